#!/bin/bash
# GPU box helper: the 10M-triangle orbit workload with the light grids off and at the given densities.
set -u
for setting in "$@"; do
    if [ $setting = off ]; then export RT_LIGHT_GRID=0; else export RT_LIGHT_GRID=1 RT_LIGHT_GRID_DENSITY=$setting; fi
    timeout -k 10 400 python -u bench.py --workload s2_10Mtri_1ksph_64lights_3bounces_4k_orbit --steps 20 --warmup 1 --no-walk-stats > gpurun_out/grid_s2_$setting.log 2>&1 || { tail -30 gpurun_out/grid_s2_$setting.log; exit 1; }
    tail -1 gpurun_out/grid_s2_$setting.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('S2 $setting', round(d['ms_per_step'],2), 'ms', round(d['value'],1), d['config']['scene_device_bytes'], d['host_seconds'])"
done
