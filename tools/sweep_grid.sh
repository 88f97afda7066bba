#!/bin/bash
# GPU box helper: parity tests with the light grids on, then the bench with the grids off and at several densities.
set -u
mkdir -p gpurun_out
[ -n "${SKIP_TESTS:-}" ] || { timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_abi.py tests/test_kernel_model.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/grid.tests.log 2>&1 || { tail -40 gpurun_out/grid.tests.log; exit 1; }; }
[ -n "${SKIP_TESTS:-}" ] || tail -2 gpurun_out/grid.tests.log
for setting in ${SETTINGS:-off 1 4 16 64}; do
    if [ $setting = off ]; then export RT_LIGHT_GRID=0; else export RT_LIGHT_GRID=1 RT_LIGHT_GRID_DENSITY=$setting; fi
    timeout -k 10 300 python -u bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-both-hierarchies --no-walk-stats > gpurun_out/grid_$setting.bench.log 2>&1 || { tail -30 gpurun_out/grid_$setting.bench.log; exit 1; }
    tail -1 gpurun_out/grid_$setting.bench.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('BENCH $setting', d['value'], d['unit'], d['ms_per_step'], 'ms', d['config'].get('scene_device_bytes'), d.get('host_seconds'))"
done
