#!/bin/bash
# GPU box helper: the 10M-triangle orbit workload (20 frames) against several builds of libhiptrace.
for lib in "$@"; do RT_HIPTRACE_LIB=$PWD/raytracer_amd/lib/$lib timeout -k 10 300 python -u bench.py --workload s2_10Mtri_1ksph_64lights_3bounces_4k_orbit --steps 20 --warmup 1 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib s2', round(d['ms_per_step'],2), 'ms', round(d['value'],1), d['config']['wide_nodes'], d['config']['tree_levels'])"; done
