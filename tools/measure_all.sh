#!/bin/bash
# GPU box helper: the table of workloads in one go (one JSON line each under gpurun_out/).
# Usage: bash tools/measure_all.sh
set -u
run() { # tag, bench args...
  local tag=$1; shift
  timeout -k 10 400 python -u bench.py --no-cpu-baseline --no-walk-stats "$@" > gpurun_out/all_$tag.log 2>&1 || { echo "$tag failed"; tail -5 gpurun_out/all_$tag.log; exit 1; }
  tail -1 gpurun_out/all_$tag.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],2), 'ms', round(d['value']/1e3,2), 'Grays/s', int(d['rays_per_frame']), 'rays/frame', 'upload', round(d['host_seconds']['flatten_and_upload'],2), 's')"
}
run s1_4k --steps 10 --warmup 2 --no-both-hierarchies &&
run s1_4k_octree --steps 10 --warmup 2 --no-both-hierarchies --accel octree &&
RT_LIGHT_GRID=0 run s1_4k_walk --steps 10 --warmup 2 --no-both-hierarchies &&
RT_LIGHT_GRID=0 run s1_4k_octree_walk --steps 10 --warmup 2 --no-both-hierarchies --accel octree &&
run s1_4k_1bounce --steps 10 --warmup 2 --no-both-hierarchies --workload s1_1Mtri_4k_8lights_1bounce &&
run s1_4k_jitter --steps 10 --warmup 2 --no-both-hierarchies --jitter hashed &&
run s1_8k --steps 5 --warmup 1 --no-both-hierarchies --workload s1_1Mtri_8k_8lights_2bounces &&
run default_4k --steps 10 --warmup 2 --workload default_scene_4k_1bounce &&
run s2_orbit --steps 60 --warmup 2 --workload s2_10Mtri_1ksph_64lights_3bounces_4k_orbit
