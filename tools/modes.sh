#!/bin/bash
# GPU box helper: short bench runs of the configurations whose shadow rays walk the hierarchy, one line each.
# Usage: bash tools/modes.sh <tag> [s1walk] [jitter] [s2] [octree] [s1]
TAG=$1; shift
mkdir -p gpurun_out
one() { # name, env..., -- bench args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout -k 10 400 python -u bench.py --no-cpu-baseline --no-both-hierarchies --no-walk-stats "$@" > gpurun_out/modes_${TAG}_$name.log 2>&1 || { tail -5 gpurun_out/modes_${TAG}_$name.log; exit 1; }
  tail -1 gpurun_out/modes_${TAG}_$name.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],3), 'ms', round(d['value'],1), 'Mrays/s')"
}
for what in "$@"; do
  case $what in
    s1) one s1 X=1 -- --steps 10 --warmup 2 ;;
    s1walk) one s1walk RT_LIGHT_GRID=0 -- --steps 10 --warmup 2 ;;
    s1walk_old) one s1walk_old RT_LIGHT_GRID=0 RT_SHARED_WALK=0 -- --steps 10 --warmup 2 ;;
    jitter) one jitter X=1 -- --steps 10 --warmup 2 --jitter hashed ;;
    octree) one octree X=1 -- --steps 5 --warmup 1 --accel octree ;;
    octreewalk) one octreewalk RT_LIGHT_GRID=0 -- --steps 5 --warmup 1 --accel octree ;;
    s2) one s2 X=1 -- --steps 10 --warmup 1 --workload s2_10Mtri_1ksph_64lights_3bounces_4k_orbit ;;
    s2_old) one s2_old RT_SHARED_WALK=0 -- --steps 10 --warmup 1 --workload s2_10Mtri_1ksph_64lights_3bounces_4k_orbit ;;
    8k) one 8k X=1 -- --steps 5 --warmup 1 --workload s1_1Mtri_8k_8lights_2bounces ;;
    default) one default X=1 -- --steps 10 --warmup 2 --workload default_scene_4k_1bounce ;;
  esac
done
