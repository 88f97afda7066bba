#!/bin/bash
# GPU box helper: rocprofv3 kernel-trace summaries (no counters) of the larger benchmark configurations.
# Usage (from the repo root on the box): bash tools/profile_configs.sh <tag>
set -u
TAG=${1:-r2}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/prof_configs_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for w in s1_1Mtri_8k_8lights_2bounces s2_10Mtri_1ksph_64lights_3bounces_4k_orbit default_scene_4k_1bounce; do
  timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$w" -- python3 "$REPO/bench.py" --workload $w --steps 10 --warmup 1 --no-cpu-baseline --no-both-hierarchies --no-walk-stats > "$OUT/$w.log" 2>&1 || { echo "$w failed"; tail -5 "$OUT/$w.log"; exit 1; }
  echo "== $w"; tail -1 "$OUT/$w.log" | cut -c1-200
  cat "$OUT/$w"/*/*_kernel_stats.csv | head -4
done
