#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's default workload on the GPU box:
# one kernel-trace/stats pass and separate PMC passes (counters never share a pass with a trace).
# Usage (from the repo root on the box): bash tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-both-hierarchies --no-walk-stats $*"
echo "python3 bench.py $ARGS${RT_LIGHT_GRID:+   (RT_LIGHT_GRID=$RT_LIGHT_GRID)}" > "$OUT/args.txt"
run() { # name, rocprof options...
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 "$REPO/bench.py" $ARGS > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc"
  return $rc
}
run trace --kernel-trace --stats || exit 1
run pmc_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY || exit 1
run pmc_sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE || exit 1
# live lanes per vector instruction: thread-cycles over instruction-cycles of the vector ALU
run pmc_lanes --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE || exit 1
run pmc_fetch --pmc FETCH_SIZE || exit 1
run pmc_write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
run pmc_tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum || true
find "$OUT" -name "*.csv" | head -50
