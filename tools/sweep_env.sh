#!/bin/bash
# GPU box helper: bench.py under several values of one tuning environment variable.
# Usage: bash tools/sweep_env.sh VAR v1 v2 ... [-- bench args]
VAR=$1; shift
VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done; [ "${1:-}" == "--" ] && shift
for v in "${VALS[@]}"; do
  env $VAR=$v timeout -k 10 300 python -u bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-both-hierarchies "$@" > gpurun_out/sweep_${VAR}_$v.log 2>&1 || { tail -5 gpurun_out/sweep_${VAR}_$v.log; exit 1; }
  tail -1 gpurun_out/sweep_${VAR}_$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],2), 'ms')"
done
