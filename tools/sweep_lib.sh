#!/bin/bash
# GPU box helper: bench.py against several builds of libhiptrace (same ABI).
# Usage: bash tools/sweep_lib.sh lib1.so lib2.so ...
for lib in "$@"; do
  RT_HIPTRACE_LIB=$PWD/raytracer_amd/lib/$lib timeout -k 10 300 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-both-hierarchies --no-walk-stats > gpurun_out/sweep_$lib.log 2>&1 || { tail -5 gpurun_out/sweep_$lib.log; exit 1; }
  tail -1 gpurun_out/sweep_$lib.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],2), 'ms')"
done
