#!/bin/bash
# GPU box helper: one short bench run with the walk counters of the profiling build; prints phase split and walk figures.
# Usage: bash tools/phase.sh <tag> [bench args...]
TAG=${1:-x}; shift || true
mkdir -p gpurun_out
timeout -k 10 400 python -u bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-both-hierarchies "$@" > gpurun_out/phase_$TAG.log 2>&1 || { tail -20 gpurun_out/phase_$TAG.log; exit 1; }
tail -1 gpurun_out/phase_$TAG.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms', round(d['ms_per_step'],3), 'Mrays/s', round(d['value'],1))
print('phase ms', {k: round(v,3) for k,v in d.get('phase_split',{}).get('ms',{}).items()})
w=d.get('actual_walk',{})
print('walk', {k: (round(v,2) if isinstance(v,float) else v) for k,v in w.items() if k!='per_ray'}, w.get('per_ray'))
print('roofline frac', d.get('roofline',{}).get('frac'))
"
