#!/bin/bash
# GPU box helper: parity tests, then the walk-statistics probe, then a short bench.  Each GPU step
# has its own time limit and the chain stops at the first failure.
set -u
TAG=${1:-x}
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_abi.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/check_$TAG.tests.log 2>&1 || { tail -30 gpurun_out/check_$TAG.tests.log; exit 1; }
tail -2 gpurun_out/check_$TAG.tests.log
export PROBE_ACCEL=${PROBE_ACCEL:-0}
ACCEL=$([ "$PROBE_ACCEL" = 0 ] && echo bvh || echo octree)
RT_HIPTRACE_LIB=$PWD/raytracer_amd/lib/libhiptrace_stats.so timeout -k 10 300 python -u tools/probe.py 960x540 > gpurun_out/check_$TAG.probe.log 2>&1 || { tail -30 gpurun_out/check_$TAG.probe.log; exit 1; }
cut -c1-320 gpurun_out/check_$TAG.probe.log
timeout -k 10 400 python -u bench.py --accel $ACCEL --steps 5 --warmup 1 --no-cpu-baseline --no-both-hierarchies > gpurun_out/check_$TAG.bench.log 2>&1 || { tail -30 gpurun_out/check_$TAG.bench.log; exit 1; }
tail -1 gpurun_out/check_$TAG.bench.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('BENCH', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', d['rays_per_frame'], d['config']['scene_device_bytes'], d['host_seconds'])"
