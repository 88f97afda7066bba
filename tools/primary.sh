#!/bin/bash
# GPU box helper: the primary-ray-only frame (no lights, depth 1) of the 1M-triangle mesh at 4K, with and without the shared walk.
mkdir -p gpurun_out
for lib in "$@"; do
for sw in 1 0; do
  RT_HIPTRACE_LIB=$PWD/raytracer_amd/lib/$lib RT_SHARED_WALK=$sw PROBE_ONLY=primary timeout -k 10 200 python -u tools/probe.py 3840x2160 > gpurun_out/primary_${lib}_$sw.log 2>&1 || { tail -5 gpurun_out/primary_${lib}_$sw.log; exit 1; }
  echo "$lib shared=$sw $(cut -c1-150 gpurun_out/primary_${lib}_$sw.log | tail -2)"
done
done
