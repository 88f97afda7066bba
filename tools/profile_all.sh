#!/bin/bash
# GPU box helper: tools/profile.sh (kernel trace + counter passes) for one named configuration, then its summary.
# Usage: bash tools/profile_all.sh <name> ; names: s1 s1_octree s1_walk s1_jitter s1_8k default s2
set -u
NAME=$1
case $NAME in
  s1) ARGS=""; TITLE="S1 4K depth 3, private BVH + light grids" ;;
  s1_octree) ARGS="--accel octree"; TITLE="S1 4K depth 3, caller's octree + light grids" ;;
  s1_walk) export RT_LIGHT_GRID=0; ARGS=""; TITLE="S1 4K depth 3, private BVH, shadow rays walk (RT_LIGHT_GRID=0)" ;;
  s1_jitter) ARGS="--jitter hashed"; TITLE="S1 4K depth 3, hashed light jitter (soft shadows), private BVH" ;;
  s1_8k) ARGS="--workload s1_1Mtri_8k_8lights_2bounces"; TITLE="S1 8K depth 3, private BVH + light grids" ;;
  default) ARGS="--workload default_scene_4k_1bounce"; TITLE="default scene 4K depth 2" ;;
  s2) ARGS="--workload s2_10Mtri_1ksph_64lights_3bounces_4k_orbit"; TITLE="S2 4K: 10M triangles + 1000 spheres, 64 lights, depth 4, first 6 frames of the orbit" ;;
  *) echo "unknown configuration $NAME"; exit 2 ;;
esac
bash tools/profile.sh r3_$NAME $ARGS > gpurun_out/prof_r3_$NAME.log 2>&1 || { tail -20 gpurun_out/prof_r3_$NAME.log; exit 1; }
python3 tools/summarize_profile.py gpurun_out/prof_r3_$NAME "$TITLE" > gpurun_out/r3_$NAME.md
cp gpurun_out/prof_r3_$NAME/trace/*/*_kernel_stats.csv gpurun_out/r3_${NAME}_kernel_stats.csv
tail -14 gpurun_out/r3_$NAME.md | head -12
