#!/bin/bash
# Host-only sanitizer pass (AddressSanitizer + UndefinedBehaviorSanitizer, no GPU involved): the plain-C
# host modules (scene, synthetic scenes, threaded octree build, OBJ ingestion, bitmap writer) and the C++
# flattening / hierarchy check, each driven by a small program.  The sanitizer runtime is linked statically.
# Usage: bash tools/sanitize/run.sh        (from the repository root; prints the drivers' output, fails on a report)
set -eu
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$(mktemp -d)
SAN="-O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -static-libasan -fno-omit-frame-pointer -fopenmp -pthread"
for f in rt_scene rt_synth rt_octree rt_object rt_image; do
  gcc -std=gnu11 $SAN -I"$ROOT/include" -I"$ROOT/raytracer_amd/host" -c "$ROOT/raytracer_amd/host/$f.c" -o "$OUT/$f.o"
done
gcc -std=gnu11 $SAN -I"$ROOT/include" -o "$OUT/host_driver" "$ROOT/tools/sanitize/host_driver.c" "$OUT"/rt_*.o -lm
g++ -std=c++17 $SAN -I"$ROOT/include" -I"$ROOT/raytracer_amd/csrc" -o "$OUT/flatten_driver" "$ROOT/tools/sanitize/flatten_driver.cpp" \
    "$ROOT/raytracer_amd/csrc/flatten.cpp" "$OUT"/rt_*.o -lm
printf 'f 1 2 3\n' > "$OUT/bad.obj"; printf 'v 1 2' > "$OUT/truncated.obj"; printf 'f' > "$OUT/f.obj"
(cd "$OUT" && ASAN_OPTIONS=detect_leaks=1 ./host_driver "$ROOT/tests/golden/cube_quads.obj" "$ROOT/tests/golden/quirks.obj" bad.obj truncated.obj f.obj missing.obj)
ASAN_OPTIONS=detect_leaks=1 "$OUT/flatten_driver"
rm -rf "$OUT"
echo "sanitizer pass clean"
