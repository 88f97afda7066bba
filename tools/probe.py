#!/usr/bin/env python3
"""Profiling aid: kernel time, ray counts and (with the RT_WALK_STATS build) walk counters for
a few variants of the synthetic mesh scene.  Usage: python tools/probe.py [WxH] [quadsX quadsZ]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import hiptrace, host  # noqa: E402


def run(tag, scene, tree, w, h, depth, reps=3):
    cam = host.camera_copy(scene.contents.camera)
    accel = int(os.environ.get("PROBE_ACCEL", "0"))   # 0: the library's own hierarchy, 1: the caller's octree
    tr = hiptrace.Tracer(w, h, depth, accel=accel)
    tr.upload_scene(scene, tree)
    best = 1e30
    for _ in range(reps):
        out = tr.frame(cam, want_aux=False, want_image=False)
        best = min(best, out["kernel_ms"])
    rays = out["rays"]
    total = sum(rays.values())
    ws = tr.walk_stats()
    line = f"{tag:34s} {w}x{h} d{depth} kernel {best:9.3f} ms  rays {total:10d}  {total / best / 1e3:8.1f} Mrays/s  {rays}"
    if ws:
        line += "  per-ray: " + " ".join(f"{k}={v / total:.2f}" for k, v in ws.items() if not k.startswith("wave"))
        line += "  per-64-rays: " + " ".join(f"{k}={v / (total / 64):.1f}" for k, v in ws.items() if k.startswith("wave"))
        hist = [v for k, v in ws.items() if k.startswith("shadowSteps")]
        if sum(hist):
            line += "  shadow walk steps histogram (log2 buckets, % of rays): " + " ".join(f"{100.0 * v / sum(hist):.1f}" for v in hist)
    print(line, flush=True)
    tr.close()


def main():
    w, h = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "960x540").split("x"))
    if os.environ.get("PROBE_ONLY") == "full":
        scene = host.lib.rt_scene_synth_s1(w, h)
        tree = host.lib.rt_octree_buildFromScene(scene)
        for depth in (1, 3):
            run("s1", scene, tree, w, h, depth, reps=2)
        return
    if os.environ.get("PROBE_ONLY") == "tiny":
        tiny = host.lib.rt_scene_synth_mesh(w, h, 2, 1, 0, 0, 0)
        run("4 triangles, no lights", tiny, None, w, h, 1, reps=3)
        return
    if os.environ.get("PROBE_ONLY") == "primary":
        scene = host.lib.rt_scene_synth_mesh(w, h, 1000, 500, 0, 0, 0)
        run("mesh, no lights", scene, None, w, h, 1, reps=5)
        run("mesh, no lights", scene, None, w, h, 3, reps=5)
        tiny = host.lib.rt_scene_synth_mesh(w, h, 2, 1, 0, 0, 0)
        run("4 triangles, no lights", tiny, None, w, h, 1, reps=5)
        return
    qx, qz = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1000, 500)
    t = time.time()
    for lights in (0, 8):
        scene = host.lib.rt_scene_synth_mesh(w, h, qx, qz, 0, lights, 0)
        tree = host.lib.rt_octree_buildFromScene(scene)
        print(f"built {qx}x{qz} lights={lights} in {time.time() - t:.1f}s nodes={tree.contents.nodeCount}", flush=True)
        for depth in ((1, 3) if lights == 0 else (1, 2, 3)):
            run(f"mesh lights={lights}", scene, tree, w, h, depth)
        host.lib.rt_octree_destroy(tree)
        host.lib.rt_scene_destroy(scene)
    scene = host.lib.rt_scene_init(w, h)
    tree = host.lib.rt_octree_buildFromScene(scene)
    run("default scene", scene, tree, w, h, 2)
    run("default scene", scene, tree, w, h, 5)


if __name__ == "__main__":
    main()
