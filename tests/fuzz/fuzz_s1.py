#!/usr/bin/env python3
"""Randomised parity sweep on the million-triangle benchmark scene: random cameras, both hierarchies,
each frame against the CPU oracle's octree walk.  Usage: python tests/fuzz/fuzz_s1.py [cameras] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import oracle  # noqa: E402
from raytracer_amd import hiptrace, host  # noqa: E402



def run(count=8, seed=5, say=print):
    """Runs the sweep; raises AssertionError at the first mismatch."""
    rng = np.random.default_rng(seed)
    w, h, depth = 320, 180, 3
    scene = host.lib.rt_scene_synth_s1(w, h)
    tree = host.lib.rt_octree_buildFromScene(scene)
    tracers = []
    for accel in (0, 1):
        tr = hiptrace.Tracer(w, h, depth, accel=accel)
        tr.upload_scene(scene, tree)
        tracers.append(tr)
    worst = 0
    for case in range(count):
        cam = scene.contents.camera.contents
        cam.position.x = float(rng.uniform(-45, 45)); cam.position.y = float(rng.uniform(0.3, 30)); cam.position.z = float(rng.uniform(-45, 45))
        cam.lookAt.x = float(rng.uniform(-15, 15)); cam.lookAt.y = float(rng.uniform(0, 4)); cam.lookAt.z = float(rng.uniform(-15, 15))
        cam.FOV = float(rng.uniform(25, 120))
        host.lib.rt_camera_setup(scene.contents.camera)
        camera = host.camera_copy(scene.contents.camera)
        cpu = oracle.render(scene, tree, camera, depth, walk=oracle.WALK_PRUNED)
        for accel, tr in enumerate(tracers):
            gpu = tr.frame(camera)
            ok = (np.array_equal(gpu["hitMask"], cpu["hitMask"]) and np.array_equal(gpu["primitiveId"], cpu["primitiveId"]) and
                  np.array_equal(gpu["hitDistance"].view(np.uint32), cpu["hitDistance"].view(np.uint32)))
            a = gpu["rgba8"].view(np.uint8).reshape(h, w, 4).astype(np.int16)
            b = cpu["rgba8"].view(np.uint8).reshape(h, w, 4).astype(np.int16)
            err = np.abs(a - b).max(axis=-1)
            worst = max(worst, int(err.max()))
            say(f"camera {case} accel {accel}: ids {'ok' if ok else 'DIFFER'}, max colour error {int(err.max())}/255, rays {sum(gpu['rays'].values())}")
            if not ok or err.max() > 2:
                raise AssertionError("mismatch: see the line printed above")
    say(f"{count} cameras passed on both hierarchies, largest colour error {worst}/255")



if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 5,
        lambda *a: print(*a, flush=True))
