#!/usr/bin/env python3
"""Randomised sweep on the GPU box for occlusion by planes: small mesh scenes with extra planes at random
orientations (normals of length 0.3 to 3 among them) that cut through the room, and lights within a few hundredths
of a plane, on either side of it and exactly on it.  Every frame is held against the CPU oracle's brute force: hit
mask, primitive id and hit distance bit-exact, RGBA8 within 1/255 on >= 99.9 % of pixels and never off by more than 2.
Usage: python tests/fuzz/fuzz_planes.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (initialises the HIP runtime first)
import oracle  # noqa: E402
from raytracer_amd import hiptrace, host  # noqa: E402
from raytracer_amd.types import Plane, Vec3  # noqa: E402



def run(cases=100, seed=31, say=print):
    """Runs the sweep; raises AssertionError at the first mismatch."""
    rng = np.random.default_rng(seed)
    for case in range(cases):
        w, h = int(rng.integers(60, 140)), int(rng.integers(40, 90))
        qx, qz = int(rng.integers(2, 40)), int(rng.integers(2, 30))
        lights, spheres = int(rng.integers(1, 10)), int(rng.integers(0, 12))
        depth = int(rng.integers(1, 5))
        scene = host.lib.rt_scene_synth_mesh(w, h, qx, qz, 0, lights, spheres)
        sc = scene.contents
        extra = int(rng.integers(0, 5))
        if int(rng.integers(0, 8)) == 0:
            extra = int(rng.integers(12, 30))           # more planes than the side masks cover
        for _ in range(extra):
            n = rng.normal(size=3)
            if int(rng.integers(0, 3)) == 0:
                n = np.eye(3)[int(rng.integers(0, 3))] * float(rng.choice([-1.0, 1.0]))
            n = n / np.linalg.norm(n) * float(rng.choice([1.0, 1.0, 0.3, 0.6, 1.9, 3.0]))
            host.lib.rt_scene_addPlane(scene, Plane(int(rng.integers(1, sc.materialCount)), Vec3(*n), float(rng.uniform(-30, 30))))
        sc = scene.contents
        for i in range(sc.pointLightCount):
            if int(rng.integers(0, 3)) == 0 or sc.planeCount == 0:
                continue
            # next to a plane n . x + d = 0: at signed distance `off` from it, above a random point of the room
            pl = sc.planes[int(rng.integers(0, sc.planeCount))]
            n = np.array([pl.normal.x, pl.normal.y, pl.normal.z], dtype=np.float64)
            length = np.linalg.norm(n)
            x = rng.uniform(-25, 25, 3); x[1] = rng.uniform(0.5, 20)
            x = x - n * ((n @ x + pl.distanceFromOrigin) / (length * length))      # foot of x on the plane
            off = float(rng.choice([0.0, 1e-4, -1e-4, 5e-3, -5e-3, 9e-3, 1.1e-2, -1.1e-2, 3e-2, -3e-2, 0.3, -0.3]))
            x = x + n / length * off
            q = sc.pointLights[i].position
            q.x, q.y, q.z = float(x[0]), float(x[1]), float(x[2])
        cam = sc.camera.contents
        cam.position.x = float(rng.uniform(-45, 45)); cam.position.y = float(rng.uniform(0.5, 25)); cam.position.z = float(rng.uniform(-45, 45))
        cam.lookAt.x = float(rng.uniform(-10, 10)); cam.lookAt.y = float(rng.uniform(0, 5)); cam.lookAt.z = float(rng.uniform(-10, 10))
        cam.FOV = float(rng.uniform(30, 120))
        host.lib.rt_camera_setup(sc.camera)
        camera = host.camera_copy(sc.camera)
        cpu = oracle.render(scene, None, camera, depth)
        tr = hiptrace.Tracer(w, h, depth, light_grid=int(rng.integers(0, 2)))
        tr.upload_scene(scene, None)
        gpu = tr.frame(camera)
        tr.close()
        ids = (np.array_equal(gpu["hitMask"], cpu["hitMask"]) and np.array_equal(gpu["primitiveId"], cpu["primitiveId"]) and
               np.array_equal(gpu["hitDistance"].view(np.uint32), cpu["hitDistance"].view(np.uint32)))
        a = gpu["rgba8"].view(np.uint8).reshape(h, w, 4).astype(np.int16)
        b = cpu["rgba8"].view(np.uint8).reshape(h, w, 4).astype(np.int16)
        err = np.abs(a - b).max(axis=-1)
        frac = float((err <= 1).mean())
        say(f"case {case}: {w}x{h} quads {qx}x{qz} lights {lights} spheres {spheres} planes {sc.planeCount} depth {depth}: "
              f"ids {'ok' if ids else 'DIFFER'}, within 1/255 {frac:.5f}, max {int(err.max())}/255")
        host.lib.rt_scene_destroy(scene)
        if not ids or frac < 0.999 or err.max() > 2:
            raise AssertionError("mismatch: see the line printed above")
    say(f"{cases} cases passed")



if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 31,
        lambda *a: print(*a, flush=True))
