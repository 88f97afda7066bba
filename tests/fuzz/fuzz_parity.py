#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box: random mesh scenes, cameras, depths, hierarchies and
jitter modes, each frame against the CPU oracle (hit mask, primitive id and hit distance bit-exact,
RGBA8 within 1/255 on >= 99.9 % of pixels).  Prints one line per case; exits non-zero on the first
mismatch.  With FUZZ_BIG=1 the scenes are larger (up to 45 000 triangles and 1000 spheres) and the oracle
walks the octree instead of testing every primitive.  Usage: python tests/fuzz/fuzz_parity.py [cases] [seed]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (initialises the HIP runtime first)
import oracle  # noqa: E402
from raytracer_amd import hiptrace, host  # noqa: E402



def run(cases=40, seed=2024, say=print):
    """Runs the sweep; raises AssertionError at the first mismatch."""
    rng = np.random.default_rng(seed)
    BIG = os.environ.get("FUZZ_BIG") == "1"
    worst = 0
    for case in range(cases):
        w, h = int(rng.integers(40, 200)), int(rng.integers(30, 120))
        qx, qz = int(rng.integers(2, 90)), int(rng.integers(2, 60))
        lights, spheres = int(rng.integers(0, 12)), int(rng.integers(0, 30))
        if BIG:
            qx, qz, spheres = int(rng.integers(60, 150)), int(rng.integers(60, 150)), int(rng.integers(100, 1001))
        if int(rng.integers(0, 6)) == 0:
            lights = int(rng.integers(30, 71))      # more than 32 lights: several shadow passes per level
        depth = int(rng.integers(1, 6))
        accel = int(rng.integers(0, 2))
        jitter = int(rng.integers(0, 2))
        # one case in three runs the general kernel: device shading model, sub-pixel grid, aperture
        general = int(rng.integers(0, 3)) == 0
        shading = int(rng.integers(0, 2)) if general else 0
        rpp = int(rng.choice([1, 2, 4, 6, 9])) if general else 1
        aperture = float(rng.choice([0.0, 0.05])) if general and shading == 1 and jitter == 1 else 0.0
        # one case in four is rendered as the ranks of a split frame and re-assembled
        world = int(rng.choice([2, 3, 8])) if int(rng.integers(0, 4)) == 0 else 1
        scene = host.lib.rt_scene_synth_mesh(w, h, qx, qz, 0, lights, spheres)
        # one case in three gets awkward triangles: zero area, collinear, exact duplicates (the smaller id must
        # win a tie), specks, sheets across the room, and every length scaled by a random factor
        factor = 1.0
        if int(rng.integers(0, 3)) == 0:
            from raytracer_amd.types import Triangle, Vec3
            sc = scene.contents
            def add(a, b, c, material):
                host.lib.rt_scene_addTriangle(scene, Triangle(material, Vec3(*a), Vec3(*b), Vec3(*c)))
            for _ in range(int(rng.integers(1, 12))):
                kind = int(rng.integers(0, 5))
                p = rng.uniform(-20, 20, 3); p[1] = rng.uniform(0.2, 8)
                material = int(rng.integers(1, sc.materialCount))
                if kind == 0:
                    add(p, p, p + rng.uniform(-1, 1, 3), material)
                elif kind == 1:
                    d = rng.uniform(-1, 1, 3)
                    add(p, p + d, p + 2 * d, material)
                elif kind == 2 and sc.triangleCount:
                    t = sc.triangles[int(rng.integers(0, sc.triangleCount))]
                    add((t.v0.x, t.v0.y, t.v0.z), (t.v1.x, t.v1.y, t.v1.z), (t.v2.x, t.v2.y, t.v2.z), material)
                elif kind == 3:
                    add(p, p + rng.uniform(-1e-5, 1e-5, 3), p + rng.uniform(-1e-5, 1e-5, 3), material)
                else:
                    add(p, p + rng.uniform(-40, 40, 3), p + rng.uniform(-40, 40, 3), material)
            if int(rng.integers(0, 2)) == 0:
                factor = float(rng.choice([0.02, 0.3, 7.0, 300.0]))
                for i in range(sc.triangleCount):
                    for v in (sc.triangles[i].v0, sc.triangles[i].v1, sc.triangles[i].v2):
                        v.x *= factor; v.y *= factor; v.z *= factor
                for i in range(sc.sphereCount):
                    q = sc.spheres[i].position
                    q.x *= factor; q.y *= factor; q.z *= factor
                    sc.spheres[i].radius *= factor
                for i in range(sc.planeCount):
                    sc.planes[i].distanceFromOrigin *= factor
                for i in range(sc.pointLightCount):
                    q = sc.pointLights[i].position
                    q.x *= factor; q.y *= factor; q.z *= factor
                    sc.pointLights[i].strength *= factor * factor
        cam = scene.contents.camera.contents
        cam.position.x = float(rng.uniform(-45, 45)); cam.position.y = float(rng.uniform(0.5, 25)); cam.position.z = float(rng.uniform(-45, 45))
        cam.lookAt.x = float(rng.uniform(-10, 10)); cam.lookAt.y = float(rng.uniform(0, 5)); cam.lookAt.z = float(rng.uniform(-10, 10))
        cam.FOV = float(rng.uniform(30, 120))
        for v in (cam.position, cam.lookAt):
            v.x *= factor; v.y *= factor; v.z *= factor
        host.lib.rt_camera_setup(scene.contents.camera)
        camera = host.camera_copy(scene.contents.camera)
        camera.apertureSize = aperture
        tree = host.lib.rt_octree_buildFromScene(scene)
        cpu = oracle.render(scene, tree if BIG else None, camera, depth, jitter=jitter, shading=shading, rays_per_pixel=rpp)
        if world == 1:
            tr = hiptrace.Tracer(w, h, depth, accel=accel, jitter=jitter, shading=shading, rays_per_pixel=rpp)
            tr.upload_scene(scene, tree)
            gpu = tr.frame(camera)
            # the second frame launches its tiles longest first (by the first frame's costs): same pixels
            again = tr.frame(camera)
            if not (np.array_equal(again["rgba8"], gpu["rgba8"]) and again["rays"] == gpu["rays"]):
                say(f"case {case}: the second frame differs from the first")
                raise AssertionError("mismatch: see the line printed above")
        else:
            # every rank renders its tiles into its slice of the gathered buffer; rank 0 re-assembles the frame
            ranks = [hiptrace.Tracer(w, h, depth, accel=accel, jitter=jitter, shading=shading, rays_per_pixel=rpp, rank=r, world_size=world)
                     for r in range(world)]
            nbytes = ranks[0].output_bytes()
            gathered = torch.zeros(world * nbytes, dtype=torch.uint8, device="cuda")
            frame = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
            for r, one in enumerate(ranks):
                one.upload_scene(scene, tree)
                one.frame_device(camera, gathered.data_ptr() + r * nbytes)
                one.last_frame_stats()
            ranks[0].untile(gathered.data_ptr(), frame.data_ptr())
            ranks[0].last_frame_stats()
            torch.cuda.synchronize()
            gpu = {"rgba8": frame.cpu().numpy().view(np.uint32).reshape(h, w), "hitMask": cpu["hitMask"], "primitiveId": cpu["primitiveId"],
                   "hitDistance": cpu["hitDistance"]}      # the split path returns pixels only
            for one in ranks:
                one.close()
        label = (f"case {case}: {w}x{h} quads {qx}x{qz} lights {lights} spheres {spheres} depth {depth} accel {accel} jitter {jitter} "
                 f"shading {shading} rays/pixel {rpp} aperture {aperture} ranks {world} triangles {scene.contents.triangleCount} scale {factor}")
        ok = (np.array_equal(gpu["hitMask"], cpu["hitMask"]) and np.array_equal(gpu["primitiveId"], cpu["primitiveId"]) and
              np.array_equal(gpu["hitDistance"].view(np.uint32), cpu["hitDistance"].view(np.uint32)))
        a = gpu["rgba8"].view(np.uint8).reshape(h, w, 4).astype(np.int16)
        b = cpu["rgba8"].view(np.uint8).reshape(h, w, 4).astype(np.int16)
        err = np.abs(a - b).max(axis=-1)
        frac = float((err <= 1).mean())
        worst = max(worst, int(err.max()))
        say(f"{label}: ids {'ok' if ok else 'DIFFER'}, within 1/255: {frac:.5f}, max {int(err.max())}/255")
        if world == 1:
            tr.close()
        host.lib.rt_octree_destroy(tree)
        host.lib.rt_scene_destroy(scene)
        if not ok or frac < 0.999:
            raise AssertionError("mismatch: see the line printed above")
    say(f"{cases} cases passed, largest colour error {worst}/255")



if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 2024,
        lambda *a: print(*a, flush=True))
