#!/usr/bin/env python3
"""Randomised sweep on the GPU box for the per-light direction grids: random mesh scenes with lights in awkward
places (just above the mesh, next to a triangle corner, inside the cloud of spheres, far outside), rendered with
the grids and without them.  The two frames must be identical byte for byte, ray counters included: the grids only
change where the candidates of an occlusion query come from.  Every tenth case is also held against the CPU
oracle.  Usage: python tests/fuzz/fuzz_grid.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (initialises the HIP runtime first)
import oracle  # noqa: E402
from raytracer_amd import hiptrace, host  # noqa: E402
from raytracer_amd.types import Triangle, Vec3  # noqa: E402



def run(cases=100, seed=77, say=print):
    """Runs the sweep; raises AssertionError at the first mismatch."""
    rng = np.random.default_rng(seed)
    used = 0
    for case in range(cases):
        w, h = int(rng.integers(60, 260)), int(rng.integers(40, 150))
        qx, qz = int(rng.integers(2, 200)), int(rng.integers(2, 120))
        lights, spheres = int(rng.integers(1, 12)), int(rng.integers(0, 60))
        if int(rng.integers(0, 6)) == 0:
            lights = int(rng.integers(30, 71))
        depth = int(rng.integers(1, 6))
        scene = host.lib.rt_scene_synth_mesh(w, h, qx, qz, 0, lights, spheres)
        sc = scene.contents
        if int(rng.integers(0, 3)) == 0:
            for _ in range(int(rng.integers(1, 8))):        # sheets across the room, specks, slivers
                p = rng.uniform(-20, 20, 3); p[1] = rng.uniform(0.2, 8)
                kind = int(rng.integers(0, 3))
                span = (40.0, 1e-5, 1.0)[kind]
                q, r = p + rng.uniform(-span, span, 3), p + rng.uniform(-span, span, 3)
                if kind == 2:
                    r = p + (q - p) * 1.0001 + rng.uniform(-1e-4, 1e-4, 3)
                host.lib.rt_scene_addTriangle(scene, Triangle(int(rng.integers(1, sc.materialCount)), Vec3(*p), Vec3(*q), Vec3(*r)))
        for i in range(sc.pointLightCount):
            where = int(rng.integers(0, 6))
            q = sc.pointLights[i].position
            if where == 0:                                      # skimming the mesh
                q.x, q.y, q.z = float(rng.uniform(-25, 25)), float(rng.uniform(2.6, 3.2)), float(rng.uniform(-25, 25))
            elif where == 1 and sc.triangleCount:               # next to a triangle's corner
                t = sc.triangles[int(rng.integers(0, sc.triangleCount))]
                d = rng.uniform(-1, 1, 3) * float(rng.choice([1e-3, 1e-2, 0.2]))
                q.x, q.y, q.z = t.v0.x + float(d[0]), t.v0.y + abs(float(d[1])), t.v0.z + float(d[2])
            elif where == 2 and sc.sphereCount:                 # next to (or inside) a sphere
                s = sc.spheres[int(rng.integers(0, sc.sphereCount))]
                d = rng.uniform(-1, 1, 3) * s.radius * float(rng.choice([0.5, 1.01, 1.5]))
                q.x, q.y, q.z = s.position.x + float(d[0]), s.position.y + float(d[1]), s.position.z + float(d[2])
            elif where == 3:                                    # far outside the room's contents
                q.x, q.y, q.z = float(rng.uniform(-48, 48)), float(rng.uniform(30, 49)), float(rng.uniform(-48, 48))
            # else: where the generator put it
        factor = 1.0
        if int(rng.integers(0, 4)) == 0:
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
        cam = sc.camera.contents
        cam.position.x = float(rng.uniform(-45, 45)); cam.position.y = float(rng.uniform(0.5, 25)); cam.position.z = float(rng.uniform(-45, 45))
        cam.lookAt.x = float(rng.uniform(-10, 10)); cam.lookAt.y = float(rng.uniform(0, 5)); cam.lookAt.z = float(rng.uniform(-10, 10))
        cam.FOV = float(rng.uniform(30, 120))
        for v in (cam.position, cam.lookAt):
            v.x *= factor; v.y *= factor; v.z *= factor
        host.lib.rt_camera_setup(sc.camera)
        camera = host.camera_copy(sc.camera)
        frames = {}
        accel = int(rng.integers(0, 2))         # the library's own hierarchy, or the caller's octree
        tree = host.lib.rt_octree_buildFromScene(scene) if accel == 1 else None
        for name, setting in (("grid", hiptrace.LIGHT_GRID_AUTO), ("walk", hiptrace.LIGHT_GRID_OFF)):
            tr = hiptrace.Tracer(w, h, depth, light_grid=setting, accel=accel)
            tr.upload_scene(scene, tree)
            frames[name] = tr.frame(camera)
            if name == "grid":
                info = tr.scene_info()
                gridLights, references = info.lightGridLights, info.lightGridReferences
            tr.close()
        used += 1 if gridLights else 0
        same = all(np.array_equal(frames["grid"][k].view(np.uint8), frames["walk"][k].view(np.uint8)) for k in ("rgba8", "primitiveId", "hitMask", "hitDistance"))
        same = same and frames["grid"]["rays"] == frames["walk"]["rays"]
        if tree is not None:
            host.lib.rt_octree_destroy(tree)
        label = (f"case {case}: {w}x{h} accel {accel} quads {qx}x{qz} lights {lights} ({gridLights} with a grid, {references} references) spheres {spheres} "
                 f"depth {depth} triangles {sc.triangleCount} scale {factor}")
        differing = int((frames["grid"]["rgba8"] != frames["walk"]["rgba8"]).sum())
        verdict = "same" if same else f"DIFFER in {differing} pixels"
        if same and case % 10 == 0:
            cpu = oracle.render(scene, None, camera, depth)
            g = frames["grid"]
            ids = (np.array_equal(g["hitMask"], cpu["hitMask"]) and np.array_equal(g["primitiveId"], cpu["primitiveId"]) and
                   np.array_equal(g["hitDistance"].view(np.uint32), cpu["hitDistance"].view(np.uint32)))
            a = g["rgba8"].view(np.uint8).reshape(h, w, 4).astype(np.int16)
            b = cpu["rgba8"].view(np.uint8).reshape(h, w, 4).astype(np.int16)
            err = np.abs(a - b).max(axis=-1)
            verdict += f"; oracle: ids {'ok' if ids else 'DIFFER'}, within 1/255 {float((err <= 1).mean()):.5f}, max {int(err.max())}/255"
            same = ids and float((err <= 1).mean()) >= 0.999
        say(f"{label}: {verdict}")
        host.lib.rt_scene_destroy(scene)
        if not same:
            raise AssertionError("mismatch: see the line printed above")
    say(f"{cases} cases passed, {used} of them with at least one light grid")



if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 77,
        lambda *a: print(*a, flush=True))
