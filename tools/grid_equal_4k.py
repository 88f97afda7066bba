#!/usr/bin/env python3
"""GPU box helper: the 1M-triangle scene at 3840x2160, ray trees of height 3, rendered with and without the light grids
over both hierarchies; all four frames must be the same bytes.  Usage: python tools/grid_equal_4k.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import hiptrace, host  # noqa: E402

w, h, depth = 3840, 2160, 3
scene = host.lib.rt_scene_synth_s1(w, h)
tree = host.lib.rt_octree_buildFromScene(scene)
cam = host.camera_copy(scene.contents.camera)
frames = {}
for accel in (0, 1):
    for setting in (hiptrace.LIGHT_GRID_AUTO, hiptrace.LIGHT_GRID_OFF):
        tr = hiptrace.Tracer(w, h, depth, accel=accel, light_grid=setting)
        tr.upload_scene(scene, tree)
        frames[(accel, setting)] = tr.frame(cam)
        print(f"accel {accel} lightGrid {setting}: kernel {frames[(accel, setting)]['kernel_ms']:.3f} ms, grids for {tr.scene_info().lightGridLights} lights", flush=True)
        tr.close()
first = frames[(0, hiptrace.LIGHT_GRID_OFF)]
bad = 0
for key, f in frames.items():
    for k in ("rgba8", "primitiveId", "hitMask", "hitDistance"):
        differing = int((f[k].view(np.uint8) != first[k].view(np.uint8)).sum())
        if differing:
            print(f"{key}: {k} differs in {differing} bytes")
            bad += 1
    if f["rays"] != first["rays"]:
        print(f"{key}: ray counters differ: {f['rays']} vs {first['rays']}")
        bad += 1
print("all four frames identical" if not bad else "FRAMES DIFFER")
sys.exit(1 if bad else 0)
