#!/usr/bin/env python3
"""Profiling aid: kernel time of single ranks of a W-way split of the 4K benchmark frame, all on one
GPU (no exchange), to see how much of the ideal 1/W the trace kernel keeps.
Usage: python tools/probe_ranks.py [WxH]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (initialises the HIP runtime first)
from raytracer_amd import hiptrace, host  # noqa: E402

w, h = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "3840x2160").split("x"))
scene = host.lib.rt_scene_synth_s1(w, h)
cam = host.camera_copy(scene.contents.camera)
base = None
for world in (1, 2, 4, 8):
    times = []
    for rank in sorted({0, world // 2, world - 1}):
        tr = hiptrace.Tracer(w, h, 3, rank=rank, world_size=world)
        tr.upload_scene(scene, None)
        best = 1e9
        for _ in range(4):
            tr.frame_device(cam)
            best = min(best, tr.last_frame_stats()[1])
        times.append((rank, best))
        tr.close()
    worst = max(t for _, t in times)
    base = base or worst
    print(f"world {world}: " + ", ".join(f"rank {r} {t:.3f} ms" for r, t in times) +
          f"  -> speed-up {base / worst:.2f} of {world}", flush=True)
