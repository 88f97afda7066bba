#!/usr/bin/env python3
"""Profiling aid: every rank of a W-way split of the 4K benchmark frame on one GPU -- kernel time (best of four frames)
and rays traced, to tell an uneven share of the work from a ragged end of the launch.
Usage: python tools/probe_rank_balance.py [W]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
from raytracer_amd import hiptrace, host  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w, h = 3840, 2160
scene = host.lib.rt_scene_synth_s1(w, h)
cam = host.camera_copy(scene.contents.camera)
for rank in range(world):
    tr = hiptrace.Tracer(w, h, 3, rank=rank, world_size=world)
    tr.upload_scene(scene, None)
    times = []
    for _ in range(5):
        tr.frame_device(cam)
        rays, ms = tr.last_frame_stats()
        times.append(ms)
    print(f"rank {rank}: first frame {times[0]:.3f} ms, then {min(times[1:]):.3f}..{max(times[1:]):.3f} ms, rays {sum(rays.values())} "
          f"(reflection {rays['reflection']})", flush=True)
    tr.close()
