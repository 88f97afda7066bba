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

# wall time per frame of one rank's call sequence when frames are enqueued back to back (what a rank of the
# multi-process bench does between exchanges): kernel time plus whatever the host side adds per frame
import time  # noqa: E402
for world in (1, 8):
    tr = hiptrace.Tracer(w, h, 3, rank=0, world_size=world)
    tr.upload_scene(scene, None)
    out = torch.empty(tr.output_bytes(), dtype=torch.uint8, device="cuda")
    for _ in range(5):
        tr.frame_device(cam, out.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frames = 200
    for _ in range(frames):
        tr.frame_device(cam, out.data_ptr())
    tr.last_frame_stats()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3 / frames
    print(f"world {world}: {wall:.3f} ms per frame back to back, kernel {tr.last_frame_stats()[1]:.3f} ms", flush=True)
    tr.close()
