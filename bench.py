#!/usr/bin/env python3
"""Throughput of the HIP trace path on the synthetic 1M-triangle octree scene.

A step is one frame: every rank traces the screen tiles it owns (interleaved 16x16 tiles,
replicated scene), the RGBA8 tiles are gathered to rank 0 over RCCL and re-assembled there.
At --gpus 1 the frame is written row-major directly and there is no exchange.

Prints ONE JSON line on rank 0.  `value` counts the rays the device actually traced
(primary + shadow + reflection + refraction, exact device counters) per second of wall time
over the timed steps, with scene and camera already resident in HBM.

`python bench.py --gpus N` with N > 1 and no rank environment starts the N ranks itself (a
`torch.distributed.run` child process, one rank per GPU over RCCL) and relays rank 0's line; under
`python -m torch.distributed.run ... bench.py --gpus N` it is one of the ranks.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene kind, width, height, lights, maxDepth)
    "s1_1Mtri_4k_8lights_2bounces": ("s1", 3840, 2160, 8, 3),
    "s1_1Mtri_4k_8lights_1bounce": ("s1", 3840, 2160, 8, 2),
    "default_scene_4k_1bounce": ("default", 3840, 2160, 1, 2),
    "s1_1Mtri_8k_8lights_2bounces": ("s1", 7680, 4320, 8, 3),
    "s1_1Mtri_960x540_8lights_2bounces": ("s1", 960, 540, 8, 3),
    # 10M triangles + 1000 spheres, 64 lights, 3 bounces, 60-frame orbit of the camera; no octree
    # is built for it (the reference builder needs > 10 GB for this scene), so only the
    # library's own hierarchy runs and there is no CPU leg
    "s2_10Mtri_1ksph_64lights_3bounces_4k_orbit": ("s2", 3840, 2160, 64, 4),
}

# Algorithmic cost per unit of work, counted from the reference arithmetic (one flop per
# add/sub/mul/div/sqrt) and its record sizes: box test 12 flop / 24 B (src/kernel.cl:428-472),
# plane 13 / 20 B (src/raytracer.c:70-86), sphere 27 / 20 B (:88-116), triangle 42 to the
# distance + 66 for the edge tests / 40 B + 4 B index (:118-161), node header 72 B
# (src/octree.h:15-25), one light sample shaded ~90 flop (:292-309), one pixel 4 B.
FLOP = {"boxTests": 12, "planeTests": 13, "sphereTests": 27, "triangleDistance": 42, "triangleInside": 66,
        "lightEvaluations": 90}
BYTES = {"nodesVisited": 72, "triangleTests": 44, "sphereTests": 24, "planeTests": 20}
PEAK_FP32_TFLOPS = 157.3   # MI355X vector fp32, packed FMA (datasheet)
PEAK_HBM_GBS = 8000.0      # MI355X HBM3E (datasheet)
# Bytes the device walk reads per unit of its own work (csrc/device_layout.h): a wide node is one
# 128-byte line, a filter record 64 B, an exact test a 16-byte plane + three 16-byte corners + a
# 4-byte index; a pixel is 4 B written.
DEVICE_BYTES = {"nodes_tested": 128, "triangle_filter_tests": 64, "triangle_exact_tests": 68}


def launch_ranks(args, argv):
    """The parent of `python bench.py --gpus N`, N > 1: touches no GPU (counting devices does not
    initialise one), starts torch.distributed.run as a child with the same arguments and returns its
    exit status; rank 0's JSON line goes to this process's stdout."""
    import torch
    if not args.selftest_exchange:
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs, this host has {have}", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    return subprocess.run(cmd, env=env).returncode


def selftest_exchange(args, rank, world):
    """No GPU: the rank launcher, the rendezvous and the frame exchange (interleaved tiles, gather to rank 0,
    re-assembly) over gloo, each rank filling its tiles with a function of the pixel position.  Not a
    measurement; prints one JSON line saying whether rank 0 re-assembled the frame."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from raytracer_amd import tiles
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    width, height = 200, 120
    local = tiles.local_tile_count(width, height, world)

    def pixel(frame, x, y):
        return np.uint32(0xFF000000) | (np.uint32(frame) << 20) | (y.astype(np.uint32) << 10) | x.astype(np.uint32)

    mine = [torch.zeros(local * tiles.TILE * tiles.TILE, dtype=torch.int32) for _ in range(2)]
    chunks = [[torch.zeros_like(mine[0]) for _ in range(world)] for _ in range(2)] if rank == 0 else None
    frames = []
    exchange = tiles.FrameExchange(rank, world, mine, chunks, lambda slot: frames.append(
        tiles.untile(torch.cat(chunks[slot]).numpy().view(np.uint32), width, height, world).copy()))
    for f in range(args.steps):
        slot = exchange.acquire()
        buf = mine[slot].numpy().view(np.uint32).reshape(local, tiles.TILE, tiles.TILE)
        buf[:] = 0
        for k, tx, ty in tiles.tiles_of_rank(width, height, rank, world):
            ys, xs = np.mgrid[ty * tiles.TILE:(ty + 1) * tiles.TILE, tx * tiles.TILE:(tx + 1) * tiles.TILE]
            buf[k] = pixel(f, xs, ys)
        exchange.submit(slot)
    exchange.drain()
    ok = True
    if rank == 0:
        ys, xs = np.mgrid[0:height, 0:width]
        ok = len(frames) == args.steps and all(np.array_equal(fr, pixel(f, xs, ys)) for f, fr in enumerate(frames))
        print(json.dumps({"metric": "selftest_exchange", "ok": bool(ok), "n_gpus": 0, "rccl_ranks": dist.get_world_size(),
                          "backend": dist.get_backend(), "steps": args.steps, "frames_reassembled": len(frames)}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


def build_scene(host, kind, width, height):
    if kind == "default":
        scene = host.lib.rt_scene_init(width, height)
    elif kind == "s1":
        scene = host.lib.rt_scene_synth_s1(width, height)
    elif kind == "s1_no_octree":
        scene = host.lib.rt_scene_synth_s1(width, height)
        if not scene:
            raise RuntimeError("scene construction failed")
        return scene, None
    elif kind == "s2":
        scene = host.lib.rt_scene_synth_s2(width, height)
        if not scene:
            raise RuntimeError("scene construction failed")
        return scene, None
    else:
        raise ValueError(kind)
    tree = host.lib.rt_octree_buildFromScene(scene)
    if not scene or not tree:
        raise RuntimeError("scene or octree construction failed")
    return scene, tree


def cpu_baseline(scene, tree, camera, depth, width, height, kind, budget_pixels):
    """The CPU oracle (a port of the reference algorithm) on a strided sample of the same frame.
    Octree scenes use the reference's own walk (every node the ray's box test admits, no
    pruning), the default scene its brute-force loops.  Also returns the per-pixel work counts
    that the roofline figures are computed from."""
    import oracle
    step = 1
    while (width // step) * (height // step) > budget_pixels:
        step *= 2
    window = oracle.Window(step // 2, step // 2, step, step, width // step, height // step)
    octree = tree if kind != "default" else None
    t0 = time.perf_counter()
    out = oracle.render(scene, octree, camera, depth, window=window, walk=oracle.WALK_REFERENCE,
                        box_pad=0.0, threads=0, want=("rgba8", "hitMask", "primitiveId", "hitDistance"))
    dt = time.perf_counter() - t0
    n = out["counters"]
    rays = n["primary"] + n["shadow"] + n["reflection"] + n["refraction"]
    pixels = window.countX * window.countY
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    threads = int(os.environ.get("OMP_NUM_THREADS", cores))
    scale = (width * height) / pixels
    # one thread, on a sample sixteen times sparser (about the same wall time per thread)
    single = None
    if min(threads, cores) > 1:
        wide = oracle.Window(2 * step, 2 * step, 4 * step, 4 * step, max(1, width // (4 * step)), max(1, height // (4 * step)))
        t1 = time.perf_counter()
        one = oracle.render(scene, octree, camera, depth, window=wide, walk=oracle.WALK_REFERENCE, box_pad=0.0, threads=1, want=("rgba8",))
        d1 = time.perf_counter() - t1
        m = one["counters"]
        single = {"value": (m["primary"] + m["shadow"] + m["reflection"] + m["refraction"]) / d1 / 1e6, "unit": "Mrays/s", "cores": 1,
                  "sample": f"{wide.countX * wide.countY} pixels (every {4 * step}th in x and y), {d1:.2f} s"}
    cpu_model = ""
    try:
        for text in open("/proc/cpuinfo"):
            if text.startswith("model name"):
                cpu_model = text.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    flops = sum(FLOP[k] * n[k] for k in FLOP) * scale
    bytes_ = (sum(BYTES[k] * n[k] for k in BYTES)) * scale + 4.0 * width * height
    return {
        "value": rays / dt / 1e6, "unit": "Mrays/s", "cores": min(threads, cores), "kind": "port",
        "sample": f"{pixels} pixels (every {step}th in x and y) of the same frame, {rays} rays, {dt:.2f} s, "
                  f"reference walk" if kind != "default" else
                  f"{pixels} pixels (every {step}th in x and y) of the same frame, {rays} rays, {dt:.2f} s, brute force",
        "frames_per_s": 1.0 / (dt * scale), "cpu_model": cpu_model, "single_thread": single,
    }, flops, bytes_, rays * scale, (out, window)


def parity_gate(gpu, cpu, window):
    """The frame the device rendered against the oracle's frame on the oracle's pixels: hit mask, primitive id and
    hit distance must be the same bits, colour within 1/255 on at least 99.9 % of the pixels."""
    import numpy as np
    ys = slice(window.y0, window.y0 + window.stepY * window.countY, window.stepY)
    xs = slice(window.x0, window.x0 + window.stepX * window.countX, window.stepX)
    pick = {k: gpu[k][ys, xs] for k in ("rgba8", "hitMask", "primitiveId", "hitDistance")}
    a = np.ascontiguousarray(pick["rgba8"]).view(np.uint8).reshape(-1, 4).astype(np.int16)
    b = np.ascontiguousarray(cpu["rgba8"]).view(np.uint8).reshape(-1, 4).astype(np.int16)
    worst = np.abs(a - b).max(axis=1)
    pixels = int(worst.size)
    return {"pixels": pixels,
            "hit_mask_mismatches": int(np.count_nonzero(pick["hitMask"] != cpu["hitMask"])),
            "id_mismatches": int(np.count_nonzero(pick["primitiveId"] != cpu["primitiveId"])),
            "distance_bit_mismatches": int(np.count_nonzero(np.ascontiguousarray(pick["hitDistance"]).view(np.uint32) !=
                                                            np.ascontiguousarray(cpu["hitDistance"]).view(np.uint32))),
            "rgb_within_1_frac": float(np.count_nonzero(worst <= 1)) / pixels, "rgb_max_err": int(worst.max()),
            "against": "CPU oracle, the same frame and pixels as cpu_baseline (untimed device frame with side outputs)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="s1_1Mtri_4k_8lights_2bounces", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-sample-pixels", type=int, default=9000000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--both-hierarchies", action=argparse.BooleanOptionalAction, default=True,
                    help="also time a few frames with the other hierarchy and report them as other_hierarchy")
    ap.add_argument("--jitter", default="zero", choices=("zero", "hashed"),
                    help="zero: hard shadows (the measured default); hashed: the reference's light jitter, deterministic")
    ap.add_argument("--accel", default="bvh", choices=("octree", "bvh"),
                    help="hierarchy the device walks: the caller's octree topology, or the library's own 8-wide BVH")
    ap.add_argument("--no-walk-stats", action="store_true",
                    help="skip the untimed frame of the profiling build (no roofline figure); for runs under a profiler")
    ap.add_argument("--selftest-exchange", action="store_true",
                    help="no GPU, no measurement: run the rank launcher and the tile exchange over gloo and say whether the frame re-assembled")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if args.selftest_exchange:
        raise SystemExit(selftest_exchange(args, rank, world))

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the trace path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from raytracer_amd import hiptrace, host

    kind, width, height, _lights, depth = WORKLOADS[args.workload]
    if world > 1:
        # several ranks: each measures the library's default path and nothing else -- no second hierarchy, and (unless
        # the caller's octree is the hierarchy asked for) no octree at all, which every rank would otherwise build on
        # its share of the host's cores while the others wait
        args.both_hierarchies = False
        if args.accel == "bvh" and kind == "s1":
            kind = "s1_no_octree"
    t_build = time.perf_counter()
    scene, tree = build_scene(host, kind, width, height)
    camera = host.camera_copy(scene.contents.camera)
    t_build = time.perf_counter() - t_build
    if tree is None:
        if args.accel != "bvh":
            raise SystemExit(f"{args.workload} has no octree: use --accel bvh")
        args.no_cpu_baseline, args.both_hierarchies = True, False
    # the stress scene is rendered along a 60-frame orbit; the others from the scene's own camera
    cameras = [camera]
    if kind == "s2":
        cameras = []
        for f in range(60):
            c = host.camera_copy(scene.contents.camera)
            host.lib.rt_camera_orbit(C.byref(c), f, 60)
            cameras.append(c)

    stream = torch.cuda.current_stream()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def accel_of(name):
        return hiptrace.ACCEL_PRIVATE_BVH if name == "bvh" else hiptrace.ACCEL_CALLER_OCTREE

    def measure(accel_name, warmup, steps):
        """Uploads the scene for one hierarchy and times `steps` frames of the whole exchange path."""
        accel = accel_of(accel_name)
        tracer = hiptrace.Tracer(width, height, depth, device=local_rank, rank=rank, world_size=world, accel=accel,
                                 jitter=hiptrace.JITTER_HASHED if args.jitter == "hashed" else hiptrace.JITTER_ZERO)
        tracer.set_stream(stream.cuda_stream)
        t_upload = time.perf_counter()
        tracer.upload_scene(scene, tree)
        t_upload = time.perf_counter() - t_upload
        info = tracer.scene_info()
        out_bytes = tracer.output_bytes()
        # Two frames in flight at N > 1: while frame i is traced, the tiles of frame i-1 travel to
        # rank 0 on RCCL's own stream and are re-assembled there.  Buffers alternate; a buffer is
        # reused only after the gather that reads it has been waited for (a stream-level wait).
        mine = [torch.empty(out_bytes, dtype=torch.uint8, device="cuda") for _ in range(2 if world > 1 else 1)]
        gathered = frame = chunks = None
        if world > 1 and rank == 0:
            gathered = [torch.empty(world * out_bytes, dtype=torch.uint8, device="cuda") for _ in range(2)]
            chunks = [list(g.split(out_bytes)) for g in gathered]
            frame = torch.empty(width * height * 4, dtype=torch.uint8, device="cuda")
        exchange = None
        if world > 1:
            from raytracer_amd.tiles import FrameExchange
            exchange = FrameExchange(rank, world, mine, chunks,
                                     lambda slot: tracer.untile(gathered[slot].data_ptr(), frame.data_ptr()))

        def run(count, starts=None, stops=None):
            for i in range(count):
                slot = exchange.acquire() if exchange else 0
                if starts:
                    starts[i].record(stream)
                tracer.frame_device(cameras[i % len(cameras)], mine[slot].data_ptr())
                if stops:
                    stops[i].record(stream)
                if exchange:
                    exchange.submit(slot)
            if exchange:
                exchange.drain()

        run(warmup)
        fence()
        starts = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        stops = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        t0 = time.perf_counter()
        run(steps, starts, stops)
        fence()
        elapsed = time.perf_counter() - t0

        rays, _ = tracer.last_frame_stats()
        rays_sum = float(sum(rays.values())) * steps
        if len(cameras) > 1:
            # the frames differ along a camera path: count their rays in an untimed replay
            # (frames are deterministic, so the counts are those of the timed frames)
            rays_sum = 0.0
            for i in range(steps):
                tracer.frame_device(cameras[i % len(cameras)], mine[0].data_ptr())
                rays_sum += float(sum(tracer.last_frame_stats()[0].values()))
        kernel_ms = sum(s.elapsed_time(e) for s, e in zip(starts, stops)) / steps
        stats = torch.tensor([elapsed, kernel_ms, rays_sum], dtype=torch.float64, device="cuda")
        mx, sm = stats.clone(), stats.clone()
        per_rank = [kernel_ms]
        if world > 1:
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            every = [torch.zeros(1, dtype=torch.float64, device="cuda") for _ in range(world)]
            dist.all_gather(every, stats[1:2].contiguous())
            per_rank = [float(v[0]) for v in every]
        result = {"elapsed": float(mx[0]), "kernel_ms_max": float(mx[1]), "kernel_ms_mean": float(sm[1]) / world,
                  "kernel_ms_per_rank": per_rank,
                  "rays_total": float(sm[2]), "rays": rays, "t_upload": t_upload,
                  "wide_nodes": int(info.wideNodeCount), "tree_levels": int(info.treeLevels),
                  # the scene's own device copy, and beside it the per-light direction grids (if the upload built any)
                  "device_bytes": int(info.deviceBytes) - int(info.lightGridBytes),
                  "light_grid_bytes": int(info.lightGridBytes), "light_grid_lights": int(info.lightGridLights),
                  "light_grid_build_peak_bytes": int(info.lightGridPeakMiB) << 20}
        tracer.close()
        del mine, gathered, frame, chunks
        return result

    m = measure(args.accel, args.warmup, args.steps)
    elapsed, kernel_ms_max, kernel_ms_mean, rays_total, rays = (m["elapsed"], m["kernel_ms_max"], m["kernel_ms_mean"],
                                                                m["rays_total"], m["rays"])
    other = None
    if args.both_hierarchies and kind != "default":
        other_name = "octree" if args.accel == "bvh" else "bvh"
        other_steps = max(2, min(args.steps, 5))
        o = measure(other_name, 1, other_steps)
        other = {"hierarchy": other_name, "value": o["rays_total"] / o["elapsed"] / 1e6, "unit": "Mrays/s",
                 "ms_per_step": o["elapsed"] * 1e3 / other_steps, "steps": other_steps,
                 "kernel_ms_max_over_ranks": o["kernel_ms_max"], "wide_nodes": o["wide_nodes"],
                 "tree_levels": o["tree_levels"], "scene_device_bytes": o["device_bytes"], "light_grid_bytes": o["light_grid_bytes"],
                 "flatten_and_upload_s": o["t_upload"]}

    # What the device walk did on one frame, from the build with walk counters (one untimed frame per
    # rank, never part of the timing above): the work the roofline figure is computed from.
    walk = None
    if args.no_walk_stats:
        pass
    elif os.path.exists(hiptrace.STATS_LIB_PATH):
        counted = hiptrace.Tracer(width, height, depth, device=local_rank, rank=rank, world_size=world, accel=accel_of(args.accel),
                                  jitter=hiptrace.JITTER_HASHED if args.jitter == "hashed" else hiptrace.JITTER_ZERO,
                                  library=hiptrace.load_library(hiptrace.STATS_LIB_PATH))
        counted.upload_scene(scene, tree)
        counted.frame_device(cameras[0])
        counted_rays, _ = counted.last_frame_stats()
        ws = counted.walk_stats()
        counted.close()
        keys = sorted(ws)
        totals = torch.tensor([float(ws[k]) for k in keys] + [float(counted_rays["shadow"]), float(sum(counted_rays.values()))],
                              dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(totals, op=dist.ReduceOp.SUM)
        walk = {k: int(v) for k, v in zip(keys, totals[:len(keys)].tolist())}
        walk["shadowRays"], walk["rays"] = int(totals[-2]), int(totals[-1])
    elif rank == 0:
        print("bench.py: libhiptrace_stats.so is missing (make -C raytracer_amd/csrc stats): no roofline figure", file=sys.stderr)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = rays_total / elapsed / 1e6
        per_rank = m["kernel_ms_per_rank"]
        line = {
            "metric": "Mrays/s (primary+shadow+bounce) at 4K" if height == 2160 else "Mrays/s (primary+shadow+bounce)",
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "frames_per_s": 1e3 / ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "backend": dist.get_backend() if world > 1 else None,
            "config": {"workload": args.workload, "width": width, "height": height, "max_depth": depth,
                       "triangles": int(scene.contents.triangleCount), "spheres": int(scene.contents.sphereCount),
                       "planes": int(scene.contents.planeCount), "lights": int(scene.contents.pointLightCount),
                       "octree_nodes": int(tree.contents.nodeCount) if tree else None,
                       "octree_indexes": int(tree.contents.indexCount) if tree else None, "camera_path_frames": len(cameras),
                       "hierarchy": args.accel, "jitter": args.jitter, "wide_nodes": m["wide_nodes"], "tree_levels": m["tree_levels"],
                       "scene_device_bytes": m["device_bytes"], "light_grid_bytes": m["light_grid_bytes"], "light_grid_lights": m["light_grid_lights"],
                       "light_grid_build_peak_bytes": m["light_grid_build_peak_bytes"],
                       "partition": f"interleaved 16x16 tiles over {world} rank(s)" + (", RCCL gather to rank 0 overlapped with the next frame" if world > 1 else "")},
            "rays_per_frame": rays_total / args.steps, "rays_by_kind": {k: int(v) for k, v in rays.items()} if world == 1 else None,
            "kernel_ms": {"per_rank": per_rank, "max_over_ranks": kernel_ms_max, "mean_over_ranks": kernel_ms_mean,
                          "max_over_mean": kernel_ms_max / kernel_ms_mean if kernel_ms_mean else None},
            "host_seconds": {"scene_and_octree_build": t_build, "flatten_and_upload": m["t_upload"]},
        }
        if other:
            line["other_hierarchy"] = other

        # Counter evidence kept from separate rocprofv3 --pmc passes over this workload (counters cannot
        # share a run with the timed region): HBM bytes per launch and how busy the vector ALUs were.
        pmc = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[args.workload][args.accel]
        except (OSError, KeyError, ValueError):
            pass

        if walk:
            # dominant (only) kernel: trace_kernel.  Algorithmic work = the per-unit prices above applied to
            # what THIS kernel's walk did on the frame (a filter test is priced as the distance part of a
            # triangle test, an exact test as a whole one, box tests are the non-empty children of the
            # nodes tested), over the slowest rank's average launch duration.
            boxes = walk.get("childBoxesTested") or 8 * walk["wideNodesTested"]
            planes, lights = int(scene.contents.planeCount), max(1, int(scene.contents.pointLightCount))
            plane_flop = FLOP["planeTests"] * planes * walk["rays"]
            if args.jitter == "zero":
                # hard shadows: a shadow ray skips the planes its point and its light are clearly on one side of; priced
                # as the side test of the point (7 flop per plane, once per point) and no plane test at all, which is the
                # least the kernel can have done
                plane_flop = FLOP["planeTests"] * planes * (walk["rays"] - walk["shadowRays"]) + 7 * planes * walk["shadowRays"] // lights
            flop = (FLOP["boxTests"] * boxes + FLOP["triangleDistance"] * walk["triangleDistanceTests"] +
                    (FLOP["triangleDistance"] + FLOP["triangleInside"]) * walk["triangleEdgeTests"] +
                    plane_flop +
                    FLOP["lightEvaluations"] * walk["shadowRays"])
            bytes_ = (DEVICE_BYTES["nodes_tested"] * walk["wideNodesTested"] + DEVICE_BYTES["triangle_filter_tests"] * walk["triangleDistanceTests"] +
                      DEVICE_BYTES["triangle_exact_tests"] * walk["triangleEdgeTests"] + 4.0 * width * height)
            t = kernel_ms_max * 1e-3
            achieved = flop / world / t / 1e12
            traffic = None
            valu = None
            if pmc:
                # per launch of one GPU rendering the whole frame; split evenly over ranks at N > 1
                traffic = (2 * pmc["fetch_counter_bytes"] + pmc["write_bytes"]) / world
                valu = pmc.get("valu")
            hbm_frac = (traffic / t / 1e9 / PEAK_HBM_GBS) if traffic else None
            valu_busy = valu.get("busy_frac") if valu else None
            # which resource the counters say the kernel is closest to: vector-ALU issue or HBM
            bound = "fp32_valu" if (valu_busy is None or hbm_frac is None or valu_busy >= hbm_frac) else "hbm"
            line["roofline"] = {
                "kernel": "trace_kernel", "bound": bound, "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_TFLOPS, "traffic": traffic, "traffic_source": pmc["source"] if pmc else None,
                "algorithmic_flop_per_frame": flop, "algorithmic_bytes_per_frame": bytes_,
                "valu_issue": valu,
                # HBM: the bytes a frame cannot avoid moving -- the resident scene once and the pixels out -- over the launch
                # time; what the counters saw moving is `traffic`.  The records the walk reads (algorithmic_bytes_per_frame)
                # are served by the vector L1 and L2 almost entirely and are reported as such, not against HBM.
                "hbm": {"achieved": (m["device_bytes"] + 4.0 * width * height / world) / t / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": (m["device_bytes"] + 4.0 * width * height / world) / t / 1e9 / PEAK_HBM_GBS,
                        "compulsory_bytes_per_frame": m["device_bytes"] + 4.0 * width * height / world,
                        "measured_traffic_frac": hbm_frac},
                "walk_reads_through_l1": {"bytes_per_frame": bytes_, "GB/s": bytes_ / world / t / 1e9},
                "note": "work = what this kernel's own walk did on the frame (device counters of the profiling build, untimed frame)",
            }
            line["actual_walk"] = {"nodes_tested": walk["wideNodesTested"], "box_tests": boxes,
                                   "children_per_node": boxes / walk["wideNodesTested"] if walk["wideNodesTested"] else None,
                                   "leaves_opened": walk["leavesOpened"], "triangle_filter_tests": walk["triangleDistanceTests"],
                                   "triangle_exact_tests": walk["triangleEdgeTests"], "wave_node_steps": walk.get("waveNodeSteps"),
                                   "wave_leaf_phases": walk.get("waveLeafPhases"), "per_ray": {
                                       "nodes": walk["wideNodesTested"] / walk["rays"], "filter_tests": walk["triangleDistanceTests"] / walk["rays"],
                                       "exact_tests": walk["triangleEdgeTests"] / walk["rays"]}}
            cycles = {k: walk.get(k, 0) for k in ("cyclesClosestHit", "cyclesShadowPass", "cyclesShading", "cyclesBarrier")}
            if sum(cycles.values()) > 0:
                whole = float(sum(cycles.values()))
                line["phase_split"] = {"unit": "fraction of wavefront cycles (profiling build)",
                                       **{k[6:].lower(): v / whole for k, v in cycles.items()},
                                       "ms": {k[6:].lower(): v / whole * kernel_ms_max for k, v in cycles.items()}}

        if not args.no_cpu_baseline and world == 1:
            # the CPU figure is reported at N = 1 only
            base, ref_flops, ref_bytes, cpu_rays_frame, (cpu_frame, cpu_window) = cpu_baseline(scene, tree, camera, depth, width, height, kind,
                                                                                                args.cpu_sample_pixels)
            line["cpu_baseline"] = base
            # the parity gate that goes with the number: one more untimed frame of the measured configuration, with the
            # side outputs, held against the frame the CPU leg has just computed
            if args.jitter == "zero":
                checked = hiptrace.Tracer(width, height, depth, device=local_rank, accel=accel_of(args.accel))
                checked.upload_scene(scene, tree)
                line["parity"] = parity_gate(checked.frame(camera), cpu_frame, cpu_window)
                checked.close()
            # how much less the device walk does than the reference's unordered, unpruned octree walk
            # (kernel.cl:552-601; oracle counters on the sample, scaled to the frame) -- not a hardware figure
            eff = {"reference_walk_flop_per_frame": ref_flops, "reference_walk_bytes_per_frame": ref_bytes,
                   "reference_walk_rays_per_frame": cpu_rays_frame,
                   "reference_walk_tflops_equivalent": ref_flops / (kernel_ms_max * 1e-3) / 1e12}
            if "roofline" in line:
                eff["device_walk_flop_per_frame"] = line["roofline"]["algorithmic_flop_per_frame"]
                eff["ratio"] = ref_flops / line["roofline"]["algorithmic_flop_per_frame"]
            line["algorithmic_efficiency"] = eff
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
