"""CPU, two processes over gloo: the multi-GPU frame assembly (interleaved tiles, gather to
rank 0, untile) with the CPU oracle standing in for the per-rank renderer."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import tiles


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def worker(rank, world, port, width, height, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from raytracer_amd import host
    scene = host.lib.rt_scene_init(width, height)
    cam = host.camera_copy(scene.contents.camera)
    local = tiles.local_tile_count(width, height, world)
    mine = np.zeros((local, tiles.TILE, tiles.TILE), np.uint32)
    for slot, tx, ty in tiles.tiles_of_rank(width, height, rank, world):
        x0, y0 = tx * tiles.TILE, ty * tiles.TILE
        nx, ny = min(tiles.TILE, width - x0), min(tiles.TILE, height - y0)
        out = oracle.render(scene, None, cam, 2, window=oracle.Window(x0, y0, 1, 1, nx, ny), threads=1, want=("rgba8",))
        mine[slot, :ny, :nx] = out["rgba8"]
    send = torch.from_numpy(mine.view(np.int32).reshape(-1))
    chunks = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send, chunks, dst=0)
    if rank == 0:
        gathered = torch.cat(chunks).numpy().view(np.uint32)
        frame = tiles.untile(gathered, width, height, world)
        full = oracle.render(scene, None, cam, 2, threads=2, want=("rgba8",))["rgba8"]
        np.save(result_path, np.array([int(np.array_equal(frame, full)), frame.shape[0], frame.shape[1]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("size", [(64, 48), (70, 37)])   # tilesX multiple of 2 (rotated), and ragged edges
def test_two_rank_frame_equals_single_frame(tmp_path, size):
    width, height = size
    result = str(tmp_path / "ok.npy")
    mp.spawn(worker, args=(2, free_port(), width, height, result), nprocs=2, join=True)
    ok, h, w = np.load(result)
    assert (h, w) == (height, width)
    assert ok == 1


def pipeline_worker(rank, world, port, frames, result_path):
    """FrameExchange over gloo: frame f of rank r is a buffer filled with 1000*f + r."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 4096
    mine = [torch.zeros(n, dtype=torch.int32) for _ in range(2)]
    chunks = [[torch.zeros(n, dtype=torch.int32) for _ in range(world)] for _ in range(2)] if rank == 0 else None
    seen = []
    ex = tiles.FrameExchange(rank, world, mine, chunks,
                             lambda slot: seen.append([int(c[0]) for c in chunks[slot]] + [int(c[-1]) for c in chunks[slot]]))
    for f in range(frames):
        slot = ex.acquire()
        mine[slot].fill_(1000 * f + rank)
        ex.submit(slot)
        assert ex.frames == f + 1
    ex.drain()
    assert ex.inflight == [None, None]
    with pytest.raises(RuntimeError):
        ex.submit(1 - (ex.frames & 1))
    if rank == 0:
        np.save(result_path, np.array(seen))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("frames", [1, 2, 5])
def test_frame_exchange_delivers_every_frame_in_order(tmp_path, frames):
    result = str(tmp_path / "seen.npy")
    mp.spawn(pipeline_worker, args=(2, free_port(), frames, result), nprocs=2, join=True)
    seen = np.load(result)
    assert seen.shape == (frames, 4)
    for f in range(frames):
        assert list(seen[f]) == [1000 * f, 1000 * f + 1, 1000 * f, 1000 * f + 1]


def test_partition_matches_c_abi():
    from raytracer_amd import hiptrace
    for tiles_x, tiles_y, world in ((240, 135, 8), (5, 3, 2), (9, 7, 4)):
        for ty in range(tiles_y):
            for tx in range(tiles_x):
                assert tiles.tile_owner(tx, ty, tiles_x, world) == hiptrace.tile_owner(tx, ty, tiles_x, world)


def test_untile_roundtrip():
    rng = np.random.default_rng(3)
    for (w, h, world) in ((64, 48, 2), (70, 37, 3), (128, 32, 8)):
        frame = rng.integers(0, 2**32, (h, w), dtype=np.uint64).astype(np.uint32)
        local = tiles.local_tile_count(w, h, world)
        g = np.zeros((world, local, tiles.TILE, tiles.TILE), np.uint32)
        for r in range(world):
            for slot, tx, ty in tiles.tiles_of_rank(w, h, r, world):
                blk = frame[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16]
                g[r, slot, :blk.shape[0], :blk.shape[1]] = blk
        assert np.array_equal(tiles.untile(g.reshape(-1), w, h, world), frame)
