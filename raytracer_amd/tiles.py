"""Host-side mirror of the screen-tile partition used by the multi-GPU path.

The frame is cut into 16x16-pixel tiles numbered row-major (k = ty * tilesX + tx).  Tile k
belongs to rank (k + s*ty) % world, s = 1 when tilesX is a multiple of world (otherwise
ownership would be constant down a column), and is that rank's local tile k // world.  Each
rank renders its tiles into a tile-major buffer of local_tile_count() * 256 pixels; rank 0
receives the concatenation in rank order and untile() restores the row-major frame.  The same
arithmetic runs in the kernels (hip_trace.hip: tile_of_block, untile_kernel).
"""
import numpy as np

TILE = 16


def tile_grid(width, height):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def rotate_flag(tiles_x, world):
    return 1 if world > 1 and tiles_x % world == 0 else 0


def tile_owner(tx, ty, tiles_x, world):
    if world <= 1:
        return 0
    return (ty * tiles_x + tx + rotate_flag(tiles_x, world) * ty) % world


def local_tile_count(width, height, world):
    tiles_x, tiles_y = tile_grid(width, height)
    return (tiles_x * tiles_y + world - 1) // world


def tiles_of_rank(width, height, rank, world):
    """[(local index, tx, ty)] of the tiles `rank` renders, in local order."""
    tiles_x, tiles_y = tile_grid(width, height)
    out = []
    for ty in range(tiles_y):
        for tx in range(tiles_x):
            if tile_owner(tx, ty, tiles_x, world) == rank:
                out.append(((ty * tiles_x + tx) // world, tx, ty))
    return sorted(out)


def untile(gathered, width, height, world):
    """gathered: uint32 array of world * local_tile_count * 256 pixels (rank-major)."""
    tiles_x, tiles_y = tile_grid(width, height)
    local = local_tile_count(width, height, world)
    g = np.asarray(gathered, np.uint32).reshape(world, local, TILE, TILE)
    frame = np.zeros((tiles_y * TILE, tiles_x * TILE), np.uint32)
    for ty in range(tiles_y):
        for tx in range(tiles_x):
            k = ty * tiles_x + tx
            frame[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE] = g[tile_owner(tx, ty, tiles_x, world), k // world]
    return frame[:height, :width]


class FrameExchange:
    """Keeps two frames in flight at world > 1: while frame i is being traced, the tiles of
    frame i-1 travel to rank 0 (torch.distributed.gather, asynchronous: on RCCL's own stream for
    device tensors) and are re-assembled there.

    mine:   two send buffers (this rank's tile-major output, alternating per frame)
    chunks: on rank 0, two lists of `world` receive buffers; None elsewhere
    untile: callable(slot), run on rank 0 once the gather into chunks[slot] has been waited for

    Per frame:  slot = acquire();  <enqueue the trace into mine[slot]>;  submit(slot).
    A buffer is handed out again only after the gather that reads it has been waited for."""

    def __init__(self, rank, world, mine, chunks, untile):
        if len(mine) != 2 or (rank == 0 and (chunks is None or len(chunks) != 2)):
            raise ValueError("FrameExchange needs two send buffers and, on rank 0, two receive lists")
        self.rank, self.world, self.mine, self.chunks, self.untile = rank, world, mine, chunks, untile
        self.inflight = [None, None]
        self.frames = 0

    def _finish(self, slot):
        work = self.inflight[slot]
        if work is None:
            return
        work.wait()
        self.inflight[slot] = None
        if self.rank == 0:
            self.untile(slot)

    def acquire(self):
        slot = self.frames & 1
        self._finish(slot)          # frame (frames - 2) used this slot; normally long finished
        return slot

    def submit(self, slot):
        import torch.distributed as dist
        if slot != (self.frames & 1) or self.inflight[slot] is not None:
            raise RuntimeError("submit() must follow acquire() of the same slot")
        self.inflight[slot] = dist.gather(self.mine[slot], self.chunks[slot] if self.rank == 0 else None, dst=0,
                                          async_op=True)
        self.frames += 1
        self._finish(slot ^ 1)      # the previous frame: its gather overlapped this frame's trace

    def drain(self):
        self._finish(self.frames & 1)
        self._finish((self.frames & 1) ^ 1)
