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
