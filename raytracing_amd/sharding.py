"""Tile ownership for multi-GPU frames (pure index math, no GPU needed).

The frame is cut into `tile` x `tile` squares numbered row-major (the reference walks 48x48 buckets,
src/sdl.cpp:253-276; pixels are independent, so any partition renders the same image).  Rank r of
`world` owns tiles t with t % world == r and stores them compacted in increasing t; every rank's
block is padded to the same number of tiles so one equal-size gather moves the framebuffer.
"""
import numpy as np


def tile_grid(width, height, tile):
    return (width + tile - 1) // tile, (height + tile - 1) // tile


def tiles_owned(width, height, tile, rank, world):
    tx, ty = tile_grid(width, height, tile)
    n = tx * ty
    return (n - rank + world - 1) // world if n > rank else 0


def tiles_per_rank_padded(width, height, tile, world):
    return tiles_owned(width, height, tile, 0, world)


def owned_tile_rects(width, height, tile, rank, world):
    """[(x0, y0, x1, y1)] of the tiles a rank owns, in storage order (clipped to the frame)."""
    tx, ty = tile_grid(width, height, tile)
    out = []
    for t in range(rank, tx * ty, world):
        x0, y0 = (t % tx) * tile, (t // tx) * tile
        out.append((x0, y0, min(x0 + tile, width), min(y0 + tile, height)))
    return out


def untile(gathered, width, height, tile, world):
    """CPU mirror of qd_untile_async: gathered[world, tiles_padded, tile, tile, 3] -> image[H, W, 3]."""
    gathered = np.asarray(gathered)
    tpr = gathered.shape[1]
    img = np.zeros((height, width, 3), gathered.dtype)
    for r in range(world):
        for k, (x0, y0, x1, y1) in enumerate(owned_tile_rects(width, height, tile, r, world)):
            assert k < tpr
            img[y0:y1, x0:x1] = gathered[r, k, : y1 - y0, : x1 - x0]
    return img


def aa_slice(n_listed, rank, world):
    """Slice of the AA pass's pixel list that `rank` traces: (pixels per rank, first pixel, count).  Blocks of equal size
    in rank order, the tail ranks possibly short or empty, so that the gathered taps are in list order."""
    per = (n_listed + world - 1) // world
    first = min(n_listed, rank * per)
    return per, first, min(n_listed, first + per) - first


class SlotSchedule:
    """Which slot (HIP stream, buffers, RCCL communicator) the n-th submitted frame uses: n % F.  Every rank walks the same
    sequence, so the gathers -- collectives -- are issued in the same order everywhere; and a slot's gathers only ever
    run on that slot's stream and communicator, i.e. strictly one after the other: no two operations of one communicator
    are ever in flight together (RCCL, like NCCL, wants the operations of ONE communicator issued in one order on every
    rank and not concurrently from several streams; different communicators may overlap).  `log` keeps what was issued,
    for tests: (frame number, slot)."""

    def __init__(self, frames_in_flight):
        if frames_in_flight < 1:
            raise ValueError("frames_in_flight must be >= 1")
        self.F = frames_in_flight
        self.submitted = 0
        self.log = []

    def next(self):
        k = self.submitted % self.F
        self.log = self.log[-255:] + [(self.submitted, k)]
        self.submitted += 1
        return k
