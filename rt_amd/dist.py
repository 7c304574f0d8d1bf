"""Multi-GPU plumbing: one process per GPU, image tiles
interleaved over the ranks, scene replicated, ONE exchange step per frame —
the gather of the compact tile buffers to rank 0 (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).  Each rank sends
its buffer straight to the root (point-to-point links, no ring)."""
import numpy as np

from .api import compact_pixels


def padded_pixels(width, height, tile_size, world):
    """Every rank's compact buffer is padded to rank 0's size (rank 0 owns the
    most tiles), so the gather moves equal-sized messages."""
    return compact_pixels(width, height, tile_size, 0, world)


def gather_compact(local, rank, world, dst=0):
    """local: torch tensor [padded_pixels, C] on this rank's device.  Returns
    the concatenation [world*padded_pixels, C] on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return local
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                      device=local.device) if rank == dst else None
    dist.gather(local, list(out.chunk(world)) if rank == dst else None, dst=dst)
    return out


def tile_index_map(width, height, tile_size, world):
    """For every image pixel (row-major, row 0 = bottom) its index in the
    gathered buffer — the same mapping untile_kernel applies on the GPU."""
    ts = tile_size
    tiles_x = (width + ts - 1) // ts
    stride = padded_pixels(width, height, ts, world)
    y, x = np.mgrid[0:height, 0:width]
    tile = (y // ts) * tiles_x + (x // ts)
    rank, j = tile % world, tile // world
    return (rank * stride + j * ts * ts + (y % ts) * ts + (x % ts)).astype(np.int64)


def untile_host(gathered, width, height, tile_size, world):
    """Host (numpy) de-interleave of a gathered buffer; index shuffle only."""
    g = np.asarray(gathered)
    idx = tile_index_map(width, height, tile_size, world)
    return g.reshape((g.shape[0],) + g.shape[1:])[idx]


def compact_from_image(image, width, height, tile_size, rank, world):
    """Inverse of untile for one rank: pick this rank's tiles out of an image
    (used by the CPU tests to stand in for a rank's compact render output)."""
    ts = tile_size
    img = np.asarray(image)
    n = padded_pixels(width, height, ts, world)
    out = np.zeros((n,) + img.shape[2:], img.dtype)
    idx = tile_index_map(width, height, ts, world)
    stride = n
    mine = (idx // stride) == rank
    out[(idx[mine] - rank * stride)] = img[mine]
    return out
