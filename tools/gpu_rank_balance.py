"""How evenly the interleaved tiles spread the C2 frame over N ranks: every rank's share is
rendered on this one GPU in turn (weak scaling as bench.py does it: 16 spp x N) and the
kernel times compared - the slowest rank sets the step time of the N-GPU job.
usage: gpu_rank_balance.py [tile sizes...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rt_amd as ra

tiles = [int(a) for a in sys.argv[1:]] or [64]
W, H = 1920, 1080
ctx = ra.Context(0)
scene, cam = ra.Scene.builtin(0, aspect=W / H)
ctx.upload(scene)
stream = torch.cuda.current_stream()
for tile in tiles:
    for world in (1, 2, 4, 8):
        times = []
        for rank in range(world):
            st = ra.default_settings(W, H, spp=16 * world, max_depth=8, tile_size=tile, rank=rank, world_size=world,
                                     flags=ra.FLAG_COMPACT if world > 1 else 0)
            n_out = ra.compact_pixels(W, H, tile, 0, world) if world > 1 else W * H
            accum = torch.zeros((n_out, 4), dtype=torch.float32, device="cuda")
            ms = []
            for _ in range(6):
                ctx.render_device(cam, st, accum.data_ptr(), None, stream.cuda_stream)
                ms.append(ctx.kernel_ms())
            times.append(float(np.median(ms[1:])))
        t = np.array(times)
        print(f"tile {tile:3d} world {world}: kernel ms per rank min {t.min():.3f} mean {t.mean():.3f} max {t.max():.3f}"
              f"  -> max/mean {t.max() / t.mean():.3f}, against world 1: {t.max():.3f}", flush=True)
ctx.close()
