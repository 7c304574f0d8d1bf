"""Tile load balance of the multi-GPU split, measured on ONE GPU: every rank's share of
the weak-scaling frame (S0 1080p, 16*N spp, 64x64 tiles interleaved) rendered in turn.
The N-GPU step time is the slowest rank's; efficiency bound = mean / max."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rt_amd as ra
ctx = ra.Context(0)
W, H = 1920, 1080
scene, cam = ra.Scene.builtin(0, aspect=W / H)
ctx.upload(scene)
for world in (1, 2, 4, 8):
    ms, rays = [], []
    for rank in range(world):
        n = ra.compact_pixels(W, H, 64, rank, world)
        acc = torch.zeros((max(n, 1), 4), dtype=torch.float32, device="cuda")
        st = ra.default_settings(W, H, spp=16 * world, max_depth=8, rank=rank, world_size=world,
                                 flags=ra.FLAG_COMPACT if world > 1 else 0)
        if world == 1:
            acc = torch.zeros((H * W, 4), dtype=torch.float32, device="cuda")
        stats = ra.Stats()
        ctx.render_device(cam, st, acc.data_ptr(), None, None, stats)
        t = []
        for _ in range(3):
            ctx.render_device(cam, st, acc.data_ptr(), None, None)
            t.append(ctx.kernel_ms())
        ms.append(float(np.median(t))); rays.append(stats.rays)
    ms = np.array(ms)
    print(f"N={world}: kernel ms per rank {np.round(ms, 3).tolist()}  max {ms.max():.3f}  mean/max {ms.mean()/ms.max():.3f}  "
          f"whole-job {sum(rays)/ms.max()/1e3:.0f} Mrays/s if ranks ran side by side", flush=True)
ctx.close()
