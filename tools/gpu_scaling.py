"""Kernel time vs spp / resolution / occupancy for S0 (looks for fixed costs and tails)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rt_amd as ra
ctx = ra.Context(0)
for (w, h) in ((1920, 1080), (3840, 2160)):
    scene, cam = ra.Scene.builtin(0, aspect=w / h)
    ctx.upload(scene)
    acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    for spp in (1, 4, 16, 64):
        if w > 2000 and spp > 16: continue
        st = ra.default_settings(w, h, spp=spp, max_depth=8)
        stats = ra.Stats()
        ctx.render_device(cam, st, acc.data_ptr(), None, None, stats)
        ms = []
        for _ in range(4):
            ctx.render_device(cam, st, acc.data_ptr(), None, None)
            ms.append(ctx.kernel_ms())
        m = float(np.median(ms))
        print(f"{w}x{h} spp={spp}: {m:.3f} ms  {stats.rays/m/1e3:.0f} Mrays/s  ms/spp={m/spp:.3f}", flush=True)
ctx.close()
