"""Tile size of the work-item order at N = 1 (the image does not depend on it): kernel time
of the C2 frame and of the C3-class frame for tile sizes 8 ... 256.
usage: gpu_tile_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rt_amd as ra

W, H = 1920, 1080
ctx = ra.Context(0)
stream = torch.cuda.current_stream()
accum = torch.zeros((W * H, 4), dtype=torch.float32, device="cuda")
for name, kind, param, spp in (("C2", 0, 0, 16), ("C3-class", 1, 8, 8)):
    scene, cam = ra.Scene.builtin(kind, param=param, aspect=W / H)
    ctx.upload(scene)
    for rep in range(2):
        out = []
        for tile in (8, 16, 32, 64, 128, 256):
            st = ra.default_settings(W, H, spp=spp, max_depth=8, tile_size=tile)
            ms = []
            for _ in range(12):
                ctx.render_device(cam, st, accum.data_ptr(), None, stream.cuda_stream)
                ms.append(ctx.kernel_ms())
            out.append(f"{tile}: {np.min(ms[2:]):.3f}/{np.median(ms[2:]):.3f}")
        print(f"{name} tile: min/median kernel ms  " + "  ".join(out), flush=True)
ctx.close()
