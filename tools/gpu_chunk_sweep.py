"""Kernel time of the S0 frame against the chunk size (samples per work item), at a
given resolution and spp.  usage: gpu_chunk_sweep.py [w h spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rt_amd as ra
w, h, spp = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (1920, 1080, 64)))
ctx = ra.Context(0)
scene, cam = ra.Scene.builtin(0, aspect=w / h)
ctx.upload(scene)
acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
stats = ra.Stats()
ctx.render_device(cam, ra.default_settings(w, h, spp=spp, max_depth=8, flags=ra.FLAG_STATS), acc.data_ptr(), None, None, stats)
rays = stats.rays
for chunk in (1, 2, 4, 8, 16):
    if chunk > spp:
        break
    st = ra.default_settings(w, h, spp=spp, max_depth=8, chunk_samples=chunk)
    ms = []
    for _ in range(3):
        ctx.render_device(cam, st, acc.data_ptr(), None, None)
        ms.append(ctx.kernel_ms())
    print(f"{w}x{h} {spp} spp, chunk {chunk:2d}: kernel {min(ms):8.3f} ms -> {rays / min(ms) / 1e3:7.0f} Mrays/s", flush=True)
ctx.close()
