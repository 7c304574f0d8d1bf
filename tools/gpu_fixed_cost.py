"""Fixed cost of one launch of the trace kernel: kernel time of the C2 frame at 2 .. 128 spp
(chunks of 2 samples throughout, so the per-item work is the same) and a straight-line fit;
the intercept is what ramp-up and the tail of the launch cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rt_amd as ra

ctx = ra.Context(0)
w, h = 1920, 1080
depth = int(os.environ.get("RT_DEPTH", 8))
scene, cam = ra.Scene.builtin(0, aspect=w / h)
ctx.upload(scene)
acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
xs, ys = [], []
for spp in (2, 4, 8, 16, 32, 64, 128):
    st = ra.default_settings(w, h, spp=spp, max_depth=depth, chunk_samples=2)
    ms = []
    for _ in range(4):
        ctx.render_device(cam, st, acc.data_ptr(), None, None)
        ms.append(ctx.kernel_ms())
    xs.append(spp); ys.append(min(ms))
    print(f"spp {spp:4d}: kernel {min(ms):8.3f} ms  ({min(ms) / spp * 1e3:7.1f} us per spp)", flush=True)
b, a = np.polyfit(xs, ys, 1)
print(f"fit: {a:.3f} ms + {b * 1e3:.1f} us per spp -> fixed cost is {a / ys[3] * 100:.1f} % of the 16 spp launch")
ctx.close()
