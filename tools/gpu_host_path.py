"""PCIe-inclusive rate of the host-buffer entry point rtgpu_render on C2 (never the bench's `value`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rt_amd as ra
ctx = ra.Context(0)
scene, cam = ra.Scene.builtin(0, aspect=1920 / 1080)
ctx.upload(scene)
st = ra.default_settings(1920, 1080, spp=16, max_depth=8)
acc = np.zeros((1080, 1920, 4), np.float32)
_, _, stats = ctx.render(cam, st, accum=acc, want_ids=False, want_stats=True)
t = []
for _ in range(5):
    t0 = time.perf_counter()
    ctx.render(cam, st, accum=acc, want_ids=False, want_stats=False)
    t.append(time.perf_counter() - t0)
ms = 1e3 * float(np.median(t))
print(f"rtgpu_render (host accum buffer, 33 MB back over PCIe): {ms:.3f} ms per call, kernel {ctx.kernel_ms():.3f} ms, "
      f"{stats.rays / ms / 1e3:.0f} Mrays/s PCIe-inclusive")
ctx.close()
