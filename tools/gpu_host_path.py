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
# the progressive loop of INTEGRATION.md §2: sums stay on the GPU, 8 MB of resolved pixels come back
st_r = ra.default_settings(1920, 1080, spp=16, max_depth=8, flags=ra.FLAG_ACCUMULATE)
ctx.render_resolved(cam, st_r)
t = []
for i in range(8):
    st_r.sample_offset = 16 * (i + 1)
    t0 = time.perf_counter()
    ctx.render_resolved(cam, st_r)
    t.append(time.perf_counter() - t0)
ms_r = 1e3 * float(np.median(t))
print(f"rtgpu_render_resolved (sums resident, 8 MB rgba8 back through pinned staging): {ms_r:.3f} ms per call, "
      f"kernel {ctx.kernel_ms():.3f} ms, {stats.rays / ms_r / 1e3:.0f} Mrays/s PCIe-inclusive")
ctx.close()
