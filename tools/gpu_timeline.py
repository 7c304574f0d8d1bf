"""Per-wave start / queue-dry / end times of one profiled launch: how long is the tail?"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rt_amd as ra
from gpu_perf import CONFIGS
ctx = ra.Context(0)
ra.lib.rtgpu_debug_wave_times.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
for name in sys.argv[1:] or ["c2"]:
    kind, level, w, h, spp, depth = CONFIGS[name][:6]
    depth = int(os.environ.get("RT_DEPTH", depth))
    scene, cam = ra.Scene.builtin(kind, param=level, aspect=w / h)
    ctx.upload(scene)
    acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    st = ra.default_settings(w, h, spp=spp, max_depth=depth, flags=ra.FLAG_PROFILE)
    ctx.render_device(cam, st, acc.data_ptr(), None, None)
    buf = np.zeros((8192, 4), np.uint64); n = C.c_uint64()
    assert ra.lib.rtgpu_debug_wave_times(ctx._h, buf.ctypes.data, 8192, C.byref(n)) == 0
    t = buf[: n.value].astype(np.float64)
    t0 = t[:, 0].min()
    start, end, dry = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0   # microseconds
    q = lambda a: " ".join(f"{np.percentile(a, p):8.1f}" for p in (0, 10, 50, 90, 99, 100))
    print(f"{name}: kernel {ctx.kernel_ms():.3f} ms, {n.value} waves; microseconds, percentiles 0/10/50/90/99/100")
    print("   start    ", q(start))
    print("   queue dry", q(dry[dry > 0]))
    print("   end      ", q(end))
    print("   end - dry", q((end - dry)[dry > 0]))
    print("   outer iterations after dry", q(t[:, 3][dry > 0]), " -> us per iteration", f"{np.median(((end - dry) / np.maximum(t[:, 3], 1))[dry > 0]):.1f}")
    print(f"   mean busy share until last end: {(end - start).sum() / (len(end) * end.max()):.3f}")
    # shader clock during the launch: cycles of the per-section timers (clock64) over the same waves' 100 MHz wall time
    c = ctx.profile_counters()
    print(f"   shader clock seen by the waves: {sum(c[16:21]) / (end - start).sum():.0f} MHz")
ctx.close()
