"""First GPU check: S0/S1 through the C ABI vs the CPU oracle (bit-exact)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rt_amd as ra
from oracle import Oracle

o = Oracle()
ctx = ra.Context(0)
ok = True
for kind, param, w, h, spp, depth in ((0, 0, 320, 180, 4, 8), (1, 5, 320, 180, 2, 8), (0, 0, 1280, 720, 1, 8)):
    sc, cam = ra.Scene.builtin(kind, param=param, aspect=w / h)
    ctx.upload(sc)
    fs = ctx.flat_scene()
    st = ra.default_settings(w, h, spp=spp, max_depth=depth, flags=ra.FLAG_STATS)
    t = time.time(); a, ids, stats = ctx.render(cam, st); dt = time.time() - t
    kms = ctx.kernel_ms()
    t = time.time(); b, ids2, stats2 = o.render(fs.c, cam, st, threads=16); dtc = time.time() - t
    same_ids = np.array_equal(ids, ids2)
    same_acc = np.array_equal(a.view(np.uint32), b.view(np.uint32))
    nbad = int((a.view(np.uint32) != b.view(np.uint32)).any(axis=-1).sum())
    print(f"kind={kind} {w}x{h}x{spp}: ids_equal={same_ids} accum_bit_equal={same_acc} bad_px={nbad} "
          f"gpu_stats={stats.as_dict()} oracle_stats={stats2} kernel_ms={kms:.3f} call_s={dt:.3f} cpu_s={dtc:.2f}")
    ok &= same_ids and same_acc and stats.as_dict() == stats2
    st2 = ra.default_settings(w, h, spp=spp, max_depth=depth)
    a2, _, s2 = ctx.render(cam, st2)
    print("   non-stats kernel: equal", np.array_equal(a2, a), "ms", ctx.kernel_ms(), "Mrays/s", s2.rays / ctx.kernel_ms() / 1e3)
print("PARITY", "OK" if ok else "FAIL")
sys.exit(0 if ok else 1)
