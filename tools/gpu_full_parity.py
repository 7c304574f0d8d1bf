"""Full-size parity, outside the test suite (minutes of oracle time): every benchmark frame
at its stated resolution and depth through the C ABI against the CPU oracle, bit for bit
(accumulators, primary-hit ids, traversal counters), for the SAH and the GPU-built tree.
The sample count is cut to what the oracle traces in about ten seconds per case; the
per-sample arithmetic does not depend on it.  usage: gpu_full_parity.py [threads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rt_amd as ra
from oracle import Oracle

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
o = Oracle()
ctx = ra.Context(0)
ok = True
CASES = (("C1", 0, 0, 1280, 720, 1), ("C2", 0, 0, 1920, 1080, 16), ("C3", 1, 8, 1920, 1080, 4),
         ("C4 frame", 0, 0, 3840, 2160, 2), ("C5", 2, 8, 1920, 1080, 2))
for name, kind, param, w, h, spp in CASES:
    scene, cam = ra.Scene.builtin(kind, param=param, aspect=w / h)
    for builder in ("sah", "lbvh"):
        ctx.upload(scene, builder=builder)
        fs = ctx.flat_scene()
        st = ra.default_settings(w, h, spp=spp, max_depth=8, flags=ra.FLAG_STATS)
        acc, ids, stats = ctx.render(cam, st)
        t = time.time()
        want, want_ids, want_stats = o.render(fs.c, cam, st, threads=threads)
        dt = time.time() - t
        same = (np.array_equal(ids, want_ids) and np.array_equal(acc.view(np.uint32), want.view(np.uint32))
                and stats.as_dict() == want_stats)
        ok &= same
        counting = ctx.kernel_name()
        # the product kernel (no statistics flag: its own instantiation, registers and launch shape) gives the same frame
        st.flags = 0
        acc2, ids2, _ = ctx.render(cam, st, want_stats=False)
        same2 = np.array_equal(ids2, ids) and np.array_equal(acc2.view(np.uint32), acc.view(np.uint32))
        ok &= same2
        print(f"{name} {builder}: {w}x{h} {spp} spp, {len(scene.triangles) + len(scene.spheres)} prims, {counting}, "
              f"{stats.rays} rays: {'bit-identical, counters equal' if same else 'MISMATCH'} (oracle {dt:.1f} s); "
              f"{ctx.kernel_name()}: {'the same frame' if same2 else 'MISMATCH'}", flush=True)
print("FULL PARITY", "OK" if ok else "FAIL", flush=True)
ctx.close()
sys.exit(0 if ok else 1)
