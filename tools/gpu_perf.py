"""Kernel-time survey over the BASELINE.json configs on one GPU (not the bench
contract: a developer tool for A/B-ing kernel changes).
usage: python tools/gpu_perf.py [c2] [c3] [c5] [--reps N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rt_amd as ra

CONFIGS = {
    # name: (kind, level, w, h, spp, depth)
    "c1": (0, 0, 1280, 720, 1, 8),
    "c2": (0, 0, 1920, 1080, 16, 8),
    "c3": (1, 8, 1920, 1080, 8, 8),     # C3 is 64 spp; 8 spp here (same per-ray work)
    "c5": (2, 7, 1920, 1080, 4, 8),     # S2 at level 7 = 2.6M tris (level 8 = 10.5M)
    "c5full": (2, 8, 1920, 1080, 4, 8),
    "c3full": (1, 8, 1920, 1080, 64, 8),   # C3 as BASELINE.json states it
    "c5full16": (2, 8, 1920, 1080, 16, 8), # C5's frame as BASELINE.json states it (16 spp), on ONE GPU
    "c4": (0, 0, 3840, 2160, 256, 8),      # C4's frame on ONE GPU (its 2/4/8-GPU runs are the driver's)
    "c2lens": (0, 0, 1920, 1080, 16, 8, 0.1),  # C2 through the thin lens (aperture 0.1)
}

def main():
    reps = 5
    args = sys.argv[1:]
    if "--reps" in args:
        k = args.index("--reps")
        reps = int(args[k + 1])
        del args[k:k + 2]
    names = [a for a in args if not a.startswith("--")] or ["c2", "c3", "c5"]
    ctx = ra.Context(0)
    for name in names:
        kind, level, w, h, spp, depth = CONFIGS[name][:6]
        t0 = time.time()
        scene, cam = ra.Scene.builtin(kind, param=level, aspect=w / h)
        if len(CONFIGS[name]) > 6:  # the RTIOW cover camera with an aperture
            cam = ra.look_at((13, 2, 3), (0, 0, 0), vfov=20.0, aspect=w / h, aperture=CONFIGS[name][6], focus_dist=10.0)
        t1 = time.time()
        ctx.upload(scene, builder=os.environ.get("RT_BUILDER", "sah"))
        t2 = time.time()
        fs = ctx.flat_scene()
        acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
        st = ra.default_settings(w, h, spp=spp, max_depth=depth, flags=ra.FLAG_STATS)
        if os.environ.get("RT_CHUNK"):
            st.chunk_samples = int(os.environ["RT_CHUNK"])
        stats = ra.Stats()
        ctx.render_device(cam, st, acc.data_ptr(), None, None, stats)
        s = stats.as_dict()
        st.flags = 0
        ms = []
        for _ in range(reps):
            ctx.render_device(cam, st, acc.data_ptr(), None, None)
            ms.append(ctx.kernel_ms())
        ms = np.array(ms)
        wide = f"/{fs.c.n_nodes4}x4" if fs.c.nodes4 else ""
        slot_tree = ctx.kernel_name().rstrip(">").split(",")[-1] in ("11", "12")
        node_bytes = 64 if slot_tree else 96 if fs.c.nodes4 else 48 if fs.c.nodes48 else 64   # slot tree / four-wide / packed nodes
        abytes = node_bytes * s["nodes"] + 48 * s["tri_tests"] + 16 * s["sphere_tests"] + 32 * s["scatters"] + 16 * w * h
        print(f"{name}: prims={scene.n_prims} nodes={fs.c.n_nodes}{wide} depth={fs.c.max_depth} gen={t1-t0:.1f}s "
              f"build+upload={t2-t1:.1f}s (bvh {fs.c.build_ms:.1f} ms) | {w}x{h}x{spp} rays={s['rays']} "
              f"nodes/ray={s['nodes']/s['rays']:.1f} tris/ray={s['tri_tests']/s['rays']:.1f} "
              f"sph/ray={s['sphere_tests']/s['rays']:.1f} B/ray={abytes/s['rays']:.0f} | "
              f"{ctx.kernel_name()} kernel ms min={ms.min():.3f} med={np.median(ms):.3f} -> {s['rays']/np.median(ms)/1e3:.0f} Mrays/s, "
              f"{abytes/np.median(ms)/1e6:.0f} GB/s algorithmic", flush=True)
    ctx.close()

if __name__ == "__main__":
    main()
