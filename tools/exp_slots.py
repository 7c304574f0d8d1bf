"""Slot tree (modes 11/12) against the four-wide nodes it is made from (modes 5/6/9/10) on the same scenes:
images, first-hit ids, counters and kernel times.  usage: python tools/exp_slots.py [c3] [c5] [c5full] ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rt_amd as ra
from gpu_perf import CONFIGS

def run(name, slots, reps=5):
    os.environ["RTGPU_SLOT_TREE"] = "1" if slots else "0"
    ctx = ra.Context(0)
    kind, level, w, h, spp, depth = CONFIGS[name][:6]
    scene, cam = ra.Scene.builtin(kind, param=level, aspect=w / h)
    ctx.upload(scene, builder=os.environ.get("RT_BUILDER", "sah"))
    acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    ids = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    st = ra.default_settings(w, h, spp=spp, max_depth=depth, flags=ra.FLAG_STATS)
    stats = ra.Stats()
    ctx.render_device(cam, st, acc.data_ptr(), ids.data_ptr(), None, stats)
    s = stats.as_dict()
    st.flags = 0
    ms = []
    for _ in range(reps):
        ctx.render_device(cam, st, acc.data_ptr(), None, None)
        ms.append(ctx.kernel_ms())
    out = (acc.cpu().numpy().copy(), ids.cpu().numpy().copy(), s, ctx.kernel_name(), min(ms), float(np.median(ms)))
    ctx.close()
    return out

for name in sys.argv[1:] or ["c3"]:
    a = run(name, False)
    b = run(name, True)
    same_px = float((a[0] == b[0]).all(axis=-1).mean())
    same_id = float((a[1] == b[1]).mean())
    print(f"{name}: {a[3]} min {a[4]:.3f} med {a[5]:.3f} ms | {b[3]} min {b[4]:.3f} med {b[5]:.3f} ms | "
          f"pixels equal {same_px:.6f} ids equal {same_id:.6f} | nodes {a[2]['nodes']} -> {b[2]['nodes']} "
          f"tris {a[2]['tri_tests']} -> {b[2]['tri_tests']} rays {a[2]['rays']} -> {b[2]['rays']}", flush=True)
