"""Minimal target for rocprofv3 passes: render one config a few times.
usage: python3 tools/gpu_prof_target.py c2|c3|c5 [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rt_amd as ra
from gpu_perf import CONFIGS

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kind, level, w, h, spp, depth = CONFIGS[name]
ctx = ra.Context(0)
scene, cam = ra.Scene.builtin(kind, param=level, aspect=w / h)
ctx.upload(scene)
acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
st = ra.default_settings(w, h, spp=spp, max_depth=depth)
for _ in range(reps):
    ctx.render_device(cam, st, acc.data_ptr(), None, None)
torch.cuda.synchronize()
print(name, "kernel ms", ctx.kernel_ms())
ctx.close()
