"""A dynamic scene end to end on the GPU: every frame a torch kernel moves the vertices of
S2 (level 8: 10.5 M triangles) in place, rtgpu_refit_device refits the uploaded tree from
those device arrays (or, second pass, rtgpu_build_device builds a NEW tree over them), and
the frame is traced - nothing crosses PCIe.  Prints ms per frame and the share of each
part.  usage: gpu_animation.py [level] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rt_amd as ra

level = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
w, h = 1920, 1080
ctx = ra.Context(0)
scene, cam = ra.Scene.builtin(2, param=level, aspect=w / h)
t0 = time.perf_counter(); ctx.upload(scene, builder=os.environ.get("RT_BUILDER", "lbvh")); t_up = time.perf_counter() - t0
st = ra.default_settings(w, h, spp=4, max_depth=8)
accum = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
# the caller's triangle records (40 bytes: v0 v1 v2 material) as a float view on the GPU
rest = torch.from_numpy(scene.triangles.view(np.float32).reshape(-1, 10).copy()).cuda()
cur = rest.clone()
n_tri = len(scene.triangles)
torch.cuda.synchronize()
for mode in ("refit", "rebuild"):
    t_move = t_refit = t_trace = 0.0
    for f in range(frames + 2):
        if f == 2:
            t_move = t_refit = t_trace = 0.0
        a = time.perf_counter()
        phase = 0.3 * f
        for v in range(3):  # y += 0.02 * sin(3 x + phase) per vertex
            cur[:, 3 * v + 1] = rest[:, 3 * v + 1] + 0.02 * torch.sin(3.0 * rest[:, 3 * v] + phase)
        torch.cuda.synchronize()
        b = time.perf_counter()
        if mode == "refit":
            ctx.refit_device(None, 0, cur.data_ptr(), n_tri)
        else:
            ctx.build_device(None, 0, cur.data_ptr(), n_tri, scene.materials)
        c = time.perf_counter()
        ctx.render_device(cam, st, accum.data_ptr())
        torch.cuda.synchronize()
        d = time.perf_counter()
        t_move += b - a; t_refit += c - b; t_trace += d - c
    per = (t_move + t_refit + t_trace) / frames * 1e3
    print(f"S2 level {level}: {n_tri} triangles, upload {t_up * 1e3:.0f} ms; {frames} animated frames at 1080p 4 spp, {mode} every frame: "
          f"{per:.2f} ms per frame = move {t_move / frames * 1e3:.2f} + {mode} {t_refit / frames * 1e3:.2f} + trace {t_trace / frames * 1e3:.2f}",
          flush=True)
ctx.close()
