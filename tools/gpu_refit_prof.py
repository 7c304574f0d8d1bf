"""Refit timing: S2 at a given level (8 = 10.5 M triangles), both builders; the
vertices are displaced by a small wave and the tree refitted, twice.  Prints the GPU
time of the refit (kernels only), the wall time of the call (incl. the H2D copy of the
caller's arrays) and the trace kernel's time before and after."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rt_amd as ra

level = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = ra.Context(0)
scene, cam = ra.Scene.builtin(2, param=level, aspect=16 / 9)
st = ra.default_settings(1920, 1080, spp=4, max_depth=8)
accum = torch.zeros(1080, 1920, 4, device="cuda")
tri = scene.triangles.copy()
for k in ("v0", "v1", "v2"):
    tri[k][:, 1] += (0.02 * np.sin(3.0 * tri[k][:, 0])).astype(np.float32)
moved = ra.Scene(scene.spheres, tri, scene.materials)

def frame():
    best = 1e9
    for _ in range(3):
        ctx.render_device(cam, st, accum.data_ptr())
        torch.cuda.synchronize()
        best = min(best, ctx.kernel_ms())
    return best

for builder in ("sah", "lbvh"):
    t0 = time.perf_counter(); ctx.upload(scene, builder=builder); t_up = (time.perf_counter() - t0) * 1e3
    ms0 = frame()
    for which, sc in (("moved", moved), ("back", scene)):
        t0 = time.perf_counter(); ctx.refit(sc); wall = (time.perf_counter() - t0) * 1e3
        # build_ms of the flat scene = GPU time of the last build / refit (this also pays the lazy copy back)
        gpu_ms = ctx.flat_scene().c.build_ms
        print(f"{builder}: {scene.n_prims} prims, upload {t_up:.0f} ms | refit {which}: GPU {gpu_ms:.2f} ms, call {wall:.1f} ms | "
              f"trace kernel {ms0:.3f} -> {frame():.3f} ms", flush=True)
# the same refit with the moved triangles already on the GPU (rtgpu_refit_device)
d_tri = torch.from_numpy(moved.triangles.view(np.uint8).copy()).cuda()
torch.cuda.synchronize()
for _ in range(2):
    t0 = time.perf_counter(); ctx.refit_device(None, 0, d_tri.data_ptr(), len(moved.triangles)); wall = (time.perf_counter() - t0) * 1e3
    print(f"{builder}: refit from device arrays: call {wall:.1f} ms (input check, leaf records, boxes, packed and four-wide copies)", flush=True)
ctx.close()
