"""Wall time of an LBVH upload (S1 level 8, S2 level 8) against the GPU time of the build: what the
call costs besides its kernels.  usage: gpu_lbvh_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rt_amd as ra
ctx = ra.Context(0)
for kind, param in ((1, 8), (2, 8)):
    scene, cam = ra.Scene.builtin(kind, param=param, aspect=16 / 9)
    for rep in range(2):
        t0 = time.perf_counter(); ctx.upload(scene, builder="lbvh"); t1 = time.perf_counter()
        fs = ctx.flat_scene()
        print(f"kind {kind}: {len(scene.triangles)} tris: upload wall {1e3 * (t1 - t0):.0f} ms, GPU build {fs.c.build_ms:.1f} ms, nodes {fs.c.n_nodes} nodes4 {fs.c.n_nodes4}", flush=True)
    os.environ["RTGPU_NO_WIDE"] = "1"
    t0 = time.perf_counter(); ctx.upload(scene, builder="lbvh"); t1 = time.perf_counter()
    print(f"   without the four-wide collapse: upload wall {1e3 * (t1 - t0):.0f} ms", flush=True)
    del os.environ["RTGPU_NO_WIDE"]
ctx.close()
