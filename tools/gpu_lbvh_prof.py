"""Target for rocprofv3: GPU LBVH build of S2 at level 8 (10.5 M triangles), twice."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rt_amd as ra
level = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = ra.Context(0)
scene, cam = ra.Scene.builtin(2, param=level)
for _ in range(2):
    ctx.upload(scene, builder="lbvh")
    print("lbvh", scene.n_prims, "prims:", round(ctx.flat_scene().c.build_ms, 2), "ms, depth", ctx.flat_scene().c.max_depth)
ctx.close()
