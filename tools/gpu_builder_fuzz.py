"""Random scenes through the GPU builders, outside the suite: for every scene the GPU LBVH
(host arrays and device arrays) must give the bytes of the oracle's restatement, its
four-wide nodes must validate, a refit with moved primitives must give the bytes of the
oracle's refit, and a small frame must match the oracle bit for bit; then the same refit
and frame through the host SAH tree.  Where the tree has a slot tree (include/rtgpu_format.h) the
array made on the device must be the oracle's byte for byte, after the builds and after the refits.
usage: gpu_builder_fuzz.py [seed] [scenes]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import rt_amd as ra
from rt_amd.api import FlatSceneC
from oracle import Oracle
from test_lbvh import _flat_bytes
from util import make_scene

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(seed)
o = Oracle()
ctx = ra.Context(0)
MATS = [(0, .5, .5, .5, 0), (1, .8, .8, .8, 0.1), (2, 1, 1, 1, 1.5), (3, 1, .9, .8, 4.0)]
bad = 0
modes = {}

def slot_tree_ok():
    """device slot tree == the oracle's (both None where the tree has no slot form or no four-wide nodes)"""
    fs = ctx.flat_scene()
    got = ctx.debug_slot_tree()
    want = o.slot_tree(fs.c) if fs.c.nodes4 else None
    return (got is None and want is None) or (got is not None and want is not None and np.array_equal(got, want))

for it in range(count):
    shape = rng.integers(0, 5)
    ns = int(rng.choice([0, 0, 1, 2, 3, 7, 40, 300, 2500])) if shape != 0 else 0
    nt = int(rng.choice([0, 1, 2, 3, 4, 5, 9, 64, 700, 3000, 9000, 30000])) if shape != 1 else 0
    spread = float(rng.choice([0.0, 1e-3, 1.0, 50.0]))          # 0: all centroids coincide
    sph = [(*(rng.normal(0, spread, 3)), float(rng.uniform(0.05, 0.5)), int(rng.integers(0, 4))) for _ in range(ns)]
    tri = []
    for _ in range(nt):
        p = rng.normal(0, spread, 3)
        tri.append((tuple(p), tuple(p + rng.normal(0, .3, 3)), tuple(p + rng.normal(0, .3, 3)), int(rng.integers(0, 4))))
    scene = make_scene(ra, spheres=sph, triangles=tri, materials=MATS)
    what = f"scene {it}: {ns} spheres, {nt} triangles, spread {spread}"
    ctx.upload(scene, builder="lbvh")
    got = ctx.flat_scene()
    want = o.build_lbvh(scene.desc, FlatSceneC)
    ok = (got.c.n_nodes, got.c.max_depth) == (want.n_nodes, want.max_depth)
    a, b = _flat_bytes(got.c), _flat_bytes(want)
    ok = ok and all(a[k] == b[k] for k in a)
    rc, _ = o.validate_bvh(got.c)
    ok = ok and rc == 0
    # the same from device arrays
    d_s = torch.from_numpy(scene.spheres.view(np.uint8).copy()).cuda() if ns else None
    d_t = torch.from_numpy(scene.triangles.view(np.uint8).copy()).cuda() if nt else None
    ctx.build_device(d_s.data_ptr() if ns else None, ns, d_t.data_ptr() if nt else None, nt, scene.materials)
    c = _flat_bytes(ctx.flat_scene().c)
    ok = ok and all(c[k] == b[k] for k in c)
    # refit with moved primitives
    moved_s, moved_t = scene.spheres.copy(), scene.triangles.copy()
    if ns: moved_s["center"] += rng.normal(0, .2, (ns, 3)).astype(np.float32)
    if nt:
        for name in ("v0", "v1", "v2"): moved_t[name] += rng.normal(0, .05, (nt, 3)).astype(np.float32)
    moved = ra.Scene(moved_s, moved_t, scene.materials)
    before = ctx.flat_scene()
    want_r = o.refit(before.c, moved.desc)
    ctx.refit(moved)
    after = ctx.flat_scene()
    d, e = _flat_bytes(after.c), _flat_bytes(want_r)
    ok = ok and all(d[k] == e[k] for k in d)
    rc, _ = o.validate_bvh(after.c)
    ok = ok and rc == 0
    # a small frame
    cam = ra.look_at((0, 1, 6 + 3 * spread), (0, 0, 0), vfov=50, aspect=1.5)
    st = ra.default_settings(48, 32, spp=2, max_depth=5, flags=ra.FLAG_STATS)
    acc, ids, stats = ctx.render(cam, st)
    modes[ctx.kernel_name().rstrip(">").split(",")[-1]] = modes.get(ctx.kernel_name().rstrip(">").split(",")[-1], 0) + 1
    w_acc, w_ids, w_stats = o.render(after.c, cam, st, threads=8)
    ok = ok and np.array_equal(ids, w_ids) and np.array_equal(acc.view(np.uint32), w_acc.view(np.uint32)) and stats.as_dict() == w_stats
    ok = ok and slot_tree_ok()
    o.free_flat(want); o.free_flat(want_r)
    # the host SAH tree: refit and frame
    ctx.upload(scene)
    before = ctx.flat_scene()
    rc, _ = o.validate_bvh(before.c)
    ok = ok and rc == 0
    want_r = o.refit(before.c, moved.desc)
    ctx.refit(moved)
    after = ctx.flat_scene()
    d, e = _flat_bytes(after.c), _flat_bytes(want_r)
    ok = ok and all(d[k] == e[k] for k in d)
    if after.c.nodes4 and want_r.nodes4:
        grab = lambda ptr, n: bytes((C.c_char * n).from_address(ptr))
        ok = ok and grab(after.c.nodes4, after.c.n_nodes4 * 96) == grab(want_r.nodes4, want_r.n_nodes4 * 96)
    acc, ids, stats = ctx.render(cam, st)
    modes[ctx.kernel_name().rstrip(">").split(",")[-1]] = modes.get(ctx.kernel_name().rstrip(">").split(",")[-1], 0) + 1
    w_acc, w_ids, w_stats = o.render(after.c, cam, st, threads=8)
    ok = ok and np.array_equal(ids, w_ids) and np.array_equal(acc.view(np.uint32), w_acc.view(np.uint32)) and stats.as_dict() == w_stats
    ok = ok and slot_tree_ok()
    o.free_flat(want_r)
    if not ok:
        bad += 1
        print("MISMATCH", what, flush=True)
print(f"builder fuzz seed {seed}: {count} scenes, {bad} mismatches; kernel modes of the frames: {dict(sorted(modes.items()))}", flush=True)
ctx.close()
sys.exit(1 if bad else 0)
