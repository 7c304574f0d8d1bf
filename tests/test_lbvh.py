"""GPU LBVH builder against its CPU restatement
(oracle rto_build_lbvh): flattened bytes identical, tree structurally valid, and
frames traced through the GPU-built tree identical to the oracle's."""
import ctypes as C
import os

import numpy as np
import pytest

from rt_amd.api import FlatSceneC
from util import bits, make_scene

MATS = [(0, .5, .5, .5, 0), (1, .8, .8, .8, 0.1), (2, 1, 1, 1, 1.5)]


def _flat_bytes(c, wide=False):
    """The arrays of a flattened scene as bytes, tails included (wide: and the four-wide nodes,
    which the restatement of the builder does not make - they are compared GPU against GPU)."""
    def grab(ptr, n):
        return bytes((C.c_char * n).from_address(ptr)) if n else b""
    return {
        "nodes": grab(c.nodes, c.n_nodes * 64),
        "tris": grab(c.tris, (c.n_tris + 8) * 48),
        "spheres": grab(c.spheres, (c.n_spheres + 8) * 16),
        "meta": grab(c.sphere_meta, (c.n_spheres * 2 + 16) * 4),
        "materials": grab(c.materials, c.n_materials * 32),
        "nodes48": grab(c.nodes48, c.n_nodes * 48) if c.nodes48 else b"",
        **({} if not wide else {
        # numbered from a scan in thread order (collapse4.hip): the same bytes on every run
        "nodes4": grab(c.nodes4, c.n_nodes4 * 96) if c.nodes4 else b"",
        "nodes4_src": grab(c.nodes4_src, c.n_nodes4 * 16) if c.nodes4_src else b""}),
    }


def _scenes(ra):
    rng = np.random.default_rng(5)
    cloud = [(float(x), float(y), float(z), float(r), int(m)) for (x, y, z), r, m in
             zip(rng.uniform(-4, 4, (300, 3)), rng.uniform(0.05, 0.4, 300), rng.integers(0, 3, 300))]
    soup = [(tuple(p), tuple(p + rng.normal(0, .3, 3)), tuple(p + rng.normal(0, .3, 3)), int(rng.integers(0, 3)))
            for p in rng.uniform(-3, 3, (500, 3))]
    return {
        "S0 (spheres + quad)": ra.Scene.builtin(0)[0],
        "S1 level 5 (20k tris)": ra.Scene.builtin(1, param=5)[0],
        "S2 level 3 (mixed materials)": ra.Scene.builtin(2, param=3)[0],
        "random mixed": make_scene(ra, spheres=cloud, triangles=soup, materials=MATS),
        "coincident": make_scene(ra, spheres=[(0, 0, 0, 1, 0)] * 40, materials=MATS),
        "collinear centroids": make_scene(ra, spheres=[(float(i), 0, 0, .4, i % 3) for i in range(33)], materials=MATS),
        "two": make_scene(ra, spheres=[(0, 0, 0, 1, 0)], triangles=[((-2, -2, -1), (2, -2, -1), (0, 2, -1), 1)], materials=MATS),
        "one leaf of two": make_scene(ra, spheres=[(0, 0, 0, 1, 0), (0.5, 0, 0, 1, 1)], materials=MATS),
        "one leaf of four": make_scene(ra, triangles=[((-1, -1, .1 * k), (1, -1, .1 * k), (0, 1, .1 * k), k % 3)
                                                      for k in range(4)], materials=MATS),
        "one sphere": make_scene(ra, spheres=[(0, 0, 0, 1, 2)], materials=MATS),
        "one triangle": make_scene(ra, triangles=[((-1, -1, 0), (1, -1, 0), (0, 1, 0), 1)], materials=MATS),
        "empty": make_scene(ra, materials=MATS),
    }


def test_oracle_lbvh_is_a_valid_tree(ra, oracle):
    """CPU only: the restatement itself produces structurally valid trees."""
    for name, scene in _scenes(ra).items():
        fs = oracle.build_lbvh(scene.desc, FlatSceneC)
        try:
            rc, depth = oracle.validate_bvh(fs)
            assert rc == 0, f"{name}: validator code {rc}"
            assert depth == fs.max_depth <= 62, name
            # one-primitive leaves would need P - 1 nodes; same-type pairs are merged into one leaf
            assert max(1, (scene.n_prims - 1) // 2) <= fs.n_nodes <= max(1, scene.n_prims - 1), name
            # every primitive id appears exactly once in the leaf arrays
            tris = np.frombuffer(_flat_bytes(fs)["tris"], np.uint32).reshape(-1, 12)[:fs.n_tris]
            meta = np.frombuffer(_flat_bytes(fs)["meta"], np.uint32).reshape(-1, 2)[:fs.n_spheres]
            ids = np.sort(np.concatenate([meta[:, 0], tris[:, 3]]))
            assert np.array_equal(ids, np.arange(scene.n_prims)), name
        finally:
            oracle.free_flat(fs)


@pytest.mark.gpu
def test_gpu_lbvh_matches_the_restatement_byte_for_byte(ra, oracle, gpu_ctx):
    for name, scene in _scenes(ra).items():
        gpu_ctx.upload(scene, builder="lbvh")
        got = gpu_ctx.flat_scene()
        want = oracle.build_lbvh(scene.desc, FlatSceneC)
        try:
            assert (got.c.n_nodes, got.c.n_tris, got.c.n_spheres, got.c.max_depth) == (
                want.n_nodes, want.n_tris, want.n_spheres, want.max_depth), name
            a, b = _flat_bytes(got.c), _flat_bytes(want)
            for k in a:
                assert a[k] == b[k], f"{name}: {k} differ"
        finally:
            oracle.free_flat(want)


@pytest.mark.gpu
def test_frames_through_the_gpu_built_tree_match_the_oracle(ra, oracle, gpu_ctx):
    for kind, param, w, h in ((0, 0, 160, 90), (1, 5, 128, 72), (2, 3, 128, 72)):
        scene, cam = ra.Scene.builtin(kind, param=param, aspect=w / h)
        gpu_ctx.upload(scene, builder="lbvh")
        st = ra.default_settings(w, h, spp=2, max_depth=8, flags=ra.FLAG_STATS)
        acc, ids, stats = gpu_ctx.render(cam, st)
        want, want_ids, want_stats = oracle.render(gpu_ctx.flat_scene().c, cam, st, threads=4)
        assert np.array_equal(ids, want_ids) and np.array_equal(bits(acc), bits(want))
        assert stats.as_dict() == want_stats
        # the SAH tree finds the same primary hits (a ray through a shared edge may name
        # either neighbour: the order of equal-distance tests depends on the tree)
        gpu_ctx.upload(scene)
        _, sah_ids, _ = gpu_ctx.render(cam, st)
        assert (ids == sah_ids).mean() > 0.999


@pytest.mark.gpu
def test_lbvh_upload_rejects_bad_input(ra, gpu_ctx):
    bad = make_scene(ra, spheres=[(0, 0, 0, 1, 5)], materials=[(0, .5, .5, .5, 0)])
    with pytest.raises(ra.RtgpuError):
        gpu_ctx.upload(bad, builder="lbvh")
    nan = make_scene(ra, triangles=[((float("nan"), 0, 0), (1, 0, 0), (0, 1, 0), 0)], materials=[(0, .5, .5, .5, 0)])
    with pytest.raises(ra.RtgpuError):
        gpu_ctx.upload(nan, builder="lbvh")


@pytest.mark.gpu
def test_four_wide_nodes_collapsed_on_the_gpu(ra, oracle, gpu_ctx):
    """collapse4.hip: the four-wide nodes of a GPU-built tree are made on the GPU.  They must
    be a valid tree over the same primitives (validator of the restatement), have the node
    count and stack bound of the host's collapse of the same BVH2 (same rule, other
    numbering), and every node must hold the boxes its source map names."""
    if os.environ.get("RTGPU_NO_WIDE"):
        pytest.skip("asserts the four-wide nodes that RTGPU_NO_WIDE switches off")
    ra.lib.rtgpu_debug_collapse4_counts.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    for kind, param in ((1, 5), (2, 4), (1, 6)):
        scene, cam = ra.Scene.builtin(kind, param=param, aspect=16 / 9)
        gpu_ctx.upload(scene, builder="lbvh")
        fs = gpu_ctx.flat_scene()
        assert fs.c.n_nodes >= ra.NODE48_MIN_NODES and fs.c.nodes4 and fs.c.nodes4_src
        rc, _ = oracle.validate_bvh(fs.c)
        assert rc == 0, f"validator code {rc}"
        n4, ms = C.c_uint64(), C.c_uint32()
        assert ra.lib.rtgpu_debug_collapse4_counts(fs.c.nodes, fs.c.n_nodes, C.byref(n4), C.byref(ms)) == 0
        assert (fs.c.n_nodes4, fs.c.max_stack4) == (n4.value, ms.value)
        # boxes: child k of node i = the box (node2, slot) of the BVH2 its source word names, moved
        # by less than 512 floats to carry the reference bytes; missing children have no source
        nodes = fs.nodes_u32.reshape(-1, 16)
        w4 = fs.nodes4_u32.reshape(-1, 4, 6)
        src = fs.nodes4_src.reshape(-1, 4)
        have = src != 0xFFFFFFFF
        boxes2 = nodes[:, :12].reshape(-1, 2, 6)                       # lo0 hi0 | lo1 hi1
        want = boxes2[src[have] >> 1, src[have] & 1]
        got = w4[have]
        key = lambda u: np.where(u & 0x80000000, ~u, u | 0x80000000).astype(np.int64)   # ordered float bits
        d = key(got) - key(want)
        assert np.abs(d).max() < 512
        assert (d[:, :3] <= 0).all() and (d[:, 3:] >= 0).all()          # lo planes down, hi planes up
        refs = (w4[..., 0] & 0xFF) | ((w4[..., 1] & 0xFF) << 8) | ((w4[..., 2] & 0xFF) << 16) | ((w4[..., 3] & 0xFF) << 24)
        assert (refs[~have] == 0xFFFFFFFF).all()
        # every inner reference names a later node, each exactly once (breadth-first numbering)
        inner = refs[have & ((refs & 0x80000000) == 0)]
        assert np.array_equal(np.sort(inner), np.arange(1, fs.c.n_nodes4))
        # and the frame through them is the oracle's, counters included
        st = ra.default_settings(96, 54, spp=2, max_depth=6, flags=ra.FLAG_STATS)
        acc, ids, stats = gpu_ctx.render(cam, st)
        want_acc, want_ids, want_stats = oracle.render(fs.c, cam, st, threads=4)
        assert np.array_equal(ids, want_ids) and np.array_equal(bits(acc), bits(want_acc))
        assert stats.as_dict() == want_stats


@pytest.mark.gpu
def test_gpu_built_tree_is_reproducible(ra, gpu_ctx):
    """The four-wide nodes of a GPU-built tree carry their child numbers in the low bytes of box
    planes, so the numbering reaches the traversal order and the counters.  It comes from a scan
    in thread order: the same scene gives the same bytes twice on one context and on a fresh
    one (what N ranks that each build their own tree rely on), and the same counters."""
    if os.environ.get("RTGPU_NO_WIDE"):
        pytest.skip("asserts the four-wide nodes that RTGPU_NO_WIDE switches off")
    scene, cam = ra.Scene.builtin(2, param=5, aspect=16 / 9)   # 8 x 20k triangles: ~100k four-wide nodes
    st = ra.default_settings(160, 90, spp=2, max_depth=8, flags=ra.FLAG_STATS)
    runs = []
    fresh = ra.Context(0)
    try:
        for ctx in (gpu_ctx, gpu_ctx, fresh):
            ctx.upload(scene, builder="lbvh")
            fs = ctx.flat_scene()
            assert fs.c.nodes4 and fs.c.n_nodes4 > 10000
            acc, ids, stats = ctx.render(cam, st)
            runs.append((_flat_bytes(fs.c, wide=True), bits(acc).copy(), ids.copy(), stats.as_dict()))
    finally:
        fresh.close()
    for other in runs[1:]:
        for k in runs[0][0]:
            assert other[0][k] == runs[0][0][k], f"{k} differ between two builds of one scene"
        assert np.array_equal(other[1], runs[0][1]) and np.array_equal(other[2], runs[0][2])
        assert other[3] == runs[0][3]


@pytest.mark.gpu
def test_build_from_device_arrays(ra, oracle, gpu_ctx):
    """rtgpu_build_device: the tree over primitive arrays in GPU memory is byte for byte the
    tree rtgpu_upload_scene_lbvh builds from the same arrays on the host; bad input is found
    on the GPU; counts may change from one build to the next."""
    import torch
    for name in ("random mixed", "S1 level 5 (20k tris)", "one triangle", "one sphere", "two", "empty"):
        scene = _scenes(ra)[name]
        gpu_ctx.upload(scene, builder="lbvh")
        want_fs = gpu_ctx.flat_scene()
        want = _flat_bytes(want_fs.c, wide=True)
        want_counts = (want_fs.c.n_nodes, want_fs.c.max_depth, want_fs.c.n_nodes4, want_fs.c.max_stack4)
        d_sph = torch.from_numpy(scene.spheres.view(np.uint8).copy()).cuda() if len(scene.spheres) else None
        d_tri = torch.from_numpy(scene.triangles.view(np.uint8).copy()).cuda() if len(scene.triangles) else None
        gpu_ctx.build_device(d_sph.data_ptr() if d_sph is not None else None, len(scene.spheres),
                             d_tri.data_ptr() if d_tri is not None else None, len(scene.triangles), scene.materials)
        got_fs = gpu_ctx.flat_scene()
        got = _flat_bytes(got_fs.c, wide=True)
        for k in want:
            assert got[k] == want[k], f"{name}: {k} differ"
        assert (got_fs.c.n_nodes, got_fs.c.max_depth, got_fs.c.n_nodes4, got_fs.c.max_stack4) == want_counts, name
        rc, _ = oracle.validate_bvh(got_fs.c)
        assert rc == 0, f"{name}: validator code {rc}"
    # a frame through a device-built tree, then a rebuild with other counts
    scene, cam = ra.Scene.builtin(1, param=5, aspect=16 / 9)
    d_tri = torch.from_numpy(scene.triangles.view(np.uint8).copy()).cuda()
    gpu_ctx.build_device(None, 0, d_tri.data_ptr(), len(scene.triangles), scene.materials)
    st = ra.default_settings(96, 54, spp=2, max_depth=6, flags=ra.FLAG_STATS)
    acc, ids, stats = gpu_ctx.render(cam, st)
    w_acc, w_ids, w_stats = oracle.render(gpu_ctx.flat_scene().c, cam, st, threads=4)
    assert np.array_equal(ids, w_ids) and np.array_equal(bits(acc), bits(w_acc)) and stats.as_dict() == w_stats
    half = len(scene.triangles) // 2
    gpu_ctx.build_device(None, 0, d_tri.data_ptr(), half, scene.materials)
    assert gpu_ctx.flat_scene().c.n_tris == half
    acc2, ids2, _ = gpu_ctx.render(cam, st)
    w_acc2, w_ids2, _ = oracle.render(gpu_ctx.flat_scene().c, cam, st, threads=4)
    assert np.array_equal(ids2, w_ids2) and np.array_equal(bits(acc2), bits(w_acc2))
    assert ids2[ids2 != 0xFFFFFFFF].max() < half
    # bad input: found on the GPU, nothing uploaded
    bad = scene.triangles.copy()
    bad["v2"][11, 0] = np.nan
    d_bad = torch.from_numpy(bad.view(np.uint8).copy()).cuda()
    with pytest.raises(ra.RtgpuError, match="non-finite"):
        gpu_ctx.build_device(None, 0, d_bad.data_ptr(), len(bad), scene.materials)
    bad = scene.triangles.copy()
    bad["material"][5] = len(scene.materials)
    d_bad = torch.from_numpy(bad.view(np.uint8).copy()).cuda()
    with pytest.raises(ra.RtgpuError):
        gpu_ctx.build_device(None, 0, d_bad.data_ptr(), len(bad), scene.materials)
    with pytest.raises(ra.RtgpuError, match="NULL"):
        gpu_ctx.build_device(None, 3, None, 0, scene.materials)
    with pytest.raises(ra.RtgpuError, match="no scene"):
        gpu_ctx.render(cam, st)


@pytest.mark.gpu
def test_repeated_device_builds_do_not_grow_gpu_memory(ra, gpu_ctx):
    """The context's pool (dev_pool.h): a frame loop of rtgpu_build_device calls reaches a
    steady state, blocks of sizes no longer asked for are freed after two builds, and an
    upload from host arrays releases what the pool kept."""
    import torch
    scene, _ = ra.Scene.builtin(1, param=7)                       # 328 k triangles
    d_tri = torch.from_numpy(scene.triangles.view(np.uint8).copy()).cuda()
    n = len(scene.triangles)
    small, _ = ra.Scene.builtin(0)
    # warm-up: what the HIP runtime allocates once per process (kernel scratch, code objects,
    # rocPRIM state) must not count as growth
    gpu_ctx.build_device(None, 0, d_tri.data_ptr(), n, scene.materials)
    gpu_ctx.build_device(None, 0, d_tri.data_ptr(), n // 16, scene.materials)
    gpu_ctx.upload(small)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    used = []
    for k in range(8):
        gpu_ctx.build_device(None, 0, d_tri.data_ptr(), n, scene.materials)
        used.append(free_before - torch.cuda.mem_get_info()[0])
    assert used[0] > 50 * n                                       # the scene itself is tens of bytes per triangle
    assert max(used[2:]) == min(used[2:]), used                   # steady state from the third build on
    for k in range(3):                                            # a much smaller scene: the big blocks must go
        gpu_ctx.build_device(None, 0, d_tri.data_ptr(), n // 16, scene.materials)
    assert free_before - torch.cuda.mem_get_info()[0] < used[-1] // 4
    gpu_ctx.upload(small)                                         # host path: pool released
    assert abs(free_before - torch.cuda.mem_get_info()[0]) < (8 << 20)


def test_oracle_lbvh_merges_triangle_pairs_and_nothing_else(ra, oracle):
    """CPU only: the leaf rule of the builder (RTGPU_LBVH_MAX_LEAF = 2, triangles only).  No
    sphere leaf holds more than one sphere, no triangle leaf more than two triangles, and no
    emitted node is left with two single-triangle leaves (such a node is merged into one
    leaf of its parent) - except the root, which always stays a node."""
    for name, scene in _scenes(ra).items():
        fs = oracle.build_lbvh(scene.desc, FlatSceneC)
        try:
            nodes = np.frombuffer(_flat_bytes(fs)["nodes"], np.uint32).reshape(-1, 16)
            refs = nodes[:, 12:14]
            leaf = (refs & 0x80000000) != 0
            empty = refs == 0xFFFFFFFF
            sphere = leaf & ~empty & ((refs & 0x40000000) != 0)
            tri = leaf & ~empty & ((refs & 0x40000000) == 0)
            count = ((refs >> 27) & 7) + 1
            assert (count[sphere] == 1).all(), name
            assert (count[tri] <= 2).all(), name
            single_tri = tri & (count == 1)
            both_single = single_tri[:, 0] & single_tri[:, 1]
            assert not both_single[1:].any(), name
            if len(scene.triangles) >= 2 and len(scene.spheres) == 0:
                assert (count[tri] == 2).any(), name                 # something did merge
            # every triangle is referenced exactly once, pairs included
            first = refs & 0x07FFFFFF
            seen = np.zeros(fs.n_tris + 1, np.int64)
            for f, c in zip(first[tri], count[tri]):
                seen[f:f + c] += 1
            assert (seen[:fs.n_tris] == 1).all(), name
        finally:
            oracle.free_flat(fs)
