"""Refit ("builder + refit"): new positions under an unchanged
tree.  CPU: the restatement (oracle rto_refit) reproduces the builders' boxes for
unchanged positions, keeps the topology, yields valid trees for moved geometry and
round-trips.  GPU: rtgpu_refit is byte-identical to the restatement for both
builders, and frames traced through a refitted tree match the oracle's."""
import ctypes as C

import numpy as np
import pytest

from rt_amd.api import FlatSceneC
from test_lbvh import MATS, _flat_bytes, _scenes
from util import bits, make_scene


def _moved(ra, scene, seed=11):
    """The same primitives displaced, scaled and re-assigned materials."""
    rng = np.random.default_rng(seed)
    sph, tri, mat = scene.spheres.copy(), scene.triangles.copy(), scene.materials.copy()
    if len(sph):
        sph["center"] += rng.normal(0, 0.3, (len(sph), 3)).astype(np.float32)
        sph["radius"] *= rng.uniform(0.7, 1.3, len(sph)).astype(np.float32)
        sph["material"] = rng.integers(0, len(mat), len(sph))
    if len(tri):
        shift = rng.normal(0, 0.2, (len(tri), 1, 3)).astype(np.float32)
        for k, name in enumerate(("v0", "v1", "v2")):
            tri[name] += shift[:, 0] + rng.normal(0, 0.02, (len(tri), 3)).astype(np.float32)
        tri["material"] = rng.integers(0, len(mat), len(tri))
    if len(mat):
        mat["albedo"] = rng.uniform(0.2, 0.9, (len(mat), 3)).astype(np.float32)
    return ra.Scene(sph, tri, mat)


def _topology(c):
    nodes = np.frombuffer(_flat_bytes(c)["nodes"], np.uint32).reshape(-1, 16)
    tris = np.frombuffer(_flat_bytes(c)["tris"], np.uint32).reshape(-1, 12)[:c.n_tris]
    meta = np.frombuffer(_flat_bytes(c)["meta"], np.uint32).reshape(-1, 2)[:c.n_spheres]
    return nodes[:, 12:14].copy(), tris[:, 3].copy(), meta[:, 0].copy()


def _assert_same_bytes(a, b, what):
    fa, fb = _flat_bytes(a), _flat_bytes(b)
    for k in fa:
        assert fa[k] == fb[k], f"{what}: {k} differ"
    # four-wide nodes and their source map, where both sides carry them
    if a.nodes4 and b.nodes4:
        assert (a.n_nodes4, a.max_stack4) == (b.n_nodes4, b.max_stack4), what
        grab = lambda ptr, n: bytes((C.c_char * n).from_address(ptr))
        assert grab(a.nodes4, a.n_nodes4 * 96) == grab(b.nodes4, b.n_nodes4 * 96), f"{what}: nodes4 differ"
        assert grab(a.nodes4_src, a.n_nodes4 * 16) == grab(b.nodes4_src, b.n_nodes4 * 16), f"{what}: nodes4_src differ"


def test_refit_with_unchanged_positions_reproduces_the_builders(ra, oracle):
    for name, scene in _scenes(ra).items():
        sah = ra.FlatScene.build(scene)
        lb = oracle.build_lbvh(scene.desc, FlatSceneC)
        for label, flat in (("sah", sah.c), ("lbvh", lb)):
            again = oracle.refit(flat, scene.desc)
            try:
                _assert_same_bytes(again, flat, f"{name} / {label}")
            finally:
                oracle.free_flat(again)
        oracle.free_flat(lb)


def test_refit_keeps_the_topology_and_bounds_the_moved_geometry(ra, oracle):
    for name, scene in _scenes(ra).items():
        flat = ra.FlatScene.build(scene)
        moved = _moved(ra, scene)
        there = oracle.refit(flat.c, moved.desc)
        back = oracle.refit(there, scene.desc)
        try:
            rc, depth = oracle.validate_bvh(there)
            assert rc == 0, f"{name}: validator code {rc}"
            assert bool(there.nodes4) == bool(flat.c.nodes4) and there.n_nodes4 == flat.c.n_nodes4, name
            assert depth == flat.c.max_depth, name
            for a, b in zip(_topology(there), _topology(flat.c)):
                assert np.array_equal(a, b), name
            # the leaf records hold the moved primitives
            tris = np.frombuffer(_flat_bytes(there)["tris"], np.float32).reshape(-1, 12)[:there.n_tris]
            ids = _topology(there)[1] - len(scene.spheres)
            if len(ids):
                assert np.array_equal(bits(tris[:, 0:3]), bits(moved.triangles["v0"][ids])), name
                assert np.array_equal(bits(tris[:, 4:7]), bits(moved.triangles["v1"][ids] - moved.triangles["v0"][ids])), name
            # and refitting back is the identity
            _assert_same_bytes(back, flat.c, name)
        finally:
            oracle.free_flat(there)
            oracle.free_flat(back)


def test_refit_rejects_other_counts(ra, oracle):
    scene = make_scene(ra, spheres=[(0, 0, 0, 1, 0), (3, 0, 0, 1, 1)], materials=MATS)
    fewer = make_scene(ra, spheres=[(0, 0, 0, 1, 0)], materials=MATS)
    flat = ra.FlatScene.build(scene)
    with pytest.raises(RuntimeError):
        oracle.refit(flat.c, fewer.desc)


@pytest.mark.gpu
def test_gpu_refit_matches_the_restatement_byte_for_byte(ra, oracle, gpu_ctx):
    for name, scene in _scenes(ra).items():
        moved = _moved(ra, scene)
        for builder in ("sah", "lbvh"):
            gpu_ctx.upload(scene, builder=builder)
            before = gpu_ctx.flat_scene()
            want = oracle.refit(before.c, moved.desc)       # copies: `before` is refreshed below
            same = oracle.refit(before.c, scene.desc)
            try:
                gpu_ctx.refit(scene)                          # unchanged positions: identity
                _assert_same_bytes(gpu_ctx.flat_scene().c, same, f"{name} / {builder} / unchanged")
                gpu_ctx.refit(moved)
                got = gpu_ctx.flat_scene()
                assert (got.c.n_nodes, got.c.max_depth) == (want.n_nodes, want.max_depth)
                _assert_same_bytes(got.c, want, f"{name} / {builder}")
            finally:
                oracle.free_flat(want)
                oracle.free_flat(same)


@pytest.mark.gpu
def test_frames_through_a_refitted_tree_match_the_oracle(ra, oracle, gpu_ctx):
    for kind, param, w, h, builder in ((0, 0, 160, 90, "sah"), (1, 5, 128, 72, "lbvh"), (2, 3, 128, 72, "sah")):
        scene, cam = ra.Scene.builtin(kind, param=param, aspect=w / h)
        moved = _moved(ra, scene, seed=3)
        gpu_ctx.upload(scene, builder=builder)
        st = ra.default_settings(w, h, spp=2, max_depth=8, flags=ra.FLAG_STATS)
        first, _, _ = gpu_ctx.render(cam, st)
        gpu_ctx.refit(moved)
        acc, ids, stats = gpu_ctx.render(cam, st)
        want, want_ids, want_stats = oracle.render(gpu_ctx.flat_scene().c, cam, st, threads=4)
        assert np.array_equal(ids, want_ids) and np.array_equal(bits(acc), bits(want))
        assert stats.as_dict() == want_stats
        assert not np.array_equal(bits(acc), bits(first))  # the geometry did move
        # a fresh build over the moved geometry sees the same primary hits (ties aside)
        gpu_ctx.upload(moved, builder=builder)
        _, fresh_ids, _ = gpu_ctx.render(cam, st)
        assert (ids == fresh_ids).mean() > 0.999
        # refitting back restores the first frame bit for bit
        gpu_ctx.upload(scene, builder=builder)
        gpu_ctx.refit(moved)
        gpu_ctx.refit(scene)
        again, _, _ = gpu_ctx.render(cam, st)
        assert np.array_equal(bits(again), bits(first))


@pytest.mark.gpu
def test_refit_rejects_bad_input_and_keeps_the_scene(ra, gpu_ctx):
    scene, cam = ra.Scene.builtin(0, aspect=2.0)
    gpu_ctx.upload(scene)
    st = ra.default_settings(64, 32, spp=1)
    want, _, _ = gpu_ctx.render(cam, st)
    fewer = ra.Scene(scene.spheres[:-1], scene.triangles, scene.materials)
    with pytest.raises(ra.RtgpuError, match="counts"):
        gpu_ctx.refit(fewer)
    bad = ra.Scene(scene.spheres.copy(), scene.triangles, scene.materials)
    bad.spheres["radius"][3] = np.nan
    with pytest.raises(ra.RtgpuError):
        gpu_ctx.refit(bad)
    got, _, _ = gpu_ctx.render(cam, st)
    assert np.array_equal(bits(got), bits(want))
    fresh = ra.Context()
    try:
        with pytest.raises(ra.RtgpuError, match="no scene"):
            fresh.refit(scene)
    finally:
        fresh.close()


@pytest.mark.gpu
def test_refit_from_device_arrays(ra, oracle, gpu_ctx):
    """rtgpu_refit_device: the moved primitives handed over as GPU arrays give the bytes of
    the host-pointer refit (and of the restatement); bad input found by the on-device
    check leaves the scene intact."""
    import torch
    for name in ("random mixed", "S1 level 5 (20k tris)"):
        scene = _scenes(ra)[name]
        moved = _moved(ra, scene, seed=5)
        moved = ra.Scene(moved.spheres, moved.triangles, scene.materials)   # materials stay as uploaded
        for builder in ("sah", "lbvh"):
            gpu_ctx.upload(scene, builder=builder)
            want = oracle.refit(gpu_ctx.flat_scene().c, moved.desc)
            try:
                d_sph = torch.from_numpy(moved.spheres.view(np.uint8).copy()).cuda()
                d_tri = torch.from_numpy(moved.triangles.view(np.uint8).copy()).cuda()
                gpu_ctx.refit_device(d_sph.data_ptr() if len(moved.spheres) else None, len(moved.spheres),
                                     d_tri.data_ptr() if len(moved.triangles) else None, len(moved.triangles))
                _assert_same_bytes(gpu_ctx.flat_scene().c, want, f"{name} / {builder} / device arrays")
                # bad input: caught on the GPU before anything is rewritten
                bad = moved.triangles.copy()
                bad["v1"][7, 1] = np.inf
                d_bad = torch.from_numpy(bad.view(np.uint8).copy()).cuda()
                with pytest.raises(ra.RtgpuError, match="non-finite"):
                    gpu_ctx.refit_device(d_sph.data_ptr() if len(moved.spheres) else None, len(moved.spheres),
                                         d_bad.data_ptr(), len(bad))
                with pytest.raises(ra.RtgpuError, match="counts"):
                    gpu_ctx.refit_device(None, 0, d_tri.data_ptr(), len(moved.triangles) - 1)
                _assert_same_bytes(gpu_ctx.flat_scene().c, want, f"{name} / {builder} / after rejected input")
            finally:
                oracle.free_flat(want)
