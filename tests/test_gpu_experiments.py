"""The kernel's experiment switches (profiles/r02/experiments.txt) change HOW the trace loop schedules its
work, never what a path computes: every switch gives the frame and the counters of the product kernel, bit for
bit.  The switches are read at rtgpu_create, so each case renders through a context of its own."""
import numpy as np
import pytest

from util import assert_bit_equal

pytestmark = pytest.mark.gpu

# switch -> kernel mode expected on (S0: the whole scene fits LDS, a mesh whose tree lives in memory: its slot tree - mode 11: the
# blocks of the slot-tree modes hold 24 stack entries per lane in LDS and this tree's bound is 29; mode 12 is covered in
# tests/test_gpu_slot_tree.py - or with RTGPU_SLOT_TREE=0 the four-wide nodes the slot tree is made from)
CASES = [
    ({}, ((7, 8, 13), 11)),                                         # (7, 8 or 13: whichever the build makes the default)
    ({"RTGPU_NO_TINY": "1"}, (2, 11)),                              # the tree in LDS, spheres behind it
    ({"RTGPU_NO_SPHERE_CACHE": "1"}, (2, 11)),                      # the tree in LDS, spheres through L1
    ({"RTGPU_RAYS2": "1"}, (8, 11)),                                # two rays per lane in lock step (scenes in LDS)
    ({"RTGPU_RAYS2": "0", "RTGPU_TINY6": "0"}, (7, 11)),            # one ray per lane, one block of 1024 threads per CU
    ({"RTGPU_RAYS2": "0", "RTGPU_TINY6": "1"}, (13, 11)),           # ... two blocks of 768 threads, the scene in LDS twice
    ({"RTGPU_STACK_LDS": "9"}, (0, 11)),                            # nine stack entries in LDS, the rest in memory: S0's tree too
    ({"RTGPU_REFILL": "0"}, (7, 11)),                               # plain while-while traversal rounds
    ({"RTGPU_NO_MATS_CACHE": "1"}, (7, 11)),                        # trees in memory: materials through L1, not LDS
    ({"RTGPU_TAIL_DRAW": "16"}, (7, 11)),                           # the end of the work queue in draws of 16 items for every wave
    ({"RTGPU_TAIL_DRAW": "32"}, (7, 11)),                           # ... of 32
    ({"RTGPU_TAIL_DRAW": "64"}, (7, 11)),                           # ... of 64, as the whole queue before round 3
]
# the same over the four-wide nodes: their boxes are tighter than the slot tree's halves, so the counters of node visits and
# triangle tests are smaller, and equal among these; frame and ids are the slot tree's
WIDE = {"RTGPU_SLOT_TREE": "0"}
WIDE_CASES = [
    ({}, (7, 6)),
    ({"RTGPU_NODE4_PAD": "1"}, (7, 10)),                            # four-wide nodes in 128-byte slots
    ({"RTGPU_NODE4_PAD": "1", "RTGPU_STACK_LDS": "14"}, (7, 9)),    # ... with the deeper stack entries in memory
    ({"RTGPU_WIDE_CACHE": "1"}, (7, 5)),                            # experiment A: top of the four-wide tree in LDS
]


def _frames(ra, ctx):
    out = []
    for kind, param, w, h, spp in ((0, 0, 160, 90, 4), (1, 5, 128, 72, 3)):
        scene, cam = ra.Scene.builtin(kind, param=param, aspect=w / h)
        ctx.upload(scene)
        st = ra.default_settings(w, h, spp=spp, max_depth=8, seed=11, flags=ra.FLAG_STATS)
        a, ia, sa = ctx.render(cam, st)
        out.append((a, ia, sa.as_dict(), int(ctx.kernel_name().rstrip(">").split(",")[-1])))
    return out


def _frames_with(ra, monkeypatch, env):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ctx = ra.Context(0)
    try:
        return _frames(ra, ctx)
    finally:
        ctx.close()
        for k in env:
            monkeypatch.delenv(k)


@pytest.mark.parametrize("env,modes", CASES[1:], ids=lambda v: "+".join(v) if isinstance(v, dict) else None)
def test_experiment_switches_leave_the_frame_alone(ra, gpu_ctx, monkeypatch, env, modes):
    want = _frames(ra, gpu_ctx)
    assert want[0][3] in CASES[0][1][0] and want[1][3] == CASES[0][1][1]
    got = _frames_with(ra, monkeypatch, env)
    if modes[0] == 7 and "RTGPU_RAYS2" not in env:
        modes = (want[0][3], modes[1])                           # the switch leaves the choice between 7, 8 and 13 alone
    assert tuple(f[3] for f in got) == modes
    for (a, ia, sa, _), (b, ib, sb, _) in zip(got, want):
        assert_bit_equal(a, b, f"{env}")
        assert np.array_equal(ia, ib) and sa == sb


@pytest.mark.parametrize("env,modes", WIDE_CASES, ids=lambda v: ("wide+" + "+".join(v)) if isinstance(v, dict) else None)
def test_four_wide_nodes_give_the_slot_trees_frame(ra, oracle, gpu_ctx, monkeypatch, env, modes):
    want = _frames(ra, gpu_ctx)
    base = _frames_with(ra, monkeypatch, WIDE)
    got = _frames_with(ra, monkeypatch, {**WIDE, **env})
    assert tuple(f[3] for f in got) == (want[0][3], modes[1])
    for (a, ia, sa, _), (b, ib, sb, _), (c, ic, sc, _) in zip(got, base, want):
        assert_bit_equal(a, b, f"{env}")
        assert np.array_equal(ia, ib) and sa == sb               # among the four-wide modes: everything
        assert_bit_equal(a, c, f"{env} vs slot tree")             # against the slot tree: frame and ids ...
        assert np.array_equal(ia, ic)
        assert {k: sa[k] for k in ("paths", "rays", "scatters")} == {k: sc[k] for k in ("paths", "rays", "scatters")}
    assert got[1][2]["nodes"] < want[1][2]["nodes"]               # ... with fewer node visits
    # and the oracle over the four-wide nodes agrees with the kernel over them, counters included
    scene, cam = ra.Scene.builtin(1, param=5, aspect=128 / 72)
    fs = ra.FlatScene.build(scene)
    st = ra.default_settings(128, 72, spp=3, max_depth=8, seed=11, flags=ra.FLAG_STATS)
    o, io, so = oracle.render(fs.c, cam, st, slot_tree=False, threads=8)
    assert_bit_equal(got[1][0], o, "oracle over the four-wide nodes")
    assert np.array_equal(got[1][1], io) and got[1][2] == so


def test_six_waves_per_simd_product_kernel(ra, oracle, monkeypatch):
    """Mode 13 as it ships (no statistics flag: two blocks of 768 threads per CU, 80 registers): S0 in several chunks through
    the thin lens, against the oracle and against mode 7; then a progressive second frame on top."""
    scene, _ = ra.Scene.builtin(0, param=0, aspect=200 / 120)
    cam = ra.look_at((13, 2, 3), (0, 0, 0), vfov=20.0, aspect=200 / 120, aperture=0.1, focus_dist=10.0)
    st = ra.default_settings(200, 120, spp=5, max_depth=8, seed=3)
    st.chunk_samples = 2
    frames = {}
    for tiny6 in ("1", "0"):
        monkeypatch.setenv("RTGPU_TINY6", tiny6)
        monkeypatch.setenv("RTGPU_RAYS2", "0")
        ctx = ra.Context(0)
        try:
            ctx.upload(scene)
            a, ia, _ = ctx.render(cam, st, want_stats=False)
            assert int(ctx.kernel_name().rstrip(">").split(",")[-1]) == (13 if tiny6 == "1" else 7)
            assert ctx.kernel_name().startswith("trace_kernel<0,")
            st2 = st.copy()
            st2.flags = ra.FLAG_ACCUMULATE
            st2.sample_offset = st.spp
            b, _, _ = ctx.render(cam, st2, accum=a.copy(), want_ids=False, want_stats=False)
            frames[tiny6] = (a, ia, b)
            if tiny6 == "1":
                fs = ctx.flat_scene()
                o, io, _ = oracle.render(fs.c, cam, st, threads=8)
                assert_bit_equal(a, o, "mode 13 against the oracle")
                assert np.array_equal(ia, io)
        finally:
            ctx.close()
    assert_bit_equal(frames["1"][0], frames["0"][0], "mode 13 against mode 7")
    assert np.array_equal(frames["1"][1], frames["0"][1])
    assert_bit_equal(frames["1"][2], frames["0"][2], "second progressive frame")
