"""The slot tree (include/rtgpu_format.h; kernel modes 11 and 12) on the GPU: the array slot_tree.hip makes from the
four-wide nodes and the primitive arrays is the oracle's restatement byte for byte - for trees of both builders,
for spheres and triangles in one tree, after a refit - frames traced through it match the oracle's bit for bit with
their counters, and a tree that has no slot form runs on the four-wide nodes."""
import numpy as np
import pytest

from util import GOLDEN, assert_bit_equal, bits, fnv1a_words, make_scene

pytestmark = pytest.mark.gpu


def _mode(ctx):
    return int(ctx.kernel_name().rstrip(">").split(",")[-1])


def _mixed(ra, seed=6, n_sph=1200, n_tri=1200):
    """Overlapping spheres and triangles with all three scattering materials (the soup of test_gpu_parity)."""
    rng = np.random.default_rng(seed)
    mats = [(0, .8, .6, .4, 0.0), (1, .9, .9, .7, 0.2), (2, 1., 1., 1., 1.5), (0, .3, .5, .8, 0.0), (1, .7, .7, .7, 0.0)]
    sph = [(*rng.uniform(-3, 3, 3).tolist(), float(rng.uniform(0.05, 0.5)), int(rng.integers(0, 5))) for _ in range(n_sph)]
    tri = []
    for _ in range(n_tri):
        c = rng.uniform(-3, 3, 3)
        tri.append((tuple(c + rng.normal(0, .4, 3)), tuple(c + rng.normal(0, .4, 3)), tuple(c + rng.normal(0, .4, 3)),
                    int(rng.integers(0, 5))))
    return make_scene(ra, spheres=sph, triangles=tri, materials=mats)


@pytest.mark.parametrize("builder", ["sah", "lbvh"])
def test_device_slot_tree_is_the_oracles(ra, oracle, gpu_ctx, builder):
    scenes = {"S1 level 5": ra.Scene.builtin(1, param=5)[0], "S2 level 4": ra.Scene.builtin(2, param=4)[0],
              "spheres and triangles": _mixed(ra)}
    for name, scene in scenes.items():
        gpu_ctx.upload(scene, builder=builder)
        fs = gpu_ctx.flat_scene()
        assert fs.c.nodes4, name
        got = gpu_ctx.debug_slot_tree()
        want = oracle.slot_tree(fs.c)
        assert got is not None and want is not None and got.shape == want.shape, name
        assert np.array_equal(got, want), f"{name} / {builder}: slot {np.nonzero((got != want).any(axis=1))[0][:5]} differ"
        if name == "S1 level 5" and builder == "sah":   # ... and the committed one
            g = np.load(f"{GOLDEN}/s1_level5_64x36_slots.npz")
            assert got.shape[0] == int(g["n_slots"][0]) and fnv1a_words(got) == int(g["slots_fnv"][0])


def test_slot_tree_fixture_without_oracle(ra, gpu_ctx):
    """Committed vectors only: S1 level 5 through its slot tree (kernel mode 11: the stack bound of this tree, 29 entries,
    is above the 24 a lane of these modes has in LDS)."""
    g = np.load(f"{GOLDEN}/s1_level5_64x36_slots.npz")
    scene, cam = ra.Scene.builtin(1, param=5, aspect=64 / 36)
    gpu_ctx.upload(scene)
    a, ids, st = gpu_ctx.render(cam, ra.default_settings(64, 36, spp=2, max_depth=8, flags=ra.FLAG_STATS))
    assert _mode(gpu_ctx) == 11
    assert np.array_equal(ids, g["ids"]) and np.array_equal(bits(a), g["accum_2spp"])
    assert [st.as_dict()[k] for k in sorted(st.as_dict())] == g["stats"].tolist()


@pytest.mark.parametrize("builder", ["sah", "lbvh"])
def test_frames_through_the_slot_tree(ra, oracle, gpu_ctx, builder):
    """Spheres (their ids and materials ride in the slot), triangles in leaves of one to four, glass and metal; the
    LDS-only stack (mode 12) and, with a shallow LDS share, the stack that spills to memory (mode 11)."""
    scene = _mixed(ra)
    cam = ra.look_at((0.5, 1.0, 9.0), (0, 0, 0), vfov=50.0, aspect=1.5)
    st = ra.default_settings(120, 80, spp=3, max_depth=6, seed=5, flags=ra.FLAG_STATS)
    gpu_ctx.upload(scene, builder=builder)
    a, ia, sa = gpu_ctx.render(cam, st)
    assert _mode(gpu_ctx) in (11, 12)
    b, ib, sb = oracle.render(gpu_ctx.flat_scene().c, cam, st, threads=8)
    assert_bit_equal(a, b, f"mixed scene / {builder}")
    assert np.array_equal(ia, ib) and sa.as_dict() == sb
    assert sb["sphere_tests"] > 0 and sb["tri_tests"] > 0 and len(np.unique(ia)) > 200


def test_rounds_of_chunks_over_the_slot_tree(ra, oracle, gpu_ctx):
    """132 spp in chunks of two = 66 chunks = two rounds: the slot-tree kernels derive an item's last sample from its chunk and the
    round's first chunk, and write the first sample's primary hit in the first round only."""
    scene, cam = ra.Scene.builtin(1, param=5, aspect=1.5)
    st = ra.default_settings(60, 40, spp=132, max_depth=6, chunk_samples=2, seed=4, flags=ra.FLAG_STATS)
    gpu_ctx.upload(scene)
    a, ia, sa = gpu_ctx.render(cam, st)
    assert _mode(gpu_ctx) == 11
    b, ib, sb = oracle.render(gpu_ctx.flat_scene().c, cam, st, threads=8)
    assert_bit_equal(a, b, "66 chunks over the slot tree")
    assert np.array_equal(ia, ib) and sa.as_dict() == sb


def test_shallow_tree_keeps_its_whole_stack_in_lds(ra, oracle, gpu_ctx):
    """A tree whose stack bound fits the LDS share of a slot-tree block runs mode 12 (no overflow stack): 2304 spheres on a grid
    (four-wide stack bound 17 + 4), every material kind, counters included; frames with and without statistics agree (their kernels
    run five and four blocks per CU)."""
    mats = [(0, .8, .3, .3, 0.0), (1, .8, .8, .8, 0.1), (2, 1., 1., 1., 1.5)]
    sph = [(float(x) - 24.0, float(y) - 24.0, 0.0, 0.3, (x + y) % 3) for x in range(48) for y in range(48)]
    scene = make_scene(ra, spheres=sph, materials=mats)
    cam = ra.look_at((3.0, -30.0, 14.0), (0, 0, 0), vfov=60.0, aspect=1.5)
    st = ra.default_settings(150, 100, spp=4, max_depth=6, seed=9, flags=ra.FLAG_STATS)
    gpu_ctx.upload(scene)
    fs = gpu_ctx.flat_scene()
    assert fs.c.nodes4 and fs.c.max_stack4 + 4 <= 24
    a, ia, sa = gpu_ctx.render(cam, st)
    assert _mode(gpu_ctx) == 12
    b, ib, sb = oracle.render(fs.c, cam, st, threads=8)
    assert_bit_equal(a, b, "shallow tree")
    assert np.array_equal(ia, ib) and sa.as_dict() == sb and sb["sphere_tests"] > 0
    st.flags = 0
    a2, ia2, _ = gpu_ctx.render(cam, st)
    assert_bit_equal(a2, a, "product kernel against its counting variant")
    assert np.array_equal(ia2, ia)


def test_hybrid_stack_over_the_slot_tree(ra, oracle, monkeypatch):
    monkeypatch.setenv("RTGPU_STACK_LDS", "10")
    ctx = ra.Context(0)
    try:
        scene, cam = ra.Scene.builtin(2, param=4, aspect=1.5)
        st = ra.default_settings(120, 80, spp=2, max_depth=8, seed=2, flags=ra.FLAG_STATS)
        ctx.upload(scene)
        a, ia, sa = ctx.render(cam, st)
        assert _mode(ctx) == 11
        b, ib, sb = oracle.render(ctx.flat_scene().c, cam, st, threads=8)
        assert_bit_equal(a, b, "hybrid stack")
        assert np.array_equal(ia, ib) and sa.as_dict() == sb
    finally:
        ctx.close()


def test_slot_tree_follows_a_refit(ra, oracle, gpu_ctx):
    from test_refit import _moved
    scene, cam = ra.Scene.builtin(1, param=5, aspect=1.5)
    moved = _moved(ra, scene)
    st = ra.default_settings(96, 64, spp=2, max_depth=6, flags=ra.FLAG_STATS)
    gpu_ctx.upload(scene)
    first, _, _ = gpu_ctx.render(cam, st)
    before = gpu_ctx.debug_slot_tree()
    for how in ("host", "device"):
        gpu_ctx.upload(scene)
        gpu_ctx.render(cam, st)                              # the slot tree of the first geometry exists
        if how == "host":
            gpu_ctx.refit(moved)
        else:
            import torch
            tri = torch.from_numpy(moved.triangles.view(np.uint8).copy()).cuda()
            gpu_ctx.refit_device(None, 0, tri.data_ptr(), len(moved.triangles))
        a, ia, sa = gpu_ctx.render(cam, st)
        assert _mode(gpu_ctx) == 11
        fs = gpu_ctx.flat_scene()
        got, want = gpu_ctx.debug_slot_tree(), oracle.slot_tree(fs.c)
        assert np.array_equal(got, want) and not np.array_equal(got, before), how
        b, ib, sb = oracle.render(fs.c, cam, st, threads=8)
        assert_bit_equal(a, b, f"after a {how} refit")
        assert np.array_equal(ia, ib) and sa.as_dict() == sb
        assert not np.array_equal(bits(a), bits(first))


def test_refit_before_the_first_frame_of_a_new_scene(ra, oracle, gpu_ctx):
    """The layout of a slot tree (kept for the in-place refresh after a refit) belongs to ONE scene: a scene uploaded after it and
    refitted before any frame was traced gets a slot tree of its own, not a refresh into the old layout."""
    from test_refit import _moved
    first, cam = ra.Scene.builtin(1, param=5, aspect=1.5)
    second, cam2 = ra.Scene.builtin(2, param=4, aspect=1.5)
    st = ra.default_settings(60, 40, spp=1, max_depth=4, flags=ra.FLAG_STATS)
    gpu_ctx.upload(first)
    gpu_ctx.render(cam, st)                       # the first scene's slot tree and layout exist
    gpu_ctx.upload(second)
    gpu_ctx.refit(_moved(ra, second))             # ... and must not be refreshed with the second scene's boxes
    fs = gpu_ctx.flat_scene()
    got, want = gpu_ctx.debug_slot_tree(), oracle.slot_tree(fs.c)
    assert got is not None and np.array_equal(got, want)
    a, ia, sa = gpu_ctx.render(cam2, st)
    b, ib, sb = oracle.render(fs.c, cam2, st, threads=8)
    assert_bit_equal(a, b, "second scene after a refit")
    assert np.array_equal(ia, ib) and sa.as_dict() == sb
    # two refits in a row without a frame between them, then back: the refresh is idempotent
    gpu_ctx.refit(second)
    gpu_ctx.refit(_moved(ra, second, seed=3))
    gpu_ctx.refit(second)
    assert np.array_equal(gpu_ctx.debug_slot_tree(), oracle.slot_tree(gpu_ctx.flat_scene().c))


def test_tree_without_a_slot_form_runs_on_the_four_wide_nodes(ra, oracle, gpu_ctx):
    """A scene 400 000 units wide: the top boxes are farther from their origin than binary16 reaches, so there is no slot
    tree; kernel and oracle both traverse the four-wide nodes and agree, counters included."""
    rng = np.random.default_rng(3)
    tri = []
    for _ in range(3000):
        c = rng.uniform(-2e5, 2e5, 3)
        tri.append((tuple(c + rng.normal(0, 3e3, 3)), tuple(c + rng.normal(0, 3e3, 3)), tuple(c + rng.normal(0, 3e3, 3)), 0))
    scene = make_scene(ra, triangles=tri, materials=[(0, .7, .7, .7, 0)])
    cam = ra.look_at((0, 0, 6e5), (0, 0, 0), vfov=50.0, aspect=1.5)
    st = ra.default_settings(96, 64, spp=2, max_depth=4, flags=ra.FLAG_STATS)
    gpu_ctx.upload(scene)
    fs = gpu_ctx.flat_scene()
    assert fs.c.nodes4 and oracle.slot_tree(fs.c) is None and gpu_ctx.debug_slot_tree() is None
    a, ia, sa = gpu_ctx.render(cam, st)
    assert _mode(gpu_ctx) == 6
    b, ib, sb = oracle.render(fs.c, cam, st, threads=8)
    assert_bit_equal(a, b, "no slot form")
    assert np.array_equal(ia, ib) and sa.as_dict() == sb and (ia != 0xFFFFFFFF).sum() > 100


@pytest.mark.parametrize("spp,chunk", [(12, 12), (7, 3), (33, 16)])
def test_idle_lanes_trace_later_samples_at_the_end_of_the_launch(ra, oracle, gpu_ctx, spp, chunk):
    """Once the queue is dry idle lanes take later samples of their wave's unfinished items (trace loop, section 2b) and the
    owner adds their sums in sample order: the frame, the ids and the counters are the oracle's.  Small frames of long items
    (a wave's whole life is the end of the launch), glass and metal in the scene (long paths), with and without statistics,
    and a progressive second frame on top."""
    scene, cam = ra.Scene.builtin(2, param=4, aspect=1.5)
    st = ra.default_settings(72, 48, spp=spp, max_depth=8, seed=5, flags=ra.FLAG_STATS)
    st.chunk_samples = chunk
    gpu_ctx.upload(scene)
    a, ia, sa = gpu_ctx.render(cam, st)
    assert _mode(gpu_ctx) in (11, 12)
    b, ib, sb = oracle.render(gpu_ctx.flat_scene().c, cam, st, threads=8)
    assert_bit_equal(a, b, "long items")
    assert np.array_equal(ia, ib) and sa.as_dict() == sb
    st.flags = 0
    a2, ia2, _ = gpu_ctx.render(cam, st, want_stats=False)
    assert_bit_equal(a2, a, "product kernel against its counting variant")
    assert np.array_equal(ia2, ia)
    st.flags = ra.FLAG_ACCUMULATE
    st.sample_offset = spp
    a3, _, _ = gpu_ctx.render(cam, st, accum=a2.copy(), want_ids=False, want_stats=False)
    st.flags = ra.FLAG_ACCUMULATE | ra.FLAG_STATS
    b3, _, _ = oracle.render(gpu_ctx.flat_scene().c, cam, st, accum=b.copy(), threads=8)
    assert_bit_equal(a3, b3, "second progressive frame")
