"""The oracle's frame loop against its golden fixtures and against itself
(BVH vs exhaustive search, threads, crops, tiles, progressive frames)."""
import numpy as np
import pytest

from util import GOLDEN, assert_bit_equal, bits, fnv1a_words, golden_json, make_scene


@pytest.fixture(scope="module")
def s0_small(ra):
    scene, cam = ra.Scene.builtin(0, aspect=64 / 36)
    return scene, cam, ra.FlatScene.build(scene)


def test_golden_crop(ra, oracle, s0_small):
    _, cam, fs = s0_small
    g = np.load(f"{GOLDEN}/s0_64x36.npz")
    chk = golden_json("checksums.json")
    acc, ids, stats = oracle.render(fs.c, cam, ra.default_settings(64, 36, spp=1, max_depth=8))
    assert np.array_equal(ids, g["ids"])
    assert np.array_equal(bits(acc), g["accum_1spp"])
    assert stats == chk["s0_64x36_stats_1spp"]
    acc4, _, stats4 = oracle.render(fs.c, cam, ra.default_settings(64, 36, spp=4, max_depth=8))
    assert np.array_equal(bits(acc4), g["accum_4spp"])
    assert stats4 == chk["s0_64x36_stats_4spp"]
    acc6, _, _ = oracle.render(fs.c, cam, ra.default_settings(64, 36, spp=6, max_depth=8, chunk_samples=2))
    assert np.array_equal(bits(acc6), g["accum_6spp_chunk2"])


@pytest.mark.parametrize("depth", [1, 8])
def test_c1_frame_checksums(ra, oracle, depth):
    """BASELINE.json configs[0]: default scene, 1280x720, 1 spp, CPU path."""
    scene, cam = ra.Scene.builtin(0, aspect=1280 / 720)
    fs = ra.FlatScene.build(scene)
    acc, ids, stats = oracle.render(fs.c, cam, ra.default_settings(1280, 720, spp=1, max_depth=depth), threads=8)
    g = golden_json("checksums.json")[f"c1_1280x720_depth{depth}"]
    assert fnv1a_words(ids) == g["ids_fnv"]
    assert fnv1a_words(bits(acc)) == g["accum_fnv"]
    assert stats == g["stats"]
    if depth == 1:  # RTIOW: a path that exhausts its depth contributes nothing
        hit = ids != ra.NO_HIT
        assert np.all(acc[hit][:, :3] == 0) and np.all(acc[~hit][:, :3] > 0)


@pytest.mark.parametrize("kind,param", [(0, 0), (1, 3), (2, 2)])
def test_bvh_equals_exhaustive_search(ra, oracle, kind, param):
    """Builder + traversal vs testing every primitive (no BVH): same closest hit
    for every segment of every path, so identical images and identical paths."""
    scene, cam = ra.Scene.builtin(kind, param=param, aspect=48 / 27)
    fs = ra.FlatScene.build(scene)
    st = ra.default_settings(48, 27, spp=2, max_depth=6)
    a, ia, sa = oracle.render(fs.c, cam, st)
    b, ib, sb = oracle.render(fs.c, cam, st, brute_force=True)
    assert np.array_equal(ia, ib)
    assert_bit_equal(a, b, "bvh vs brute force")
    assert (sa["rays"], sa["scatters"]) == (sb["rays"], sb["scatters"])
    assert sb["nodes"] == 0 and sa["nodes"] > 0


def test_threads_crop_and_tiles_do_not_change_pixels(ra, oracle, s0_small):
    _, cam, fs = s0_small
    st = ra.default_settings(64, 36, spp=2, max_depth=8, tile_size=16)
    full, ids, stats = oracle.render(fs.c, cam, st)
    par, ids_p, stats_p = oracle.render(fs.c, cam, st, threads=5)
    assert_bit_equal(full, par, "threads")
    assert stats == stats_p
    crop, ids_c, _ = oracle.render(fs.c, cam, st, crop=(10, 5, 30, 20))
    assert_bit_equal(crop[5:20, 10:30], full[5:20, 10:30], "crop")
    assert np.all(crop[:5] == 0) and np.all(ids_c[20:] == ra.NO_HIT)
    # union of the ranks' tiles == the full frame, for any world size
    for world in (2, 3):
        acc = np.zeros_like(full)
        total_rays = 0
        for r in range(world):
            st_r = ra.default_settings(64, 36, spp=2, max_depth=8, tile_size=16, rank=r, world_size=world)
            acc, _, s = oracle.render(fs.c, cam, st_r, accum=acc)
            total_rays += s["rays"]
        assert_bit_equal(acc, full, f"world {world}")
        assert total_rays == stats["rays"]


def test_progressive_frames(ra, oracle, s0_small):
    """Two 2-spp frames with ACCUMULATE and sample_offset see the same samples as
    one 4-spp frame (sum order differs, so compare to a tolerance) and w = 4."""
    _, cam, fs = s0_small
    one, _, _ = oracle.render(fs.c, cam, ra.default_settings(64, 36, spp=4, max_depth=8))
    a, _, _ = oracle.render(fs.c, cam, ra.default_settings(64, 36, spp=2, max_depth=8))
    b, _, _ = oracle.render(fs.c, cam, ra.default_settings(64, 36, spp=2, max_depth=8, sample_offset=2,
                                                           flags=ra.FLAG_ACCUMULATE), accum=a.copy())
    assert np.all(b[..., 3] == 4.0)
    np.testing.assert_allclose(b, one, rtol=1e-6, atol=1e-6)
    # ...and exactly: frame sums added in order
    f2, _, _ = oracle.render(fs.c, cam, ra.default_settings(64, 36, spp=2, max_depth=8, sample_offset=2))
    assert_bit_equal(b[..., :3], a[..., :3] + f2[..., :3], "accumulate")


def test_chunked_frame_sum(ra, oracle, s0_small):
    """chunk_samples defines the summation tree: frame = ((C0 + C1) + C2)..., each
    chunk the in-order sum of its samples; one chunk == plain sequential sum."""
    _, cam, fs = s0_small
    mk = lambda **kw: ra.default_settings(64, 36, max_depth=8, **kw)
    whole, _, _ = oracle.render(fs.c, cam, mk(spp=5, chunk_samples=2))      # chunks {0,1} {2,3} {4}
    parts = [oracle.render(fs.c, cam, mk(spp=n, sample_offset=o, chunk_samples=8))[0]
             for o, n in ((0, 2), (2, 2), (4, 1))]
    want = (parts[0][..., :3] + parts[1][..., :3]) + parts[2][..., :3]
    assert_bit_equal(whole[..., :3], want, "chunk tree")
    assert np.all(whole[..., 3] == 5.0)
    seq, _, _ = oracle.render(fs.c, cam, mk(spp=5, chunk_samples=5))
    seq2, _, _ = oracle.render(fs.c, cam, mk(spp=5, chunk_samples=99))
    assert_bit_equal(seq, seq2, "one chunk")
    np.testing.assert_allclose(whole, seq, rtol=1e-6, atol=1e-6)


def test_sky_only_and_materials(ra, oracle):
    cam = ra.look_at((0, 0, 5), (0, 0, 0), vfov=40.0, aspect=1.0)
    st = ra.default_settings(32, 32, spp=4, max_depth=8)
    # empty scene: every path is the sky gradient, in [bottom, top] per channel
    fs = ra.FlatScene.build(make_scene(ra))  # keep it alive: .c borrows its memory
    acc, ids, stats = oracle.render(fs.c, cam, st)
    assert np.all(ids == ra.NO_HIT) and stats["rays"] == stats["paths"] == 32 * 32 * 4
    mean = acc[..., :3] / 4
    assert np.all(mean[..., 2] == 1.0) and np.all((mean[..., 0] >= 0.5) & (mean[..., 0] <= 1.0))
    assert np.all(np.diff(mean[:, 16, 0]) < 0)   # row 0 is the bottom: whiter below, bluer above
    # white-furnace style bounds: albedo-1 surfaces never create energy
    for mat in ((0, 1, 1, 1, 0), (1, 1, 1, 1, 0.0), (1, 1, 1, 1, 0.5), (2, 1, 1, 1, 1.5)):
        sc = make_scene(ra, spheres=[(0, 0, 0, 1, 0), (0, -101, 0, 100, 0)], materials=[mat])
        fs = ra.FlatScene.build(sc)
        acc, ids, _ = oracle.render(fs.c, cam, st)
        assert (ids == 0).sum() > 100
        assert np.all(acc[..., :3] / 4 <= 1.0 + 1e-6) and np.all(np.isfinite(acc))
    # a mirror (metal, fuzz 0) facing the camera reflects the sky exactly once: 2 segments per hit path
    sc = make_scene(ra, triangles=[((-50, -50, 0), (50, -50, 0), (0, 50, 0), 0)], materials=[(1, 1, 1, 1, 0)])
    fs = ra.FlatScene.build(sc)
    acc, ids, stats = oracle.render(fs.c, cam, st)
    assert np.all(ids == 0) and stats["rays"] == 2 * stats["paths"] and stats["scatters"] == stats["paths"]


def test_thin_lens_changes_rays_but_not_determinism(ra, oracle):
    scene, cam = ra.Scene.builtin(0, aspect=1.0)
    lens = ra.look_at((13, 2, 3), (0, 0, 0), vfov=20.0, aspect=1.0, aperture=0.1, focus_dist=10.0)
    fs = ra.FlatScene.build(scene)
    st = ra.default_settings(32, 32, spp=2, max_depth=8)
    a, _, _ = oracle.render(fs.c, cam, st)
    b, _, _ = oracle.render(fs.c, lens, st)
    c, _, _ = oracle.render(fs.c, lens, st, threads=3)
    assert not np.array_equal(a, b)
    assert_bit_equal(b, c, "lens determinism")


def test_deep_tree_traversal(ra, oracle):
    """Geometric chain of spheres -> a BVH much deeper than the kernel's LDS
    short stack (16 entries); BVH and exhaustive search still agree."""
    sph = [(float(3.0 ** i), 0.0, 0.0, float(0.25 * 3.0 ** i), 0) for i in range(-16, 40)]
    sc = make_scene(ra, spheres=sph, materials=[(0, .5, .5, .5, 0)])
    fs = ra.FlatScene.build(sc)
    assert fs.c.max_depth >= 28 and oracle.validate_bvh(fs.c)[0] == 0
    cam = ra.look_at((-1e-9, 0.0, 0.0), (1, 0, 0), vfov=30.0, aspect=1.0)
    st = ra.default_settings(16, 16, spp=1, max_depth=3)
    oracle.lib.rto_debug_max_stack(1)
    a, ia, _ = oracle.render(fs.c, cam, st)
    assert oracle.lib.rto_debug_max_stack(1) > 20, "scene does not overflow the kernel's LDS short stack"
    b, ib, _ = oracle.render(fs.c, cam, st, brute_force=True)
    assert np.array_equal(ia, ib) and (ia != ra.NO_HIT).any()
    assert_bit_equal(a, b, "deep tree")


BLACK = dict(sky_bottom=(0, 0, 0), sky_top=(0, 0, 0))


def test_emissive_materials(ra, oracle):
    """RTGPU_MAT_EMISSIVE ("Lights", notes/plan.md:16): a light ends the path with
    albedo * param from either face; nothing else emits under a black sky."""
    from util import lit_box
    scene, cam = lit_box(ra)
    fs = ra.FlatScene.build(scene)
    st = ra.default_settings(48, 48, spp=4, max_depth=8, **BLACK)
    acc, ids, stats = oracle.render(fs.c, cam, st)
    g = np.load(f"{GOLDEN}/lit_box_48x48.npz")
    assert np.array_equal(ids, g["ids"]) and np.array_equal(bits(acc), g["accum_4spp"])
    assert [stats[k] for k in sorted(stats)] == g["stats"].tolist()
    # pixels whose primary hit is the small emissive sphere (prim id 2, material 6) hold
    # exactly spp * albedo * param: throughput 1, and the product is one rounded multiply
    acc1, ids1, _ = oracle.render(fs.c, cam, ra.default_settings(48, 48, spp=1, **BLACK))
    direct = ids1 == 2
    assert direct.sum() > 4
    assert np.all(acc1[direct][:, :3] == np.float32([1.0, 0.5, 0.2]) * np.float32(2.0))
    # BVH traversal vs exhaustive search with lights in the scene
    b, ib, _ = oracle.render(fs.c, cam, st, brute_force=True)
    assert np.array_equal(ids, ib)
    assert_bit_equal(acc, b, "lit box: bvh vs brute force")
    # linearity: scaling every light by 2 (exact in binary floating point) scales the frame by 2
    twice = ra.Scene(scene.spheres, scene.triangles, scene.materials.copy())
    twice.materials["param"][[3, 6]] *= 2
    fs2 = ra.FlatScene.build(twice)  # keep the owner alive while its arrays are read
    acc2, ids2, _ = oracle.render(fs2.c, cam, st)
    assert np.array_equal(ids2, ids)
    assert_bit_equal(acc2[..., :3], acc[..., :3] * np.float32(2), "lights x2")
    # no light, black sky: nothing arrives anywhere; the lit frame is not black
    dark = ra.Scene(scene.spheres, scene.triangles, scene.materials.copy())
    dark.materials["param"][[3, 6]] = 0
    fs0 = ra.FlatScene.build(dark)
    acc0, _, _ = oracle.render(fs0.c, cam, st)
    assert np.all(acc0[..., :3] == 0) and (acc[..., :3] > 0).mean() > 0.2
    # seen from behind (camera above the ceiling light, looking down through it): it still emits
    above = ra.look_at((0, 3.995, 0), (0, 0, 0), up=(0, 0, 1), vfov=20.0, aspect=1.0)
    _, ids_b, _ = oracle.render(fs.c, above, ra.default_settings(8, 8, spp=1, **BLACK))
    assert set(np.unique(ids_b)) <= {3 + 10, 3 + 11}  # the two light triangles (3 spheres come first)


def test_unknown_material_type_is_rejected(ra):
    bad = make_scene(ra, spheres=[(0, 0, 0, 1, 0)], materials=[(4, .5, .5, .5, 0)])
    with pytest.raises(ra.RtgpuError, match="material type"):
        ra.FlatScene.build(bad)


def test_large_tree_fixture(ra, oracle):
    """S1 level 5 (a tree with packed and four-wide nodes) against the committed frame:
    pins the collapse, both packers and the four-wide traversal of the oracle."""
    g = np.load(f"{GOLDEN}/s1_level5_64x36.npz")
    scene, cam = ra.Scene.builtin(1, param=5, aspect=64 / 36)
    fs = ra.FlatScene.build(scene)
    assert [fs.c.n_nodes, fs.c.n_nodes4, fs.c.max_stack4] == g["n_nodes"].tolist()
    assert fnv1a_words(fs.nodes4_u32) == int(g["nodes4_fnv"][0])
    assert fnv1a_words(fs.nodes48_u32) == int(g["nodes48_fnv"][0])
    acc, ids, stats = oracle.render(fs.c, cam, ra.default_settings(64, 36, spp=2, max_depth=8), slot_tree=False)
    assert np.array_equal(ids, g["ids"]) and np.array_equal(bits(acc), g["accum_2spp"])
    assert [stats[k] for k in sorted(stats)] == g["stats"].tolist()


def test_slot_tree_fixture(ra, oracle):
    """The same scene through its slot tree (rtgpu_format.h): the committed array and frame; the frame is the
    four-wide nodes' (the boxes only grow, by the rounding of their planes to binary16), the counters are not."""
    g = np.load(f"{GOLDEN}/s1_level5_64x36_slots.npz")
    g4 = np.load(f"{GOLDEN}/s1_level5_64x36.npz")
    scene, cam = ra.Scene.builtin(1, param=5, aspect=64 / 36)
    fs = ra.FlatScene.build(scene)
    slots = oracle.slot_tree(fs.c)
    assert slots.shape == (int(g["n_slots"][0]), 16) and fnv1a_words(slots) == int(g["slots_fnv"][0])
    assert np.array_equal(slots[0], g["root"])
    st = ra.default_settings(64, 36, spp=2, max_depth=8)
    acc, ids, stats = oracle.render(fs.c, cam, st)                      # builds the tree itself
    acc2, ids2, stats2 = oracle.render(fs.c, cam, st, slots=slots)      # traverses the one given
    assert np.array_equal(ids, g["ids"]) and np.array_equal(bits(acc), g["accum_2spp"])
    assert np.array_equal(ids, ids2) and np.array_equal(bits(acc), bits(acc2)) and stats == stats2
    assert [stats[k] for k in sorted(stats)] == g["stats"].tolist()
    assert np.array_equal(g["ids"], g4["ids"]) and np.array_equal(g["accum_2spp"], g4["accum_2spp"])
    want4 = dict(zip(g4["stats_keys"].tolist(), g4["stats"].tolist()))
    assert stats["nodes"] > want4["nodes"] and stats["rays"] == want4["rays"] and stats["scatters"] == want4["scatters"]


def test_slot_tree_structure(ra, oracle):
    """Every slot of the tree is reached exactly once from the root, every child box (origin + halves) contains the
    four-wide node's box for that child, and the primitives in the slots are the leaf-ordered arrays' records."""
    for kind, param in ((1, 5), (2, 4)):
        scene, _ = ra.Scene.builtin(kind, param=param, aspect=1.5)
        fs = ra.FlatScene.build(scene)
        slots = oracle.slot_tree(fs.c)
        n4 = fs.nodes4_u32.reshape(-1, 24)
        tris = np.frombuffer(ra.api.C.string_at(fs.c.tris, fs.c.n_tris * 48), np.uint32).reshape(-1, 12)
        seen = np.zeros(len(slots), np.int32)
        # walk both trees together: (slot, four-wide node)
        todo = [(0, 0)]
        while todo:
            s, i = todo.pop()
            seen[s] += 1
            w = slots[s]
            org = w[:3].view(np.float32).astype(np.float64)
            family = int(w[3] & 0xFFFFFF)
            byte = [int(w[0] & 0xFF), int(w[1] & 0xFF), int(w[2] & 0xFF), int(w[3] >> 24)]
            at = 0
            for k in range(4):
                cw = n4[i, 6 * k:6 * k + 6]
                ref = int(cw[0] & 0xFF) | int(cw[1] & 0xFF) << 8 | int(cw[2] & 0xFF) << 16 | int(cw[3] & 0xFF) << 24
                halves = w[4 + 3 * k:7 + 3 * k]
                if ref == 0xFFFFFFFF:
                    assert all(int(h) == 0x7C007C00 for h in halves) and byte[k] == at << 4
                    continue
                lo = (halves & 0xFFFF).astype(np.uint16).view(np.float16).astype(np.float64) + org
                hi = (halves >> 16).astype(np.uint16).view(np.float16).astype(np.float64) + org
                box = cw.view(np.float32).astype(np.float64)
                assert (lo <= box[:3]).all() and (hi >= box[3:]).all()
                # ... and tight: within 2^-10 of the offset (a binary16 step) or of the smallest normal half
                assert ((box[:3] - lo) <= np.maximum((box[:3] - org) * 2.0 ** -10, 2.0 ** -14)).all()
                assert ((hi - box[3:]) <= np.maximum((box[3:] - org) * 2.0 ** -10, 2.0 ** -14)).all()
                assert byte[k] >> 4 == at
                if ref & 0x80000000:
                    count = ((ref >> 27) & 7) + 1
                    assert byte[k] & 0xF == 8 | (4 if ref & 0x40000000 else 0) | (count - 1)
                    for j in range(count):
                        seen[family + at + j] += 1
                        if not ref & 0x40000000:
                            assert np.array_equal(slots[family + at + j, :12], tris[(ref & 0x07FFFFFF) + j])
                    at += count
                else:
                    assert byte[k] & 0xF == 0
                    todo.append((family + at, ref))
                    at += 1
        assert (seen == 1).all()
