"""Host-side logic of the product library that needs no GPU: BVH builder and
flattener, builtin scenes, camera reduction, tile bookkeeping."""
import ctypes as C

import numpy as np
import pytest

from util import bits, fnv1a_words, golden_json, make_scene


def test_flatten_is_deterministic_and_pinned(ra):
    scene, _ = ra.Scene.builtin(0)
    a, b = ra.FlatScene.build(scene), ra.FlatScene.build(scene)
    assert np.array_equal(a.nodes_u32, b.nodes_u32) and np.array_equal(a.tris_u32, b.tris_u32)
    assert np.array_equal(a.spheres_f32.view(np.uint32), b.spheres_f32.view(np.uint32))
    g = golden_json("checksums.json")["flat_s0"]
    assert (a.c.n_nodes, a.c.max_depth, a.c.n_spheres, a.c.n_tris) == (
        g["n_nodes"], g["max_depth"], g["n_spheres"], g["n_tris"])
    assert fnv1a_words(a.nodes_u32) == g["nodes_fnv"]


def test_parallel_build_is_thread_count_independent(ra, oracle, monkeypatch):
    """Large scenes are built by parallel tasks near the root (Builder::build_tasks)
    and threaded scans; the flattened bytes must not depend on that."""
    scene, _ = ra.Scene.builtin(1, param=7)   # 327,682 prims: task splits and threaded scans both engage
    monkeypatch.setenv("RTGPU_BUILD_TASK_DEPTH", "0")
    one = ra.FlatScene.build(scene)
    monkeypatch.setenv("RTGPU_BUILD_TASK_DEPTH", "6")
    many = ra.FlatScene.build(scene)
    monkeypatch.setenv("RTGPU_BUILD_TASK_DEPTH", "2")
    few = ra.FlatScene.build(scene)
    for other in (many, few):
        assert np.array_equal(one.nodes_u32, other.nodes_u32) and np.array_equal(one.tris_u32, other.tris_u32)
        assert one.c.max_depth == other.c.max_depth
    rc, depth = oracle.validate_bvh(many.c)
    assert rc == 0 and depth == many.c.max_depth


@pytest.mark.parametrize("kind,param", [(0, 0), (1, 3), (1, 6), (2, 3)])
def test_bvh_is_structurally_valid(ra, oracle, kind, param):
    scene, _ = ra.Scene.builtin(kind, param=param)
    fs = ra.FlatScene.build(scene)
    rc, depth = oracle.validate_bvh(fs.c)
    assert rc == 0, f"validator code {rc}"
    assert depth == fs.c.max_depth <= 62
    assert fs.c.n_tris == len(scene.triangles) and fs.c.n_spheres == len(scene.spheres)
    # leaf-ordered arrays are a permutation of the input primitives
    if len(scene.spheres):
        ids = np.sort(fs.sphere_meta[:, 0])
        assert np.array_equal(ids, np.arange(len(scene.spheres)))
    if len(scene.triangles):
        ids = np.sort(fs.tris_u32[:, 3])
        assert np.array_equal(ids, len(scene.spheres) + np.arange(len(scene.triangles)))


def test_builtin_scene_sizes(ra):
    s0, _ = ra.Scene.builtin(0)
    assert 450 <= s0.n_prims <= 490 and len(s0.triangles) == 2
    assert s0.spheres[0]["radius"] == 1000.0          # prim id 0 = ground sphere
    s1, _ = ra.Scene.builtin(1, param=4)
    assert len(s1.triangles) == 20 * 4 ** 4 + 2
    s2, _ = ra.Scene.builtin(2, param=2)
    assert len(s2.triangles) == 8 * 20 * 4 ** 2
    assert set(np.unique(s2.materials["type"])) == {0, 1, 2}
    # same seed -> same bytes; other seed -> other layout
    s0b, _ = ra.Scene.builtin(0)
    s0c, _ = ra.Scene.builtin(0, seed=5)
    assert s0.spheres.tobytes() == s0b.spheres.tobytes() and s0.spheres.tobytes() != s0c.spheres.tobytes()


def test_icosphere_is_watertight(ra):
    """Shared edges are bit-identical from both triangles (no vertex cache needed)."""
    s1, _ = ra.Scene.builtin(1, param=3)
    t = s1.triangles[:-2]
    v = np.stack([t["v0"], t["v1"], t["v2"]], 1).view(np.uint32)          # [n,3,3] bit patterns
    edges = {}
    for tri in v:
        for a, b in ((0, 1), (1, 2), (2, 0)):
            key = frozenset((tri[a].tobytes(), tri[b].tobytes()))
            edges[key] = edges.get(key, 0) + 1
    assert set(edges.values()) == {2}


def test_flatten_rejects_bad_input(ra):
    L = ra.lib
    from rt_amd.api import FlatSceneC
    out = FlatSceneC()
    bad_mat = make_scene(ra, spheres=[(0, 0, 0, 1, 3)], materials=[(0, .5, .5, .5, 0)])
    assert L.rtgpu_flatten_scene_host(C.byref(bad_mat.desc), C.byref(out)) != 0
    assert b"material" in L.rtgpu_last_error(None)
    nan = make_scene(ra, spheres=[(float("nan"), 0, 0, 1, 0)], materials=[(0, .5, .5, .5, 0)])
    assert L.rtgpu_flatten_scene_host(C.byref(nan.desc), C.byref(out)) != 0
    neg = make_scene(ra, spheres=[(0, 0, 0, -1, 0)], materials=[(0, .5, .5, .5, 0)])
    assert L.rtgpu_flatten_scene_host(C.byref(neg.desc), C.byref(out)) != 0
    nomat = make_scene(ra, spheres=[(0, 0, 0, 1, 0)])
    assert L.rtgpu_flatten_scene_host(C.byref(nomat.desc), C.byref(out)) != 0
    inf_tri = make_scene(ra, triangles=[((0, 0, 0), (1, 0, 0), (0, float("inf"), 0), 0)],
                         materials=[(0, .5, .5, .5, 0)])
    assert L.rtgpu_flatten_scene_host(C.byref(inf_tri.desc), C.byref(out)) != 0


def test_flatten_tiny_scenes(ra, oracle):
    empty = ra.FlatScene.build(make_scene(ra))
    assert empty.c.n_nodes == 1 and empty.nodes_u32[0, 12] == 0xFFFFFFFF == empty.nodes_u32[0, 13]
    one = ra.FlatScene.build(make_scene(ra, spheres=[(0, 0, 0, 1, 0)], materials=[(0, .5, .5, .5, 0)]))
    assert one.c.n_nodes == 1 and one.nodes_u32[0, 13] == 0xFFFFFFFF
    assert one.nodes_u32[0, 12] == 0x80000000 | 0x40000000  # leaf, sphere, count 1, first 0
    assert oracle.validate_bvh(one.c)[0] == 0
    # 2..4 primitives of one type that SAH keeps together: one leaf under a root with an empty sibling
    for k in (2, 3, 4):
        few = ra.FlatScene.build(make_scene(ra, spheres=[(0.3 * i, 0, 0, 1, 0) for i in range(k)],
                                            materials=[(0, .5, .5, .5, 0)]))
        assert few.c.n_nodes == 1 and few.c.max_depth == 1 and few.nodes_u32[0, 13] == 0xFFFFFFFF
        assert few.nodes_u32[0, 12] == 0x80000000 | 0x40000000 | ((k - 1) << 27)
        assert np.all(np.isinf(few.nodes_f32[0, 6:12])) and oracle.validate_bvh(few.c) == (0, 1)
    # 64 coincident spheres: SAH cannot split them, median split must
    same = ra.FlatScene.build(make_scene(ra, spheres=[(0, 0, 0, 1, 0)] * 64, materials=[(0, .5, .5, .5, 0)]))
    assert oracle.validate_bvh(same.c)[0] == 0
    # mixed leaf candidates are split by type
    mixed = ra.FlatScene.build(make_scene(
        ra, spheres=[(0, 0, 0, 1, 0), (0.1, 0, 0, 1, 0)],
        triangles=[((0, 0, 0), (1, 0, 0), (0, 1, 0), 0)], materials=[(0, .5, .5, .5, 0)]))
    assert oracle.validate_bvh(mixed.c)[0] == 0


def test_camera_look_at(ra):
    cam = ra.look_at((13, 2, 3), (0, 0, 0), (0, 1, 0), vfov=20.0, aspect=16 / 9, aperture=0.1, focus_dist=10.0)
    frm, at, up = np.array([13, 2, 3.0]), np.zeros(3), np.array([0, 1.0, 0])
    h = np.tan(np.radians(np.float64(np.float32(20.0))) / 2)
    w = (frm - at) / np.linalg.norm(frm - at)
    u = np.cross(up, w); u /= np.linalg.norm(u)
    v = np.cross(w, u)
    hor, ver = 10.0 * np.float64(np.float32(16 / 9)) * 2 * h * u, 10.0 * 2 * h * v
    np.testing.assert_allclose(np.array(cam.horizontal), hor, rtol=2e-7)
    np.testing.assert_allclose(np.array(cam.vertical), ver, rtol=2e-7, atol=1e-7)
    np.testing.assert_allclose(np.array(cam.lower_left), frm - hor / 2 - ver / 2 - 10.0 * w, rtol=2e-7)
    assert list(cam.origin) == [13.0, 2.0, 3.0] and cam.lens_radius == np.float32(0.05)


def test_tile_bookkeeping(ra):
    from rt_amd import dist
    for (w, h, ts, world) in ((1920, 1080, 64, 1), (1920, 1080, 64, 8), (100, 70, 16, 3), (64, 64, 64, 2), (8, 8, 8, 4)):
        tiles = ((w + ts - 1) // ts) * ((h + ts - 1) // ts)
        per = [ra.compact_pixels(w, h, ts, r, world) // (ts * ts) for r in range(world)]
        assert sum(per) == tiles and max(per) - min(per) <= 1 and per[0] == max(per)
        idx = dist.tile_index_map(w, h, ts, world)
        assert len(np.unique(idx)) == w * h             # injective
        assert idx.max() < world * dist.padded_pixels(w, h, ts, world)
        img = np.arange(w * h * 2, dtype=np.float32).reshape(h, w, 2)
        parts = [dist.compact_from_image(img, w, h, ts, r, world) for r in range(world)]
        back = dist.untile_host(np.concatenate(parts), w, h, ts, world)
        assert np.array_equal(back, img)
    assert ra.compact_pixels(64, 64, 64, 5, 4) == 0     # rank >= world


def test_camera_from_matrix_follows_the_reference_row_vector_convention(ra, oracle):
    """rtgpu_camera_from_matrix (rt::Mat4x4, code/math/mat.cxx:12-20: world = p * M): the eye is
    (0,0,0,1) * M and the view direction (0,0,-1,1) * M - eye, both computed with the oracle's
    transform_point, which tests/test_oracle_kat.py pins to the REFERENCE's own
    transformed_point on these same fixture matrices (tests/golden/ref_math.json)."""
    import ctypes as C
    from oracle import fptr
    from util import f32, golden_json
    g = golden_json("ref_math.json")
    mats = f32(g["m"]).reshape(-1, 16)
    out = np.zeros(4, np.float32)

    def xf(p, m):
        oracle.lib.rto_transform_point4(fptr(np.ascontiguousarray(p, np.float32)), fptr(np.ascontiguousarray(m, np.float32)), fptr(out))
        return out.copy()
    as_bytes = lambda cam: bytes((C.c_char * C.sizeof(cam)).from_buffer_copy(cam))
    n = 0
    for m in mats:
        if not np.all(np.isfinite(m)) or np.abs(m).max() > 1e6:
            continue
        eye, at = xf((0, 0, 0, 1), m), xf((0, 0, -1, 1), m)
        up = m[4:7]
        if np.linalg.norm(np.cross(up.astype(np.float64), (eye - at)[:3].astype(np.float64))) < 1e-6:
            continue
        want = ra.look_at(tuple(eye[:3]), tuple(at[:3]), up=tuple(up), vfov=33.0, aspect=1.5, aperture=0.1, focus_dist=7.0)
        got = ra.camera_from_matrix(m.reshape(4, 4), vfov=33.0, aspect=1.5, aperture=0.1, focus_dist=7.0)
        assert as_bytes(got) == as_bytes(want)
        assert list(got.origin) == [m[12], m[13], m[14]]          # row 3 is the translation (mat.cxx:93-102)
        n += 1
    assert n >= 8
    # the textbook camera: rows = right, up, back, eye reproduce rtgpu_camera_look_at exactly
    eye, at = np.array([13.0, 2.0, 3.0]), np.zeros(3)
    w = (eye - at) / np.linalg.norm(eye - at)
    u = np.cross([0, 1, 0], w); u /= np.linalg.norm(u)
    v = np.cross(w, u)
    M = np.zeros((4, 4), np.float32)
    M[0, :3], M[1, :3], M[2, :3], M[3, :3], M[3, 3] = u, v, w, eye, 1
    a, b = ra.camera_from_matrix(M, vfov=20.0), ra.look_at(tuple(eye), tuple(eye - M[2, :3].astype(np.float64)), up=tuple(M[1, :3]), vfov=20.0)
    assert np.allclose(list(a.lower_left), list(b.lower_left), atol=1e-5) and np.allclose(list(a.horizontal), list(b.horizontal), atol=1e-5)
    c = ra.look_at((13, 2, 3), (0, 0, 0), vfov=20.0)
    assert np.allclose(list(a.lower_left), list(c.lower_left), atol=2e-5) and np.allclose(list(a.vertical), list(c.vertical), atol=2e-5)
