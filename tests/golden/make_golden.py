#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.

Two kinds of fixture, kept apart on purpose:

  ref_math.json   OUTPUTS OF THE REFERENCE ITSELF: rt::dot / rt::len /
                  rt::normalized / Vec3 operators / rt::transformed_point
                  (code/math/vec.cxx, mat.cxx) compiled from /root/reference
                  into oracle/_ref/libref_math.so (oracle/Makefile target
                  `ref`) and run here on seeded inputs.  This is the only part
                  of the hot path that exists in the reference (SURVEY.md §8a
                  rows a9/a10), so it is the only part that CAN be pinned by it.

  rng_kat.json, isect_kat.json, scatter_kat.json, s0_64x36.npz, checksums.json,
  lit_box_48x48.npz, s1_level5_64x36.npz
                  outputs of THIS repo's CPU oracle (oracle/rt_oracle.c), for
                  which the reference has no counterpart ("parity unpinned",
                  SURVEY.md §8c).  They pin the oracle against regressions and
                  give the GPU tests vectors that do not need the oracle at all.

Run from the repo root in the build container:  python tests/golden/make_golden.py
Float values are stored as u32 bit patterns so the comparison is exact.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import rt_amd as ra  # noqa: E402
from oracle import Oracle, fptr, load_ref_math  # noqa: E402


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32).tolist()


def fnv1a(u32):
    h = np.uint64(0xCBF29CE484222325)
    p = np.uint64(0x100000001B3)
    with np.errstate(over="ignore"):
        for b in np.asarray(u32, np.uint32).ravel().view(np.uint8):
            h = (h ^ np.uint64(b)) * p
    return int(h)


def fnv1a_fast(u32):
    """FNV-1a over 32-bit words (word-wise variant: fast enough for 1M words)."""
    h = 0xCBF29CE484222325
    for w in np.asarray(u32, np.uint32).ravel().tolist():
        h = ((h ^ w) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def make_ref_math():
    ref = load_ref_math()
    if ref is None:
        print("oracle/_ref/libref_math.so absent (no /root/reference?): ref_math.json not regenerated")
        return
    rng = np.random.default_rng(0x52545F31)
    n = 256
    scale = np.float32(10.0) ** rng.integers(-3, 4, size=(n, 1)).astype(np.float32)
    a = (rng.standard_normal((n, 3)).astype(np.float32) * scale).astype(np.float32)
    b = (rng.standard_normal((n, 3)).astype(np.float32) * scale).astype(np.float32)
    s = rng.standard_normal(n).astype(np.float32)
    p4 = rng.standard_normal((n, 4)).astype(np.float32)
    m = rng.standard_normal((n, 16)).astype(np.float32)
    out = {k: [] for k in ("dot", "len", "normalize", "add", "sub", "mul", "scale", "transform")}
    tmp3 = np.zeros(3, np.float32)
    tmp4 = np.zeros(4, np.float32)
    for i in range(n):
        ai, bi = np.ascontiguousarray(a[i]), np.ascontiguousarray(b[i])
        out["dot"].append(bits([ref.ref_dot3(fptr(ai), fptr(bi))])[0])
        out["len"].append(bits([ref.ref_len3(fptr(ai))])[0])
        ref.ref_normalize3(fptr(ai), fptr(tmp3)); out["normalize"].append(bits(tmp3))
        ref.ref_add3(fptr(ai), fptr(bi), fptr(tmp3)); out["add"].append(bits(tmp3))
        ref.ref_sub3(fptr(ai), fptr(bi), fptr(tmp3)); out["sub"].append(bits(tmp3))
        ref.ref_mul3(fptr(ai), fptr(bi), fptr(tmp3)); out["mul"].append(bits(tmp3))
        ref.ref_scale3(fptr(ai), float(s[i]), fptr(tmp3)); out["scale"].append(bits(tmp3))
        pi, mi = np.ascontiguousarray(p4[i]), np.ascontiguousarray(m[i])
        ref.ref_transform_point4(fptr(pi), fptr(mi), fptr(tmp4)); out["transform"].append(bits(tmp4))
    doc = {
        "source": "reference code/math/vec.cxx + mat.cxx compiled from /root/reference "
                  "(clang++ -O2 -ffp-contract=off), run by tests/golden/make_golden.py",
        "a": bits(a), "b": bits(b), "s": bits(s), "p4": bits(p4), "m": bits(m), "out": out,
    }
    json.dump(doc, open(os.path.join(HERE, "ref_math.json"), "w"))
    print("wrote ref_math.json")


def make_rng(o):
    keys = [0, 1, 2, 0xFFFFFFFF, 0x9E3779B9, 0x52545F31, 12345678]
    doc = {
        "pcg_in": keys, "pcg_out": [int(o.lib.rto_pcg(k)) for k in keys],
        "path_key": [[p, s, seed, int(o.lib.rto_path_key(p, s, seed))]
                     for p in (0, 1, 921599, 2073599) for s in (0, 1, 255) for seed in (1, 7)],
        "u01": [[k, c, bits([o.lib.rto_u01(k, c)])[0]] for k in (0, 0xDEADBEEF, 0x52545F31) for c in range(8)],
    }
    json.dump(doc, open(os.path.join(HERE, "rng_kat.json"), "w"))
    print("wrote rng_kat.json")


def isect_cases():
    """Known-answer cases incl. the edge cases of SURVEY.md §8c: tangent,
    t ~ tmin, parallel, degenerate, origin inside, behind."""
    sph = [
        # o, d, (c, r), tmin, tbest
        ([0, 0, 5], [0, 0, -1], [0, 0, 0, 1], 1e-3, 1e30),      # head-on
        ([0, 1, 5], [0, 0, -1], [0, 0, 0, 1], 1e-3, 1e30),      # tangent
        ([0, 1.0000001, 5], [0, 0, -1], [0, 0, 0, 1], 1e-3, 1e30),  # just outside tangent
        ([0, 0, 0], [0, 0, -1], [0, 0, 0, 1], 1e-3, 1e30),      # origin inside -> far root
        ([0, 0, 5], [0, 0, 1], [0, 0, 0, 1], 1e-3, 1e30),       # behind
        ([0, 0, 1.0005], [0, 0, -1], [0, 0, 0, 1], 1e-3, 1e30),  # near root below tmin
        ([0, 0, 5], [0, 0, -1], [0, 0, 0, 1], 1e-3, 3.5),       # tbest cuts it off
        ([13, 2, 3], [-0.9636, -0.1482, -0.2224], [0, -1000, 0, 1000], 1e-3, 1e30),  # ground sphere
    ]
    tri = [
        # o, d, v0, v1, v2, tmin, tbest
        ([0.25, 0.25, 1], [0, 0, -1], [0, 0, 0], [1, 0, 0], [0, 1, 0], 1e-3, 1e30),  # inside
        ([0.25, 0.25, -1], [0, 0, 1], [0, 0, 0], [1, 0, 0], [0, 1, 0], 1e-3, 1e30),  # back face
        ([0, 0, 1], [0, 0, -1], [0, 0, 0], [1, 0, 0], [0, 1, 0], 1e-3, 1e30),        # on vertex
        ([0.5, 0.5, 1], [0, 0, -1], [0, 0, 0], [1, 0, 0], [0, 1, 0], 1e-3, 1e30),    # on edge u+v=1
        ([0.6, 0.6, 1], [0, 0, -1], [0, 0, 0], [1, 0, 0], [0, 1, 0], 1e-3, 1e30),    # outside
        ([0.25, 0.25, 1], [1, 0, 0], [0, 0, 0], [1, 0, 0], [0, 1, 0], 1e-3, 1e30),   # parallel
        ([0.25, 0.25, 1], [0, 0, -1], [0, 0, 0], [1, 0, 0], [2, 0, 0], 1e-3, 1e30),  # degenerate
        ([0.25, 0.25, 0.0005], [0, 0, -1], [0, 0, 0], [1, 0, 0], [0, 1, 0], 1e-3, 1e30),  # t < tmin
        ([0.25, 0.25, 1], [0, 0, -1], [0, 0, 0], [1, 0, 0], [0, 1, 0], 1e-3, 0.5),   # tbest cut
    ]
    slab = [
        # o, d, lo, hi, tmin, tbest
        ([0, 0, 5], [0, 0, -1], [-1, -1, -1], [1, 1, 1], 1e-3, 1e30),
        ([0, 0, 5], [0, 0, 1], [-1, -1, -1], [1, 1, 1], 1e-3, 1e30),
        ([0, 0, 0], [0, 0, -1], [-1, -1, -1], [1, 1, 1], 1e-3, 1e30),    # inside
        ([1, 0, 5], [0, 0, -1], [-1, -1, -1], [1, 1, 1], 1e-3, 1e30),    # on a face plane, d=0 there
        ([2, 0, 5], [0, 0, -1], [-1, -1, -1], [1, 1, 1], 1e-3, 1e30),    # miss, d has zeros
        ([0, 0, 5], [0, 0, -1], [-1, -1, 0], [1, 1, 0], 1e-3, 1e30),     # flat box
        ([0, 0, 5], [0.57735026, 0.57735026, -0.57735026], [-1, -1, -1], [1, 1, 1], 1e-3, 1e30),
        ([0, 0, 5], [0, 0, -1], [-1, -1, -1], [1, 1, 1], 1e-3, 3.0),     # tbest before entry
    ]
    return sph, tri, slab


def norm(d):
    d = np.asarray(d, np.float32)
    return (d / np.sqrt((d * d).sum(dtype=np.float32))).astype(np.float32)


def make_isect(o):
    sph, tri, slab = isect_cases()
    f = lambda v: np.ascontiguousarray(np.asarray(v, np.float32))  # noqa: E731
    t = np.zeros(1, np.float32)
    doc = {"sphere": [], "tri": [], "slab": []}
    for (org, d, s4, tmin, tbest) in sph:
        d = norm(d)
        hit = o.lib.rto_kat_sphere(fptr(f(org)), fptr(d), fptr(f(s4)), tmin, tbest, fptr(t))
        doc["sphere"].append({"o": bits(f(org)), "d": bits(d), "s": bits(f(s4)), "tmin": bits([tmin])[0],
                              "tbest": bits([tbest])[0], "hit": int(hit), "t": bits(t)[0]})
    for (org, d, v0, v1, v2, tmin, tbest) in tri:
        d = norm(d)
        hit = o.lib.rto_kat_tri(fptr(f(org)), fptr(d), fptr(f(v0)), fptr(f(v1)), fptr(f(v2)), tmin, tbest, fptr(t))
        doc["tri"].append({"o": bits(f(org)), "d": bits(d), "v0": bits(f(v0)), "v1": bits(f(v1)), "v2": bits(f(v2)),
                           "tmin": bits([tmin])[0], "tbest": bits([tbest])[0], "hit": int(hit), "t": bits(t)[0]})
    for (org, d, lo, hi, tmin, tbest) in slab:
        d = norm(d)
        hit = o.lib.rto_kat_slab(fptr(f(org)), fptr(d), fptr(f(lo)), fptr(f(hi)), tmin, tbest, fptr(t))
        doc["slab"].append({"o": bits(f(org)), "d": bits(d), "lo": bits(f(lo)), "hi": bits(f(hi)),
                            "tmin": bits([tmin])[0], "tbest": bits([tbest])[0], "hit": int(hit), "tnear": bits(t)[0]})
    json.dump(doc, open(os.path.join(HERE, "isect_kat.json"), "w"))
    print("wrote isect_kat.json")


def scatter_cases():
    """(type, param, d, n, key): per material, front and back faces, grazing
    incidence, total internal reflection, fuzz that absorbs, perfect mirror."""
    LAMB, METAL, DIEL = 0, 1, 2
    cases = []
    for key in (1, 0x52545F31, 0xDEADBEEF):
        cases += [
            (LAMB, 0.0, [0, -1, 0], [0, 1, 0], key),                    # head-on, front face
            (LAMB, 0.0, [0.6, 0.8, 0], [0, 1, 0], key),                 # from behind: normal flips
            (METAL, 0.0, [0.6, -0.8, 0], [0, 1, 0], key),               # perfect mirror, no draw
            (METAL, 0.3, [0.6, -0.8, 0], [0, 1, 0], key),               # fuzzy
            (METAL, 1.0, [0.9999, -0.01414, 0], [0, 1, 0], key),        # grazing + big fuzz: may be absorbed
            (DIEL, 1.5, [0, -1, 0], [0, 1, 0], key),                    # entering, normal incidence
            (DIEL, 1.5, [0.7071068, -0.7071068, 0], [0, 1, 0], key),    # entering at 45 deg
            (DIEL, 1.5, [0.7071068, 0.7071068, 0], [0, 1, 0], key),     # leaving at 45 deg: total internal reflection
            (DIEL, 1.5, [0.3, 0.9539392, 0], [0, 1, 0], key),           # leaving, refracts or reflects by Schlick
            (DIEL, 1.5, [0.9999, -0.01414, 0], [0, 1, 0], key),         # grazing: reflectance near 1
        ]
    for key in range(2, 10):  # enough keys that some grazing fuzzy reflections end below the surface
        cases.append((METAL, 1.0, [0.9999, -0.01414, 0], [0, 1, 0], key))
    return cases


def make_scatter(o):
    import ctypes as C
    out = np.zeros(3, np.float32)
    doc = []
    for (mtype, param, d, n, key) in scatter_cases():
        d, n = norm(d), norm(n)
        draws = C.c_uint32()
        ok = o.lib.rto_kat_scatter(mtype, param, fptr(d), fptr(n), key, 0, fptr(out), C.byref(draws))
        doc.append({"type": mtype, "param": bits([param])[0], "d": bits(d), "n": bits(n), "key": key,
                    "scattered": int(ok), "dir": bits(out), "draws": int(draws.value)})
    json.dump(doc, open(os.path.join(HERE, "scatter_kat.json"), "w"))
    print("wrote scatter_kat.json")


def make_images(o):
    # primary-hit id map + 1 spp radiance of a 64x36 rendering of S0 (depth 8)
    scene, cam = ra.Scene.builtin(0, aspect=64 / 36)
    fs = ra.FlatScene.build(scene)
    st = ra.default_settings(64, 36, spp=1, max_depth=8)
    acc, ids, stats = o.render(fs.c, cam, st)
    st4 = ra.default_settings(64, 36, spp=4, max_depth=8)
    acc4, _, stats4 = o.render(fs.c, cam, st4)
    st6 = ra.default_settings(64, 36, spp=6, max_depth=8, chunk_samples=2)   # three chunks of two
    acc6, _, _ = o.render(fs.c, cam, st6)
    np.savez_compressed(os.path.join(HERE, "s0_64x36.npz"), ids=ids, accum_1spp=acc.view(np.uint32),
                        accum_4spp=acc4.view(np.uint32), accum_6spp_chunk2=acc6.view(np.uint32))
    # checksums of the full C1 frame (1280x720, 1 spp, depth 1 and depth 8) and counters
    scene, cam = ra.Scene.builtin(0, aspect=1280 / 720)
    fs = ra.FlatScene.build(scene)
    doc = {"s0_64x36_stats_1spp": stats, "s0_64x36_stats_4spp": stats4,
           "flat_s0": {"n_nodes": int(fs.c.n_nodes), "max_depth": int(fs.c.max_depth),
                       "nodes_fnv": fnv1a_fast(fs.nodes_u32), "n_spheres": int(fs.c.n_spheres),
                       "n_tris": int(fs.c.n_tris)}}
    for depth in (1, 8):
        st = ra.default_settings(1280, 720, spp=1, max_depth=depth)
        acc, ids, stats = o.render(fs.c, cam, st, threads=8)
        doc[f"c1_1280x720_depth{depth}"] = {
            "ids_fnv": fnv1a_fast(ids), "accum_fnv": fnv1a_fast(acc.view(np.uint32)), "stats": stats}
    json.dump(doc, open(os.path.join(HERE, "checksums.json"), "w"), indent=1)
    print("wrote s0_64x36.npz, checksums.json")


def make_lit_box(o):
    # emissive materials (RTGPU_MAT_EMISSIVE): the lit box of tests/util.py, black sky
    sys.path.insert(0, os.path.dirname(HERE))
    from util import lit_box
    scene, cam = lit_box(ra)
    fs = ra.FlatScene.build(scene)
    st = ra.default_settings(48, 48, spp=4, max_depth=8, sky_bottom=(0, 0, 0), sky_top=(0, 0, 0))
    acc, ids, stats = o.render(fs.c, cam, st)
    np.savez_compressed(os.path.join(HERE, "lit_box_48x48.npz"), ids=ids, accum_4spp=acc.view(np.uint32),
                        stats=np.array([stats[k] for k in sorted(stats)], np.uint64),
                        stats_keys=np.array(sorted(stats)))
    print("wrote lit_box_48x48.npz")


def make_mesh(o):
    # a tree large enough for the packed and four-wide nodes (S1 level 5: 20 k triangles,
    # 11.9 k BVH2 nodes): what the trace loop traverses there is the four-wide array
    scene, cam = ra.Scene.builtin(1, param=5, aspect=64 / 36)
    fs = ra.FlatScene.build(scene)
    assert fs.c.nodes4 and fs.c.nodes48
    st = ra.default_settings(64, 36, spp=2, max_depth=8)
    acc, ids, stats = o.render(fs.c, cam, st, slot_tree=False)
    np.savez_compressed(os.path.join(HERE, "s1_level5_64x36.npz"), ids=ids, accum_2spp=acc.view(np.uint32),
                        stats=np.array([stats[k] for k in sorted(stats)], np.uint64),
                        stats_keys=np.array(sorted(stats)),
                        n_nodes=np.array([fs.c.n_nodes, fs.c.n_nodes4, fs.c.max_stack4], np.uint64),
                        nodes4_fnv=np.array([fnv1a_fast(fs.nodes4_u32)], np.uint64),
                        nodes48_fnv=np.array([fnv1a_fast(fs.nodes48_u32)], np.uint64))
    print("wrote s1_level5_64x36.npz")


def make_mesh_slots(o):
    # the same scene through its slot tree (include/rtgpu_format.h), which the trace loop traverses in
    # preference to the four-wide nodes: the array itself, and the frame with its counters
    scene, cam = ra.Scene.builtin(1, param=5, aspect=64 / 36)
    fs = ra.FlatScene.build(scene)
    slots = o.slot_tree(fs.c)
    assert slots is not None
    st = ra.default_settings(64, 36, spp=2, max_depth=8)
    acc, ids, stats = o.render(fs.c, cam, st, slot_tree=True)
    np.savez_compressed(os.path.join(HERE, "s1_level5_64x36_slots.npz"), ids=ids, accum_2spp=acc.view(np.uint32),
                        stats=np.array([stats[k] for k in sorted(stats)], np.uint64),
                        stats_keys=np.array(sorted(stats)),
                        n_slots=np.array([slots.shape[0]], np.uint64),
                        slots_fnv=np.array([fnv1a_fast(slots)], np.uint64),
                        root=slots[0], first_family=slots[1:int(slots[0, 3] & 0xFFFFFF) + 4])
    print("wrote s1_level5_64x36_slots.npz")


if __name__ == "__main__":
    o = Oracle()
    makers = {"ref_math": make_ref_math, "rng": lambda: make_rng(o), "isect": lambda: make_isect(o),
              "scatter": lambda: make_scatter(o), "images": lambda: make_images(o), "lit_box": lambda: make_lit_box(o),
              "mesh": lambda: make_mesh(o), "mesh_slots": lambda: make_mesh_slots(o)}
    for name in (sys.argv[1:] or list(makers)):   # all of them, or the ones named
        makers[name]()
