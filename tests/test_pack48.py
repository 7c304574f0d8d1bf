"""Packed 48-byte nodes (include/rtgpu_format.h): the product's packer (pack48.h, ordered
integers) against the oracle's independent restatement (float comparisons on candidate
blocks), the format's properties, and where the packed copy exists."""
import ctypes as C

import numpy as np
import pytest

from util import make_scene


def _product_pack(ra, words):
    ra.lib.rtgpu_debug_pack_node48.argtypes = [C.c_void_p, C.c_void_p]
    src = np.ascontiguousarray(words, np.uint32)
    out = np.zeros(12, np.uint32)
    return out if ra.lib.rtgpu_debug_pack_node48(src.ctypes.data, out.ctypes.data) == 0 else None


def _ordered(u):
    u = np.asarray(u, np.uint32).astype(np.int64)
    return np.where(u & 0x80000000, -(u & 0x7FFFFFFF) - 1, u)  # -0 just below +0


def _check_properties(node, packed):
    assert np.array_equal(packed[4:8], node[4:8])                      # exact planes untouched
    c0 = sum(int(packed[k] & 0xFF) << (8 * k) for k in range(4))
    c1 = sum(int(packed[8 + k] & 0xFF) << (8 * k) for k in range(4))
    assert (c0, c1) == (int(node[12]), int(node[13]))
    a, b = _ordered(packed), _ordered(node[:12])
    for k in list(range(4)) + list(range(8, 12)):
        lo = (k % 6) < 3
        d = int(b[k] - a[k]) if lo else int(a[k] - b[k])
        crosses_zero = (int(node[k]) ^ int(packed[k])) >> 31
        assert 0 <= d <= (511 if crosses_zero else 255), (k, hex(int(node[k])), hex(int(packed[k])))
    assert np.all(np.isfinite(packed.view(np.float32)))


def test_packers_agree_on_known_answers_and_random_planes(ra, oracle):
    rng = np.random.default_rng(48)
    specials = np.array([0.0, -0.0, 1e-45, -1e-45, 1.17549435e-38, -1.17549435e-38, 1.0, -1.0, 255.0, 256.0,
                         -3.999, 1e30, -1e30, 3.0e38, -3.0e38], np.float32).view(np.uint32)
    cases = []
    for _ in range(3000):
        node = np.zeros(16, np.uint32)
        kind = rng.integers(0, 4)
        if kind == 0:
            node[:12] = rng.choice(specials, 12)
        elif kind == 1:   # any finite bit pattern
            bits = rng.integers(0, 2**32, 12, dtype=np.uint64).astype(np.uint32)
            bits[(bits & 0x7F800000) == 0x7F800000] &= 0xBFFFFFFF
            node[:12] = bits
        elif kind == 2:   # around zero: low blocks of both signs
            node[:12] = (rng.integers(0, 600, 12).astype(np.uint32) | (rng.integers(0, 2, 12).astype(np.uint32) << 31))
        else:             # ordinary coordinates
            node[:12] = rng.normal(0, 10, 12).astype(np.float32).view(np.uint32)
        node[12:14] = rng.integers(0, 2**32, 2, dtype=np.uint64).astype(np.uint32)
        cases.append(node)
    for node in cases:
        got, want = _product_pack(ra, node), oracle.pack_node48(node)
        assert (got is None) == (want is None), [hex(int(v)) for v in node]
        if got is None:
            continue
        assert np.array_equal(got, want), [hex(int(v)) for v in node]
        _check_properties(node, got)
    # a known answer worked by hand: +1.0f (0x3F800000) carrying byte 0x7F
    node = np.zeros(16, np.uint32)
    node[:12] = np.float32(1.0).view(np.uint32)
    node[12], node[13] = 0x7F7F7F7F, 0x7F7F7F7F
    p = _product_pack(ra, node)
    assert [hex(int(v)) for v in p[:4]] == ["0x3f7fff7f"] * 3 + ["0x3f80007f"]       # lo planes down, hi0.x up
    assert [hex(int(v)) for v in p[8:]] == ["0x3f7fff7f"] + ["0x3f80007f"] * 3       # lo1.z down, hi1 up
    # negative planes are stored with the byte as is; down means larger magnitude
    node[:12] = np.float32(-1.0).view(np.uint32)
    p = _product_pack(ra, node)
    assert hex(int(p[0])) == "0xbf80007f" and hex(int(p[3])) == "0xbf7fff7f"
    # a non-finite plane has no packed form, nor does FLT_MAX pushed upwards
    node[:12] = np.float32(1.0).view(np.uint32)
    node[5] = np.float32(np.inf).view(np.uint32)
    assert _product_pack(ra, node) is None and oracle.pack_node48(node) is None
    node[5] = np.float32(1.0).view(np.uint32)
    node[3] = 0x7F7FFFFF                      # hi0.x = FLT_MAX, byte 0x7F < 0xFF: must move up -> inf
    assert _product_pack(ra, node) is None and oracle.pack_node48(node) is None


def test_flattened_scenes_carry_the_packed_copy_from_the_threshold_up(ra, oracle):
    small = ra.FlatScene.build(ra.Scene.builtin(0)[0])
    assert small.c.n_nodes < ra.NODE48_MIN_NODES and not small.c.nodes48 and len(small.nodes48_u32) == 0
    big = ra.FlatScene.build(ra.Scene.builtin(1, param=5)[0])
    assert big.c.n_nodes >= ra.NODE48_MIN_NODES and big.c.nodes48
    assert oracle.validate_bvh(big.c)[0] == 0          # includes the packed-vs-exact checks
    nodes, packed = big.nodes_u32, big.nodes48_u32
    for i in range(0, len(nodes), 7):
        want = oracle.pack_node48(nodes[i])
        assert want is not None and np.array_equal(packed[i], want), i
        _check_properties(nodes[i], packed[i])
    # a corrupted packed copy is caught by the validator
    bad = packed.copy()
    bad[5, 4] ^= 1
    keep = big.c.nodes48
    big.c.nodes48 = bad.ctypes.data
    assert oracle.validate_bvh(big.c)[0] == 12
    big.c.nodes48 = keep


def test_oracle_traverses_the_packed_nodes_and_sees_the_same_scene(ra, oracle):
    """Packed planes only grow boxes (by < 3.1e-5 relative), so the closest hits are those
    of the exact tree; the count of visited nodes moves by a hair (a different near/far
    order can also save visits)."""
    scene, cam = ra.Scene.builtin(1, param=5, aspect=1.5)
    flat = ra.FlatScene.build(scene)
    st = ra.default_settings(96, 64, spp=2, max_depth=6)
    a, ia, sa = oracle.render(flat.c, cam, st, threads=4)
    keep = flat.c.nodes48
    flat.c.nodes48 = None
    b, ib, sb = oracle.render(flat.c, cam, st, threads=4)
    flat.c.nodes48 = keep
    assert np.array_equal(ia, ib) and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert sa["rays"] == sb["rays"] and abs(sa["nodes"] - sb["nodes"]) < sb["nodes"] * 0.01


def test_four_wide_nodes(ra, oracle):
    """rtgpu_node4: present from the same threshold, structurally valid (every node and
    primitive reached once, boxes nested, stack bound honoured), the oracle's four-wide
    traversal finds the hits of the BVH2 in about half as many node steps, and a corrupted
    node is caught."""
    small = ra.FlatScene.build(ra.Scene.builtin(0)[0])
    assert not small.c.nodes4 and small.c.n_nodes4 == 0
    for kind, param in ((1, 5), (2, 3)):
        scene, cam = ra.Scene.builtin(kind, param=param, aspect=1.5)
        flat = ra.FlatScene.build(scene)
        assert flat.c.nodes4 and flat.c.n_nodes // 3 <= flat.c.n_nodes4 <= flat.c.n_nodes
        assert 3 <= flat.c.max_stack4 <= 3 * flat.c.max_depth
        assert oracle.validate_bvh(flat.c)[0] == 0
        st = ra.default_settings(96, 64, spp=2, max_depth=6)
        oracle.lib.rto_debug_max_stack(1)
        a, ia, sa = oracle.render(flat.c, cam, st, threads=4)
        assert oracle.lib.rto_debug_max_stack(1) <= flat.c.max_stack4
        keep = flat.c.nodes4
        flat.c.nodes4 = None
        b, ib, sb = oracle.render(flat.c, cam, st, threads=4)
        flat.c.nodes4 = keep
        assert np.array_equal(ia, ib) and np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert sa["rays"] == sb["rays"] and 0.4 < sa["nodes"] / sb["nodes"] < 0.7
        assert sa["tri_tests"] < sb["tri_tests"] * 1.05
        # brute force over the same primitives agrees as well
        c, ic, _ = oracle.render(flat.c, cam, st, brute_force=True, threads=4)
        assert np.array_equal(ia, ic) and np.array_equal(a.view(np.uint32), c.view(np.uint32))
    words = flat.nodes4_u32
    bad = words.copy()
    bad[3, 4] = np.float32(-1e30).view(np.uint32)     # child 0 of node 3: hi.y far below its lo
    flat.c.nodes4 = bad.ctypes.data
    assert oracle.validate_bvh(flat.c)[0] in (16, 17)
    bad = words.copy()
    bad[3, 0] ^= 0x01                                  # a reference byte: another node / primitive
    flat.c.nodes4 = bad.ctypes.data
    assert oracle.validate_bvh(flat.c)[0] != 0
    flat.c.nodes4 = keep
