"""Shared helpers of the test-suite."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def f32(bits):
    return np.asarray(bits, np.uint32).view(np.float32)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def fnv1a_words(u32):
    h = 0xCBF29CE484222325
    for w in np.asarray(u32, np.uint32).ravel().tolist():
        h = ((h ^ w) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def assert_bit_equal(a, b, what=""):
    a, b = bits(a), bits(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bad = a != b
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} words differ, first at {np.argwhere(bad)[0]}"


def psnr(a, b, peak=1.0):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return float("inf") if mse == 0 else 10.0 * np.log10(peak * peak / mse)


def make_scene(ra, spheres=(), triangles=(), materials=()):
    """spheres: (cx,cy,cz,r,mat); triangles: (v0,v1,v2,mat); materials: (type,r,g,b,param)."""
    s = np.zeros(len(spheres), ra.SPHERE_DTYPE)
    for i, (x, y, z, r, m) in enumerate(spheres):
        s[i] = ((x, y, z), r, m)
    t = np.zeros(len(triangles), ra.TRIANGLE_DTYPE)
    for i, (a, b, c, m) in enumerate(triangles):
        t[i] = (a, b, c, m)
    mm = np.zeros(len(materials), ra.MATERIAL_DTYPE)
    for i, (ty, r, g, b, p) in enumerate(materials):
        mm[i] = ((r, g, b), p, ty, (0, 0, 0))
    return ra.Scene(s, t, mm)
