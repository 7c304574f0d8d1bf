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


def lit_box(ra, light_power=8.0, light_mat=3):
    """A closed room of ten triangles lit by a two-triangle ceiling light and a small
    emissive sphere, with a glass and a metal sphere inside (the "Lights" item of the
    reference's plan, notes/plan.md:16).  Materials: 0 white, 1 red, 2 green lambertian,
    3 light (emissive, albedo * light_power), 4 glass, 5 metal, 6 warm emissive.
    Returns (scene, camera for a square image)."""
    def quad(a, b, c, d, m):
        return [(a, b, c, m), (a, c, d, m)]
    L, H = 2.0, 2.0
    tris = []
    tris += quad((-L, 0, -L), (L, 0, -L), (L, 0, L), (-L, 0, L), 0)            # floor
    tris += quad((-L, 2 * H, -L), (-L, 2 * H, L), (L, 2 * H, L), (L, 2 * H, -L), 0)  # ceiling
    tris += quad((-L, 0, -L), (-L, 2 * H, -L), (L, 2 * H, -L), (L, 0, -L), 0)  # back wall
    tris += quad((-L, 0, -L), (-L, 0, L), (-L, 2 * H, L), (-L, 2 * H, -L), 1)  # left, red
    tris += quad((L, 0, -L), (L, 2 * H, -L), (L, 2 * H, L), (L, 0, L), 2)      # right, green
    tris += quad((-0.7, 2 * H - 0.01, -0.7), (0.7, 2 * H - 0.01, -0.7), (0.7, 2 * H - 0.01, 0.7),
                 (-0.7, 2 * H - 0.01, 0.7), light_mat)                          # ceiling light
    spheres = [(-0.9, 0.8, -0.5, 0.8, 5), (0.9, 0.7, 0.6, 0.7, 4), (0.0, 0.25, 1.2, 0.25, 6)]
    mats = [(0, .73, .73, .73, 0), (0, .65, .05, .05, 0), (0, .12, .45, .15, 0),
            (3, 1.0, 0.9, 0.8, light_power), (2, 1, 1, 1, 1.5), (1, .8, .8, .8, 0.05), (3, 1.0, 0.5, 0.2, 2.0)]
    cam = ra.look_at((0, 2, 7.5), (0, 2, 0), vfov=38.0, aspect=1.0)
    return make_scene(ra, spheres=spheres, triangles=tris, materials=mats), cam
