"""Exact analytic checks of the trace loop ("furnace" tests), on the oracle (CPU) and on the GPU.

The reference holds no renderer, so the oracle is unpinned for the hot path; what
can be pinned independently of BOTH implementations is physics with an exact answer:
  * a convex lambertian sphere of albedo 1/2 under a uniform white sky: every path is
    sky x 1/2 - each pixel that only sees the sphere sums to spp/2 EXACTLY, each pixel that only
    sees the sky to spp;
  * a glass sphere of albedo 1 under the same sky: every path that ends carries exactly 1, a path
    that runs out of segments 0 - sums are integers, spp everywhere once the depth is large;
  * a perfect mirror sphere under the gradient sky: the pixel is the sky seen along the reflected
    direction - recomputed here in f64 from the camera, the jitter draws and the sphere."""
import numpy as np
import pytest

from util import make_scene

WHITE = dict(sky_bottom=(1, 1, 1), sky_top=(1, 1, 1))
W = H = 96


def _cam(ra):
    return ra.look_at((0, 0, 4), (0, 0, 0), vfov=40.0, aspect=1.0)


def _disk_masks():
    """Pixels whose whole footprint lies inside / outside the sphere's silhouette (r=1 at the origin seen from z=4)."""
    half = 4.0 * np.tan(np.radians(20.0))                     # half-height of the image plane through the origin
    ys, xs = np.mgrid[0:H, 0:W]
    px = ((xs + 0.5) / W * 2 - 1) * half
    py = ((ys + 0.5) / H * 2 - 1) * half
    r = np.hypot(px, py)
    silhouette = 1.0 / np.sqrt(1 - 1 / 16.0) * (4.0 - 0.25) / 4.0 * 1.0   # tangent cone cut by the plane z=0: ~0.968
    silhouette = 4.0 * np.tan(np.arcsin(1 / 4.0))             # = 1.0328 on the plane through the origin
    pix = 2 * half / W
    return r < silhouette - 2 * pix, r > silhouette + 2 * pix


def _furnace(ra, render):
    inside, outside = _disk_masks()
    assert inside.sum() > 1000 and outside.sum() > 1000
    cam = _cam(ra)
    # 1. lambertian, albedo 1/2
    spp = 8
    scene = make_scene(ra, spheres=[(0, 0, 0, 1, 0)], materials=[(0, .5, .5, .5, 0)])
    acc, ids = render(scene, cam, ra.default_settings(W, H, spp=spp, max_depth=8, **WHITE))
    assert np.all(acc[inside][:, :3] == np.float32(spp * 0.5)), "lambertian sphere in a white furnace"
    assert np.all(acc[outside][:, :3] == np.float32(spp)) and np.all(ids[inside] == 0) and np.all(ids[outside] == ra.NO_HIT)
    edge = acc[~inside & ~outside][:, 0]
    assert np.all((edge >= spp * 0.5) & (edge <= spp)) and np.all(edge * 2 == np.round(edge * 2))
    # ... and a path of ONE segment never reaches the sky
    acc1, _ = render(scene, cam, ra.default_settings(W, H, spp=spp, max_depth=1, **WHITE))
    assert np.all(acc1[inside][:, :3] == 0) and np.all(acc1[outside][:, :3] == np.float32(spp))
    # 2. glass, albedo 1: integer sums; exhausted paths vanish as the depth grows
    glass = make_scene(ra, spheres=[(0, 0, 0, 1, 0)], materials=[(2, 1, 1, 1, 1.5)])
    lost = {}
    for depth in (2, 3, 4, 64):
        a, _ = render(glass, cam, ra.default_settings(W, H, spp=spp, max_depth=depth, **WHITE))
        assert np.all(a[..., :3] == np.round(a[..., :3])) and np.all(a[..., :3] <= spp) and np.all(a[..., :3] >= 0)
        assert np.all(a[..., 0] == a[..., 1]) and np.all(a[..., 1] == a[..., 2])
        assert np.all(a[outside][:, :3] == np.float32(spp))
        lost[depth] = int(spp * W * H - a[..., 0].astype(np.float64).sum())   # paths that ran out of segments
    assert lost[2] > lost[3] > lost[4] >= 0 and lost[64] == 0, lost
    return cam


def _mirror_expected(ra, oracle, cam, st):
    """f64 restatement of ONE-bounce mirror paths: camera ray through the jittered film point, nearest
    root of the sphere quadratic, reflection, sky gradient along the reflected direction."""
    L = oracle.lib
    ys, xs = np.mgrid[0:H, 0:W]
    pix = (ys * W + xs).astype(np.uint32)
    jx = np.zeros((H, W)); jy = np.zeros((H, W))
    for y in range(H):
        for x in range(W):
            key = L.rto_path_key(int(pix[y, x]), st.sample_offset, st.seed)
            jx[y, x] = L.rto_u01(key, 0)
            jy[y, x] = L.rto_u01(key, 1)
    o = np.array(list(cam.origin), np.float64)
    ll, hor, ver = (np.array(list(v), np.float64) for v in (cam.lower_left, cam.horizontal, cam.vertical))
    sx, sy = (xs + jx) / W, (ys + jy) / H
    d = ll + sx[..., None] * hor + sy[..., None] * ver - o
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    b = (d * o).sum(-1)
    disc = b * b - ((o * o).sum() - 1.0)
    hit = disc > 0
    t = -b - np.sqrt(np.where(hit, disc, 0))
    p = o + t[..., None] * d
    n = p / np.linalg.norm(p, axis=-1, keepdims=True)
    r = d - 2 * (d * n).sum(-1, keepdims=True) * n
    r /= np.linalg.norm(r, axis=-1, keepdims=True)
    out_dir = np.where(hit[..., None], r, d)
    tt = 0.5 * (out_dir[..., 1] + 1.0)
    bottom, top = np.array(list(st.sky_bottom), np.float64), np.array(list(st.sky_top), np.float64)
    return (1 - tt)[..., None] * bottom + tt[..., None] * top, hit, disc


def _mirror(ra, oracle, render):
    cam = _cam(ra)
    st = ra.default_settings(W, H, spp=1, max_depth=8, seed=21)
    scene = make_scene(ra, spheres=[(0, 0, 0, 1, 0)], materials=[(1, 1, 1, 1, 0.0)])
    acc, ids = render(scene, cam, st)
    want, hit, disc = _mirror_expected(ra, oracle, cam, st)
    sure = np.abs(disc) > 1e-4                                  # away from the silhouette, where f32 and f64 may disagree on hit / miss
    assert np.array_equal(ids[sure] == 0, hit[sure])
    err = np.abs(acc[..., :3].astype(np.float64) - want)[sure]
    # f32 chain against f64: a few ulp of values <= 1; grazing hits amplify them by ~1/sqrt(discriminant)
    assert err.max() < 1e-4 and np.median(err) < 1e-6, (err.max(), np.median(err))
    assert (hit & sure).sum() > 1500


def test_furnace_on_the_oracle(ra, oracle):
    def render(scene, cam, st):
        fs = ra.FlatScene.build(scene)            # keep it alive across the call: its arrays are freed with it
        a, i, _ = oracle.render(fs.c, cam, st, threads=8)
        return a, i
    _furnace(ra, render)
    _mirror(ra, oracle, render)


@pytest.mark.gpu
def test_furnace_on_the_gpu(ra, oracle, gpu_ctx):
    def render(scene, cam, st):
        gpu_ctx.upload(scene)
        a, i, _ = gpu_ctx.render(cam, st)
        return a, i
    _furnace(ra, render)
    _mirror(ra, oracle, render)
