"""The C++ facade (rt_amd/host/rt.hpp, the module a reference maintainer would drop into
code/rt/) driven by its headless front-end stand-in, rt_headless, as a fresh process:
its PPM must be the oracle's resolved frame - one GPU (begin / poll / end), the
progressive loop (rt_render_resolved_or_panic) and the multi-GPU group (--devices)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "rt_amd", "host", "rt_headless")


def _ppm(path, w, h):
    raw = open(path, "rb").read()
    head = f"P6\n{w} {h}\n255\n".encode()
    assert raw.startswith(head) and len(raw) == len(head) + w * h * 3
    return np.frombuffer(raw[len(head):], np.uint8).reshape(h, w, 3)


def _rgb_top_down(rgba8):
    """0xRRGGBBAA, row 0 = bottom -> [h, w, 3] bytes, row 0 = top (what rtgpu_write_ppm writes)."""
    a = rgba8[::-1]
    return np.stack([(a >> 24) & 0xFF, (a >> 16) & 0xFF, (a >> 8) & 0xFF], axis=-1).astype(np.uint8)


@pytest.mark.parametrize("opts,frames", [([], 1), (["--frames", "3"], 3), (["--devices", "0,0,0", "--frames", "2"], 2),
                                         (["--gpus", "1"], 1)])
def test_rt_headless_writes_the_oracles_frame(ra, oracle, tmp_path, opts, frames):
    assert os.path.exists(EXE), "build it with make -C rt_amd/host (graft build() does)"
    w, h, spp, depth = 320, 180, 4, 8
    out = str(tmp_path / "frame.ppm")
    r = subprocess.run([EXE, *opts, "0", str(w), str(h), str(spp), str(depth), "0", out], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    scene, cam = ra.Scene.builtin(0, aspect=np.float32(w) / np.float32(h))
    fs = ra.FlatScene.build(scene)
    acc = np.zeros((h, w, 4), np.float32)
    rays = 0
    for f in range(frames):
        st = ra.default_settings(w, h, spp=spp, max_depth=depth, sample_offset=f * spp, flags=ra.FLAG_ACCUMULATE if f else 0)
        acc, _, s = oracle.render(fs.c, cam, st, accum=acc, threads=16)
        rays += s["rays"]
    assert np.array_equal(_ppm(out, w, h), _rgb_top_down(oracle.resolve(acc)))
    assert f"{rays} rays" in r.stdout, r.stdout


def test_rt_headless_panics_like_the_reference(tmp_path):
    """errf convention (code/first.cpp:27-43): banner on stderr, exit code 1."""
    r = subprocess.run([EXE, str(tmp_path / "missing.obj"), "64", "64", "1", "2", "0", str(tmp_path / "x.ppm")],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "!!! ERROR" in r.stderr


def _basis(eye, at):
    """Camera-to-world Mat4x4 in the reference's row-vector convention (code/math/mat.cxx:12-20,93-102):
    rows right / up / back / eye."""
    eye, at = np.asarray(eye, np.float64), np.asarray(at, np.float64)
    back = (eye - at) / np.linalg.norm(eye - at)
    right = np.cross([0.0, 1.0, 0.0], back)
    right /= np.linalg.norm(right)
    up = np.cross(back, right)
    m = np.zeros((4, 4), np.float32)
    m[0, :3], m[1, :3], m[2, :3], m[3, :3], m[3, 3] = right, up, back, eye, 1.0
    return m


def _matrix_arg(m):
    return ",".join(repr(float(x)) for x in m.reshape(16))      # repr of a float32 round-trips through strtof


def test_rt_headless_camera_from_basis(ra, oracle, tmp_path):
    """rt_camera_from_basis (rt.hpp; Mat4x4 rows right / up / back / eye, code/math/mat.cxx:12-20) executed by
    the facade's own driver: the frame is the oracle's through rtgpu_camera_from_matrix of the same floats."""
    w, h, spp, depth = 256, 144, 3, 6
    m = _basis((9.0, 3.5, -4.0), (0.0, 0.5, 0.0))
    out = str(tmp_path / "m.ppm")
    r = subprocess.run([EXE, "--camera-matrix", _matrix_arg(m), "--vfov", "27.5", "0", str(w), str(h), str(spp), str(depth), "0", out],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    scene, _ = ra.Scene.builtin(0, aspect=np.float32(w) / np.float32(h))
    cam = ra.camera_from_matrix(m, vfov=27.5, aspect=np.float32(w) / np.float32(h))
    fs = ra.FlatScene.build(scene)
    acc, _, _ = oracle.render(fs.c, cam, ra.default_settings(w, h, spp=spp, max_depth=depth), threads=16)
    assert np.array_equal(_ppm(out, w, h), _rgb_top_down(oracle.resolve(acc)))
    default_cam = _ppm(out, w, h).copy()
    r = subprocess.run([EXE, "0", str(w), str(h), str(spp), str(depth), "0", out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and not np.array_equal(_ppm(out, w, h), default_cam)   # the matrix did move the camera


def test_rt_headless_parses_an_obj_from_a_buffer(ra, oracle, tmp_path):
    """rt_scene_parse_obj_or_panic(Buffer) (code/first.hpp:29-32): the file read whole into a Buffer, parsed by the
    facade, traced; against the oracle on the scene the Python binding parses from the same text."""
    w, h, spp, depth = 192, 128, 2, 5
    src, _ = ra.Scene.builtin(1, param=3, aspect=1.5)
    text = "# written by tests/test_headless.py\n"
    for i, t in enumerate(src.triangles):
        for v in (t["v0"], t["v1"], t["v2"]):
            text += "v " + " ".join(repr(float(np.float32(c))) for c in v) + "\n"
        text += f"f {3 * i + 1} {3 * i + 2} {3 * i + 3}\n"
    obj = tmp_path / "mesh.obj"
    obj.write_text(text)
    m = _basis((0.5, 1.5, 4.0), (0.0, 0.0, 0.0))
    outs = []
    for via in ("buffer", "path"):
        out = str(tmp_path / f"{via}.ppm")
        r = subprocess.run([EXE, "--obj-via", via, "--camera-matrix", _matrix_arg(m), "--vfov", "40", str(obj), str(w), str(h), str(spp),
                            str(depth), "0", out], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(_ppm(out, w, h))
    mesh = ra.Scene.from_obj(text=text)
    fs = ra.FlatScene.build(mesh)
    cam = ra.camera_from_matrix(m, vfov=40.0, aspect=np.float32(w) / np.float32(h))
    acc, ids, _ = oracle.render(fs.c, cam, ra.default_settings(w, h, spp=spp, max_depth=depth), threads=16)
    assert (ids != ra.NO_HIT).sum() > 2000
    want = _rgb_top_down(oracle.resolve(acc))
    assert np.array_equal(outs[0], want) and np.array_equal(outs[1], want)


def test_panel_state_machine_through_rt_headless_script(ra, oracle, tmp_path):
    """the control panel's logic (rt_panel_init / _apply / _step_or_panic, rt.hpp) - what
    restarts the accumulation, how sample_offset and the ACCUMULATE flag advance - run by rt_headless --script
    in place of the ImGui widgets.  Every displayed frame must be the oracle's resolved frame of the sum the
    panel's rules imply: samples-per-frame keeps the sum; depth, camera, sky and the button restart it."""
    w, h = 160, 90
    script = tmp_path / "panel.txt"
    script.write_text("""# k: what the frame must show
frame
frame
spp 3
frame
depth 3
frame
depth 3
frame
camera 5 4 9 0 0.5 0 35 0 10
frame
frame
sky 0.9 0.4 0.2 0.1 0.2 0.6
frame
restart
spp 1
frame
""")
    out = str(tmp_path / "p.ppm")
    r = subprocess.run([EXE, "--script", str(script), "0", str(w), str(h), "2", "8", "0", out], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stdout + r.stderr
    scene, cam0 = ra.Scene.builtin(0, aspect=np.float32(w) / np.float32(h))
    fs = ra.FlatScene.build(scene)
    cam1 = ra.look_at((5, 4, 9), (0, 0.5, 0), vfov=35.0, aspect=np.float32(w) / np.float32(h), aperture=0.0, focus_dist=10.0)
    sky = dict(sky_bottom=(0.9, 0.4, 0.2), sky_top=(0.1, 0.2, 0.6))
    # (camera, depth, sky kwargs, [(sample_offset, spp) of every frame in the sum so far])
    expect = [
        (cam0, 8, {}, [(0, 2)]),
        (cam0, 8, {}, [(0, 2), (2, 2)]),
        (cam0, 8, {}, [(0, 2), (2, 2), (4, 3)]),           # spp changed: the sum goes on
        (cam0, 3, {}, [(0, 3)]),                           # depth changed: restart
        (cam0, 3, {}, [(0, 3), (3, 3)]),                   # depth "changed" to the same value: no restart
        (cam1, 3, {}, [(0, 3)]),                           # camera: restart
        (cam1, 3, {}, [(0, 3), (3, 3)]),
        (cam1, 3, sky, [(0, 3)]),                          # sky: restart
        (cam1, 3, sky, [(0, 1)]),                          # button: restart (and 1 spp from here on)
    ]
    shown = [int(line.split(":")[1].split()[0]) for line in r.stdout.splitlines() if line.startswith("frame ")]
    assert shown == [sum(n for _, n in parts) for *_, parts in expect], r.stdout
    for k, (cam, depth, kw, parts) in enumerate(expect):
        acc = np.zeros((h, w, 4), np.float32)
        for j, (off, n) in enumerate(parts):
            st = ra.default_settings(w, h, spp=n, max_depth=depth, sample_offset=off, flags=ra.FLAG_ACCUMULATE if j else 0, **kw)
            acc, _, _ = oracle.render(fs.c, cam, st, accum=acc, threads=16)
        assert np.array_equal(_ppm(f"{out}.{k}.ppm", w, h), _rgb_top_down(oracle.resolve(acc))), f"frame {k}"
