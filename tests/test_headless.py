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
