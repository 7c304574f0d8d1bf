"""The C-ABI library loads and exports every symbol include/rtgpu.h declares
(no compute calls: runs without a GPU)."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "rtgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rtgpu_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(ra):
    names = declared_functions()
    assert len(names) >= 20, names
    for n in names:
        assert hasattr(ra.lib, n), f"librtgpu.so does not export {n}"


def test_abi_version_and_struct_sizes(ra):
    assert ra.lib.rtgpu_abi_version() == 5
    from rt_amd import api
    assert C.sizeof(api.Camera) == 80
    assert C.sizeof(api.Settings) == 72
    assert C.sizeof(api.Stats) == 48
    assert C.sizeof(api.SceneDesc) == 48
    assert C.sizeof(api.FlatSceneC) == 128
    assert api.MATERIAL_DTYPE.itemsize == 32 and api.SPHERE_DTYPE.itemsize == 20
    assert api.TRIANGLE_DTYPE.itemsize == 40


def test_library_is_gfx950_only():
    blob = open(os.path.join(ROOT, "rt_amd", "librtgpu.so"), "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob
    for other in (b"--gfx906", b"--gfx90a", b"--gfx942"):
        assert other not in blob


def test_product_does_not_reference_the_oracle():
    """The product path must never route through oracle/ (parity would be void)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rt_amd")):
        if "build" in dirpath or "__pycache__" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in text and "rto_" not in text, f
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f


def test_create_without_gpu_fails_cleanly(ra):
    import torch
    if torch.cuda.is_available():
        return  # covered by the gpu tests
    h = C.c_void_p()
    rc = ra.lib.rtgpu_create(C.byref(h), 0)
    assert rc != 0 and not h.value
    assert b"HIP" in ra.lib.rtgpu_last_error(None) or ra.lib.rtgpu_last_error(None)


def test_default_settings(ra):
    s = ra.default_settings(1920, 1080, spp=16, max_depth=8)
    assert (s.width, s.height, s.spp, s.max_depth, s.seed, s.tile_size, s.world_size, s.chunk_samples) == (1920, 1080, 16, 8, 1, 64, 1, 4)
    assert np.float32(s.tmin) == np.float32(1e-3)
    assert list(s.sky_top) == [0.5, np.float32(0.7), 1.0] and list(s.sky_bottom) == [1.0, 1.0, 1.0]
    # chunk rule: 2 samples per work item below 16 spp, 4 up to 256 spp, then at most 64 chunks per pixel
    for spp, want in ((1, 2), (15, 2), (16, 4), (64, 4), (128, 4), (256, 4), (257, 5), (4096, 64)):
        st = ra.default_settings(64, 64, spp=spp)
        assert st.chunk_samples == want and (spp + want - 1) // want <= 64
