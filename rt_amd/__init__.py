"""rt_amd — Python plumbing over librtgpu.so (the C ABI of include/rtgpu.h).

This is a thin ctypes binding used by the tests and bench.py; the product is
the shared library (HIP kernels + C ABI + host BVH builder) and the C++ facade
in rt_amd/host/.  There is NO fallback: if the library is missing, importing
this package raises, and no CPU path exists behind any call here.
"""
from .api import (  # noqa: F401
    LIB_PATH,
    Camera,
    Context,
    FlatScene,
    RtgpuError,
    Scene,
    Settings,
    Stats,
    compact_pixels,
    default_settings,
    lib,
    look_at,
    write_png,
    write_ppm,
    Group,
    camera_from_matrix,
    MATERIAL_DTYPE,
    SPHERE_DTYPE,
    TRIANGLE_DTYPE,
    FLAG_ACCUMULATE,
    FLAG_COMPACT,
    FLAG_STATS,
    FLAG_PROFILE,
    NO_HIT,
    NODE48_MIN_NODES,
    MAT_LAMBERTIAN,
    MAT_METAL,
    MAT_DIELECTRIC,
    MAT_EMISSIVE,
    SCENE_SEED,
)
