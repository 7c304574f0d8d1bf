"""ctypes binding of include/rtgpu.h.  Struct layouts mirror the header field
for field; tests/test_abi.py checks sizes and that every declared symbol is
exported."""
import ctypes as C
import os

import numpy as np

# RTGPU_LIB: an experimental build of the same library (developer knob, see tools/)
LIB_PATH = os.environ.get("RTGPU_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "librtgpu.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C rt_amd/csrc` "
        "(or python -c 'import __graft_entry__ as g; g.build()'). "
        "rt_amd has no CPU fallback."
    )



def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64 (soname libamdhip64.so.7,
    file name libamdhip64.so).  A process that loads BOTH the system runtime
    (through librtgpu.so) and torch's copy ends up with two HIP/HSA runtimes and
    the second one finds no GPU.  Loading torch's copy first makes the dynamic
    loader resolve librtgpu.so's NEEDED libamdhip64.so.7 to it by soname, so the
    two share one runtime whichever is imported first.  Without torch installed
    the system runtime (/opt/rocm/lib) is used, as for any C/C++ caller."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(p):
        C.CDLL(p, mode=C.RTLD_GLOBAL)


_share_hip_runtime_with_torch()
lib = C.CDLL(LIB_PATH)

FLAG_ACCUMULATE = 1
FLAG_COMPACT = 2
FLAG_STATS = 4
FLAG_PROFILE = 8
NO_HIT = 0xFFFFFFFF
SCENE_SEED = 0x52545F31  # "RT_1" 

MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_EMISSIVE = 0, 1, 2, 3
NODE48_MIN_NODES = 1366

MATERIAL_DTYPE = np.dtype(
    [("albedo", "<f4", (3,)), ("param", "<f4"), ("type", "<u4"), ("_pad", "<u4", (3,))]
)
SPHERE_DTYPE = np.dtype([("center", "<f4", (3,)), ("radius", "<f4"), ("material", "<u4")])
TRIANGLE_DTYPE = np.dtype(
    [("v0", "<f4", (3,)), ("v1", "<f4", (3,)), ("v2", "<f4", (3,)), ("material", "<u4")]
)
assert MATERIAL_DTYPE.itemsize == 32 and SPHERE_DTYPE.itemsize == 20 and TRIANGLE_DTYPE.itemsize == 40


class RtgpuError(RuntimeError):
    pass


class SceneDesc(C.Structure):
    _fields_ = [
        ("spheres", C.c_void_p), ("n_spheres", C.c_uint64),
        ("triangles", C.c_void_p), ("n_triangles", C.c_uint64),
        ("materials", C.c_void_p), ("n_materials", C.c_uint64),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("origin", C.c_float * 3), ("lower_left", C.c_float * 3),
        ("horizontal", C.c_float * 3), ("vertical", C.c_float * 3),
        ("u", C.c_float * 3), ("v", C.c_float * 3),
        ("lens_radius", C.c_float), ("_pad", C.c_float),
    ]


class Settings(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32),
        ("sample_offset", C.c_uint32), ("max_depth", C.c_uint32), ("seed", C.c_uint32),
        ("tmin", C.c_float), ("sky_bottom", C.c_float * 3), ("sky_top", C.c_float * 3),
        ("tile_size", C.c_uint32), ("rank", C.c_uint32), ("world_size", C.c_uint32),
        ("flags", C.c_uint32), ("chunk_samples", C.c_uint32),
    ]

    def copy(self):
        s = Settings()
        C.memmove(C.byref(s), C.byref(self), C.sizeof(Settings))
        return s


class Stats(C.Structure):
    _fields_ = [
        ("paths", C.c_uint64), ("rays", C.c_uint64), ("nodes", C.c_uint64),
        ("tri_tests", C.c_uint64), ("sphere_tests", C.c_uint64), ("scatters", C.c_uint64),
    ]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class FlatSceneC(C.Structure):
    _fields_ = [
        ("nodes", C.c_void_p), ("n_nodes", C.c_uint64),
        ("tris", C.c_void_p), ("n_tris", C.c_uint64),
        ("spheres", C.c_void_p), ("n_spheres", C.c_uint64),
        ("sphere_meta", C.c_void_p),
        ("materials", C.c_void_p), ("n_materials", C.c_uint64),
        ("max_depth", C.c_uint32), ("_pad", C.c_uint32),
        ("build_ms", C.c_double),
        ("nodes48", C.c_void_p),
        ("nodes4", C.c_void_p), ("n_nodes4", C.c_uint64),
        ("max_stack4", C.c_uint32), ("_pad2", C.c_uint32),
        ("nodes4_src", C.c_void_p),
    ]


assert C.sizeof(Camera) == 80 and C.sizeof(Settings) == 72 and C.sizeof(Stats) == 48

_P = C.POINTER
lib.rtgpu_create.argtypes = [_P(C.c_void_p), C.c_int]
lib.rtgpu_create.restype = C.c_int
lib.rtgpu_destroy.argtypes = [C.c_void_p]
lib.rtgpu_destroy.restype = None
lib.rtgpu_last_error.argtypes = [C.c_void_p]
lib.rtgpu_last_error.restype = C.c_char_p
lib.rtgpu_abi_version.restype = C.c_uint32
lib.rtgpu_upload_scene.argtypes = [C.c_void_p, _P(SceneDesc)]
lib.rtgpu_get_flat_scene.argtypes = [C.c_void_p, _P(FlatSceneC)]
lib.rtgpu_flatten_scene_host.argtypes = [_P(SceneDesc), _P(FlatSceneC)]
lib.rtgpu_flat_scene_free.argtypes = [_P(FlatSceneC)]
lib.rtgpu_flat_scene_free.restype = None
lib.rtgpu_render.argtypes = [C.c_void_p, _P(Camera), _P(Settings), C.c_void_p, C.c_void_p, _P(Stats)]
lib.rtgpu_render_begin.argtypes = [C.c_void_p, _P(Camera), _P(Settings), C.c_void_p, C.c_void_p]
lib.rtgpu_render_poll.argtypes = [C.c_void_p, _P(C.c_int)]
lib.rtgpu_render_end.argtypes = [C.c_void_p, _P(Stats)]
lib.rtgpu_render_device.argtypes = [
    C.c_void_p, _P(Camera), _P(Settings), C.c_void_p, C.c_void_p, C.c_void_p, _P(Stats)]
lib.rtgpu_render_device_static.argtypes = [C.c_void_p, _P(Camera), _P(Settings), C.c_void_p, C.c_void_p, C.c_void_p]
lib.rtgpu_static_scratch_bytes.restype = C.c_uint64
lib.rtgpu_render_resolved.argtypes = [C.c_void_p, _P(Camera), _P(Settings), C.c_void_p, _P(Stats)]
lib.rtgpu_resolved_frame.argtypes = [C.c_void_p]
lib.rtgpu_resolved_frame.restype = C.c_void_p
lib.rtgpu_resident_accum.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
lib.rtgpu_resident_reset.argtypes = [C.c_void_p]
lib.rtgpu_resident_reset.restype = None
lib.rtgpu_group_create.argtypes = [_P(C.c_void_p), _P(C.c_int), C.c_int]
lib.rtgpu_group_destroy.argtypes = [C.c_void_p]
lib.rtgpu_group_destroy.restype = None
lib.rtgpu_group_last_error.argtypes = [C.c_void_p]
lib.rtgpu_group_last_error.restype = C.c_char_p
lib.rtgpu_group_size.argtypes = [C.c_void_p]
lib.rtgpu_group_comm_ranks.argtypes = [C.c_void_p]
lib.rtgpu_group_upload_scene.argtypes = [C.c_void_p, _P(SceneDesc), C.c_int]
lib.rtgpu_group_render_begin.argtypes = [C.c_void_p, _P(Camera), _P(Settings)]
lib.rtgpu_group_render_end.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, _P(Stats)]
lib.rtgpu_group_render.argtypes = [C.c_void_p, _P(Camera), _P(Settings), C.c_void_p, C.c_void_p, _P(Stats)]
lib.rtgpu_group_kernel_ms.argtypes = [C.c_void_p, C.c_int, _P(C.c_float)]
lib.rtgpu_profile_counters.argtypes = [C.c_void_p, _P(C.c_uint64)]
lib.rtgpu_last_kernel_ms.argtypes = [C.c_void_p, _P(C.c_float)]
lib.rtgpu_last_kernel_name.argtypes = [C.c_void_p]
lib.rtgpu_last_kernel_name.restype = C.c_char_p
lib.rtgpu_resolve_rgba8_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
lib.rtgpu_resolve_rgba8.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
lib.rtgpu_compact_pixels.argtypes = [C.c_uint32] * 5
lib.rtgpu_compact_pixels.restype = C.c_uint64
lib.rtgpu_untile_device.argtypes = [
    C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
    C.c_void_p]
lib.rtgpu_camera_look_at.argtypes = [
    _P(Camera), C.c_float * 3, C.c_float * 3, C.c_float * 3, C.c_float, C.c_float, C.c_float, C.c_float]
lib.rtgpu_camera_look_at.restype = None
lib.rtgpu_camera_from_matrix.argtypes = [_P(Camera), C.c_float * 16, C.c_float, C.c_float, C.c_float, C.c_float]
lib.rtgpu_camera_from_matrix.restype = None
lib.rtgpu_default_settings.argtypes = [_P(Settings), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
lib.rtgpu_default_settings.restype = None
lib.rtgpu_scene_builtin.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, _P(SceneDesc), _P(Camera), C.c_float]
lib.rtgpu_upload_scene_lbvh.argtypes = [C.c_void_p, _P(SceneDesc)]
lib.rtgpu_refit.argtypes = [C.c_void_p, _P(SceneDesc)]
lib.rtgpu_refit_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
lib.rtgpu_build_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
lib.rtgpu_scene_free.argtypes = [_P(SceneDesc)]
lib.rtgpu_scene_free.restype = None
lib.rtgpu_scene_parse_obj.argtypes = [C.c_char_p, C.c_uint64, _P(SceneDesc)]
lib.rtgpu_scene_load_obj.argtypes = [C.c_char_p, _P(SceneDesc)]
lib.rtgpu_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
lib.rtgpu_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]


def _global_error():
    return (lib.rtgpu_last_error(None) or b"").decode()


def look_at(frm, at, up=(0, 1, 0), vfov=20.0, aspect=16 / 9, aperture=0.0, focus_dist=10.0):
    cam = Camera()
    f3 = C.c_float * 3
    lib.rtgpu_camera_look_at(C.byref(cam), f3(*frm), f3(*at), f3(*up), vfov, aspect, aperture, focus_dist)
    return cam


def camera_from_matrix(m, vfov=20.0, aspect=16 / 9, aperture=0.0, focus_dist=10.0):
    """m: 4x4 camera-to-world matrix, row-vector convention of the reference (rows = right, up,
    back, eye)."""
    cam = Camera()
    flat = [float(x) for x in np.asarray(m, np.float32).reshape(16)]
    lib.rtgpu_camera_from_matrix(C.byref(cam), (C.c_float * 16)(*flat), vfov, aspect, aperture, focus_dist)
    return cam


def default_settings(width, height, spp=1, max_depth=8, **kw):
    s = Settings()
    lib.rtgpu_default_settings(C.byref(s), width, height, spp, max_depth)
    for k, v in kw.items():
        if k in ("sky_bottom", "sky_top"):
            getattr(s, k)[:] = v
        else:
            setattr(s, k, v)
    return s


def write_ppm(path, rgba8):
    """rgba8: uint32 [h, w] of 0xRRGGBBAA, row 0 = bottom (what Context.resolve returns)."""
    a = np.ascontiguousarray(rgba8, np.uint32)
    h, w = a.shape
    if lib.rtgpu_write_ppm(str(path).encode(), a.ctypes.data, w, h):
        raise RtgpuError(_global_error())


def write_png(path, rgba8):
    """The same image as an 8-bit RGBA PNG (rtgpu_write_png)."""
    a = np.ascontiguousarray(rgba8, np.uint32)
    h, w = a.shape
    if lib.rtgpu_write_png(str(path).encode(), a.ctypes.data, w, h):
        raise RtgpuError(_global_error())


def compact_pixels(width, height, tile_size, rank, world):
    return int(lib.rtgpu_compact_pixels(width, height, tile_size, rank, world))


class Scene:
    """Caller-owned scene arrays (numpy structured arrays) + the C descriptor."""

    def __init__(self, spheres=None, triangles=None, materials=None):
        self.spheres = np.ascontiguousarray(
            spheres if spheres is not None else np.zeros(0, SPHERE_DTYPE), dtype=SPHERE_DTYPE)
        self.triangles = np.ascontiguousarray(
            triangles if triangles is not None else np.zeros(0, TRIANGLE_DTYPE), dtype=TRIANGLE_DTYPE)
        self.materials = np.ascontiguousarray(
            materials if materials is not None else np.zeros(0, MATERIAL_DTYPE), dtype=MATERIAL_DTYPE)
        self.desc = SceneDesc(
            self.spheres.ctypes.data if len(self.spheres) else None, len(self.spheres),
            self.triangles.ctypes.data if len(self.triangles) else None, len(self.triangles),
            self.materials.ctypes.data if len(self.materials) else None, len(self.materials))

    @property
    def n_prims(self):
        return len(self.spheres) + len(self.triangles)

    @staticmethod
    def _from_desc(d):
        """Copies the malloc'ed arrays of a library-made descriptor into numpy and frees them."""
        try:
            def grab(ptr, n, dt):
                if not n:
                    return np.zeros(0, dt)
                buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
                return np.frombuffer(buf, dtype=dt).copy()
            return Scene(grab(d.spheres, d.n_spheres, SPHERE_DTYPE),
                         grab(d.triangles, d.n_triangles, TRIANGLE_DTYPE),
                         grab(d.materials, d.n_materials, MATERIAL_DTYPE))
        finally:
            lib.rtgpu_scene_free(C.byref(d))

    @staticmethod
    def builtin(kind, seed=SCENE_SEED, param=0, aspect=16 / 9):
        """S0/S1/S2 of SURVEY.md §8d -> (Scene, Camera).  Arrays are copied into numpy."""
        d, cam = SceneDesc(), Camera()
        if lib.rtgpu_scene_builtin(kind, seed, param, C.byref(d), C.byref(cam), aspect):
            raise RtgpuError(_global_error())
        return Scene._from_desc(d), cam

    @staticmethod
    def from_obj(path=None, text=None):
        """Wavefront OBJ file or text -> Scene of triangles (rtgpu_scene_load_obj / _parse_obj)."""
        d = SceneDesc()
        if text is not None:
            raw = text.encode() if isinstance(text, str) else bytes(text)
            rc = lib.rtgpu_scene_parse_obj(raw, len(raw), C.byref(d))
        else:
            rc = lib.rtgpu_scene_load_obj(str(path).encode(), C.byref(d))
        if rc:
            raise RtgpuError(_global_error())
        return Scene._from_desc(d)


class FlatScene:
    """Flattened scene (include/rtgpu_format.h).  Either built on the host
    (FlatScene.build) or borrowed from a Context (Context.flat_scene)."""

    def __init__(self, c_struct, owned, keepalive=None):
        self.c = c_struct
        self._owned = owned
        self._keep = keepalive

    @staticmethod
    def build(scene):
        fs = FlatSceneC()
        if lib.rtgpu_flatten_scene_host(C.byref(scene.desc), C.byref(fs)):
            raise RtgpuError(_global_error())
        return FlatScene(fs, True)

    def _view(self, ptr, n, dtype, words):
        if not n:
            return np.zeros((0, words), dtype)
        buf = (C.c_char * (n * words * 4)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).reshape(n, words)

    @property
    def nodes_f32(self):
        return self._view(self.c.nodes, self.c.n_nodes, np.float32, 16)

    @property
    def nodes_u32(self):
        return self._view(self.c.nodes, self.c.n_nodes, np.uint32, 16)

    @property
    def nodes48_u32(self):
        """Packed 48-byte nodes (12 words each); empty below RTGPU_NODE48_MIN_NODES."""
        if not self.c.nodes48:
            return np.zeros((0, 12), np.uint32)
        return self._view(self.c.nodes48, self.c.n_nodes, np.uint32, 12)

    @property
    def nodes4_u32(self):
        """Four-wide packed nodes (24 words each); empty when absent."""
        if not self.c.nodes4:
            return np.zeros((0, 24), np.uint32)
        return self._view(self.c.nodes4, self.c.n_nodes4, np.uint32, 24)

    @property
    def nodes4_src(self):
        if not self.c.nodes4_src:
            return np.zeros((0, 4), np.uint32)
        return self._view(self.c.nodes4_src, self.c.n_nodes4, np.uint32, 4)

    @property
    def tris_f32(self):
        return self._view(self.c.tris, self.c.n_tris, np.float32, 12)

    @property
    def tris_u32(self):
        return self._view(self.c.tris, self.c.n_tris, np.uint32, 12)

    @property
    def spheres_f32(self):
        return self._view(self.c.spheres, self.c.n_spheres, np.float32, 4)

    @property
    def sphere_meta(self):
        return self._view(self.c.sphere_meta, self.c.n_spheres, np.uint32, 2)

    def __del__(self):
        if getattr(self, "_owned", False):
            lib.rtgpu_flat_scene_free(C.byref(self.c))
            self._owned = False


class Context:
    """One GPU (one process per GPU is the intended multi-GPU layout)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        if lib.rtgpu_create(C.byref(self._h), device):
            raise RtgpuError(_global_error())

    def _check(self, rc):
        if rc:
            raise RtgpuError((lib.rtgpu_last_error(self._h) or b"").decode())

    def close(self):
        if self._h:
            lib.rtgpu_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, scene, builder="sah"):
        """builder: "sah" = binned-SAH BVH on the host (default), "lbvh" = LBVH on the GPU."""
        if builder == "lbvh":
            self._check(lib.rtgpu_upload_scene_lbvh(self._h, C.byref(scene.desc)))
        elif builder == "sah":
            self._check(lib.rtgpu_upload_scene(self._h, C.byref(scene.desc)))
        else:
            raise ValueError(f"unknown builder {builder!r}")

    def refit(self, scene):
        """New positions / materials under the uploaded tree (rtgpu_refit): `scene` must
        hold the uploaded scene's counts, primitive i being the moved primitive i."""
        self._check(lib.rtgpu_refit(self._h, C.byref(scene.desc)))

    def refit_device(self, spheres_ptr, n_spheres, triangles_ptr, n_triangles):
        """rtgpu_refit_device: the moved primitives as device arrays (SPHERE_DTYPE /
        TRIANGLE_DTYPE records, e.g. the data_ptr() of a torch uint8 tensor)."""
        self._check(lib.rtgpu_refit_device(self._h, spheres_ptr, n_spheres, triangles_ptr, n_triangles))

    def build_device(self, spheres_ptr, n_spheres, triangles_ptr, n_triangles, materials):
        """rtgpu_build_device: LBVH over primitive arrays in device memory (SPHERE_DTYPE /
        TRIANGLE_DTYPE records); `materials`: numpy array of MATERIAL_DTYPE on the host."""
        mats = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
        self._check(lib.rtgpu_build_device(self._h, spheres_ptr, n_spheres, triangles_ptr, n_triangles,
                                           mats.ctypes.data, len(mats)))

    def flat_scene(self):
        fs = FlatSceneC()
        self._check(lib.rtgpu_get_flat_scene(self._h, C.byref(fs)))
        return FlatScene(fs, False, keepalive=self)

    def render(self, cam, settings, accum=None, want_ids=True, want_stats=True):
        """Host-buffer entry point (rtgpu_render).  Returns (accum, prim_id, Stats)."""
        s = settings
        if s.flags & FLAG_COMPACT:
            n = compact_pixels(s.width, s.height, s.tile_size, s.rank, s.world_size)
            shape = (n, 4)
        else:
            n = s.width * s.height
            shape = (s.height, s.width, 4)
        if accum is None:
            accum = np.zeros(shape, np.float32)
        assert accum.dtype == np.float32 and accum.size == n * 4 and accum.flags.c_contiguous
        ids = np.full(shape[:-1], NO_HIT, np.uint32) if want_ids else None
        st = Stats()
        self._check(lib.rtgpu_render(
            self._h, C.byref(cam), C.byref(s), accum.ctypes.data,
            ids.ctypes.data if ids is not None else None, C.byref(st) if want_stats else None))
        return accum, ids, st

    def render_begin(self, cam, settings, accum, ids=None):
        """Non-blocking rtgpu_render_begin into caller-owned numpy buffers (kept alive here
        until render_end)."""
        assert accum.dtype == np.float32 and accum.flags.c_contiguous
        assert ids is None or (ids.dtype == np.uint32 and ids.flags.c_contiguous)
        self._check(lib.rtgpu_render_begin(self._h, C.byref(cam), C.byref(settings), accum.ctypes.data,
                                           ids.ctypes.data if ids is not None else None))
        self._pending = (accum, ids)

    def render_poll(self):
        done = C.c_int()
        self._check(lib.rtgpu_render_poll(self._h, C.byref(done)))
        return bool(done.value)

    def render_end(self, want_stats=True):
        st = Stats()
        self._check(lib.rtgpu_render_end(self._h, C.byref(st) if want_stats else None))
        accum, ids = getattr(self, "_pending", (None, None))
        self._pending = None
        return accum, ids, st

    def render_device(self, cam, settings, accum_ptr, ids_ptr=None, stream=None, stats=None):
        """Device-pointer entry point (rtgpu_render_device); async unless stats is given."""
        self._check(lib.rtgpu_render_device(
            self._h, C.byref(cam), C.byref(settings), accum_ptr, ids_ptr, stream,
            C.byref(stats) if stats is not None else None))

    def render_device_static(self, cam, settings, accum_ptr, ids_ptr=None, stream=None):
        """rtgpu_render_device_static: the same frame with the launch's scratch in the library's
        static device array - the entry point that allocates nothing."""
        self._check(lib.rtgpu_render_device_static(
            self._h, C.byref(cam), C.byref(settings), accum_ptr, ids_ptr, stream))

    def render_resolved(self, cam, settings, want_stats=False):
        """rtgpu_render_resolved: the accumulation stays in the context, the resolved
        0xRRGGBBAA frame [h, w] comes back.  Returns (rgba8, Stats or None)."""
        out = np.zeros((settings.height, settings.width), np.uint32)
        st = Stats() if want_stats else None
        self._check(lib.rtgpu_render_resolved(self._h, C.byref(cam), C.byref(settings), out.ctypes.data,
                                              C.byref(st) if want_stats else None))
        return out, st

    def resident_accum(self, width, height):
        out = np.zeros((height, width, 4), np.float32)
        self._check(lib.rtgpu_resident_accum(self._h, out.ctypes.data, width * height))
        return out

    def resident_reset(self):
        lib.rtgpu_resident_reset(self._h)

    def profile_counters(self):
        out = (C.c_uint64 * 32)()
        self._check(lib.rtgpu_profile_counters(self._h, out))
        return [int(v) for v in out]

    def kernel_ms(self):
        ms = C.c_float()
        self._check(lib.rtgpu_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def kernel_name(self):
        """Instantiation of the last trace-kernel launch, e.g. 'trace_kernel<0,16,2>'."""
        return (lib.rtgpu_last_kernel_name(self._h) or b"").decode()

    def debug_slot_tree(self):
        """The slot tree (include/rtgpu_format.h) of the uploaded scene as the trace loop traverses it:
        u32 array [n_slots, 16], or None if the scene has none.  Diagnostics, not part of the C ABI."""
        fn = lib.rtgpu_debug_slot_tree
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        n = C.c_uint64()
        self._check(fn(self._h, None, 0, C.byref(n)))
        if not n.value:
            return None
        out = np.zeros((n.value, 16), np.uint32)
        self._check(fn(self._h, out.ctypes.data, n.value, C.byref(n)))
        return out

    def resolve(self, accum):
        accum = np.ascontiguousarray(accum, np.float32)
        n = accum.size // 4
        out = np.zeros(accum.shape[:-1], np.uint32)
        self._check(lib.rtgpu_resolve_rgba8(self._h, accum.ctypes.data, out.ctypes.data, n))
        return out

    def resolve_device(self, accum_ptr, out_ptr, n_pixels, stream=None):
        self._check(lib.rtgpu_resolve_rgba8_device(self._h, accum_ptr, out_ptr, n_pixels, stream))

    def untile_device(self, gathered_ptr, image_ptr, width, height, tile_size, world, elem_bytes,
                      stream=None):
        self._check(lib.rtgpu_untile_device(
            self._h, gathered_ptr, image_ptr, width, height, tile_size, world, elem_bytes, stream))


class Group:
    """Several GPUs of one node behind one handle (rtgpu_group_*): one process, one worker
    thread and one context per GPU, RCCL tile gather to the first device of the list."""

    def __init__(self, devices):
        self._h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        if lib.rtgpu_group_create(C.byref(self._h), arr, len(devices)):
            raise RtgpuError((lib.rtgpu_group_last_error(None) or b"").decode())
        self.size = len(devices)

    def _check(self, rc):
        if rc:
            raise RtgpuError((lib.rtgpu_group_last_error(self._h) or b"").decode())

    def close(self):
        if self._h:
            lib.rtgpu_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def comm_ranks(self):
        """Ranks of the RCCL communicator (0: the tiles move by device copies, a rehearsal)."""
        return int(lib.rtgpu_group_comm_ranks(self._h))

    def upload(self, scene, builder="sah"):
        self._check(lib.rtgpu_group_upload_scene(self._h, C.byref(scene.desc), {"sah": 0, "lbvh": 1}[builder]))

    def render_begin(self, cam, settings):
        self._check(lib.rtgpu_group_render_begin(self._h, C.byref(cam), C.byref(settings)))
        self._shapes = getattr(self, "_shapes", []) + [(settings.height, settings.width)]  # only frames that began

    def render_end(self, want_accum=True, want_rgba8=True):
        shapes = getattr(self, "_shapes", [])
        h, w = shapes.pop(0) if shapes else (0, 0)     # no frame begun: the library says so
        acc = np.zeros((h, w, 4), np.float32) if want_accum else None
        rgba = np.zeros((h, w), np.uint32) if want_rgba8 else None
        st = Stats()
        self._check(lib.rtgpu_group_render_end(self._h, acc.ctypes.data if want_accum else None,
                                               rgba.ctypes.data if want_rgba8 else None, C.byref(st)))
        return acc, rgba, st

    def render(self, cam, settings, want_accum=True, want_rgba8=True):
        self.render_begin(cam, settings)
        return self.render_end(want_accum, want_rgba8)

    def kernel_ms(self, index):
        ms = C.c_float()
        self._check(lib.rtgpu_group_kernel_ms(self._h, index, C.byref(ms)))
        return float(ms.value)
