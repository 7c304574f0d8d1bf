"""Loader for the CPU oracle (oracle/rt_oracle.c) and, when present, the
reference's own math module (oracle/_ref/libref_math.so).

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this package; nothing under rt_amd/ does.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def _cpu_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


def lib_path():
    name = "liboracle.so" if _cpu_has_fma() else "liboracle_nofma.so"
    return os.path.join(_HERE, name)


def load():
    p = lib_path()
    if not os.path.exists(p):
        raise ImportError(f"{p} missing: run `make -C oracle`")
    lib = C.CDLL(p)
    f3 = C.POINTER(C.c_float)
    lib.rto_pcg.argtypes = [C.c_uint32]
    lib.rto_pcg.restype = C.c_uint32
    lib.rto_path_key.argtypes = [C.c_uint32] * 3
    lib.rto_path_key.restype = C.c_uint32
    lib.rto_u01.argtypes = [C.c_uint32, C.c_uint32]
    lib.rto_u01.restype = C.c_float
    lib.rto_dot3.argtypes = [f3, f3]
    lib.rto_dot3.restype = C.c_float
    lib.rto_len3.argtypes = [f3]
    lib.rto_len3.restype = C.c_float
    lib.rto_normalize3.argtypes = [f3, f3]
    lib.rto_cross3.argtypes = [f3, f3, f3]
    lib.rto_transform_point4.argtypes = [f3, f3, f3]
    lib.rto_kat_slab.argtypes = [f3, f3, f3, f3, C.c_float, C.c_float, f3]
    lib.rto_kat_sphere.argtypes = [f3, f3, f3, C.c_float, C.c_float, f3]
    lib.rto_kat_tri.argtypes = [f3, f3, f3, f3, f3, C.c_float, C.c_float, f3]
    lib.rto_kat_scatter.argtypes = [C.c_uint32, C.c_float, f3, f3, C.c_uint32, C.c_uint32, f3,
                                    C.POINTER(C.c_uint32)]
    lib.rto_render.argtypes = [
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int]
    lib.rto_render_slots.argtypes = lib.rto_render.argtypes + [C.c_void_p]
    lib.rto_set_slot_tree.argtypes = [C.c_int]
    lib.rto_set_slot_tree.restype = None
    lib.rto_build_slot_tree.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.rto_build_slot_tree.restype = C.c_void_p
    lib.rto_free_words.argtypes = [C.c_void_p]
    lib.rto_free_words.restype = None
    lib.rto_resolve_rgba8.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    lib.rto_resolve_rgba8.restype = None
    lib.rto_validate_bvh.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    lib.rto_build_lbvh.argtypes = [C.c_void_p, C.c_void_p]
    lib.rto_refit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rto_pack_node48.argtypes = [C.c_void_p, C.c_void_p]
    lib.rto_flat_free.argtypes = [C.c_void_p]
    lib.rto_flat_free.restype = None
    lib.rto_has_openmp.restype = C.c_int
    lib.rto_debug_max_stack.argtypes = [C.c_int]
    lib.rto_debug_max_stack.restype = C.c_int
    return lib


def load_ref_math():
    """The reference's code/math compiled from /root/reference (oracle/Makefile
    target `ref`).  Returns None when the prebuilt library is absent."""
    p = os.path.join(_HERE, "_ref", "libref_math.so")
    if not os.path.exists(p):
        return None
    lib = C.CDLL(p)
    f3 = C.POINTER(C.c_float)
    lib.ref_dot3.argtypes = [f3, f3]
    lib.ref_dot3.restype = C.c_float
    lib.ref_len3.argtypes = [f3]
    lib.ref_len3.restype = C.c_float
    lib.ref_normalize3.argtypes = [f3, f3]
    for n in ("ref_add3", "ref_sub3", "ref_mul3"):
        getattr(lib, n).argtypes = [f3, f3, f3]
    lib.ref_scale3.argtypes = [f3, C.c_float, f3]
    lib.ref_transform_point4.argtypes = [f3, f3, f3]
    return lib


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Oracle:
    """Convenience wrapper: render a flattened scene on the CPU."""

    def __init__(self):
        self.lib = load()

    def slot_tree(self, flat_c):
        """The slot tree (include/rtgpu_format.h) of a flattened scene with four-wide nodes: u32 array
        [n_slots, 16], or None if the tree has no slot form (then the four-wide nodes are traversed)."""
        n = C.c_uint64()
        p = self.lib.rto_build_slot_tree(C.addressof(flat_c), C.byref(n))
        if not p:
            if n.value:
                raise MemoryError("rto_build_slot_tree")
            return None
        try:
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n.value, 16)).copy()
        finally:
            self.lib.rto_free_words(p)

    def render(self, flat_c, cam, settings, accum=None, crop=None, brute_force=False, threads=1,
               want_ids=True, slot_tree=True, slots=None):
        """flat_c: rtgpu_flat_scene ctypes struct; cam/settings: ctypes structs of
        rt_amd.api (plain data).  Returns (accum[h,w,4], prim_id[h,w], stats dict).
        slot_tree: traverse the slot tree of a scene with four-wide nodes when it has one (what the
        product does by default; False = the four-wide nodes, RTGPU_SLOT_TREE=0 there).  slots: a tree
        from slot_tree() to traverse, instead of building it for this call."""
        w, h = settings.width, settings.height
        if accum is None:
            accum = np.zeros((h, w, 4), np.float32)
        ids = np.full((h, w), 0xFFFFFFFF, np.uint32) if want_ids else None
        stats = (C.c_uint64 * 6)()
        x0, y0, x1, y1 = crop if crop is not None else (0, 0, w, h)
        args = (C.addressof(flat_c), C.addressof(cam), C.addressof(settings), accum.ctypes.data,
                ids.ctypes.data if ids is not None else None, C.addressof(stats), x0, y0, x1, y1,
                1 if brute_force else 0, threads)
        if slots is not None:
            assert slots.dtype == np.uint32 and slots.flags.c_contiguous
            rc = self.lib.rto_render_slots(*args, slots.ctypes.data)
        else:
            self.lib.rto_set_slot_tree(1 if slot_tree else 0)
            rc = self.lib.rto_render(*args)
            self.lib.rto_set_slot_tree(1)
        if rc:
            raise RuntimeError(f"rto_render failed with {rc}")
        keys = ("paths", "rays", "nodes", "tri_tests", "sphere_tests", "scatters")
        return accum, ids, dict(zip(keys, (int(v) for v in stats)))

    def resolve(self, accum):
        accum = np.ascontiguousarray(accum, np.float32)
        out = np.zeros(accum.shape[:-1], np.uint32)
        self.lib.rto_resolve_rgba8(accum.ctypes.data, out.ctypes.data, accum.size // 4)
        return out

    def build_lbvh(self, scene_desc, flat_struct_type):
        """CPU restatement of the GPU LBVH builder: scene descriptor -> flattened scene
        (a ctypes struct of the caller's rtgpu_flat_scene type; free with free_flat)."""
        fs = flat_struct_type()
        rc = self.lib.rto_build_lbvh(C.addressof(scene_desc), C.addressof(fs))
        if rc:
            raise RuntimeError(f"rto_build_lbvh failed with {rc}")
        return fs

    def pack_node48(self, node_words):
        """node_words: 16 u32 of one rtgpu_node -> 12 u32 of its packed form, or None."""
        import numpy as np
        src = np.ascontiguousarray(node_words, np.uint32)
        out = np.zeros(12, np.uint32)
        return out if self.lib.rto_pack_node48(src.ctypes.data, out.ctypes.data) else None

    def refit(self, flat_c, scene_desc):
        """CPU restatement of the GPU refit: same topology, geometry and materials of
        `scene_desc`.  Returns a new flattened scene (free with free_flat)."""
        fs = type(flat_c)()
        rc = self.lib.rto_refit(C.addressof(flat_c), C.addressof(scene_desc), C.addressof(fs))
        if rc:
            raise RuntimeError(f"rto_refit failed with {rc}")
        return fs

    def free_flat(self, fs):
        self.lib.rto_flat_free(C.addressof(fs))

    def validate_bvh(self, flat_c):
        d = C.c_uint32()
        rc = self.lib.rto_validate_bvh(C.addressof(flat_c), C.byref(d))
        return rc, int(d.value)
