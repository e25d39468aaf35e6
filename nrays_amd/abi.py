"""ctypes mirror of include/nrays_abi.h (the C-ABI drop-in boundary for scene::render).

Every structure here is the `#[repr(C)]` twin of the C declaration of the same name; field order
and types must match include/nrays_abi.h exactly (tests/test_abi.py checks sizes and symbols).
"""
import ctypes as C
import os

ABI_VERSION = 5
COUNT_AS_TIMED = 1  # nrays_render_device_counted: count the work of the plain (timed) render, include/nrays_abi.h

# NraysStatus
OK = 0
ERR_BAD_ARG = -1
ERR_HIP = -2
ERR_OOM = -3
ERR_UNSUPPORTED = -4
ERR_NO_DEVICE = -5
ERR_QUEUE_OVERFLOW = -6
ERR_RCCL = -7
UNIQUE_ID_BYTES = 128

# NraysShapeKind (examples/loader3d.rs:593-695)
SHAPE_BALL, SHAPE_CUBOID, SHAPE_CYLINDER, SHAPE_CAPSULE, SHAPE_CONE, SHAPE_PLANE, SHAPE_TRIMESH = range(7)
# NraysMaterialKind
MAT_PHONG, MAT_NORMAL, MAT_UV = range(3)
TEXEL_RGBA8, TEXEL_RGBA32F = 0, 1
INTERP_BILINEAR, INTERP_NEAREST = 0, 1
OVERFLOW_WRAP, OVERFLOW_CLAMP = 0, 1


class NraysLight(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("radius", C.c_double), ("racsample", C.c_uint32), ("color", C.c_float * 3)]


class NraysTexture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32), ("interp", C.c_uint32),
                ("overflow", C.c_uint32), ("reserved", C.c_uint32), ("texels", C.c_void_p)]


class NraysMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("ambiant", C.c_float * 3), ("diffuse", C.c_float * 3),
                ("specular", C.c_float * 3), ("shininess", C.c_float), ("texture_id", C.c_int32),
                ("alpha_texture_id", C.c_int32)]


class NraysMesh(C.Structure):
    _fields_ = [("num_vertices", C.c_uint32), ("num_triangles", C.c_uint32), ("vertices", C.POINTER(C.c_double)),
                ("uvs", C.POINTER(C.c_double)), ("indices", C.POINTER(C.c_uint32))]


class NraysNode(C.Structure):
    _fields_ = [("shape_kind", C.c_uint32), ("solid", C.c_uint32), ("params", C.c_double * 3),
                ("translation", C.c_double * 3), ("axis_angle", C.c_double * 3), ("refl_mix", C.c_float),
                ("refl_atenuation", C.c_float), ("alpha", C.c_float), ("reserved0", C.c_float),
                ("refr_coeff", C.c_double), ("material_id", C.c_uint32), ("mesh_id", C.c_int32)]


class NraysSceneDesc(C.Structure):
    _fields_ = [("background", C.c_float * 3), ("num_lights", C.c_uint32), ("lights", C.POINTER(NraysLight)),
                ("num_materials", C.c_uint32), ("materials", C.POINTER(NraysMaterial)),
                ("num_textures", C.c_uint32), ("textures", C.POINTER(NraysTexture)),
                ("num_meshes", C.c_uint32), ("meshes", C.POINTER(NraysMesh)),
                ("num_nodes", C.c_uint32), ("nodes", C.POINTER(NraysNode))]


class NraysRenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("ray_per_pixel", C.c_uint32),
                ("max_depth", C.c_uint32), ("window_width", C.c_double), ("camera_eye", C.c_double * 3),
                ("inv_proj_view", C.c_double * 16), ("seed", C.c_uint64), ("band_rows", C.c_uint32),
                ("band_owner", C.c_uint32), ("band_owners", C.c_uint32), ("reserved", C.c_uint32)]


class NraysStats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_reflection", C.c_uint64), ("rays_refraction", C.c_uint64),
                ("rays_shadow", C.c_uint64), ("node_tests", C.c_uint64), ("tri_tests", C.c_uint64),
                ("prim_tests", C.c_uint64), ("hit_records", C.c_uint64), ("tex_samples", C.c_uint64),
                ("generations", C.c_uint32), ("instrumented", C.c_uint32), ("kernel_ms_primary", C.c_double),
                ("kernel_ms_total", C.c_double), ("frames_timed", C.c_uint32), ("reserved", C.c_uint32),
                ("rays_primary_traced", C.c_uint64), ("rays_shadow_elided", C.c_uint64), ("node_fetches", C.c_uint64)]

    def rays_traced(self):
        """Rays that went through a BVT query (total_rays() counts every primary ray the reference would trace)."""
        return self.rays_primary_traced + self.rays_reflection + self.rays_refraction + self.rays_shadow - self.rays_shadow_elided

    def total_rays(self):
        return self.rays_primary + self.rays_reflection + self.rays_refraction + self.rays_shadow

    def algorithmic_bytes(self, width=0, rows=0):
        """B = 32 N_node + 36 N_tri + 64 N_prim + 64 N_hit + 16 N_texsample + 64 per ray
        (+ 12 B per framebuffer pixel)."""
        return (32 * self.node_tests + 36 * self.tri_tests + 64 * self.prim_tests + 64 * self.hit_records
                + 16 * self.tex_samples + 64 * self.total_rays() + 12 * width * rows)

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class NraysMultiTimings(C.Structure):
    _fields_ = [("render_ms", C.c_double), ("exchange_ms", C.c_double), ("untile_ms", C.c_double), ("frames", C.c_uint32), ("owner", C.c_uint32)]


class NraysTileCosts(C.Structure):
    _fields_ = [("tiles", C.c_uint64), ("sum_cycles", C.c_uint64), ("max_cycles", C.c_uint64), ("resident_waves", C.c_uint64), ("shader_clock_hz", C.c_double), ("kernel_ms", C.c_double)]


class NraysCastResult(C.Structure):
    _fields_ = [("toi", C.c_double), ("normal", C.c_double * 3), ("uv", C.c_double * 2), ("node_id", C.c_int32), ("flags", C.c_uint32)]


class NraysBlasDump(C.Structure):
    _fields_ = [("num_nodes", C.c_uint32), ("num_refs", C.c_uint32), ("root", C.c_int32), ("max_depth", C.c_int32), ("hairy", C.c_uint32),
                ("node_capacity", C.c_uint32), ("ref_capacity", C.c_uint32), ("pad", C.c_uint32), ("nodes", C.POINTER(C.c_float)), ("tri_ids", C.POINTER(C.c_uint32))]


# Every symbol include/nrays_abi.h declares, with its ctypes signature.
HIP_SYMBOLS = {
    "nrays_debug_blas_build": (C.c_int, [C.POINTER(NraysMesh), C.c_uint32, C.POINTER(NraysBlasDump)]),
    "nrays_debug_node_aabb": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_double)]),
    "nrays_debug_scene_flags": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "nrays_get_tile_costs": (C.c_int, [C.c_void_p, C.POINTER(NraysTileCosts)]),
    "nrays_debug_cast_batch": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                         C.POINTER(NraysCastResult)]),
    "nrays_scene_create": (C.c_int, [C.POINTER(NraysSceneDesc), C.POINTER(C.c_void_p)]),
    "nrays_render": (C.c_int, [C.c_void_p, C.POINTER(NraysRenderParams), C.POINTER(C.c_float)]),
    "nrays_render_rgb8": (C.c_int, [C.c_void_p, C.POINTER(NraysRenderParams), C.POINTER(C.c_uint8)]),
    "nrays_render_device": (C.c_int, [C.c_void_p, C.POINTER(NraysRenderParams), C.c_void_p, C.c_void_p]),
    "nrays_render_device_instrumented": (C.c_int, [C.c_void_p, C.POINTER(NraysRenderParams), C.c_void_p, C.c_void_p]),
    "nrays_render_device_counted": (C.c_int, [C.c_void_p, C.POINTER(NraysRenderParams), C.c_void_p, C.c_void_p, C.c_uint32]),
    "nrays_tile_rows": (C.c_uint32, [C.POINTER(NraysRenderParams)]),
    "nrays_untile_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "nrays_get_stats": (C.c_int, [C.c_void_p, C.POINTER(NraysStats)]),
    "nrays_get_primary_kernel_stats": (C.c_int, [C.c_void_p, C.POINTER(NraysStats)]),
    "nrays_scene_device_bytes": (C.c_uint64, [C.c_void_p]),
    "nrays_scene_destroy": (None, [C.c_void_p]),
    "nrays_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "nrays_comm_create": (C.c_int, [C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "nrays_comm_create_local": (C.c_int, [C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]),
    "nrays_comm_owners": (C.c_uint32, [C.c_void_p]),
    "nrays_comm_destroy": (None, [C.c_void_p]),
    "nrays_scene_set_create": (C.c_int, [C.POINTER(NraysSceneDesc), C.c_void_p, C.POINTER(C.c_void_p)]),
    "nrays_scene_set_destroy": (None, [C.c_void_p]),
    "nrays_scene_set_num_local": (C.c_uint32, [C.c_void_p]),
    "nrays_scene_set_local_scene": (C.c_void_p, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "nrays_render_multi": (C.c_int, [C.c_void_p, C.POINTER(NraysRenderParams), C.POINTER(C.c_float)]),
    "nrays_render_multi_device": (C.c_int, [C.c_void_p, C.POINTER(NraysRenderParams), C.c_void_p]),
    "nrays_multi_sync": (C.c_int, [C.c_void_p]),
    "nrays_multi_get_stats": (C.c_int, [C.c_void_p, C.POINTER(NraysStats)]),
    "nrays_multi_get_timings": (C.c_int, [C.c_void_p, C.POINTER(NraysMultiTimings)]),
    "nrays_last_error": (C.c_char_p, []),
    "nrays_abi_version": (C.c_uint32, []),
}

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# NRAYS_HIP_LIB overrides the library path (kernel A/B tuning with tools/kbench.py only).
HIP_LIB_PATH = os.environ.get("NRAYS_HIP_LIB") or os.path.join(_REPO, "nrays_amd", "lib", "libnrays_hip.so")
HOST_LIB_PATH = os.path.join(_REPO, "nrays_amd", "lib", "libnrays_host.so")

_hip_lib = None


class NraysError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("nrays status %d: %s" % (status, message))
        self.status = status


def load_hip_lib():
    """Loads the HIP product library.  Fails loudly: there is no CPU fallback in the product path."""
    global _hip_lib
    if _hip_lib is not None:
        return _hip_lib
    if not os.path.exists(HIP_LIB_PATH):
        raise ImportError("libnrays_hip.so is not built (%s missing); run `python -c 'import __graft_entry__ as g; "
                          "g.build()'` — the nrays_amd product path has no CPU fallback" % HIP_LIB_PATH)
    # torch bundles its own libamdhip64.so (SONAME libamdhip64.so.7).  It must be loaded FIRST so that
    # this library's NEEDED libamdhip64.so.7 binds to the same HIP runtime; two runtimes in one process
    # cannot both own the GPU ("no HIP device visible").  torch is plumbing here: device memory,
    # streams and torch.distributed.
    import torch  # noqa: F401
    lib = C.CDLL(HIP_LIB_PATH)
    for name, (res, args) in HIP_SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    if lib.nrays_abi_version() != ABI_VERSION:
        raise ImportError("libnrays_hip.so ABI version %d != %d" % (lib.nrays_abi_version(), ABI_VERSION))
    _hip_lib = lib
    return lib


def check(status):
    if status != OK:
        msg = load_hip_lib().nrays_last_error()
        raise NraysError(status, msg.decode() if msg else "?")
