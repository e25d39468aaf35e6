"""ctypes loader of the CPU oracle (oracle/nrays_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (nrays_amd) never does.  PARITY UNPINNED — see the header of nrays_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from nrays_amd import abi

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NRAYS_ORACLE_LIB") or os.path.join(_DIR, "_build", "libnrays_oracle.so")  # (NRAYS_ORACLE_LIB: the sanitizer build, tests/test_oracle_sanitizers.py)
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _DIR])


def use_native():
    """bench.py's cpu_baseline leg only: builds the oracle on THIS host with -march=native (a separate file; the portable build the tests
    use stays as it is) and makes it the library of this process.  Returns True if the native build is now in use."""
    global _lib, LIB_PATH
    native = os.path.join(_DIR, "_build", "libnrays_oracle_native.so")
    try:
        subprocess.check_call(["make", "-s", "-C", _DIR, "native"])
    except Exception:
        return False
    if not os.path.exists(native):
        return False
    LIB_PATH, _lib = native, None
    return True


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        l = C.CDLL(LIB_PATH)
        l.nrays_oracle_render.restype = C.c_int
        l.nrays_oracle_render.argtypes = [C.POINTER(abi.NraysSceneDesc), C.POINTER(abi.NraysRenderParams),
                                          C.POINTER(C.c_float), C.c_int, C.POINTER(abi.NraysStats)]
        l.nrays_oracle_render_timed.restype = C.c_int
        l.nrays_oracle_render_timed.argtypes = [C.POINTER(abi.NraysSceneDesc), C.POINTER(abi.NraysRenderParams),
                                                C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(abi.NraysStats),
                                                C.POINTER(C.c_double)]
        l.nrays_oracle_last_thread_cpu.restype = None
        l.nrays_oracle_last_thread_cpu.argtypes = [C.POINTER(C.c_double)]
        l.nrays_oracle_cast_batch.restype = C.c_int
        l.nrays_oracle_cast_batch.argtypes = [C.POINTER(abi.NraysSceneDesc), C.c_uint32, C.POINTER(C.c_double),
                                              C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double),
                                              C.POINTER(C.c_int32)]
        l.nrays_oracle_shadow.restype = C.c_int
        l.nrays_oracle_shadow.argtypes = [C.POINTER(abi.NraysSceneDesc), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.c_double, C.POINTER(C.c_float)]
        l.nrays_oracle_tex_sample.restype = None
        l.nrays_oracle_tex_sample.argtypes = [C.POINTER(abi.NraysTexture), C.c_double, C.c_double, C.POINTER(C.c_float)]
        l.nrays_oracle_node_aabb.restype = C.c_int
        l.nrays_oracle_node_aabb.argtypes = [C.POINTER(abi.NraysSceneDesc), C.c_uint32, C.POINTER(C.c_double)]
        l.nrays_oracle_rng_u01.restype = C.c_double
        l.nrays_oracle_rng_u01.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
        l.nrays_oracle_last_error.restype = C.c_char_p
        _lib = l
    return _lib


def render(descriptor, params, num_threads=1):
    """Oracle counterpart of nrays_render.  Returns (image (rows, W, 3) float32, NraysStats)."""
    l = lib()
    rows = params.height
    if params.band_rows and params.band_owners > 1:
        nb = (params.height + params.band_rows - 1) // params.band_rows
        rows = ((nb + params.band_owners - 1) // params.band_owners) * params.band_rows
    out = np.zeros((rows, params.width, 3), dtype=np.float32)
    st = abi.NraysStats()
    rc = l.nrays_oracle_render(descriptor.pointer(), C.byref(params), out.ctypes.data_as(C.POINTER(C.c_float)),
                               int(num_threads), C.byref(st))
    if rc != 0:
        raise RuntimeError("oracle render failed: %d %s" % (rc, l.nrays_oracle_last_error()))
    return out, st


def render_timed(descriptor, params, num_threads, reps):
    """bench.py's cpu_baseline leg: `reps` frames on persistent threads with the scene built before the clock starts.
    Returns (seconds of the threaded region, NraysStats summed over the `reps` frames)."""
    l = lib()
    rows = params.height
    if params.band_rows and params.band_owners > 1:
        nb = (params.height + params.band_rows - 1) // params.band_rows
        rows = ((nb + params.band_owners - 1) // params.band_owners) * params.band_rows
    out = np.zeros((rows, params.width, 3), dtype=np.float32)
    st = abi.NraysStats()
    sec = C.c_double(0.0)
    rc = l.nrays_oracle_render_timed(descriptor.pointer(), C.byref(params), out.ctypes.data_as(C.POINTER(C.c_float)),
                                     int(num_threads), int(reps), C.byref(st), C.byref(sec))
    if rc != 0:
        raise RuntimeError("oracle render failed: %d %s" % (rc, l.nrays_oracle_last_error()))
    return sec.value, st


def last_thread_cpu():
    """CPU seconds of the threads of the last render_timed call of this thread: (sum, min, max)."""
    v = (C.c_double * 3)()
    lib().nrays_oracle_last_thread_cpu(v)
    return v[0], v[1], v[2]


def cast(descriptor, origins, dirs, bruteforce=False):
    """Closest hits (ClosestRayTOICostFn) of a batch of rays.  Returns (hit mask, (n, 8) array of
    toi, nx, ny, nz, has_uv, u, v, node)."""
    o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
    d = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
    n = len(o)
    out = np.zeros((n, 8), dtype=np.float64)
    hit = np.zeros(n, dtype=np.int32)
    rc = lib().nrays_oracle_cast_batch(descriptor.pointer(), n, o.ctypes.data_as(C.POINTER(C.c_double)),
                                       d.ctypes.data_as(C.POINTER(C.c_double)), 1 if bruteforce else 0,
                                       out.ctypes.data_as(C.POINTER(C.c_double)), hit.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        raise RuntimeError("oracle cast failed: %d" % rc)
    return hit.astype(bool), out


def shadow(descriptor, origin, direction, maxtoi):
    """Scene::intersects_ray: returns None if blocked, else the f32 colour filter."""
    o = (C.c_double * 3)(*origin)
    d = (C.c_double * 3)(*direction)
    f = (C.c_float * 3)()
    lit = lib().nrays_oracle_shadow(descriptor.pointer(), o, d, float(maxtoi), f)
    return np.array(f[:], dtype=np.float32) if lit == 1 else None


def tex_sample(texture, u, v):
    """Texture2d::sample on an nrays_amd.scene.Texture2d."""
    t = abi.NraysTexture()
    t.width, t.height = texture.data.dims
    t.format, t.interp, t.overflow = texture.data.format, texture.interpol, texture.overflow
    t.texels = texture.data.pixels.ctypes.data
    out = (C.c_float * 4)()
    lib().nrays_oracle_tex_sample(C.byref(t), float(u), float(v), out)
    return np.array(out[:], dtype=np.float32)


def node_aabb(descriptor, i):
    out = (C.c_double * 6)()
    rc = lib().nrays_oracle_node_aabb(descriptor.pointer(), int(i), out)
    if rc != 0:
        raise RuntimeError("bad node")
    return np.array(out[:])


def rng_u01(seed, pixel, sample, dim):
    return lib().nrays_oracle_rng_u01(seed, pixel, sample, dim)
