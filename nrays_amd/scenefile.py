"""ctypes binding of the C++ loader3d front-end (nrays_amd/host → lib/libnrays_host.so): `.scene`,
`.mtl`, `.obj` parsing and the PNG codec.  A FileScene exposes the same two members the render and
oracle entry points need from nrays_amd.scene.Scene: `.descriptor.pointer()` and `.device_handle()`."""
import ctypes as C
import os

import numpy as np

from . import abi


class NraysHostCamera(C.Structure):
    _fields_ = [("eye", C.c_double * 3), ("at", C.c_double * 3), ("fovy", C.c_double), ("resolution", C.c_double * 2),
                ("aa", C.c_double * 2), ("output", C.c_char * 256)]


_lib = None


def host_lib():
    global _lib
    if _lib is None:
        if not os.path.exists(abi.HOST_LIB_PATH):
            raise ImportError("libnrays_host.so is not built; run __graft_entry__.build()")
        l = C.CDLL(abi.HOST_LIB_PATH)
        l.nrays_host_load_scene.restype = C.c_int
        l.nrays_host_load_scene.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        l.nrays_host_scene_desc.restype = C.POINTER(abi.NraysSceneDesc)
        l.nrays_host_scene_desc.argtypes = [C.c_void_p]
        l.nrays_host_num_cameras.restype = C.c_uint32
        l.nrays_host_num_cameras.argtypes = [C.c_void_p]
        l.nrays_host_camera.restype = C.c_int
        l.nrays_host_camera.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(NraysHostCamera)]
        l.nrays_host_inverse_projection.restype = C.c_int
        l.nrays_host_inverse_projection.argtypes = [C.POINTER(NraysHostCamera), C.c_double, C.c_double, C.POINTER(C.c_double)]
        l.nrays_host_num_warnings.restype = C.c_uint32
        l.nrays_host_num_warnings.argtypes = [C.c_void_p]
        l.nrays_host_warning.restype = C.c_char_p
        l.nrays_host_warning.argtypes = [C.c_void_p, C.c_uint32]
        l.nrays_host_free_scene.restype = None
        l.nrays_host_free_scene.argtypes = [C.c_void_p]
        l.nrays_host_last_error.restype = C.c_char_p
        l.nrays_host_write_png.restype = C.c_int
        l.nrays_host_write_png.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_uint32, C.c_uint32]
        l.nrays_host_write_ppm.restype = C.c_int
        l.nrays_host_write_ppm.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_uint32, C.c_uint32]
        l.nrays_host_read_png.restype = C.c_int
        l.nrays_host_read_png.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        l.nrays_host_read_image.restype = C.c_int
        l.nrays_host_read_image.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        _lib = l
    return _lib


class _Descriptor:
    def __init__(self, ptr):
        self._ptr = ptr
        self.desc = ptr.contents

    def pointer(self):
        return self._ptr


class FileScene:
    """A scene loaded from a `.scene` file by the C++ front-end (examples/loader3d.rs:57-65)."""

    def __init__(self, path, allow_standins=False):
        l = host_lib()
        h = C.c_void_p()
        if l.nrays_host_load_scene(os.fsencode(path), 1 if allow_standins else 0, C.byref(h)) != 0:
            raise RuntimeError(l.nrays_host_last_error().decode())
        self._h = h
        self.descriptor = _Descriptor(l.nrays_host_scene_desc(h))
        self.warnings = [l.nrays_host_warning(h, i).decode() for i in range(l.nrays_host_num_warnings(h))]
        self.cameras = []
        for i in range(l.nrays_host_num_cameras(h)):
            c = NraysHostCamera()
            l.nrays_host_camera(h, i, C.byref(c))
            self.cameras.append(c)
        self._handle = None

    def camera_dict(self, i=0):
        c = self.cameras[i]
        return dict(eye=tuple(c.eye), at=tuple(c.at), fovy=c.fovy, resolution=(int(c.resolution[0]), int(c.resolution[1])),
                    aa=(int(c.aa[0]), c.aa[1]), output=c.output.decode())

    def inverse_projection(self, i, width, height):
        out = (C.c_double * 16)()
        if host_lib().nrays_host_inverse_projection(C.byref(self.cameras[i]), float(width), float(height), out) != 0:
            raise RuntimeError(host_lib().nrays_host_last_error().decode())
        return np.array(out[:]).reshape(4, 4).T  # column-major -> row-major 4x4

    def device_handle(self):
        if self._handle is None:
            lib = abi.load_hip_lib()
            h = C.c_void_p()
            abi.check(lib.nrays_scene_create(self.descriptor.pointer(), C.byref(h)))
            self._handle = h
        return self._handle

    def close(self):
        if self._handle is not None:
            abi.load_hip_lib().nrays_scene_destroy(self._handle)
            self._handle = None
        if self._h is not None:
            host_lib().nrays_host_free_scene(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_png(path, image):
    """Image::to_png (src/image.rs:60-90): c*255 clamped to [0,255], truncated to u8, RGB8."""
    img = np.ascontiguousarray(image, dtype=np.float32)
    if host_lib().nrays_host_write_png(os.fsencode(path), img.ctypes.data_as(C.POINTER(C.c_float)), img.shape[1], img.shape[0]) != 0:
        raise RuntimeError(host_lib().nrays_host_last_error().decode())


def read_png(path):
    w, h = C.c_uint32(), C.c_uint32()
    ch = host_lib().nrays_host_read_png(os.fsencode(path), None, 0, C.byref(w), C.byref(h))
    if ch < 0:
        raise RuntimeError(host_lib().nrays_host_last_error().decode())
    buf = np.empty((h.value, w.value, ch), dtype=np.uint8)
    host_lib().nrays_host_read_png(os.fsencode(path), buf.ctypes.data, buf.nbytes, C.byref(w), C.byref(h))
    return buf


def read_image(path):
    """What stb_image's load hands to Texture2d::from_png (src/texture2d.rs:95): PNG, JPEG, BMP or TGA, recognised by content;
    (H, W, channels) uint8, top row first."""
    w, h = C.c_uint32(), C.c_uint32()
    ch = host_lib().nrays_host_read_image(os.fsencode(path), None, 0, C.byref(w), C.byref(h))
    if ch < 0:
        raise RuntimeError(host_lib().nrays_host_last_error().decode())
    buf = np.empty((h.value, w.value, ch), dtype=np.uint8)
    host_lib().nrays_host_read_image(os.fsencode(path), buf.ctypes.data, buf.nbytes, C.byref(w), C.byref(h))
    return buf
