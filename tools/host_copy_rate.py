#!/usr/bin/env python
"""PCIe-inclusive frame times of the host-buffer entry points (GPU box): nrays_render (float frame, 24.9 MB at 1080p) and
nrays_render_rgb8 (bytes, 6.2 MB), pageable numpy buffers, balls scene.   python tools/host_copy_rate.py"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import nrays_amd as nr
from nrays_amd import abi
from tools import scenes_util as su
lib = abi.load_hip_lib()
sc, cam = su.balls_scene()
p, _ = su.camera_params(cam, 1920, 1080)
f = np.empty((1080, 1920, 3), np.float32); q = np.empty((1080, 1920, 3), np.uint8)
res = {}
for name, fn, buf, ty in (("nrays_render", lib.nrays_render, f, C.c_float), ("nrays_render_rgb8", lib.nrays_render_rgb8, q, C.c_uint8)):
    for _ in range(5): abi.check(fn(sc.device_handle(), C.byref(p), buf.ctypes.data_as(C.POINTER(ty))))
    t0 = time.perf_counter()
    for _ in range(50): abi.check(fn(sc.device_handle(), C.byref(p), buf.ctypes.data_as(C.POINTER(ty))))
    dt = (time.perf_counter() - t0) / 50
    res[name] = {"ms_per_frame": round(dt * 1e3, 3), "host_bytes": int(buf.nbytes), "GBs": round(buf.nbytes / dt / 1e9, 1), "Mrays_s": round(nr.get_stats(sc).total_rays() / dt / 1e6, 1)}
print(json.dumps(res))
