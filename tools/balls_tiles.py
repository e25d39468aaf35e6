#!/usr/bin/env python
"""Steady-state wave-tile costs of the balls frame by reflection depth (GPU box; -DNR_DEBUG_TILE_COSTS build, NRAYS_DEBUG_RECORD_ALWAYS=1):
distribution, and where the most expensive tiles are."""
import ctypes as C, json, os, sys
os.environ["NRAYS_DEBUG_RECORD_ALWAYS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from nrays_amd import abi
from tools import scenes_util as su
lib = abi.load_hip_lib()
lib.nrays_debug_tile_costs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
for depth in (1, 4):
    sc, cam = su.balls_scene()
    p, _ = su.camera_params(cam, 1920, 1080, max_depth=depth)
    out = torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda")
    for _ in range(10): abi.check(lib.nrays_render_device(sc.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
    buf = np.zeros(1 << 20, np.uint32); n = C.c_uint32()
    abi.check(lib.nrays_debug_tile_costs(sc.device_handle(), buf.ctypes.data, 1 << 20, C.byref(n)))
    c = buf[:n.value].astype(np.float64) * 16
    nz = c[c > 0]
    order = np.argsort(c)[::-1]
    print(json.dumps({"max_depth": depth, "tiles": int(n.value), "tiles_with_cost": int(len(nz)), "sum_cycles": float(c.sum()),
                      "percentiles_50_90_99_100": [float(x) for x in np.percentile(nz, [50, 90, 99, 100])],
                      "tiles_above_20k_cycles": int((c > 20000).sum()), "tiles_above_40k": int((c > 40000).sum()),
                      "top8": [[int(i), float(c[i])] for i in order[:8]]}), flush=True)
