#!/usr/bin/env python
"""How concentrated a mesh frame's work is: distribution of the per-wave-tile cycle counts k_primary records for the
cost-ordered work lists.  Needs a -DNR_DEBUG_TILE_COSTS (or -DNR_PHASE_TIMING) build, which exports nrays_debug_tile_costs.
  NRAYS_HIP_LIB=nrays_amd/lib/v/tc.so python tools/tile_costs.py sponza hairball balls"""
import ctypes as C, json, os, sys
os.environ.setdefault("NRAYS_LPT_ANALYTIC", "1")  # analytic scenes only record tile costs with their (rejected) cost-ordered lists on
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from nrays_amd import abi
from tools import scenes_util as su, standins
lib = abi.load_hip_lib()
for name in sys.argv[1:] or ["sponza"]:
    sc, cam = {"sponza": standins.sponza_scene, "hairball": standins.hairball_scene, "sponza8": lambda: standins.sponza_scene(n_lights=8),
               "balls": su.balls_scene, "primitives": lambda: su.primitives_scene(0.0, 1)}[name]()
    p, _ = su.camera_params(cam, 1920, 1080)
    out = torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda")
    for _ in range(8 if os.environ.get("NRAYS_DEBUG_RECORD_ALWAYS") else 3):  # with NRAYS_DEBUG_RECORD_ALWAYS=1: the costs of a steady-state frame
        abi.check(lib.nrays_render_device(sc.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
    cap = 1 << 20
    buf = np.zeros(cap, np.uint32); n = C.c_uint32()
    lib.nrays_debug_tile_costs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    abi.check(lib.nrays_debug_tile_costs(sc.device_handle(), buf.ctypes.data, cap, C.byref(n)))
    c = np.sort(buf[:n.value].astype(np.float64) * 16)[::-1]
    tot = c.sum()
    res = {"scene": name, "wave_tiles": int(n.value), "total_wave_cycles": float(tot), "max_tile_cycles": float(c[0]), "median_tile_cycles": float(np.median(c))}
    res["top10_tile_cycles"] = [float(x) for x in c[:10]]
    res["tiles_above_half_of_max"] = int((c > 0.5 * c[0]).sum()); res["tiles_above_a_tenth_of_max"] = int((c > 0.1 * c[0]).sum())
    for q in (0.001, 0.01, 0.05, 0.10, 0.25, 0.5):
        k = max(1, int(q * len(c)))
        res["share_of_top_%g%%" % (q * 100)] = round(float(c[:k].sum() / tot), 3)
    print(json.dumps(res), flush=True)
