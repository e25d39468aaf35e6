#!/usr/bin/env python
"""Per-wave-tile cycle counts for offline scheduling experiments -> gpurun_out/r06/tiles_<scene>.npz:
  cold    the FIRST frame of a fresh handle (analytic scenes: image-order lists; mesh scenes: k_seed_costs' order)
  seed    k_seed_costs' guess for that frame (mesh scenes)
  exact   the frame that sorts the camera's own recorded costs (what a resting camera keeps)
  moved1 / moved   the same for the cameras 1 and 8 bench-steps further along bench.py's moving path
Needs a -DNR_DEBUG_TILE_COSTS build (exports nrays_debug_tile_costs / nrays_debug_seed_costs).
  NRAYS_HIP_LIB=nrays_amd/lib/v/tc.so python tools/tile_dump.py balls sponza"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from nrays_amd import abi
from tools import scenes_util as su, standins
lib = abi.load_hip_lib()
os.makedirs(os.path.join(ROOT, "gpurun_out", "r06"), exist_ok=True)
for f in (lib.nrays_debug_tile_costs, lib.nrays_debug_seed_costs):
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]


def grab(fn, h, tiles):
    buf = np.zeros(1 << 20, np.uint32); n = C.c_uint32()
    if fn(h, buf.ctypes.data, 1 << 20, C.byref(n)) != 0:
        return None
    return buf[:tiles].copy()


for name in sys.argv[1:] or ["balls"]:
    make = {"sponza": standins.sponza_scene, "hairball": standins.hairball_scene, "sponza8": lambda: standins.sponza_scene(n_lights=8),
            "balls": su.balls_scene, "primitives": lambda: su.primitives_scene(0.0, 1)}[name]
    res = {}
    for tag, k in (("", 0), ("moved1", 1), ("moved", 8)):
        sc, cam = make()
        eye0 = np.array(cam["eye"], dtype=np.float64); at = np.array(cam["at"], dtype=np.float64)
        cam = dict(cam, eye=tuple(eye0 + k * 1e-3 * np.linalg.norm(eye0 - at) * np.array([1.0, 0.0, 0.0])))
        p, _ = su.camera_params(cam, 1920, 1080)
        out = torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda")
        h = sc.device_handle()
        abi.check(lib.nrays_render_device(h, C.byref(p), C.c_void_p(out.data_ptr()), None)); torch.cuda.synchronize()
        tc = abi.NraysTileCosts(); abi.check(lib.nrays_get_tile_costs(h, C.byref(tc)))
        n = int(tc.tiles)
        if not tag:
            res["cold"] = grab(lib.nrays_debug_tile_costs, h, n)
            seed = grab(lib.nrays_debug_seed_costs, h, n)
            if seed is not None: res["seed"] = seed
        abi.check(lib.nrays_render_device(h, C.byref(p), C.c_void_p(out.data_ptr()), None)); torch.cuda.synchronize()
        res[tag if tag else "exact"] = grab(lib.nrays_debug_tile_costs, h, n)
        res["waves"] = np.array([int(tc.resident_waves)])
    np.savez(os.path.join(ROOT, "gpurun_out", "r06", "tiles_%s.npz" % name), **res)
    print(json.dumps({"scene": name, "tiles": n, "keys": sorted(res)}), flush=True)
