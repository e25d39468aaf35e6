#!/usr/bin/env python
"""Config 5 (hairball stand-in, 64 spp, window 1.0) at a given resolution under the scheduling switches of the environment
(NRAYS_GRAB, NRAYS_LPT, NRAYS_LANE_LOG2): does the per-tile dequeue bound a frame of 8 M one-pixel wave tiles?
  NRAYS_GRAB=4 NRAYS_LPT=0 python tools/aa_grab.py [width height spp [hairball|sponza|sponza8]]"""
import ctypes as C, sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import nrays_amd as nr
from nrays_amd import abi
from tools import scenes_util as su, standins
lib = abi.load_hip_lib()
w, h, spp = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160, 64)
scene = sys.argv[4] if len(sys.argv) > 4 else "hairball"
sc, cam = standins.hairball_scene() if scene == "hairball" else (su.balls_scene() if scene == "balls" else standins.sponza_scene(n_lights=8 if scene == "sponza8" else 1))
p, _ = su.camera_params(cam, w, h, spp=spp, window=1.0, seed=1)
out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
hd = sc.device_handle()
for _ in range(5):
    abi.check(lib.nrays_render_device(hd, C.byref(p), C.c_void_p(out.data_ptr()), None))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2):
    abi.check(lib.nrays_render_device(hd, C.byref(p), C.c_void_p(out.data_ptr()), None))
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
print(json.dumps({"scene": scene, "env": {k: os.environ[k] for k in ("NRAYS_GRAB", "NRAYS_LPT", "NRAYS_LANE_LOG2", "NRAYS_LPT_ANALYTIC") if k in os.environ}, "res": [w, h], "spp": spp,
                  "ms": round(dt * 1e3, 2), "checksum": float(out.double().sum())}), flush=True)
