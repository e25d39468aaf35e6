#!/usr/bin/env python
"""First frames of FRESH handles (what a caller that renders each camera once gets): mean of the first / second / fifth frame over a
few handles, host-synchronised.   python tools/cold_frame.py [sponza|sponza8|hairball|balls] ..."""
import ctypes as C, sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from nrays_amd import abi
from tools import scenes_util as su, standins
lib = abi.load_hip_lib()
for name in sys.argv[1:] or ["sponza"]:
    mk = {"sponza": standins.sponza_scene, "sponza8": lambda: standins.sponza_scene(n_lights=8), "config4": lambda: standins.sponza_scene(n_lights=8),
          "hairball": standins.hairball_scene, "balls": su.balls_scene}[name]
    W, H = (3840, 2160) if name == "config4" else (1920, 1080)
    t = [[], [], []]
    for rep in range(5 if name == "config4" else 9):
        sc, cam = mk()
        p, _ = su.camera_params(cam, W, H)
        out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        h = sc.device_handle(); torch.cuda.synchronize()
        for k in range(5):
            t0 = time.perf_counter()
            abi.check(lib.nrays_render_device(h, C.byref(p), C.c_void_p(out.data_ptr()), None)); torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            if rep and k in (0, 1, 4): t[(0, 1, None, None, 2)[k]].append(dt)
    print(json.dumps({"scene": name, "cost_seed": os.environ.get("NRAYS_COST_SEED", "1"), "first_ms": round(sum(t[0]) / len(t[0]), 3), "second_ms": round(sum(t[1]) / len(t[1]), 3), "fifth_ms": round(sum(t[2]) / len(t[2]), 3)}), flush=True)
