#!/usr/bin/env python
"""Sweeps the BLAS builder's quality knobs on the GPU box (the device builder reads them from the environment per scene:
NRAYS_PRESPLIT_BUDGET[_HAIRY], NRAYS_PRESPLIT_MINGAIN[_HAIRY], NRAYS_PRIM_COST[_HAIRY], NRAYS_MAX_LEAF): frame time, AABB / triangle
tests per ray, references and build time of one scene per setting.

  python tools/build_sweep.py hairball NRAYS_PRESPLIT_BUDGET_HAIRY=5,8,12 NRAYS_PRIM_COST_HAIRY=0.5,0.7,1.0
"""
import ctypes as C
import itertools
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import nrays_amd as nr
from nrays_amd import abi
from tools import scenes_util as su, standins


def measure(scene_name, w=1920, h=1080, steps=20):
    make = {"hairball": standins.hairball_scene, "sponza": standins.sponza_scene, "sponza8": lambda: standins.sponza_scene(n_lights=8), "hair300": lambda: standins.hairball_scene(strands=300)}[scene_name]
    sc, cam = make()
    lib = abi.load_hip_lib()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hd = sc.device_handle()
    torch.cuda.synchronize(); build = time.perf_counter() - t0
    p, _ = su.camera_params(cam, w, h)
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    abi.check(lib.nrays_render_device_instrumented(hd, C.byref(p), C.c_void_p(out.data_ptr()), None))
    st = nr.get_stats(sc)
    for _ in range(4):
        abi.check(lib.nrays_render_device(hd, C.byref(p), C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        abi.check(lib.nrays_render_device(hd, C.byref(p), C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / steps * 1e3
    return dict(ms=round(ms, 4), node_per_ray=round(st.node_tests / st.total_rays(), 1), tri_per_ray=round(st.tri_tests / st.total_rays(), 2), build_s=round(build, 3),
                scene_mb=round(lib.nrays_scene_device_bytes(hd) / 1e6, 1))


def main():
    scene = sys.argv[1]
    knobs = [a.split("=") for a in sys.argv[2:]]
    names = [k for k, _ in knobs]
    measure(scene, steps=2)  # warm the process (first hipMemcpy, kernel load)
    for combo in itertools.product(*[v.split(",") for _, v in knobs]):
        for k, v in zip(names, combo):
            os.environ[k] = v
        r = measure(scene)
        r.update(dict(zip(names, combo)), scene=scene)
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
