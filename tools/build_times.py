#!/usr/bin/env python
"""Scene-build phases of the hairball stand-in (GPU box): NRAYS_BUILD_TIMES=1 makes nrays_scene_create print presplit / build_bvh times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["NRAYS_BUILD_TIMES"] = "1"
import torch
from tools import standins
for name, make in (("hairball", standins.hairball_scene), ("hairball again", standins.hairball_scene), ("sponza", standins.sponza_scene)):
    t = time.perf_counter(); sc, cam = make(); t1 = time.perf_counter()
    h = sc.device_handle(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%s: python scene %.2f s, nrays_scene_create %.4f s" % (name, t1 - t, t2 - t1), flush=True)
