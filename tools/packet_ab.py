#!/usr/bin/env python
"""A/B of packet traversal for anti-aliased frames (GPU box): the hairball stand-in at 64 / 16 / 4 samples per pixel with NRAYS_PACKET off and
at several thresholds; every variant must render the bit-identical frame.   python tools/packet_ab.py [--full]  (--full: config 5, 4K x 64 spp)"""
import ctypes as C, hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import nrays_amd as nr
from nrays_amd import abi
from tools import standins, scenes_util as su
lib = abi.load_hip_lib()
full = "--full" in sys.argv
cases = [("hairball 1080p 64 spp", standins.hairball_scene, 1920, 1080, 64, 2), ("hairball 1080p 16 spp", standins.hairball_scene, 1920, 1080, 16, 3), ("hairball 1080p 4 spp", standins.hairball_scene, 1920, 1080, 4, 5),
         ("sponza 1080p 16 spp", standins.sponza_scene, 1920, 1080, 16, 3)]
if full: cases = [("config 5: hairball 4K 64 spp", standins.hairball_scene, 3840, 2160, 64, 1)] + cases
for name, make, w, h, spp, steps in cases:
    ref = None
    for setting in (None, "6", "4", "2"):
        if setting is None: os.environ.pop("NRAYS_PACKET", None)
        else: os.environ["NRAYS_PACKET"] = setting
        if setting is not None and (1 << int(setting)) > spp: continue
        sc, cam = make()
        p, _ = su.camera_params(cam, w, h, spp=spp, window=1.0, seed=1)
        out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
        hd = sc.device_handle()
        abi.check(lib.nrays_render_device(hd, C.byref(p), C.c_void_p(out.data_ptr()), None))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): abi.check(lib.nrays_render_device(hd, C.byref(p), C.c_void_p(out.data_ptr()), None))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        digest = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]
        if ref is None: ref = digest
        st = nr.get_stats(sc)
        print(json.dumps({"case": name, "NRAYS_PACKET": setting, "ms": round(dt * 1e3, 3), "frame_sha1": digest, "identical_to_per_lane": digest == ref, "rays": st.total_rays()}), flush=True)
        del sc
