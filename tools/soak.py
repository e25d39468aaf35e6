#!/usr/bin/env python
"""Soak run (GPU box): many frames per scene handle, alternating resolutions / cameras / sample counts, checking that
every repeat of a configuration reproduces its first image bit for bit, that statistics stay constant and that
device memory does not grow.  Exit code 1 on any deviation."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import nrays_amd as nr
from nrays_amd import abi
from tools import scenes_util as su, standins
lib = abi.load_hip_lib()

def soak(name, sc, cam, configs, rounds):
    ref = {}
    bad = 0
    free0 = None
    mid = 0
    t0 = time.time()
    n = 0
    outs = [torch.empty((h, w, 3), dtype=torch.float32, device="cuda") for (w, h, _) in configs]  # no allocator traffic in the loop
    for r in range(rounds):
        for ci, (w, h, kw) in enumerate(configs):
            p, _ = su.camera_params(cam if "cam" not in kw else kw["cam"], w, h, **{k: v for k, v in kw.items() if k != "cam"})
            out = outs[ci]
            abi.check(lib.nrays_render_device(sc.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
            n += 1
            if r % 50 == 0 or r == rounds - 1:
                st = nr.get_stats(sc)
                key = (ci,)
                sig = (out.double().sum().item(), st.total_rays())
                if key not in ref:
                    ref[key] = (out.clone(), sig)
                else:
                    if not torch.equal(out, ref[key][0]) or sig != ref[key][1]:
                        bad += 1
                        print("DEVIATION", name, "config", ci, "round", r, sig, ref[key][1])
        if r == 2:
            free0, res0 = torch.cuda.mem_get_info()[0], torch.cuda.memory_reserved()
        if r == rounds // 2 and free0 is not None:
            mid = (free0 - torch.cuda.mem_get_info()[0]) - (torch.cuda.memory_reserved() - res0)
        if r in (10, 50, 100, 200) and free0 is not None:  # one-off lazy allocations show as a step, a leak as a slope
            print("  %s round %d: drift %d bytes" % (name, r, (free0 - torch.cuda.mem_get_info()[0]) - (torch.cuda.memory_reserved() - res0)), flush=True)
    free1, res1 = torch.cuda.mem_get_info()[0], torch.cuda.memory_reserved()
    # device memory that went away and is not held by torch's caching allocator (the checks above allocate temporaries)
    leak = ((free0 - free1) - (res1 - res0)) if free0 is not None else 0
    print("%s: %d frames in %.1f s, deviations %d, device memory drift %d bytes" % (name, n, time.time() - t0, bad, leak), flush=True)
    # a LEAK grows with the frames: what counts is the growth over the second half of the run (the HIP runtime itself takes a
    # 16 MiB step after a few hundred launches of a process — with or without this library's lazily allocated buffers)
    # ... and an absolute bound on top (a leak per camera / configuration change would all land in the first half): the documented
    # 16 MiB runtime step plus 8 MiB
    return bad + (1 if (leak - mid > (4 << 20) or leak > (24 << 20)) else 0)

bad = 0
sc, cam = su.balls_scene()
bad += soak("balls", sc, cam, [(1920, 1080, {}), (640, 360, {}), (333, 77, dict(spp=4, window=1.0, seed=3)),
                               (640, 360, dict(cam=dict(cam, eye=(2.0, 4.0, -9.0))))], 1500)
sc, cam = su.primitives_scene()
bad += soak("primitives", sc, cam, [(800, 600, {}), (320, 240, dict(spp=2, window=1.0, seed=1))], 300)
sc, cam = standins.sponza_scene()
bad += soak("sponza", sc, cam, [(1920, 1080, {}), (480, 270, {}), (480, 270, dict(max_depth=2))], 150)
sys.exit(1 if bad else 0)
