#!/usr/bin/env python
"""A/B of the staged ("wavefront") path against the megakernel (GPU box): every workload in its own process with
NRAYS_WAVEFRONT=0 and =1 (optionally for several builds of the library); prints ms per frame and a checksum of the
frame's bits (equal checksums = identical frames).

  python tools/wf_ab.py [--libs a.so,b.so] [--work sponza,sponza8,hairball,sponza4k,config4,config5] [--steps 20] [--modes 0,1]
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WORK = {  # name: (scene, lights, width, height, camera kwargs, steps divisor)
    "sponza": ("sponza", 1, 1920, 1080, {}, 1), "sponza8": ("sponza", 8, 1920, 1080, {}, 1), "hairball": ("hairball", 1, 1920, 1080, {}, 1),
    "sponza4k": ("sponza", 1, 3840, 2160, {}, 2), "config4": ("sponza", 8, 3840, 2160, {}, 4),
    "hair16": ("hairball", 1, 1920, 1080, dict(spp=16, window=1.0, seed=1), 4),
    "config5": ("hairball", 1, 3840, 2160, dict(spp=64, window=1.0, seed=1), 20),
}


def child(name, steps):
    import torch
    import nrays_amd as nr
    from nrays_amd import abi
    from tools import scenes_util as su, standins
    scene, lights, w, h, kw, div = WORK[name]
    sc, cam = standins.sponza_scene(n_lights=lights) if scene == "sponza" else standins.hairball_scene()
    lib = abi.load_hip_lib()
    p, _ = su.camera_params(cam, w, h, **kw)
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    hd = sc.device_handle()
    t0 = time.perf_counter()
    abi.check(lib.nrays_render_device(hd, C.byref(p), C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize()
    cold = time.perf_counter() - t0
    st = nr.get_stats(sc)
    for _ in range(3 if div < 20 else 0):
        abi.check(lib.nrays_render_device(hd, C.byref(p), C.c_void_p(out.data_ptr()), None))
    n = max(1, steps // div)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        abi.check(lib.nrays_render_device(hd, C.byref(p), C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    ts = nr.get_stats(sc)
    chk = int(out.view(torch.int32).to(torch.int64).sum().item())
    print(json.dumps({"work": name, "wavefront": os.environ.get("NRAYS_WAVEFRONT", "auto"), "ms": round(dt * 1e3, 4), "cold_ms": round(cold * 1e3, 3),
                      "gpu_ms": round(ts.kernel_ms_total, 4), "rays": st.total_rays(), "shadow": st.rays_shadow, "gens": st.generations, "checksum": chk}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="")
    ap.add_argument("--work", default="sponza,sponza8,hairball,config4,config5")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--modes", default="0,1")
    ap.add_argument("--child", default="")
    a = ap.parse_args()
    if a.child:
        child(a.child, a.steps)
        return
    for lib in ([("" if l == "default" else l) for l in a.libs.split(",") if l] or [""]):
        print("== lib:", lib or "default", flush=True)
        for wk in a.work.split(","):
            for mode in a.modes.split(","):
                env = dict(os.environ, NRAYS_WAVEFRONT=mode)
                if lib:
                    env["NRAYS_HIP_LIB"] = os.path.abspath(lib)
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", wk, "--steps", str(a.steps)], env=env,
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
                sys.stdout.write(r.stdout)
                if r.returncode != 0:
                    print("FAILED", wk, mode, r.stderr[-1500:], flush=True)


if __name__ == "__main__":
    main()
