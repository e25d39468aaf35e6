#!/usr/bin/env python
"""Kernel tuning harness (GPU box): times the HIP path on the BASELINE scenes for one or several
builds of libnrays_hip.so (A/B in one gpurun call; each build runs in its own process).

  python tools/kbench.py [--libs a.so,b.so] [--scenes balls,sponza,hairball] [--steps 20]
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


def run_one(scene_name, steps, width, height, check):
    import numpy as np
    import torch
    import nrays_amd as nr
    from nrays_amd import abi
    from tools import scenes_util as su, standins
    torch.cuda.set_device(0)
    lib = abi.load_hip_lib()
    if scene_name == "balls":
        sc, cam = su.balls_scene()
    elif scene_name == "ballsaway":  # every wave tile misses the scene: the cheap path alone
        sc, cam = su.balls_scene()
        cam = dict(cam, at=(0.0, 5.0, -30.0))
    elif scene_name == "ballszoom":  # every pixel starts a deep reflection chain between the balls (use with a small frame)
        sc, cam = su.balls_scene()
        cam = dict(eye=(0.0, 0.6, -6.0), at=(1.05, 0.0, 0.0), fovy=6.0)
    elif scene_name == "ballsfar":  # the scene covers a few wave tiles only: the cost of a tile outside its screen bounds
        sc, cam = su.balls_scene()
        cam = dict(cam, eye=(0.0, 150.0, -300.0))
    elif scene_name == "sponza":
        sc, cam = standins.sponza_scene()
    elif scene_name in ("sponza8", "config4"):
        sc, cam = standins.sponza_scene(n_lights=8)
    elif scene_name in ("hairball", "config5"):
        sc, cam = standins.hairball_scene()
    elif scene_name == "primitives":
        sc, cam = su.primitives_scene(0.0, 1)
    else:
        raise SystemExit("unknown scene " + scene_name)
    t0 = time.perf_counter()
    h = sc.device_handle()
    t_build = time.perf_counter() - t0
    p, _ = su.camera_params(cam, width, height, **(dict(spp=64, window=1.0, seed=1) if scene_name == "config5" else {}))  # config5: aa 64 1.0
    out = torch.empty((height, width, 3), dtype=torch.float32, device="cuda")

    def render(instr=False):
        fn = lib.nrays_render_device_instrumented if instr else lib.nrays_render_device
        abi.check(fn(h, C.byref(p), C.c_void_p(out.data_ptr()), None))
    render(True)
    st = nr.get_stats(sc)
    for _ in range(4):  # the per-camera scheduling state of a handle settles in three plain frames
        render()
    nr.get_stats(sc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        render()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ts = nr.get_stats(sc)
    res = {"scene": scene_name, "res": [width, height], "ms": round(dt * 1e3, 4), "mrays_s": round(st.total_rays() / dt / 1e6, 1),
           "rays": st.total_rays(), "primary_ms": round(ts.kernel_ms_primary, 4), "gpu_ms": round(ts.kernel_ms_total, 4),
           "node_per_ray": round(st.node_tests / st.total_rays(), 1), "tri_per_ray": round(st.tri_tests / st.total_rays(), 2),
           "gens": st.generations, "build_s": round(t_build, 2), "shadow": st.rays_shadow, "shadow_not_traced": ts.rays_shadow_elided,
           "GBs_alg": round(st.algorithmic_bytes(width, height) / (ts.kernel_ms_total * 1e-3) / 1e9, 1) if ts.kernel_ms_total > 0 else None}
    if check:
        import oracle
        cw, ch = 160, 90
        pc, _ = su.camera_params(cam, cw, ch)
        oc = torch.empty((ch, cw, 3), dtype=torch.float32, device="cuda")
        abi.check(lib.nrays_render_device(h, C.byref(pc), C.c_void_p(oc.data_ptr()), None))
        torch.cuda.synchronize()
        ref, _ = oracle.render(sc.descriptor, pc, 32)
        res["max_err_160x90"] = float(np.abs(oc.cpu().numpy() - ref).max())
    print(json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="")
    ap.add_argument("--scenes", default="balls,sponza")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--child", default="")
    a = ap.parse_args()
    if a.child:
        run_one(a.child, a.steps, a.width, a.height, a.check)
        return
    libs = [l for l in a.libs.split(",") if l] or [""]
    for lib in libs:
        env = dict(os.environ)
        if lib:
            env["NRAYS_HIP_LIB"] = os.path.abspath(lib)
        print("== lib:", lib or "default", flush=True)
        for s in a.scenes.split(","):
            cmd = [sys.executable, os.path.abspath(__file__), "--child", s, "--steps", str(a.steps), "--width", str(a.width), "--height", str(a.height)]
            if a.check:
                cmd.append("--check")
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
            sys.stdout.write(r.stdout)
            if r.returncode != 0:
                print("FAILED:", r.stderr[-1500:], flush=True)


if __name__ == "__main__":
    main()
