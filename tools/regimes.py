#!/usr/bin/env python
"""The three regimes a caller of nrays_render_device can be in, side by side (reference caller: examples/loader3d.rs:67-93 renders
each camera ONCE):
  steady   a resting camera re-rendered (pipelined: ms per frame of N launches and one synchronisation; sync: host-synchronised wall time per frame)
  cold     the FIRST frame of a fresh handle in a warm process: host call time, host-synchronised wall time, HIP-event GPU time
  moving   a camera that moves every frame (eye shifted by 1e-3 of its distance to `at` per frame, bench.py's path), pipelined
and the frames are the same frames: the last moving frame and the cold frame are compared bit for bit with a settled handle's render of the same camera.
  python tools/regimes.py balls|sponza|hairball|sponza8|primitives [width height] [--cold N] [--frames N]"""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import nrays_amd as nr
from nrays_amd import abi
from tools import scenes_util as su, standins

ap = argparse.ArgumentParser()
ap.add_argument("scene"); ap.add_argument("width", nargs="?", type=int, default=1920); ap.add_argument("height", nargs="?", type=int, default=1080)
ap.add_argument("--cold", type=int, default=5); ap.add_argument("--frames", type=int, default=40); ap.add_argument("--steady", type=int, default=200)
a = ap.parse_args()
lib = abi.load_hip_lib()
make = {"sponza": standins.sponza_scene, "hairball": standins.hairball_scene, "balls": su.balls_scene, "sponza8": lambda: standins.sponza_scene(n_lights=8),
        "primitives": lambda: su.primitives_scene(0.0, 1)}[a.scene]
W, H = a.width, a.height
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream


def render(h, p, o=out):
    abi.check(lib.nrays_render_device(h, C.byref(p), C.c_void_p(o.data_ptr()), C.c_void_p(stream)))


sc, cam = make()
p, _ = su.camera_params(cam, W, H)
h = sc.device_handle()
for _ in range(6):  # settle (and warm the process: code objects loaded)
    render(h, p)
torch.cuda.synchronize()
ref = out.clone()
res = {"scene": a.scene, "res": [W, H]}
# steady, pipelined
nr.get_stats(sc)
t0 = time.perf_counter()
for _ in range(a.steady):
    render(h, p)
torch.cuda.synchronize()
res["steady_ms"] = round((time.perf_counter() - t0) / a.steady * 1e3, 5)
res["steady_gpu_ms"] = round(nr.get_stats(sc).kernel_ms_total, 5)
# steady, host-synchronised frame by frame
ws = []
for _ in range(20):
    t0 = time.perf_counter(); render(h, p); torch.cuda.synchronize(); ws.append((time.perf_counter() - t0) * 1e3)
res["steady_sync_wall_ms"] = round(float(np.median(ws)), 5)
# cold: first frame of fresh handles
cold = []
for k in range(a.cold):
    sc2, _ = make()
    h2 = sc2.device_handle(); torch.cuda.synchronize()
    o2 = torch.empty_like(out)
    t0 = time.perf_counter(); render(h2, p, o2); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    st = nr.get_stats(sc2)
    t3 = time.perf_counter(); render(h2, p, o2); torch.cuda.synchronize(); t4 = time.perf_counter()
    st2 = nr.get_stats(sc2)
    cold.append({"call_ms": round((t1 - t0) * 1e3, 4), "wall_ms": round((t2 - t0) * 1e3, 4), "gpu_ms": round(st.kernel_ms_total, 4), "primary_ms": round(st.kernel_ms_primary, 4),
                 "second_wall_ms": round((t4 - t3) * 1e3, 4), "second_gpu_ms": round(st2.kernel_ms_total, 4), "identical": bool(torch.equal(o2, ref))})
    del sc2
res["cold"] = cold
res["cold_wall_ms"] = round(float(np.median([c["wall_ms"] for c in cold])), 4)
res["cold_gpu_ms"] = round(float(np.median([c["gpu_ms"] for c in cold])), 4)
res["cold_identical"] = all(c["identical"] for c in cold)
# moving camera (bench.py's path), pipelined; twice: the second pass starts from a handle that has seen the neighbourhood
eye0 = np.array(cam["eye"], dtype=np.float64); at = np.array(cam["at"], dtype=np.float64)
step = 1e-3 * np.linalg.norm(eye0 - at) * np.array([1.0, 0.0, 0.0])
params = [su.camera_params(dict(cam, eye=tuple(eye0 + (k + 1) * step)), W, H)[0] for k in range(a.frames)]
mv = []
for rep in range(3):
    torch.cuda.synchronize(); nr.get_stats(sc)
    t0 = time.perf_counter()
    for q in params:
        render(h, q)
    torch.cuda.synchronize()
    mv.append(round((time.perf_counter() - t0) / a.frames * 1e3, 5))
res["moving_ms"] = mv
res["moving_gpu_ms"] = round(nr.get_stats(sc).kernel_ms_total, 5)
last = out.clone()
# the same cameras at rest (a settled handle each): what the moving frames would cost if nothing had to be learnt about them
along = []
for k in sorted(set([0, a.frames // 4, a.frames // 2, 3 * a.frames // 4, a.frames - 1])):
    sc3, _ = make(); h3 = sc3.device_handle()
    o3 = torch.empty_like(out)
    for _ in range(6):
        render(h3, params[k], o3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        render(h3, params[k], o3)
    torch.cuda.synchronize()
    along.append(round((time.perf_counter() - t0) / 40 * 1e3, 5))
    if k == a.frames - 1:
        res["moving_last_frame_identical_to_a_settled_render"] = bool(torch.equal(last, o3))
    del sc3
res["steady_along_the_path_ms"] = along
res["steady_along_the_path_mean_ms"] = round(float(np.mean(along)), 5)
# a camera that jumps far away and back (not "nearby": must be treated as cold, pixels identical)
far = su.camera_params(dict(cam, eye=tuple(eye0 * 1.7 + np.array([3.0, 1.0, 0.5]))), W, H)[0]
render(h, far); render(h, p); torch.cuda.synchronize()
res["after_a_jump_identical"] = bool(torch.equal(out, ref))
print(json.dumps(res), flush=True)
