#!/usr/bin/env python
"""Load balance of the multi-GPU tiling measured on ONE GPU: for N = 1, 2, 4, 8 the tile of every rank is
rendered in turn (same kernel, same band parameters a real rank would get) and the slowest rank's GPU time
bounds the N-GPU frame from below (the RCCL gather, overlapped with the next render by
nrays_amd.tiling.FramePipeline, comes on top).  Prints one JSON line per workload.

  python tools/tile_scaling.py [balls|sponza8_4k|all]
"""
import ctypes as C, json, os, sys
os.environ.setdefault("NRAYS_EVENT_STRIDE", "1")  # HIP events on every frame of a handle (read when the handle is created)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import nrays_amd as nr
from nrays_amd import abi, tiling
from tools import scenes_util as su, standins

lib = abi.load_hip_lib()


def rank_ms(sc, full, rank, world, steps):
    p = tiling.tile_params(full, rank, world, int(os.environ.get("BAND_ROWS", tiling.DEFAULT_BAND_ROWS)))
    rows = lib.nrays_tile_rows(C.byref(p))
    out = torch.empty((rows, full.width, 3), dtype=torch.float32, device="cuda")
    h = sc.device_handle()
    for _ in range(int(os.environ.get("SETTLE", "5"))):  # the per-camera scheduling state of a handle settles in three plain frames
        abi.check(lib.nrays_render_device(h, C.byref(p), C.c_void_p(out.data_ptr()), None))
    nr.get_stats(sc)
    for _ in range(steps):
        abi.check(lib.nrays_render_device(h, C.byref(p), C.c_void_p(out.data_ptr()), None))
    return nr.get_stats(sc).kernel_ms_total


def run(name, sc, cam, w, h, steps):
    full, _ = su.camera_params(cam, w, h)
    res = {"workload": name, "res": [w, h], "band_rows": int(os.environ.get("BAND_ROWS", tiling.DEFAULT_BAND_ROWS))}
    t1 = None
    for world in [int(x) for x in os.environ.get("WORLDS", "1,2,4,8").split(",")]:
        ts = [rank_ms(sc, full, r, world, steps) for r in range(world)]
        if world == 1:
            t1 = ts[0]
        if t1 is None:
            t1 = float(os.environ.get("T1_MS", "0")) or sum(ts)
        res["N=%d" % world] = {"slowest_rank_ms": round(max(ts), 4), "fastest_rank_ms": round(min(ts), 4), "mean_rank_ms": round(sum(ts) / len(ts), 4),
                               "render_speedup_bound": round(t1 / max(ts), 2), "rank_ms": [round(t, 3) for t in ts]}
    print(json.dumps(res), flush=True)


which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("balls", "all"):
    sc, cam = su.balls_scene()
    run("balls 1080p 4 bounces", sc, cam, 1920, 1080, 50)
if which in ("sponza", "all"):
    sc, cam = standins.sponza_scene()
    run("sponza stand-in 1080p", sc, cam, 1920, 1080, 10)
if which in ("sponza8_4k", "all"):
    sc, cam = standins.sponza_scene(n_lights=8)
    run("sponza stand-in 4K, 8 lights (config 4)", sc, cam, 3840, 2160, 3)
