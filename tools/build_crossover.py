#!/usr/bin/env python
"""From how many triangles is the device BLAS builder faster than the host builder?  nrays_scene_create of a one-mesh scene (a random
triangle soup) with NRAYS_GPU_BUILD=0 and with NRAYS_GPU_BUILD_MIN=1, best of three, in a warm process (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import nrays_amd as nr


def scene(n, seed):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-1, 1, (n, 1, 3)); pts = (c + rng.uniform(-0.03, 0.03, (n, 3, 3))).reshape(-1, 3).astype(np.float32).astype(np.float64)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    mat = nr.PhongMaterial((0.1, 0.1, 0.1), (1, 1, 1), (1, 1, 1), None, None, 100.0)
    node = nr.SceneNode(mat, 0.0, 0.0, 1.0, 1.0, nr.Isometry3((0.0, 0.0, 0.0), (0.0, 0.0, 0.0)), nr.TriMesh(pts, idx, None))
    return nr.Scene([node], [nr.Light((0.0, 3.0, -5.0), 0.0, 1, (1, 1, 1))], (1, 1, 1))


def create_ms(n, env):
    for k in ("NRAYS_GPU_BUILD", "NRAYS_GPU_BUILD_MIN"): os.environ.pop(k, None)
    os.environ.update(env)
    best = 1e9
    for rep in range(3):
        sc = scene(n, 7)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sc.device_handle(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
        del sc
    return best


create_ms(100000, {"NRAYS_GPU_BUILD_MIN": "1"})  # warm the process
for n in [int(x) for x in (sys.argv[1:] or "1000 3000 10000 20000 50000 100000 300000 1000000".split())]:
    print("%8d triangles: host %.2f ms, device %.2f ms" % (n, create_ms(n, {"NRAYS_GPU_BUILD": "0"}), create_ms(n, {"NRAYS_GPU_BUILD_MIN": "1"})), flush=True)
