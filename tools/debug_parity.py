import ctypes as C, sys, os, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import nrays_amd as nr
from nrays_amd import abi
from tests import scenes_util as su, standins
import oracle
lib = abi.load_hip_lib()
sc, cam = standins.sponza_scene()
W, H = 160, 90
def gpu(p):
    rows = lib.nrays_tile_rows(C.byref(p))
    out = torch.empty((rows, p.width, 3), dtype=torch.float32, device="cuda")
    abi.check(lib.nrays_render_device(sc.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
    st = nr.get_stats(sc)
    return out.cpu().numpy(), st
for md in (1, 2, 0):
    p, _ = su.camera_params(cam, W, H, max_depth=md)
    g, gs = gpu(p)
    o, os_ = oracle.render(sc.descriptor, p, 32)
    err = np.abs(g - o).max(axis=2)
    bad = np.argwhere(err > 1e-4)
    print("max_depth", md, "max err", err.max(), "bad pixels", len(bad), "gpu rays", gs.rays_primary, gs.rays_refraction, gs.rays_shadow, "oracle", os_.rays_primary, os_.rays_refraction, os_.rays_shadow)
    for (j, i) in bad[:12]:
        print("  px", i, j, "gpu", g[j, i], "oracle", o[j, i])
    np.save(os.path.join(ROOT, "gpurun_out", "dbg_gpu_%d.npy" % md), g)
    np.save(os.path.join(ROOT, "gpurun_out", "dbg_ora_%d.npy" % md), o)
