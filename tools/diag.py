import ctypes as C, sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import nrays_amd as nr
from nrays_amd import abi
from tests import scenes_util as su, standins
lib = abi.load_hip_lib()
def run(sc, cam, w, h, steps=10, **kw):
    p, _ = su.camera_params(cam, w, h, **kw)
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    hd = sc.device_handle()
    for _ in range(3): abi.check(lib.nrays_render_device(hd, C.byref(p), C.c_void_p(out.data_ptr()), None))
    st = nr.get_stats(sc)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): abi.check(lib.nrays_render_device(hd, C.byref(p), C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    return dt * 1e3, st.total_rays(), st.generations
sc, cam = standins.sponza_scene()
for md in (0, 1, 2, 4):
    ms, rays, g = run(sc, cam, 1920, 1080, max_depth=md)
    print("sponza 1080p max_depth", md, "ms %.3f rays %d gens %d Mrays/s %.0f" % (ms, rays, g, rays / ms / 1e3))
for (w, h) in ((960, 540), (3840, 2160)):
    ms, rays, g = run(sc, cam, w, h, steps=5)
    print("sponza %dx%d ms %.3f rays %d Mrays/s %.0f" % (w, h, ms, rays, rays / ms / 1e3))
sc8, cam8 = standins.sponza_scene(n_lights=8)
ms, rays, g = run(sc8, cam8, 3840, 2160, steps=3)
print("sponza 4K 8 lights ms %.3f rays %d Mrays/s %.0f" % (ms, rays, rays / ms / 1e3))
