"""balls: k_primary time vs resolution (looks for a step where the flagged tiles exceed the resident waves)."""
import ctypes as C, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import nrays_amd as nr
from nrays_amd import abi
from tools import scenes_util as su
lib = abi.load_hip_lib()
def run(sc, cam, w, h, steps=50):
    p, _ = su.camera_params(cam, w, h)
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    hd = sc.device_handle()
    for _ in range(5): abi.check(lib.nrays_render_device(hd, C.byref(p), C.c_void_p(out.data_ptr()), None))
    nr.get_stats(sc)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): abi.check(lib.nrays_render_device(hd, C.byref(p), C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    st = nr.get_stats(sc)
    hit = (out.reshape(-1, 3) != 1.0).any(dim=1).sum().item()
    return dt * 1e3, st.kernel_ms_primary, st.kernel_ms_total, st.total_rays(), hit
sc1, cam = su.balls_scene(refl=(0.2, 0.25))
for s in [0.5, 0.6, 0.7, 0.8, 0.85, 0.9, 0.95, 1.0, 1.1, 1.25, 1.5, 2.0]:
    w, h = int(1920 * s) // 16 * 16, int(1080 * s) // 8 * 8
    r = run(sc1, cam, w, h)
    print("balls %4dx%4d  ms %.4f primary %.4f total %.4f rays %d hit_px %d (%.0f wave tiles)" % ((w, h) + r + (r[4] / 64.0,)))
