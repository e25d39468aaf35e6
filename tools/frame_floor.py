"""Where a cheap analytic frame spends its time (GPU box): empty scene, all-miss scene, balls with and without
reflection / depth cap, then empty and balls frames over a range of resolutions (fixed launch cost vs per-pixel cost)."""
import ctypes as C, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import nrays_amd as nr
from nrays_amd import abi
from tools import scenes_util as su
lib = abi.load_hip_lib()
def run(sc, cam, w=1920, h=1080, steps=50, **kw):
    p, _ = su.camera_params(cam, w, h, **kw)
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    hd = sc.device_handle()
    for _ in range(5): abi.check(lib.nrays_render_device(hd, C.byref(p), C.c_void_p(out.data_ptr()), None))
    nr.get_stats(sc)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): abi.check(lib.nrays_render_device(hd, C.byref(p), C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    st = nr.get_stats(sc)
    return dt * 1e3, st.kernel_ms_primary, st.total_rays()
cam = dict(eye=(0.0, 5.0, -10.0), at=(0.0, 0.0, 0.0), fovy=45.0)
empty = nr.Scene([], [nr.Light((0, 10, 0), 0, 1, (1, 1, 1))])
print("empty scene          ms %.4f kernel %.4f rays %d" % run(empty, cam))
far = nr.Scene([nr.SceneNode(su.default_material(), 0, 0, 1, 1, nr.Isometry3((0, 0, -100.0)), nr.Ball(1.0))], [nr.Light((0, 10, 0), 0, 1, (1, 1, 1))])
print("one ball behind cam  ms %.4f kernel %.4f rays %d" % run(far, cam))
sc0, _ = su.balls_scene(refl=(0.0, 0.0))
print("balls no reflection  ms %.4f kernel %.4f rays %d" % run(sc0, cam))
sc1, _ = su.balls_scene(refl=(0.2, 0.25))
print("balls 4 bounces      ms %.4f kernel %.4f rays %d" % run(sc1, cam))
print("balls 4b max_depth 1 ms %.4f kernel %.4f rays %d" % run(sc1, cam, max_depth=1))
for (w, h) in [(64, 64), (480, 270), (960, 540), (1920, 1080), (3840, 2160)]:
    print("empty %4dx%4d      ms %.4f kernel %.4f rays %d" % ((w, h) + run(empty, cam, w, h)))
for (w, h) in [(64, 64), (480, 270), (960, 540), (1920, 1080), (3840, 2160)]:
    print("balls %4dx%4d      ms %.4f kernel %.4f rays %d" % ((w, h) + run(sc1, cam, w, h)))
