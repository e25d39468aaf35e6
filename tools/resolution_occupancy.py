import ctypes as C, sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import nrays_amd as nr
from nrays_amd import abi
from tools import scenes_util as su, standins
lib = abi.load_hip_lib()
def run(name, sc, cam, w, h, steps):
    p, _ = su.camera_params(cam, w, h)
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    hd = sc.device_handle()
    for _ in range(4): abi.check(lib.nrays_render_device(hd, C.byref(p), C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): abi.check(lib.nrays_render_device(hd, C.byref(p), C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(name, w, h, round(dt * 1e3, 3), flush=True)
sc, cam = standins.sponza_scene(n_lights=8)
for (w, h) in [(1920, 1080), (2560, 1440), (3840, 2160)]: run("sponza8", sc, cam, w, h, 5)
del sc
sc, cam = standins.sponza_scene()
for (w, h) in [(1920, 1080), (2560, 1440), (3840, 2160)]: run("sponza1", sc, cam, w, h, 8)
