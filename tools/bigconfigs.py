"""One-off run of the large BASELINE configs on ONE GPU (configs 4 and 5 are 8-GPU configs; this gives
the single-GPU rate and checks that the sample-batching path survives 530 M primary rays)."""
import ctypes as C, sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import nrays_amd as nr
from nrays_amd import abi
from tools import scenes_util as su, standins
lib = abi.load_hip_lib()
def run(name, sc, cam, w, h, steps, **kw):
    p, _ = su.camera_params(cam, w, h, **kw)
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    hd = sc.device_handle()
    abi.check(lib.nrays_render_device(hd, C.byref(p), C.c_void_p(out.data_ptr()), None))
    st = nr.get_stats(sc)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): abi.check(lib.nrays_render_device(hd, C.byref(p), C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"config": name, "res": [w, h], "ms": round(dt * 1e3, 2), "rays": st.total_rays(), "primary": st.rays_primary,
                      "shadow": st.rays_shadow, "mrays_s": round(st.total_rays() / dt / 1e6, 1), "finite": bool(torch.isfinite(out).all()),
                      "mean": float(out.mean())}), flush=True)
sc, cam = standins.sponza_scene(n_lights=8)
run("config4 sponza 4K 8 lights (1 of 8 GPUs' worth: whole frame on one GPU)", sc, cam, 3840, 2160, 3)
del sc
hs, hc = standins.hairball_scene()
run("config5 hairball 4K 64 spp window 1.0 (whole frame on one GPU)", hs, hc, 3840, 2160, 1, spp=64, window=1.0, seed=1)
