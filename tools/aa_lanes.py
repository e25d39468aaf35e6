import ctypes as C, sys, os, time, json
sys.path.insert(0, os.getcwd())
import torch
import nrays_amd as nr
from nrays_amd import abi
from tools import scenes_util as su, standins
lib = abi.load_hip_lib()
hs, hc = standins.hairball_scene()
hd = hs.device_handle()
for (w, h, spp) in [(1920, 1080, 16), (3840, 2160, 64)]:
    p, _ = su.camera_params(hc, w, h, spp=spp, window=1.0, seed=1)
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        abi.check(lib.nrays_render_device(hd, C.byref(p), C.c_void_p(out.data_ptr()), None))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = nr.get_stats(hs)
    print(json.dumps({"lane_log2_cap": os.environ.get("NRAYS_LANE_LOG2", "none"), "res": [w, h], "spp": spp, "ms": round(dt * 1e3, 1), "rays": st.total_rays(), "mrays_s": round(st.total_rays() / dt / 1e6, 1), "sum": float(out.double().sum())}), flush=True)
