import ctypes as C, sys, os, time, json
sys.path.insert(0, os.getcwd())
import torch
import nrays_amd as nr
from nrays_amd import abi
from tools import scenes_util as su
lib = abi.load_hip_lib()
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
for depth in (1, 2, 4):
    sc, cam = su.balls_scene()
    p, _ = su.camera_params(cam, W, H, max_depth=depth)
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    for _ in range(30): abi.check(lib.nrays_render_device(sc.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 400
    for _ in range(n): abi.check(lib.nrays_render_device(sc.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
    t_submit = (time.perf_counter() - t) / n
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    st = nr.get_stats(sc)
    print(json.dumps({"res": [W, H], "max_depth": depth, "wall_ms": round(dt * 1e3, 4), "host_submit_ms": round(t_submit * 1e3, 4), "kernel_ms_primary": round(st.kernel_ms_primary, 4), "kernel_ms_total": round(st.kernel_ms_total, 4)}), flush=True)
