"""First frames of a fresh handle (warm process): host wall time against the GPU time of the frame's kernels (HIP events of the library).
  python tools/cold_probe.py sponza|hairball|balls [repeats]"""
import ctypes as C, json, os, sys, time
os.environ.setdefault("NRAYS_EVENT_STRIDE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import nrays_amd as nr
from nrays_amd import abi
from tools import scenes_util as su, standins
lib = abi.load_hip_lib()
name = sys.argv[1] if len(sys.argv) > 1 else "sponza"
make = {"sponza": standins.sponza_scene, "hairball": standins.hairball_scene, "balls": su.balls_scene, "sponza8": lambda: standins.sponza_scene(n_lights=8)}[name]
out = torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda")
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    sc, cam = make()
    p, _ = su.camera_params(cam, 1920, 1080)
    h = sc.device_handle(); torch.cuda.synchronize()
    rows = []
    for f in range(5):
        t0 = time.perf_counter()
        abi.check(lib.nrays_render_device(h, C.byref(p), C.c_void_p(out.data_ptr()), None))
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        st = nr.get_stats(sc)
        rows.append({"frame": f, "call_ms": round((t1 - t0) * 1e3, 4), "wall_ms": round((t2 - t0) * 1e3, 4), "gpu_ms": round(st.kernel_ms_total, 4), "primary_ms": round(st.kernel_ms_primary, 4)})
    print(json.dumps({"scene": name, "rep": rep, "frames": rows}), flush=True)
    del sc
