"""Frame time against the two bounds of its schedule (longest wave tile; sum of tile cycles per resident wave) under the occupancy / grid switches.
  [NRAYS_OCC=3] [NRAYS_GRID_WG_PER_CU=1] python tools/occ_probe.py sponza|sponza8|hairball [WxH]"""
import ctypes as C, json, os, sys
os.environ.setdefault("NRAYS_EVENT_STRIDE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import nrays_amd as nr
from nrays_amd import abi
from tools import scenes_util as su, standins
lib = abi.load_hip_lib()
name = sys.argv[1] if len(sys.argv) > 1 else "sponza"
W, H = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1920x1080").split("x")]
sc, cam = {"sponza": standins.sponza_scene, "sponza8": lambda: standins.sponza_scene(n_lights=8), "hairball": standins.hairball_scene}[name]()
p, _ = su.camera_params(cam, W, H)
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
h = sc.device_handle()
for _ in range(5): abi.check(lib.nrays_render_device(h, C.byref(p), C.c_void_p(out.data_ptr()), None))
torch.cuda.synchronize()
tc = abi.NraysTileCosts(); ok = lib.nrays_get_tile_costs(h, C.byref(tc)) == 0
nr.get_stats(sc)
for _ in range(20): abi.check(lib.nrays_render_device(h, C.byref(p), C.c_void_p(out.data_ptr()), None))
st = nr.get_stats(sc)
r = {"scene": name, "res": [W, H], "ms": round(st.kernel_ms_total, 4), "env": {k: v for k, v in os.environ.items() if k.startswith("NRAYS_") and k != "NRAYS_EVENT_STRIDE"}}
if ok:
    hz = 2.4e9
    r.update({"wave_tiles": tc.tiles, "resident_waves": tc.resident_waves, "longest_tile_ms": round(tc.max_cycles / hz * 1e3, 4),
              "sum_cycles_per_resident_wave_ms": round(tc.sum_cycles / tc.resident_waves / hz * 1e3, 4), "sum_cycles_wave_ms": round(tc.sum_cycles / hz * 1e3, 1)})
if os.environ.get("RANK_TIMELINE"):  # a -DNR_DEBUG_TILE_COSTS build: when the waves of the last frame entered and left the kernel (10 ns ticks)
    import numpy as np
    lib.nrays_debug_wave_times.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    buf = np.zeros((16384, 4), np.uint32); n = C.c_uint32()
    abi.check(lib.nrays_debug_wave_times(h, buf.ctypes.data, 16384, C.byref(n)))
    w = buf[:n.value].astype(np.int64); w = w[w[:, 2] != 0]
    t0 = w[:, 0].min(); ex = (w[:, 2] - t0) / 100.0
    r.update({"waves": int(len(w)), "span_us": round(float(ex.max()), 1),
              "exit_us_p1_10_25_50_75_90_99_100": [round(float(x), 1) for x in np.percentile(ex, [1, 10, 25, 50, 75, 90, 99, 100])],
              "idle_wave_time_before_the_end_frac": round(float((ex.max() - ex).sum() / (ex.max() * len(w))), 4),
              "tiles_per_wave_p0_50_100": [int(x) for x in np.percentile(w[:, 3] & 0xfff, [0, 50, 100])]})
print(json.dumps(r), flush=True)
