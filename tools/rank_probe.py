"""One rank of the 8-way split of config 4 on one GPU: frame time against the two bounds of its schedule (longest wave tile; sum of tile cycles per resident wave).
  [NRAYS_OCC=.. NRAYS_LIGHT_SPLIT=..] python tools/rank_probe.py [world] [rank ...]"""
import ctypes as C, json, os, sys
os.environ.setdefault("NRAYS_EVENT_STRIDE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import nrays_amd as nr
from nrays_amd import abi, tiling
from tools import scenes_util as su, standins
lib = abi.load_hip_lib()
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ranks = [int(x) for x in sys.argv[2:]] or [0, 5]
sc, cam = standins.sponza_scene(n_lights=8)
W, H = [int(x) for x in os.environ.get("RES", "3840x2160").split("x")]
full, _ = su.camera_params(cam, W, H)
h = sc.device_handle()
for rank in ranks:
    p = tiling.tile_params(full, rank, world, tiling.DEFAULT_BAND_ROWS)
    rows = lib.nrays_tile_rows(C.byref(p))
    out = torch.empty((rows, full.width, 3), dtype=torch.float32, device="cuda")
    for _ in range(5): abi.check(lib.nrays_render_device(h, C.byref(p), C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize()
    tc = abi.NraysTileCosts(); ok = lib.nrays_get_tile_costs(h, C.byref(tc)) == 0
    nr.get_stats(sc)
    per = []
    for _ in range(12):
        abi.check(lib.nrays_render_device(h, C.byref(p), C.c_void_p(out.data_ptr()), None))
        st = nr.get_stats(sc); per.append(round(st.kernel_ms_total, 3))
    ms = sum(per) / len(per)
    r = {"world": world, "rank": rank, "rows": rows, "ms": round(ms, 4), "frames_ms": per}
    if ok:
        hz = 2.4e9
        r.update({"wave_tiles": tc.tiles, "resident_waves": tc.resident_waves, "longest_tile_ms": round(tc.max_cycles / hz * 1e3, 4),
                  "sum_cycles_per_resident_wave_ms": round(tc.sum_cycles / tc.resident_waves / hz * 1e3, 4)})
    if os.environ.get("RANK_TIMELINE"):  # a -DNR_DEBUG_TILE_COSTS build: when the waves of the last frame entered and left the kernel (10 ns ticks)
        import numpy as np
        lib.nrays_debug_wave_times.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        buf = np.zeros((16384, 4), np.uint32); n = C.c_uint32()
        abi.check(lib.nrays_debug_wave_times(h, buf.ctypes.data, 16384, C.byref(n)))
        w = buf[:n.value].astype(np.int64); w = w[w[:, 2] != 0]
        t0 = w[:, 0].min(); ex = (w[:, 2] - t0) / 100.0; en = (w[:, 0] - t0) / 100.0
        r.update({"waves": int(len(w)), "span_us": round(float(ex.max()), 1), "entry_us_p50_100": [round(float(x), 1) for x in np.percentile(en, [50, 100])],
                  "exit_us_p1_10_25_50_75_90_99_100": [round(float(x), 1) for x in np.percentile(ex, [1, 10, 25, 50, 75, 90, 99, 100])],
                  "idle_wave_time_before_the_end_frac": round(float((ex.max() - ex).sum() / (ex.max() * len(w))), 4)})
    print(json.dumps(r), flush=True)
