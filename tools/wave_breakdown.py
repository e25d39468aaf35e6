#!/usr/bin/env python
"""Where the waves of an analytic frame spend their life (GPU box; -DNR_DEBUG_TILE_COSTS build):
  NRAYS_HIP_LIB=nrays_amd/lib/v/tc.so python tools/wave_breakdown.py balls [frames] [max_depth]
Per wave: time in work tiles, in tiles that traced nothing, in background rows, and the rest (entry, dequeue, exit); printed as
percentiles over the waves, for lead / other workgroups, and for the waves that exit last."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from nrays_amd import abi
from tools import scenes_util as su
lib = abi.load_hip_lib()
name = sys.argv[1] if len(sys.argv) > 1 else "balls"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc, cam = {"balls": su.balls_scene, "primitives": lambda: su.primitives_scene(0.0, 1)}[name]()
W, H = 1920, 1080
p, _ = su.camera_params(cam, W, H, **({"max_depth": int(sys.argv[3])} if len(sys.argv) > 3 else {}))
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
for fn in ("nrays_debug_wave_times", "nrays_debug_wave_times2"):
    getattr(lib, fn).argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
q = lambda a: [round(float(x), 1) for x in np.percentile(a, [0, 50, 90, 99, 100])]
for f in range(frames):
    abi.check(lib.nrays_render_device(sc.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
    a = np.zeros((16384, 4), np.uint32); b = np.zeros((16384, 4), np.uint32); n = C.c_uint32()
    abi.check(lib.nrays_debug_wave_times(sc.device_handle(), a.ctypes.data, 16384, C.byref(n)))
    abi.check(lib.nrays_debug_wave_times2(sc.device_handle(), b.ctypes.data, 16384, C.byref(n)))
    a = a[:n.value].astype(np.int64); b = b[:n.value].astype(np.int64)
    ok = a[:, 2] != 0; idx = np.nonzero(ok)[0]; a = a[ok]; b = b[ok]
    t0 = a[:, 0].min()
    ent, ex = (a[:, 0] - t0) / 100.0, (a[:, 2] - t0) / 100.0
    work, miss, rows = b[:, 0] / 100.0, b[:, 1] / 100.0, b[:, 2] / 100.0
    nwork, nmiss, nrows, longest = b[:, 3] & 0xff, (b[:, 3] >> 8) & 0xff, (b[:, 3] >> 16) & 0xff, (b[:, 3] >> 24) * 16 / 100.0
    other = ex - ent - work - miss - rows
    lead = (idx // 4) < 256
    last = np.argsort(ex)[::-1][:8]
    print(json.dumps({"frame": f, "span_us": round(float(ex.max()), 1), "exit_us": q(ex), "work_us": q(work), "miss_us": q(miss), "rows_us": q(rows), "other_us": q(other),
                      "lead_wg_waves": {"exit": q(ex[lead]), "work": q(work[lead]), "miss": q(miss[lead]), "rows": q(rows[lead]), "n_work": q(nwork[lead]), "n_miss": q(nmiss[lead]), "n_rows": q(nrows[lead])},
                      "other_wg_waves": {"exit": q(ex[~lead]), "work": q(work[~lead]), "miss": q(miss[~lead]), "rows": q(rows[~lead]), "n_work": q(nwork[~lead]), "n_miss": q(nmiss[~lead]), "n_rows": q(nrows[~lead])},
                      "sum_us": {"work": round(float(work.sum()), 0), "miss": round(float(miss.sum()), 0), "rows": round(float(rows.sum()), 0), "other": round(float(other.sum()), 0), "wave_life": round(float((ex - ent).sum()), 0)},
                      "longest_work_tile_us": q(longest),
                      "last_waves": [{"wave": int(idx[i]), "exit": round(float(ex[i]), 1), "work": round(float(work[i]), 1), "longest": round(float(longest[i]), 1), "miss": round(float(miss[i]), 1), "rows": round(float(rows[i]), 1),
                                      "n": [int(nwork[i]), int(nmiss[i]), int(nrows[i])]} for i in last]}), flush=True)
