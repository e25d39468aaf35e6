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
out = torch.empty((H + int(os.environ.get("NRAYS_BREAKDOWN_PAD", "0")), W, 3), dtype=torch.float32, device="cuda")  # (padding: is the end of the allocation special?)
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
    if os.environ.get("NRAYS_DEBUG_WAVE_WORK") == "5":  # b[:, 1] = the wave's slowest row entry: row << 14 | quarter << 12 | ticks
        sr, sq, st = b[:, 1] >> 14, (b[:, 1] >> 12) & 3, (b[:, 1] & 0xfff) / 100.0
        o = np.argsort(st)[::-1][:24]
        print(json.dumps({"frame": f, "slowest_row_entries": [{"wave": int(idx[i]), "wg": int(idx[i]) // (len(a) // 256 // 1 if False else 4), "row": int(sr[i]), "quarter": int(sq[i]), "us": round(float(st[i]), 1), "row_mod_8": int(sr[i] % 8)} for i in o],
                          "row_us_by_row_mod_8_p50_p99": {int(m): [round(float(v), 1) for v in np.percentile(st[(sr % 8 == m) & (st > 0)], [50, 99])] for m in range(8) if ((sr % 8 == m) & (st > 0)).any()}}), flush=True)
        continue
    lead = (idx // 4) < 256
    xcc = (a[:, 3] >> 28) & 7
    by_xcd = {int(x): {"waves": int((xcc == x).sum()), "exit_p50_max": [round(float(v), 1) for v in np.percentile(ex[xcc == x], [50, 100])], "rows_p50_p99_max": [round(float(v), 1) for v in np.percentile(rows[xcc == x], [50, 99, 100])],
                       "other_p50_max": [round(float(v), 1) for v in np.percentile(other[xcc == x], [50, 100])], "work_sum": round(float(work[xcc == x].sum()), 0), "rows_n": int(nrows[xcc == x].sum())} for x in range(8) if (xcc == x).any()}
    last = np.argsort(ex)[::-1][:8]
    print(json.dumps({"frame": f, "span_us": round(float(ex.max()), 1), "exit_us": q(ex), "work_us": q(work), "miss_us": q(miss), "rows_us": q(rows), "other_us": q(other),
                      "lead_wg_waves": {"exit": q(ex[lead]), "work": q(work[lead]), "miss": q(miss[lead]), "rows": q(rows[lead]), "n_work": q(nwork[lead]), "n_miss": q(nmiss[lead]), "n_rows": q(nrows[lead])},
                      "other_wg_waves": {"exit": q(ex[~lead]), "work": q(work[~lead]), "miss": q(miss[~lead]), "rows": q(rows[~lead]), "n_work": q(nwork[~lead]), "n_miss": q(nmiss[~lead]), "n_rows": q(nrows[~lead])},
                      "sum_us": {"work": round(float(work.sum()), 0), "miss": round(float(miss.sum()), 0), "rows": round(float(rows.sum()), 0), "other": round(float(other.sum()), 0), "wave_life": round(float((ex - ent).sum()), 0)},
                      "longest_work_tile_us": q(longest), "by_xcd": by_xcd,
                      "last_waves": [{"wave": int(idx[i]), "exit": round(float(ex[i]), 1), "work": round(float(work[i]), 1), "longest": round(float(longest[i]), 1), "miss": round(float(miss[i]), 1), "rows": round(float(rows[i]), 1),
                                      "n": [int(nwork[i]), int(nmiss[i]), int(nrows[i])]} for i in last]}), flush=True)
