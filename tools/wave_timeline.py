#!/usr/bin/env python
"""When the waves of the primary kernel start, reach their first tile and exit (GPU box; needs a -DNR_DEBUG_TILE_COSTS build):
  NRAYS_HIP_LIB=nrays_amd/lib/v/tc.so python tools/wave_timeline.py balls [frames]
Prints the span of the launch (first entry -> last exit), percentiles of the waves' entry / first-tile / exit times relative to the
first entry, and the tiles of the waves that exit last."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from nrays_amd import abi
from tools import scenes_util as su, standins
lib = abi.load_hip_lib()
name = sys.argv[1] if len(sys.argv) > 1 else "balls"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc, cam = {"sponza": standins.sponza_scene, "hairball": standins.hairball_scene, "balls": su.balls_scene,
           "ballsfar": lambda: (su.balls_scene()[0], dict(su.balls_scene()[1], eye=(0.0, 150.0, -300.0))),
           "primitives": lambda: su.primitives_scene(0.0, 1), "config4": lambda: standins.sponza_scene(n_lights=8), "sponza8": lambda: standins.sponza_scene(n_lights=8),
           "config5": standins.hairball_scene}[name]()
W, H = (3840, 2160) if name in ("config4", "config5") else (1920, 1080)
if os.environ.get("NRAYS_TIMELINE_RES"): W, H = map(int, os.environ["NRAYS_TIMELINE_RES"].split("x"))
p, _ = su.camera_params(cam, W, H, **({"max_depth": int(sys.argv[3])} if len(sys.argv) > 3 else {}), **(dict(spp=64, window=1.0, seed=1) if name == "config5" else {}))  # optional third argument: max_depth
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
lib.nrays_debug_wave_times.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
for f in range(frames):
    abi.check(lib.nrays_render_device(sc.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
    buf = np.zeros((16384, 4), np.uint32); n = C.c_uint32()
    abi.check(lib.nrays_debug_wave_times(sc.device_handle(), buf.ctypes.data, 16384, C.byref(n)))
    w = buf[:n.value].astype(np.int64)
    w = w[w[:, 2] != 0]
    if os.environ.get("NRAYS_DEBUG_WAVE_WORK") == "4":  # w[:, 1] = 10 ns ticks inside fill_row (20 bits) | row entries << 20
        t0 = w[:, 0].min()
        ex = (w[:, 2] - t0) / 100.0
        rt, rn = (w[:, 1] & 0xfffff) / 100.0, w[:, 1] >> 20
        last = np.argsort(ex)[::-1][:6]
        print(json.dumps({"scene": name, "frame": f, "span_us": round(float(ex.max()), 1), "row_entries_per_wave_p50_90_100": [float(x) for x in np.percentile(rn, [50, 90, 100])],
                          "row_us_per_wave_p50_90_100": [round(float(x), 1) for x in np.percentile(rt, [50, 90, 100])],
                          "last_waves": [{"exit_us": round(float(ex[i]), 1), "row_entries": int(rn[i]), "row_us": round(float(rt[i]), 1), "entries": int(w[i, 3] & 0xfff)} for i in last]}), flush=True)
        continue
    if os.environ.get("NRAYS_DEBUG_WAVE_WORK") == "3":  # w[:, 1] = when the wave's last work tile ended (0: it had none)
        t0 = w[:, 0].min()
        ex = (w[:, 2] - t0) / 100.0
        had = w[:, 1] != 0
        le = np.where(had, (w[:, 1] - t0) / 100.0, np.nan)
        tail = ex - le
        last = np.argsort(ex)[::-1][:6]
        xcc = (w[:, 3] >> 28) & 7
        print(json.dumps({"scene": name, "frame": f, "last_work_tile_end_us_by_xcd_p10_p50_max": {int(x): [round(float(v), 0) for v in np.nanpercentile(le[xcc == x], [10, 50, 100])] for x in range(8)},
                          "entries_per_wave_p0_50_100": [int(v) for v in np.percentile(w[:, 3] & 0xfff, [0, 50, 100])]}), flush=True)
        print(json.dumps({"scene": name, "frame": f, "span_us": round(float(ex.max()), 1), "waves_with_work": int(had.sum()),
                          "last_work_tile_end_us_p50_90_99_100": [round(float(x), 1) for x in np.nanpercentile(le, [50, 90, 99, 100])],
                          "exit_after_last_work_tile_us_p0_50_90_100": [round(float(x), 1) for x in np.nanpercentile(tail, [0, 50, 90, 100])],
                          "last_waves": [{"exit_us": round(float(ex[i]), 1), "last_work_tile_end_us": round(float(le[i]), 1), "entries": int(w[i, 3] & 0xfff)} for i in last]}), flush=True)
        continue
    if os.environ.get("NRAYS_DEBUG_WAVE_WORK") == "2":  # w[:, 1] = the wave's whole life in s_memtime ticks / 16
        life_us = (w[:, 2] - w[:, 0]) / 100.0
        mhz = w[:, 1] * 16 / np.maximum(life_us, 1e-3)
        print(json.dumps({"scene": name, "frame": f, "s_memtime_ticks_per_us_p0_50_100": [round(float(x), 1) for x in np.percentile(mhz, [0, 50, 100])]}), flush=True)
        continue
    if os.environ.get("NRAYS_DEBUG_WAVE_WORK"):  # w[:, 1] = work-tile cycles / 16 (26 bits) | work tiles << 26
        t0 = w[:, 0].min()
        ex = (w[:, 2] - t0) / 100.0
        wc, wt = (w[:, 1] & 0x03ffffff) * 16 / 2400.0, w[:, 1] >> 26   # us at 2.4 GHz, tiles
        last = np.argsort(ex)[::-1][:6]
        print(json.dumps({"scene": name, "frame": f, "span_us": round(float(ex.max()), 1),
                          "work_tiles_per_wave_p50_90_99_100": [float(x) for x in np.percentile(wt, [50, 90, 99, 100])],
                          "work_us_per_wave_p50_90_99_100": [round(float(x), 1) for x in np.percentile(wc, [50, 90, 99, 100])],
                          "corr_exit_vs_work_us": round(float(np.corrcoef(ex, wc)[0, 1]), 3),
                          "exit_minus_work_us_p0_50_100": [round(float(x), 1) for x in np.percentile(ex - wc, [0, 50, 100])],
                          "last_waves": [{"exit_us": round(float(ex[i]), 1), "work_us": round(float(wc[i]), 1), "work_tiles": int(wt[i]), "entries": int(w[i, 3] & 0xfff)} for i in last]}), flush=True)
        continue
    hw = w[:, 3] >> 12; w[:, 3] &= 0xfff  # xcc << 16 | HW_ID[15:0] (gfx9: cu_id 11:8, sh_id 12, se_id 15:13)
    cu = (hw >> 8) & 0xfff  # xcc, se, sh, cu: one value per CU
    wg = np.arange(len(w)) // 4
    pairs = {}
    for c, g in zip(cu, wg): pairs.setdefault(int(c), set()).add(int(g))
    lead = sum(1 for v in pairs.values() if sum(1 for g in v if g < 256) == 1)
    simd = (hw >> 4) & 3
    simd_is_wave_index = float((simd == (np.arange(len(w)) % 4)).mean())
    distinct = [len(set(simd[i:i + 4].tolist())) for i in range(0, len(w) - 3, 4)]
    simd_hist = np.bincount(simd, minlength=4).tolist(); distinct_hist = np.bincount(distinct, minlength=5).tolist()
    t0 = w[:, 0].min()
    ent, first, ex = (w[:, 0] - t0) / 100.0, np.where(w[:, 3] > 0, (w[:, 1] - t0) / 100.0, np.nan), (w[:, 2] - t0) / 100.0
    q = lambda a: [round(float(x), 1) for x in np.nanpercentile(a, [0, 50, 90, 99, 100])]
    last = np.argsort(ex)[-3:][::-1]
    import nrays_amd as _nr
    _st = _nr.get_stats(sc)
    print(json.dumps({"scene": name, "frame": f, "waves": int(len(w)), "span_us": round(float(ex.max()), 1), "kernel_ms_primary_of_the_last_timed_frame": round(_st.kernel_ms_primary, 4),
                      "entry_us_p0_50_90_99_100": q(ent), "first_tile_us": q(first), "exit_us": q(ex),
                      "cus": len(pairs), "cus_with_exactly_one_of_the_first_256_workgroups": lead, "waves_whose_simd_id_equals_their_index_in_the_workgroup": round(simd_is_wave_index, 3), "simd_id_histogram": simd_hist, "workgroups_by_number_of_distinct_simd_ids": distinct_hist, "last_waves": [{"exit_us": round(float(ex[i]), 1), "entry_us": round(float(ent[i]), 1), "tiles": int(w[i, 3])} for i in last]}), flush=True)
