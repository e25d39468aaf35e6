import ctypes as C, json, os, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("NRAYS_EVENT_STRIDE", "1")
import torch
import nrays_amd as nr
from nrays_amd import abi, tiling
from tools import scenes_util as su, standins
lib = abi.load_hip_lib()
sc, cam = standins.sponza_scene(n_lights=8)
full, _ = su.camera_params(cam, 3840, 2160)
for world in (1, 4, 8):
    for rank in ((0,) if world == 1 else (0, 2, world - 1)):
        p = tiling.tile_params(full, rank, world, tiling.DEFAULT_BAND_ROWS)
        rows = lib.nrays_tile_rows(C.byref(p))
        out = torch.empty((rows, full.width, 3), dtype=torch.float32, device="cuda")
        h = sc.device_handle()
        abi.check(lib.nrays_render_device_instrumented(h, C.byref(p), C.c_void_p(out.data_ptr()), None))  # (measures the shader clock under this load)
        for _ in range(6): abi.check(lib.nrays_render_device(h, C.byref(p), C.c_void_p(out.data_ptr()), None))
        nr.get_stats(sc)
        for _ in range(4): abi.check(lib.nrays_render_device(h, C.byref(p), C.c_void_p(out.data_ptr()), None))
        ms = nr.get_stats(sc).kernel_ms_total
        tc = abi.NraysTileCosts(); abi.check(lib.nrays_get_tile_costs(h, C.byref(tc)))
        clk = (tc.shader_clock_hz or 2.4e9) / 1e3  # cycles per ms, measured by the recording launch
        print(json.dumps({"world": world, "rank": rank, "kernel_ms": round(ms, 3), "recording_launch_ms": round(tc.kernel_ms, 3), "clock_ghz": round(clk / 1e6, 3), "tiles": tc.tiles, "resident_waves": tc.resident_waves, "sum_cycles_per_wave_ms": round(tc.sum_cycles / tc.resident_waves / clk, 3),
                          "max_tile_ms": round(tc.max_cycles / clk, 3), "tiles_per_wave": round(tc.tiles / tc.resident_waves, 2)}), flush=True)
