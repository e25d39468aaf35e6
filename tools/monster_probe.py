#!/usr/bin/env python
"""What the longest wave tile of a mesh frame is made of (GPU box; -DNR_PHASE_TIMING build: tools/build_variant.sh pt -DNR_PHASE_TIMING).
Finds the most expensive 8x8 wave tile of the full frame (recorded tile costs), then renders exactly that tile as a frame of its own — an off-centre sub-frustum of the same
camera: the same rays up to rounding — so that every phase counter of the launch belongs to that one wave: cycles in the closest-hit queries of the primary / continuation
rays, in shadow queries, outside the queries; node-loop and triangle-loop iterations and their SIMD efficiency; generations.
  NRAYS_HIP_LIB=nrays_amd/lib/v/pt.so python tools/monster_probe.py sponza [rank ...]      (rank 0 = the longest tile, 1 = the second, ...)"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("NRAYS_LIGHT_SPLIT", "0")   # whole tiles: the question is what an UNSPLIT deep tile costs
os.environ.setdefault("NRAYS_EVENT_STRIDE", "1")
import numpy as np, torch
import nrays_amd as nr
from nrays_amd import abi
from tools import scenes_util as su, standins
lib = abi.load_hip_lib()
lib.nrays_debug_tile_costs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
lib.nrays_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
name = sys.argv[1] if len(sys.argv) > 1 else "sponza"
ranks = [int(x) for x in sys.argv[2:]] or [0, 1, 5, 40, 300]
make = {"sponza": standins.sponza_scene, "sponza8": lambda: standins.sponza_scene(n_lights=8), "hairball": standins.hairball_scene}[name]
W, H = 1920, 1080
sc, cam = make()
p, _ = su.camera_params(cam, W, H)
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
for _ in range(2):
    abi.check(lib.nrays_render_device(sc.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
torch.cuda.synchronize()
tc = abi.NraysTileCosts(); abi.check(lib.nrays_get_tile_costs(sc.device_handle(), C.byref(tc)))
buf = np.zeros(1 << 20, np.uint32); n = C.c_uint32()
abi.check(lib.nrays_debug_tile_costs(sc.device_handle(), buf.ctypes.data, 1 << 20, C.byref(n)))
cost = (buf[:int(tc.tiles)] & 0x7fffffff).astype(np.int64) * 16
order = np.argsort(-cost)
nx = (W + 15) // 16
M = np.array(list(p.inv_proj_view), dtype=np.float64).reshape(4, 4).T  # column-major -> M[:, c] is column c
for r in ranks:
    wt = int(order[r]); tile, sub = wt >> 2, wt & 3
    i0 = (tile % nx) * 16 + (sub & 1) * 8; j0 = (tile // nx) * 16 + (sub >> 1) * 8
    w = h = 8
    sx, sy = w / W, h / H
    tx, ty = (w + 2 * i0) / W - 1.0, 1.0 - (h + 2 * j0) / H
    Mc = M.copy(); Mc[:, 0] = sx * M[:, 0]; Mc[:, 1] = sy * M[:, 1]; Mc[:, 3] = M[:, 3] + tx * M[:, 0] + ty * M[:, 1]
    q = abi.NraysRenderParams.from_buffer_copy(p)
    q.width, q.height = w, h
    for c in range(4):
        for rr in range(4):
            q.inv_proj_view[4 * c + rr] = Mc[rr, c]
    s2, _ = make()
    o2 = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    for _ in range(2):
        abi.check(lib.nrays_render_device(s2.device_handle(), C.byref(q), C.c_void_p(o2.data_ptr()), None))
    st = nr.get_stats(s2)
    dbg = (C.c_ulonglong * 16)(); abi.check(lib.nrays_debug_counters(s2.device_handle(), dbg))
    full = out[j0:j0 + 8, i0:i0 + 8].cpu().numpy(); crop = o2.cpu().numpy()
    tot = max(st.prim_tests, 1)
    print(json.dumps({"scene": name, "rank": r, "tile_pixels": [i0, j0], "full_frame_tile_cycles": int(cost[wt]), "alone_ms": round(st.kernel_ms_primary, 4), "alone_wave_cycles": int(st.prim_tests),
                      "same_pixels_as_in_the_full_frame": bool(np.abs(full - crop).max() < 1e-3), "generations": int(st.generations),
                      "rays": {"refraction": int(st.rays_refraction), "reflection": int(st.rays_reflection), "shadow": int(st.rays_shadow), "shadow_not_traced": int(st.rays_shadow_elided)},
                      "share_closest_primary": round(dbg[4] / tot, 3), "share_closest_continuation": round(dbg[5] / tot, 3), "share_shadow": round(dbg[6] / tot, 3),
                      "share_node_loops": round(st.node_tests / tot, 3), "share_leaf_phases": round(st.tri_tests / tot, 3), "share_triangle_leaves": round(st.hit_records / tot, 3),
                      "outside_queries": {k: round(dbg[8 + i] / tot, 4) for i, k in enumerate(["dequeue_wait", "raygen_and_root_test", "hit_reconstruction_and_gates", "shadow_ray_setup", "material", "weights_and_continuation", "opacity_sample(split build)", "material_compute(split build)"])},
                      "node_loop": {"wave_iterations": int(dbg[0]), "lane_iterations": int(dbg[1]), "simd_efficiency": round(dbg[1] / max(64 * dbg[0], 1), 3), "cycles_per_wave_iteration": round(st.node_tests / max(dbg[0], 1)),
                                    "uniform_share": round(dbg[7] / max(dbg[0], 1), 3)},
                      "triangle_loop": {"wave_iterations": int(dbg[2]), "lane_iterations": int(dbg[3]), "simd_efficiency": round(dbg[3] / max(64 * dbg[2], 1), 3), "cycles_per_wave_iteration": round(st.hit_records / max(dbg[2], 1))}}), flush=True)
    if os.environ.get("DEPTH_SCAN"):  # the same tile with the recursion cut at depth d: what each generation adds
        scan = []
        for d in range(1, 11):
            q.max_depth = d
            for _ in range(2):
                abi.check(lib.nrays_render_device(s2.device_handle(), C.byref(q), C.c_void_p(o2.data_ptr()), None))
            s_ = nr.get_stats(s2)
            scan.append({"max_depth": d, "ms": round(s_.kernel_ms_primary, 4), "cycles": int(s_.prim_tests), "refraction": int(s_.rays_refraction), "shadow": int(s_.rays_shadow), "not_traced": int(s_.rays_shadow_elided),
                         "node_iters": None})
        print(json.dumps({"rank": r, "depth_scan": scan}), flush=True)
    del s2
