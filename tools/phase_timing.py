#!/usr/bin/env python
"""Where the wave cycles of the mesh kernels go (GPU box).  Needs a build with -DNR_PHASE_TIMING (tools/kres.py -o
nrays_amd/lib/v/lib_pt.so -DNR_PHASE_TIMING): the kernels then accumulate s_memtime differences per wave —
node loops, leaf phases (of which triangle leaves), whole wave — into the counter fields read here.

  NRAYS_HIP_LIB=nrays_amd/lib/v/lib_pt.so python tools/phase_timing.py sponza hairball
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import nrays_amd as nr
    from nrays_amd import abi
    from tools import scenes_util as su, standins
    lib = abi.load_hip_lib()
    for name in sys.argv[1:] or ["sponza", "hairball"]:
        name, _, spp = name.partition("@")  # hairball@16: anti-aliased, 16 rays per pixel, window 1
        sc, cam = {"sponza": standins.sponza_scene, "hairball": standins.hairball_scene,
                   "sponza8": lambda: standins.sponza_scene(n_lights=8), "balls": su.balls_scene}[name]()
        p, _ = su.camera_params(cam, 1920, 1080, **(dict(spp=int(spp), window=1.0, seed=1) if spp else {}))
        out = torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda")
        for _ in range(3):
            abi.check(lib.nrays_render_device(sc.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
        st = nr.get_stats(sc)
        tot = max(st.prim_tests, 1)
        dbg = (C.c_ulonglong * 16)()
        lib.nrays_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
        abi.check(lib.nrays_debug_counters(sc.device_handle(), dbg))
        print(json.dumps({"scene": name, "ray_per_pixel": int(spp or 1), "ms": round(st.kernel_ms_primary, 3), "wave_cycles": st.prim_tests,
                          "node_loops": round(st.node_tests / tot, 3), "leaf_phases": round(st.tri_tests / tot, 3),
                          "triangle_leaves": round(st.hit_records / tot, 3),
                          "outside_traversal": round(1 - (st.node_tests + st.tri_tests) / tot, 3),
                          "closest_queries_of_primary_rays": round(dbg[4] / tot, 3), "closest_queries_of_continuation_rays": round(dbg[5] / tot, 3),
                          "shadow_queries": round(dbg[6] / tot, 3),
                          "outside_queries": {k: round(dbg[8 + i] / tot, 4) for i, k in enumerate(["dequeue_wait", "raygen_and_root_test", "hit_reconstruction_and_gates",
                                                                                                      "shadow_ray_setup", "material", "weights_and_continuation"])},
                          "node_loop_wave_iterations_with_one_node_for_the_whole_wave": round(dbg[7] / max(dbg[0], 1), 3),
                          "node_loop": {"wave_iterations": dbg[0], "lane_iterations": dbg[1], "simd_efficiency": round(dbg[1] / max(64 * dbg[0], 1), 3),
                                        "lanes_still_in_the_query": round(dbg[14] / max(64 * dbg[0], 1), 3),
                                        "cycles_per_wave_iteration": round(st.node_tests / max(dbg[0], 1))},
                          "triangle_loop": {"wave_iterations": dbg[2], "lane_iterations": dbg[3], "simd_efficiency": round(dbg[3] / max(64 * dbg[2], 1), 3),
                                            "lanes_still_in_the_query": round(dbg[15] / max(64 * dbg[2], 1), 3),
                                            "cycles_per_wave_iteration": round(st.hit_records / max(dbg[2], 1))}}), flush=True)


if __name__ == "__main__":
    main()
