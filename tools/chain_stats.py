"""Longest per-pixel chain of a frame (instrumented render): AABB tests spent on one pixel's whole chain
(`NraysStats.reserved`) next to the frame's average, for the workloads whose tail bounds the frame."""
import ctypes as C, sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import nrays_amd as nr
from nrays_amd import abi
from tools import scenes_util as su, standins
lib = abi.load_hip_lib()
def run(name, sc, cam, w, h):
    p, _ = su.camera_params(cam, w, h)
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    abi.check(lib.nrays_render_device_instrumented(sc.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
    st = nr.get_stats(sc)
    print(json.dumps({"scene": name, "res": [w, h], "rays": st.total_rays(), "node_tests": st.node_tests, "tri_tests": st.tri_tests,
                      "avg_node_tests_per_pixel": round(st.node_tests / (w * h), 1), "max_chain_node_tests": st.reserved,
                      "generations": st.generations, "kernel_ms_instrumented": round(st.kernel_ms_total, 3)}), flush=True)
sc, cam = standins.sponza_scene(); run("sponza", sc, cam, 1920, 1080)
sc, cam = standins.sponza_scene(n_lights=8); run("sponza8", sc, cam, 1920, 1080)
sc, cam = standins.hairball_scene(); run("hairball", sc, cam, 1920, 1080)
sc, cam = su.balls_scene(); run("balls", sc, cam, 1920, 1080)
