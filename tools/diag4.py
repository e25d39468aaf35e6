import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import nrays_amd as nr
from nrays_amd import abi
from tests import scenes_util as su, standins
lib = abi.load_hip_lib()
for name, (sc, cam) in {"sponza": standins.sponza_scene(), "hairball": standins.hairball_scene(), "balls": su.balls_scene()}.items():
    p, _ = su.camera_params(cam, 1920, 1080)
    out = torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda")
    for _ in range(3):
        abi.check(lib.nrays_render_device(sc.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
    st = nr.get_stats(sc)
    tot = st.prim_tests
    print(name, "kernel ms %.3f" % st.kernel_ms_primary, "wave-cycles total %.3e  node-loop %.1f%%  leaf+sentinel phase %.1f%%  outside traversal %.1f%%" % (
        tot, 100.0 * st.node_tests / tot, 100.0 * st.tri_tests / tot, 100.0 * (tot - st.node_tests - st.tri_tests) / tot))
