import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import nrays_amd as nr
from nrays_amd import abi
from tests import scenes_util as su, standins
lib = abi.load_hip_lib()
for name, (sc, cam) in {"sponza": standins.sponza_scene(), "hairball": standins.hairball_scene()}.items():
    for md in (1, 0):
        p, _ = su.camera_params(cam, 1920, 1080, max_depth=md)
        out = torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda")
        abi.check(lib.nrays_render_device_instrumented(sc.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
        st = nr.get_stats(sc)
        print(name, "max_depth", md, "avg node tests/pixel %.1f" % (st.node_tests / (1920 * 1080)), "max node tests on one pixel chain", st.reserved)
