"""Two handles of one scene rendering on two streams at the same time (what bench.py times as `two_frames_in_flight_ms_per_frame`):
the launches overlap on the device, every handle keeps its own counters, work lists and cost history, and each frame must equal the
frame a single handle renders alone (reference: `scene::render` is a pure function of scene and camera, src/scene.rs:29-36)."""
import ctypes as C

import numpy as np
import pytest

from nrays_amd import abi
from tools import scenes_util as su, standins

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("make", [su.balls_scene, lambda: standins.sponza_scene(detail=0.2), lambda: standins.hairball_scene(strands=300)],
                         ids=["balls", "sponza", "hairball"])
def test_alternating_handles_on_two_streams(gpu, make):
    import torch
    lib = abi.load_hip_lib()
    w, h = 320, 180
    sc0, cam = make()
    p, _ = su.camera_params(cam, w, h)
    alone = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    for _ in range(3):  # image-order frame, cost-ordered frame, steady state: all the same pixels
        abi.check(lib.nrays_render_device(sc0.device_handle(), C.byref(p), C.c_void_p(alone.data_ptr()), None))
    torch.cuda.synchronize()
    want = alone.cpu().numpy()
    sc1, _ = make()
    sc2, _ = make()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    o1, o2 = torch.zeros_like(alone), torch.zeros_like(alone)
    for k in range(12):
        abi.check(lib.nrays_render_device(sc1.device_handle(), C.byref(p), C.c_void_p(o1.data_ptr()), C.c_void_p(s1.cuda_stream)))
        abi.check(lib.nrays_render_device(sc2.device_handle(), C.byref(p), C.c_void_p(o2.data_ptr()), C.c_void_p(s2.cuda_stream)))
        if k in (0, 1, 5, 11):
            torch.cuda.synchronize()
            assert np.array_equal(o1.cpu().numpy(), want), k
            assert np.array_equal(o2.cpu().numpy(), want), k
