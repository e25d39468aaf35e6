"""The per-camera scheduling state of a handle (nrays_hip.hip: render_impl — resting camera / nearby camera / cold camera; k_seed_costs, k_tile_order, order reuse, cost
recording) never changes a pixel: a camera's first frame on a fresh handle, the frames of a camera that moves a little every frame, a jump to a far camera and back are
the bit-identical frames of a settled handle (reference: the thread partition of src/scene.rs:49-66 never changes a pixel either; examples/loader3d.rs:67-93 renders every
camera once).  Also: the instrumented render that counts what the timed kernel does (NRAYS_COUNT_AS_TIMED) and the recording launch's own duration / clock."""
import ctypes as C

import numpy as np
import pytest

import nrays_amd as nr
from nrays_amd import abi
from tools import scenes_util as su, standins

pytestmark = pytest.mark.gpu


def _render(lib, sc, p, h, w):
    img = np.empty((h, w, 3), np.float32)
    abi.check(lib.nrays_render(sc.device_handle(), C.byref(p), img.ctypes.data_as(C.POINTER(C.c_float))))
    return img


@pytest.mark.parametrize("make,res", [(su.balls_scene, (320, 180)), (lambda: standins.sponza_scene(detail=0.2), (320, 180)),
                                      (lambda: standins.sponza_scene(detail=0.2, n_lights=8), (192, 108)), (lambda: su.primitives_scene(0.0, 1), (200, 120))])
def test_cold_moving_and_jumping_cameras_render_the_settled_frames(gpu, make, res):
    lib = abi.load_hip_lib()
    w, h = res
    sc, cam = make()
    eye0 = np.array(cam["eye"], dtype=np.float64); at = np.array(cam["at"], dtype=np.float64)
    step = 2e-3 * np.linalg.norm(eye0 - at) * np.array([1.0, 0.3, 0.0])
    cams = [dict(cam, eye=tuple(eye0 + k * step)) for k in range(24)]              # a nearby camera every frame: orders reused, re-sorted every few frames
    far = dict(cam, eye=tuple(eye0 * 1.6 + np.array([2.0, 1.0, 0.5])))              # not "nearby": treated as a cold camera
    params = [su.camera_params(c, w, h)[0] for c in cams]
    pfar = su.camera_params(far, w, h)[0]

    def settled(p):  # a handle of its own, the frame after the scheduling state has settled
        s2, _ = make()
        for _ in range(3):
            img = _render(lib, s2, p, h, w)
        return img
    ref0, ref_last, ref_far = settled(params[0]), settled(params[-1]), settled(pfar)
    cold = _render(lib, make()[0], params[0], h, w)                                  # first frame of a fresh handle
    assert np.array_equal(cold, ref0)
    frames = [_render(lib, sc, p, h, w) for p in params]                             # one handle, moving camera
    assert np.array_equal(frames[0], ref0) and np.array_equal(frames[-1], ref_last)
    assert np.array_equal(_render(lib, sc, pfar, h, w), ref_far)                     # the jump ...
    assert np.array_equal(_render(lib, sc, params[-1], h, w), ref_last)              # ... and back
    for _ in range(3):                                                               # and at rest again
        assert np.array_equal(_render(lib, sc, params[-1], h, w), ref_last)


def test_counting_what_the_timed_kernel_does(gpu):
    """nrays_render_device_counted(NRAYS_COUNT_AS_TIMED): same pixels, the reference's ray counts, the shadow rays a plain render skips reported as skipped and their
    traversal work gone from the counters; node_fetches: at most one 128-byte record per lane and box group, at least one per wave."""
    import torch
    lib = abi.load_hip_lib()
    sc, cam = standins.sponza_scene(detail=0.2, n_lights=8)
    w, h = 256, 144
    p, _ = su.camera_params(cam, w, h)
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    hnd = sc.device_handle()
    abi.check(lib.nrays_render_device(hnd, C.byref(p), C.c_void_p(out.data_ptr()), None)); plain_img = out.cpu().numpy().copy(); plain = nr.get_stats(sc)
    abi.check(lib.nrays_render_device_counted(hnd, C.byref(p), C.c_void_p(out.data_ptr()), None, 0)); ref_img = out.cpu().numpy().copy(); ref = nr.get_stats(sc)
    abi.check(lib.nrays_render_device_counted(hnd, C.byref(p), C.c_void_p(out.data_ptr()), None, abi.COUNT_AS_TIMED)); tim_img = out.cpu().numpy().copy(); tim = nr.get_stats(sc)
    assert np.array_equal(plain_img, ref_img) and np.array_equal(plain_img, tim_img)
    for k in ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow"):
        assert getattr(plain, k) == getattr(ref, k) == getattr(tim, k)
    assert ref.rays_shadow_elided == 0 and tim.rays_shadow_elided == plain.rays_shadow_elided > 0
    assert tim.node_tests < ref.node_tests and tim.tri_tests <= ref.tri_tests and tim.hit_records <= ref.hit_records
    for st in (ref, tim):
        assert 0 < st.node_fetches * 4 and st.node_fetches <= st.node_tests           # a fetched record holds up to four boxes; a lane that tests a box fetched or shared it
        assert st.node_fetches * 4 * 64 >= st.node_tests                               # and one fetch serves at most a wave of lanes with four boxes each
    assert lib.nrays_render_device_counted(hnd, C.byref(p), C.c_void_p(out.data_ptr()), None, 8) != 0   # unknown flag


def test_the_recording_launch_measures_itself(gpu):
    """NraysTileCosts: duration of the launch that recorded the tile costs (its own events) and the shader clock under the scene's load (measured by instrumented launches);
    the longest unit and the sum per resident wave are fractions of THAT launch."""
    import torch
    lib = abi.load_hip_lib()
    for make in (su.balls_scene, lambda: standins.sponza_scene(detail=0.3)):
        sc, cam = make()
        p, _ = su.camera_params(cam, 640, 360)
        out = torch.empty((360, 640, 3), dtype=torch.float32, device="cuda")
        for _ in range(2):
            abi.check(lib.nrays_render_device(sc.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
        tc0 = abi.NraysTileCosts()
        abi.check(lib.nrays_get_tile_costs(sc.device_handle(), C.byref(tc0)))
        assert tc0.shader_clock_hz == 0.0 and tc0.kernel_ms > 0.0                   # the clock is measured by instrumented launches only (the plain kernels carry no code for it)
        abi.check(lib.nrays_render_device_instrumented(sc.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
        tc = abi.NraysTileCosts()
        abi.check(lib.nrays_get_tile_costs(sc.device_handle(), C.byref(tc)))
        assert tc.tiles > 0 and tc.kernel_ms > 0.0
        assert 0.5e9 < tc.shader_clock_hz < 3.0e9
        t = tc.kernel_ms * 1e-3
        assert tc.max_cycles / tc.shader_clock_hz <= 1.05 * t
        assert tc.sum_cycles / tc.resident_waves / tc.shader_clock_hz <= 1.05 * t
