"""Screen bounds of the scene (nrays_hip.hip: screen_bounds()): wave tiles without a pixel inside the projected bounding box
of the scene write the background without generating a ray.  It must never change a pixel or a ray count: every frame here
is rendered by two handles of the same scene, one with the bounds in use and one created under NRAYS_SCREEN_CULL=0, and
compared bit for bit (and a few against the oracle, which knows nothing of screen bounds).  Cameras: the bench view, the
scene far away / off to one side / partly outside the frame / behind the camera, the eye inside the bounding box, a ray
grazing a ball, jittered anti-aliased frames (the window widens the bounds), band tiling, ragged resolutions."""
import ctypes as C
import math

import numpy as np
import pytest

import nrays_amd as nr
import oracle
from nrays_amd import abi
from tools import scenes_util as su

pytestmark = pytest.mark.gpu
CLASSES = ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow")


def _render(scene, p):
    out = np.empty((p.height, p.width, 3), dtype=np.float32)
    rows = p.height
    if p.band_owners > 1:  # compact buffer of one owner's bands
        from nrays_amd import tiling
        rows = tiling.tile_rows(p.height, p.band_rows, p.band_owners)
        out = np.empty((rows, p.width, 3), dtype=np.float32)
    abi.check(abi.load_hip_lib().nrays_render(scene.device_handle(), C.byref(p), out.ctypes.data_as(C.POINTER(C.c_float))))
    return out, nr.get_stats(scene)


def _pair(make_scene, monkeypatch):
    monkeypatch.delenv("NRAYS_SCREEN_CULL", raising=False)
    a, cam = make_scene()
    a.device_handle()
    monkeypatch.setenv("NRAYS_SCREEN_CULL", "0")  # read once per handle, at creation
    b, _ = make_scene()
    b.device_handle()
    monkeypatch.delenv("NRAYS_SCREEN_CULL", raising=False)
    return a, b, cam


CAMERAS = {
    "bench view": dict(eye=(0.0, 5.0, -10.0), at=(0.0, 0.0, 0.0), fovy=45.0),
    "far away": dict(eye=(0.0, 60.0, -240.0), at=(0.0, 0.0, 0.0), fovy=45.0),
    "scene in a corner": dict(eye=(0.0, 5.0, -10.0), at=(9.0, 4.0, 0.0), fovy=45.0),
    "scene cut by the frame edge": dict(eye=(0.0, 5.0, -10.0), at=(4.5, 0.0, 0.0), fovy=30.0),
    "scene behind the camera": dict(eye=(0.0, 5.0, -10.0), at=(0.0, 10.0, -20.0), fovy=45.0),
    "scene beside the camera": dict(eye=(0.0, 0.0, -2.0), at=(0.0, 0.0, -10.0), fovy=120.0),
    "eye inside the bounding box": dict(eye=(1.05, 0.9, 0.0), at=(0.0, 0.0, 0.0), fovy=70.0),
    "grazing": dict(eye=(-10.0, 1.0, 0.0), at=(10.0, 1.0000001, 0.0), fovy=20.0),
    "wide angle": dict(eye=(0.0, 1.0, -3.2), at=(0.0, 0.0, 0.0), fovy=150.0),
}


@pytest.mark.parametrize("name", sorted(CAMERAS))
def test_balls_frames_do_not_depend_on_the_screen_bounds(gpu, monkeypatch, name):
    a, b, _ = _pair(su.balls_scene, monkeypatch)
    for (w, h) in ((640, 360), (333, 217)):
        p, _ = su.camera_params(CAMERAS[name], w, h)
        ia, sa = _render(a, p)
        ib, sb = _render(b, p)
        assert np.array_equal(ia, ib), name
        for k in CLASSES:
            assert getattr(sa, k) == getattr(sb, k), (name, k)
    p, _ = su.camera_params(CAMERAS[name], 320, 180)
    img, st = _render(a, p)
    ref, ost = oracle.render(a.descriptor, p, 8)
    assert np.abs(img - ref).max() <= 1e-4, name
    for k in CLASSES:
        assert getattr(st, k) == getattr(ost, k), (name, k)


@pytest.mark.parametrize("spp,window", [(4, 1.0), (5, 7.0), (16, 2.5)])
def test_jittered_frames_do_not_depend_on_the_screen_bounds(gpu, monkeypatch, spp, window):
    """The jitter window moves a pixel's samples by up to window / 2 pixels: the bounds are widened by it."""
    a, b, cam = _pair(su.balls_scene, monkeypatch)
    far = dict(eye=(0.0, 20.0, -60.0), at=(6.0, 0.0, 0.0), fovy=45.0)
    for c in (cam, far):
        p, _ = su.camera_params(c, 480, 270, spp=spp, window=window, seed=11)
        ia, sa = _render(a, p)
        ib, sb = _render(b, p)
        assert np.array_equal(ia, ib)
        for k in CLASSES:
            assert getattr(sa, k) == getattr(sb, k), k


def test_mesh_scene_and_band_tiling(gpu, monkeypatch):
    a, b, cam = _pair(lambda: su.mesh_scene(), monkeypatch)
    far = dict(eye=(3.0, 14.0, -45.0), at=(0.0, 0.0, 0.0), fovy=40.0)
    for c in (cam, far):
        for owners, owner, band in ((1, 0, 0), (3, 1, 16), (2, 1, 1)):
            p, _ = su.camera_params(c, 400, 232, band_rows=band, band_owner=owner, band_owners=owners)
            ia, sa = _render(a, p)
            ib, sb = _render(b, p)
            assert np.array_equal(ia, ib), (owners, owner, band)
            for k in CLASSES:
                assert getattr(sa, k) == getattr(sb, k), k


def test_degenerate_cameras_fall_back_to_every_pixel(gpu, monkeypatch):
    """A singular / non-finite inverse projection must not crash or cull: whatever the kernel computes for such rays, it
    computes it for every pixel, with and without the bounds."""
    a, b, cam = _pair(su.balls_scene, monkeypatch)
    p, proj = su.camera_params(cam, 160, 90)
    for k in range(16):
        p.inv_proj_view[k] = 0.0
    p.inv_proj_view[15] = 1.0
    ia, _ = _render(a, p)
    ib, _ = _render(b, p)
    assert np.array_equal(ia, ib, equal_nan=True)
    # eye exactly on a face of the bounding box (x = 3.1 is the +x face of the right ball's box up to its f32 rounding)
    c = dict(eye=(3.1, 0.0, 0.0), at=(0.0, 0.0, 0.0), fovy=60.0)
    p, _ = su.camera_params(c, 160, 90)
    ia, _ = _render(a, p)
    ib, _ = _render(b, p)
    assert np.array_equal(ia, ib)
    assert not math.isnan(float(ia.sum()))


@pytest.mark.parametrize("make", [su.balls_scene, lambda: su.primitives_scene(0.0, 1), lambda: su.primitives_scene(0.1, 10),
                                  lambda: su.random_shapes_scene(5, n=6, with_mesh=False)], ids=["balls", "primitives", "primitives area light", "random shapes"])
def test_scene_records_in_lds_do_not_change_a_frame(gpu, monkeypatch, make):
    """Small analytic scenes are rendered by kernels that read nodes / instances / shading records from an LDS copy
    (DScene::lds_blob); NRAYS_LDS_SCENE=0 keeps a handle on the kernels that read them from HBM.  Same pixels, same rays."""
    monkeypatch.delenv("NRAYS_LDS_SCENE", raising=False)
    a, cam = make()
    a.device_handle()
    monkeypatch.setenv("NRAYS_LDS_SCENE", "0")
    b, _ = make()
    b.device_handle()
    monkeypatch.delenv("NRAYS_LDS_SCENE", raising=False)
    for kw in (dict(), dict(spp=4, window=1.0, seed=3)):
        p, _ = su.camera_params(cam, 320, 200, **kw)
        ia, sa = _render(a, p)
        ib, sb = _render(b, p)
        assert np.array_equal(ia, ib)
        for k in CLASSES:
            assert getattr(sa, k) == getattr(sb, k), k


@pytest.mark.parametrize("make", [su.balls_scene, lambda: su.primitives_scene(0.0, 1), lambda: su.mesh_scene()], ids=["balls", "primitives", "mesh"])
def test_repeated_frames_of_a_resting_camera_are_identical(gpu, make):
    """A handle settles its per-camera scheduling state over the first frames of a camera (tile costs recorded, sorted, then —
    for frames that are a handful of long tiles — cost-ordered work lists on one workgroup per CU; mesh scenes re-sort every
    frame).  None of it may change a pixel or a ray count: six frames of one camera, a second camera in between, all equal to
    what a fresh handle renders first."""
    a, cam = make()
    other = dict(cam, eye=(cam["eye"][0] + 1.5, cam["eye"][1] + 0.5, cam["eye"][2]))
    # anti-aliased frames of the same handle go through the same states with sample-major wave tiles
    pa, _ = su.camera_params(cam, 256, 144, spp=4, window=1.0, seed=9)
    aa_first, aa_s0 = _render(a, pa)
    for k in range(5):
        img, st = _render(a, pa)
        assert np.array_equal(img, aa_first), k
        for c in CLASSES:
            assert getattr(st, c) == getattr(aa_s0, c), (k, c)
    p, _ = su.camera_params(cam, 512, 288)
    q, _ = su.camera_params(other, 512, 288)
    first, s0 = _render(a, p)
    for k in range(6):
        img, st = _render(a, p)
        assert np.array_equal(img, first), k
        for c in CLASSES:
            assert getattr(st, c) == getattr(s0, c), (k, c)
        if k == 3:  # a different camera in between: its own costs, the stale order of the first
            fresh, _ = make()
            want, _ = _render(fresh, q)
            got, _ = _render(a, q)
            assert np.array_equal(got, want)
    ref, ost = oracle.render(a.descriptor, p, 8)
    assert np.abs(first - ref).max() <= 1e-4


def test_frames_enqueued_without_synchronisation(gpu):
    """nrays_render_device only enqueues: fifty frames of one camera issued back to back, so that the asynchronous read-back
    behind the scheduling decision is still in flight ("not ready") while the following frames are launched.  No call may fail
    and the last frame equals a synchronous first one."""
    import torch
    lib = abi.load_hip_lib()
    a, cam = su.balls_scene()
    p, _ = su.camera_params(cam, 960, 540)
    want, _ = _render(su.balls_scene()[0], p)
    out = torch.empty((540, 960, 3), dtype=torch.float32, device="cuda")
    for _ in range(50):
        abi.check(lib.nrays_render_device(a.device_handle(), C.byref(p), C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)


def test_one_handle_rendered_on_alternating_streams(gpu):
    """A caller that double-buffers renders ONE handle on two streams in turn.  The handle owns per-frame device state, so the
    library orders every render behind its predecessor — with an event recorded on the predecessor's stream when that frame
    recorded none of its own (three frames out of four: the timing events are sampled), never with a host synchronisation.  Every
    frame must equal the render of its camera on a fresh handle."""
    import torch
    lib = abi.load_hip_lib()
    for make in (su.balls_scene, su.mesh_scene):
        a, cam = make()
        cams = [cam, dict(cam, eye=(cam["eye"][0] + 0.7, cam["eye"][1], cam["eye"][2])), dict(cam, fovy=cam["fovy"] * 1.2)]
        params = [su.camera_params(c, 320, 180)[0] for c in cams]
        want = [_render(make()[0], q)[0] for q in params]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs = [torch.empty((180, 320, 3), dtype=torch.float32, device="cuda") for _ in range(12)]
        for k, o in enumerate(outs):
            s = streams[k & 1]
            abi.check(lib.nrays_render_device(a.device_handle(), C.byref(params[k % 3]), C.c_void_p(o.data_ptr()), C.c_void_p(s.cuda_stream)))
        torch.cuda.synchronize()
        for k, o in enumerate(outs):
            assert np.array_equal(o.cpu().numpy(), want[k % 3]), (make.__name__, k)
