"""The two limits the trace loop has that the reference's recursion does not (reference: src/scene.rs:196-252 recurses without a
bound): the 64-generation cap of a chain, and the capacity of the HBM queue that holds the second children of
double-branching hits (trace_device.h: emit_rays).  Both must behave exactly as documented: the capped frame equals the oracle's
(which carries the same cap), an overflowing frame is REPORTED (NRAYS_ERR_QUEUE_OVERFLOW, never a silently incomplete image) and
the handle renders a clean frame afterwards."""
import ctypes as C

import numpy as np
import pytest

import nrays_amd as nr
import oracle
from nrays_amd import abi
from tools import scenes_util as su

pytestmark = pytest.mark.gpu
CLASSES = ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow")


def _render(sc, p):
    img = np.empty((p.height, p.width, 3), np.float32)
    abi.check(abi.load_hip_lib().nrays_render(sc.device_handle(), C.byref(p), img.ctypes.data_as(C.POINTER(C.c_float))))
    return img, nr.get_stats(sc)


def _mirror_box():
    """Camera inside a hollow (non-solid) cuboid of perfect mirrors that never lose energy (refl 0.9 0.0): every chain is cut by the
    generation cap only."""
    nodes = [nr.SceneNode(su.default_material(), 0.9, 0.0, 1.0, 1.0, nr.Isometry3((0, 0, 0)), nr.Cuboid((4.0, 3.0, 5.0)))]
    lights = [nr.Light((0.5, 1.0, -0.5), 0.0, 1, (1, 1, 1))]
    return nr.Scene(nodes, lights, (1, 1, 1)), dict(eye=(0.3, 0.2, -1.0), at=(1.0, 0.6, 4.0), fovy=60.0)


def test_mirror_box_reaches_the_generation_cap(gpu):
    sc, cam = _mirror_box()
    p, _ = su.camera_params(cam, 64, 48)
    img, st = _render(sc, p)
    ref, ost = oracle.render(sc.descriptor, p, 8)
    assert st.generations == 64  # kMaxGenerations (device_types.h) == HARD_DEPTH_CAP (oracle)
    assert np.abs(img - ref).max() <= 1e-4
    for k in CLASSES:
        assert getattr(st, k) == getattr(ost, k), (k, st.as_dict(), ost.as_dict())
    # every chain that stays inside: 64 reflections, the 65th is refused (a few chains leave through an edge: the reflected ray
    # starts 0.001 along its direction, scene.rs:209, which near an edge is outside the box)
    assert 0.99 * 64 * 64 * 48 <= st.rays_reflection <= 64 * 64 * 48


def _facing_panes(alpha=0.5):
    """Two facing planes, both reflective AND transparent (every hit spawns a reflection and a refraction), the camera between them:
    the reflection stays in registers and bounces between the panes until the cap, the refraction of every bounce goes to the queue."""
    def pane(z, nz):
        return nr.SceneNode(su.default_material(), 0.5, 0.001, alpha, 1.0, nr.Isometry3((0, 0, z)), nr.Plane((0, 0, nz)))
    lights = [nr.Light((0.0, 3.0, 0.0), 0.0, 1, (1, 1, 1))]
    return nr.Scene([pane(5.0, -1.0), pane(-5.0, 1.0)], lights, (0.2, 0.3, 0.4)), dict(eye=(0.0, 0.0, 0.0), at=(0.3, 0.2, 5.0), fovy=50.0)


def test_capped_double_branching_chain_equals_the_oracle(gpu):
    sc, cam = _facing_panes()
    p, _ = su.camera_params(cam, 32, 24)  # 768 pixels x 64 queued refractions stay below the queue's minimum capacity (65 536)
    img, st = _render(sc, p)
    ref, ost = oracle.render(sc.descriptor, p, 8)
    assert st.generations == 64 and st.rays_refraction >= 32 * 24 * 60
    assert np.abs(img - ref).max() <= 1e-4
    for k in CLASSES:
        assert getattr(st, k) == getattr(ost, k), (k, st.as_dict(), ost.as_dict())


def test_queue_overflow_is_reported_and_the_next_frame_is_clean(gpu):
    sc, cam = _facing_panes()
    w = h = 256  # capacity = 4 rays per pixel (nrays_hip.hip: render_impl); every pixel queues one refraction per generation
    p, _ = su.camera_params(cam, w, h)
    img = np.empty((h, w, 3), np.float32)
    lib = abi.load_hip_lib()
    rc = lib.nrays_render(sc.device_handle(), C.byref(p), img.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == abi.ERR_QUEUE_OVERFLOW, rc
    assert b"overflow" in lib.nrays_last_error()
    with pytest.raises(abi.NraysError) as ei:  # the stats of that frame say so too
        nr.get_stats(sc)
    assert ei.value.status == abi.ERR_QUEUE_OVERFLOW
    # the same handle, a frame that fits (three generations: at most three queued rays per pixel): complete and equal to the oracle
    p3, _ = su.camera_params(cam, w, h, max_depth=3)
    for _ in range(2):
        img3, st3 = _render(sc, p3)
        ref3, ost3 = oracle.render(sc.descriptor, p3, 16)
        assert np.abs(img3 - ref3).max() <= 1e-4
        for k in CLASSES:
            assert getattr(st3, k) == getattr(ost3, k), (k, st3.as_dict(), ost3.as_dict())
    # ... and a fresh handle renders the same frame (nothing of the overflowing frame was left behind in the old one)
    sc2, _ = _facing_panes()
    img_fresh, _ = _render(sc2, p3)
    assert np.array_equal(img_fresh, img3)
