"""GPU tests of the library's multi-GPU entry points (include/nrays_abi.h: nrays_comm_*, nrays_scene_set_*,
nrays_render_multi*).  The GPU box has one device, so the N-owner path runs with every owner on device 0 (their tiles
move by device-to-device copies instead of RCCL send / receive — same partition, same buffers, same pipeline, same
k_untile); the RCCL communicator itself is exercised in its 1-rank form.  The frame must be bit-identical to the
single-GPU render for any number of owners (replaces the thread partition of src/scene.rs:49-66)."""
import ctypes as C

import numpy as np
import pytest

import nrays_amd as nr
import oracle
from nrays_amd import abi, tiling
from tests import scenes_util as su

pytestmark = pytest.mark.gpu


def _single(scene, params):
    lib = abi.load_hip_lib()
    out = np.empty((params.height, params.width, 3), dtype=np.float32)
    abi.check(lib.nrays_render(scene.device_handle(), C.byref(params), out.ctypes.data_as(C.POINTER(C.c_float))))
    return out, nr.get_stats(scene)


@pytest.mark.parametrize("owners", [1, 2, 3, 8])
def test_render_multi_is_bit_identical_for_any_number_of_owners(gpu, owners):
    lib = abi.load_hip_lib()
    sc, cam = su.mesh_scene()
    p, _ = su.camera_params(cam, 200, 117)  # 8 bands of 16 rows, the last one ragged
    ref, rst = _single(sc, p)
    comm = tiling.local_comm(owners, [0] * owners)
    assert lib.nrays_comm_owners(comm) == owners
    ss = tiling.SceneSet(sc.descriptor, comm)
    for _ in range(3):  # later frames run through the cost-ordered work lists and the second buffer set
        img = ss.render(p)
        assert np.array_equal(img, ref)
    st = ss.stats()
    for k in ("rays_primary", "rays_reflection", "rays_refraction", "rays_shadow"):
        assert getattr(st, k) == getattr(rst, k), k
    ss.close()
    lib.nrays_comm_destroy(comm)


def test_pipelined_device_frames_and_aa(gpu):
    """nrays_render_multi_device back to back (render k + 1 overlaps the exchange of frame k), changing camera every
    frame; each frame is checked after the final sync against its own single-GPU render.  AA jitter + area light: the
    RNG is keyed by the global pixel index, so the tiled frames still match bit for bit."""
    import torch
    lib = abi.load_hip_lib()
    sc, cam = su.primitives_scene(light_radius=0.1, nsample=10)
    comm = tiling.local_comm(4, [0, 0, 0, 0])
    ss = tiling.SceneSet(sc.descriptor, comm)
    cams = [cam, dict(cam, eye=(3.0, 4.0, -9.0)), dict(cam, fovy=60.0), cam, dict(cam, at=(0.5, 0.0, 0.0))]
    outs = [torch.empty((72, 96, 3), dtype=torch.float32, device="cuda") for _ in cams]
    params = [su.camera_params(c, 96, 72, spp=3, window=1.0, seed=11)[0] for c in cams]
    for p, o in zip(params, outs):
        ss.render_device(p, o.data_ptr())
    ss.sync()
    for p, o in zip(params, outs):
        ref, _ = _single(sc, p)
        assert np.array_equal(o.cpu().numpy(), ref)
    ref, _ = oracle.render(sc.descriptor, params[0], 8)
    assert np.abs(outs[0].cpu().numpy() - ref).max() <= 1e-4
    ss.close()
    lib.nrays_comm_destroy(comm)


def test_one_set_many_resolutions(gpu):
    """The set's tile / gather buffers only grow: a smaller frame after a larger one must still be exchanged with ITS tile
    size (and a larger one after that must re-allocate while nothing is in flight)."""
    lib = abi.load_hip_lib()
    sc, cam = su.balls_scene(tex_size=(64, 32))
    comm = tiling.local_comm(3, [0, 0, 0])
    ss = tiling.SceneSet(sc.descriptor, comm)
    for (w, h) in [(320, 200), (96, 50), (33, 17), (400, 300), (96, 50)]:
        p, _ = su.camera_params(cam, w, h)
        ref, _ = _single(sc, p)
        assert np.array_equal(ss.render(p), ref), (w, h)
    ss.close()
    lib.nrays_comm_destroy(comm)


def test_ranked_communicator_one_rank_and_errors(gpu):
    lib = abi.load_hip_lib()
    uid = tiling.unique_id()
    assert len(uid) == abi.UNIQUE_ID_BYTES and any(uid)
    comm = tiling.ranked_comm(uid, 1, 0)
    sc, cam = su.balls_scene(tex_size=(64, 32))
    p, _ = su.camera_params(cam, 160, 90)
    ss = tiling.SceneSet(sc.descriptor, comm)
    ref, _ = _single(sc, p)
    assert np.array_equal(ss.render(p), ref)
    # errors are reported, not fatal
    rc = lib.nrays_render_multi(ss._h, C.byref(p), None)
    assert rc == abi.ERR_BAD_ARG and b"output buffer" in lib.nrays_last_error()
    bad = C.c_void_p()
    assert lib.nrays_comm_create_local(2, (C.c_int32 * 2)(0, 99), C.byref(bad)) == abi.ERR_BAD_ARG
    assert lib.nrays_comm_create_local(0, None, C.byref(bad)) == abi.ERR_BAD_ARG
    p.ray_per_pixel = 0
    out = np.zeros((90, 160, 3), np.float32)
    assert lib.nrays_render_multi(ss._h, C.byref(p), out.ctypes.data_as(C.POINTER(C.c_float))) == abi.ERR_BAD_ARG
    ss.close()
    lib.nrays_comm_destroy(comm)


def test_config4_through_the_library_partition(gpu):
    """BASELINE config 4's partition (8 owners, 3840x2160, 8 lights) through nrays_render_multi on a reduced-detail
    stand-in (the full-detail frame is covered by tests/test_configs_gpu.py through the band parameters)."""
    from tests import standins
    lib = abi.load_hip_lib()
    sc, cam = standins.sponza_scene(detail=0.25, n_lights=8)
    p, _ = su.camera_params(cam, 3840, 2160)
    ref, rst = _single(sc, p)
    comm = tiling.local_comm(8, [0] * 8)
    ss = tiling.SceneSet(sc.descriptor, comm)
    img = ss.render(p)
    assert np.array_equal(img, ref)
    assert ss.stats().total_rays() == rst.total_rays()
    ss.close()
    lib.nrays_comm_destroy(comm)
