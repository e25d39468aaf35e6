"""GPU tests of the library's multi-GPU entry points (include/nrays_abi.h: nrays_comm_*, nrays_scene_set_*,
nrays_render_multi*).  The GPU box has one device, so the N-owner path runs with every owner on device 0 (their tiles
move by device-to-device copies instead of RCCL send / receive — same partition, same buffers, same pipeline, same
k_untile); the RCCL communicator itself is exercised in its 1-rank form.  The frame must be bit-identical to the
single-GPU render for any number of owners (replaces the thread partition of src/scene.rs:49-66)."""
import ctypes as C
import os

import numpy as np
import pytest

import nrays_amd as nr
import oracle
from nrays_amd import abi, tiling
from tools import scenes_util as su

pytestmark = pytest.mark.gpu


def _single(scene, params):
    lib = abi.load_hip_lib()
    out = np.empty((params.height, params.width, 3), dtype=np.float32)
    abi.check(lib.nrays_render(scene.device_handle(), C.byref(params), out.ctypes.data_as(C.POINTER(C.c_float))))
    return out, nr.get_stats(scene)


@pytest.mark.parametrize("direct", [False, True])
@pytest.mark.parametrize("owners", [1, 2, 3, 8])
def test_render_multi_is_bit_identical_for_any_number_of_owners(gpu, owners, direct, monkeypatch):
    """direct: NRAYS_MULTI_DIRECT=1 (read when the scene set is created) — every band goes straight into its rows of the frame (one strided
    2-D copy per same-device owner, the ragged last band apart), no gather buffer and no k_untile pass."""
    if direct:
        monkeypatch.setenv("NRAYS_MULTI_DIRECT", "1")
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
    from tools import standins
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


# ---------------------------------------------------------------------------------------------------------------------
# More than one rank / more than one device.  The GPU box of this build has ONE device: the tests below that need two
# skip themselves there (and run on a multi-GPU node, where the driver's scaling bench exercises the same entry points);
# what CAN run on one device is the failure path of ncclCommInitRank.
def _device_count():
    import torch
    return torch.cuda.device_count()


def _spawn_ranks(tmp_path, world, devices, extra=()):
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multi_rank_probe.py")
    idfile = str(tmp_path / "uid.bin")
    procs = [subprocess.Popen([sys.executable, probe, idfile, str(world), str(r), str(devices[r])] + list(extra),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for pr in procs:
            outs.append(pr.communicate(timeout=240)[0])
    finally:
        for pr in procs:  # exactly the processes started here
            if pr.poll() is None:
                pr.kill()
    return outs


def test_failed_comm_init_rank_is_an_error_not_a_hang(gpu, tmp_path):
    """Two ranks on ONE device: RCCL refuses the duplicate GPU, ncclCommInitRank fails on both ranks, and
    nrays_comm_create reports NRAYS_ERR_RCCL with the RCCL message instead of hanging or aborting."""
    outs = _spawn_ranks(tmp_path, 2, [0, 0])
    for o in outs:
        line = [l for l in o.splitlines() if l.startswith("rc=")]
        assert line, o
        assert line[-1].startswith("rc=%d " % abi.ERR_RCCL) and "ncclCommInitRank" in line[-1], o
    lib = abi.load_hip_lib()
    bad = C.c_void_p()
    uid = (C.c_uint8 * abi.UNIQUE_ID_BYTES)(*tiling.unique_id())
    assert lib.nrays_comm_create(uid, 2, 5, C.byref(bad)) == abi.ERR_BAD_ARG  # rank outside the group
    assert lib.nrays_comm_create(uid, 0, 0, C.byref(bad)) == abi.ERR_BAD_ARG


def test_two_devices_one_process_exchange_over_rccl(gpu):
    """nrays_comm_create_local over two DISTINCT devices: ncclCommInitAll, the grouped ncclSend / ncclRecv branch of
    nrays_render_multi_device and k_untile — frame bit-identical to the single-GPU render (skipped on a 1-GPU box)."""
    if _device_count() < 2:
        pytest.skip("needs two GPUs")
    lib = abi.load_hip_lib()
    sc, cam = su.mesh_scene()
    p, _ = su.camera_params(cam, 200, 117)
    ref, rst = _single(sc, p)
    for devices in ([0, 1], [0, 1, 0, 1], [1, 0, 1]):
        comm = tiling.local_comm(len(devices), devices)
        ss = tiling.SceneSet(sc.descriptor, comm)
        for _ in range(3):
            assert np.array_equal(ss.render(p), ref), devices
        assert ss.stats().total_rays() == rst.total_rays()
        ss.close()
        lib.nrays_comm_destroy(comm)


def test_two_ranks_one_process_per_gpu(gpu, tmp_path):
    """The ranked form (one process per GPU, what torch.distributed.run launches): rank r on device r, the unique id shipped
    through a file, three pipelined frames through nrays_render_multi; then bench.py --gpus 2 under the launcher."""
    if _device_count() < 2:
        pytest.skip("needs two GPUs")
    import json
    import subprocess
    import sys
    outs = _spawn_ranks(tmp_path, 2, [0, 1], extra=("--render",))
    assert "rc=0" in outs[0] and "rc=0" in outs[1], outs
    assert "identical=1" in outs[0], outs[0]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["tiled_frame_identical_to_single_gpu_render"] is True
