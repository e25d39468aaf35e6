"""CPU coverage of the multi-GPU path (world size 2, gloo): band ownership, the single gather and
the band layout contract of the compact tile buffers.  The tiles are rendered by the oracle here
(no GPU in this suite); on the GPU box the same plumbing moves HIP-rendered tiles over RCCL and the
un-permutation runs in the k_untile kernel (tests/test_parity_gpu.py covers that side)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

from nrays_amd import abi, tiling


def test_band_bookkeeping_matches_the_c_abi(built):
    lib = abi.load_hip_lib()
    for (h, band, world) in [(1080, 16, 8), (37, 8, 3), (100, 16, 4), (16, 16, 2), (5, 16, 2)]:
        p = abi.NraysRenderParams()
        p.width, p.height = 7, h
        rows = set()
        for r in range(world):
            q = tiling.tile_params(p, r, world, band)
            assert lib.nrays_tile_rows(C.byref(q)) == tiling.tile_rows(h, band, world)
            mine = tiling.owned_rows(h, band, r, world)
            assert len(mine) <= tiling.tile_rows(h, band, world)
            rows |= set(mine)
        assert rows == set(range(h))
    q = tiling.tile_params(p, 0, 1, 16)
    assert q.band_owners == 1 and q.band_rows == 0


def _worker(rank, world, port, w, h, band, result_path):
    import torch
    import torch.distributed as dist
    import oracle
    from tools import scenes_util as su
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc, cam = su.balls_scene(tex_size=(64, 32))
    full, _ = su.camera_params(cam, w, h)
    p = tiling.tile_params(full, rank, world, band)
    tile, _ = oracle.render(sc.descriptor, p, 1)
    assert tile.shape[0] == tiling.tile_rows(h, band, world)
    g = tiling.gather_tiles(torch.from_numpy(tile), rank, world)
    if rank == 0:
        g = g.numpy()
        frame = np.zeros((h, w, 3), np.float32)
        for r in range(world):
            for k, j in enumerate(tiling.owned_rows(h, band, r, world)):
                frame[j] = g[r, k]
        ref, _ = oracle.render(sc.descriptor, full, 1)
        np.save(result_path, np.array([float(np.abs(frame - ref).max())]))
    else:
        assert g is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gather_a_bit_identical_frame(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "err.npy")
    mp.spawn(_worker, args=(2, port, 40, 37, 8, out), nprocs=2, join=True)
    assert np.load(out)[0] == 0.0


def _pipeline_worker(rank, world, port, result_path):
    """FramePipeline with a fake renderer (frame k, rank r -> constant tile k*10 + r): every frame must
    reach rank 0 intact and in order although render k+1 is enqueued before gather k is consumed."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tiles = [torch.zeros((4, 5, 3)), torch.zeros((4, 5, 3))]
    state = {"k": 0, "seen": []}

    def render(t):
        t.fill_(state["k"] * 10 + rank)
        state["k"] += 1

    def untile(g, idx):
        state["seen"].append((idx, [float(g[r].mean()) for r in range(world)], bool((g[0] == g[0, 0, 0, 0]).all())))

    pipe = tiling.FramePipeline(rank, world, tiles, render, untile)
    for _ in range(5):
        pipe.step()
    pipe.flush()
    if rank == 0:
        ok = len(state["seen"]) == 5 and all(idx == k and vals == [k * 10.0 + r for r in range(world)] and uniform
                                              for k, (idx, vals, uniform) in enumerate(state["seen"]))
        np.save(result_path, np.array([1.0 if ok else 0.0]))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_pipeline_two_ranks(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "ok.npy")
    mp.spawn(_pipeline_worker, args=(2, port, out), nprocs=2, join=True)
    assert np.load(out)[0] == 1.0
