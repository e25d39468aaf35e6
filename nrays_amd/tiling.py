"""Multi-GPU framebuffer tiling: one process per GPU, rows dealt to ranks in bands.

The reference's only parallelism is a contiguous static split of the pixel range over CPU threads
(src/scene.rs:49-66).  Here the scene is replicated on every GPU, band b of `band_rows` rows is
rendered by rank b % world, every rank renders into a compact tile buffer, and ONE collective — a
gather to rank 0 over RCCL/xGMI (every peer has its own direct link to GPU 0, so no ring) — is the
only exchange step; rank 0 then un-permutes the bands with the k_untile HIP kernel.
The RNG is keyed by the global pixel index, so the frame does not depend on the number of GPUs.
"""
import ctypes as C

from . import abi

DEFAULT_BAND_ROWS = 16  # one 16x16 workgroup tile high


def band_owner(row, band_rows, world):
    return (row // band_rows) % world


def owned_rows(height, band_rows, rank, world):
    """Global row indices rendered by `rank`, in the order they appear in its compact tile buffer."""
    return [j for j in range(height) if band_owner(j, band_rows, world) == rank]


def tile_rows(height, band_rows, world):
    """Rows of every rank's compact buffer (bands are padded so that all ranks gather equal sizes)."""
    if world <= 1:
        return height
    nb = (height + band_rows - 1) // band_rows
    return ((nb + world - 1) // world) * band_rows


def tile_params(params, rank, world, band_rows=DEFAULT_BAND_ROWS):
    """Copy of `params` restricted to the bands of `rank`."""
    p = abi.NraysRenderParams()
    C.memmove(C.byref(p), C.byref(params), C.sizeof(p))
    if world > 1:
        p.band_rows, p.band_owner, p.band_owners = band_rows, rank, world
    else:
        p.band_rows, p.band_owner, p.band_owners = 0, 0, 1
    return p


def gather_tiles(tile, rank, world, group=None):
    """The single exchange step: gathers every rank's compact tile on rank 0 ((world, rows, W, 3) there,
    None elsewhere).  `tile` is a torch tensor on the rank's device (CPU tensors work with gloo)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return tile.unsqueeze(0)
    if rank == 0:
        out = torch.empty((world,) + tuple(tile.shape), dtype=tile.dtype, device=tile.device)
        dist.gather(tile, gather_list=list(out.unbind(0)), dst=0, group=group)
        return out
    dist.gather(tile, gather_list=None, dst=0, group=group)
    return None


def untile_device(gathered, width, height, band_rows, world, out=None, stream=None):
    """Band interleave -> row-major frame on the GPU (k_untile).  `gathered` is the (world, rows, W, 3)
    CUDA tensor from gather_tiles on rank 0."""
    import torch
    lib = abi.load_hip_lib()
    if out is None:
        out = torch.empty((height, width, 3), dtype=torch.float32, device=gathered.device)
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    abi.check(lib.nrays_untile_device(C.c_void_p(gathered.data_ptr()), C.c_void_p(out.data_ptr()), width, height,
                                      band_rows if world > 1 else height, world, C.c_void_p(stream)))
    return out


class FramePipeline:
    """Depth-1 software pipeline of the multi-GPU frame loop: while the gather of frame k is in flight
    on the communication stream (RCCL), the render of frame k + 1 already runs on the compute stream;
    rank 0 un-permutes frame k right after enqueuing render k + 1.  Two tile buffers and two gather
    buffers alternate, so a buffer is never rewritten before its collective has consumed it.

        render(tile_tensor)            enqueue the rank's tile render into `tile_tensor`
        untile(gathered, frame_index)  rank 0 only: consume the gathered (world, rows, W, 3) tensor
    """

    def __init__(self, rank, world, tiles, render, untile, group=None):
        import torch
        assert len(tiles) == 2
        self.rank, self.world, self.tiles, self.render, self.untile, self.group = rank, world, tiles, render, untile, group
        self.gathered = None
        if world > 1 and rank == 0:
            self.gathered = [torch.empty((world,) + tuple(tiles[0].shape), dtype=tiles[0].dtype, device=tiles[0].device)
                             for _ in range(2)]
            self.gather_lists = [list(g.unbind(0)) for g in self.gathered]  # built once, not per step
        self.pending = None  # (work, slot, frame index)
        self.k = 0

    def _finish(self):
        if self.pending is None:
            return
        work, slot, idx = self.pending
        if work is not None:
            work.wait()  # the compute stream waits for the collective; no host block for NCCL
        if self.rank == 0:
            self.untile(self.gathered[slot] if self.world > 1 else self.tiles[slot].unsqueeze(0), idx)
        self.pending = None

    def step(self):
        import torch.distributed as dist
        slot = self.k & 1
        self.render(self.tiles[slot])
        self._finish()  # frame k - 1: its gather overlapped the render just enqueued
        work = None
        if self.world > 1:
            if self.rank == 0:
                work = dist.gather(self.tiles[slot], gather_list=self.gather_lists[slot], dst=0,
                                   group=self.group, async_op=True)
            else:
                work = dist.gather(self.tiles[slot], gather_list=None, dst=0, group=self.group, async_op=True)
        self.pending = (work, slot, self.k)
        self.k += 1

    def flush(self):
        self._finish()


class SceneSet:
    """The library's own multi-GPU path (include/nrays_abi.h, nrays_amd/csrc/multi_gpu.cpp): a scene replicated on the
    GPUs of a communicator, one call per frame.  `comm` comes from local_comm() (one process drives every owner) or
    ranked_comm() (one process per GPU)."""

    def __init__(self, descriptor, comm):
        lib = abi.load_hip_lib()
        self._lib, self._comm = lib, comm
        h = C.c_void_p()
        abi.check(lib.nrays_scene_set_create(descriptor.pointer(), comm, C.byref(h)))
        self._h = h
        self.owners = lib.nrays_comm_owners(comm)

    def render(self, params, out=None):
        """scene::render on the group; returns the (H, W, 3) float32 frame on the process that drives owner 0."""
        import numpy as np
        if out is None:
            out = np.empty((params.height, params.width, 3), dtype=np.float32)
        abi.check(self._lib.nrays_render_multi(self._h, C.byref(params), out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def render_device(self, params, out_ptr):
        abi.check(self._lib.nrays_render_multi_device(self._h, C.byref(params), C.c_void_p(out_ptr)))

    def sync(self):
        abi.check(self._lib.nrays_multi_sync(self._h))

    def stats(self):
        st = abi.NraysStats()
        abi.check(self._lib.nrays_multi_get_stats(self._h, C.byref(st)))
        return st

    def timings(self):
        """Average render / exchange / un-permute milliseconds of this process's first owner over the frames since the previous call."""
        tm = abi.NraysMultiTimings()
        abi.check(self._lib.nrays_multi_get_timings(self._h, C.byref(tm)))
        return tm

    def local_scenes(self):
        """[(owner index, NraysScene* handle)] of the owners this process drives (handles stay owned by the set)."""
        out = []
        for k in range(self._lib.nrays_scene_set_num_local(self._h)):
            o = C.c_uint32()
            out.append((o.value, C.c_void_p(self._lib.nrays_scene_set_local_scene(self._h, k, C.byref(o)))))
            out[-1] = (o.value, out[-1][1])
        return out

    def close(self):
        if self._h is not None:
            self._lib.nrays_scene_set_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def local_comm(owners, devices=None):
    """One process, `owners` band owners on `devices` (None: owner o on device o % device_count)."""
    lib = abi.load_hip_lib()
    c = C.c_void_p()
    dv = (C.c_int32 * owners)(*devices) if devices is not None else None
    abi.check(lib.nrays_comm_create_local(owners, dv, C.byref(c)))
    return c


def unique_id():
    lib = abi.load_hip_lib()
    buf = (C.c_uint8 * abi.UNIQUE_ID_BYTES)()
    abi.check(lib.nrays_comm_unique_id(buf))
    return bytes(buf)


def ranked_comm(id_bytes, world, rank):
    """One process per GPU: rank `rank` of `world`, on the calling thread's current device."""
    lib = abi.load_hip_lib()
    c = C.c_void_p()
    buf = (C.c_uint8 * abi.UNIQUE_ID_BYTES)(*id_bytes)
    abi.check(lib.nrays_comm_create(buf, world, rank, C.byref(c)))
    return c
