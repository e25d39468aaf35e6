#!/usr/bin/env python
"""Host-side cost of the multi-GPU frame loop, measured on ONE GPU: a 1-rank NCCL (RCCL) group runs bench.py's
FramePipeline (render -> dist.gather -> k_untile, depth-1 overlap) so that every PyTorch / RCCL call of an N-GPU
step is issued, with no peer to wait for.  Compares with the plain single-GPU loop."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import nrays_amd as nr
from nrays_amd import abi, tiling
from tools import scenes_util as su
lib = abi.load_hip_lib()
sc, cam = su.balls_scene()
W, H = 1920, 1080
full, _ = su.camera_params(cam, W, H)
handle = sc.device_handle()
stream = torch.cuda.current_stream().cuda_stream
tiles = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
frame = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
def render(t):
    abi.check(lib.nrays_render_device(handle, C.byref(full), C.c_void_p(t.data_ptr()), C.c_void_p(stream)))
class FakeWorld2(tiling.FramePipeline):
    """world = 1 has no collective in FramePipeline; force the gather calls of the N > 1 branch."""
    def step(self):
        slot = self.k & 1
        self.render(self.tiles[slot])
        self._finish()
        work = dist.gather(self.tiles[slot], gather_list=[self.gathered[slot][0]], dst=0, async_op=True)
        self.pending = (work, slot, self.k)
        self.k += 1
pipe = FakeWorld2(0, 1, tiles, render, lambda g, idx: tiling.untile_device(g, W, H, H, 1, out=frame))
pipe.world = 2  # take the gathered-buffer path in _finish
pipe.gathered = [torch.empty((1, H, W, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
def timeit(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t_enq = time.perf_counter() - t0  # host time to enqueue n steps (no wait for the GPU)
    if fn is pipe.step: pipe.flush()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, t_enq / n * 1e3
print("plain loop        %.4f ms/step (host enqueue %.4f)" % timeit(lambda: render(tiles[0])))
print("pipeline (1 rank) %.4f ms/step (host enqueue %.4f)" % timeit(pipe.step))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(300): pipe.step()
pr.disable(); pipe.flush(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
dist.destroy_process_group()
