#!/usr/bin/env python
"""bench.py — headline benchmark of the nrays trace loop on MI355X.

Metric (BASELINE.json): Mrays/s (primary + shadow + reflection), 1920x1080, 4 bounces.
A step = one scene::render of the workload: scenes/balls.scene at 1920x1080 with `refl 0.2 0.25`
(exactly four reflection generations), 1 ray per pixel, inputs resident in HBM.
A ray = one BVT query (primary, reflection, refraction or shadow; src/scene.rs:153,166).

  python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches it under torch.distributed.run, one rank per GPU: the frame is tiled in
16-row bands dealt round-robin to the ranks, each rank renders its compact tile, one RCCL gather
brings the tiles to rank 0, a HIP kernel un-permutes them (strong scaling: the frame is fixed).
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default="balls", choices=["balls", "sponza"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU-baseline leg")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the nrays_amd product path has no CPU fallback")
    # Plumbing test on a 1-GPU box (never the measured configuration): NRAYS_BENCH_ONE_DEVICE=1 puts every rank on
    # cuda:0 and moves the tiles with gloo, so that the whole N > 1 code path of this file can be executed.
    one_device = os.environ.get("NRAYS_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as graft
    if rank == 0:
        graft.build()
    if world > 1:
        dist.barrier()
    import nrays_amd as nr
    from nrays_amd import abi, tiling
    from tests import scenes_util as su

    lib = abi.load_hip_lib()
    W, H = args.width, args.height
    if args.scene == "balls":
        scene, cam = su.balls_scene()
        workload = "scenes/balls.scene %dx%d, refl 0.2 0.25 (4 reflection bounces), 1 ray/pixel, procedural globe texture" % (W, H)
    else:
        from tests import standins
        scene, cam = standins.sponza_scene()
        workload = "crytek_sponza stand-in (procedural, %d tris) %dx%d, 1 light" % (standins.SPONZA_TRIS, W, H)
    full, _ = su.camera_params(cam, W, H)
    band = tiling.DEFAULT_BAND_ROWS
    p = tiling.tile_params(full, rank, world, band)
    rows = lib.nrays_tile_rows(C.byref(p))
    tiles = [torch.zeros((rows, W, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
    tile = tiles[0]
    handle = scene.device_handle()
    stream = torch.cuda.current_stream().cuda_stream

    def render(instrumented=False, into=None):
        fn = lib.nrays_render_device_instrumented if instrumented else lib.nrays_render_device
        abi.check(fn(handle, C.byref(p), C.c_void_p((tile if into is None else into).data_ptr()), C.c_void_p(stream)))

    frame = torch.empty((H, W, 3), dtype=torch.float32, device="cuda") if rank == 0 else None

    # N > 1: render k+1 overlaps the RCCL gather of frame k (nrays_amd.tiling.FramePipeline); every frame is
    # gathered and un-permuted on rank 0 before the timed region ends (flush()).
    pipe = tiling.FramePipeline(rank, world, tiles, lambda t: render(into=t),
                                lambda g, idx: tiling.untile_device(g, W, H, band, world, out=frame)) if world > 1 else None

    def step():
        if pipe is None:
            render()
        else:
            pipe.step()

    # ---- untimed: instrumented frame -> ray counts and algorithmic bytes of the dominant kernel ----
    render(instrumented=True)
    st = nr.get_stats(scene)
    pk = abi.NraysStats()
    abi.check(lib.nrays_get_primary_kernel_stats(handle, C.byref(pk)))
    rays_local = st.total_rays()
    rays_t = torch.tensor([rays_local, st.rays_primary, st.rays_reflection, st.rays_refraction, st.rays_shadow],
                          dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(rays_t)
    rays_total = float(rays_t[0].item())
    owned = len(tiling.owned_rows(H, band, rank, world)) if world > 1 else H
    bytes_primary = pk.algorithmic_bytes(W, owned)

    for _ in range(args.warmup):
        step()
    if pipe is not None:
        pipe.flush()
    nr.get_stats(scene)  # drains the event ring so that the averages below cover the timed steps only
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if pipe is not None:
        pipe.flush()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt_t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
    dt = float(dt_t.item())
    tst = nr.get_stats(scene)  # HIP-event timings of the timed steps (render stream)

    if one_device and world > 1 and rank == 0:  # plumbing test: the gathered, un-permuted frame equals a direct render
        direct = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        abi.check(lib.nrays_render_device(handle, C.byref(full), C.c_void_p(direct.data_ptr()), C.c_void_p(stream)))
        torch.cuda.synchronize()
        print("one-device plumbing test: gathered frame identical to a direct render: %s" % bool(torch.equal(direct, frame)), file=sys.stderr, flush=True)

    result = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = rays_total * args.steps / dt / 1e6
        t_primary = tst.kernel_ms_primary * 1e-3
        achieved = bytes_primary / t_primary / 1e9 if t_primary > 0 else 0.0
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_k_primary.json")  # collected for the balls workload at 1920x1080
        if os.path.exists(pmc_path) and args.scene == "balls" and (W, H) == (1920, 1080) and world == 1:
            try:
                traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result = {
            "metric": "Mrays/s (primary+shadow+reflection), 1920x1080, 4 bounces",
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "resolution": [W, H], "ray_per_pixel": 1,
                       "parallelism": "framebuffer bands x%d + RCCL gather" % world if world > 1 else "1 GPU",
                       "rays_per_frame": {"total": int(rays_total), "primary": int(rays_t[1].item()),
                                          "reflection": int(rays_t[2].item()), "refraction": int(rays_t[3].item()),
                                          "shadow": int(rays_t[4].item())}},
            "roofline": {"bound": "hbm", "kernel": "k_primary", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(bytes_primary),
                         "kernel_ms": round(tst.kernel_ms_primary, 5), "frame_gpu_ms": round(tst.kernel_ms_total, 5),
                         "launches_timed": int(tst.frames_timed),
                         "units_per_launch": {"rays": int(pk.total_rays()), "node_tests": int(pk.node_tests),
                                              "tri_tests": int(pk.tri_tests), "prim_tests": int(pk.prim_tests),
                                              "hit_records": int(pk.hit_records), "tex_samples": int(pk.tex_samples)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(scene, full, args.cpu_seconds)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


def cpu_baseline(scene, params, budget_s):
    """The CPU leg: the oracle (a port of the reference algorithm: f64, best-first two-level BVT,
    recursive trace) on all host cores with the reference's static pixel partition (scene.rs:49-66),
    on a bounded sample of the same workload: the full frame if it fits the time budget, else the
    top rows of it (ray counts scale the rate, which is what is reported)."""
    import oracle  # the checker, used here only as the timed CPU baseline
    from nrays_amd import abi
    cores = os.cpu_count() or 1
    probe = abi.NraysRenderParams()
    C.memmove(C.byref(probe), C.byref(params), C.sizeof(probe))
    # probe: every 16th band of 16 rows (a 1/16 sample of the frame spread over its height)
    probe.band_rows, probe.band_owner, probe.band_owners = 16, 7, 16
    t0 = time.perf_counter()
    _, st = oracle.render(scene.descriptor, probe, cores)
    t_probe = max(time.perf_counter() - t0, 1e-6)
    rate = st.total_rays() / t_probe
    est_full = 16.0 * t_probe
    if est_full <= budget_s:
        # whole frames, repeated until the sample is worth ~10-30 s of CPU time (cores x wall), at most `budget_s` of wall
        t0 = time.perf_counter()
        oracle.render(scene.descriptor, params, cores)
        t_frame = max(time.perf_counter() - t0, 1e-6)  # the probe above pays the thread start-up: time one real frame
        reps = int(max(1, min(256, 20.0 / (t_frame * cores), budget_s / t_frame)))
        rays = 0
        t0 = time.perf_counter()
        for _ in range(reps):
            _, st = oracle.render(scene.descriptor, params, cores)
            rays += st.total_rays()
        dt = time.perf_counter() - t0
        rate = rays / dt
        sample = "%d x full %dx%d frame, %d rays, %.2f s wall on %d threads (%.0f CPU-s)" % (
            reps, params.width, params.height, rays, dt, cores, dt * cores)
    else:
        sample = "1/16 of the frame (every 16th 16-row band), %d rays, %.2f s" % (st.total_rays(), t_probe)
    rays = max(st.total_rays(), 1)
    # the same rays through the reference-equivalent tree (median split, one primitive per leaf, best-first
    # search), reported beside the shipped BVH's counts in roofline.units_per_launch
    ref_counts = {"aabb_tests_per_ray": round(st.node_tests / rays, 2), "tri_tests_per_ray": round(st.tri_tests / rays, 2),
                  "prim_tests_per_ray": round(st.prim_tests / rays, 3)}
    return {"value": round(rate / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port", "sample": sample,
            "reference_tree_counts": ref_counts}


if __name__ == "__main__":
    main()
