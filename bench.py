#!/usr/bin/env python
"""bench.py — headline benchmark of the nrays trace loop on MI355X.

Metric (BASELINE.json): Mrays/s (primary + shadow + reflection), 1920x1080, 4 bounces.
A step = one scene::render of the workload: scenes/balls.scene (loaded through the loader3d front-end) at
1920x1080 with `refl 0.2 0.25` (exactly four reflection generations), 1 ray per pixel,
inputs resident in HBM.  A ray = one BVT query (primary, reflection, refraction or shadow; src/scene.rs:153,166).

  python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches it under torch.distributed.run, one rank per GPU: the frame is tiled in 16-row bands
dealt round-robin to the ranks, each rank renders its compact tile, grouped RCCL send / receive brings the tiles to
GPU 0, a HIP kernel un-permutes them — all inside the library (nrays_render_multi_device, one C call per frame; strong
scaling: the frame is fixed).  Rank 0 prints ONE JSON line.  (`--gpus N` without a launcher: one process drives N owners.)

At N = 1 the line also carries
  roofline       contract fields (achieved = algorithmic bytes / kernel time against the 8 TB/s HBM peak) PLUS what
                 actually bounds the kernel: `bound` / `limiter` (longest wave tile vs sum of tile cycles per resident
                 wave, from the library's per-tile cycle counts), dram_frac, valu_active_frac, wave_wait_frac,
                 fp64_issue_frac, l2_gbs, compulsory_bytes — from rocprofv3 --pmc passes run live by this script
                 (traffic_source says whether the counters are live or replayed from profiles/);
  value_cold, value_moving, scene_build_s, cold_frame_ms, moving_camera_ms_per_frame   what a caller gets outside the resting-camera steady state
  two_frames_in_flight_ms_per_frame   two handles / streams / frame buffers alternating (information; never `value`)
                 the contract's loop times (the reference's only caller renders each camera ONCE);
  value_traced   the rate on the rays that went through a BVT query (wave tiles outside the scene's screen bounds are
                 written without one);
  cpu_baseline   the oracle (a port of the reference algorithm) on all host cores, scene built before the clock starts;
  secondary      the crytek_sponza stand-in (BASELINE config 3, the north star's >= 100x target) and the hairball
                 stand-in with their own roofline and cpu_baseline blocks; at N > 1 BASELINE config 4 (3840x2160,
                 8 lights), the frame that can scale.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
NUM_SIMDS = 256 * 4    # same guide: 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9       # same guide: max clock (the effective clock under load is lower: fractions below are lower bounds)
# SIMD cycles (s_memtime ticks) per wave64 instruction with the SIMD saturated, measured with tools/probe/valu_rate.hip on
# MI355X (profiles/r02_valu_rate.json): v_fma/mul/add_f64 1.35 (f32: 1.05-1.27), v_rcp_f64 5.2.  A wave ALONE on its SIMD
# issues one VALU instruction per ~5.4 cycles whatever its type: the kernels are bound by that per-wave issue latency at
# 2-3 waves per SIMD, not by the f64 rate.
CYCLES_PER_F64_WAVE_INSTR = 1.35
CYCLES_PER_F64_TRANS_WAVE_INSTR = 5.2


def load_workload(name):
    """Returns (scene object with .descriptor / .device_handle(), camera dict, description with %d x %d for the resolution)."""
    if name == "balls":
        from tools import gen_assets
        from nrays_amd import scenefile
        gen_assets.gen_globe()  # scenes/media/ is not shipped (as upstream): deterministic procedural globe.png
        fs = scenefile.FileScene(os.path.join(ROOT, "scenes", "balls.scene"))
        cam = fs.camera_dict()
        return fs, cam, "scenes/balls.scene via the loader3d front-end, %dx%d, refl 0.2 0.25 (4 reflection bounces), 1 ray/pixel, procedural globe.png"
    from tools import standins
    if name in ("hairball", "config5"):
        sc, cam = standins.hairball_scene()
        if name == "config5":  # BASELINE config 5: 3840x2160, 64 AA samples per pixel in a one-pixel window, counter-based RNG seed 1
            cam = dict(cam, spp=64, window=1.0, seed=1)
            return sc, cam, "hairball STAND-IN (procedural, 2.88 M triangles; the real asset is not shipped upstream), %dx%d, 1 light, 64 rays/pixel (aa 64 1.0, BASELINE config 5)"
        return sc, cam, "hairball STAND-IN (procedural, 2.88 M triangles: 3000 strands x 8 sides x 60 segments; the real asset is not shipped upstream), %dx%d, 1 light, 1 ray/pixel"
    sc, cam = standins.sponza_scene(n_lights=8 if name == "config4" else 1)
    return sc, cam, "crytek_sponza STAND-IN (procedural, %d triangles, 276 nodes, alpha-mapped foliage; the real asset is not shipped upstream), " % standins.SPONZA_TRIS + (
        "%dx%d, 8 lights (BASELINE config 4), 1 ray/pixel" if name == "config4" else "%dx%d, 1 light, 1 ray/pixel")


def camera_params(cam, W, H):
    import nrays_amd as nr
    from nrays_amd import math3d
    proj = math3d.inverse_projection(cam["eye"], cam["at"], cam["fovy"], W, H)
    return nr.make_params((W, H), cam.get("spp", 1), cam.get("window", 0.0), cam["eye"], proj, seed=cam.get("seed", 0))


L2_PEAK_GBS = 34500.0  # same guide, "L2 (per XCD)": 32 MiB aggregate, ~34.5 TB/s


def roofline_block(pk, tst, W, owned_rows, scene_bytes, pmc, pmc_source, tile_costs=None, step_ms=None, pk_ref=None):
    """Roofline of the dominant kernel (k_primary).  ONE kernel, one set of units: `pk` holds the counters of an instrumented render that skips exactly what
    the timed plain kernel skips (nrays_render_device_counted, NRAYS_COUNT_AS_TIMED), `pk_ref` the reference algorithm's counts (every ray scene.rs traces),
    reported beside them as `reference_units_per_launch`.

    `frac` = the LARGEST of the kernel's utilisations of real ceilings, each <= 1 by construction:
      hbm    DRAM bytes moved (PMC FETCH_SIZE / WRITE_SIZE, the guide's gfx950 corrections) / kernel time / 8 TB/s
      l2     bytes the waves request from the L2s (TCC requests x 128 B: a wave-uniform scalar node fetch is ONE request, not 64) / kernel time / 34.5 TB/s
      valu   SIMD cycles with a VALU instruction issuing (SQ_ACTIVE_INST_VALU x 4) / SIMD cycles of the kernel at the MEASURED shader clock
    `contract_*` = SURVEY 8(d): algorithmic record bytes of the timed kernel's tests (32 B per box per lane, ...) / kernel time / HBM peak — a rate of useful
    bytes served mostly by caches and scalar broadcasts, not a utilisation; `unique_fetch_*` = the same with the node bytes counted as FETCHED: 128 B once
    per wave for a wave-uniform visit, 128 B per lane otherwise (NraysStats::node_fetches) — the byte count the scalar-broadcast design is held to.
    `limiter`: longest dealt unit and sum of tile cycles per resident wave of the cost-recording launch over THAT launch's own duration, cycles converted
    at the clock that launch measured (NraysTileCosts::kernel_ms, shader_clock_hz): fractions of one launch, <= 1."""
    # Kernel time: HIP events around the launch (every 4th frame) also see its dispatch latency (~3 us) — in the steady loop the dispatch of frame k + 1 hides
    # behind frame k.  A single-launch frame's kernel cannot take longer than the step, so the fractions use min(event time, ms_per_step); `kernel_ms_events`
    # keeps the raw figure.  The rocprofv3 average of the same kernel is in profiles/r06_rocprofv3_kernel_stats_<scene>.csv (tools/final_run.sh).
    t_events = tst.kernel_ms_primary * 1e-3
    single = abs(tst.kernel_ms_total - tst.kernel_ms_primary) <= 1e-9
    t = min(t_events, step_ms * 1e-3) if (step_ms and single and t_events > 0) else t_events
    fb = 12 * W * owned_rows
    other = 36 * pk.tri_tests + 64 * pk.prim_tests + 64 * pk.hit_records + 16 * pk.tex_samples + fb
    record_bytes = 32 * pk.node_tests + other
    unique_bytes = 128 * pk.node_fetches + other
    contract = record_bytes / t / 1e9 if t > 0 else 0.0
    unique = unique_bytes / t / 1e9 if t > 0 else 0.0
    clk = tile_costs.shader_clock_hz if (tile_costs is not None and tile_costs.shader_clock_hz > 1e8) else None

    def units(k):
        return {"rays": int(k.total_rays()), "rays_traced": int(k.rays_traced()), "rays_shadow_counted_not_traced": int(k.rays_shadow_elided),
                "node_tests": int(k.node_tests), "node_fetches_128B": int(k.node_fetches), "tri_tests": int(k.tri_tests), "prim_tests": int(k.prim_tests),
                "hit_records": int(k.hit_records), "tex_samples": int(k.tex_samples)}
    r = {"bound": None, "contract_bound": "hbm", "kernel": "k_primary", "achieved": None, "peak": None, "unit": None, "frac": None,
         "traffic": None, "traffic_source": None, "ceilings": {},
         "contract_achieved": round(contract, 2), "contract_peak": HBM_PEAK_GBS, "contract_unit": "GB/s", "contract_frac": round(contract / HBM_PEAK_GBS, 5),
         "algorithmic_bytes_per_launch": int(record_bytes),
         "unique_fetch_bytes_per_launch": int(unique_bytes), "unique_fetch_achieved": round(unique, 2), "unique_fetch_frac": round(unique / HBM_PEAK_GBS, 5),
         "kernel_ms": round(t * 1e3, 5), "kernel_ms_events": round(tst.kernel_ms_primary, 5),
         "frame_gpu_ms": round(tst.kernel_ms_total, 5), "launches_timed": int(tst.frames_timed),
         "compulsory_bytes": int(scene_bytes + fb),
         "shader_clock_ghz": round(clk / 1e9, 4) if clk else None,
         "units_per_launch": units(pk), "units_source": "instrumented render that skips what the timed kernel skips (NRAYS_COUNT_AS_TIMED)",
         "note": "frac = max over real ceilings (hbm: PMC DRAM bytes, l2: TCC requests x 128 B, valu: VALU-issue cycles at the measured clock), each <= 1; "
                 "contract_* = SURVEY 8d algorithmic record bytes of the TIMED kernel's own tests / kernel time / HBM peak (useful bytes served by caches and SGPR "
                 "broadcasts: a rate, not a utilisation); unique_fetch_* = node bytes as fetched (128 B once per wave-uniform visit); reference_units_per_launch = the "
                 "reference algorithm's counts (every shadow ray traced); the kernels are latency-bound (wave_wait_frac): `limiter` names what the schedule waits for"}
    if pk_ref is not None:
        r["reference_units_per_launch"] = units(pk_ref)
    if tile_costs is not None and tile_costs.tiles and tile_costs.kernel_ms > 0 and clk:
        tk = tile_costs.kernel_ms * 1e-3  # the recording launch's own duration
        longest = tile_costs.max_cycles / clk
        through = tile_costs.sum_cycles / max(tile_costs.resident_waves, 1) / clk
        lim = {"longest_tile_cycles": int(tile_costs.max_cycles), "sum_tile_cycles": int(tile_costs.sum_cycles), "wave_tiles": int(tile_costs.tiles),
               "resident_waves": int(tile_costs.resident_waves), "recording_launch_ms": round(tile_costs.kernel_ms, 5),
               "longest_tile_frac": round(longest / tk, 4), "wave_throughput_frac": round(through / tk, 4),
               "source": "nrays_get_tile_costs: s_memtime cycles of every unit the cost-recording launch of this camera dealt (a split tile: by part), converted at the clock "
                         "that launch measured, over that launch's own duration (first wave's start to last wave's end)"}
        lim["name"] = "latency/longest-tile" if longest >= through else "throughput/wave-cycles"
        lim["frac"] = round(max(longest, through) / tk, 4)
        r["limiter"] = lim
    if pmc and t > 0:
        r["traffic_source"] = pmc_source
        ceil = r["ceilings"]
        if pmc.get("hbm_bytes_per_launch") is not None:
            r["traffic"] = float(pmc["hbm_bytes_per_launch"])
            a = pmc["hbm_bytes_per_launch"] / t / 1e9
            ceil["hbm"] = {"achieved": round(a, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(min(a / HBM_PEAK_GBS, 1.0), 5)}
            r["dram_frac"] = ceil["hbm"]["frac"]
        simd_cycles = NUM_SIMDS * t * (clk or CLOCK_HZ)
        if pmc.get("TCC_HIT_sum") is not None and pmc.get("TCC_MISS_sum") is not None:
            req = pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"]
            a = req * 128.0 / t / 1e9  # 128-byte L2 requests (MI355X_MICROARCH.md, HBM section)
            ceil["l2"] = {"achieved": round(a, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(min(a / L2_PEAK_GBS, 1.0), 5)}
            r["l2_hit_rate"] = round(pmc["TCC_HIT_sum"] / max(req, 1.0), 4)
            r["l2_gbs"] = ceil["l2"]["achieved"]
        if pmc.get("SQ_ACTIVE_INST_VALU") is not None:  # quad-cycles summed over waves
            a = 4.0 * pmc["SQ_ACTIVE_INST_VALU"] / t
            ceil["valu"] = {"achieved": round(a / 1e9, 2), "peak": round(NUM_SIMDS * (clk or CLOCK_HZ) / 1e9, 1), "unit": "G SIMD-cycles/s with a VALU instruction issuing",
                            "frac": round(min(4.0 * pmc["SQ_ACTIVE_INST_VALU"] / simd_cycles, 1.0), 4),
                            "clock": "measured by the cost-recording launch" if clk else "the guide's 2.4 GHz maximum (no measured clock)"}
            r["valu_active_frac"] = ceil["valu"]["frac"]
        if pmc.get("SQ_WAIT_ANY") is not None and pmc.get("SQ_WAVE_CYCLES"):
            r["wave_wait_frac"] = round(pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"], 4)  # share of the waves' lifetime spent in s_waitcnt
        f64 = [pmc.get(k) for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64")]
        if all(v is not None for v in f64):
            r["f64_wave_instructions"] = int(sum(f64))
            r["fp64_issue_frac"] = round((sum(f64[:3]) * CYCLES_PER_F64_WAVE_INSTR + f64[3] * CYCLES_PER_F64_TRANS_WAVE_INSTR) / simd_cycles, 4)
        if pmc.get("SQ_INSTS_VALU") is not None:
            r["valu_wave_instructions"] = int(pmc["SQ_INSTS_VALU"])
        if pmc.get("SQ_INSTS_VMEM_RD") is not None:
            r["vector_load_wave_instructions"] = int(pmc["SQ_INSTS_VMEM_RD"])
        if pmc.get("errors"):
            r["pmc_errors"] = pmc["errors"][:2]
        if ceil:
            name = max(ceil, key=lambda k: ceil[k]["frac"])
            r["bound"] = name
            r["achieved"], r["peak"], r["unit"], r["frac"] = ceil[name]["achieved"], ceil[name]["peak"], ceil[name]["unit"], ceil[name]["frac"]
    if r["frac"] is None:
        r["bound"] = "unmeasured: no hardware counters in this run (contract_* only)"
    return r


def pmc_for(workload, W, H, live):
    """Counters of the primary kernel on `workload`: live rocprofv3 passes, else the committed collection."""
    from tools import pmc_collect
    if live:
        try:
            res = pmc_collect.collect(workload, pmc_collect.TRAFFIC_PASSES, steps=3, width=W, height=H, timeout=240)
            if res.get("hbm_bytes_per_launch") is not None or res.get("SQ_ACTIVE_INST_VALU") is not None:
                return res, "live: rocprofv3 --pmc (2 passes, tools/pmc_collect.py TRAFFIC_PASSES) on tools/kbench.py --child %s in this run" % workload
        except Exception as e:  # the bench line must not depend on the profiler
            print("live PMC collection failed: %r" % (e,), file=sys.stderr)
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (rnd, workload))
        if os.path.exists(path) and (W, H) == (1920, 1080):
            try:
                return json.load(open(path)), "replayed from profiles/%s_pmc_%s.json (python tools/pmc_collect.py %s ...; not measured in this run)" % (rnd, workload, workload)
            except Exception:
                pass
    return None, None


_ORACLE_NATIVE = None


def cpu_baseline(scene, params, budget_s):
    """The CPU leg: the oracle (a port of the reference algorithm: f64, best-first two-level BVT, recursive trace) on
    all host cores with the reference's static pixel partition (scene.rs:49-66).  The BVTs are built before the clock
    starts (Scene::new is outside scene::render, loader3d.rs:57-93), the threads are created once and each renders its
    range `reps` times.  The sample is sized from a WARM call (the first one pays thread start-up and page faults) to at
    least 1.5 s of wall time — 10-30 CPU-seconds and far more on a many-core host — and at most `budget_s`."""
    import oracle  # the checker, used here only as the timed CPU baseline
    global _ORACLE_NATIVE
    if _ORACLE_NATIVE is None:  # once per process: the oracle rebuilt on this host for its own cores, accepted only if it renders the portable build's frame
        import numpy as np
        from tools import scenes_util as su
        probe_sc, probe_cam = su.balls_scene()
        pp, _ = su.camera_params(probe_cam, 96, 54)
        want, _st = oracle.render(probe_sc.descriptor, pp, 4)
        _ORACLE_NATIVE = bool(oracle.use_native())
        if _ORACLE_NATIVE:
            got, _st = oracle.render(probe_sc.descriptor, pp, 4)
            if not np.array_equal(got, want):
                raise RuntimeError("the -march=native oracle renders another frame than the portable build")
    cores = os.cpu_count() or 1
    oracle.render_timed(scene.descriptor, params, cores, 1)            # cold: not used
    sec, st = oracle.render_timed(scene.descriptor, params, cores, 1)  # warm: sizes the sample
    per_frame, reps, spent = sec, 1, 2.0 * sec
    while sec < 1.5 and reps < 8192 and spent < budget_s:
        want = int(math.ceil(1.8 / max(per_frame, 1e-5)))                       # frames for ~1.8 s
        afford = int((budget_s - spent) / max(per_frame, 1e-5))                # frames the budget still pays for
        new_reps = max(reps + 1, min(8192, want, max(afford, 1)))
        if new_reps <= reps:
            break
        reps = new_reps
        sec, st = oracle.render_timed(scene.descriptor, params, cores, reps)
        spent += sec
        per_frame = sec / reps
    rays = max(st.total_rays(), 1)
    cpu_sum, cpu_min, cpu_max = oracle.last_thread_cpu()  # CLOCK_THREAD_CPUTIME_ID of every worker thread, gate to last pixel
    sample = ("%d x full %dx%d frame, %d rays, %.2f s wall on %d persistent threads; CPU time of the threads: %.1f s in all, %.3f s the least "
              "loaded, %.3f s the most (static contiguous pixel ranges, scene.rs:61-63: the wall time is the slowest thread's); BVT build "
              "excluded, sized from a warm call") % (reps, params.width, params.height, rays, sec, cores, cpu_sum, cpu_min, cpu_max)
    # the same rays through the reference-equivalent tree (median split, one primitive per leaf, best-first
    # search), reported beside the shipped BVH's counts in roofline.units_per_launch
    ref_counts = {"aabb_tests_per_ray": round(st.node_tests / rays, 2), "tri_tests_per_ray": round(st.tri_tests / rays, 2),
                  "prim_tests_per_ray": round(st.prim_tests / rays, 3)}
    # `cores` = what the host actually gave the threads (CPU seconds of the threads / wall seconds; a container's quota can be far below the
    # logical CPUs it shows: this pool grants 16 to the 256 threads), never the thread count; `value` is the rate ON THOSE cores.
    # value_at_full_host: every logical CPU running its thread at the measured per-thread speed, perfectly balanced — the number a
    # comparison with the whole host has to use (an upper bound: SMT siblings and memory bandwidth are not modelled).
    eff = cpu_sum / max(sec, 1e-9)
    return {"value": round(rays / sec / 1e6, 4), "unit": "Mrays/s", "cores": round(eff, 1), "threads": cores, "logical_cpus": cores, "cpu_quota": cpu_quota(),
            "kind": "port", "build": "gcc -O3 -march=native -ffp-contract=off (built on this host)" if _ORACLE_NATIVE else "gcc -O3 -ffp-contract=off (portable build)",
            "sample": sample,
            "sample_seconds": round(sec, 3), "thread_cpu_seconds": {"sum": round(cpu_sum, 3), "min": round(cpu_min, 4), "max": round(cpu_max, 4)},
            "effective_cores": round(eff, 1),
            "value_per_effective_core": round(rays / max(cpu_sum, 1e-9) / 1e6, 4),
            "value_at_full_host": round(rays / max(cpu_sum / cores, 1e-9) / 1e6, 4),
            "value_at_perfect_balance": round(rays / max(cpu_sum / cores, 1e-9) / 1e6, 4),
            "reference_tree_counts": ref_counts}


def cpu_quota():
    """The container's CPU quota in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(float(q) / float(per), 2)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / per, 2)
    except Exception:
        return None


def reference_toolchain_probe():
    """if a Rust toolchain AND the upstream assets were on the box, real loader3d would be timed too."""
    import shutil
    cargo = shutil.which("cargo")
    media = os.path.join(ROOT, "scenes", "media", "crytek-sponza", "sponza.obj")
    return {"cargo": cargo or "absent", "upstream_assets": "present" if os.path.exists(media) else "absent (scenes/media is not shipped upstream)",
            "rust_reference_timed": False}


def single_gpu_measure(name, W, H, steps, warmup, args, pmc=True, moving=True):
    """One workload on the current device: what a caller gets on a fresh handle (scene build, cold first frame), the
    steady-state loop the contract times, a moving-camera figure, roofline + limiter, CPU baseline."""
    import torch
    import nrays_amd as nr
    from nrays_amd import abi
    lib = abi.load_hip_lib()
    stream = torch.cuda.current_stream().cuda_stream
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")

    def render_on(handle, params, instrumented=False):
        fn = lib.nrays_render_device_instrumented if instrumented else lib.nrays_render_device
        abi.check(fn(handle, C.byref(params), C.c_void_p(out.data_ptr()), C.c_void_p(stream)))

    # (1) process warm-up on a throw-away handle: the first launch of a kernel in a process loads its code object
    warm_scene, cam, desc = load_workload(name)
    p = camera_params(cam, W, H)
    render_on(warm_scene.device_handle(), p)
    torch.cuda.synchronize()
    del warm_scene
    # (2) what a drop-in scene::render delivers for ONE render of a camera (loader3d.rs:67-93 renders each camera once):
    # Scene::new -> nrays_scene_create (flatten, BVH build, upload), then the FIRST frame of the fresh handle — no cost
    # history (mesh scenes: k_seed_costs' guess; analytic scenes: image-order work lists), cost recording on.  Median over
    # several fresh handles (the frame is host-synchronised: launch latency + GPU time + wake-up).
    cold_walls, cold_gpu = [], []
    for _ in range(3):
        cs, _, _ = load_workload(name)
        ch = cs.device_handle()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        render_on(ch, p)
        torch.cuda.synchronize()
        cold_walls.append((time.perf_counter() - t0) * 1e3)
        cold_gpu.append(nr.get_stats(cs).kernel_ms_total)
        del cs
    scene, cam, desc = load_workload(name)
    t0 = time.perf_counter()
    handle = scene.device_handle()
    torch.cuda.synchronize()
    scene_build_s = time.perf_counter() - t0
    first = []
    for _ in range(3):
        t0 = time.perf_counter()
        render_on(handle, p)
        torch.cuda.synchronize()
        first.append((time.perf_counter() - t0) * 1e3)
    cold_walls.append(first[0])
    cold_ms = sorted(cold_walls)[len(cold_walls) // 2]

    # two instrumented renders: the reference algorithm's counts (every ray scene.rs traces: rays_per_frame, the metric's rays), and the counts of the work the
    # timed plain kernel really does (the shadow rays it skips skipped: the roofline's units)
    render_on(handle, p, True)
    st = nr.get_stats(scene)
    pk_ref = abi.NraysStats()
    abi.check(lib.nrays_get_primary_kernel_stats(handle, C.byref(pk_ref)))
    abi.check(lib.nrays_render_device_counted(handle, C.byref(p), C.c_void_p(out.data_ptr()), C.c_void_p(stream), abi.COUNT_AS_TIMED))
    pk = abi.NraysStats()
    abi.check(lib.nrays_get_primary_kernel_stats(handle, C.byref(pk)))
    # (3) the contract's loop.  Untimed, before --warmup: the library settles its per-camera scheduling state in three plain
    # frames of a resting camera (tile costs recorded -> sorted -> order decided; pixels never depend on it), so the timed
    # steps are RESTING-CAMERA STEADY-STATE frames whatever --warmup says (cold_frame_ms / moving_camera_ms_per_frame are the others)
    settle = 4
    for _ in range(settle):
        render_on(handle, p)
    torch.cuda.synchronize()
    tile_costs = abi.NraysTileCosts()
    if lib.nrays_get_tile_costs(handle, C.byref(tile_costs)) != 0:
        tile_costs = None
    for _ in range(warmup):
        render_on(handle, p)
    nr.get_stats(scene)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        render_on(handle, p)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tst = nr.get_stats(scene)
    # the same resting frame, host-synchronised one at a time: what the cold frame's wall time has to be compared with
    sync_walls = []
    for _ in range(10):
        t0 = time.perf_counter()
        render_on(handle, p)
        torch.cuda.synchronize()
        sync_walls.append((time.perf_counter() - t0) * 1e3)
    nr.get_stats(scene)
    plain = st  # the instrumented frame's counters: rays_primary_traced (what reached a BVT query) is only counted there
    # shadow rays whose result is multiplied by exactly 0 (light samples behind the surface; hits that contribute nothing of their own: fully transparent
    # points, perfect mirrors): counted in rays_shadow — the reference traces them — but not traced by the plain (timed) frames; not part of the traced rate
    elided = int(tst.rays_shadow_elided)
    res = {"workload": desc % (W, H), "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 5), "ray_per_pixel": int(cam.get("spp", 1)),
           "value": round(st.total_rays() * steps / dt / 1e6, 3), "unit": "Mrays/s",
           "steady_state": "resting camera: %d untimed settle frames before --warmup (per-camera cost order decided); kernel_ms from HIP events on every 4th frame" % settle,
           "rays_per_frame": {"total": int(st.total_rays()), "primary": int(st.rays_primary), "reflection": int(st.rays_reflection),
                              "refraction": int(st.rays_refraction), "shadow": int(st.rays_shadow)},
           # rays that went through a BVT query: the wave tiles the scene's screen bounds or the root test decide write the
           # background without one (same pixels; the reference would have queried) — the rate on them is the honest traversal rate
           "rays_traced_per_frame": int(plain.rays_traced()) - elided,
           "rays_shadow_counted_not_traced_per_frame": elided,
           "value_traced": round((plain.rays_traced() - elided) * steps / dt / 1e6, 3),
           "scene_build_s": round(scene_build_s, 4),
           "scene_build_note": "nrays_scene_create as the caller sees it; BLASes of >= 2 000 triangles are built on the GPU (nrays_amd/csrc/bvh_device.hip), smaller ones and the TLASes on the host",
           "cold_frame_ms": round(cold_ms, 4), "cold_frame_gpu_ms": round(sorted(cold_gpu)[len(cold_gpu) // 2], 4),
           "second_frame_ms": round(first[1], 4), "third_frame_ms": round(first[2], 4),
           "steady_frame_sync_ms": round(sorted(sync_walls)[len(sync_walls) // 2], 4),
           # the rate a caller gets who renders every camera ONCE (the reference's only call pattern): the frame's rays over the host-synchronised
           # wall time of the first frame of a fresh handle (steady_frame_sync_ms is the resting frame measured the same way)
           "value_cold": round(st.total_rays() / (cold_ms * 1e-3) / 1e6, 3),
           "cold_frame_note": "fresh handle in a warm process (kernels loaded), median of %d: first scene::render of a camera — no cost history "
                              "(mesh scenes: guessed order; analytic scenes: image-order lists), cost recording; host-synchronised wall time" % len(cold_walls)}
    if moving and cam.get("spp", 1) == 1:
        # a camera that moves every frame (eye shifted by 1e-3 of its distance per frame): the library re-sorts mesh frames from the previous
        # frame's costs, analytic frames keep a nearby camera's order for a few frames; frames pipelined like the steady loop
        import numpy as np
        eye0 = np.array(cam["eye"], dtype=np.float64); at = np.array(cam["at"], dtype=np.float64)
        frames = max(10, min(steps, 40))
        params = [camera_params(dict(cam, eye=tuple(eye0 + (k + 1) * 1e-3 * np.linalg.norm(eye0 - at) * np.array([1.0, 0.0, 0.0]))), W, H) for k in range(frames)]
        best = None
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for q in params:
                render_on(handle, q)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / frames * 1e3
            best = ms if best is None else min(best, ms)
        res["moving_camera_ms_per_frame"] = round(best, 5)
        # (the rays of the moving frames differ from the resting frame's by a fraction of a percent: the resting frame's count is used)
        res["value_moving"] = round(st.total_rays() / (best * 1e-3) / 1e6, 3)
        # the last camera of the path at rest on a handle of its own: what that frame costs when nothing has to be learnt about it
        rs, _, _ = load_workload(name)
        rh = rs.device_handle()
        for _ in range(6):
            render_on(rh, params[-1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            render_on(rh, params[-1])
        torch.cuda.synchronize()
        res["moving_path_last_camera_at_rest_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 5)
        del rs
    if moving and cam.get("spp", 1) == 1:
        # two frames in flight: a second handle of the same scene, a second stream and a second frame buffer, renders alternating —
        # what a caller who double-buffers an animation gets (the tail of one launch overlaps the next one; a handle serialises its own
        # frames).  Information only: `value` stays the one-frame-at-a-time rate.
        scene2, _, _ = load_workload(name)
        handle2 = scene2.device_handle()
        out2 = torch.empty_like(out)
        s2 = torch.cuda.Stream()
        def render2():
            abi.check(lib.nrays_render_device(handle2, C.byref(p), C.c_void_p(out2.data_ptr()), C.c_void_p(s2.cuda_stream)))
        for _ in range(settle + 4):
            render_on(handle, p); render2()
        torch.cuda.synchronize()
        frames = max(20, min(steps, 200)) // 2 * 2
        t0 = time.perf_counter()
        for k in range(frames // 2):
            render_on(handle, p); render2()
        torch.cuda.synchronize()
        res["two_frames_in_flight_ms_per_frame"] = round((time.perf_counter() - t0) / frames * 1e3, 5)
        res["two_frames_in_flight_identical"] = bool(torch.equal(out, out2))
        del scene2
    pmc_res, src = (None, None) if (args.no_pmc or not pmc) else pmc_for(name, W, H, live=not args.replay_pmc)
    res["roofline"] = roofline_block(pk, tst, W, H, lib.nrays_scene_device_bytes(handle), pmc_res, src, tile_costs, step_ms=res["ms_per_step"], pk_ref=pk_ref)
    if not args.no_cpu_baseline:
        # bounded sample: a 64-spp 4K frame is ~200 CPU-core-minutes — the CPU leg of config 5 renders the same camera at an eighth
        # of the resolution in each direction (same scene, same samples per pixel), and says so
        cp = camera_params(cam, max(W // 8, 1), max(H // 8, 1)) if cam.get("spp", 1) > 1 else p
        res["cpu_baseline"] = cpu_baseline(scene, cp, args.cpu_seconds)
        if cp is not p:
            res["cpu_baseline"]["sample"] += " (the workload's camera at %dx%d)" % (cp.width, cp.height)
        res["gpu_over_cpu"] = round(res["value"] / max(res["cpu_baseline"]["value"], 1e-9), 1)  # against the `cores` the host granted
        res["gpu_over_cpu_at_full_host"] = round(res["value"] / max(res["cpu_baseline"]["value_at_full_host"], 1e-9), 1)  # against every logical CPU, balanced
        # the same two ratios with the GPU's rate on the rays it really sent through a BVT query (the CPU port traces every ray it counts)
        res["gpu_traced_over_cpu"] = round(res["value_traced"] / max(res["cpu_baseline"]["value"], 1e-9), 1)
        res["gpu_traced_over_cpu_at_full_host"] = round(res["value_traced"] / max(res["cpu_baseline"]["value_at_full_host"], 1e-9), 1)
    return res


def drop_in_end_to_end(args):
    """What the reference's only caller does, end to end (examples/loader3d.rs:34-101): start a process, parse the .scene / OBJ / MTL / textures, Scene::new, render
    every camera ONCE, quantise, write the PNG.  Reported, not optimised: the C++ loader3d CLI of this repo (nrays_amd/host/loader3d.cpp --times) on
    scenes/crytek_sponza.scene over the stand-in's files (tools/gen_assets.py; written before the clock starts), run twice (the second process finds the files in
    the page cache), beside the CPU leg on the same file: the same parser + the oracle (BVT build + one frame on every host thread)."""
    import shutil
    import subprocess
    import tempfile
    from tools import gen_assets
    from nrays_amd import scenefile
    import oracle
    gen_assets.gen_sponza(1.0)
    scene = os.path.join(ROOT, "scenes", "crytek_sponza.scene")
    exe = os.path.join(ROOT, "nrays_amd", "lib", "loader3d")
    runs = []
    tmp = tempfile.mkdtemp(prefix="nrays_e2e_")
    try:
        for _ in range(2):
            t0 = time.perf_counter()
            r = subprocess.run([exe, scene, "--times"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
            wall = time.perf_counter() - t0
            line = [l for l in r.stdout.splitlines() if l.startswith("{\"loader3d_times_ms\"")]
            if r.returncode != 0 or not line:
                return {"error": (r.stderr or r.stdout)[-400:]}
            d = json.loads(line[-1])
            d["process_wall_ms"] = round(wall * 1e3, 1)
            d["process_start_and_exit_ms"] = round(wall * 1e3 - d["loader3d_times_ms"]["total"], 1)
            runs.append(d)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {"command": "nrays_amd/lib/loader3d scenes/crytek_sponza.scene --times (1920x1080, one camera, stand-in OBJ %.1f MB + MTL + PNG textures)" % (os.path.getsize(os.path.join(ROOT, "scenes", "media", "crytek-sponza", "sponza.obj")) / 1e6),
           "first_process": runs[0], "second_process": runs[1]}
    if not args.no_cpu_baseline:
        t0 = time.perf_counter()
        fs = scenefile.FileScene(scene)
        cam = fs.camera_dict()
        parse_s = time.perf_counter() - t0
        W, H = 1920, 1080
        import nrays_amd as nr
        pp = nr.make_params((W, H), 1, 0.0, cam["eye"], fs.inverse_projection(0, W, H))
        cores = os.cpu_count() or 1
        t0 = time.perf_counter()
        _img, st = oracle.render(fs.descriptor, pp, cores)  # BVT build + one frame
        total_s = time.perf_counter() - t0
        sec, _ = oracle.render_timed(fs.descriptor, pp, cores, 1)
        res["cpu_leg"] = {"kind": "port", "threads": cores, "parse_scene_obj_mtl_textures_ms": round(parse_s * 1e3, 1), "bvt_build_and_first_frame_ms": round(total_s * 1e3, 1),
                          "frame_only_ms": round(sec * 1e3, 1), "rays": int(st.total_rays()),
                          "note": "the same C++ parser (libnrays_host.so) + the oracle (a port of the reference algorithm) on every host thread; PNG write not included"}
    g = runs[1]["loader3d_times_ms"]
    res["note"] = ("the %.1f ms frame is %.1f %% of the second process's %.0f ms: parsing the OBJ (%.0f ms) and nrays_scene_create with the process's HIP initialisation and first copies "
                   "(%.0f ms) dominate a one-shot run" % (g["render_gpu_events"], 100.0 * g["render_gpu_events"] / max(g["total"], 1e-9), g["total"], g["parse_scene_obj_mtl_textures"], g["nrays_scene_create_incl_hip_init"]))
    return res


WORKLOAD_RES = {"config4": (3840, 2160), "config5": (3840, 2160)}  # BASELINE configs 4 (sponza, 8 lights) and 5 (hairball, 64 spp)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=0, help="default: 1920 (3840 for --scene config4)")
    ap.add_argument("--height", type=int, default=0, help="default: 1080 (2160 for --scene config4)")
    ap.add_argument("--scene", default="balls", choices=["balls", "sponza", "hairball", "config4", "config5"],
                    help="balls = BASELINE config 2 (the metric's workload); config4 = crytek_sponza stand-in 3840x2160 with 8 lights")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary blocks (sponza / hairball stand-ins; config 4 at N > 1)")
    ap.add_argument("--steady-only", action="store_true", help="no moving-camera / two-frames-in-flight extras (for a rocprofv3 --stats run whose per-kernel averages should be the steady-state launches)")
    ap.add_argument("--no-pmc", action="store_true", help="no hardware counters at all (traffic: null)")
    ap.add_argument("--replay-pmc", action="store_true", help="take the counters from profiles/ instead of running rocprofv3")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-time cap of each CPU-baseline leg")
    args = ap.parse_args()
    dw, dh = WORKLOAD_RES.get(args.scene, (1920, 1080))
    args.width = args.width or dw
    args.height = args.height or dh

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the nrays_amd product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as graft
    if rank == 0:
        graft.build()
        from tools import gen_assets
        gen_assets.gen_globe()  # before the other ranks load scenes/balls.scene
    if world > 1:
        dist.barrier()

    if world == 1 and args.gpus == 1:
        result = run_single(args)
    else:
        result = run_tiled(args, rank, world, args.gpus)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


METRIC = "Mrays/s (primary+shadow+reflection), 1920x1080, 4 bounces"


def run_single(args):
    W, H = args.width, args.height
    m = single_gpu_measure(args.scene, W, H, args.steps, args.warmup, args, moving=not args.steady_only)
    result = {
        "metric": METRIC,
        "value": m["value"], "unit": "Mrays/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": m["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": m["workload"], "resolution": [W, H], "ray_per_pixel": m["ray_per_pixel"], "parallelism": "1 GPU",
                   "rays_per_frame": m["rays_per_frame"], "rays_traced_per_frame": m["rays_traced_per_frame"],
                   "steady_state": m["steady_state"]},
        "value_traced": m["value_traced"],
        # the other two regimes a caller can be in, as rates of the same rays: one render per camera on a fresh handle
        # (host-synchronised: compare with steady_frame_sync_ms, not with ms_per_step), and a camera that moves every frame (pipelined)
        "value_cold": m["value_cold"], "value_moving": m.get("value_moving"),
        "scene_build_s": m["scene_build_s"], "cold_frame_ms": m["cold_frame_ms"], "cold_frame_gpu_ms": m["cold_frame_gpu_ms"],
        "steady_frame_sync_ms": m["steady_frame_sync_ms"], "second_frame_ms": m["second_frame_ms"],
        "third_frame_ms": m["third_frame_ms"], "cold_frame_note": m["cold_frame_note"],
        "moving_camera_ms_per_frame": m.get("moving_camera_ms_per_frame"), "moving_path_last_camera_at_rest_ms": m.get("moving_path_last_camera_at_rest_ms"),
        "two_frames_in_flight_ms_per_frame": m.get("two_frames_in_flight_ms_per_frame"),
        "two_frames_in_flight_identical": m.get("two_frames_in_flight_identical"),
        "roofline": m["roofline"],
    }
    if "cpu_baseline" in m:
        result["cpu_baseline"] = m["cpu_baseline"]
    if args.scene == "balls" and not args.no_secondary:
        # BASELINE config 3 / the north star's ">= 100x CPU on crytek_sponza at 1 GPU", and the hairball (BVH stress): same process, same run
        sec_steps, sec_warm = max(10, min(args.steps, 60)), max(3, min(args.warmup, 10))
        result["secondary"] = {"sponza_standin": single_gpu_measure("sponza", W, H, sec_steps, sec_warm, args),
                               "hairball_standin": single_gpu_measure("hairball", W, H, max(10, min(args.steps, 30)), sec_warm, args, moving=False)}
    sp = result.get("secondary", {}).get("sponza_standin", m if args.scene == "sponza" else None)
    if sp and "gpu_over_cpu" in sp:
        # the north star's ">= 100x CPU-baseline Mrays/s on crytek_sponza at 1 GPU", on the stand-in; both legs count the same rays
        cb = sp["cpu_baseline"]
        result["north_star_sponza"] = {"gpu_over_cpu": sp["gpu_over_cpu"], "cpu_effective_cores": cb["cores"], "cpu_threads": cb["threads"],
                                       "gpu_over_cpu_at_full_host": sp["gpu_over_cpu_at_full_host"], "cpu_logical_cpus": cb["logical_cpus"],
                                       "gpu_traced_over_cpu": sp["gpu_traced_over_cpu"], "gpu_traced_over_cpu_at_full_host": sp["gpu_traced_over_cpu_at_full_host"],
                                       "gpu_mrays_s": sp["value"], "gpu_mrays_s_traced": sp["value_traced"], "cpu_mrays_s": cb["value"],
                                       "cpu_mrays_s_at_full_host": cb["value_at_full_host"], "target": 100.0,
                                       "note": "gpu_over_cpu is GPU vs the cpu_effective_cores the container granted; gpu_over_cpu_at_full_host is the figure for the whole host (stand-in scene, CPU port of the reference algorithm)"}
    if args.scene == "balls" and not args.no_secondary:
        try:
            result["secondary"]["drop_in_end_to_end"] = drop_in_end_to_end(args)
        except Exception as e:  # a report, never a reason to lose the bench line
            result["secondary"]["drop_in_end_to_end"] = {"error": repr(e)}
    result["reference_toolchain"] = reference_toolchain_probe()
    return result


def tiled_measure(name, W, H, steps, warmup, rank, world, owners):
    """N > 1 through the library's own multi-GPU path (nrays_render_multi_device, multi_gpu.cpp): band
    tiling, grouped RCCL send / receive to owner 0, k_untile — one C call per frame and rank, no Python in the step.
      world > 1   one process per GPU (torch.distributed.run): rank r drives owner r; torch.distributed only ships the
                  128-byte RCCL unique id, the barriers and the final reductions;
      world == 1  `--gpus N` without a launcher: ONE process drives N owners on the visible devices (owner o on device
                  o % device_count: on a 1-GPU box a plumbing run of the N-owner path, never a measurement)."""
    import torch
    import torch.distributed as dist
    import nrays_amd as nr
    from nrays_amd import abi, tiling
    lib = abi.load_hip_lib()
    scene, cam, desc = load_workload(name)
    full = camera_params(cam, W, H)
    if world > 1:
        uid = torch.zeros(abi.UNIQUE_ID_BYTES, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid.copy_(torch.tensor(list(tiling.unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, 0)
        comm = tiling.ranked_comm(bytes(uid.cpu().tolist()), world, rank)
        mode = "one process per GPU, framebuffer bands x%d + RCCL send/recv to GPU 0 (nrays_render_multi_device)" % owners
    else:
        ndev = torch.cuda.device_count()
        comm = tiling.local_comm(owners, [o % ndev for o in range(owners)])
        mode = "ONE process drives %d owners on %d device(s), framebuffer bands + %s (nrays_render_multi_device)" % (
            owners, ndev, "RCCL send/recv to GPU 0" if ndev > 1 else "device-to-device copies: plumbing run, not a measurement")
    ss = tiling.SceneSet(scene.descriptor, comm)
    frame = torch.empty((H, W, 3), dtype=torch.float32, device="cuda") if rank == 0 else None
    fptr = frame.data_ptr() if frame is not None else 0

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: this rank's first owner renders its tile instrumented -> units of the dominant kernel; one plain frame -> ray classes
    owner0, h0 = ss.local_scenes()[0]
    band = tiling.DEFAULT_BAND_ROWS
    tp = tiling.tile_params(full, owner0, owners, band)
    rows = lib.nrays_tile_rows(C.byref(tp))
    scratch = torch.empty((rows, W, 3), dtype=torch.float32, device="cuda")
    abi.check(lib.nrays_render_device_counted(h0, C.byref(tp), C.c_void_p(scratch.data_ptr()), None, abi.COUNT_AS_TIMED))  # the timed tile kernel's own work
    pk = abi.NraysStats()
    abi.check(lib.nrays_get_primary_kernel_stats(h0, C.byref(pk)))
    for _ in range(4):  # plain frames: ray classes, and the per-camera scheduling state of every owner settles (see single_gpu_measure)
        ss.render_device(full, fptr)
    ss.sync()
    st = ss.stats()
    tile_costs = abi.NraysTileCosts()
    if lib.nrays_get_tile_costs(h0, C.byref(tile_costs)) != 0:
        tile_costs = None
    rays_t = torch.tensor([st.total_rays(), st.rays_primary, st.rays_reflection, st.rays_refraction, st.rays_shadow],
                          dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(rays_t)
    rays_total = float(rays_t[0].item())

    for _ in range(warmup):
        ss.render_device(full, fptr)
    ss.sync()
    ss.stats()  # drains the event rings so that the averages below cover the timed steps only
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        ss.render_device(full, fptr)
    ss.sync()  # every frame is rendered, exchanged and un-permuted on owner 0 before the clock stops
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        dt_t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
        dt = float(dt_t.item())
    tst = ss.stats()  # HIP-event timings of this rank's first owner over the timed steps
    # where a frame spends its time on every rank (nrays_multi_get_timings: events on the owner's render / communication streams)
    tm = ss.timings()
    tm_t = torch.tensor([tm.render_ms, tm.exchange_ms, tm.untile_ms, float(tm.owner)], dtype=torch.float64, device="cuda")
    if world > 1:
        all_tm = [torch.zeros_like(tm_t) for _ in range(world)]
        dist.all_gather(all_tm, tm_t)
    else:
        all_tm = [tm_t]
    per_rank = [{"owner": int(v[3].item()), "render_ms": round(v[0].item(), 4), "exchange_ms": round(v[1].item(), 4), "untile_ms": round(v[2].item(), 4)} for v in all_tm]

    check = None
    if rank == 0:  # the gathered, un-permuted frame equals a direct single-GPU render of the whole frame (untimed)
        direct = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        abi.check(lib.nrays_render_device(scene.device_handle(), C.byref(full), C.c_void_p(direct.data_ptr()), None))
        torch.cuda.synchronize()
        check = bool(torch.equal(direct, frame))
        elided = int(nr.get_stats(scene).rays_shadow_elided)  # the plain frame's shadow rays that are multiplied by 0: counted, not traced
        abi.check(lib.nrays_render_device_instrumented(scene.device_handle(), C.byref(full), C.c_void_p(direct.data_ptr()), None))
        traced = nr.get_stats(scene).rays_traced() - elided  # (rays_primary_traced is counted by instrumented renders only)
    res = None
    if rank == 0:
        owned = len(tiling.owned_rows(H, band, owner0, owners))
        res = {"value": round(rays_total * steps / dt / 1e6, 3), "unit": "Mrays/s", "steps": steps, "warmup": warmup,
               "ms_per_step": round(dt / steps * 1e3, 5), "value_traced": round(traced * steps / dt / 1e6, 3),
               # per rank (one process per GPU) or of the first owner (one process for all owners): tile render / exchange / un-permute
               "per_rank_ms": per_rank,
               "config": {"workload": desc % (W, H), "resolution": [W, H], "ray_per_pixel": int(cam.get("spp", 1)), "parallelism": mode,
                          "tiled_frame_identical_to_single_gpu_render": check,
                          "rays_per_frame": {"total": int(rays_total), "primary": int(rays_t[1].item()), "reflection": int(rays_t[2].item()),
                                             "refraction": int(rays_t[3].item()), "shadow": int(rays_t[4].item())},
                          "rays_traced_per_frame": int(traced), "rays_shadow_counted_not_traced_per_frame": elided},
               # owner 0's tile kernel; no counters at N > 1 (the profiler leg runs at N = 1 only)
               "roofline": roofline_block(pk, tst, W, owned, lib.nrays_scene_device_bytes(h0), None, None, tile_costs)}
    ss.close()
    lib.nrays_comm_destroy(comm)
    return res


def run_tiled(args, rank, world, owners):
    W, H = args.width, args.height
    m = tiled_measure(args.scene, W, H, args.steps, args.warmup, rank, world, owners)
    # the frame DESIGN.md 7 names as the one to tile (the metric's balls frame takes 0.05 ms on one GPU: the worst possible
    # strong-scaling workload): BASELINE config 4, reported beside the contract's line at every N > 1
    sec = None
    if args.scene == "balls" and not args.no_secondary:
        cw, ch = WORKLOAD_RES["config4"]
        sec = tiled_measure("config4", cw, ch, max(5, min(args.steps, 20)), max(2, min(args.warmup, 5)), rank, world, owners)
    if rank != 0:
        return None
    result = {"metric": METRIC, "value": m["value"], "unit": "Mrays/s", "n_gpus": owners, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": m["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
              "dtype": "f64", "data": "synthetic", "config": m["config"], "value_traced": m["value_traced"], "per_rank_ms": m["per_rank_ms"],
              "roofline": m["roofline"]}
    if sec is not None:
        result["secondary"] = {"config4_sponza_standin_4k_8_lights": sec}
    return result


if __name__ == "__main__":
    main()
