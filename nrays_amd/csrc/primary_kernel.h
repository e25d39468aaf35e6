// primary_kernel.h — k_primary, the persistent-grid megakernel of the trace loop (replaces the pixel loop of scene::render,
// src/scene.rs:49-116), with the helpers it shares with k_bounce / k_cast_batch (nrays_hip.hip).  Device code only: the
// permutations are instantiated by primary_inst.hip, one translation unit per group (NR_PRIMARY_GROUP), so that the ~57
// kernels compile side by side; nrays_hip.hip holds the host side and the small kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#ifndef NR_TILE_PRIO
#define NR_TILE_PRIO 0 // k < NR_TILE_PRIO: priority 3, < 3x: 2, < 8x: 1 (0 = off)
#endif

#include "device_types.h"
#include "scene_handle.h"
#include "tile_device.h"
#include "trace_device.h"

#ifndef NR_STATIC_FIRST
#define NR_STATIC_FIRST 1 // mesh kernels: a wave's first work-list entry is assigned statically (no atomic storm at the start of a launch)
#endif
#ifndef NR_PEEK_STEAL
#define NR_PEEK_STEAL 1 // mesh kernels look at the eight work counters before they try to steal (k_primary)
#endif
#ifndef NR_NT_STORES
#define NR_NT_STORES 1 // frame-buffer stores carry the non-temporal hint: the 25 MB of a 1080p frame do not sweep the scene out of the L2s
#endif
namespace nrays {

#ifndef NRAYS_WAVES_PER_SIMD
#define NRAYS_WAVES_PER_SIMD 2 // second __launch_bounds__ argument: caps the VGPR budget at 512 / this
#endif
constexpr int kTile = 16;          // four consecutive 8x8 wave tiles form a 16x16 pixel block

// XCD-aware dynamic scheduling of the persistent grid (scenes with meshes; analytic-only scenes use per-workgroup
// lists through an LDS counter, see k_primary).  Every XCD owns a work list with its own counter in HBM: without
// history a contiguous range of the image — so one XCD's private 4 MiB L2 keeps seeing the same region of the image
// and of the BVH — with history every 8th entry of the cost-sorted order.  A wave reads the id of the XCD it runs
// on, pulls from that XCD's list and, when it is exhausted, steals from the next one, which removes the tail of a
// static split (foliage pixels cost 10x+ a wall pixel).  Placement only affects speed: any assignment yields the
// same pixels.
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}
// Per-WAVE dequeue of 8x8-pixel wave tiles: lane 0 pulls, the index is broadcast through an SGPR; no workgroup
// barrier is involved, so a wave that drew cheap tiles never waits for a sibling that drew expensive ones.  The
// atomic for the NEXT grab is issued before the current tile is traced, so its latency hides behind the work
// (device-scope atomics on one address retire at only ~9 M/s on this part: fine for 50+ us mesh tiles, useless for
// the ~1 us tiles of analytic scenes).  Wave tiles are numbered so that four consecutive ones form a 16x16 block.
// `grab` = wave tiles per atomic (1; NRAYS_GRAB overrides it for A/B runs).
__device__ __forceinline__ uint32_t issue_grab(uint32_t* work_counters, uint32_t victim, uint32_t grab) {
    uint32_t k = 0;
    if (__lane_id() == 0) k = atomicAdd(&work_counters[victim], grab);
    return k; // valid in lane 0; consumed later through readfirstlane
}

__device__ __forceinline__ void flush_counters(DeviceCounters* ctr, const Cnt& c, bool stats) {
    // wave-level reduction, then one atomic per wave and class
    unsigned sh = c.shadow, rl = c.refl, rf = c.refr, md = c.max_depth, mc = c.max_chain_nodes, el = c.elided;
    unsigned nd = c.node, tr = c.tri, pr = c.prim, ht = c.hit, tx = c.tex, tc = c.traced, nf = c.fetch;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sh += __shfl_down(sh, off); rl += __shfl_down(rl, off); rf += __shfl_down(rf, off); el += __shfl_down(el, off);
        unsigned om = __shfl_down(md, off); md = om > md ? om : md;
        if (stats) { unsigned oc = __shfl_down(mc, off); mc = oc > mc ? oc : mc; }
        if (stats) { nd += __shfl_down(nd, off); tr += __shfl_down(tr, off); pr += __shfl_down(pr, off); ht += __shfl_down(ht, off); tx += __shfl_down(tx, off); tc += __shfl_down(tc, off); nf += __shfl_down(nf, off); }
    }
#ifdef NR_PHASE_TIMING
    unsigned pn = c.cyc_node, pl = c.cyc_leaf, pt = c.cyc_tri; // accumulators only advance in active lanes: take the max over the wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { unsigned a = __shfl_down(pn, off), b = __shfl_down(pl, off), t = __shfl_down(pt, off); pn = a > pn ? a : pn; pl = b > pl ? b : pl; pt = t > pt ? t : pt; }
    unsigned q0 = c.cyc_closest0, q1 = c.cyc_closestN, q2 = c.cyc_shadow, u0 = c.wv_uni;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { unsigned a = __shfl_down(q0, off), b = __shfl_down(q1, off), d = __shfl_down(q2, off); q0 = a > q0 ? a : q0; q1 = b > q1 ? b : q1; q2 = d > q2 ? d : q2; u0 += __shfl_down(u0, off); }
    for (int k_ = 0; k_ < 8; ++k_) {
        unsigned x_ = c.cyc_x[k_];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { unsigned a = __shfl_down(x_, off); x_ = a > x_ ? a : x_; }
        if (__lane_id() == 0) atomicAdd(&ctr->dbg2[k_], (unsigned long long)x_);
    }
    if (__lane_id() == 0) { atomicAdd(&ctr->dbg[4], (unsigned long long)q0); atomicAdd(&ctr->dbg[5], (unsigned long long)q1); atomicAdd(&ctr->dbg[6], (unsigned long long)q2); atomicAdd(&ctr->dbg[7], (unsigned long long)u0); }
    unsigned i0 = c.wv_node, i1 = c.ln_node, i2 = c.wv_tri, i3 = c.ln_tri, i4 = c.inq_node, i5 = c.inq_tri;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { i0 += __shfl_down(i0, off); i1 += __shfl_down(i1, off); i2 += __shfl_down(i2, off); i3 += __shfl_down(i3, off); i4 += __shfl_down(i4, off); i5 += __shfl_down(i5, off); }
    if (__lane_id() == 0) {
#ifndef NR_PT_SPLIT_MATERIAL // (tools/monster_probe.py builds: dbg2[6], [7] hold two more "outside the queries" buckets — opacity sample, material_compute — instead)
        atomicAdd(&ctr->dbg2[6], (unsigned long long)i4); atomicAdd(&ctr->dbg2[7], (unsigned long long)i5);
#endif
        atomicAdd(&ctr->dbg[0], (unsigned long long)i0); atomicAdd(&ctr->dbg[1], (unsigned long long)i1);
        atomicAdd(&ctr->dbg[2], (unsigned long long)i2); atomicAdd(&ctr->dbg[3], (unsigned long long)i3);
        atomicAdd(&ctr->hit_records, (unsigned long long)pt);  // hit_records = triangle leaves (part of the leaf phases)
        atomicAdd(&ctr->node_tests, (unsigned long long)pn);   // tuning builds reuse the instrumented fields:
        atomicAdd(&ctr->tri_tests, (unsigned long long)pl);    // node_tests = cycles in node loops, tri_tests = leaf phases,
        atomicAdd(&ctr->prim_tests, (unsigned long long)c.cyc_other);  // prim_tests = whole-wave cycles
    }
#endif
    if (__lane_id() == 0) {
        if (sh) atomicAdd(&ctr->rays_shadow, (unsigned long long)sh);
        if (el) atomicAdd(&ctr->shadow_elided, (unsigned long long)el);
        if (rl) atomicAdd(&ctr->rays_reflection, (unsigned long long)rl);
        if (rf) atomicAdd(&ctr->rays_refraction, (unsigned long long)rf);
        if (md) atomicMax(&ctr->max_depth, md);
        if (stats && mc) atomicMax(&ctr->max_chain_nodes, mc);
        if (stats) {
            atomicAdd(&ctr->node_tests, (unsigned long long)nd); atomicAdd(&ctr->tri_tests, (unsigned long long)tr);
            atomicAdd(&ctr->prim_tests, (unsigned long long)pr); atomicAdd(&ctr->hit_records, (unsigned long long)ht);
            atomicAdd(&ctr->tex_samples, (unsigned long long)tx);
            if (tc) atomicAdd(&ctr->rays_primary_traced, (unsigned long long)tc);
            if (nf) atomicAdd(&ctr->node_fetches, (unsigned long long)nf);
        }
    }
}

// Waves per SIMD requested per permutation (measured on MI355X, tools/kbench.py): the opaque mesh kernel is throughput-bound (hair:
// 17 % SIMD efficiency in the node loops, no deep chains) and prefers 4 waves with 88 spilled dwords — hairball 2.82 -> 2.66 ms, config 5
// 203.9 -> 194.0 ms against 3 waves; 5 / 6 waves: 3.59 / 5.21 ms (round 3); the others run best at 2 without spills.
#ifndef NR_OPAQUE_MESH_WAVES
#define NR_OPAQUE_MESH_WAVES 4
#endif
constexpr int waves_per_simd(int feat) { return (feat & ~(kFeatMultiSample | kFeatLdsScene | kFeatNoXform | kFeatPark)) == kFeatMesh ? (NRAYS_WAVES_PER_SIMD > NR_OPAQUE_MESH_WAVES ? NRAYS_WAVES_PER_SIMD : NR_OPAQUE_MESH_WAVES) : NRAYS_WAVES_PER_SIMD; }

// OCC != 0: the alpha-shadow mesh permutations also exist at three waves per SIMD (168 VGPRs, ~64 dwords of scratch per lane): a
// wave then runs slower, which lengthens a frame that is as long as its longest tile (sponza 1080p: 1.40 -> 1.67 ms) and shortens a
// frame that is bound by the sum of its tiles (sponza 4K: 4.12 -> 3.65 ms, config 4: 14.6 -> 12.5 ms) — render_impl chooses per frame.
#ifndef NR_PIXEL_SPLIT
#define NR_PIXEL_SPLIT 1 // long tiles of one-light alpha-mapped mesh frames are split by pixels (the light-parallel machinery with one light)
#endif
#ifndef NR_OCC3_AS
#define NR_OCC3_AS 3 // waves per SIMD the OCC = 3 permutations are compiled and launched for (experiments: 4)
#endif
template <bool STATS, int FEAT, bool PLAIN = false, int OCC = 0>
__global__ void __launch_bounds__(kBlock, OCC ? NR_OCC3_AS : waves_per_simd(FEAT)) k_primary(DScene S0, DRender R, QueueOut qo, float* __restrict__ out, DeviceCounters* ctr,
                                                     uint32_t* spill, uint32_t tiles_x, uint32_t tiles_y, uint32_t* work_counters, uint32_t grab_arg,
                                                     uint32_t* zero_counts, DeviceCounters* zero_ctr) {
    // The scheduling path is fixed by the permutation — workgroup lists (0) for analytic-only scenes, XCD-aware HBM
    // dequeue (>= 1) for scenes with meshes — so that each kernel carries one of them; only the full-featured kernels
    // (instrumented renders, double branching) take it from the host at run time.
    const uint32_t grab = (FEAT == kFeatAll || FEAT == 15) ? grab_arg : ((FEAT & kFeatMesh) ? (grab_arg ? grab_arg : 1u) : 0u);
    // kFeatLdsScene: the workgroup's LDS copy of the scene records (DScene::lds_blob); S then points into it, and since the
    // traversal and shading code is inlined here the compiler sees LDS addresses and issues ds_read for every record
    constexpr bool kLdsScene = (FEAT & kFeatLdsScene) != 0;
    __shared__ __attribute__((aligned(16))) uint32_t lds_scene[kLdsScene ? kLdsSceneBytes / 4 : 4];
    DScene S = S0;
    if (kLdsScene) {
        const __attribute__((address_space(1))) uint32_t* src = (const __attribute__((address_space(1))) uint32_t*)S0.lds_blob;
        for (uint32_t w = threadIdx.x; w < S0.lds_bytes / 4u; w += kBlock) lds_scene[w] = src[w];
        __syncthreads();
        const char* lb = (const char*)lds_scene;
        S.nodes = (const BvhNode*)(lb + S0.lds_off[kLdsNodes]);
        S.instances = (const Instance*)(lb + S0.lds_off[kLdsInstances]);
        S.shadow_instances = (const Instance*)(lb + S0.lds_off[kLdsShadowInstances]);
        S.links = (const InstLink*)(lb + S0.lds_off[kLdsLinks]);
        S.shadow_links = (const InstLink*)(lb + S0.lds_off[kLdsShadowLinks]);
        S.shade = (const ShadeRec*)(lb + S0.lds_off[kLdsShade]);
        S.node_aabbs = (const double*)(lb + S0.lds_off[kLdsNodeAabbs]);
        S.lights = (const LightRec*)(lb + S0.lds_off[kLdsLights]);
        S.planes = (const int32_t*)(lb + S0.lds_off[kLdsPlanes]);
        S.shadow_planes = (const int32_t*)(lb + S0.lds_off[kLdsShadowPlanes]);
    }
    __shared__ uint32_t lds_stack[kLdsStack * kBlock];
    __shared__ uint32_t block_next; // grab == 0: next entry of this workgroup's tile list
    if (grab == 0u) { // workgroup-uniform
        if (threadIdx.x == 0) block_next = 0u;
        __syncthreads();
    }
    // Counters are double-buffered: this launch clears the set the NEXT launch / frame will use (nothing
    // else touches it while this kernel runs), which removes every hipMemsetAsync from the frame.
    if (blockIdx.x == 0) {
        if (threadIdx.x < kNumCounts) zero_counts[threadIdx.x] = 0u;
        if (zero_ctr && threadIdx.x < sizeof(DeviceCounters) / 4) ((uint32_t*)zero_ctr)[threadIdx.x] = 0u;
    }
    constexpr bool kPark = (FEAT & kFeatPark) != 0;
    __shared__ uint32_t lds_park[kPark ? park_slots(FEAT) * kBlock : 4];
    Stack st;
    st.lds = (lds_u32*)(lds_stack + threadIdx.x);
    st.spill_stride = gridDim.x * kBlock;
    st.spill = spill ? (global_u32*)(spill + (size_t)blockIdx.x * kBlock + threadIdx.x) : nullptr;
    st.lds0 = Stack::addr((lds_u32*)lds_stack);
    st.park = (lds_u32*)(lds_park + threadIdx.x);
    st.init();
    // a launch that records its tile costs also records itself (DRender::cost_meta): its first wave brackets its lifetime with both clocks, every wave leaves its end tick
    // The instrumented kernel also measures the shader clock (DRender::cost_meta): the first wave of every 32nd workgroup brackets its lifetime with both clocks.  Not the
    // plain kernels: every wave doing so in a cost-recording launch (three atomics each on one line) cost the balls frame 0.062 -> 0.10 ms, and even the sampled form's code
    // 0.5 - 1 % of the analytic kernels' steady frames.
    __shared__ uint32_t clk_start[STATS ? 2 : 1];
    if (STATS && R.cost_meta && (blockIdx.x & 31u) == 0u && threadIdx.x == 0u) { clk_start[0] = (uint32_t)__builtin_readcyclecounter(); clk_start[1] = (uint32_t)__builtin_amdgcn_s_memrealtime(); }
    Cnt cnt; cnt.node = cnt.tri = cnt.prim = cnt.hit = cnt.tex = cnt.shadow = cnt.refl = cnt.refr = cnt.max_depth = cnt.max_chain_nodes = cnt.traced = cnt.elided = cnt.fetch = 0;
#ifdef NR_PHASE_TIMING
    cnt.cyc_node = cnt.cyc_leaf = cnt.cyc_other = cnt.cyc_tri = 0; cnt.wv_node = cnt.ln_node = cnt.wv_tri = cnt.ln_tri = 0; cnt.cyc_closest0 = cnt.cyc_closestN = cnt.cyc_shadow = 0; cnt.wv_uni = 0; cnt.inq_node = cnt.inq_tri = 0; for (int k_ = 0; k_ < 8; ++k_) cnt.cyc_x[k_] = 0;
    unsigned long long twave = __builtin_readcyclecounter();
#endif

    // DRender::m as it lies in the kernel-argument segment (R is the second argument, behind S0; both 8-byte aligned): generate_primary<PLAIN> reads it with vector loads
    static_assert(alignof(DScene) == 8 && alignof(DRender) == 8 && sizeof(DScene) % 8 == 0, "kernel-argument layout of k_primary");
    const double* m_mem = (const double*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(DScene) + offsetof(DRender, m));
    const uint32_t lane = threadIdx.x & 63u;
    // (one-light scenes with transparent nodes too — NR_PIXEL_SPLIT: a part is then 64 >> lsl PIXELS of the tile, every pixel's 2^lsl lanes tracing the same rays; what it
    // buys is that the deep, divergent chains through alpha-mapped layers of 8 pixels serialise in a wave instead of those of 64)
    constexpr bool kLightSplit = !STATS && (FEAT & kFeatMesh) && !(FEAT & kFeatDouble) && ((FEAT & kFeatMultiSample) || (NR_PIXEL_SPLIT && (FEAT & kFeatAlphaShadow)));
#ifdef NR_DEBUG_TILE_COSTS
    const uint32_t dbg_t_entry = (uint32_t)__builtin_amdgcn_s_memrealtime();
    const unsigned long long dbg_c_entry = __builtin_readcyclecounter();
    uint32_t dbg_t_first = 0u, dbg_tiles = 0u, dbg_work_tiles = 0u, dbg_work_cycles = 0u, dbg_t_last_end = 0u, dbg_row_ticks = 0u, dbg_rows = 0u;
    uint32_t dbg_slow_row = 0u, dbg_work_ticks = 0u, dbg_miss_ticks = 0u, dbg_miss_tiles = 0u, dbg_longest_ticks = 0u; // second record (nrays_debug_wave_times2): 10 ns ticks in work tiles / in tiles that traced nothing
#endif
    // Sample-major lane mapping of anti-aliased frames (ray_per_pixel >= 2): 2^lane_log2 lanes share ONE pixel and trace
    // its samples side by side, so a wave covers 64 >> lane_log2 pixels (8x4, 4x4, 4x2, 2x2, 2x1, 1x1) instead of 8x8 and
    // its 64 rays start within a few pixels of each other — the coherence that thin geometry (hair: 16 % SIMD efficiency
    // in the node loops at one lane per pixel) otherwise lacks.  The samples of a pixel are then summed in sample order
    // (tot_c = tot_c + trace(ray), scene.rs:72-91) by an in-wave ordered reduction, so the frame is bit-identical to the
    // pixel-major one.  One lane per pixel (lane_log2 = 0): wave tiles are 8x8, four per 16x16 block.
    const uint32_t lane_log2 = PLAIN ? 0u : R.lane_log2;
    const uint32_t nwt = lane_log2 ? R.win_nx * R.win_ny : R.win_nx * R.win_ny * 4u; // wave tiles of the window (device_types.h: DRender::win_*)

    // The pixels outside the window cannot be reached by the scene (screen_bounds()): every sample returns the background, the
    // pixel holds their f32 sum in sample order (padding rows of the last band: zero).  Rows are dealt round-robin to the
    // workgroups; the window's own pixels are written by the tiles below, so no pixel has two writers.  Mesh kernels fill their
    // rows in a prologue, all threads of the workgroup on one row; with workgroup lists (grab == 0) the rows are the TAIL of
    // the list — a quarter of a row per entry — so they are written by whichever waves run out of tiles first, not by the wave that
    // still sits on the frame's longest tile.
    const bool fill_rows = R.win_nx < tiles_x || R.win_ny < tiles_y;
    auto fill_row = [&](uint32_t rl, uint32_t t0, uint32_t tstep) { // threads t0, t0 + tstep, ... of the row's W * 3 floats
        fill_background_row(S.background[0], S.background[1], S.background[2], R.spp, out, R.width, R.height, R.band_rows, R.band_owner, R.band_owners,
                            R.win_x0, R.win_nx, R.win_y0, R.win_ny, lane_log2, rl, t0, tstep);
    };
    auto fill_row_inline = [&](uint32_t rl, uint32_t t0, uint32_t tstep) { // the same, without a call (and its wait for the stores)
        fill_background_row_body(S.background[0], S.background[1], S.background[2], R.spp, out, R.width, R.height, R.band_rows, R.band_owner, R.band_owners,
                                 R.win_x0, R.win_nx, R.win_y0, R.win_ny, lane_log2, rl, t0, tstep);
    };
    if (fill_rows && grab != 0u)
        for (uint32_t rl = blockIdx.x; rl < R.rows_local; rl += gridDim.x) fill_row(rl, threadIdx.x, kBlock);

    // grab == 0 (cheap analytic scenes, ~1 us tiles): the wave tiles are dealt round-robin to the workgroups and
    // the four waves of a workgroup pull from their list through an LDS counter — list scheduling inside the
    // workgroup (a wave that drew a deep reflection chain does not hold back its siblings' share) without a
    // single HBM atomic; device-scope atomics on one address retire at ~9 M/s on this part, far too slow
    // for 32k tiles per 100 us frame.  grab >= 1: XCD-aware dynamic dequeue from HBM counters (mesh scenes).
    // wave-uniform by construction: read through an SGPR so that the tile arithmetic stays on the scalar unit
    const uint32_t total_waves = gridDim.x * (kBlock / 64);
    const uint32_t my_wave = blockIdx.x * (kBlock / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t victim = xcc_id();
    const uint32_t g = grab ? grab : 1u;
    const uint32_t per = (((nwt + 7u) / 8u) + g - 1u) / g * g; // wave tiles per XCD range
    const bool prefetch = grab > 1u;
    // The FIRST entry of a wave comes without an atomic: 2048 - 4096 waves incrementing eight counters at the start of the launch
    // retire one every ~100 ns per counter — the last wave of an XCD reached its first tile after 25 us (sponza) / 51 us (hairball).
    // Workgroup b statically owns entry (b / 8) * 4 + wave of list b mod 8 (dispatch deals the workgroups round-robin to the XCDs, so
    // that is the list of its own XCD; nothing depends on it); the counters then count from the number of static owners of a list.
    auto static_owners = [&](uint32_t x) -> uint32_t { return grab == 1u ? (uint32_t)(kBlock / 64) * ((gridDim.x + 7u - x) >> 3) : 0u; };
    bool static_first = NR_STATIC_FIRST && grab == 1u;
    uint32_t pending = (grab && !static_first) ? issue_grab(work_counters, victim, grab) : 0u;
    for (;;) {
      NR_TIC(tdq);
      uint32_t first, last;
      if (grab == 0u) {
          uint32_t k = 0;
          if (lane == 0u) k = atomicAdd(&block_next, 1u); // LDS: ~100 cycles, no prefetch needed
          const uint32_t kk = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
          // this workgroup's list: (lead workgroups only) its share of the `heavy` most expensive entries of the order, then
          // its share of the other entries, then its share of the rows outside the window (a quarter of a row per entry)
          const uint32_t G = gridDim.x, b = blockIdx.x, Gh = R.tile_order ? R.lead_wgs : 0u;
          const uint32_t heavy = Gh ? (nwt < R.lead_entries ? nwt : R.lead_entries) : 0u;
          const uint32_t nhb = (b < Gh && heavy > b) ? (heavy - b + Gh - 1u) / Gh : 0u;
          const uint32_t ncb = nwt - heavy > b ? (nwt - heavy - b + G - 1u) / G : 0u;
          if (kk < nhb) first = b + Gh * kk;
          else if (kk - nhb < ncb) first = heavy + (kk - nhb) * G + b;
          else { // the tiles are taken: rows
              if (!fill_rows) break;
              const uint32_t part = kk - nhb - ncb;
              const uint32_t rl = (part >> 2) * G + b;
              if (rl >= R.rows_local) break;
#ifdef NR_DEBUG_TILE_COSTS
              { const uint32_t tr0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
                fill_row_inline(rl, (part & 3u) * 64u + lane, kBlock);
                const uint32_t dtr_ = (uint32_t)__builtin_amdgcn_s_memrealtime() - tr0;
                if (dtr_ >= (dbg_slow_row & 0xfffu)) dbg_slow_row = (rl << 14) | ((part & 3u) << 12) | (dtr_ > 0xfffu ? 0xfffu : dtr_);
                dbg_row_ticks += dtr_; dbg_rows++; }
#else
              fill_row_inline(rl, (part & 3u) * 64u + lane, kBlock);
#endif
              continue;
          }
          // with the costs of an earlier frame of this camera: entry e of the descending-cost order instead of wave tile e,
          // so the few hundred expensive tiles of a frame (deep reflection chains) are dealt one to a WAVE and start first
          // instead of piling up in the workgroups whose columns cross them
          if (R.tile_order) first = R.tile_order[first];
          last = first + 1u;
      } else {
          // XCD x owns a list of wave tiles: without history the contiguous range [x * per, (x + 1) * per) of the
          // image; with the previous frame's costs the entries x, x + 8, x + 16, ... of the descending-cost order
          // (longest-processing-time-first: the deep alpha / reflection chains start first, the frame ends on
          // cheap tiles).
          const bool ordered = R.tile_order != nullptr;
          uint32_t k;
          if (static_first) { victim = blockIdx.x & 7u; k = (blockIdx.x >> 3) * (uint32_t)(kBlock / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
          else k = (uint32_t)__builtin_amdgcn_readfirstlane((int)pending) + (NR_STATIC_FIRST ? static_owners(victim) : 0u);
          auto list_len = [&](uint32_t x) -> uint32_t {
              if (ordered) return R.order_len ? R.order_len[x] : (nwt > x ? (nwt - x + 7u) / 8u : 0u);
              return x * per < nwt ? (nwt - x * per < per ? nwt - x * per : per) : 0u;
          };
          uint32_t len = list_len(victim);
          if (k >= len) { // this XCD's list is exhausted: steal from the next non-empty one
              bool found = false;
#if NR_PEEK_STEAL
              // one look at all eight counters (lanes 0..7, one round trip): lists that are exhausted are not even tried — the
              // failed atomics of the waves that run dry at the end of a frame delayed the dequeues of the waves still working
              // (not in anti-aliased frames: millions of small tiles, the lists are image bands that run dry one after the other and
              // most steals succeed — the look costs them 0.7 %)
              uint32_t cnt8 = 0u;
              if (lane < 8u && lane_log2 == 0u) cnt8 = __hip_atomic_load(&work_counters[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
              for (uint32_t tries = 0; tries < 7u && !found; ++tries) {
                  victim = (victim + 1u) & 7u;
                  len = list_len(victim);
                  if (len == 0u) continue;
#if NR_PEEK_STEAL
                  if ((uint32_t)__builtin_amdgcn_readlane((int)cnt8, (int)victim) + (NR_STATIC_FIRST ? static_owners(victim) : 0u) >= len) continue;
#endif
                  k = (uint32_t)__builtin_amdgcn_readfirstlane((int)issue_grab(work_counters, victim, grab)) + (NR_STATIC_FIRST ? static_owners(victim) : 0u);
                  if (k < len) found = true;
              }
              if (!found) break; // wave-uniform
          }
          if (ordered) { first = R.tile_order[k * 8u + victim]; last = first + 1u; }
          else { first = victim * per + k; last = k + grab < len ? first + grab : victim * per + len; }
#if NR_TILE_PRIO
          // cost-ordered lists: the earlier an entry sits in its list the longer its tile — the frame cannot end before the longest chains do,
          // so their waves issue ahead of the SIMD's other wave (s_setprio only orders the waves of one SIMD; results do not depend on it)
          if (ordered) { if (k < NR_TILE_PRIO) __builtin_amdgcn_s_setprio(3); else if (k < 3u * NR_TILE_PRIO) __builtin_amdgcn_s_setprio(2); else if (k < 8u * NR_TILE_PRIO) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#endif
          if (static_first) { static_first = false; victim = xcc_id(); } // from here on: this XCD's counter
          if (prefetch) pending = issue_grab(work_counters, victim, grab); // issued now, consumed after the tiles below
      }
      NR_TOC(cyc_x[0], tdq);
      for (uint32_t ent = first; ent < last; ++ent) {
        // (an entry of a cost-ordered list may stand for one part of a light-parallel tile: DRender::light_lsl)
        const uint32_t wt = kLightSplit ? (ent & kEntryTileMask) : ent;
        const uint32_t lsl = (kLightSplit && (ent & kEntrySplit)) ? R.light_lsl : 0u, part = kLightSplit ? ((ent >> 28) & 7u) : 0u; // wave-uniform
#ifdef NR_DEBUG_TILE_COSTS
        const unsigned long long tile_t0 = __builtin_readcyclecounter();
        bool dbg_worked = false;
        const uint32_t dbg_tile_r0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
        if (dbg_tiles++ == 0u) dbg_t_first = dbg_tile_r0;
#else
        const unsigned long long tile_t0 = R.tile_cost ? __builtin_readcyclecounter() : 0ULL;
#endif
        uint32_t i, rl; // column, local (compact) row
        uint32_t q = 0u; // which of the pixel's side-by-side samples this lane traces
        if (lane_log2 == 0u) {
            uint32_t tile = wt >> 2, sub = wt & 3u;
            uint32_t tx = R.win_x0 + tile % R.win_nx, ty = R.win_y0 + tile / R.win_nx;
            const uint32_t p = lane >> lsl;                               // pixel of this lane inside the entry's part of the tile
            uint32_t lx = ((sub & 1u) << 3) | (p & 7u), ly = ((sub >> 1) << 3) | ((p >> 3) + part * (8u >> lsl));
            i = tx * kTile + lx; rl = ty * kTile + ly;
        } else {
            const uint32_t bwl = (7u - lane_log2) >> 1, bhl = (6u - lane_log2) >> 1; // the wave's pixel block is 2^bwl x 2^bhl
            const uint32_t p = lane >> lane_log2;
            q = lane & ((1u << lane_log2) - 1u);
            uint32_t tx = R.win_x0 + wt % R.win_nx, ty = R.win_y0 + wt / R.win_nx;
            i = (tx << bwl) + (p & ((1u << bwl) - 1u)); rl = (ty << bhl) + (p >> bwl);
        }
        // local row -> global row (framebuffer bands dealt round-robin to owners)
        uint32_t j = rl;
        if (R.band_rows != 0 && R.band_owners > 1) {
            uint32_t lb = rl / R.band_rows;
            j = (lb * R.band_owners + R.band_owner) * R.band_rows + (rl % R.band_rows);
        }
        bool active = i < R.width && rl < R.rows_local && j < R.height;
        uint32_t pix = rl * R.width + i;
        // no pixel of this wave tile inside the screen bounds of the scene: every sample is a miss (Scene::trace returns the
        // background, scene.rs:157-161) and no ray has to be generated to know it
        const bool tile_misses = __ballot(active && (int32_t)i >= R.cull_i0 && (int32_t)i <= R.cull_i1 && (int32_t)j >= R.cull_j0 && (int32_t)j <= R.cull_j1) == 0ULL;
        // tot_c = tot_c + trace(ray) sample after sample (scene.rs:72-91): a later sample batch continues the running sum
        // of the earlier ones, so the f32 summation order — and with it the frame — does not depend on the batching
        f3 tot = F3(0.0f, 0.0f, 0.0f);
        if (!PLAIN && !R.first_batch && active) { const float* o = out + (size_t)pix * 3; tot = F3(o[0], o[1], o[2]); }
        const uint32_t s_begin = PLAIN ? 0u : R.sample_begin, s_end = PLAIN ? 1u : R.sample_end;
        for (uint32_t g = s_begin; g < s_end; g += 1u << lane_log2) {
            const uint32_t s = g + q;
            const bool sample_active = active && s < s_end;
            RayState ray;
            unsigned node_before = cnt.node;
            f3 c;
            bool wave_may_hit = false; // wave-uniform
            NR_TIC(trg);
            if (!tile_misses) {
                // lanes outside the frame (ragged right edge, padding rows of a band) still execute the ray generation: keep
                // their table reads inside the tables
                generate_primary<PLAIN>(R, i < R.width ? i : R.width - 1u, j < R.height ? j : R.height - 1u, s, pix, ray, PLAIN ? m_mem : nullptr);
                wave_may_hit = __ballot(sample_active && primary_may_hit(S, ray.o, ray.d)) != 0ULL;
            }
            NR_TOC(cyc_x[1], trg);
            if (!wave_may_hit) {
                // no ray of this wave tile gets past the root of the BVT: Scene::trace returns the background for
                // all of them (scene.rs:157-161), without entering the trace loop
                c = F3(S.background[0], S.background[1], S.background[2]);
                // (tiles the screen bounds decide run no box test at all: node_tests excludes them)
                if (STATS && !tile_misses && sample_active && S.closest_root >= 0) cnt.node += root_children(S);
            } else {
                // instrumented renders only: a uniform counter in the tile loop of the plain kernels costs 25 us of the 52 us balls
                // frame (profiles/r03 notes)
                if (STATS && sample_active) cnt.traced++;
#ifdef NR_DEBUG_TILE_COSTS
                dbg_worked = true;
#endif
                c = trace_chain<STATS, FEAT>(S, st, sample_active, ray, 0u, R.max_depth, qo, cnt, !PLAIN && R.use_rng != 0u, lsl);
            }
            if (STATS) { unsigned dn = cnt.node - node_before; if (dn > cnt.max_chain_nodes) cnt.max_chain_nodes = dn; }
            if (lane_log2 == 0u) { tot.x = tot.x + c.x; tot.y = tot.y + c.y; tot.z = tot.z + c.z; }
            else { // the pixel's lanes hold samples g .. g + 2^lane_log2 - 1: add them in sample order (every lane of the group keeps the same sum)
                const uint32_t base = lane & ~((1u << lane_log2) - 1u);
                for (uint32_t k = 0; k < (1u << lane_log2) && g + k < s_end; ++k) { // wave-uniform bounds
                    const float cx = __shfl(c.x, (int)(base + k)), cy = __shfl(c.y, (int)(base + k)), cz = __shfl(c.z, (int)(base + k));
                    tot.x = tot.x + cx; tot.y = tot.y + cy; tot.z = tot.z + cz;
                }
            }
        }
        if (active && (q != 0u || (lane & ((1u << lsl) - 1u)) != 0u)) { /* the group's first lane writes the pixel */ }
        else if (active) {
            float* o = out + (size_t)pix * 3;
#if NR_NT_STORES
            __builtin_nontemporal_store(tot.x, o); __builtin_nontemporal_store(tot.y, o + 1); __builtin_nontemporal_store(tot.z, o + 2);
#else
            o[0] = tot.x; o[1] = tot.y; o[2] = tot.z;
#endif
        } else if (q == 0u && (lane & ((1u << lsl) - 1u)) == 0u && i < R.width && rl < R.rows_local && (PLAIN || R.first_batch)) { // padding rows of the last band
            float* o = out + (size_t)pix * 3;
            o[0] = 0.0f; o[1] = 0.0f; o[2] = 0.0f;
        }
#ifdef NR_DEBUG_TILE_COSTS
        if (dbg_worked) { dbg_work_tiles++; dbg_work_cycles += (uint32_t)((__builtin_readcyclecounter() - tile_t0) >> 4); dbg_t_last_end = (uint32_t)__builtin_amdgcn_s_memrealtime(); }
        { const uint32_t dt_ = (uint32_t)__builtin_amdgcn_s_memrealtime() - dbg_tile_r0;
          if (dbg_worked) { dbg_work_ticks += dt_; if (dt_ > dbg_longest_ticks) dbg_longest_ticks = dt_; } else { dbg_miss_ticks += dt_; dbg_miss_tiles++; } }
#endif
        if (R.tile_cost && lane == 0u) { // wave cycles spent on this tile, for the next frame's order
            // (a part's cycles stand for the tile's: x 2^lsl for a light-parallel part; x 3 for a part of a one-light tile — 8 of its pixels: measured, a whole foliage tile takes
            // 2 - 3 parts' time — so that a tile a first guess split without need, e.g. in a 4K frame, is whole again once its cost is known)
            unsigned long long dt = (__builtin_readcyclecounter() - tile_t0) >> 4;
            if (kLightSplit && lsl) dt = (FEAT & kFeatMultiSample) ? dt << lsl : dt * 3u;
            const uint32_t c = dt > (unsigned long long)kCostMask ? kCostMask : (uint32_t)dt;
            // a light-parallel tile: its most expensive part stands for all (the array is cleared before a frame that records into split entries) — the
            // first part alone is eight of the tile's pixels, and a tile priced by a cheap row stayed whole and late in the order: one rank of eight
            // of config 4 ran 1.82 ms for 1.25 (profiles/r05_rank_occupancy.log)
            // (kCostSplit marks the record of a tile that ran in parts: the next sort keeps it split down to half the threshold — a part x 3 can underestimate the whole tile, and a
            // tile that is split one frame and whole the next makes every second frame of a moving camera wait for an unsplit monster — and nrays_get_tile_costs knows the units dealt)
            if (kLightSplit && lsl) atomicMax(&R.tile_cost[wt], c | kCostSplit); else R.tile_cost[wt] = c;
        }
      }
      if (grab == 1u) pending = issue_grab(work_counters, victim, grab);
    }
#ifdef NR_PHASE_TIMING
    cnt.cyc_other = (unsigned)(__builtin_readcyclecounter() - twave);
#endif
#ifdef NR_DEBUG_TILE_COSTS
    if (R.wave_times && lane == 0u) {
        uint32_t* w = R.wave_times + 4u * (blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));
        uint32_t hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        w[0] = dbg_t_entry; w[1] = R.dbg_mode == 4u ? (dbg_row_ticks & 0xfffffu) | (dbg_rows << 20) : R.dbg_mode == 3u ? dbg_t_last_end : R.dbg_mode == 2u ? (uint32_t)((__builtin_readcyclecounter() - dbg_c_entry) >> 4) : R.dbg_mode ? (dbg_work_cycles & 0x03ffffffu) | (dbg_work_tiles << 26) : dbg_t_first; w[2] = (uint32_t)__builtin_amdgcn_s_memrealtime(); w[3] = dbg_tiles | (xcc_id() << 28) | ((hw & 0xffffu) << 12);
        uint32_t* w2 = R.wave_times + 4u * (uint32_t)kMaxGrid * (kBlock / 64) + 4u * (blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));
        w2[0] = dbg_work_ticks; w2[1] = R.dbg_mode == 5u ? dbg_slow_row : dbg_miss_ticks; w2[2] = dbg_row_ticks; w2[3] = (dbg_work_tiles & 0xffu) | ((dbg_miss_tiles & 0xffu) << 8) | ((dbg_rows & 0xffu) << 16) | ((dbg_longest_ticks >> 4) << 24);
    }
#endif
    if (STATS && R.cost_meta && (blockIdx.x & 31u) == 0u && threadIdx.x == 0u) { // lifetime in both clocks; sums over the sampled waves and over the handle's recording launches (never cleared: only the ratio is used)
        atomicAdd(&R.cost_meta[2], (unsigned long long)((uint32_t)__builtin_readcyclecounter() - clk_start[0]));
        atomicAdd(&R.cost_meta[3], (unsigned long long)((uint32_t)__builtin_amdgcn_s_memrealtime() - clk_start[STATS ? 1 : 0]));
    }
    flush_counters(ctr, cnt, STATS);
}

} // namespace nrays

// ---------------------------------------------------------------------------------------------------------------------
// The permutations of k_primary and the translation units that hold them.  X(group, STATS, FEAT, PLAIN, OCC); FEAT is a
// sum of device_types.h: Features bits.  primary_inst.hip is compiled once per group (-DNR_PRIMARY_GROUP=g) and defines
// launch_primary_group<g>(); nrays_hip.hip: launch_primary() names the permutation a frame wants and asks the groups in
// turn.  A tuning build (-DNR_ONLY=33 or -DNR_ONLY=6,70,...: the FEAT codes an A/B run touches, tools/build_variant.sh)
// compiles only those permutations + the two full kernels; every other frame then renders with the full kernel — same
// pixels, slower.
#define NR_PRIMARY_PERMUTATIONS(X)                                                                                          \
    /* group 0: the full-featured kernels (instrumented frames, double-branching scenes, the fall-back) */                    \
    X(0, true, 31, false, 0) X(0, false, 31, false, 0) X(0, false, 15, false, 0) X(0, false, 3, false, 0) X(0, false, 19, false, 0) \
    /* group 1: analytic scenes, plain frames (tables, no RNG keys, one sample per pixel); 32 = records in LDS */            \
    X(1, false, 33, true, 0) X(1, false, 37, true, 0) X(1, false, 49, true, 0) X(1, false, 53, true, 0)                       \
    X(1, false, 1, true, 0) X(1, false, 5, true, 0) X(1, false, 17, true, 0) X(1, false, 21, true, 0)                         \
    /* group 2: analytic scenes, general frames */                                                                           \
    X(2, false, 33, false, 0) X(2, false, 37, false, 0) X(2, false, 49, false, 0) X(2, false, 53, false, 0)                   \
    X(2, false, 1, false, 0) X(2, false, 5, false, 0) X(2, false, 17, false, 0) X(2, false, 21, false, 0)                     \
    /* group 3: mesh scenes */                                                                                               \
    X(3, false, 2, true, 0) X(3, false, 6, true, 0) X(3, false, 18, true, 0) X(3, false, 22, true, 0)                         \
    X(3, false, 2, false, 0) X(3, false, 6, false, 0) X(3, false, 18, false, 0) X(3, false, 22, false, 0)                     \
    X(3, false, 7, false, 0) X(3, false, 23, false, 0)                                                                       \
    /* group 4: mesh scenes whose BLASes all sit in world space (64 = kFeatNoXform) */                                       \
    X(4, false, 66, true, 0) X(4, false, 70, true, 0) X(4, false, 82, true, 0) X(4, false, 86, true, 0)                       \
    X(4, false, 66, false, 0) X(4, false, 70, false, 0) X(4, false, 82, false, 0) X(4, false, 86, false, 0)                   \
    /* group 5: the three-wave builds of the alpha-shadow mesh permutations (128 = kFeatPark) */                             \
    X(5, false, 134, true, 3) X(5, false, 134, false, 3) X(5, false, 6, true, 3) X(5, false, 6, false, 3)                     \
    X(5, false, 150, true, 3) X(5, false, 150, false, 3) X(5, false, 22, true, 3) X(5, false, 22, false, 3)                   \
    X(5, false, 7, false, 3) X(5, false, 23, false, 3)                                                                       \
    /* group 6: ... of the untransformed ones */                                                                             \
    X(6, false, 198, true, 3) X(6, false, 198, false, 3) X(6, false, 70, true, 3) X(6, false, 70, false, 3)                   \
    X(6, false, 214, true, 3) X(6, false, 214, false, 3) X(6, false, 86, true, 3) X(6, false, 86, false, 3)
constexpr int kPrimaryGroups = 7;

namespace nrays {

struct PrimaryLaunch { // the arguments of one k_primary launch
    uint32_t grid; hipStream_t stream; const DScene* d; const DRender* R; const QueueOut* qo; float* out; DeviceCounters* ctr; uint32_t* spill;
    uint32_t tiles_x, tiles_y; uint32_t* work; uint32_t grab; uint32_t* zero_counts; DeviceCounters* zero_ctr;
};
// true = this group holds the permutation and has launched it (hipGetLastError() tells how that went)
bool launch_primary_group0(const PrimaryLaunch& a, bool stats, int feat, bool plain, int occ);
bool launch_primary_group1(const PrimaryLaunch& a, bool stats, int feat, bool plain, int occ);
bool launch_primary_group2(const PrimaryLaunch& a, bool stats, int feat, bool plain, int occ);
bool launch_primary_group3(const PrimaryLaunch& a, bool stats, int feat, bool plain, int occ);
bool launch_primary_group4(const PrimaryLaunch& a, bool stats, int feat, bool plain, int occ);
bool launch_primary_group5(const PrimaryLaunch& a, bool stats, int feat, bool plain, int occ);
bool launch_primary_group6(const PrimaryLaunch& a, bool stats, int feat, bool plain, int occ);

} // namespace nrays
