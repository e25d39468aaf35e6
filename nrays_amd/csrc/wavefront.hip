// wavefront.hip — the trace loop of scene::render (src/scene.rs:29-116) as STAGES over compacted ray queues in HBM, for
// scenes of TriMesh nodes: the north star's "wavefront ballot / prefix-sum ray compaction" on the main path.
//
// The megakernel (nrays_hip.hip: k_primary) keeps a pixel's whole chain — closest hit, shadow rays, Phong, continuation —
// in the registers of one lane; a wave then carries the union of every code path (207 - 249 VGPRs: two waves per SIMD),
// lanes whose ray missed or whose chain ended idle until the wave tile ends, and a frame cannot end before its deepest
// tile does.  Here one kernel does ONE thing to 64 consecutive rays of a queue:
//
//   k_wf_primary    raygen + closest hit of the primary rays (8x8 wave tiles / the samples of a pixel side by side, as in
//                   k_primary); misses write the background, hits are appended — wave ballot -> popcount prefix -> the
//                   wave's current 1024-slot block of the queue (one atomic per block) — as (ray, hit) records: generation 0.
//   k_wf_closest    generations >= 1: closest hit (ClosestRayTOICostFn, src/scene.rs:262-283) of the queued rays.
//   k_wf_shadow     scenes with several lights: one (chunk, light) item per wave — the shadow query
//                   (TransparentShadowsRayTOICostFn, src/scene.rs:285-339) of 64 hits towards ONE light; result per
//                   (light, slot) in HBM.  Single-light scenes trace their one shadow ray inside k_wf_shade.
//   k_wf_shade      hit reconstruction, Phong (src/phong_material.rs:72-151) from the shadow results, weight algebra of
//                   Scene::trace (src/scene.rs:163-252), the chain's running sum, and the continuation ray appended to the
//                   next generation's queue (again ballot -> prefix -> own block).
//   k_wf_resolve    anti-aliased frames: the samples of a pixel summed in sample order (src/scene.rs:72-91).
//
// Work is dealt as contiguous RANGES, one per wave (adjacent wave tiles / adjacent chunks: the rays a wave compacts into one
// chunk come from neighbouring pixels and walk the same nodes), each behind its own counter; a wave that finishes its range
// steals single items from the ranges that still hold some (WfClaim).
// Every kernel holds one traversal's state at most, runs at four waves per SIMD, and its lanes are dense.  Arithmetic and
// summation orders are those of the megakernel: frames are bit-identical (tests/test_wavefront_gpu.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/nrays_abi.h"
#include "device_types.h"
#include "scene_build.h"
#include "scene_handle.h"
#include "tile_device.h"
#include "trace_device.h"
#include "stream_device.h"
#include "wavefront.h"

#ifndef NR_WF_OCC
#define NR_WF_OCC 4 // waves per SIMD of the stage kernels (second __launch_bounds__ argument: 128 VGPRs)
#endif
#ifndef NR_WF_OCC_RF
#define NR_WF_OCC_RF 3 // waves per SIMD of the stages with lane refill (their per-lane stream state costs registers)
#endif
#ifndef NR_WF_OCC_SHADE
#define NR_WF_OCC_SHADE NR_WF_OCC
#endif

namespace nrays {

constexpr uint32_t kWfMiss = 0xffffffffu; // WfQueue::hit.z of a ray that left the scene
constexpr int kWfMaxSegs = kMaxGrid * (kBlock / 64);
constexpr int kTile = 16;

// One generation of rays, SoA of 16-byte vectors over `slots` (every lane of a wave moves whole dwordx4s, consecutive
// lanes consecutive vectors).  The slots are handed out in blocks of kWfBlock: a producer wave fills its current block
// (fill[block] = rays in it) and takes the next one from `nblocks`; a consumer's 64-ray chunk c is the slots
// [64 c, 64 c + 64) with fill[c / 16] - 64 (c % 16) of them (clamped to 0..64) valid.
constexpr uint32_t kWfBlock = 1024u, kWfChunksPerBlock = kWfBlock / 64u;
struct WfQueue {
    double2* o01;        // (o.x, o.y)
    double2* o2d0;       // (o.z, d.x)
    double2* d12;        // (d.y, d.z)
    uint4* re;           // RayWithEnergy::refr (f64), energy, weight (product of the blend factors down to this ray)
    uint4* kp;           // RNG path key (u64), path (index of the chain's running sum), unused
    uint4* hit;          // closest hit: toi (f64), instance, triangle slot; instance = kWfMiss: none
    uint32_t* fill;      // rays in each block
    uint32_t* nblocks;   // blocks handed out
    uint32_t slots;
};

// A producer wave's place in a queue: ballot -> popcount prefix -> slots of the current block, a new block when it is full.
struct WfOut {
    uint32_t block, used; // wave-uniform; block = ~0u: none yet
    NR_DEV void init() { block = ~0u; used = 0u; }
    // slot of this lane's ray (`has` lanes; ~0u: the queue is full), must be reached by the whole wave
    NR_DEV uint32_t append(const WfQueue& q, bool has) {
        const unsigned long long m = __ballot(has);
        const uint32_t n = (uint32_t)__popcll(m);
        if (n == 0u) return ~0u; // wave-uniform
        const uint32_t lane = __lane_id(), pre = (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL));
        const uint32_t room = block == ~0u ? 0u : kWfBlock - used;
        uint32_t at = block * kWfBlock + used + pre;
        if (n > room) { // wave-uniform: the rays beyond `room` open a new block
            uint32_t nb = 0u;
            if (lane == 0u) { if (block != ~0u) q.fill[block] = kWfBlock; nb = atomicAdd(q.nblocks, 1u); }
            nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
            if (pre >= room) at = nb * kWfBlock + (pre - room);
            block = nb; used = n - room;
        } else used += n;
        return at < q.slots ? at : ~0u;
    }
    NR_DEV void finish(const WfQueue& q) { if (block != ~0u && __lane_id() == 0u) q.fill[block] = used; }
};
NR_DEV uint32_t wf_chunk_valid(const WfQueue& q, uint32_t c) {
    const uint32_t f = q.fill[c / kWfChunksPerBlock], b = (c % kWfChunksPerBlock) * 64u;
    return f > b ? (f - b < 64u ? f - b : 64u) : 0u;
}

// Items [0, n) as one contiguous range per wave of the grid, each behind its own counter: a wave claims the items of its range
// one by one (the claim of the next one is in flight while it works) and then steals from the ranges that still hold some —
// one look at 64 counters, the fullest range wins.  next[]: zero at launch (the previous kernel of the stream clears it).
struct WfClaim {
    uint32_t* next;
    uint32_t n, per, nranges, victim, pending, seed;
    NR_DEV uint32_t range_len(uint32_t r) const { const uint32_t b = r * per; return b < n ? (n - b < per ? n - b : per) : 0u; }
    NR_DEV uint32_t issue(uint32_t r) const { uint32_t k = 0u; if (__lane_id() == 0u) k = atomicAdd(&next[r], 1u); return k; }
    NR_DEV void init(uint32_t* next_, uint32_t n_, uint32_t my_wave, uint32_t total_waves) {
        next = next_; n = n_; nranges = total_waves; per = (n_ + total_waves - 1u) / total_waves; victim = my_wave; seed = my_wave + 1u;
        pending = n_ ? issue(victim) : 0u;
    }
    NR_DEV bool get(uint32_t& item) { // wave-uniform
        if (n == 0u) return false;
        for (;;) {
            const uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)pending);
            if (k < range_len(victim)) { item = victim * per + k; pending = issue(victim); return true; }
            bool found = false;
            const uint32_t lane = __lane_id();
            for (uint32_t round = 0u; round * 64u < nranges && !found; ++round) {
                const uint32_t idx = round * 64u + lane;
                uint32_t cand = seed + idx; cand = cand >= nranges ? cand % nranges : cand;
                uint32_t rem = 0u;
                if (idx < nranges) { const uint32_t c = __hip_atomic_load(&next[cand], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), l = range_len(cand); rem = c < l ? l - c : 0u; }
                uint32_t best = ((rem > 0xffffu ? 0xffffu : rem) << 6) | lane;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)best, off); best = o > best ? o : best; }
                best = (uint32_t)__builtin_amdgcn_readfirstlane((int)best);
                if ((best >> 6) != 0u) { victim = (uint32_t)__builtin_amdgcn_readlane((int)cand, (int)(best & 63u)); found = true; }
            }
            if (!found) return false;
            pending = issue(victim);
        }
    }
};
NR_DEV void wf_clear_next(uint32_t* clear_next, uint32_t n) { // the claim counters of the NEXT kernel of the stream
    if (blockIdx.x == 0) for (uint32_t k = threadIdx.x; k < n; k += kBlock) clear_next[k] = 0u;
}

NR_DEV void wf_store_ray(const WfQueue& q, uint32_t i, const RayState& r, uint32_t path) {
    q.o01[i] = make_double2(r.o.x, r.o.y); q.o2d0[i] = make_double2(r.o.z, r.d.x); q.d12[i] = make_double2(r.d.y, r.d.z);
    const unsigned long long rb = (unsigned long long)__double_as_longlong(r.refr);
    q.re[i] = make_uint4((uint32_t)rb, (uint32_t)(rb >> 32), __float_as_uint(r.energy), __float_as_uint(r.weight));
    q.kp[i] = make_uint4((uint32_t)r.key, (uint32_t)(r.key >> 32), path, 0u);
}
NR_DEV void wf_load_od(const WfQueue& q, uint32_t i, d3& o, d3& d) {
    const double2 a = q.o01[i], b = q.o2d0[i], c = q.d12[i];
    o = D3(a.x, a.y, b.x); d = D3(b.y, c.x, c.y);
}
NR_DEV void wf_load_ray(const WfQueue& q, uint32_t i, RayState& r, uint32_t& path) {
    wf_load_od(q, i, r.o, r.d);
    const uint4 e = q.re[i], k = q.kp[i];
    r.refr = __longlong_as_double((long long)(((unsigned long long)e.y << 32) | e.x));
    r.energy = __uint_as_float(e.z); r.weight = __uint_as_float(e.w);
    r.key = ((unsigned long long)k.y << 32) | k.x; path = k.z; r.pixel = k.z;
}
NR_DEV void wf_store_hit(const WfQueue& q, uint32_t i, bool any, const Hit& h) {
    const unsigned long long tb = (unsigned long long)__double_as_longlong(h.t);
    q.hit[i] = make_uint4((uint32_t)tb, (uint32_t)(tb >> 32), any ? h.inst : kWfMiss, h.prim);
}
NR_DEV bool wf_load_hit(const WfQueue& q, uint32_t i, Hit& h) {
    const uint4 v = q.hit[i];
    h.t = __longlong_as_double((long long)(((unsigned long long)v.y << 32) | v.x)); h.inst = v.z; h.prim = v.w;
    return v.z != kWfMiss;
}

// Scene::trace's closest-hit query with the deferred exact gates, exactly as shade_hit runs it: ungated traversal, the
// winner checked against the reference's AABB gates, fully gated repeat for knife-edge rays.
template <int FEAT>
NR_DEV bool wf_closest(const DScene& S, Stack& st, d3 o, d3 d, Hit& hit, Cnt& cnt) {
    static_assert((FEAT & kFeatMesh) != 0, "the staged path renders scenes with TriMesh nodes");
    f3 nofilter = F3(1.0f, 1.0f, 1.0f);
    Isect is; uint32_t node_id;
    bool gated = false;
    for (;;) {
        if (!traverse<false, false, FEAT>(S, st, o, d, kDblMax, hit, nofilter, cnt, gated, &is)) return false;
        if (resolve_hit<false, FEAT, true>(S, o, d, hit, is, node_id) || gated) return true;
        gated = true;
    }
}
template <int FEAT>
NR_DEV uint32_t wf_hit_node(const DScene& S, const Hit& h) {
    if (FEAT & kFeatAnalytic) {
        const Instance& in = S.instances[h.inst];
        if (in.kind != NRAYS_SHAPE_TRIMESH) return (uint32_t)in.node_id;
    }
    return S.tris[h.prim].node_id;
}

struct WfStackInit {
    NR_DEV static void make(Stack& st, uint32_t* lds_stack, uint32_t* spill) {
        st.lds = (lds_u32*)(lds_stack + threadIdx.x);
        st.spill_stride = gridDim.x * kBlock;
        st.spill = spill ? (global_u32*)(spill + (size_t)blockIdx.x * kBlock + threadIdx.x) : nullptr;
        st.lds0 = Stack::addr((lds_u32*)lds_stack);
        st.park = nullptr;
        st.init();
    }
};
NR_DEV void wf_cnt_init(Cnt& cnt) {
    cnt.node = cnt.tri = cnt.prim = cnt.hit = cnt.tex = cnt.shadow = cnt.refl = cnt.refr = cnt.max_depth = cnt.max_chain_nodes = cnt.traced = cnt.elided = cnt.fetch = 0;
#ifdef NR_PHASE_TIMING
    cnt.cyc_node = cnt.cyc_leaf = cnt.cyc_other = cnt.cyc_tri = 0; cnt.wv_node = cnt.ln_node = cnt.wv_tri = cnt.ln_tri = 0; cnt.cyc_closest0 = cnt.cyc_closestN = cnt.cyc_shadow = 0; cnt.wv_uni = 0; cnt.inq_node = cnt.inq_tri = 0; for (int k_ = 0; k_ < 8; ++k_) cnt.cyc_x[k_] = 0;
#endif
}
// ray classes of the frame (NraysStats): wave-level reduction, one atomic per wave and class
NR_DEV void wf_flush(DeviceCounters* ctr, const Cnt& c) {
    unsigned sh = c.shadow, rl = c.refl, rf = c.refr, md = c.max_depth;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sh += __shfl_down(sh, off); rl += __shfl_down(rl, off); rf += __shfl_down(rf, off);
        const unsigned om = __shfl_down(md, off); md = om > md ? om : md;
    }
    if (__lane_id() == 0) {
        if (sh) atomicAdd(&ctr->rays_shadow, (unsigned long long)sh);
        if (rl) atomicAdd(&ctr->rays_reflection, (unsigned long long)rl);
        if (rf) atomicAdd(&ctr->rays_refraction, (unsigned long long)rf);
        if (md) atomicMax(&ctr->max_depth, md);
    }
}

// Pixel of lane-group `p` of wave tile `wt` (k_primary's numbering: four consecutive 8x8 wave tiles form a 16x16 block at one
// lane per pixel; anti-aliased frames: the wave's 2^bwl x 2^bhl pixel block).  Returns column and local (compact) row.
NR_DEV void wf_tile_pixel(const DRender& R, uint32_t lane_log2, uint32_t wt, uint32_t p, uint32_t& i, uint32_t& rl) {
    if (lane_log2 == 0u) {
        const uint32_t tile = wt >> 2, sub = wt & 3u;
        const uint32_t tx = R.win_x0 + tile % R.win_nx, ty = R.win_y0 + tile / R.win_nx;
        i = tx * kTile + (((sub & 1u) << 3) | (p & 7u)); rl = ty * kTile + (((sub >> 1) << 3) | (p >> 3));
    } else {
        const uint32_t bwl = (7u - lane_log2) >> 1, bhl = (6u - lane_log2) >> 1;
        const uint32_t tx = R.win_x0 + wt % R.win_nx, ty = R.win_y0 + wt / R.win_nx;
        i = (tx << bwl) + (p & ((1u << bwl) - 1u)); rl = (ty << bhl) + (p >> bwl);
    }
}
NR_DEV uint32_t wf_global_row(const DRender& R, uint32_t rl) { // local row -> global row (framebuffer bands dealt round-robin to owners)
    if (R.band_rows != 0 && R.band_owners > 1) return ((rl / R.band_rows) * R.band_owners + R.band_owner) * R.band_rows + (rl % R.band_rows);
    return rl;
}

// ------------------------------------------------------------------------------------------------ stage: primary
// scene.rs:67-95 for the wave tiles [tile_begin, tile_end) of the window, EVERY sample of their pixels.  `acc`: where a chain's
// running sum lives — the frame itself at one sample per pixel (path = pixel), else the per-sample array k_wf_resolve folds
// (path = ((tile - tile_begin) * pixels per tile + pixel) * spp + sample).
template <int FEAT, bool PLAIN>
__global__ void __launch_bounds__(kBlock, NR_WF_OCC) k_wf_primary(DScene S, DRender R, WfQueue q, float* __restrict__ out, float* __restrict__ acc, DeviceCounters* ctr,
                                                                  uint32_t* spill, uint32_t tiles_x, uint32_t tiles_y, uint32_t tile_begin, uint32_t tile_end,
                                                                  uint32_t* claim_next, uint32_t* clear_next, uint32_t* zero_counts, DeviceCounters* zero_ctr) {
    __shared__ uint32_t lds_stack[kLdsStack * kBlock];
    if (blockIdx.x == 0) { // the counter sets of the NEXT launch / frame (double-buffered: nrays_hip.hip, k_primary)
        if (zero_counts && threadIdx.x < kNumCounts) zero_counts[threadIdx.x] = 0u;
        if (zero_ctr && threadIdx.x < sizeof(DeviceCounters) / 4) ((uint32_t*)zero_ctr)[threadIdx.x] = 0u;
    }
    wf_clear_next(clear_next, kWfMaxSegs);
    Stack st; WfStackInit::make(st, lds_stack, spill);
    Cnt cnt; wf_cnt_init(cnt);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t lane_log2 = PLAIN ? 0u : R.lane_log2;
    const bool fill_rows = (R.win_nx < tiles_x || R.win_ny < tiles_y) && tile_begin == 0u;
    if (fill_rows)
        for (uint32_t rl = blockIdx.x; rl < R.rows_local; rl += gridDim.x)
            fill_background_row(S.background[0], S.background[1], S.background[2], R.spp, out, R.width, R.height, R.band_rows, R.band_owner, R.band_owners,
                                R.win_x0, R.win_nx, R.win_y0, R.win_ny, lane_log2, rl, threadIdx.x, kBlock);
    const uint32_t total_waves = gridDim.x * (kBlock / 64);
    const uint32_t my_wave = blockIdx.x * (kBlock / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    WfOut wo; wo.init();
    const uint32_t spp = PLAIN ? 1u : R.spp, ppt = 64u >> lane_log2;
    const uint32_t p = lane >> lane_log2, sq = lane & ((1u << lane_log2) - 1u);
    WfClaim work; work.init(claim_next, tile_end - tile_begin, my_wave, total_waves);
    uint32_t item;
    while (work.get(item)) {
        const uint32_t wt = tile_begin + item;
        uint32_t i, rl;
        wf_tile_pixel(R, lane_log2, wt, p, i, rl);
        const uint32_t j = wf_global_row(R, rl);
        const bool active = i < R.width && rl < R.rows_local && j < R.height;
        const uint32_t pix = rl * R.width + i;
        const bool tile_misses = __ballot(active && (int32_t)i >= R.cull_i0 && (int32_t)i <= R.cull_i1 && (int32_t)j >= R.cull_j0 && (int32_t)j <= R.cull_j1) == 0ULL;
        for (uint32_t g = 0u; g < spp; g += 1u << lane_log2) {
            const uint32_t s = g + sq;
            const bool sample_active = active && s < spp;
            RayState ray;
            bool wave_may_hit = false; // wave-uniform
            if (!tile_misses) {
                generate_primary<PLAIN>(R, i < R.width ? i : R.width - 1u, j < R.height ? j : R.height - 1u, s, pix, ray);
                wave_may_hit = __ballot(sample_active && primary_may_hit(S, ray.o, ray.d)) != 0ULL;
            }
            Hit hit; hit.t = 0.0; hit.inst = 0u; hit.prim = 0u;
            bool got = false;
            if (wave_may_hit && sample_active) got = wf_closest<FEAT>(S, st, ray.o, ray.d, hit, cnt);
            const uint32_t path = spp == 1u ? pix : (item * ppt + p) * spp + s;
            if (sample_active && !got) { // Scene::trace returns the background (scene.rs:157-161): the chain's sum is 0 + background
                float* a = acc + (size_t)path * 3;
                const float c0 = 0.0f + S.background[0], c1 = 0.0f + S.background[1], c2 = 0.0f + S.background[2];
#if NR_NT_STORES
                __builtin_nontemporal_store(c0, a); __builtin_nontemporal_store(c1, a + 1); __builtin_nontemporal_store(c2, a + 2);
#else
                a[0] = c0; a[1] = c1; a[2] = c2;
#endif
            }
            const uint32_t at = wo.append(q, got);
            if (got) { if (at != ~0u) { wf_store_ray(q, at, ray, path); wf_store_hit(q, at, true, hit); } else atomicOr(&ctr->overflow, 1u); }
        }
        if (!active && sq == 0u && i < R.width && rl < R.rows_local) { // padding rows of the last band
            float* o = out + (size_t)pix * 3;
            o[0] = 0.0f; o[1] = 0.0f; o[2] = 0.0f;
        }
    }
    wo.finish(q);
    wf_flush(ctr, cnt);
}

// ------------------------------------------------------------------------------------------------ stage: closest hit
template <int FEAT>
__global__ void __launch_bounds__(kBlock, NR_WF_OCC) k_wf_closest(DScene S, WfQueue q, uint32_t* spill, uint32_t* claim_next, uint32_t* clear_next) {
    __shared__ uint32_t lds_stack[kLdsStack * kBlock];
    wf_clear_next(clear_next, kWfMaxSegs);
    Stack st; WfStackInit::make(st, lds_stack, spill);
    Cnt cnt; wf_cnt_init(cnt);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total_waves = gridDim.x * (kBlock / 64);
    const uint32_t my_wave = blockIdx.x * (kBlock / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    WfClaim work; work.init(claim_next, *q.nblocks * kWfChunksPerBlock, my_wave, total_waves);
    uint32_t c;
    while (work.get(c)) {
        if (lane < wf_chunk_valid(q, c)) {
            const uint32_t at = c * 64u + lane;
            d3 o, d; wf_load_od(q, at, o, d);
            Hit hit; hit.t = 0.0; hit.inst = 0u; hit.prim = 0u;
            const bool got = wf_closest<FEAT>(S, st, o, d, hit, cnt);
            wf_store_hit(q, at, got, hit);
        }
    }
}

// One light sample position (light.rs:57-63); `k` = sample index (the staged path renders lights with racsample 1: k = 0).
NR_DEV d3 wf_light_pos(const LightRec& light, unsigned long long ray_key, uint32_t li, uint32_t k) {
    d3 pos = D3(light.pos[0], light.pos[1], light.pos[2]);
    if (light.radius != 0.0) {
        const unsigned long long sk = rng_hash(rng_hash(ray_key, kSaltLight + li), k);
        pos = pos + D3(rng_u01(sk, 0), rng_u01(sk, 1), rng_u01(sk, 2)) * light.radius;
    }
    return pos;
}

// ------------------------------------------------------------------------------------------------ stage: shadow rays
// Item = (chunk, light): the 64 hits of a chunk towards one light (phong_material.rs:108-116 + scene.rs:147-161).
// shres[light * q.slots + slot] = (blocked, filter rgb).
template <int FEAT>
__global__ void __launch_bounds__(kBlock, NR_WF_OCC) k_wf_shadow(DScene S, WfQueue q, uint4* __restrict__ shres, DeviceCounters* ctr, uint32_t* spill, uint32_t* claim_next, uint32_t* clear_next) {
    __shared__ uint32_t lds_stack[kLdsStack * kBlock];
    wf_clear_next(clear_next, kWfMaxSegs);
    Stack st; WfStackInit::make(st, lds_stack, spill);
    Cnt cnt; wf_cnt_init(cnt);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total_waves = gridDim.x * (kBlock / 64);
    const uint32_t my_wave = blockIdx.x * (kBlock / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t nl = S.num_lights;
    WfClaim work; work.init(claim_next, *q.nblocks * kWfChunksPerBlock * nl, my_wave, total_waves);
    uint32_t it;
    while (work.get(it)) {
        const uint32_t c = it / nl, li = it - c * nl;
        if (lane < wf_chunk_valid(q, c)) {
            const uint32_t at = c * 64u + lane;
            Hit hit;
            if (wf_load_hit(q, at, hit) && ((S.shade[wf_hit_node<FEAT>(S, hit)].flags >> 8) & 0xffu) == NRAYS_MAT_PHONG) {
                d3 o, d; wf_load_od(q, at, o, d);
                const LightRec& light = S.lights[li];
                unsigned long long key = 0ULL;
                if (light.radius != 0.0) { const uint4 k = q.kp[at]; key = ((unsigned long long)k.y << 32) | k.x; }
                const d3 pos = wf_light_pos(light, key, li, 0u);
                const d3 point = o + d * hit.t;
                d3 ldir = pos - point;
                const double nrm = norm(ldir);
                ldir = ldir / nrm;
                f3 filter = F3(1.0f, 1.0f, 1.0f);
                cnt.shadow++;
                const bool blocked = shadow_query<false, FEAT>(S, st, point + ldir * 0.001, ldir, nrm - 0.001, filter, cnt);
                shres[(size_t)li * q.slots + at] = make_uint4(blocked ? 1u : 0u, __float_as_uint(filter.x), __float_as_uint(filter.y), __float_as_uint(filter.z));
            }
        }
    }
    wf_flush(ctr, cnt);
}

// PhongMaterial::compute (phong_material.rs:72-151) with the shadow queries of the hit already answered (k_wf_shadow):
// the operations and their order are material_compute's (trace_device.h), lights with one sample each.
NR_DEV f4 wf_material_lit(const DScene& S, const ShadeRec& m, const RayState& ray, d3 point, const Isect& in, Cnt& cnt, const uint4* __restrict__ shres, size_t slots, uint32_t at) {
    if (((m.flags >> 8) & 0xffu) != NRAYS_MAT_PHONG) return material_ambiant<false>(m, in, cnt);
    f4 tex; tex.x = tex.y = tex.z = tex.w = 1.0f;
    float alpha = 1.0f;
    if (in.has_uv && m.tex.texels) tex = tex_sample<false>(m.tex, in.u, in.v, cnt);
    if (in.has_uv && m.alpha_tex.texels) alpha = tex_sample<false>(m.alpha_tex, in.u, in.v, cnt).w;
    f3 res = F3(m.ka[0] * tex.x, m.ka[1] * tex.y, m.ka[2] * tex.z);
    const d3 normal = in.n;
#pragma nounroll
    for (uint32_t li = 0; li < S.num_lights; ++li) {
        const LightRec& light = S.lights[li];
        f3 acc = F3(0.0f, 0.0f, 0.0f);
        const uint4 sr = shres[(size_t)li * slots + at];
        if (sr.x == 0u) { // lit
            const f3 filter = F3(__uint_as_float(sr.y), __uint_as_float(sr.z), __uint_as_float(sr.w));
            const d3 pos = wf_light_pos(light, ray.key, li, 0u);
            d3 ldir = pos - point;
            const double nrm = norm(ldir);
            ldir = ldir / nrm;
            const double dot_ldir_norm = dot(ldir, normal);
            float dcoeff = (float)dot_ldir_norm;
            dcoeff = dcoeff > 0.0f ? dcoeff : 0.0f;
            const f3 diffuse_color = F3(m.kd[0] * tex.x, m.kd[1] * tex.y, m.kd[2] * tex.z);
            const f3 diffuse = F3(diffuse_color.x * dcoeff, diffuse_color.y * dcoeff, diffuse_color.z * dcoeff);
            const d3 lproj = normal * dot_ldir_norm;
            const d3 rldir = normalize((-ldir) + lproj * 2.0);
            float scoeff = (float)(-dot(rldir, ray.d));
            if (scoeff > 0.0f) {
                scoeff = powf(scoeff, m.shininess);
                const f3 sp = F3(m.ks[0] * scoeff, m.ks[1] * scoeff, m.ks[2] * scoeff);
                acc.x = acc.x + light.color[0] * (filter.x * (diffuse.x + sp.x));
                acc.y = acc.y + light.color[1] * (filter.y * (diffuse.y + sp.y));
                acc.z = acc.z + light.color[2] * (filter.z * (diffuse.z + sp.z));
            } else {
                acc.x = acc.x + light.color[0] * (filter.x * diffuse.x);
                acc.y = acc.y + light.color[1] * (filter.y * diffuse.y);
                acc.z = acc.z + light.color[2] * (filter.z * diffuse.z);
            }
        }
        const uint32_t rs = light.racsample;
        const float inv = 1.0f / (float)(rs * rs);
        res.x = inv * acc.x + res.x; res.y = inv * acc.y + res.y; res.z = inv * acc.z + res.z;
    }
    f4 out; out.x = res.x; out.y = res.y; out.z = res.z; out.w = alpha;
    return out;
}

// ------------------------------------------------------------------------------------------------ stage: shade
// One step of Scene::trace (scene.rs:163-252) for the rays of generation `depth`: shade_hit (trace_device.h) without its
// closest-hit query.  MULTI: the shadow results come from k_wf_shadow; else the scene has ONE light and its shadow ray is
// traced here, before the shading state exists.  The ray's own weighted contribution is added to its chain's running sum
// (sum = sum + c generation after generation, trace_chain's order), the continuation goes to the wave's current block of `qn`.
template <int FEAT, bool MULTI>
__global__ void __launch_bounds__(kBlock, NR_WF_OCC_SHADE) k_wf_shade(DScene S, WfQueue q, WfQueue qn, const uint4* __restrict__ shres, float* __restrict__ acc, DeviceCounters* ctr,
                                                                       uint32_t* spill, uint32_t depth, uint32_t max_depth, uint32_t keyed, uint32_t emit, uint32_t* claim_next, uint32_t* clear_next) {
    __shared__ uint32_t lds_stack[kLdsStack * kBlock];
    wf_clear_next(clear_next, kWfMaxSegs);
    Stack st; WfStackInit::make(st, lds_stack, spill);
    Cnt cnt; wf_cnt_init(cnt);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total_waves = gridDim.x * (kBlock / 64);
    const uint32_t my_wave = blockIdx.x * (kBlock / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    WfOut wo; wo.init();
    WfClaim work; work.init(claim_next, *q.nblocks * kWfChunksPerBlock, my_wave, total_waves);
    uint32_t c;
    while (work.get(c)) {
        const uint32_t nvalid = wf_chunk_valid(q, c);
        if (nvalid == 0u) continue; // wave-uniform
        bool has_next = false;
        RayState ray; uint32_t path = 0u;
        ray.o = D3(0, 0, 0); ray.d = D3(0, 0, 1); ray.refr = 1.0; ray.energy = 0.0f; ray.weight = 0.0f; ray.key = 0ULL; ray.pixel = 0u;
        if (lane < nvalid) {
            const uint32_t at = c * 64u + lane;
            wf_load_ray(q, at, ray, path);
            Hit hit;
            f3 contrib;
            if (!wf_load_hit(q, at, hit)) { // the ray left the scene: background (scene.rs:157-161), the chain ends
                contrib = F3(S.background[0] * ray.weight, S.background[1] * ray.weight, S.background[2] * ray.weight);
            } else {
                Isect is; uint32_t node_id;
                resolve_hit<false, FEAT, false>(S, ray.o, ray.d, hit, is, node_id);
                bool pre = false, pre_lit = false; f3 pre_filter = F3(1.0f, 1.0f, 1.0f);
                if constexpr (!MULTI) if (S.num_lights == 1 && ((S.shade[node_id].flags >> 8) & 0xffu) == NRAYS_MAT_PHONG) { // shade_hit: the single shadow ray, traced first
                    const LightRec& light = S.lights[0];
                    if (light.racsample == 1u) {
                        d3 pos = D3(light.pos[0], light.pos[1], light.pos[2]);
                        if (light.radius != 0.0) {
                            unsigned long long sk = rng_hash(rng_hash(ray.key, kSaltLight), 0);
                            pos = pos + D3(rng_u01(sk, 0), rng_u01(sk, 1), rng_u01(sk, 2)) * light.radius;
                        }
                        d3 point = ray.o + ray.d * hit.t;
                        d3 ldir = pos - point;
                        double nrm = norm(ldir);
                        ldir = ldir / nrm;
                        cnt.shadow++;
                        pre = true;
                        pre_lit = !shadow_query<false, FEAT>(S, st, point + ldir * 0.001, ldir, nrm - 0.001, pre_filter, cnt);
                    }
                }
                is.toi = hit.t;
                const ShadeRec& sn = S.shade[node_id];
                d3 pt = ray.o + ray.d * hit.t;
                f4 obj;
                if constexpr (MULTI) obj = wf_material_lit(S, sn, ray, pt, is, cnt, shres, (size_t)q.slots, at);
                else obj = material_compute<false, FEAT>(S, st, sn, ray, pt, is, cnt, pre, pre_lit, pre_filter, 0u);
                // weight algebra and continuation: shade_hit's, line by line (scenes with double branching are not rendered here)
                const bool may_recurse = depth < (uint32_t)kMaxGenerations && (max_depth == 0 || depth < max_depth);
                const float mix = sn.refl_mix;
                const float alpha = obj.w * sn.alpha;
                const float wa = alpha == 1.0f ? ray.weight : ray.weight * alpha; // scene.rs:183-190
                const float wo = wa * (1.0f - mix);
                contrib = F3(obj.x * wo, obj.y * wo, obj.z * wo);
                const bool do_refl = mix != 0.0f && ray.energy > 0.1f && may_recurse; // scene.rs:204
                const bool do_refr = alpha != 1.0f && may_recurse;                    // scene.rs:229
                const d3 dirn = is.n * dot(ray.d, is.n);
                if (do_refr) { // scene.rs:229-248
                    double n1, n2;
                    if (ray.refr == 1.0) { n1 = 1.0; n2 = sn.refr_coeff; } else { n1 = sn.refr_coeff; n2 = 1.0; }
                    const d3 tangent = ray.d - dirn;
                    const d3 new_dir = normalize(dirn + tangent * (n2 / n1));
                    const float w = ray.weight * (1.0f - alpha);
                    ray.o = pt + new_dir * 0.001; ray.d = new_dir; ray.refr = n2; ray.weight = w;
                    ray.key = keyed ? rng_hash(ray.key, kSaltRefr) : 0ULL;
                    cnt.refr++; has_next = true;
                } else if (do_refl) { // scene.rs:204-214
                    const d3 rdir = ray.d - dirn * 2.0;
                    ray.o = pt + rdir * 0.001; ray.d = rdir; ray.energy = ray.energy - sn.refl_atenuation;
                    ray.weight = wa * mix; ray.key = keyed ? rng_hash(ray.key, kSaltRefl) : 0ULL;
                    cnt.refl++; has_next = true;
                }
                if (depth > cnt.max_depth) cnt.max_depth = depth;
            }
            float* a = acc + (size_t)path * 3;
            if (depth == 0u) { a[0] = 0.0f + contrib.x; a[1] = 0.0f + contrib.y; a[2] = 0.0f + contrib.z; }
            else { a[0] = a[0] + contrib.x; a[1] = a[1] + contrib.y; a[2] = a[2] + contrib.z; }
        }
        if (emit) {
            const uint32_t at = wo.append(qn, has_next);
            if (has_next) { if (at != ~0u) wf_store_ray(qn, at, ray, path); else atomicOr(&ctr->overflow, 1u); }
        }
    }
    if (emit) wo.finish(qn);
    wf_flush(ctr, cnt);
}

// ------------------------------------------------------------------------------------------------ anti-aliased frames
// tot_c = tot_c + trace(ray) sample after sample (scene.rs:72-91): one thread per pixel of the tile range.
__global__ void k_wf_resolve(DRender R, const float* __restrict__ acc, float* __restrict__ out, uint32_t tile_begin, uint32_t tile_end) {
    const uint32_t lane_log2 = R.lane_log2, ppt = 64u >> lane_log2;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)(tile_end - tile_begin) * ppt) return;
    const uint32_t wt = tile_begin + (uint32_t)(t / ppt), p = (uint32_t)(t % ppt);
    uint32_t i, rl;
    wf_tile_pixel(R, lane_log2, wt, p, i, rl);
    const uint32_t j = wf_global_row(R, rl);
    if (!(i < R.width && rl < R.rows_local && j < R.height)) return;
    const float* a = acc + t * R.spp * 3;
    float tx = 0.0f, ty = 0.0f, tz = 0.0f;
    for (uint32_t s = 0; s < R.spp; ++s) { tx = tx + a[3 * s]; ty = ty + a[3 * s + 1]; tz = tz + a[3 * s + 2]; }
    float* o = out + ((size_t)rl * R.width + i) * 3;
    o[0] = tx; o[1] = ty; o[2] = tz;
}

// ================================================================================================ stages with lane refill
// The same three traversal stages over traverse_stream() (stream_device.h): a lane whose ray is finished delivers its result and
// takes the next ray of the wave's stream instead of idling until the slowest lane of its 64 is done.  The streams below hand out
// the rays of the items a wave claims (WfClaim) in item order, so a wave's lanes still hold rays of neighbouring pixels.

// Primary rays of the wave tiles of a pass: position r of a tile = lane slot r % 64 of round r / 64 of k_wf_primary's loops (the same
// pixel, sample and path for the same r).  Per lane: the key and path of the ray in flight (origin and direction live in traverse_stream).
template <int FEAT, bool PLAIN>
struct WfPrimarySource {
    const DScene& S; const DRender& R; const WfQueue& q; WfOut& wo; WfClaim& work; float* acc; float* out; DeviceCounters* ctr;
    uint32_t tile_begin, lane_log2, spp, total_pos;
    uint32_t item, pos; bool have, exhausted; // wave-uniform
    unsigned long long key; uint32_t path;    // per lane
    NR_DEV void init(uint32_t tile_begin_) {
        tile_begin = tile_begin_; lane_log2 = PLAIN ? 0u : R.lane_log2; spp = PLAIN ? 1u : R.spp;
        total_pos = ((spp + (1u << lane_log2) - 1u) >> lane_log2) * 64u;
        item = 0u; pos = 0u; have = false; exhausted = false; key = 0ULL; path = 0u;
    }
    NR_DEV bool more() const { return !exhausted; }
    NR_DEV void background(uint32_t pth) const { // Scene::trace returns the background (scene.rs:157-161): the chain's sum is 0 + background
        float* a = acc + (size_t)pth * 3;
        const float c0 = 0.0f + S.background[0], c1 = 0.0f + S.background[1], c2 = 0.0f + S.background[2];
        __builtin_nontemporal_store(c0, a); __builtin_nontemporal_store(c1, a + 1); __builtin_nontemporal_store(c2, a + 2);
    }
    // pixel / sample / path of position r of wave tile `it` (relative to the pass); false: no sample there
    NR_DEV bool locate(uint32_t it, uint32_t r, uint32_t& i, uint32_t& j, uint32_t& s, uint32_t& pix, uint32_t& pth, bool& pad) const {
        const uint32_t slot = r & 63u, p = slot >> lane_log2, sq = slot & ((1u << lane_log2) - 1u);
        s = ((r >> 6) << lane_log2) + sq;
        uint32_t rl;
        wf_tile_pixel(R, lane_log2, tile_begin + it, p, i, rl);
        j = wf_global_row(R, rl);
        const bool active = i < R.width && rl < R.rows_local && j < R.height;
        pix = rl * R.width + i;
        pth = spp == 1u ? pix : (it * (64u >> lane_log2) + p) * spp + s;
        pad = !active && sq == 0u && r < 64u && i < R.width && rl < R.rows_local; // padding rows of the last band
        return active && s < spp;
    }
    NR_DEV void refill(bool idle, bool& got, d3& o, d3& d, double& tlimit) {
        const uint32_t lane = __lane_id();
        for (int round = 0; round < 3; ++round) {
            if (!have) {
                if (!work.get(item)) { exhausted = true; return; }
                have = true; pos = 0u;
                // the tile's own lane slots: padding rows, and whether any of its pixels lies inside the scene's screen bounds
                uint32_t i, j, s, pix, pth; bool pad;
                locate(item, lane, i, j, s, pix, pth, pad);
                if (pad) { float* p = out + (size_t)pix * 3; p[0] = 0.0f; p[1] = 0.0f; p[2] = 0.0f; }
                const bool inside = i < R.width && j < R.height && (int32_t)i >= R.cull_i0 && (int32_t)i <= R.cull_i1 && (int32_t)j >= R.cull_j0 && (int32_t)j <= R.cull_j1;
                if (__ballot(inside) == 0ULL) { // no ray of this tile can reach the scene: every sample is the background
                    for (uint32_t r = lane; r < total_pos; r += 64u) { if (locate(item, r, i, j, s, pix, pth, pad)) background(pth); }
                    have = false;
                    continue;
                }
            }
            const bool want = idle && !got;
            const unsigned long long m = __ballot(want);
            if (m == 0ULL) return;
            const uint32_t n = (uint32_t)__popcll(m), left = total_pos - pos, take = n < left ? n : left;
            const uint32_t rank = (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL));
            if (want && rank < take) {
                uint32_t i, j, s, pix, pth; bool pad;
                if (locate(item, pos + rank, i, j, s, pix, pth, pad)) {
                    RayState ray;
                    generate_primary<PLAIN>(R, i, j, s, pix, ray);
                    if (primary_may_hit(S, ray.o, ray.d)) { got = true; o = ray.o; d = ray.d; key = ray.key; path = pth; }
                    else background(pth);
                }
            }
            pos += take;
            if (pos >= total_pos) have = false;
        }
    }
    NR_DEV bool deliver(bool fin, d3 o, d3 d, bool hit, const Hit& h, bool blocked, f3 filter, bool gated) {
        (void)blocked; (void)filter;
        bool ok = false, again = false;
        if (fin) {
            if (hit) {
                Isect is; uint32_t node_id;
                if (resolve_hit<false, FEAT, true>(S, o, d, h, is, node_id) || gated) ok = true; else again = true;
            } else background(path);
        }
        const uint32_t at = wo.append(q, ok);
        if (ok) {
            if (at != ~0u) {
                RayState ray; ray.o = o; ray.d = d; ray.refr = 1.0; ray.energy = 1.0f; ray.weight = 1.0f; ray.key = key; ray.pixel = path;
                wf_store_ray(q, at, ray, path); wf_store_hit(q, at, true, h);
            } else atomicOr(&ctr->overflow, 1u);
        }
        return again;
    }
};

template <int FEAT, bool PLAIN>
__global__ void __launch_bounds__(kBlock, NR_WF_OCC_RF) k_wf_primary_rf(DScene S, DRender R, WfQueue q, float* __restrict__ out, float* __restrict__ acc, DeviceCounters* ctr,
                                                                     uint32_t* spill, uint32_t tiles_x, uint32_t tiles_y, uint32_t tile_begin, uint32_t tile_end,
                                                                     uint32_t* claim_next, uint32_t* clear_next, uint32_t* zero_counts, DeviceCounters* zero_ctr) {
    __shared__ uint32_t lds_stack[kLdsStack * kBlock];
    if (blockIdx.x == 0) {
        if (zero_counts && threadIdx.x < kNumCounts) zero_counts[threadIdx.x] = 0u;
        if (zero_ctr && threadIdx.x < sizeof(DeviceCounters) / 4) ((uint32_t*)zero_ctr)[threadIdx.x] = 0u;
    }
    wf_clear_next(clear_next, kWfMaxSegs);
    Stack st; WfStackInit::make(st, lds_stack, spill);
    Cnt cnt; wf_cnt_init(cnt);
    const uint32_t lane_log2 = PLAIN ? 0u : R.lane_log2;
    const bool fill_rows = (R.win_nx < tiles_x || R.win_ny < tiles_y) && tile_begin == 0u;
    if (fill_rows)
        for (uint32_t rl = blockIdx.x; rl < R.rows_local; rl += gridDim.x)
            fill_background_row(S.background[0], S.background[1], S.background[2], R.spp, out, R.width, R.height, R.band_rows, R.band_owner, R.band_owners,
                                R.win_x0, R.win_nx, R.win_y0, R.win_ny, lane_log2, rl, threadIdx.x, kBlock);
    const uint32_t total_waves = gridDim.x * (kBlock / 64);
    const uint32_t my_wave = blockIdx.x * (kBlock / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    WfOut wo; wo.init();
    WfClaim work; work.init(claim_next, tile_end - tile_begin, my_wave, total_waves);
    WfPrimarySource<FEAT, PLAIN> src{S, R, q, wo, work, acc, out, ctr};
    src.init(tile_begin);
    traverse_stream<false, FEAT>(S, st, src, cnt);
    wo.finish(q);
    wf_flush(ctr, cnt);
}

// The rays of a queue for their closest hit (generations >= 1).
template <int FEAT>
struct WfClosestSource {
    const DScene& S; const WfQueue& q; WfClaim& work;
    uint32_t chunk, pos, nvalid; bool have, exhausted; // wave-uniform
    uint32_t at;                                        // per lane
    NR_DEV void init() { chunk = 0u; pos = 0u; nvalid = 0u; have = false; exhausted = false; at = 0u; }
    NR_DEV bool more() const { return !exhausted; }
    NR_DEV void refill(bool idle, bool& got, d3& o, d3& d, double& tlimit) {
        const uint32_t lane = __lane_id();
        for (int round = 0; round < 3; ++round) {
            if (!have) {
                if (!work.get(chunk)) { exhausted = true; return; }
                nvalid = wf_chunk_valid(q, chunk); pos = 0u; have = nvalid != 0u;
                if (!have) continue;
            }
            const bool want = idle && !got;
            const unsigned long long m = __ballot(want);
            if (m == 0ULL) return;
            const uint32_t n = (uint32_t)__popcll(m), left = nvalid - pos, take = n < left ? n : left;
            const uint32_t rank = (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL));
            if (want && rank < take) { at = chunk * 64u + pos + rank; wf_load_od(q, at, o, d); got = true; }
            pos += take;
            if (pos >= nvalid) have = false;
        }
    }
    NR_DEV bool deliver(bool fin, d3 o, d3 d, bool hit, const Hit& h, bool blocked, f3 filter, bool gated) {
        (void)blocked; (void)filter;
        bool again = false;
        if (fin) {
            bool ok = false;
            if (hit) { Isect is; uint32_t node_id; if (resolve_hit<false, FEAT, true>(S, o, d, h, is, node_id) || gated) ok = true; else again = true; }
            if (!again) wf_store_hit(q, at, ok, h);
        }
        return again;
    }
};
template <int FEAT>
__global__ void __launch_bounds__(kBlock, NR_WF_OCC_RF) k_wf_closest_rf(DScene S, WfQueue q, uint32_t* spill, uint32_t* claim_next, uint32_t* clear_next) {
    __shared__ uint32_t lds_stack[kLdsStack * kBlock];
    wf_clear_next(clear_next, kWfMaxSegs);
    Stack st; WfStackInit::make(st, lds_stack, spill);
    Cnt cnt; wf_cnt_init(cnt);
    const uint32_t total_waves = gridDim.x * (kBlock / 64);
    const uint32_t my_wave = blockIdx.x * (kBlock / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    WfClaim work; work.init(claim_next, *q.nblocks * kWfChunksPerBlock, my_wave, total_waves);
    WfClosestSource<FEAT> src{S, q, work};
    src.init();
    traverse_stream<false, FEAT>(S, st, src, cnt);
}

// The shadow rays of the items (chunk, light) of a queue (k_wf_shadow's arithmetic).
template <int FEAT>
struct WfShadowSource {
    const DScene& S; const WfQueue& q; WfClaim& work; uint4* shres; Cnt& cnt;
    uint32_t chunk, li, pos, nvalid; bool have, exhausted; // wave-uniform
    uint32_t at, my_li;                                     // per lane
    NR_DEV void init() { chunk = 0u; li = 0u; pos = 0u; nvalid = 0u; have = false; exhausted = false; at = 0u; my_li = 0u; }
    NR_DEV bool more() const { return !exhausted; }
    NR_DEV void refill(bool idle, bool& got, d3& o, d3& d, double& tlimit) {
        const uint32_t lane = __lane_id(), nl = S.num_lights;
        for (int round = 0; round < 3; ++round) {
            if (!have) {
                uint32_t it;
                if (!work.get(it)) { exhausted = true; return; }
                chunk = it / nl; li = it - chunk * nl;
                nvalid = wf_chunk_valid(q, chunk); pos = 0u; have = nvalid != 0u;
                if (!have) continue;
            }
            const bool want = idle && !got;
            const unsigned long long m = __ballot(want);
            if (m == 0ULL) return;
            const uint32_t n = (uint32_t)__popcll(m), left = nvalid - pos, take = n < left ? n : left;
            const uint32_t rank = (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL));
            if (want && rank < take) {
                const uint32_t a = chunk * 64u + pos + rank;
                Hit hit;
                if (wf_load_hit(q, a, hit) && ((S.shade[wf_hit_node<FEAT>(S, hit)].flags >> 8) & 0xffu) == NRAYS_MAT_PHONG) {
                    d3 ro, rd; wf_load_od(q, a, ro, rd);
                    const LightRec& light = S.lights[li];
                    unsigned long long key = 0ULL;
                    if (light.radius != 0.0) { const uint4 k = q.kp[a]; key = ((unsigned long long)k.y << 32) | k.x; }
                    const d3 lp = wf_light_pos(light, key, li, 0u);
                    const d3 point = ro + rd * hit.t;
                    d3 ldir = lp - point;
                    const double nrm = norm(ldir);
                    ldir = ldir / nrm;
                    cnt.shadow++;
                    o = point + ldir * 0.001; d = ldir; tlimit = nrm - 0.001; at = a; my_li = li; got = true;
                }
            }
            pos += take;
            if (pos >= nvalid) have = false;
        }
    }
    NR_DEV bool deliver(bool fin, d3 o, d3 d, bool hit, const Hit& h, bool blocked, f3 filter, bool gated) {
        (void)o; (void)d; (void)hit; (void)h; (void)gated;
        if (fin) shres[(size_t)my_li * q.slots + at] = make_uint4(blocked ? 1u : 0u, __float_as_uint(filter.x), __float_as_uint(filter.y), __float_as_uint(filter.z));
        return false;
    }
};
template <int FEAT>
__global__ void __launch_bounds__(kBlock, NR_WF_OCC_RF) k_wf_shadow_rf(DScene S, WfQueue q, uint4* __restrict__ shres, DeviceCounters* ctr, uint32_t* spill, uint32_t* claim_next, uint32_t* clear_next) {
    __shared__ uint32_t lds_stack[kLdsStack * kBlock];
    wf_clear_next(clear_next, kWfMaxSegs);
    Stack st; WfStackInit::make(st, lds_stack, spill);
    Cnt cnt; wf_cnt_init(cnt);
    const uint32_t total_waves = gridDim.x * (kBlock / 64);
    const uint32_t my_wave = blockIdx.x * (kBlock / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    WfClaim work; work.init(claim_next, *q.nblocks * kWfChunksPerBlock * S.num_lights, my_wave, total_waves);
    WfShadowSource<FEAT> src{S, q, work, shres, cnt};
    src.init();
    traverse_stream<true, FEAT>(S, st, src, cnt);
    wf_flush(ctr, cnt);
}

// ================================================================================================ host side
#define WF_TRY(expr)                                                                                               \
    do {                                                                                                           \
        hipError_t e_ = (expr);                                                                                    \
        if (e_ != hipSuccess)                                                                                      \
            return set_last_error(e_ == hipErrorOutOfMemory ? NRAYS_ERR_OOM : NRAYS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct WavefrontState {
    void* block[2] = {nullptr, nullptr}; // ray records of the two alternating generations
    WfQueue q[2];
    uint32_t slots = 0;
    uint32_t* d_nblocks = nullptr;  // kMaxGenerations + 2 block counters (queue of generation g: d_nblocks[g]), cleared per pass
    uint32_t* h_nblocks = nullptr;  // pinned: the next generation's block count, read back between generations
    uint32_t* d_claim[2] = {nullptr, nullptr}; // range counters of the stage kernels, alternating: a kernel claims from one and clears the other
    int claim_parity = 0;           // d_claim[claim_parity] is all zero
    uint4* d_shres = nullptr; size_t shres_vecs = 0;
    float* d_acc = nullptr; size_t acc_floats = 0;
    uint64_t max_paths = 64ull << 20; // NRAYS_WF_MAX_PATHS: (pixel, sample) paths per pass over a range of wave tiles
    bool refill = true;               // NRAYS_WF_REFILL=0: the traversal stages run every 64 rays from start to end together (traverse()) instead of refilling free lanes
    bool fuse = false;                // NRAYS_WF_FUSE=1: single-light scenes trace their shadow ray inside k_wf_shade instead of k_wf_shadow (A/B)
};

static int wf_ensure(NraysScene* sc, uint32_t slots, size_t shres_vecs, size_t acc_floats) {
    if (!sc->wf) {
        sc->wf = new (std::nothrow) WavefrontState();
        if (!sc->wf) return set_last_error(NRAYS_ERR_OOM, "host allocation failed");
        WavefrontState& w = *sc->wf;
        if (const char* e = getenv("NRAYS_WF_MAX_PATHS")) w.max_paths = (uint64_t)std::max(4096ll, atoll(e));
        if (const char* e = getenv("NRAYS_WF_FUSE")) w.fuse = atoi(e) != 0;
        if (const char* e = getenv("NRAYS_WF_REFILL")) w.refill = atoi(e) != 0; // 2: also in anti-aliased frames
        WF_TRY(hipMalloc((void**)&w.d_nblocks, (kMaxGenerations + 2) * sizeof(uint32_t)));
        WF_TRY(hipHostMalloc((void**)&w.h_nblocks, 4 * sizeof(uint32_t), hipHostMallocDefault));
        for (int k = 0; k < 2; ++k) {
            WF_TRY(hipMalloc((void**)&w.d_claim[k], kWfMaxSegs * sizeof(uint32_t)));
            WF_TRY(hipMemset(w.d_claim[k], 0, kWfMaxSegs * sizeof(uint32_t)));
        }
    }
    WavefrontState& w = *sc->wf;
    if (slots > w.slots) {
        for (int k = 0; k < 2; ++k) if (w.block[k]) { (void)hipFree(w.block[k]); w.block[k] = nullptr; }
        w.slots = 0;
        const size_t per_slot = 6 * 16; // o01, o2d0, d12, re, kp, hit
        const size_t nblk = slots / kWfBlock;
        const size_t bytes = (size_t)slots * per_slot + nblk * sizeof(uint32_t) + 256;
        for (int k = 0; k < 2; ++k) {
            WF_TRY(hipMalloc(&w.block[k], bytes));
            char* c = (char*)w.block[k];
            WfQueue& q = w.q[k];
            q.o01 = (double2*)c; c += (size_t)slots * 16; q.o2d0 = (double2*)c; c += (size_t)slots * 16; q.d12 = (double2*)c; c += (size_t)slots * 16;
            q.re = (uint4*)c; c += (size_t)slots * 16; q.kp = (uint4*)c; c += (size_t)slots * 16; q.hit = (uint4*)c; c += (size_t)slots * 16;
            q.fill = (uint32_t*)c; c += nblk * sizeof(uint32_t);
            q.nblocks = nullptr;
            q.slots = slots;
        }
        w.slots = slots;
    }
    if (shres_vecs > w.shres_vecs) {
        if (w.d_shres) { (void)hipFree(w.d_shres); w.d_shres = nullptr; w.shres_vecs = 0; }
        WF_TRY(hipMalloc((void**)&w.d_shres, shres_vecs * sizeof(uint4)));
        w.shres_vecs = shres_vecs;
    }
    if (acc_floats > w.acc_floats) {
        if (w.d_acc) { (void)hipFree(w.d_acc); w.d_acc = nullptr; w.acc_floats = 0; }
        WF_TRY(hipMalloc((void**)&w.d_acc, acc_floats * sizeof(float)));
        w.acc_floats = acc_floats;
    }
    return NRAYS_OK;
}

void wavefront_release(NraysScene* sc) {
    if (!sc || !sc->wf) return;
    WavefrontState& w = *sc->wf;
    for (int k = 0; k < 2; ++k) if (w.block[k]) (void)hipFree(w.block[k]);
    for (int k = 0; k < 2; ++k) if (w.d_claim[k]) (void)hipFree(w.d_claim[k]);
    if (w.d_nblocks) (void)hipFree(w.d_nblocks);
    if (w.h_nblocks) (void)hipHostFree(w.h_nblocks);
    if (w.d_shres) (void)hipFree(w.d_shres);
    if (w.d_acc) (void)hipFree(w.d_acc);
    delete sc->wf;
    sc->wf = nullptr;
}

static bool wf_eligible(const NraysScene* sc) {
    const int f = sc->features;
    if (!(f == 2 || f == 6 || f == 18 || f == 22)) return false; // TriMesh nodes only (+ alpha shadows, + several lights)
    if (sc->host.any_double_branch || sc->max_primary_forced) return false;
    for (const LightRec& l : sc->host.lights) if (l.racsample != 1u) return false;
    if (sc->host.lights.empty()) return false;
    return true;
}

bool wavefront_wanted(const NraysScene* sc, const NraysRenderParams* p, uint32_t lane_log2) {
    if (sc->wavefront_mode == 0 || !wf_eligible(sc)) return false;
    if (sc->wavefront_mode == 1) return true;
    // The library's rule (profiles/r04_wavefront_ab.log): frames bound by the SUM of their rays gain from dense lanes and four waves
    // per SIMD; a frame that is as long as its deepest chain (one light, 1080p) is faster in the megakernel, whose long tiles start
    // first and overlap with everything else.
    (void)p; (void)lane_log2;
    return false;
}

template <int FEAT>
static int wf_render_feat(NraysScene* sc, const NraysRenderParams* p, DRender R, float* d_out, hipStream_t stream, uint32_t tiles_x, uint32_t tiles_y,
                          bool timed, int slot, DeviceCounters* next_ctr, uint32_t* next_counts) {
    constexpr bool kMulti = (FEAT & kFeatMultiSample) != 0;
    const uint32_t lane_log2 = R.lane_log2, spp = p->ray_per_pixel;
    const uint32_t nwt = lane_log2 ? R.win_nx * R.win_ny : R.win_nx * R.win_ny * 4u;
    const uint64_t paths_per_tile = (uint64_t)(64u >> lane_log2) * spp;
    // lane refill pays where the lanes of a wave part ways early (one ray per pixel: hairball 2.98 -> 2.67 ms); the samples of ONE pixel walk
    // together, and mixing in the next pixel's costs them their wave-uniform node fetches (16 spp: 22.0 -> 26.5 ms) — profiles/r04_wavefront_refill_ab.log
    const bool refill_on = (sc->wf ? sc->wf->refill : !(getenv("NRAYS_WF_REFILL") && atoi(getenv("NRAYS_WF_REFILL")) == 0)) && (lane_log2 == 0u || (getenv("NRAYS_WF_REFILL") && atoi(getenv("NRAYS_WF_REFILL")) == 2));
    const uint32_t grid_full = std::min<uint32_t>((uint32_t)kMaxGrid, (uint32_t)sc->num_cus * (uint32_t)(refill_on ? NR_WF_OCC_RF : NR_WF_OCC));
    const uint32_t waves_full = grid_full * (kBlock / 64);
    // tile ranges: every sample of a range's pixels in one pass (the samples of a pixel stay side by side in the queues)
    uint64_t max_paths = 64ull << 20;
    if (sc->wf) max_paths = sc->wf->max_paths; else if (const char* e = getenv("NRAYS_WF_MAX_PATHS")) max_paths = (uint64_t)std::max(4096ll, atoll(e));
    const uint32_t tiles_per_pass = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::max<uint32_t>(nwt, 1u), max_paths / std::max<uint64_t>(paths_per_tile, 1)));
    // slots: every path of a pass may hit, plus one partly filled block per producer wave — of the grid this frame launches (every stage
    // kernel runs on at most grid_full workgroups), not of the largest grid the library knows: that was 0.8 GB per queue for any frame
    const uint64_t slots64 = (((uint64_t)tiles_per_pass * paths_per_tile + kWfBlock - 1) / kWfBlock + (uint64_t)waves_full + 1) * kWfBlock;
    if (slots64 >= (1ull << 31)) return set_last_error(NRAYS_ERR_UNSUPPORTED, "staged path: pass too large");
    const uint32_t slots = (uint32_t)slots64;
    const size_t acc_floats = spp > 1 ? (size_t)tiles_per_pass * paths_per_tile * 3 : 0;
    int rc = wf_ensure(sc, slots, (size_t)slots * sc->d.num_lights, acc_floats);
    if (rc != NRAYS_OK) return rc;
    WavefrontState& w = *sc->wf;
    float* acc = spp > 1 ? w.d_acc : d_out;
    const bool plain = R.window_width == 0.0 && !R.use_rng && spp == 1u && R.width <= 16384u && R.height <= 16384u;
    const bool can_continue = sc->host.any_reflective || sc->host.any_transparent;
    const uint32_t keyed = R.use_rng ? 1u : 0u;
    R.sample_begin = 0; R.sample_end = spp; R.first_batch = 1u;
    // every stage kernel claims its items from d_claim[parity] (all zero) and clears the other set for its successor
    auto claim = [&]() { uint32_t* c = w.d_claim[w.claim_parity]; w.claim_parity ^= 1; return c; };
    bool first_pass = true;
    for (uint32_t t0 = 0; t0 < std::max<uint32_t>(nwt, 1u); t0 += tiles_per_pass) {
        const uint32_t t1 = std::min<uint32_t>(nwt, t0 + tiles_per_pass);
        if (first_pass && timed) WF_TRY(hipEventRecord(sc->ev_pbegin[slot], stream));
        WF_TRY(hipMemsetAsync(w.d_nblocks, 0, (kMaxGenerations + 2) * sizeof(uint32_t), stream));
        WfQueue q0 = w.q[0]; q0.nblocks = w.d_nblocks;
        {
            uint32_t* cn = claim(); uint32_t* cl = w.d_claim[w.claim_parity];
            uint32_t* zc = first_pass ? next_counts : nullptr; DeviceCounters* zctr = first_pass ? next_ctr : nullptr;
            if (refill_on) {
                if (plain) hipLaunchKernelGGL((k_wf_primary_rf<FEAT, true>), dim3(grid_full), dim3(kBlock), 0, stream, sc->d, R, q0, d_out, acc, sc->d_counters, sc->d_spill, tiles_x, tiles_y, t0, t1, cn, cl, zc, zctr);
                else hipLaunchKernelGGL((k_wf_primary_rf<FEAT, false>), dim3(grid_full), dim3(kBlock), 0, stream, sc->d, R, q0, d_out, acc, sc->d_counters, sc->d_spill, tiles_x, tiles_y, t0, t1, cn, cl, zc, zctr);
            } else if (plain) hipLaunchKernelGGL((k_wf_primary<FEAT, true>), dim3(grid_full), dim3(kBlock), 0, stream, sc->d, R, q0, d_out, acc, sc->d_counters, sc->d_spill, tiles_x, tiles_y, t0, t1, cn, cl, zc, zctr);
            else hipLaunchKernelGGL((k_wf_primary<FEAT, false>), dim3(grid_full), dim3(kBlock), 0, stream, sc->d, R, q0, d_out, acc, sc->d_counters, sc->d_spill, tiles_x, tiles_y, t0, t1, cn, cl, zc, zctr);
            WF_TRY(hipGetLastError());
        }
        uint32_t grid = grid_full; // workgroups of the generation's stage kernels (sized from its block count once that is known)
        for (uint32_t g = 0; g <= (uint32_t)kMaxGenerations; ++g) {
            WfQueue q = w.q[g & 1], qn = w.q[(g + 1) & 1];
            q.nblocks = w.d_nblocks + g; qn.nblocks = w.d_nblocks + g + 1;
            if (g > 0) {
                uint32_t* cn = claim(); uint32_t* cl = w.d_claim[w.claim_parity];
                if (refill_on) hipLaunchKernelGGL((k_wf_closest_rf<FEAT>), dim3(grid), dim3(kBlock), 0, stream, sc->d, q, sc->d_spill, cn, cl);
                else hipLaunchKernelGGL((k_wf_closest<FEAT>), dim3(grid), dim3(kBlock), 0, stream, sc->d, q, sc->d_spill, cn, cl);
                WF_TRY(hipGetLastError());
            }
            const bool fused = !kMulti && w.fuse;
            if (!fused) {
                const uint32_t grid_sh = g == 0 ? grid_full : std::min<uint32_t>(grid_full, grid * sc->d.num_lights);
                uint32_t* cn = claim(); uint32_t* cl = w.d_claim[w.claim_parity];
                if (refill_on) hipLaunchKernelGGL((k_wf_shadow_rf<FEAT>), dim3(grid_sh), dim3(kBlock), 0, stream, sc->d, q, w.d_shres, sc->d_counters, sc->d_spill, cn, cl);
                else hipLaunchKernelGGL((k_wf_shadow<FEAT>), dim3(grid_sh), dim3(kBlock), 0, stream, sc->d, q, w.d_shres, sc->d_counters, sc->d_spill, cn, cl);
                WF_TRY(hipGetLastError());
            }
            const bool emit = can_continue && g < (uint32_t)kMaxGenerations;
            {
                uint32_t* cn = claim(); uint32_t* cl = w.d_claim[w.claim_parity];
                if (fused) {
                    if constexpr (!kMulti) hipLaunchKernelGGL((k_wf_shade<FEAT, false>), dim3(grid), dim3(kBlock), 0, stream, sc->d, q, qn, (const uint4*)w.d_shres, acc, sc->d_counters, sc->d_spill,
                                                              g, p->max_depth, keyed, emit ? 1u : 0u, cn, cl);
                } else hipLaunchKernelGGL((k_wf_shade<FEAT, true>), dim3(grid), dim3(kBlock), 0, stream, sc->d, q, qn, (const uint4*)w.d_shres, acc, sc->d_counters, sc->d_spill,
                                          g, p->max_depth, keyed, emit ? 1u : 0u, cn, cl);
                WF_TRY(hipGetLastError());
            }
            if (!emit) break;
            WF_TRY(hipMemcpyAsync(w.h_nblocks, qn.nblocks, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            WF_TRY(hipStreamSynchronize(stream));
            if (w.h_nblocks[0] == 0u) break;
            grid = std::max<uint32_t>(1u, std::min<uint32_t>(grid_full, (w.h_nblocks[0] * kWfChunksPerBlock + 3u) / 4u));
        }
        if (spp > 1 && t1 > t0) {
            const size_t n = (size_t)(t1 - t0) * (64u >> lane_log2);
            hipLaunchKernelGGL(k_wf_resolve, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, R, (const float*)w.d_acc, d_out, t0, t1);
            WF_TRY(hipGetLastError());
        }
        if (first_pass && timed) WF_TRY(hipEventRecord(sc->ev_pend[slot], stream));
        first_pass = false;
    }
    return NRAYS_OK;
}

int wavefront_render(NraysScene* sc, const NraysRenderParams* p, DRender R, float* d_out, hipStream_t stream, uint32_t tiles_x, uint32_t tiles_y,
                     bool timed, int slot, DeviceCounters* next_ctr, uint32_t* next_counts) {
    switch (sc->features) {
    case 2: return wf_render_feat<2>(sc, p, R, d_out, stream, tiles_x, tiles_y, timed, slot, next_ctr, next_counts);
    case 6: return wf_render_feat<6>(sc, p, R, d_out, stream, tiles_x, tiles_y, timed, slot, next_ctr, next_counts);
    case 18: return wf_render_feat<18>(sc, p, R, d_out, stream, tiles_x, tiles_y, timed, slot, next_ctr, next_counts);
    case 22: return wf_render_feat<22>(sc, p, R, d_out, stream, tiles_x, tiles_y, timed, slot, next_ctr, next_counts);
    default: return set_last_error(NRAYS_ERR_UNSUPPORTED, "staged path: scene not eligible");
    }
}

} // namespace nrays
