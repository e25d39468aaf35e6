// nrays_hip.hip — gfx950 kernels and the C-ABI entry points of include/nrays_abi.h.
//
// Launch structure of one nrays_render (replaces scene::render, src/scene.rs:29-116):
//   k_tile_order     mesh scenes, from the second frame of a geometry on: wave tiles sorted by the previous frame's
//               per-tile cost (8 LDS counting sorts), so that the deep chains start first.
//   k_primary   persistent grid; one lane per pixel of an 8x8 wave tile, looping over the AA samples of the batch:
//               raygen -> [no ray of the tile passes the root box: background] -> closest hit -> Phong + shadow
//               rays -> continuation kept in registers (trace_chain) -> pixel write.  Only the second child of a
//               hit that spawns a reflection AND a refraction goes to the compacted HBM queue (wave ballots).
//   k_bounce    rounds over that queue (double-branching scenes only): same per-ray work, the weighted contribution added to
//               the pixel's 64-bit fixed-point sum (order-independent), second children appended to the next round's queue;
//   k_fold_fixed  adds those sums to the frame after the rounds of a sample batch.
//   k_resolve   divides by ray_per_pixel when it is > 1 (scene.rs:94).
//   k_untile    un-permutes gathered multi-GPU tile buffers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/nrays_abi.h"
#include "device_types.h"
#include "scene_build.h"
#include "scene_handle.h"
#include "tile_device.h"
#include "trace_device.h"
#include "wavefront.h"
#include "primary_kernel.h"

namespace nrays {

// Value range of the fixed-point sums (2^-32 units in a signed 64-bit integer): a contribution is clamped to +-9.0e18 units (|x| <= 2.1e9 — colours
// are O(1)), a NaN contribution counts as 0 (the float atomicAdd it replaced would have poisoned the pixel; the reference's f32 sum too), and a sum of
// several clamped contributions can wrap — none of which a frame of finite O(1) radiances can reach.
__device__ __forceinline__ long long to_fixed(float x) {
    double v = (double)x * 4294967296.0;
    v = v > 9.0e18 ? 9.0e18 : (v < -9.0e18 ? -9.0e18 : v); // (NaN -> 0 below)
    return v == v ? __double2ll_rn(v) : 0ll;
}
// out += fixed-point sums of the queued chains (k_bounce), which are cleared for the next sample batch.
__global__ void k_fold_fixed(float* __restrict__ out, long long* __restrict__ fixed, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long f = fixed[i];
    if (f != 0ll) { out[i] = out[i] + (float)((double)f * (1.0 / 4294967296.0)); fixed[i] = 0ll; }
}

template <bool STATS>
__global__ void __launch_bounds__(kBlock, NRAYS_WAVES_PER_SIMD) k_bounce(DScene S, RayQueue qin, const uint32_t* __restrict__ count_in, uint32_t capacity,
                                                    QueueOut qo, long long* __restrict__ fixed, DeviceCounters* ctr, uint32_t* spill,
                                                    uint32_t max_depth) {
    __shared__ uint32_t lds_stack[kLdsStack * kBlock];
    Stack st;
    st.lds = (lds_u32*)(lds_stack + threadIdx.x);
    st.spill_stride = gridDim.x * kBlock;
    st.spill = spill ? (global_u32*)(spill + (size_t)blockIdx.x * kBlock + threadIdx.x) : nullptr;
    st.lds0 = Stack::addr((lds_u32*)lds_stack);
    st.park = nullptr;
    st.init();
    Cnt cnt; cnt.node = cnt.tri = cnt.prim = cnt.hit = cnt.tex = cnt.shadow = cnt.refl = cnt.refr = cnt.max_depth = cnt.max_chain_nodes = cnt.traced = cnt.elided = cnt.fetch = 0;
#ifdef NR_PHASE_TIMING
    cnt.cyc_node = cnt.cyc_leaf = cnt.cyc_other = cnt.cyc_tri = 0; cnt.wv_node = cnt.ln_node = cnt.wv_tri = cnt.ln_tri = 0; cnt.cyc_closest0 = cnt.cyc_closestN = cnt.cyc_shadow = 0; cnt.wv_uni = 0; cnt.inq_node = cnt.inq_tri = 0; for (int k_ = 0; k_ < 8; ++k_) cnt.cyc_x[k_] = 0;
#endif
    uint32_t n = *count_in;
    if (n > capacity) n = capacity;
    for (uint32_t base = blockIdx.x * kBlock; base < n; base += gridDim.x * kBlock) { // block-uniform trip count
        uint32_t idx = base + threadIdx.x;
        bool active = idx < n;
        RayState ray;
        ray.o = D3(0, 0, 0); ray.d = D3(0, 0, 1); ray.refr = 1.0; ray.energy = 0.0f; ray.weight = 0.0f; ray.key = 0; ray.pixel = 0;
        uint32_t depth = 0;
        if (active) queue_load(qin, idx, ray, depth);
        f3 c = trace_chain<STATS, kFeatAll>(S, st, active, ray, depth, max_depth, qo, cnt, true);
        if (active) {
            // The queued chains of a pixel finish in no particular order.  Their contributions are therefore summed as 64-bit
            // FIXED-POINT numbers (2^-32 units: integer addition is associative, so the sum does not depend on the order) and
            // folded into the frame by k_fold_fixed once the rounds of the batch are over: frames of double-branching scenes are
            // bit-reproducible.  The 2.3e-10 quantum is far below the f32 resolution of a pixel value.
            unsigned long long* f = (unsigned long long*)(fixed + (size_t)ray.pixel * 3);
            atomicAdd(f, (unsigned long long)to_fixed(c.x)); atomicAdd(f + 1, (unsigned long long)to_fixed(c.y)); atomicAdd(f + 2, (unsigned long long)to_fixed(c.z));
        }
    }
    flush_counters(ctr, cnt, STATS);
}

// (FEAT: kFeatAll, or kFeatMesh for scenes of opaque TriMesh nodes only — the permutation whose node phases end by quorum in
// hair-like meshes, so that the independent fixtures also cover that path.)
// nrays_debug_cast_batch: the closest-hit query with the deferred exact gates exactly as shade_hit runs it (ungated traversal, the
// winner checked against the reference's AABB gates, fully gated repeat for knife-edge rays), or the shadow query, on rays from memory.
template <int FEAT>
__global__ void __launch_bounds__(kBlock, NRAYS_WAVES_PER_SIMD) k_cast_batch(DScene S, uint32_t mode, uint32_t n, const double* __restrict__ ro, const double* __restrict__ rd,
                                                                              const double* __restrict__ max_toi, NraysCastResult* __restrict__ out, uint32_t* spill) {
    __shared__ uint32_t lds_stack[kLdsStack * kBlock];
    Stack st;
    st.lds = (lds_u32*)(lds_stack + threadIdx.x);
    st.spill_stride = gridDim.x * kBlock;
    st.spill = spill ? (global_u32*)(spill + (size_t)blockIdx.x * kBlock + threadIdx.x) : nullptr;
    st.lds0 = Stack::addr((lds_u32*)lds_stack);
    st.park = nullptr;
    st.init();
    Cnt cnt; cnt.node = cnt.tri = cnt.prim = cnt.hit = cnt.tex = cnt.shadow = cnt.refl = cnt.refr = cnt.max_depth = cnt.max_chain_nodes = cnt.traced = cnt.elided = cnt.fetch = 0;
#ifdef NR_PHASE_TIMING
    cnt.cyc_node = cnt.cyc_leaf = cnt.cyc_other = cnt.cyc_tri = 0; cnt.wv_node = cnt.ln_node = cnt.wv_tri = cnt.ln_tri = 0; cnt.cyc_closest0 = cnt.cyc_closestN = cnt.cyc_shadow = 0; cnt.wv_uni = 0; cnt.inq_node = cnt.inq_tri = 0; for (int k_ = 0; k_ < 8; ++k_) cnt.cyc_x[k_] = 0;
#endif
    for (uint32_t base = blockIdx.x * kBlock; base < n; base += gridDim.x * kBlock) {
        const uint32_t i = base + threadIdx.x;
        if (i >= n) continue;
        const d3 o = D3(ro[3 * (size_t)i], ro[3 * (size_t)i + 1], ro[3 * (size_t)i + 2]), d = D3(rd[3 * (size_t)i], rd[3 * (size_t)i + 1], rd[3 * (size_t)i + 2]);
        NraysCastResult r; r.toi = 0.0; r.normal[0] = r.normal[1] = r.normal[2] = 0.0; r.uv[0] = r.uv[1] = 0.0; r.node_id = -1; r.flags = 0u;
        Hit hit; f3 filter = F3(1.0f, 1.0f, 1.0f);
        if (mode == 1u) {
            const bool blocked = traverse<true, false, FEAT>(S, st, o, d, max_toi[i], hit, filter, cnt);
            r.flags = blocked ? 1u : 0u; r.normal[0] = filter.x; r.normal[1] = filter.y; r.normal[2] = filter.z;
        } else {
            Isect is; uint32_t node_id = 0; bool gated = false, any = false;
            for (;;) {
                any = traverse<false, false, FEAT>(S, st, o, d, kDblMax, hit, filter, cnt, gated, &is);
                if (!any) break;
                if (resolve_hit<false, FEAT, true>(S, o, d, hit, is, node_id) || gated) break;
                gated = true;
            }
            if (any) {
                r.toi = hit.t; r.normal[0] = is.n.x; r.normal[1] = is.n.y; r.normal[2] = is.n.z; r.uv[0] = is.u; r.uv[1] = is.v;
                r.node_id = (int32_t)node_id; r.flags = 1u | (is.has_uv ? 2u : 0u);
            }
        }
        out[i] = r;
    }
}

// Wave tiles in descending order of last frame's cost.  XCD x's work list is the subset { i : i mod 8 == x } of the
// wave tiles (a uniform sample of the image), sorted by workgroup x with a 256-bucket counting sort in LDS
// on (exponent, 3 mantissa bits) of the cycle counts — an approximate order is all a work queue needs.
// Entry k of list x is stored at order[8 k + x].
__device__ __forceinline__ uint32_t cost_bucket(uint32_t c) {
    if (c == 0u) return 0u;
    uint32_t e = 31u - (uint32_t)__clz((int)c);
    uint32_t m = e >= 3u ? (c >> (e - 3u)) & 7u : (c << (3u - e)) & 7u;
    return e * 8u + m;
}
// Light-parallel tiles (split_lsl > 0, multi-light mesh frames): a tile whose cost exceeds split_factor x the frame's work per
// resident wave (its list's sum x 8 / waves: the lists are uniform samples of the image) would sit on the frame's critical path —
// it enters the list as 2^split_lsl entries (its parts, DRender::light_lsl), each priced at a third of the tile.  split_factor < 0
// splits every tile (tests).  order_len[x] receives the list's length.
// clear != 0: every cost is zeroed after its last read here — the frame that follows records into split entries by atomicMax, and a memset of its own
// would be one more launch between this kernel and k_primary.
__global__ void __launch_bounds__(1024) k_tile_order(uint32_t* __restrict__ cost, uint32_t* __restrict__ order, uint32_t n, unsigned long long* stats,
                                                     uint32_t split_lsl, float split_factor, uint32_t waves, uint32_t* __restrict__ order_len, uint32_t clear, float split_hyst) {
    __shared__ uint32_t hist[256];
    __shared__ unsigned long long wg_sum;
    __shared__ uint32_t wg_max, wg_total;
    const uint32_t x = blockIdx.x; // 0..7
    if (threadIdx.x < 256u) hist[threadIdx.x] = 0u;
    if (threadIdx.x == 0u) { wg_sum = 0ULL; wg_max = 0u; wg_total = 0u; }
    __syncthreads();
    unsigned long long my_sum = 0ULL; uint32_t my_max = 0u;
    for (uint32_t i = x + 8u * threadIdx.x; i < n; i += 8u * 1024u) { const uint32_t c = cost[i] & kCostMask; my_sum += c; my_max = c > my_max ? c : my_max; }
    if (stats || split_lsl) { // stats[2x] = sum of list x's tile costs, stats[2x + 1] = its largest one (both in the 16-cycle units of the cost array)
        if (my_sum) atomicAdd(&wg_sum, my_sum);
        if (my_max) atomicMax(&wg_max, my_max);
    }
    __syncthreads();
    if (stats && threadIdx.x == 0u) { stats[2u * x] = wg_sum; stats[2u * x + 1u] = (unsigned long long)wg_max; } // per list: no memset before the launch, the host adds them up
    const double per_wave = (double)(wg_sum * 8ULL) / (double)(waves ? waves : 1u);
    const unsigned long long thr = !split_lsl ? ~0ULL : (split_factor < 0.0f ? 0ULL : (unsigned long long)(split_factor * per_wave));
    const unsigned long long thr_keep = !split_lsl ? ~0ULL : (split_factor < 0.0f ? 0ULL : (unsigned long long)(split_factor * split_hyst * per_wave)); // a tile that ran in parts stays split down to here
    const uint32_t parts = 1u << split_lsl;
    auto heavy = [&](uint32_t rec) { const uint32_t c = rec & kCostMask; return split_lsl != 0u && (unsigned long long)c >= ((rec & kCostSplit) ? thr_keep : thr) && (split_factor < 0.0f || c != 0u); };
    for (uint32_t i = x + 8u * threadIdx.x; i < n; i += 8u * 1024u) {
        const uint32_t rec = cost[i], c = rec & kCostMask;
        if (heavy(rec)) atomicAdd(&hist[cost_bucket(c / 3u)], parts); else atomicAdd(&hist[cost_bucket(c)], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) { // exclusive prefix, most expensive bucket first
        uint32_t acc = 0u;
        for (int b = 255; b >= 0; --b) { uint32_t c = hist[b]; hist[b] = acc; acc += c; }
        wg_total = acc;
    }
    __syncthreads();
    for (uint32_t i = x + 8u * threadIdx.x; i < n; i += 8u * 1024u) {
        const uint32_t rec = cost[i], c = rec & kCostMask;
        if (heavy(rec)) {
            const uint32_t at = atomicAdd(&hist[cost_bucket(c / 3u)], parts);
            for (uint32_t s = 0; s < parts; ++s) order[8u * (at + s) + x] = i | (s << 28) | kEntrySplit;
        } else order[8u * atomicAdd(&hist[cost_bucket(c)], 1u) + x] = i;
        if (clear) cost[i] = 0u;
    }
    if (order_len && threadIdx.x == 0u) order_len[x] = wg_total;
}

// A first guess of the wave-tile costs of a camera that has no history yet (the reference's caller renders every camera ONCE,
// examples/loader3d.rs:67-93: the first frame is the one that counts for it).  A mesh frame is as long as its deepest chains, and a
// chain is deep where the primary ray crosses many nodes that can continue it — alpha-mapped / transparent layers (scene.rs:229) and
// mirrors (scene.rs:204).  cost = 1 + 4 x (boxes of such nodes the ray through the tile's centre pixel crosses), f32 slab tests against
// their world AABBs: a few microseconds, and k_tile_order then starts those tiles first, as it does from recorded costs on later
// frames.  Scheduling only: pixels do not depend on it.
__global__ void k_seed_costs(DRender R, const float* __restrict__ boxes, uint32_t nboxes, uint32_t* __restrict__ cost, uint32_t nwt, uint32_t nrays) {
    const uint32_t wt = blockIdx.x * blockDim.x + threadIdx.x;
    if (wt >= nwt) return;
    const uint32_t tile = wt >> 2, sub = wt & 3u;
    const uint32_t tx = R.win_x0 + tile % R.win_nx, ty = R.win_y0 + tile / R.win_nx;
    uint32_t n = 0u;
    for (uint32_t q = 0; q < nrays; ++q) { // the tile's centre pixel, or the centres of its four quadrants
        const uint32_t px = nrays == 1u ? 4u : 2u + 4u * (q & 1u), py = nrays == 1u ? 4u : 2u + 4u * (q >> 1);
        const uint32_t i = tx * kTile + ((sub & 1u) << 3) + px, rl = ty * kTile + ((sub >> 1) << 3) + py;
        uint32_t j = rl;
        if (R.band_rows != 0 && R.band_owners > 1) j = ((rl / R.band_rows) * R.band_owners + R.band_owner) * R.band_rows + (rl % R.band_rows);
        const double dx = ((double)i / (double)R.width - 0.5) * 2.0, dy = -((double)j / (double)R.height - 0.5) * 2.0;
        double h[4];
        for (int r = 0; r < 4; ++r) h[r] = R.m[r] * dx + R.m[4 + r] * dy - R.m[8 + r] + R.m[12 + r];
        const float ox = (float)R.eye[0], oy = (float)R.eye[1], oz = (float)R.eye[2];
        const float ix = 1.0f / (float)(h[0] / h[3] - R.eye[0]), iy = 1.0f / (float)(h[1] / h[3] - R.eye[1]), iz = 1.0f / (float)(h[2] / h[3] - R.eye[2]);
        for (uint32_t b = 0; b < nboxes; ++b) {
            const float* bx = boxes + 6u * b;
            float t0 = (bx[0] - ox) * ix, t1 = (bx[3] - ox) * ix; float lo = fminf(t0, t1), hi = fmaxf(t0, t1);
            t0 = (bx[1] - oy) * iy; t1 = (bx[4] - oy) * iy; lo = fmaxf(lo, fminf(t0, t1)); hi = fminf(hi, fmaxf(t0, t1));
            t0 = (bx[2] - oz) * iz; t1 = (bx[5] - oz) * iz; lo = fmaxf(lo, fminf(t0, t1)); hi = fminf(hi, fmaxf(t0, t1));
            n += (hi >= fmaxf(lo, 0.0f)) ? 1u : 0u;
        }
    }
    cost[wt] = 1u + (nrays == 1u ? 4u : 1u) * n;
}

// Screen bounds of the scene for one camera: the pixel rectangle outside of which no primary ray can reach the scene's
// bounding box, so that k_primary can write the background for whole wave tiles without generating their rays.
// Raygen (generate_primary, scene.rs:74-89) sends the ray of sample position (ox, oy) from `eye` through
// P = h.xyz / h.w with h = M (dx, dy, -1, 1), dx = (ox / W - 0.5) 2, dy = -(oy / H - 0.5) 2; its direction is a positive
// multiple of sgn(h.w) D(dx, dy), D = h.xyz - eye h.w = Dc + dx Dx + dy Dy — AFFINE in (dx, dy).  A corner c of the
// box lies on the ray of (dx, dy) iff c - eye = a Dc + b Dx + g Dy with dx = b / a, dy = g / a and a sgn > 0 (in front).
// If that holds for all eight corners, every point of the box is a combination of the corners with weights of one sign,
// so its (dx, dy) lies between the corners' extremes: rays outside that rectangle miss the box, hence every node, hence
// return the background; a box entirely behind the eye is missed by every ray.  Everything else — a corner beside the eye, h.w changing sign over the frame, a
// degenerate matrix, planes in the scene — gives "every pixel may hit".  Two pixels of slack plus the jitter window
// cover the rounding of this f64 computation and of the rays themselves by many orders of magnitude.
struct ScreenBounds { int32_t i0, i1, j0, j1; };
static ScreenBounds screen_bounds(const HostScene& h, const NraysRenderParams* p) {
    const ScreenBounds all = {INT32_MIN, INT32_MAX, INT32_MIN, INT32_MAX}, none = {0, -1, 0, -1};
    if (!h.bounded) return all;
    for (int a = 0; a < 3; ++a) {
        if (!(h.bounds_mn[a] <= h.bounds_mx[a])) return none; // no bounded node at all
        if (!std::isfinite(h.bounds_mn[a]) || !std::isfinite(h.bounds_mx[a])) return all;
    }
    const double* M = p->inv_proj_view; // column-major
    const double* e = p->camera_eye;
    double hc[4], hx[4], hy[4];
    for (int r = 0; r < 4; ++r) { hc[r] = M[12 + r] - M[8 + r]; hx[r] = M[r]; hy[r] = M[4 + r]; }
    // h.w keeps one sign over the frame (it is affine in dx, dy: check the corners of a slightly larger rectangle)
    const double ext = 1.0 + 4.0 / std::min<double>(p->width, p->height) + std::fabs(p->window_width);
    double wmin = INFINITY, wmax = -INFINITY, wscale = std::fabs(hc[3]) + std::fabs(hx[3]) + std::fabs(hy[3]);
    for (int k = 0; k < 4; ++k) { double w = hc[3] + ((k & 1) ? ext : -ext) * hx[3] + ((k & 2) ? ext : -ext) * hy[3]; wmin = std::min(wmin, w); wmax = std::max(wmax, w); }
    if (!(wscale > 0.0) || !std::isfinite(wscale) || !(wmin > 1e-9 * wscale || wmax < -1e-9 * wscale)) return all;
    const double sgn = wmin > 0.0 ? 1.0 : -1.0;
    double Dc[3], Dx[3], Dy[3];
    for (int a = 0; a < 3; ++a) { Dc[a] = hc[a] - e[a] * hc[3]; Dx[a] = hx[a] - e[a] * hx[3]; Dy[a] = hy[a] - e[a] * hy[3]; }
    auto det3 = [](const double* u, const double* v, const double* w) {
        return u[0] * (v[1] * w[2] - v[2] * w[1]) - u[1] * (v[0] * w[2] - v[2] * w[0]) + u[2] * (v[0] * w[1] - v[1] * w[0]);
    };
    auto len = [](const double* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
    const double det = det3(Dc, Dx, Dy);
    if (!std::isfinite(det) || !(std::fabs(det) > 1e-9 * len(Dc) * len(Dx) * len(Dy))) return all;
    double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
    double A[8], B[8], G[8];
    int behind = 0;
    for (int k = 0; k < 8; ++k) {
        double c[3];
        for (int a = 0; a < 3; ++a) c[a] = (double)((k >> a) & 1 ? h.bounds_mx[a] : h.bounds_mn[a]) - e[a];
        A[k] = det3(c, Dx, Dy) / det; B[k] = det3(Dc, c, Dy) / det; G[k] = det3(Dc, Dx, c) / det; // Cramer
        if (!std::isfinite(A[k]) || !std::isfinite(B[k]) || !std::isfinite(G[k])) return all;
        if (A[k] * sgn < -1e-9 * (std::fabs(A[k]) + std::fabs(B[k]) + std::fabs(G[k]))) ++behind;
    }
    if (behind == 8) return none; // the whole box lies behind the eye: its points are NEGATIVE multiples of every ray direction
    for (int k = 0; k < 8; ++k) {
        const double a_ = A[k], b_ = B[k], g_ = G[k];
        if (!(a_ * sgn > 1e-9 * (std::fabs(a_) + std::fabs(b_) + std::fabs(g_)))) return all; // beside / behind the eye
        const double dx = b_ / a_, dy = g_ / a_;
        if (!std::isfinite(dx) || !std::isfinite(dy)) return all;
        const double ox = (dx * 0.5 + 0.5) * (double)p->width, oy = (-dy * 0.5 + 0.5) * (double)p->height;
        xmin = std::min(xmin, ox); xmax = std::max(xmax, ox); ymin = std::min(ymin, oy); ymax = std::max(ymax, oy);
    }
    // pixel i takes its samples at ox in [i - window / 2, i + window / 2]
    const double slack = 2.0 + 0.5 * std::fabs(p->window_width);
    auto clampi = [](double v) { return (int32_t)std::max(-1.0e9, std::min(1.0e9, v)); };
    ScreenBounds r = {clampi(std::floor(xmin - slack)), clampi(std::ceil(xmax + slack)), clampi(std::floor(ymin - slack)), clampi(std::ceil(ymax + slack))};
    return r;
}

// What a cost order recorded for one camera is worth for another: the larger of (a) the angle between the two cameras' rays through each
// corner of the frame and (b) the parallax of the nearest geometry — |eye shift| over the distance from the eye to the scene's bounding box
// (at least a twentieth of its diagonal: a camera inside the scene) — both in pixels of the frame.  Tile costs vary over blocks of pixels, so
// an order stays useful while the view has shifted by less than a block (kNearPixels).  Scheduling only.
constexpr double kNearPixels = 16.0;
static CamSnap cam_snapshot(const HostScene& h, const NraysRenderParams* p) {
    CamSnap c; c.valid = false;
    const double* M = p->inv_proj_view;
    for (int a = 0; a < 3; ++a) c.eye[a] = p->camera_eye[a];
    for (int k = 0; k < 4; ++k) {
        const double dx = (k & 1) ? 1.0 : -1.0, dy = (k & 2) ? 1.0 : -1.0;
        double hh[4];
        for (int r = 0; r < 4; ++r) hh[r] = M[r] * dx + M[4 + r] * dy - M[8 + r] + M[12 + r];
        double d[3], n = 0.0;
        for (int a = 0; a < 3; ++a) { d[a] = hh[a] / hh[3] - c.eye[a]; n += d[a] * d[a]; }
        n = std::sqrt(n);
        if (!(n > 0.0) || !std::isfinite(n)) return c;
        for (int a = 0; a < 3; ++a) c.dir[k][a] = d[a] / n;
    }
    // angle of one pixel: the frame's diagonal chord over its diagonal in pixels
    double chord = 0.0;
    for (int a = 0; a < 3; ++a) chord += (c.dir[3][a] - c.dir[0][a]) * (c.dir[3][a] - c.dir[0][a]);
    c.pix_angle = std::sqrt(chord) / std::sqrt((double)p->width * p->width + (double)p->height * p->height);
    // distance to the nearest point of the bounded part of the scene
    double diag = 0.0, dist = 0.0; bool box = true;
    for (int a = 0; a < 3; ++a) {
        const double mn = h.bounds_mn[a], mx = h.bounds_mx[a];
        if (!(mn <= mx) || !std::isfinite(mn) || !std::isfinite(mx)) { box = false; break; }
        diag += (mx - mn) * (mx - mn);
        const double o = c.eye[a] < mn ? mn - c.eye[a] : (c.eye[a] > mx ? c.eye[a] - mx : 0.0);
        dist += o * o;
    }
    c.depth = box ? std::max(std::sqrt(dist), 0.05 * std::sqrt(diag)) : 1.0;
    c.valid = c.pix_angle > 0.0 && std::isfinite(c.pix_angle) && c.depth > 0.0;
    return c;
}
static double cam_shift_px(const CamSnap& a, const CamSnap& b) {
    if (!a.valid || !b.valid) return INFINITY;
    const double pa = std::min(a.pix_angle, b.pix_angle);
    double rot = 0.0, tr = 0.0;
    for (int k = 0; k < 4; ++k) { double q = 0.0; for (int x = 0; x < 3; ++x) q += (a.dir[k][x] - b.dir[k][x]) * (a.dir[k][x] - b.dir[k][x]); rot = std::max(rot, std::sqrt(q)); }
    for (int x = 0; x < 3; ++x) tr += (a.eye[x] - b.eye[x]) * (a.eye[x] - b.eye[x]);
    const double v = std::max(rot, std::sqrt(tr) / std::min(a.depth, b.depth)) / pa;
    return std::isfinite(v) ? v : INFINITY;
}

// Image::to_png quantisation (src/image.rs:66-76) of a finished frame: c * 255, clamped to [0, 255], truncated; NaN and
// negatives -> 0 (Rust's saturating `as u8`).  Same f32 operations as the host front-end's quantize_rgb8 (png_codec.cpp).
__global__ void k_quantize_rgb8(const float* __restrict__ rgb, uint8_t* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = rgb[i] * 255.0f;
    v = (v > 0.0f) ? v : 0.0f;
    v = v > 255.0f ? 255.0f : v;
    out[i] = (uint8_t)(uint32_t)v;
}

__global__ void k_resolve(float* out, size_t n, float spp) { // pxs.push(tot_c / ray_per_pixel as f32), scene.rs:94
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = out[i] / spp;
}

__global__ void k_untile(const float* __restrict__ gathered, float* __restrict__ out, uint32_t width, uint32_t height,
                         uint32_t band_rows, uint32_t owners, uint32_t rows_local) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)width * height * 3;
    if (idx >= n) return;
    uint32_t c = (uint32_t)(idx % 3);
    size_t p = idx / 3;
    uint32_t i = (uint32_t)(p % width), j = (uint32_t)(p / width);
    uint32_t band = j / band_rows, owner = band % owners, lb = band / owners;
    uint32_t rl = lb * band_rows + (j % band_rows);
    out[idx] = gathered[((size_t)owner * rows_local + rl) * width * 3 + (size_t)i * 3 + c];
}

// =============================================================================================
// host side
// =============================================================================================
static thread_local std::string g_last_error;
static int fail(int status, const std::string& msg) { g_last_error = msg; return status; }
int set_last_error(int status, const std::string& msg) { return fail(status, msg); } // for the library's other translation units (multi_gpu.cpp)

#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return fail(e_ == hipErrorOutOfMemory ? NRAYS_ERR_OOM : NRAYS_ERR_HIP,                        \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                               \
    } while (0)

} // namespace nrays

using namespace nrays;

namespace nrays {

template <typename T>
static int upload(NraysScene* sc, const std::vector<T>& v, const T** out) {
    *out = nullptr;
    if (v.empty()) return NRAYS_OK;
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, v.size() * sizeof(T)));
    sc->allocs.push_back(p);
    sc->scene_bytes += v.size() * sizeof(T);
    HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)p;
    return NRAYS_OK;
}

// Device-built segments first (device-to-device), then the host-built part: the layout HostScene's refs address.
template <typename T>
static int upload_joined(NraysScene* sc, const std::vector<std::pair<const T*, size_t>>& segs, const std::vector<T>& v, const T** out) {
    *out = nullptr;
    size_t total = v.size();
    for (const auto& s : segs) total += s.second;
    if (total == 0) return NRAYS_OK;
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, total * sizeof(T)));
    sc->allocs.push_back(p);
    sc->scene_bytes += total * sizeof(T);
    size_t at = 0;
    for (const auto& s : segs) { if (s.second) HIP_TRY(hipMemcpy((T*)p + at, s.first, s.second * sizeof(T), hipMemcpyDeviceToDevice)); at += s.second; }
    if (!v.empty()) HIP_TRY(hipMemcpy((T*)p + at, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)p;
    return NRAYS_OK;
}

static int ensure_queue(NraysScene* sc, uint32_t capacity) {
    if (capacity <= sc->queue_capacity) return NRAYS_OK;
    for (int k = 0; k < 2; ++k) {
        if (sc->queue[k].block) { (void)hipFree(sc->queue[k].block); sc->queue[k].block = nullptr; }
        size_t cap = capacity;
        size_t bytes = cap * (8 * 7 + 4 * 4 + 8);
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, bytes));
        sc->queue[k].block = p;
        char* c = (char*)p;
        RayQueue& q = sc->queue[k].q;
        for (int a = 0; a < 3; ++a) { q.o[a] = (double*)c; c += cap * 8; }
        for (int a = 0; a < 3; ++a) { q.d[a] = (double*)c; c += cap * 8; }
        q.refr = (double*)c; c += cap * 8;
        q.key = (unsigned long long*)c; c += cap * 8;
        q.energy = (float*)c; c += cap * 4;
        q.weight = (float*)c; c += cap * 4;
        q.pixel = (uint32_t*)c; c += cap * 4;
        q.depth = (uint32_t*)c; c += cap * 4;
    }
    sc->queue_capacity = capacity;
    return NRAYS_OK;
}

// Words of the cost-ordered work list (k_tile_order): entry k of XCD list x lives at order[8 k + x], and a list holds the wave
// tiles i = x (mod 8) — up to ceil(nwt / 8) of them — each as up to 2^lsl light-parallel parts.  The array therefore needs
// 8 * ceil(nwt / 8) << lsl words, not nwt << lsl: with nwt = 4 (mod 8) and every tile of one of the lists 0..3 split, the last
// entries of that list lie up to (4 << lsl) - 4 words beyond nwt << lsl.
static size_t order_slots(uint32_t nwt, uint32_t lsl) { return ((((size_t)nwt + 7u) / 8u) * 8u) << lsl; }

static uint32_t tile_rows(const NraysRenderParams* p) {
    if (p->band_rows == 0 || p->band_owners <= 1) return p->height;
    uint32_t nb = (p->height + p->band_rows - 1) / p->band_rows;
    return ((nb + p->band_owners - 1) / p->band_owners) * p->band_rows;
}

// The primary kernel is instantiated per feature set (primary_kernel.h: NR_PRIMARY_PERMUTATIONS, one translation unit per group);
// instrumented renders and k_bounce use the full-featured code (their results are identical, only slower).  The frame names the
// permutations it could run, most specialised first; the first one the library holds is launched (a tuning build holds few).
static bool primary_permutation_exists(int feat) { // (of the full build; a tuning build, NR_ONLY, may fall back to the full kernel)
#define X(G, S, F, P, O) if (!S && (F & ~(int)kFeatLdsScene) == feat) return true;
    NR_PRIMARY_PERMUTATIONS(X)
#undef X
    return false;
}
static void launch_primary(bool instrumented, int features, bool noxform, bool park, int occ, uint32_t grid, hipStream_t stream, const DScene& d, const DRender& R,
                           const QueueOut& qo, float* out, DeviceCounters* ctr, uint32_t* spill, uint32_t tx, uint32_t ty, uint32_t* work, uint32_t grab,
                           uint32_t* zero_counts, DeviceCounters* zero_ctr) {
    const PrimaryLaunch a{grid, stream, &d, &R, &qo, out, ctr, spill, tx, ty, work, grab, zero_counts, zero_ctr};
    auto launch = [&](bool stats, int feat, bool plain_, int occ_) {
        return launch_primary_group0(a, stats, feat, plain_, occ_) || launch_primary_group1(a, stats, feat, plain_, occ_) || launch_primary_group2(a, stats, feat, plain_, occ_) ||
               launch_primary_group3(a, stats, feat, plain_, occ_) || launch_primary_group4(a, stats, feat, plain_, occ_) || launch_primary_group5(a, stats, feat, plain_, occ_) ||
               launch_primary_group6(a, stats, feat, plain_, occ_);
    };
    if (instrumented) { launch(true, kFeatAll, false, 0); return; }
    // plain frames: no RNG keys, one sample per pixel
    const bool plain = R.window_width == 0.0 && !R.use_rng && R.first_batch && R.sample_begin == 0u && R.sample_end == 1u && R.width <= 16384u && R.height <= 16384u;
    const bool mesh_only = features == 2 || features == 6 || features == 18 || features == 22;
    if (occ == 3) { // the three-wave builds of the alpha-shadow mesh permutations: + kFeatNoXform when every BLAS is untransformed, + kFeatPark
        if (features == 6 || features == 22) { if (launch(false, features + (noxform ? (int)kFeatNoXform : 0) + (park ? (int)kFeatPark : 0), plain, 3)) return; }
        else if (features == 7 || features == 23) { if (launch(false, features, false, 3)) return; }
    }
    if (noxform && mesh_only && launch(false, features + (int)kFeatNoXform, plain, 0)) return;
    if (plain && launch(false, features, true, 0)) return;
    if (launch(false, features, false, 0)) return;
    launch(false, kFeatAll, false, 0); // bit 8 (double branching) only in the full kernels
}

// The ring's timing events are created by the first frame that records into a slot (1 024 hipEventCreate cost 0.6 ms of every scene creation; the
// first slots are created with the handle).  Every handle of the slot is checked: a creation that failed half-way is retried by the next frame.
static int ensure_ring_slot(NraysScene* sc, int slot) {
    hipEvent_t* ev[4] = {&sc->ev_begin[slot], &sc->ev_pbegin[slot], &sc->ev_pend[slot], &sc->ev_end[slot]};
    for (hipEvent_t* e : ev) if (!*e && hipEventCreate(e) != hipSuccess) { *e = nullptr; return fail(NRAYS_ERR_HIP, "event creation failed"); }
    return NRAYS_OK;
}

// analytic scenes: the sums / maxima k_tile_order reports per list, their pinned landing place and the event behind the read-back
static int alloc_cost_stats(NraysScene* sc) {
    HIP_TRY(hipMalloc((void**)&sc->d_cost_stats, 16 * sizeof(unsigned long long)));
    HIP_TRY(hipHostMalloc((void**)&sc->h_cost_stats, 16 * sizeof(unsigned long long), hipHostMallocDefault));
    HIP_TRY(hipEventCreateWithFlags(&sc->ev_stats, hipEventDisableTiming));
    return NRAYS_OK;
}

static int render_impl(NraysScene* sc, const NraysRenderParams* p, float* d_out, hipStream_t stream, bool instrumented, uint32_t count_flags = 0u) {
    if (!sc || !p || !d_out) return fail(NRAYS_ERR_BAD_ARG, "null argument");
    // NRAYS_HOST_TIMES=1 (read by nrays_scene_create): microseconds of host time this call spends up to a few marks, for the handle's first frames (tools/cold_probe.py)
    const bool host_times = sc->host_times && sc->frames_total < 4;
    const unsigned long long ht_frame = sc->frames_total;
    const auto ht0 = std::chrono::steady_clock::now();
    auto ht = [&](const char* what) { if (host_times) fprintf(stderr, "  render_impl frame %llu: +%.1f us %s\n", ht_frame, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ht0).count(), what); };
    if (p->ray_per_pixel == 0) return fail(NRAYS_ERR_BAD_ARG, "ray_per_pixel must be > 0 (scene.rs:37)");
    if (p->width == 0 || p->height == 0) return fail(NRAYS_ERR_BAD_ARG, "empty resolution");
    if (p->band_owners > 1 && (p->band_rows == 0 || p->band_owner >= p->band_owners)) return fail(NRAYS_ERR_BAD_ARG, "bad band parameters");
    HIP_TRY(hipSetDevice(sc->device));

    ht("hipSetDevice");
    const uint32_t rows = tile_rows(p);
    const uint64_t npix_local = (uint64_t)rows * p->width;
    if (npix_local >= (1ull << 31)) return fail(NRAYS_ERR_UNSUPPORTED, "tile too large");

    // sample batching keeps the number of primary rays (and hence continuation rays) per launch bounded
    // Continuation rays stay in registers (trace_chain); the HBM queue is only needed when one hit can
    // spawn both a reflection and a refraction.
    const bool queued = sc->host.any_double_branch;
    // Sample batching bounds the continuation rays one launch can append to that queue; a frame without a queue renders
    // all its samples in ONE launch (NRAYS_MAX_PRIMARY forces batching for the tests).
    const uint64_t kMaxPrimaryPerLaunch = sc->max_primary_per_launch;
    uint32_t batch = (queued || sc->max_primary_forced)
        ? (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(p->ray_per_pixel, kMaxPrimaryPerLaunch / std::max<uint64_t>(1, npix_local)))
        : p->ray_per_pixel;
    if (queued) {
        uint64_t want = std::min<uint64_t>(std::max<uint64_t>(4 * npix_local * batch, 1u << 16), 1ull << 27);
        int rc = ensure_queue(sc, (uint32_t)want);
        if (rc != NRAYS_OK) return rc;
        const size_t slots = (size_t)npix_local * 3;
        if (slots > sc->fixed_slots) {
            if (sc->d_fixed) { (void)hipFree(sc->d_fixed); sc->d_fixed = nullptr; sc->fixed_slots = 0; }
            HIP_TRY(hipMalloc((void**)&sc->d_fixed, slots * sizeof(long long)));
            HIP_TRY(hipMemsetAsync(sc->d_fixed, 0, slots * sizeof(long long), stream)); // k_fold_fixed leaves it cleared
            sc->fixed_slots = slots;
        }
        // a frame that failed between its k_bounce rounds and k_fold_fixed left sums behind: they must not reach this frame
        if (sc->fixed_dirty) { HIP_TRY(hipMemsetAsync(sc->d_fixed, 0, sc->fixed_slots * sizeof(long long), stream)); sc->fixed_dirty = false; }
    }
    if (sc->spill_entries && !sc->d_spill) {
        HIP_TRY(hipMalloc((void**)&sc->d_spill, (size_t)kMaxGrid * kBlock * sc->spill_entries * sizeof(uint32_t)));
    }

    DRender R; std::memset(&R, 0, sizeof R);
    R.width = p->width; R.height = p->height; R.rows_local = rows; R.spp = p->ray_per_pixel;
    R.max_depth = p->max_depth;
    R.band_rows = p->band_rows; R.band_owner = p->band_owner; R.band_owners = p->band_owners ? p->band_owners : 1;
    R.window_width = p->window_width; R.inv_width = 1.0 / (double)p->width; R.inv_height = 1.0 / (double)p->height;
    for (int a = 0; a < 3; ++a) R.eye[a] = p->camera_eye[a];
    for (int a = 0; a < 16; ++a) R.m[a] = p->inv_proj_view[a];
    R.seed = p->seed;
    { const ScreenBounds sb = sc->cull_enabled ? screen_bounds(sc->host, p) : ScreenBounds{INT32_MIN, INT32_MAX, INT32_MIN, INT32_MAX};
      R.cull_i0 = sb.i0; R.cull_i1 = sb.i1; R.cull_j0 = sb.j0; R.cull_j1 = sb.j1; }

    // lanes per pixel of an anti-aliased frame (sample-major mapping, see k_primary): the largest power of two <= min(batch, 64)
    uint32_t lane_log2 = 0;
    if (batch >= 2) { while (lane_log2 < 6u && (2u << lane_log2) <= batch) ++lane_log2; }
    if (sc->lane_log2_override >= 0) lane_log2 = std::min<uint32_t>((uint32_t)sc->lane_log2_override, lane_log2);
    R.lane_log2 = lane_log2;
    const uint32_t bwl = lane_log2 ? (7u - lane_log2) >> 1 : 4u, bhl = lane_log2 ? (6u - lane_log2) >> 1 : 4u; // pixel block of a scheduling unit
    const uint32_t tiles_x = (p->width + (1u << bwl) - 1) >> bwl, tiles_y = (rows + (1u << bhl) - 1) >> bhl;
    const uint32_t ntiles = lane_log2 ? (tiles_x * tiles_y + 3u) / 4u : tiles_x * tiles_y; // in units of four wave tiles
    // window of scheduling blocks that can see the scene (DRender::win_*); a block row of the compact buffer maps to
    // consecutive global rows as long as the bands are whole blocks high
    R.win_x0 = 0; R.win_nx = tiles_x; R.win_y0 = 0; R.win_ny = tiles_y;
    const bool banded = p->band_rows != 0 && R.band_owners > 1;
    if (batch >= p->ray_per_pixel && !instrumented && (!banded || p->band_rows % (1u << bhl) == 0) && (R.cull_i0 != INT32_MIN || R.cull_i1 != INT32_MAX)) {
        const int64_t i0 = std::max<int64_t>(R.cull_i0, 0), i1 = std::min<int64_t>(R.cull_i1, (int64_t)p->width - 1);
        uint32_t x0 = 0, nx = 0, y0 = 0, ny = 0;
        if (i0 <= i1) { x0 = (uint32_t)(i0 >> bwl); nx = (uint32_t)(i1 >> bwl) - x0 + 1u; }
        for (uint32_t by = 0; by < tiles_y && nx; ++by) {
            const uint32_t rl0 = by << bhl;
            const int64_t j0 = banded ? (int64_t)((rl0 / p->band_rows) * R.band_owners + R.band_owner) * p->band_rows + (rl0 % p->band_rows) : (int64_t)rl0;
            const int64_t j1 = std::min<int64_t>(j0 + (1 << bhl) - 1, (int64_t)p->height - 1);
            if (j0 > j1 || j1 < R.cull_j0 || j0 > R.cull_j1) continue; // padding rows / outside the bounds
            if (ny == 0) y0 = by;
            ny = by - y0 + 1u;
        }
        if (ny == 0) nx = 0;
        R.win_x0 = x0; R.win_nx = nx; R.win_y0 = y0; R.win_ny = ny;
    }
    const uint32_t win_units = R.win_nx * R.win_ny; // scheduling blocks inside the window
    uint32_t grab = sc->host.any_mesh ? 1u : 0u; // 0 = workgroup lists through LDS; the specialised kernels fix their path at compile time
    if (sc->grab_override >= 0) grab = (uint32_t)sc->grab_override; // tiles per dequeue of the mesh kernels, A/B only (NRAYS_GRAB); pixels do not depend on it
    // persistent grid: exactly the workgroups that can be resident (one 4-wave workgroup per CU per wave/SIMD)
    // Waves per SIMD of the alpha-shadow mesh permutations (k_primary's OCC): three for multi-light frames (their long tiles are split
    // into light-parallel parts, so the frame is bound by its sum) and for frames with many tiles per resident wave, two otherwise
    // (the frame is as long as its longest tile, and that tile's wave is fastest at two).  NRAYS_OCC overrides.
    int occ = 0;
    if (!instrumented && (sc->features == 6 || sc->features == 7 || sc->features == 22 || sc->features == 23) && lane_log2 == 0u) {
        const uint64_t wave_tiles = (uint64_t)ntiles * 4u, waves2 = (uint64_t)sc->num_cus * 8u;
        // (multi-light frames: from 6 wave tiles per resident wave on.  Round 5, after the shadow rays that are multiplied by 0 stopped being traced
        // (light_is_dark()): an owner's eighth of a 4K frame, 16 320 wave tiles, runs 1.22 - 1.29 ms at three waves against 1.40 - 1.44 at two, half a
        // 1080p frame 1.48 against 1.74; at 8 160 - 8 640 wave tiles the frame is as long as its longest split tile and two waves win, 1.10 / 0.97 ms
        // against 1.46 / 1.24: profiles/r05_rank_occupancy.log.  Round 4's threshold was 12: the eighth then ran 1.9 ms at two against 2.0 - 2.9.)
        const bool multi = (sc->features & kFeatMultiSample) && sc->light_lsl && sc->light_split_factor != 0.0f;
        // One light: from 14 wave tiles per resident wave on (round 5: the 1080p sponza stand-in, 32 640 wave tiles, 1.125 ms at three waves against 1.23 at two — its sum of
        // tile cycles per resident wave, 1.14 ms at two waves, had passed its longest tile, 0.89; at 1600 x 900, 22 800 wave tiles, the longest tile still leads and two waves
        // win, 0.93 against 1.12: profiles/r05_rank_occupancy.log.  Round 4's threshold was 24.)
        // (with the long tiles of one-light frames split by pixels, NR_PIXEL_SPLIT, the longest tile stops leading earlier: 22 800 wave tiles 0.86 ms at three waves against 0.95,
        // 14 400 wave tiles 0.75 against 0.73 — from 9 on)
        occ = wave_tiles >= (multi ? 6u : (NR_PIXEL_SPLIT && sc->light_lsl ? 9u : 14u)) * waves2 ? 3 : 0;
        if (sc->occ_override >= 0) occ = sc->occ_override == 3 ? 3 : 0;
    }
    uint32_t grid_primary = std::min<uint32_t>(std::min<uint32_t>(((ntiles + 7u) / 8u) * 8u, (uint32_t)kMaxGrid),
                                                     (uint32_t)sc->num_cus * (uint32_t)(occ ? NR_OCC3_AS : waves_per_simd(instrumented ? kFeatAll : sc->features)) * 256u / (uint32_t)kBlock);
    if (sc->grid_wg_per_cu > 0) grid_primary = std::min<uint32_t>(grid_primary, (uint32_t)sc->num_cus * (uint32_t)sc->grid_wg_per_cu); // NRAYS_GRID_WG_PER_CU: occupancy sensitivity runs

    // All per-handle state (double-buffered counters, queues, raygen tables, tile costs) assumes that the renders of one
    // handle execute one after the other: a render on a different stream than its predecessor is ordered behind it.
    if (sc->have_last && sc->last_stream != stream) {
        if (sc->last_timed && sc->last_done) HIP_TRY(hipStreamWaitEvent(stream, sc->last_done, 0));
        else { // the previous frame recorded no event (event_stride): mark the end of ITS stream now and wait on that — no host stall
            if (!sc->ev_switch) HIP_TRY(hipEventCreateWithFlags(&sc->ev_switch, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(sc->ev_switch, sc->last_stream));
            HIP_TRY(hipStreamWaitEvent(stream, sc->ev_switch, 0));
        }
    }
    ht("parameters, screen bounds, window");
    const bool timed = instrumented || (sc->frames_total % sc->event_stride) == 0;
    sc->frames_total++;
    const int slot = (int)(sc->frames_recorded % NraysScene::kRing);
    if (timed) { const int rc_ = ensure_ring_slot(sc, slot); if (rc_ != NRAYS_OK) return rc_; }
    // events: [pbegin .. pend] brackets the first primary launch; the frame spans [pbegin .. end], and
    // `end` is only recorded separately when something follows the primary kernel
    // The staged ("wavefront") form of the trace loop (wavefront.hip) renders this frame instead of k_primary when the scene is eligible and
    // NRAYS_WAVEFRONT / the library's rule say so; pixels are identical either way.
    const bool staged = !instrumented && wavefront_wanted(sc, p, lane_log2);
    const bool single_launch = !staged && !instrumented && !queued && p->ray_per_pixel <= batch && p->ray_per_pixel == 1;
    sc->d_counters = sc->d_counters_set[sc->frame_index & 1];
    DeviceCounters* next_ctr = sc->d_counters_set[(sc->frame_index + 1) & 1];
    sc->frame_index++;
    R.use_rng = (p->window_width != 0.0 || sc->host.any_area_light) ? 1u : 0u;
    // mesh scenes: longest-processing-time-first from the previous frame of the same geometry (pixels do not depend on it)
    R.tile_cost = nullptr; R.tile_order = nullptr;
    sc->has_prepass[slot] = false;
    // (not for the sample-major frames of anti-aliased renders: their wave tiles are a few pixels each — 8 M of them for config 5 —
    // and far more even; recording, sorting and following the order costs more than the tail it removes: hairball 4K 64 spp
    // 251 -> 222 ms without it, sponza 1080p 4 / 16 / 64 spp 2-4 %, profiles/r02_aa_lpt.log)
    if (staged) {
        sc->d_counts = sc->d_counts_set[sc->launch_index & 1];
        uint32_t* next_counts = sc->d_counts_set[(sc->launch_index + 1) & 1];
        sc->launch_index++;
        const int rc = wavefront_render(sc, p, R, d_out, stream, tiles_x, tiles_y, timed, slot, next_ctr, next_counts);
        if (rc != NRAYS_OK) return rc;
    } else {
    // ---- per-camera scheduling state (pixels never depend on it) -----------------------------------------------------------------
    // The reference's caller renders every camera ONCE (examples/loader3d.rs:67-93), an interactive caller moves it a little every
    // frame: what a frame may cost besides its tiles is decided here.
    //   resting camera    the order recorded for it is reused; nothing is recorded, nothing sorted;
    //   nearby camera     (shift of the view since the order's camera below kNearPixels, cam_shift_px()) the order is reused as it is for
    //                     up to kMaxOrderAge frames; the last of them records its tile costs, the next one sorts them (ONE k_tile_order,
    //                     which also clears the cost array) — a moving camera pays the sort every kMaxOrderAge + 1 frames;
    //   cold camera       no usable history: mesh scenes guess (k_seed_costs + k_tile_order), analytic scenes run image-order lists;
    //                     the frame records its costs, its successor sorts them.
    const uint64_t sched_key = (((uint64_t)p->width << 40) ^ ((uint64_t)rows << 20) ^ ((uint64_t)p->band_rows << 8) ^ ((uint64_t)p->band_owner << 4) ^ (uint64_t)R.band_owners ^ ((uint64_t)lane_log2 << 60))
                               + 0x9E3779B97F4A7C15ull * (((uint64_t)R.win_x0 << 48) ^ ((uint64_t)R.win_nx << 32) ^ ((uint64_t)R.win_y0 << 16) ^ (uint64_t)R.win_ny);
    uint64_t cam = 0xcbf29ce484222325ull; // FNV-1a over everything a tile's cost depends on besides the scene (which a handle never changes)
    { auto mix = [&](const void* q, size_t n) { const unsigned char* b_ = (const unsigned char*)q; for (size_t i = 0; i < n; ++i) { cam ^= b_[i]; cam *= 0x100000001b3ull; } };
      mix(p->inv_proj_view, sizeof p->inv_proj_view); mix(p->camera_eye, sizeof p->camera_eye); mix(&p->window_width, sizeof p->window_width);
      mix(&p->ray_per_pixel, sizeof p->ray_per_pixel); mix(&p->max_depth, sizeof p->max_depth); }
    const CamSnap snap = cam_snapshot(sc->host, p);
    auto near_cam = [&](const CamSnap& other) { return sc->near_reuse && cam_shift_px(other, snap) <= sc->near_pixels; };
    const uint32_t kMaxOrderAge = sc->max_order_age;
    bool lpt = grab >= 1u && lane_log2 == 0u;
    lpt = lpt && sc->lpt_enabled; // A/B switch (NRAYS_LPT=0)
    if (instrumented && sc->light_lsl) lpt = false; // the instrumented kernel does not decode the split entries a plain frame's order may hold
    if (lpt) {
        const uint32_t nwt = std::max<uint32_t>(1u, lane_log2 ? win_units : win_units * 4u);
        // light-parallel tiles (DRender::light_lsl): multi-light mesh scenes; the order array then holds up to 2^lsl entries per tile
        // (one-light frames, NR_PIXEL_SPLIT: only while a single tile can lead the frame — below 24 wave tiles per resident wave at two waves per SIMD; beyond, no tile comes near
        // the frame's work per wave and the split machinery costs 0.7 %: profiles/r05_pixel_split_ab.log)
        const bool pixel_split_only = sc->light_lsl && !(sc->features & kFeatMultiSample);
        const uint32_t split_lsl = (sc->light_lsl && sc->light_split_factor != 0.0f && !instrumented && !(pixel_split_only && (uint64_t)nwt >= 24ull * (uint64_t)sc->num_cus * 8ull)) ? sc->light_lsl : 0u;
        if (nwt > sc->tile_slots) {
            if (sc->d_tile_cost) { (void)hipFree(sc->d_tile_cost); sc->d_tile_cost = nullptr; }
            if (sc->d_tile_order) { (void)hipFree(sc->d_tile_order); sc->d_tile_order = nullptr; }
            sc->tile_slots = 0; sc->cost_valid = false; sc->order_valid = false;
            HIP_TRY(hipMalloc((void**)&sc->d_tile_cost, (size_t)nwt * sizeof(uint32_t)));
            HIP_TRY(hipMalloc((void**)&sc->d_tile_order, order_slots(nwt, sc->light_lsl) * sizeof(uint32_t)));
            sc->tile_slots = nwt;
        }
        if (split_lsl && !sc->d_order_len) HIP_TRY(hipMalloc((void**)&sc->d_order_len, 8 * sizeof(uint32_t)));
        R.light_lsl = split_lsl; R.order_len = split_lsl ? sc->d_order_len : nullptr;
        const uint64_t key = sched_key ^ ((uint64_t)split_lsl << 56); // (an order that holds split entries is not one without them)
        const bool order_here = sc->order_valid && sc->order_key == key && !sc->order_seeded && sc->lpt_reuse;
        bool record = false;
        if (order_here && sc->order_cam == cam) {
            R.tile_order = sc->d_tile_order; // resting camera
        } else if (order_here && kMaxOrderAge != 0u && sc->order_age < kMaxOrderAge && near_cam(sc->order_snap)) {
            R.tile_order = sc->d_tile_order; // nearby camera: the order as it is
            record = ++sc->order_age == kMaxOrderAge;
            if (record && split_lsl) HIP_TRY(hipMemsetAsync(sc->d_tile_cost, 0, (size_t)nwt * sizeof(uint32_t), stream)); // split entries record by atomicMax (rare frame: every kMaxOrderAge-th)
        } else {
            const bool costs_here = sc->cost_valid && sc->cost_key == key && (sc->cost_cam == cam || near_cam(sc->cost_snap));
            // no history for this view: a first guess from the boxes of the nodes that can continue a chain (k_seed_costs)
            const bool seeded = !costs_here && sc->seed_enabled && sc->seed_boxes != 0u && win_units > 0;
            if (seeded) {
                if (timed) HIP_TRY(hipEventRecord(sc->ev_begin[slot], stream));
                hipLaunchKernelGGL(k_seed_costs, dim3((nwt + 255u) / 256u), dim3(256), 0, stream, R, (const float*)sc->d_seed_boxes, sc->seed_boxes, sc->d_tile_cost, nwt, sc->seed_rays);
                HIP_TRY(hipGetLastError());
#ifdef NR_DEBUG_TILE_COSTS
                if (!sc->d_seed_copy) HIP_TRY(hipMalloc((void**)&sc->d_seed_copy, (size_t)sc->tile_slots * sizeof(uint32_t)));
                HIP_TRY(hipMemcpyAsync(sc->d_seed_copy, sc->d_tile_cost, (size_t)nwt * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream)); // tools/tile_dump.py: the guess beside the recorded costs
#endif
            }
            if (seeded || costs_here) {
                if (timed && !seeded) HIP_TRY(hipEventRecord(sc->ev_begin[slot], stream));
                sc->has_prepass[slot] = true;
                hipLaunchKernelGGL(k_tile_order, dim3(8), dim3(1024), 0, stream, sc->d_tile_cost, sc->d_tile_order, nwt, (unsigned long long*)nullptr,
                                   split_lsl, sc->light_split_factor, grid_primary * (uint32_t)(kBlock / 64), split_lsl ? sc->d_order_len : (uint32_t*)nullptr, split_lsl ? 1u : 0u, sc->split_hyst);
                HIP_TRY(hipGetLastError());
                R.tile_order = sc->d_tile_order;
                sc->order_valid = true; sc->order_key = key; sc->order_seeded = seeded; sc->order_age = 0;
                if (!seeded) { sc->order_cam = sc->cost_cam; sc->order_snap = sc->cost_snap; }
            } else if (split_lsl) HIP_TRY(hipMemsetAsync(sc->d_tile_cost, 0, (size_t)nwt * sizeof(uint32_t), stream));
            // a guessed order is replaced by the recorded one on the next frame; an order sorted from a NEARBY camera's costs serves this
            // one as it is (it ages like any other).  The frame that sorts its OWN camera's costs records once more: under the order it will
            // keep (nrays_get_tile_costs reports these).
            record = seeded || !costs_here || !sc->lpt_reuse || sc->cost_cam == cam || kMaxOrderAge == 0u;
            if (!record) sc->cost_valid = false; // consumed (and, with split entries, cleared) by the sort
        }
        if (R.tile_order) grab = 1u;
        if (record) { R.tile_cost = sc->d_tile_cost; sc->cost_key = key; sc->cost_cam = cam; sc->cost_snap = snap; sc->cost_valid = true; }
    }
    // Analytic scenes (workgroup lists): the frames are a few hundred long tiles (deep reflection chains, ~10^5 cycles each) among
    // thousands of short ones, and a long tile runs ~1.5x faster when it does not share its SIMD with another long one.  The first
    // frame of a camera records the tile costs, the second sorts them (k_tile_order) and reads back their sum and maximum; when the
    // frame's parallelism sum / max is below ~1.5 waves per SIMD of the chip, the following frames of that camera are rendered from
    // the cost order with the long tiles on the first workgroup of each CU (DRender::lead_wgs, k_primary) — otherwise image order,
    // as before (profiles/r02_analytic_lpt.log: balls 70.0 -> 52.4 us; primitives, whose every tile is long, stays at 201 us).
    if (grab == 0u && sc->lpt_analytic && !instrumented && win_units > 0) {
        const uint32_t nwt = lane_log2 ? win_units : win_units * 4u;
        if (nwt > sc->tile_slots) {
            if (sc->d_tile_cost) { (void)hipFree(sc->d_tile_cost); sc->d_tile_cost = nullptr; }
            if (sc->d_tile_order) { (void)hipFree(sc->d_tile_order); sc->d_tile_order = nullptr; }
            sc->tile_slots = 0; sc->cost_valid = false; sc->order_valid = false;
            HIP_TRY(hipMalloc((void**)&sc->d_tile_cost, (size_t)nwt * sizeof(uint32_t)));
            HIP_TRY(hipMalloc((void**)&sc->d_tile_order, (size_t)nwt * sizeof(uint32_t)));
            sc->tile_slots = nwt;
        }
        if (!sc->d_cost_stats) { int rc_ = alloc_cost_stats(sc); if (rc_ != NRAYS_OK) return rc_; }
        const uint64_t key = sched_key;
        const hipError_t stats_ready = sc->stats_pending ? hipEventQuery(sc->ev_stats) : hipErrorNotReady;
        if (sc->stats_pending && stats_ready != hipSuccess) (void)hipGetLastError(); // "not ready" must not surface as the launch error checked below
        if (sc->stats_pending && stats_ready == hipSuccess) { // the sums / maxima of the last sort's eight lists have arrived
            double sum = 0.0, mx = 0.0;
            for (int x = 0; x < 8; ++x) { sum += (double)sc->h_cost_stats[2 * x]; mx = std::max(mx, (double)sc->h_cost_stats[2 * x + 1]); }
            sc->lone_waves = mx > 0.0 && sum / mx < sc->lone_factor * 4.0 * (double)sc->num_cus;
            sc->lone_known = true; sc->lone_key = sc->stats_key;
            sc->stats_pending = false;
        }
        auto sort_costs = [&]() -> int {
            if (sc->stats_pending) HIP_TRY(hipEventSynchronize(sc->ev_stats)); // (a camera that changes every few frames: the previous read-back is long done)
            if (timed) HIP_TRY(hipEventRecord(sc->ev_begin[slot], stream));
            sc->has_prepass[slot] = true;
            hipLaunchKernelGGL(k_tile_order, dim3(8), dim3(1024), 0, stream, sc->d_tile_cost, sc->d_tile_order, nwt, sc->d_cost_stats, 0u, 0.0f, 0u, (uint32_t*)nullptr, 0u, 1.0f);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(sc->h_cost_stats, sc->d_cost_stats, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipEventRecord(sc->ev_stats, stream));
            sc->stats_pending = true; sc->stats_key = key;
            sc->order_valid = true; sc->order_key = key; sc->order_cam = sc->cost_cam; sc->order_snap = sc->cost_snap; sc->order_age = 0;
            return NRAYS_OK;
        };
        const bool order_here = sc->order_valid && sc->order_key == key;
        bool record = false;
        if (order_here && sc->order_cam == cam) {
            // steady state of a resting camera: nothing recorded, nothing sorted
        } else if (order_here && sc->order_age < kMaxOrderAge && near_cam(sc->order_snap)) {
            record = ++sc->order_age == kMaxOrderAge; // nearby camera: the order as it is; its last frame records for the re-sort
        } else if (sc->cost_valid && sc->cost_key == key && (sc->cost_cam == cam || near_cam(sc->cost_snap))) {
            const int rc = sort_costs(); if (rc != NRAYS_OK) return rc; // the frame after a recording one
        } else {
            sc->order_valid = false; // a cold camera: image-order lists, costs recorded
            record = true;
        }
        if (record) { R.tile_cost = sc->d_tile_cost; sc->cost_key = key; sc->cost_cam = cam; sc->cost_snap = snap; sc->cost_valid = true; }
        // (while the sums of a re-sort are on their way the decision of the previous sort of this geometry stands)
        if (sc->order_valid && sc->order_key == key && sc->lone_known && sc->lone_key == key && sc->lone_waves) {
            R.tile_order = sc->d_tile_order;
            if (sc->lead_mode) { R.lead_wgs = std::min<uint32_t>(grid_primary, (uint32_t)sc->num_cus); R.lead_entries = R.lead_wgs * (uint32_t)sc->lead_per_wg; } // two workgroups per CU: one of them owns the long tiles
            else grid_primary = std::min<uint32_t>(grid_primary, (uint32_t)sc->num_cus);             // NRAYS_LEAD_WGS=0: one workgroup per CU
        }
#ifdef NR_DEBUG_TILE_COSTS
        if (getenv("NRAYS_DEBUG_RECORD_ALWAYS")) R.tile_cost = sc->d_tile_cost; // tools/tile_costs.py: the costs of the steady-state frames
#endif
    }
#ifdef NR_DEBUG_TILE_COSTS
    if (!sc->d_wave_times) HIP_TRY(hipMalloc((void**)&sc->d_wave_times, (size_t)kMaxGrid * (kBlock / 64) * 8 * sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(sc->d_wave_times, 0, (size_t)kMaxGrid * (kBlock / 64) * 8 * sizeof(uint32_t), stream));
    R.wave_times = sc->d_wave_times; sc->dbg_grid = grid_primary; R.dbg_mode = getenv("NRAYS_DEBUG_WAVE_WORK") ? (uint32_t)atoi(getenv("NRAYS_DEBUG_WAVE_WORK")) : 0u;
#endif
    if (R.tile_cost) { sc->cost_tiles = lane_log2 ? win_units : win_units * 4u; sc->cost_grid = grid_primary; sc->cost_split_lsl = R.light_lsl; }
    R.cost_meta = sc->d_cost_meta; // (read by the instrumented kernel only)
    ht("scheduling state (seed / sort launches)");
    bool first_primary = true;
    for (uint32_t s0 = 0; s0 < p->ray_per_pixel; s0 += batch) {
        R.sample_begin = s0; R.sample_end = std::min<uint32_t>(p->ray_per_pixel, s0 + batch);
        R.first_batch = s0 == 0 ? 1u : 0u;
        sc->d_counts = sc->d_counts_set[sc->launch_index & 1];
        uint32_t* next_counts = sc->d_counts_set[(sc->launch_index + 1) & 1];
        sc->launch_index++;
        QueueOut qo; qo.q = sc->queue[1].q; qo.capacity = queued ? sc->queue_capacity : 0; qo.count = sc->d_counts + 1;
        qo.overflow = &sc->d_counters->overflow;
        if (first_primary && timed) HIP_TRY(hipEventRecord(sc->ev_pbegin[slot], stream));
        if (first_primary) ht("event record before the launch");
        // a launch that records its tile costs is timed (nrays_get_tile_costs: NraysTileCosts::kernel_ms): by the ring's events when the frame has them, by a pair of its own otherwise
        const bool rec_events = first_primary && R.tile_cost && !timed && sc->ev_rec[0] && sc->ev_rec[1];
        if (first_primary && R.tile_cost) { sc->rec_events_valid = rec_events; sc->rec_slot = timed ? slot : -1; }
        if (rec_events) HIP_TRY(hipEventRecord(sc->ev_rec[0], stream));
        DScene dsc = sc->d;
        if (instrumented && (count_flags & NRAYS_COUNT_AS_TIMED) && !sc->d.no_elide) { // what the scene's plain kernel skips (trace_device.h: light_is_dark everywhere; shade_hit in the alpha-mapped mesh kernels)
            const int f = primary_permutation_exists(sc->features & ~(int)kFeatLdsScene) ? sc->features : (int)kFeatAll; // the FEAT a plain frame of this scene is launched with
            dsc.stats_elide = 1u | (((f & kFeatMesh) && (f & kFeatAlphaShadow)) ? 2u : 0u);
        }
        // (a scene with a non-finite light / colour / texel: every frame by the kernel that skips nothing)
        launch_primary(instrumented || sc->d.no_elide != 0u, sc->features, sc->noxform, sc->park, occ, grid_primary, stream, dsc, R, qo, d_out, sc->d_counters, sc->d_spill, tiles_x, tiles_y, sc->d_counts + kMaxGenerations + 2, grab, next_counts, R.first_batch ? next_ctr : nullptr);
        HIP_TRY(hipGetLastError());
        if (first_primary) ht("k_primary launch");
        if (rec_events) HIP_TRY(hipEventRecord(sc->ev_rec[1], stream));
        if (first_primary) {
            if (timed) HIP_TRY(hipEventRecord(sc->ev_pend[slot], stream));
            if (instrumented) HIP_TRY(hipMemcpyAsync(sc->d_counters_primary, sc->d_counters, sizeof(DeviceCounters), hipMemcpyDeviceToDevice, stream));
            first_primary = false;
        }
        // rounds of queued second children (host-controlled: the count is read back after every round)
        // k_bounce takes its ray count from device memory and strides over it, so a round can be launched without knowing the
        // count: the host only looks (one small copy + a stream synchronisation) before every FOURTH round — to stop, and to size
        // that group's grids — instead of before every round; a group's later rounds may find an empty queue and return at once.
        uint32_t n_seen = 0;
        bool folded = true;
        for (uint32_t r = 1; queued && r <= (uint32_t)kMaxGenerations; ++r) {
            if ((r - 1u) % 4u == 0u) {
                HIP_TRY(hipMemcpyAsync(&n_seen, sc->d_counts + r, sizeof n_seen, hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
                if (n_seen == 0) break;
            }
            uint32_t launch_n = std::min<uint32_t>(n_seen, sc->queue_capacity); // the group's first count bounds nothing, it only sizes the grid
            uint32_t grid = std::max<uint32_t>(std::min<uint32_t>((launch_n + kBlock - 1) / kBlock, kMaxGrid), std::min<uint32_t>((uint32_t)sc->num_cus, kMaxGrid));
            QueueOut qn; qn.q = sc->queue[(r + 1) & 1].q; qn.capacity = sc->queue_capacity; qn.count = sc->d_counts + r + 1;
            qn.overflow = &sc->d_counters->overflow;
            if (instrumented || sc->d.no_elide) hipLaunchKernelGGL(k_bounce<true>, dim3(grid), dim3(kBlock), 0, stream, dsc, sc->queue[r & 1].q, sc->d_counts + r, sc->queue_capacity, qn, sc->d_fixed, sc->d_counters, sc->d_spill, p->max_depth);
            else hipLaunchKernelGGL(k_bounce<false>, dim3(grid), dim3(kBlock), 0, stream, sc->d, sc->queue[r & 1].q, sc->d_counts + r, sc->queue_capacity, qn, sc->d_fixed, sc->d_counters, sc->d_spill, p->max_depth);
            HIP_TRY(hipGetLastError());
            folded = false; sc->fixed_dirty = true;
        }
        if (queued && !folded) { // the next batch's k_primary continues the running sums in d_out: fold this batch's queued chains in first
            const size_t n = (size_t)npix_local * 3;
            hipLaunchKernelGGL(k_fold_fixed, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_out, sc->d_fixed, n);
            HIP_TRY(hipGetLastError());
            folded = true; sc->fixed_dirty = false;
        }
    }
    } // !staged
    if (p->ray_per_pixel > 1) {
        size_t n = (size_t)npix_local * 3;
        hipLaunchKernelGGL(k_resolve, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_out, n, (float)p->ray_per_pixel);
        HIP_TRY(hipGetLastError());
    }
    if (!single_launch && timed) HIP_TRY(hipEventRecord(sc->ev_end[slot], stream));
    sc->last_timed = timed;
    if (timed) {
        sc->single_launch[slot] = single_launch;
        sc->last_done = single_launch ? sc->ev_pend[slot] : sc->ev_end[slot];
        sc->frames_recorded++;
    }
    sc->last_stream = stream; sc->have_last = true;
    ht("end (event records after the launch)");
    // owned rows only (padding rows of the last band carry no rays)
    uint64_t owned_rows = 0;
    if (p->band_rows == 0 || p->band_owners <= 1) owned_rows = p->height;
    else for (uint32_t j = 0; j < p->height; ++j) if (((j / p->band_rows) % p->band_owners) == p->band_owner) ++owned_rows;
    sc->last_primary = owned_rows * p->width * p->ray_per_pixel;
    sc->last_primary_first_batch = owned_rows * p->width * std::min<uint32_t>(batch, p->ray_per_pixel);
    sc->last_instrumented = instrumented;
    return NRAYS_OK;
}

} // namespace nrays

extern "C" {

uint32_t nrays_abi_version(void) { return NRAYS_ABI_VERSION; }
const char* nrays_last_error(void) { return g_last_error.c_str(); }
uint32_t nrays_tile_rows(const NraysRenderParams* params) { return params ? tile_rows(params) : 0; }
uint64_t nrays_scene_device_bytes(const NraysScene* scene) { return scene ? scene->scene_bytes : 0; }

int nrays_scene_create(const NraysSceneDesc* desc, NraysScene** out_scene) {
    if (!desc || !out_scene) return fail(NRAYS_ERR_BAD_ARG, "null argument");
    *out_scene = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(NRAYS_ERR_NO_DEVICE, "no HIP device visible");
    NraysScene* sc = new (std::nothrow) NraysScene();
    if (!sc) return fail(NRAYS_ERR_OOM, "host allocation failed");
    auto bail = [&](int rc) { nrays_scene_destroy(sc); return rc; };
    if (hipGetDevice(&sc->device) != hipSuccess) return bail(fail(NRAYS_ERR_HIP, "hipGetDevice failed"));
    std::string err;
    const auto t_create0 = std::chrono::steady_clock::now();
    auto t_stage = t_create0;
    const bool stage_times = getenv("NRAYS_BUILD_TIMES") != nullptr;
    auto stage = [&](const char* what) { // NRAYS_BUILD_TIMES: where nrays_scene_create spends its time
        if (!stage_times) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "  nrays_scene_create: %s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_stage).count());
        t_stage = now;
    };
    int rc = build_host_scene(desc, sc->host, err);
    stage("build_host_scene (BLAS + TLAS builds)");
    if (rc != NRAYS_OK) return bail(fail(rc, err));
    HostScene& h = sc->host;
    const auto t_create1 = std::chrono::steady_clock::now();
    std::memset(&sc->d, 0, sizeof sc->d);
    if (h.dev_blas.size() == 1 && h.tris.empty() && h.triuvs.empty() && h.dev_blas[0].nodes && h.dev_blas[0].num_nodes + h.nodes.size() <= h.dev_blas[0].node_capacity) {
        // ONE device-built BLAS and nothing but TLAS nodes from the host (a single large mesh): the builder's arrays ARE the scene's arrays — the host
        // nodes go into the spare slots behind the BLAS; no second allocation, no copy and no release of gigabytes (0.2 s for the hairball stand-in)
        nrays::DeviceBlas& b = h.dev_blas[0];
        if (!h.nodes.empty() && hipMemcpy(b.nodes + b.num_nodes, h.nodes.data(), h.nodes.size() * sizeof(BvhNode), hipMemcpyHostToDevice) != hipSuccess)
            return bail(fail(NRAYS_ERR_HIP, "upload of the TLAS nodes failed"));
        sc->d.nodes = b.nodes; sc->d.tris = b.tris; sc->d.triuvs = b.uvs;
        sc->allocs.push_back(b.nodes); sc->allocs.push_back(b.tris); sc->allocs.push_back(b.uvs);
        sc->scene_bytes += (b.num_nodes + h.nodes.size()) * sizeof(BvhNode) + b.num_refs * (sizeof(TriRec) + sizeof(TriUv));
        b.nodes = nullptr; b.tris = nullptr; b.uvs = nullptr;
        h.dev_blas.clear();
    } else {
        std::vector<std::pair<const BvhNode*, size_t>> nseg; std::vector<std::pair<const TriRec*, size_t>> tseg; std::vector<std::pair<const TriUv*, size_t>> useg;
        for (const nrays::DeviceBlas& b : h.dev_blas) { nseg.push_back({b.nodes, b.num_nodes}); tseg.push_back({b.tris, b.num_refs}); useg.push_back({b.uvs, b.num_refs}); }
        if ((rc = upload_joined(sc, nseg, h.nodes, &sc->d.nodes)) != NRAYS_OK) return bail(rc);
        if ((rc = upload_joined(sc, tseg, h.tris, &sc->d.tris)) != NRAYS_OK) return bail(rc);
        if ((rc = upload_joined(sc, useg, h.triuvs, &sc->d.triuvs)) != NRAYS_OK) return bail(rc);
        for (nrays::DeviceBlas& b : h.dev_blas) nrays::free_device_blas(b);
        h.dev_blas.clear();
    }
    {   // the scene's small record arrays: ONE allocation and ONE copy (eight synchronous hipMalloc + hipMemcpy pairs before)
        struct Part { const void* src; size_t bytes; const void** out; size_t at; };
        std::vector<Part> parts;
        size_t total = 0;
        auto add = [&](const auto& v, auto** out) {
            *out = nullptr;
            if (v.empty()) return;
            total = (total + 255u) & ~(size_t)255u;
            parts.push_back(Part{v.data(), v.size() * sizeof(v[0]), (const void**)out, total});
            total += v.size() * sizeof(v[0]);
        };
        add(h.instances, &sc->d.instances); add(h.shadow_instances, &sc->d.shadow_instances); add(h.links, &sc->d.links); add(h.shadow_links, &sc->d.shadow_links);
        add(h.node_aabbs, &sc->d.node_aabbs); add(h.lights, &sc->d.lights); add(h.planes, &sc->d.planes); add(h.shadow_planes, &sc->d.shadow_planes);
        if (total) {
            void* blk = nullptr;
            if (hipMalloc(&blk, total) != hipSuccess) return bail(fail(NRAYS_ERR_OOM, "record allocation failed"));
            sc->allocs.push_back(blk); sc->scene_bytes += total;
            std::vector<char> stage(total);
            for (const Part& pt : parts) { std::memcpy(stage.data() + pt.at, pt.src, pt.bytes); *pt.out = (const char*)blk + pt.at; }
            if (hipMemcpy(blk, stage.data(), total, hipMemcpyHostToDevice) != hipSuccess) return bail(fail(NRAYS_ERR_HIP, "record upload failed"));
        }
    }
    {   // the elisions (trace_device.h: light_is_dark, shade_hit) need x * 0 == 0 for everything they skip: any non-finite light, material colour or float texel, or a
        // negative shininess (0 * inf), switches them off for this scene (phong_material.rs:109-141, scene.rs:179-190 then produce NaN, and so do we)
        bool finite = true;
        for (const LightRec& l : h.lights) { for (int a = 0; a < 3; ++a) finite = finite && std::isfinite(l.pos[a]) && std::isfinite(l.color[a]); finite = finite && std::isfinite(l.radius); }
        for (const ShadeRec& m : h.shade) {
            for (int a = 0; a < 3; ++a) finite = finite && std::isfinite(m.ka[a]) && std::isfinite(m.kd[a]) && std::isfinite(m.ks[a]);
            finite = finite && std::isfinite(m.shininess) && m.shininess >= 0.0f && std::isfinite(m.alpha) && std::isfinite(m.refl_mix) && std::isfinite(m.refl_atenuation) && std::isfinite(m.refr_coeff);
        }
        for (const HostTexture& t : h.textures) {
            if (t.rec.format != NRAYS_TEXEL_RGBA32F) continue;
            const float* f = (const float*)t.bytes.data();
            for (size_t i = 0, n = t.bytes.size() / sizeof(float); i < n && finite; ++i) finite = std::isfinite(f[i]);
        }
        for (int a = 0; a < 3; ++a) finite = finite && std::isfinite(h.background[a]);
        const char* e = getenv("NRAYS_ELIDE"); // =0: never (A/B)
        sc->d.no_elide = (!finite || (e && atoi(e) == 0)) ? 1u : 0u;
        // (such a scene's frames are rendered by the instrumented kernel, which does not decode the split entries of a cost-ordered list: no light-parallel / pixel-split tiles — light_lsl stays 0 below)
    }
    std::vector<TextureRec> trecs;
    for (HostTexture& t : h.textures) {
        void* p = nullptr;
        if (hipMalloc(&p, t.bytes.size()) != hipSuccess) return bail(fail(NRAYS_ERR_OOM, "texture allocation failed"));
        sc->allocs.push_back(p);
        sc->scene_bytes += t.bytes.size();
        if (hipMemcpy(p, t.bytes.data(), t.bytes.size(), hipMemcpyHostToDevice) != hipSuccess) return bail(fail(NRAYS_ERR_HIP, "texture upload failed"));
        TextureRec r = t.rec; r.texels = p; trecs.push_back(r);
        std::vector<uint8_t>().swap(t.bytes);
    }
    for (size_t i = 0; i < h.shade.size(); ++i) { // patch the device texel pointers into the per-node shading records
        if (h.shade_tex[i] >= 0) h.shade[i].tex.texels = trecs[h.shade_tex[i]].texels;
        if (h.shade_alpha_tex[i] >= 0) h.shade[i].alpha_tex.texels = trecs[h.shade_alpha_tex[i]].texels;
    }
    if ((rc = upload(sc, h.shade, &sc->d.shade)) != NRAYS_OK) return bail(rc);
    stage("uploads (nodes, triangles, records, textures)");
    if (getenv("NRAYS_BUILD_TIMES") && h.tris.size() + h.dev_tris > 1000000)
        fprintf(stderr, "  nrays_scene_create: build_host_scene %.2f s, uploads %.2f s\n", std::chrono::duration<double>(t_create1 - t_create0).count(),
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_create1).count());
    sc->d.closest_root = h.closest_root; sc->d.shadow_root = h.shadow_root;
    sc->d.num_planes = (uint32_t)h.planes.size(); sc->d.num_lights = (uint32_t)h.lights.size();
    for (int a = 0; a < 3; ++a) sc->d.background[a] = h.background[a];
    {   // NRAYS_NODE_QUORUM=0 keeps every node phase running until its last lane holds a leaf (A/B switch)
        const char* e = getenv("NRAYS_NODE_QUORUM");
        sc->d.incoherent = h.any_incoherent && !(e && atoi(e) == 0) ? 1u : 0u;
    }
    // stack bound: one deferred sibling per level of TLAS and BLAS, plus the sentinel
    // worst-case stack use: up to 3 deferred siblings per level of TLAS + BLAS (max_bvh_depth bounds each),
    // one sentinel, the plane pseudo-leaves, a little slack; whatever exceeds the LDS part spills to HBM
    {
        uint32_t need = 6u * (uint32_t)(h.max_bvh_depth + 1) + (uint32_t)h.planes.size() + 9u; // + the bottom marker
        sc->spill_entries = need > (uint32_t)kLdsStack ? need - (uint32_t)kLdsStack : 0u;
    }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, sc->device) == hipSuccess && cus > 0) sc->num_cus = cus;
    }
    sc->features = h.features ? h.features : kFeatAll;
    {   // kFeatNoXform: scenes of TriMesh nodes only whose BLASes all sit in world space (NRAYS_NOXFORM=0: the general permutations, A/B)
        const int f = sc->features;
        bool all = (f == 2 || f == 6 || f == 18 || f == 22) && !h.links.empty();
        for (const InstLink& l : h.links) all = all && (l.flags & kInstNoXform);
        for (const InstLink& l : h.shadow_links) all = all && (l.flags & kInstNoXform);
        const char* e = getenv("NRAYS_NOXFORM");
        sc->noxform = all && !(e && atoi(e) == 0);
        const char* pe = getenv("NRAYS_PARK"); // =0: the three-wave multi-light kernels keep a hit's shading state in registers / scratch (A/B)
        sc->park = !(pe && atoi(pe) == 0);
    }
    // small analytic scenes: one packed copy of the records for the kernels that read them from LDS (DScene::lds_blob)
    sc->d.lds_blob = nullptr; sc->d.lds_bytes = 0;
    { const char* e = getenv("NRAYS_LDS_SCENE");
      const int f = sc->features;
      if ((f == 1 || f == 5 || f == 17 || f == 21) && !(e && atoi(e) == 0)) {
        std::vector<uint8_t> blob;
        auto section = [&](int k, const void* q, size_t n) { // 16-byte aligned sections
            blob.resize((blob.size() + 15u) & ~(size_t)15u);
            sc->d.lds_off[k] = (uint32_t)blob.size();
            const uint8_t* b = (const uint8_t*)q;
            blob.insert(blob.end(), b, b + n);
        };
        section(kLdsNodes, h.nodes.data(), h.nodes.size() * sizeof(BvhNode));
        section(kLdsInstances, h.instances.data(), h.instances.size() * sizeof(Instance));
        section(kLdsShadowInstances, h.shadow_instances.data(), h.shadow_instances.size() * sizeof(Instance));
        section(kLdsLinks, h.links.data(), h.links.size() * sizeof(InstLink));
        section(kLdsShadowLinks, h.shadow_links.data(), h.shadow_links.size() * sizeof(InstLink));
        section(kLdsShade, h.shade.data(), h.shade.size() * sizeof(ShadeRec));
        section(kLdsNodeAabbs, h.node_aabbs.data(), h.node_aabbs.size() * sizeof(double));
        section(kLdsLights, h.lights.data(), h.lights.size() * sizeof(LightRec));
        section(kLdsPlanes, h.planes.data(), h.planes.size() * sizeof(int32_t));
        section(kLdsShadowPlanes, h.shadow_planes.data(), h.shadow_planes.size() * sizeof(int32_t));
        blob.resize((blob.size() + 15u) & ~(size_t)15u);
        if (!blob.empty() && blob.size() <= kLdsSceneBytes) {
            std::vector<uint32_t> words(blob.size() / 4);
            std::memcpy(words.data(), blob.data(), blob.size());
            if ((rc = upload(sc, words, &sc->d.lds_blob)) != NRAYS_OK) return bail(rc);
            sc->d.lds_bytes = (uint32_t)blob.size();
            sc->features |= kFeatLdsScene;
        }
      }
    }
    {   // k_seed_costs: world AABBs of the nodes that can continue a chain (transparent / alpha-mapped / reflective), as f32
        std::vector<float> boxes;
        for (size_t i = 0; i < h.shade.size() && boxes.size() < 6u * 2048u; ++i) {
            const ShadeRec& sr = h.shade[i];
            if (!(sr.alpha < 1.0f || h.shade_alpha_tex[i] >= 0 || sr.refl_mix != 0.0f)) continue;
            const double* b = h.node_aabbs.data() + 6 * i;
            bool finite = true;
            for (int a = 0; a < 6; ++a) finite = finite && std::isfinite(b[a]) && std::fabs(b[a]) < 1e30;
            if (!finite) continue;
            for (int a = 0; a < 6; ++a) boxes.push_back((float)b[a]);
        }
        // (scenes without such a node — opaque hair — get no guess: a frame in image order.  Their nodes' boxes priced by the chord a ray spends inside, round 6: hairball
        // cold frame 2.18 ms against 2.13 without — the order it buys is worth less than the two launches it costs: profiles/r06_regimes_sweep.log)
        const float* dptr = nullptr;
        if ((rc = upload(sc, boxes, &dptr)) != NRAYS_OK) return bail(rc);
        sc->d_seed_boxes = dptr; sc->seed_boxes = (uint32_t)(boxes.size() / 6);
        if (const char* e = getenv("NRAYS_COST_SEED")) { sc->seed_enabled = atoi(e) != 0; if (atoi(e) == 4) sc->seed_rays = 4u; if (atoi(e) == 1) sc->seed_rays = 1u; }
    }
    if (const char* e = getenv("NRAYS_MAX_PRIMARY")) { sc->max_primary_per_launch = (uint64_t)std::max(1ll, atoll(e)); sc->max_primary_forced = true; }
    if (const char* e = getenv("NRAYS_LANE_LOG2")) sc->lane_log2_override = std::max(0, std::min(6, atoi(e)));
    if (const char* e = getenv("NRAYS_EVENT_STRIDE")) sc->event_stride = (uint32_t)std::max(1, atoi(e));
    if (const char* e = getenv("NRAYS_GRAB")) sc->grab_override = std::max(0, atoi(e));
    if (const char* e = getenv("NRAYS_LPT")) sc->lpt_enabled = atoi(e) != 0;
    if (const char* e = getenv("NRAYS_SCREEN_CULL")) sc->cull_enabled = atoi(e) != 0;
    { const int f = sc->features; // multi-light mesh scenes without double branching: 2, 4 or 8 lanes per pixel in a split tile
      if (sc->d.no_elide) sc->light_lsl = 0;
      else if ((f & kFeatMultiSample) && (f & kFeatMesh) && !(f & kFeatDouble) && h.lights.size() >= 2) { uint32_t l = 1; while (l < 3u && (2u << l) <= h.lights.size()) ++l; sc->light_lsl = l; }
      else if (NR_PIXEL_SPLIT && !(f & kFeatMultiSample) && (f & kFeatMesh) && (f & kFeatAlphaShadow) && !(f & kFeatDouble)) sc->light_lsl = 3; } // pixel split: 8 pixels per part
    if (const char* e = getenv("NRAYS_LIGHT_SPLIT")) sc->light_split_factor = (float)atof(e);
    if (const char* e = getenv("NRAYS_OCC")) sc->occ_override = atoi(e);
    if (const char* e = getenv("NRAYS_WAVEFRONT")) sc->wavefront_mode = atoi(e);
    if (const char* e = getenv("NRAYS_LPT_ANALYTIC")) sc->lpt_analytic = atoi(e) != 0;
    if (const char* e = getenv("NRAYS_LPT_REUSE")) sc->lpt_reuse = atoi(e) != 0;
    if (const char* e = getenv("NRAYS_NEAR_REUSE")) sc->near_reuse = atoi(e) != 0;
    // Mesh scenes re-sort on every frame of a moving camera (age 0): the deep foliage chains of the sponza stand-in move between tiles with every pixel of camera motion, and a
    // frame that reuses an order a few frames old waits for tiles it started late — 1.20 ms against 1.145 with the previous frame's costs, 1.04 at rest
    // (profiles/r06_regimes_sweep.log).  Analytic scenes keep an order for 16 frames of a camera within two blocks (balls, moving: 0.0576 ms at 8 frames / 16 pixels,
    // 0.0565 at 16 / 64, 0.0550 with an order that is never refreshed; 0.049 at rest).
    sc->near_pixels = sc->host.any_mesh ? kNearPixels : 2.0 * kNearPixels; sc->max_order_age = sc->host.any_mesh ? 0u : 16u;
    if (const char* e = getenv("NRAYS_SPLIT_HYST")) sc->split_hyst = (float)atof(e);
    sc->host_times = getenv("NRAYS_HOST_TIMES") != nullptr;
    if (const char* e = getenv("NRAYS_NEAR_PIXELS")) sc->near_pixels = atof(e);
    if (const char* e = getenv("NRAYS_ORDER_AGE")) sc->max_order_age = (uint32_t)std::max(0, atoi(e));
    if (const char* e = getenv("NRAYS_LEAD_WGS")) sc->lead_mode = atoi(e) != 0;
    if (const char* e = getenv("NRAYS_LONE_FACTOR")) sc->lone_factor = atof(e);
    if (const char* e = getenv("NRAYS_LEAD_PER_WG")) sc->lead_per_wg = std::max(1, std::min(64, atoi(e)));
    if (const char* e = getenv("NRAYS_GRID_WG_PER_CU")) sc->grid_wg_per_cu = std::max(0, atoi(e));
    // release bulk host copies
    std::vector<BvhNode>().swap(h.nodes); std::vector<TriRec>().swap(h.tris); std::vector<TriUv>().swap(h.triuvs);

    for (int k = 0; k < 2; ++k) {
        if (hipMalloc((void**)&sc->d_counts_set[k], kNumCounts * sizeof(uint32_t)) != hipSuccess ||
            hipMalloc((void**)&sc->d_counters_set[k], sizeof(DeviceCounters)) != hipSuccess)
            return bail(fail(NRAYS_ERR_OOM, "counter allocation failed"));
        if (hipMemset(sc->d_counts_set[k], 0, kNumCounts * sizeof(uint32_t)) != hipSuccess ||
            hipMemset(sc->d_counters_set[k], 0, sizeof(DeviceCounters)) != hipSuccess)
            return bail(fail(NRAYS_ERR_HIP, "counter memset failed"));
    }
    sc->d_counts = sc->d_counts_set[0]; sc->d_counters = sc->d_counters_set[0];
    if (hipMalloc((void**)&sc->d_cost_meta, 4 * sizeof(unsigned long long)) != hipSuccess || hipMemset(sc->d_cost_meta, 0, 4 * sizeof(unsigned long long)) != hipSuccess)
        return bail(fail(NRAYS_ERR_OOM, "cost-meta allocation failed"));
    stage("records, switches, counters");
    if (hipMalloc((void**)&sc->d_counters_primary, sizeof(DeviceCounters)) != hipSuccess)
        return bail(fail(NRAYS_ERR_OOM, "counter allocation failed"));
    if (hipStreamCreate(&sc->own_stream) != hipSuccess) return bail(fail(NRAYS_ERR_HIP, "stream creation failed"));
    // (the ring's timing events are created by the first frame that records into a slot: 1 024 hipEventCreate cost 0.6 ms of every scene creation)
    stage("stream + event ring");
    {   // What a first frame would allocate, sized for frames up to 4K (larger ones re-allocate as before): the reference's caller
        // renders a camera ONCE (loader3d.rs:67-93), so the first frame of a handle is the one that counts for it.
        const char* e = getenv("NRAYS_PREALLOC"); // =0: allocate on the first frame (A/B switch)
        if (!(e && atoi(e) == 0)) {
            const uint32_t nwt = (3840u / 16u) * (2160u / 16u) * 4u;
            if (hipMalloc((void**)&sc->d_tile_cost, (size_t)nwt * sizeof(uint32_t)) == hipSuccess &&
                hipMalloc((void**)&sc->d_tile_order, order_slots(nwt, sc->light_lsl) * sizeof(uint32_t)) == hipSuccess) sc->tile_slots = nwt;
            else { if (sc->d_tile_cost) (void)hipFree(sc->d_tile_cost); sc->d_tile_cost = nullptr; sc->d_tile_order = nullptr; (void)hipGetLastError(); }
            if (sc->light_lsl && hipMalloc((void**)&sc->d_order_len, 8 * sizeof(uint32_t)) != hipSuccess) { sc->d_order_len = nullptr; (void)hipGetLastError(); }
            // ... the analytic scenes' read-back buffers, the event a render on another stream waits for, and the ring's first slots
            if (!sc->host.any_mesh && alloc_cost_stats(sc) != NRAYS_OK) { (void)hipGetLastError(); }
            if (hipEventCreateWithFlags(&sc->ev_switch, hipEventDisableTiming) != hipSuccess) { sc->ev_switch = nullptr; (void)hipGetLastError(); }
            for (int k = 0; k < 2; ++k) if (hipEventCreate(&sc->ev_rec[k]) != hipSuccess) { sc->ev_rec[k] = nullptr; (void)hipGetLastError(); }
            for (int k = 0; k < 8; ++k) (void)ensure_ring_slot(sc, k);
            if (sc->spill_entries && hipMalloc((void**)&sc->d_spill, (size_t)kMaxGrid * kBlock * sc->spill_entries * sizeof(uint32_t)) != hipSuccess) { sc->d_spill = nullptr; (void)hipGetLastError(); }
        }
    }
    stage("first-frame buffers");
    *out_scene = sc;
    return NRAYS_OK;
}

void nrays_scene_destroy(NraysScene* sc) {
    if (!sc) return;
    (void)hipSetDevice(sc->device);
    if (sc->have_last) (void)hipStreamSynchronize(sc->last_stream);
    for (void* p : sc->allocs) (void)hipFree(p);
    for (nrays::DeviceBlas& b : sc->host.dev_blas) nrays::free_device_blas(b); // a creation that failed between the build and the upload
    for (int k = 0; k < 2; ++k) if (sc->queue[k].block) (void)hipFree(sc->queue[k].block);
    for (int k = 0; k < 2; ++k) {
        if (sc->d_counts_set[k]) (void)hipFree(sc->d_counts_set[k]);
        if (sc->d_counters_set[k]) (void)hipFree(sc->d_counters_set[k]);
    }
    if (sc->d_spill) (void)hipFree(sc->d_spill);
    if (sc->d_fixed) (void)hipFree(sc->d_fixed);
    if (sc->d_frame) (void)hipFree(sc->d_frame);
    if (sc->d_tile_cost) (void)hipFree(sc->d_tile_cost);
    if (sc->d_tile_order) (void)hipFree(sc->d_tile_order);
    if (sc->d_order_len) (void)hipFree(sc->d_order_len);
    if (sc->d_cost_stats) (void)hipFree(sc->d_cost_stats);
    if (sc->d_cost_meta) (void)hipFree(sc->d_cost_meta);
    if (sc->d_rgb8) (void)hipFree(sc->d_rgb8);
    if (sc->h_cost_stats) (void)hipHostFree(sc->h_cost_stats);
    if (sc->ev_stats) (void)hipEventDestroy(sc->ev_stats);
    if (sc->ev_switch) (void)hipEventDestroy(sc->ev_switch);
    for (int k = 0; k < 2; ++k) if (sc->ev_rec[k]) (void)hipEventDestroy(sc->ev_rec[k]);
    if (sc->d_counters_primary) (void)hipFree(sc->d_counters_primary);
    for (int k = 0; k < NraysScene::kRing; ++k) {
        if (sc->ev_begin[k]) (void)hipEventDestroy(sc->ev_begin[k]);
        if (sc->ev_pbegin[k]) (void)hipEventDestroy(sc->ev_pbegin[k]);
        if (sc->ev_pend[k]) (void)hipEventDestroy(sc->ev_pend[k]);
        if (sc->ev_end[k]) (void)hipEventDestroy(sc->ev_end[k]);
    }
    wavefront_release(sc);
    if (sc->own_stream) (void)hipStreamDestroy(sc->own_stream);
    delete sc;
}

int nrays_render_device(NraysScene* scene, const NraysRenderParams* params, float* out_rgb_device, void* hip_stream) {
    return render_impl(scene, params, out_rgb_device, (hipStream_t)hip_stream, false);
}

int nrays_render_device_instrumented(NraysScene* scene, const NraysRenderParams* params, float* out_rgb_device, void* hip_stream) {
    return render_impl(scene, params, out_rgb_device, (hipStream_t)hip_stream, true);
}

int nrays_render_device_counted(NraysScene* scene, const NraysRenderParams* params, float* out_rgb_device, void* hip_stream, uint32_t flags) {
    if (flags & ~NRAYS_COUNT_AS_TIMED) return fail(NRAYS_ERR_BAD_ARG, "unknown count flags");
    return render_impl(scene, params, out_rgb_device, (hipStream_t)hip_stream, true, flags);
}

static void fill_counters(NraysStats* out, const DeviceCounters& c) {
    out->rays_reflection = c.rays_reflection; out->rays_refraction = c.rays_refraction; out->rays_shadow = c.rays_shadow;
    out->node_tests = c.node_tests; out->tri_tests = c.tri_tests; out->prim_tests = c.prim_tests;
    out->hit_records = c.hit_records; out->tex_samples = c.tex_samples; out->rays_primary_traced = c.rays_primary_traced; out->node_fetches = c.node_fetches;
}

int nrays_get_stats(NraysScene* sc, NraysStats* out) {
    if (!sc || !out) return fail(NRAYS_ERR_BAD_ARG, "null argument");
    std::memset(out, 0, sizeof *out);
    if (!sc->have_last) return NRAYS_OK;
    HIP_TRY(hipSetDevice(sc->device));
    HIP_TRY(hipStreamSynchronize(sc->last_stream));
    DeviceCounters c;
    HIP_TRY(hipMemcpy(&c, sc->d_counters, sizeof c, hipMemcpyDeviceToHost));
    out->rays_primary = sc->last_primary;
    fill_counters(out, c);
    out->generations = c.max_depth; out->instrumented = sc->last_instrumented ? 1u : 0u;
    out->reserved = c.max_chain_nodes; // instrumented renders: most AABB tests spent on one pixel's whole chain
    out->rays_shadow_elided = c.shadow_elided;
    // average the event timings of the frames recorded since the previous call (at most kRing)
    uint64_t first = sc->frames_reported;
    if (sc->frames_recorded - first > (uint64_t)NraysScene::kRing) first = sc->frames_recorded - NraysScene::kRing;
    double sum_p = 0.0, sum_t = 0.0; uint64_t n = 0;
    for (uint64_t f = first; f < sc->frames_recorded; ++f) {
        int k = (int)(f % NraysScene::kRing);
        float ms_p = 0.f, ms_t = 0.f;
        if (hipEventElapsedTime(&ms_p, sc->ev_pbegin[k], sc->ev_pend[k]) == hipSuccess &&
            hipEventElapsedTime(&ms_t, sc->has_prepass[k] ? sc->ev_begin[k] : sc->ev_pbegin[k], sc->single_launch[k] ? sc->ev_pend[k] : sc->ev_end[k]) == hipSuccess) { sum_p += ms_p; sum_t += ms_t; ++n; }
    }
    sc->frames_reported = sc->frames_recorded;
    if (n) { out->kernel_ms_primary = sum_p / (double)n; out->kernel_ms_total = sum_t / (double)n; }
    out->frames_timed = (uint32_t)n;
    if (c.overflow) return fail(NRAYS_ERR_QUEUE_OVERFLOW, "continuation-ray queue overflow: image is incomplete");
    return NRAYS_OK;
}

int nrays_get_tile_costs(NraysScene* sc, NraysTileCosts* out) {
    if (!sc || !out) return fail(NRAYS_ERR_BAD_ARG, "null argument");
    std::memset(out, 0, sizeof *out);
    if (!sc->have_last || !sc->d_tile_cost || !sc->cost_valid || sc->cost_tiles == 0) return fail(NRAYS_ERR_BAD_ARG, "no frame of this handle has recorded its tile costs");
    HIP_TRY(hipSetDevice(sc->device));
    HIP_TRY(hipStreamSynchronize(sc->last_stream));
    std::vector<uint32_t> c(sc->cost_tiles);
    HIP_TRY(hipMemcpy(c.data(), sc->d_tile_cost, c.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    // The unit the schedule deals is a wave tile — or ONE PART of a tile the cost-ordered lists split: k_primary marks the record of a tile that ran in parts
    // (kCostSplit; the value is its most expensive part's cycles x 2^lsl, or x 3 for the pixel-split tiles of one-light frames).  max_cycles is the longest such
    // unit, sum_cycles what the waves spend: every part of a split tile counted (at its most expensive part's price: an upper bound).
    const uint32_t lsl = sc->cost_split_lsl;
    const bool multi = (sc->features & kFeatMultiSample) != 0;
    for (uint32_t rec : c) {
        const uint32_t v = rec & kCostMask;
        uint64_t unit = v, n = 1;
        if (lsl && (rec & kCostSplit)) { unit = multi ? (uint64_t)(v >> lsl) : (uint64_t)(v / 3u); n = 1ull << lsl; }
        out->sum_cycles += unit * n * 16u; out->max_cycles = std::max<uint64_t>(out->max_cycles, unit * 16u);
    }
    out->tiles = c.size(); out->resident_waves = (uint64_t)sc->cost_grid * (kBlock / 64);
    if (sc->d_cost_meta) { // the recording launch about itself (DRender::cost_meta)
        unsigned long long m[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpy(m, sc->d_cost_meta, sizeof m, hipMemcpyDeviceToHost));
        out->shader_clock_hz = m[3] ? (double)m[2] / ((double)m[3] * 1e-8) : 0.0;
    }
    {   // the recording launch between its events
        float ms = 0.0f;
        hipError_t e = hipErrorInvalidValue;
        if (sc->rec_events_valid) e = hipEventElapsedTime(&ms, sc->ev_rec[0], sc->ev_rec[1]);
        else if (sc->rec_slot >= 0 && sc->ev_pbegin[sc->rec_slot] && sc->ev_pend[sc->rec_slot]) e = hipEventElapsedTime(&ms, sc->ev_pbegin[sc->rec_slot], sc->ev_pend[sc->rec_slot]); // (the ring holds 256 timed frames)
        if (e == hipSuccess) out->kernel_ms = ms; else (void)hipGetLastError();
    }
    return NRAYS_OK;
}

#if defined(NR_PHASE_TIMING) || defined(NR_DEBUG_TILE_COSTS)
// Tuning builds only (tools/tile_costs.py): the per-wave-tile cycle counts (>> 4) of the last frame that recorded them.
int nrays_debug_tile_costs(NraysScene* sc, uint32_t* out, uint32_t capacity, uint32_t* out_count) {
    if (!sc || !sc->have_last || !sc->d_tile_cost) return NRAYS_ERR_BAD_ARG;
    HIP_TRY(hipStreamSynchronize(sc->last_stream));
    const uint32_t n = std::min(capacity, sc->tile_slots);
    HIP_TRY(hipMemcpy(out, sc->d_tile_cost, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    *out_count = n;
    return NRAYS_OK;
}
#endif
#ifdef NR_DEBUG_TILE_COSTS
// Tuning builds only (tools/tile_dump.py): k_seed_costs' guess of the last cold frame.
int nrays_debug_seed_costs(NraysScene* sc, uint32_t* out, uint32_t capacity, uint32_t* out_count) {
    if (!sc || !sc->have_last || !sc->d_seed_copy) return NRAYS_ERR_BAD_ARG;
    HIP_TRY(hipStreamSynchronize(sc->last_stream));
    const uint32_t n = std::min(capacity, sc->tile_slots);
    HIP_TRY(hipMemcpy(out, sc->d_seed_copy, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    *out_count = n;
    return NRAYS_OK;
}
// Tuning builds only (tools/wave_timeline.py): {kernel entry, first tile, exit, tiles} per wave of the last primary launch, 10 ns ticks.
int nrays_debug_wave_times(NraysScene* sc, uint32_t* out, uint32_t capacity_waves, uint32_t* out_waves) {
    if (!sc || !sc->have_last || !sc->d_wave_times) return NRAYS_ERR_BAD_ARG;
    HIP_TRY(hipStreamSynchronize(sc->last_stream));
    const uint32_t n = std::min<uint32_t>(capacity_waves, sc->dbg_grid * (kBlock / 64));
    HIP_TRY(hipMemcpy(out, sc->d_wave_times, (size_t)n * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    *out_waves = n;
    return NRAYS_OK;
}
// The second record per wave: {ticks in work tiles, ticks in tiles that traced nothing, ticks in background rows, work tiles | miss tiles << 8 | rows << 16 | longest work tile / 16 ticks << 24}.
int nrays_debug_wave_times2(NraysScene* sc, uint32_t* out, uint32_t capacity_waves, uint32_t* out_waves) {
    if (!sc || !sc->have_last || !sc->d_wave_times) return NRAYS_ERR_BAD_ARG;
    HIP_TRY(hipStreamSynchronize(sc->last_stream));
    const uint32_t n = std::min<uint32_t>(capacity_waves, sc->dbg_grid * (kBlock / 64));
    HIP_TRY(hipMemcpy(out, sc->d_wave_times + (size_t)kMaxGrid * (kBlock / 64) * 4, (size_t)n * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    *out_waves = n;
    return NRAYS_OK;
}
#endif
#ifdef NR_PHASE_TIMING
// Tuning builds only (tools/phase_timing.py): wave / lane iteration counts of the node loops and the triangle loops.
int nrays_debug_counters(NraysScene* sc, unsigned long long out[16]) {
    if (!sc || !sc->have_last) return NRAYS_ERR_BAD_ARG;
    HIP_TRY(hipStreamSynchronize(sc->last_stream));
    DeviceCounters c;
    HIP_TRY(hipMemcpy(&c, sc->d_counters, sizeof c, hipMemcpyDeviceToHost));
    for (int k = 0; k < 8; ++k) out[k] = c.dbg[k];
    for (int k = 0; k < 8; ++k) out[8 + k] = c.dbg2[k];
    return NRAYS_OK;
}
#endif

int nrays_get_primary_kernel_stats(NraysScene* sc, NraysStats* out) {
    if (!sc || !out) return fail(NRAYS_ERR_BAD_ARG, "null argument");
    std::memset(out, 0, sizeof *out);
    if (!sc->have_last || !sc->last_instrumented) return fail(NRAYS_ERR_BAD_ARG, "the last render was not instrumented");
    HIP_TRY(hipSetDevice(sc->device));
    HIP_TRY(hipStreamSynchronize(sc->last_stream));
    DeviceCounters c;
    HIP_TRY(hipMemcpy(&c, sc->d_counters_primary, sizeof c, hipMemcpyDeviceToHost));
    out->rays_primary = sc->last_primary_first_batch;
    fill_counters(out, c);
    out->rays_shadow_elided = c.shadow_elided;
    // single continuations (reflection OR refraction) are traced by the primary kernel itself (trace_chain); only the
    // second child of a double branch goes through the queue to k_bounce, after this snapshot was taken
    out->instrumented = 1;
    return NRAYS_OK;
}

int nrays_render(NraysScene* sc, const NraysRenderParams* p, float* out_rgb) {
    if (!sc || !p || !out_rgb) return fail(NRAYS_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(sc->device));
    size_t floats = (size_t)tile_rows(p) * p->width * 3;
    if (floats > sc->frame_floats) {
        if (sc->d_frame) { (void)hipFree(sc->d_frame); sc->d_frame = nullptr; sc->frame_floats = 0; }
        HIP_TRY(hipMalloc((void**)&sc->d_frame, floats * sizeof(float)));
        sc->frame_floats = floats;
    }
    int rc = render_impl(sc, p, sc->d_frame, sc->own_stream, false);
    if (rc != NRAYS_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out_rgb, sc->d_frame, floats * sizeof(float), hipMemcpyDeviceToHost, sc->own_stream));
    HIP_TRY(hipStreamSynchronize(sc->own_stream));
    if (sc->host.any_double_branch) { // only scenes with a continuation queue can overflow it: the others skip the extra blocking copy
        unsigned int overflow = 0;
        HIP_TRY(hipMemcpy(&overflow, &sc->d_counters->overflow, sizeof overflow, hipMemcpyDeviceToHost));
        if (overflow) return fail(NRAYS_ERR_QUEUE_OVERFLOW, "continuation-ray queue overflow: image is incomplete");
    }
    return NRAYS_OK;
}

int nrays_render_rgb8(NraysScene* sc, const NraysRenderParams* p, uint8_t* out_rgb8) {
    if (!sc || !p || !out_rgb8) return fail(NRAYS_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(sc->device));
    const size_t n = (size_t)tile_rows(p) * p->width * 3;
    if (n > sc->frame_floats) {
        if (sc->d_frame) { (void)hipFree(sc->d_frame); sc->d_frame = nullptr; sc->frame_floats = 0; }
        HIP_TRY(hipMalloc((void**)&sc->d_frame, n * sizeof(float)));
        sc->frame_floats = n;
    }
    if (n > sc->rgb8_bytes) {
        if (sc->d_rgb8) { (void)hipFree(sc->d_rgb8); sc->d_rgb8 = nullptr; sc->rgb8_bytes = 0; }
        HIP_TRY(hipMalloc((void**)&sc->d_rgb8, n));
        sc->rgb8_bytes = n;
    }
    int rc = render_impl(sc, p, sc->d_frame, sc->own_stream, false);
    if (rc != NRAYS_OK) return rc;
    if (n) {
        hipLaunchKernelGGL(k_quantize_rgb8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sc->own_stream, sc->d_frame, sc->d_rgb8, n);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out_rgb8, sc->d_rgb8, n, hipMemcpyDeviceToHost, sc->own_stream));
    }
    HIP_TRY(hipStreamSynchronize(sc->own_stream));
    if (sc->host.any_double_branch) { // only scenes with a continuation queue can overflow it: the others skip the extra blocking copy
        unsigned int overflow = 0;
        HIP_TRY(hipMemcpy(&overflow, &sc->d_counters->overflow, sizeof overflow, hipMemcpyDeviceToHost));
        if (overflow) return fail(NRAYS_ERR_QUEUE_OVERFLOW, "continuation-ray queue overflow: image is incomplete");
    }
    return NRAYS_OK;
}

int nrays_debug_scene_flags(const NraysScene* sc, uint32_t out[2]) {
    if (!sc || !out) return fail(NRAYS_ERR_BAD_ARG, "null argument");
    out[0] = (uint32_t)sc->host.features; out[1] = sc->d.incoherent;
    return NRAYS_OK;
}

int nrays_debug_blas_build(const NraysMesh* mesh, uint32_t flags, NraysBlasDump* out) {
    if (!mesh || !out) return fail(NRAYS_ERR_BAD_ARG, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(NRAYS_ERR_NO_DEVICE, "no HIP device visible");
    nrays::BlasProbe probe; std::string err;
    const int rc = nrays::build_blas_probe(mesh, (flags & 1u) != 0, (flags & 2u) == 0, probe, err);
    if (rc != NRAYS_OK) return fail(rc, err);
    out->num_nodes = (uint32_t)probe.nodes.size(); out->num_refs = (uint32_t)probe.tri_ids.size();
    out->root = probe.root; out->max_depth = probe.max_depth; out->hairy = probe.hairy ? 1u : 0u;
    if (probe.nodes.size() > out->node_capacity || probe.tri_ids.size() > out->ref_capacity) return fail(NRAYS_ERR_BAD_ARG, "dump buffers too small");
    if (!probe.nodes.empty()) { if (!out->nodes) return fail(NRAYS_ERR_BAD_ARG, "null node buffer"); std::memcpy(out->nodes, probe.nodes.data(), probe.nodes.size() * sizeof(nrays::BvhNode)); }
    if (!probe.tri_ids.empty()) { if (!out->tri_ids) return fail(NRAYS_ERR_BAD_ARG, "null reference buffer"); std::memcpy(out->tri_ids, probe.tri_ids.data(), probe.tri_ids.size() * 4u); }
    return NRAYS_OK;
}

int nrays_debug_node_aabb(NraysScene* sc, uint32_t node, double out[6]) {
    if (!sc || !out) return fail(NRAYS_ERR_BAD_ARG, "null argument");
    if (!sc->d.node_aabbs || (size_t)node * 6 + 6 > sc->host.node_aabbs.size()) return fail(NRAYS_ERR_BAD_ARG, "node index out of range");
    HIP_TRY(hipSetDevice(sc->device));
    HIP_TRY(hipMemcpy(out, sc->d.node_aabbs + 6 * (size_t)node, 6 * sizeof(double), hipMemcpyDeviceToHost));
    return NRAYS_OK;
}

int nrays_debug_cast_batch(NraysScene* sc, uint32_t mode, uint32_t n, const double* origins, const double* dirs, const double* max_toi, NraysCastResult* out) {
    if (!sc || !origins || !dirs || !out || mode > 1u || (mode == 1u && !max_toi)) return fail(NRAYS_ERR_BAD_ARG, "bad cast-batch arguments");
    if (n == 0) return NRAYS_OK;
    HIP_TRY(hipSetDevice(sc->device));
    if (sc->have_last) HIP_TRY(hipStreamSynchronize(sc->last_stream));
    if (sc->spill_entries && !sc->d_spill) HIP_TRY(hipMalloc((void**)&sc->d_spill, (size_t)kMaxGrid * kBlock * sc->spill_entries * sizeof(uint32_t)));
    double *d_o = nullptr, *d_d = nullptr, *d_t = nullptr; NraysCastResult* d_r = nullptr;
    auto release = [&]() { if (d_o) (void)hipFree(d_o); if (d_d) (void)hipFree(d_d); if (d_t) (void)hipFree(d_t); if (d_r) (void)hipFree(d_r); };
#define CAST_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { release(); return fail(e_ == hipErrorOutOfMemory ? NRAYS_ERR_OOM : NRAYS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
    const size_t vb = (size_t)n * 3 * sizeof(double);
    CAST_TRY(hipMalloc((void**)&d_o, vb)); CAST_TRY(hipMalloc((void**)&d_d, vb)); CAST_TRY(hipMalloc((void**)&d_r, (size_t)n * sizeof(NraysCastResult)));
    CAST_TRY(hipMemcpy(d_o, origins, vb, hipMemcpyHostToDevice)); CAST_TRY(hipMemcpy(d_d, dirs, vb, hipMemcpyHostToDevice));
    if (mode == 1u) { CAST_TRY(hipMalloc((void**)&d_t, (size_t)n * sizeof(double))); CAST_TRY(hipMemcpy(d_t, max_toi, (size_t)n * sizeof(double), hipMemcpyHostToDevice)); }
    const uint32_t grid = std::min<uint32_t>((n + kBlock - 1) / kBlock, (uint32_t)kMaxGrid);
    if ((sc->features & ~(int)kFeatMultiSample) == (int)kFeatMesh) hipLaunchKernelGGL((k_cast_batch<kFeatMesh>), dim3(grid), dim3(kBlock), 0, sc->own_stream, sc->d, mode, n, d_o, d_d, d_t, d_r, sc->d_spill);
    else hipLaunchKernelGGL((k_cast_batch<kFeatAll>), dim3(grid), dim3(kBlock), 0, sc->own_stream, sc->d, mode, n, d_o, d_d, d_t, d_r, sc->d_spill);
    CAST_TRY(hipGetLastError());
    CAST_TRY(hipStreamSynchronize(sc->own_stream));
    CAST_TRY(hipMemcpy(out, d_r, (size_t)n * sizeof(NraysCastResult), hipMemcpyDeviceToHost));
#undef CAST_TRY
    release();
    return NRAYS_OK;
}

int nrays_untile_device(const float* gathered, float* out_rgb_device, uint32_t width, uint32_t height, uint32_t band_rows,
                        uint32_t band_owners, void* hip_stream) {
    if (!gathered || !out_rgb_device || width == 0 || height == 0 || band_rows == 0 || band_owners == 0)
        return fail(NRAYS_ERR_BAD_ARG, "bad untile arguments");
    NraysRenderParams p; std::memset(&p, 0, sizeof p);
    p.width = width; p.height = height; p.band_rows = band_rows; p.band_owners = band_owners;
    uint32_t rows_local = band_owners > 1 ? tile_rows(&p) : height;
    size_t n = (size_t)width * height * 3;
    hipLaunchKernelGGL(k_untile, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, gathered, out_rgb_device,
                       width, height, band_rows, band_owners, rows_local);
    HIP_TRY(hipGetLastError());
    return NRAYS_OK;
}

} // extern "C"
