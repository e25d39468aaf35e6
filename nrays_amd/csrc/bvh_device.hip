// bvh_device.hip — the BLAS of a TriMesh group built on the GPU (see bvh_device.h; nrays_scene_create takes this path from 2 000 triangles).
//
// Stages (all on the current device, null stream):
//   1. k_tri_records      triangle records / uvs / f32 boxes from the caller's f64 vertex arrays (+ the f32-exactness checks)
//   2. k_presplit         pre-splitting of thin diagonal triangles (presplit_clip.h: the host builder's clip arithmetic); the
//                         threshold a budget amounts to comes from a histogram of the empty areas of ALL pieces instead of the
//                         host's heap over a sample (a depth-capped survey pass first, then full-depth passes until the count fits)
//   3. binned-SAH binary build, the split rule of bvh_build.cpp bit for bit (32 bins on the centroid bounds, three axes, the first
//      minimum in (axis, bin) order, the SAH leaf criterion):
//        large nodes (> kSmall references) level by level — k_bin (LDS bins per 1024-reference chunk, merged with encoded
//        atomics), k_select (one wave per node), k_chunk_lefts + k_part_scan + k_scatter (stable partition between two order buffers);
//        small nodes — k_small: ONE WAVE builds the whole subtree of a node in LDS (two-reference nodes are decided without bins).
//      A binary node is stored at index (split position - 1): no allocation, no atomics, deterministic.
//   4. collapse into the 128-byte 4-wide nodes of device_types.h (the host Collapser's rule) level by level, subtree sizes bottom-up,
//      then the host builder's depth-first node order top-down, refs rebased to the scene's arrays
//   5. k_gather           leaf-ordered triangle records and uvs
// min / max / counts are exact and the costs are evaluated with the host's f32 operations (-ffp-contract=off), so from the same
// references both builders produce the same tree (tests/test_device_build_gpu.py compares the node arrays).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>

#include "../../include/nrays_abi.h"
#include "bvh_build.h"
#include "bvh_device.h"
#include "presplit_clip.h"

namespace nrays {
namespace {

static_assert(kSahBins == 32, "the device builder maps one SAH bin to one lane of a 32-lane half wave");
constexpr uint32_t kChunk = 1024;  // references per workgroup of the large-node kernels
constexpr uint32_t kSmall = 256;   // nodes up to this many references are finished by one wave in LDS
constexpr int kBinWords = 96 * 6;  // min (or max) words of one node's bins: [axis][bin][box xyz, centroid xyz]
constexpr uint32_t kHistShift = 13, kHistBins = 1u << (32 - kHistShift);
#define INF_F __builtin_huge_valf()

struct Task {
    uint32_t first, count;
    int32_t parent;  // binary node whose child this range is (-1: the root)
    uint32_t flags;  // kTaskSide: right child; kTaskBuf: which order buffer holds the range
    float bmn[3], bmx[3], cmn[3], cmx[3]; // bounds of the boxes / of the centroids
};
static_assert(sizeof(Task) == 64, "Task layout");
enum : uint32_t { kTaskSide = 1u, kTaskBuf = 2u };

struct Node2 { float lmin[3], lmax[3], rmin[3], rmax[3]; int32_t left, right; };

struct SplitInfo { int32_t axis; int32_t split; uint32_t mid; float lo, scale; }; // axis < 0: the range is not scattered

struct Counters {
    uint32_t n_next, n_small, overflow, n_binary;
    int32_t root_ref;
    uint32_t err;        // bit 0: index out of range, bit 1: vertex not f32-exact, bit 2: uv not f32-exact
    uint32_t next_id;    // collapse: compact ids handed out
    uint32_t pad;
    unsigned long long phase[8]; // tuning builds (NR_BUILD_PHASES): wave cycles of k_small's phases
    uint32_t bounds[12]; // encoded: box min xyz, centroid min xyz, box max xyz, centroid max xyz (root pass, BLAS bounds)
};

// ---- order-preserving float <-> uint encoding (min / max through integer atomics) -----------------------------------
__device__ inline uint32_t enc(float f) { uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ inline float dec(uint32_t e) { return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e); }
__device__ inline float dec_min(uint32_t e) { return e == 0xffffffffu ? INF_F : dec(e); } // identity of min = all ones (memset 0xff)
__device__ inline float dec_max(uint32_t e) { return e == 0u ? -INF_F : dec(e); }          // identity of max = zero (memset 0)

__device__ inline float half_area3(const float mn[3], const float mx[3]) { // bvh_build.cpp: Box::half_area
    float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    if (!(dx >= 0.f) || !(dy >= 0.f) || !(dz >= 0.f)) return 0.f;
    return dx * dy + dy * dz + dz * dx;
}
#ifdef NR_BUILD_PHASES
#define PHASE_BEGIN() unsigned long long ph_t0 = __builtin_readcyclecounter(); unsigned long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PHASE(k) do { unsigned long long ph_t1 = __builtin_readcyclecounter(); ph_acc[k] += ph_t1 - ph_t0; ph_t0 = ph_t1; } while (0)
#define PHASE_END(ctr) do { if (lane_id() == 0) for (int k_ = 0; k_ < 8; ++k_) if (ph_acc[k_]) atomicAdd(&(ctr)->phase[k_], ph_acc[k_]); } while (0)
#else
#define PHASE_BEGIN() do {} while (0)
#define PHASE(k) do {} while (0)
#define PHASE_END(ctr) do {} while (0)
#endif
__device__ inline void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
__device__ inline float sel3(const float v[3], int a) { return a == 0 ? v[0] : (a == 1 ? v[1] : v[2]); } // (a private array indexed at run time would be placed in scratch or LDS)
__device__ inline int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ inline float bcast_f(float v, int src) { return __shfl(v, src, 64); }
__device__ inline uint32_t bcast_u(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }

// Bins settle after their first few entries: most updates would not change them — a plain LDS read (same-address reads are a broadcast) filters those out
// before the atomic (which serialises the lanes that hit one address): k_bin's 1024-reference chunks, large-node levels 37.8 -> 28.8 ms.
__device__ inline void lds_min(uint32_t* p, uint32_t v) { if (v < *(volatile uint32_t*)p) atomicMin(p, v); }
__device__ inline void lds_max(uint32_t* p, uint32_t v) { if (v > *(volatile uint32_t*)p) atomicMax(p, v); }
__device__ inline void load_box(const float* ref_box, uint32_t id, float mn[3], float mx[3]) {
    const float2* p = reinterpret_cast<const float2*>(ref_box + 6 * (size_t)id);
    float2 a = p[0], b = p[1], c = p[2];
    mn[0] = a.x; mn[1] = a.y; mn[2] = b.x; mx[0] = b.y; mx[1] = c.x; mx[2] = c.y;
}
__device__ inline int bin_of(float cent, float lo, float scale) {
    int b = (int)((cent - lo) * scale);
    return min(max(b, 0), kSahBins - 1);
}

// ---- scans inside the two 32-lane halves of a wave with DPP (no LDS traffic: the wave's LDS port is busy with the bins) -----------
// row_shr:1,2,4,8 scan a row of 16 lanes, row_bcast:15 on rows 1 and 3 adds the total of the row below: lane 31 / lane 63 end up with
// the totals of lanes 0..31 / 32..63.  A lane whose source lies outside its row keeps `identity`.
template <int CTRL, int ROW_MASK> __device__ inline int dpp_i(int identity, int x) { return __builtin_amdgcn_update_dpp(identity, x, CTRL, ROW_MASK, 0xf, false); }
template <int CTRL, int ROW_MASK> __device__ inline float dpp_f(float identity, float x) { return __int_as_float(dpp_i<CTRL, ROW_MASK>(__float_as_int(identity), __float_as_int(x))); }
#define NR_SCAN32(x, identity, OP, DPP)                     \
    do {                                                    \
        x = OP(x, DPP<0x111, 0xf>(identity, x));            \
        x = OP(x, DPP<0x112, 0xf>(identity, x));            \
        x = OP(x, DPP<0x114, 0xf>(identity, x));            \
        x = OP(x, DPP<0x118, 0xf>(identity, x));            \
        x = OP(x, DPP<0x142, 0xa>(identity, x));            \
    } while (0)
__device__ inline int add_i(int a, int b) { return a + b; }
__device__ inline float scan32_min(float x) { NR_SCAN32(x, INF_F, fminf, dpp_f); return x; }
__device__ inline float scan32_max(float x) { NR_SCAN32(x, -INF_F, fmaxf, dpp_f); return x; }
__device__ inline uint32_t scan32_add(uint32_t v) { int x = (int)v; NR_SCAN32(x, 0, add_i, dpp_i); return (uint32_t)x; }
__device__ inline float lane_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ inline uint32_t lane_u(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }

// ---- the split rule of Builder::build for one node, evaluated by one wave ---------------------------------------------
// bins: mn / mx = kBinWords encoded words each, cnt = 96 counts ([axis][bin]); every argument and every result is wave-uniform.
// (An axis whose centroids coincide has no entries in its bins: its candidates have a zero count on one side and never compete.)
struct SplitResult {
    int axis, split; // axis < 0: every centroid coincides (split by index)
    bool leaf;
    uint32_t nleft;
    float lb_mn[3], lb_mx[3], lc_mn[3], lc_mx[3], rb_mn[3], rb_mx[3], rc_mn[3], rc_mx[3];
};
__device__ void select_split(const uint32_t* mn, const uint32_t* mx, const uint32_t* cnt, const Task& t, int max_leaf, float prim_cost, SplitResult& r) {
    const int lane = lane_id(), half = lane >> 5, j = lane & 31;
    const int bin = half ? 31 - j : j; // lower half: prefix over bins 0..j; upper half: suffix over bins 31-j..31
    float area[3]; uint32_t c[3];
    for (int axis = 0; axis < 3; ++axis) { // the three axes side by side: one LDS round trip, then VALU only
        const int idx = axis * 32 + bin;
        float bmn[3], bmx[3];
        for (int a = 0; a < 3; ++a) { bmn[a] = scan32_min(dec_min(mn[idx * 6 + a])); bmx[a] = scan32_max(dec_max(mx[idx * 6 + a])); }
        c[axis] = scan32_add(cnt[idx]);
        area[axis] = half_area3(bmn, bmx);
    }
    float best_cost = INF_F; int best_key = -1;
    const int src = lane <= 30 ? 62 - lane : lane; // the suffix over bins lane+1..31 sits in lane 32 + (31 - (lane + 1))
    for (int axis = 0; axis < 3; ++axis) {
        const float rarea = __shfl(area[axis], src, 64);
        const uint32_t rc = (uint32_t)__shfl((int)c[axis], src, 64);
        if (lane <= 30 && c[axis] > 0u && rc > 0u) {
            const float cost = area[axis] * (float)c[axis] + rarea * (float)rc;
            if (cost < best_cost) { best_cost = cost; best_key = axis * 32 + lane; }
        }
    }
    if (lane > 30) { best_cost = INF_F; best_key = -1; }
    // first minimum in (axis, bin) order, like the sequential sweep: a scan over the lower half, lane 31 holds the winner
#define NR_ARGMIN_STEP(CTRL, ROW_MASK)                                                                                            \
    do {                                                                                                                          \
        const float oc = dpp_f<CTRL, ROW_MASK>(INF_F, best_cost);                                                                 \
        const int ok = dpp_i<CTRL, ROW_MASK>(-1, best_key);                                                                       \
        if (ok >= 0 && (best_key < 0 || oc < best_cost || (oc == best_cost && ok < best_key))) { best_cost = oc; best_key = ok; } \
    } while (0)
    NR_ARGMIN_STEP(0x111, 0xf); NR_ARGMIN_STEP(0x112, 0xf); NR_ARGMIN_STEP(0x114, 0xf); NR_ARGMIN_STEP(0x118, 0xf); NR_ARGMIN_STEP(0x142, 0xa);
#undef NR_ARGMIN_STEP
    best_cost = lane_f(best_cost, 31); best_key = __builtin_amdgcn_readlane(best_key, 31);
    r.leaf = false;
    if ((int)t.count <= max_leaf) { // SAH: an exact f64 ray / triangle test costs about prim_cost times a node visit
        const float ha = half_area3(t.bmn, t.bmx);
        const float leaf_cost = ha * (float)t.count * prim_cost;
        const float split_cost = best_key < 0 ? INF_F : best_cost * prim_cost + ha * 1.0f;
        if (!(split_cost < leaf_cost)) r.leaf = true;
    }
    r.axis = best_key < 0 ? -1 : best_key >> 5;
    r.split = best_key < 0 ? 0 : best_key & 31;
    r.nleft = 0;
    if (r.leaf || best_key < 0) return;
    // children: lower half reduces the bins <= split, upper half the bins above
    const int idx = r.axis * 32 + j;
    const bool mine = half == 0 ? j <= r.split : j > r.split;
    float v[12]; // box min, centroid min, box max, centroid max
    for (int a = 0; a < 6; ++a) { v[a] = scan32_min(mine ? dec_min(mn[idx * 6 + a]) : INF_F); v[6 + a] = scan32_max(mine ? dec_max(mx[idx * 6 + a]) : -INF_F); }
    const uint32_t cc = scan32_add(mine ? cnt[idx] : 0u);
    for (int a = 0; a < 3; ++a) {
        r.lb_mn[a] = lane_f(v[a], 31); r.lc_mn[a] = lane_f(v[3 + a], 31); r.lb_mx[a] = lane_f(v[6 + a], 31); r.lc_mx[a] = lane_f(v[9 + a], 31);
        r.rb_mn[a] = lane_f(v[a], 63); r.rc_mn[a] = lane_f(v[3 + a], 63); r.rb_mx[a] = lane_f(v[6 + a], 63); r.rc_mx[a] = lane_f(v[9 + a], 63);
    }
    r.nleft = lane_u(cc, 31);
}

// Twelve-value wave reduction (min of v[0..5], max of v[6..11]); the result is uniform.
__device__ inline void wave_reduce12(float v[12]) {
    for (int a = 0; a < 6; ++a) {
        const float lo = scan32_min(v[a]), hi = scan32_max(v[6 + a]);
        v[a] = fminf(lane_f(lo, 31), lane_f(lo, 63)); v[6 + a] = fmaxf(lane_f(hi, 31), lane_f(hi, 63));
    }
}
__device__ inline void acc12(float v[12], const float mn[3], const float mx[3]) {
    for (int a = 0; a < 3; ++a) {
        const float ce = 0.5f * mn[a] + 0.5f * mx[a];
        v[a] = fminf(v[a], mn[a]); v[3 + a] = fminf(v[3 + a], ce); v[6 + a] = fmaxf(v[6 + a], mx[a]); v[9 + a] = fmaxf(v[9 + a], ce);
    }
}
__device__ inline void init12(float v[12]) { for (int a = 0; a < 6; ++a) { v[a] = INF_F; v[6 + a] = -INF_F; } }

__device__ inline void set_ref(Node2* node2, Counters* ctr, int32_t parent, uint32_t flags, int32_t ref) {
    if (parent < 0) ctr->root_ref = ref;
    else if (flags & kTaskSide) node2[parent].right = ref; else node2[parent].left = ref;
}
__device__ inline void make_task(Task& c, uint32_t first, uint32_t count, int32_t parent, uint32_t flags, const float bmn[3], const float bmx[3], const float cmn[3], const float cmx[3]) {
    c.first = first; c.count = count; c.parent = parent; c.flags = flags;
    for (int a = 0; a < 3; ++a) { c.bmn[a] = bmn[a]; c.bmx[a] = bmx[a]; c.cmn[a] = cmn[a]; c.cmx[a] = cmx[a]; }
}

// ---- stage 1: triangle records ----------------------------------------------------------------------------------------
struct PartDev { const double* vertices; const double* uvs; const uint32_t* indices; uint32_t num_vertices, num_triangles, node_id, tri_base, block_base, pad; };

// One launch over all the meshes of the BLAS (a merged group of the sponza stand-in: 270 parts — one launch per part was 1.7 ms of an 8 ms build): block b serves the
// part with the largest block_base <= b; the per-block sums are those of the per-part launches (same triangles per block, same fixed tree).
__global__ __launch_bounds__(256) void k_tri_records(const PartDev* __restrict__ parts, uint32_t nparts, TriRec* recs, TriUv* uvs, float* tbox, double* part_area, double* part_tri2,
                                                     Counters* ctr) {
    uint32_t lo = 0, hi = nparts;
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (parts[mid].block_base <= blockIdx.x) lo = mid; else hi = mid; }
    const PartDev part = parts[lo];
    const uint32_t block_base = part.block_base, local_block = blockIdx.x - block_base;
    const uint32_t t = local_block * 256u + threadIdx.x;
    double ha = 0.0, ta = 0.0;
    float v12[12]; init12(v12);
    uint32_t err = 0;
    if (t < part.num_triangles) {
        TriRec r; TriUv uv; PrimBounds b;
        for (int k = 0; k < 6; ++k) uv.uv[k] = 0.0f;
        for (int a = 0; a < 3; ++a) { b.mn[a] = INF_F; b.mx[a] = -INF_F; }
        float* vs[3] = {r.v0, r.v1, r.v2};
        for (int k = 0; k < 3; ++k) {
            uint32_t vi = part.indices[3 * (size_t)t + k];
            if (vi >= part.num_vertices) { err |= 1u; vi = 0; }
            for (int a = 0; a < 3; ++a) {
                const double x = part.vertices[3 * (size_t)vi + a];
                const float f = (float)x;
                if (!((double)f == x)) err |= 2u; // also rejects NaN
                vs[k][a] = f;
                b.mn[a] = fminf(b.mn[a], f); b.mx[a] = fmaxf(b.mx[a], f);
            }
            if (part.uvs) for (int a = 0; a < 2; ++a) {
                const double x = part.uvs[2 * (size_t)vi + a];
                const float f = (float)x;
                if (!((double)f == x)) err |= 4u;
                uv.uv[2 * k + a] = f;
            }
        }
        r.node_id = part.node_id; r.tri_id = t; r.pad = 0;
        const size_t g = (size_t)part.tri_base + t;
        recs[g] = r; uvs[g] = uv;
        for (int a = 0; a < 3; ++a) { tbox[6 * g + a] = b.mn[a]; tbox[6 * g + 3 + a] = b.mx[a]; }
        ClipPoly p; tri_poly(r, p);
        ha = box_half_area(b); ta = poly_area2(p);
        for (int a = 0; a < 3; ++a) { v12[a] = b.mn[a]; v12[6 + a] = b.mx[a]; }
    }
    // deterministic block sums (fixed tree), mesh bounds through encoded atomics
    __shared__ double s_a[256], s_t[256];
    s_a[threadIdx.x] = ha; s_t[threadIdx.x] = ta;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) { s_a[threadIdx.x] += s_a[threadIdx.x + s]; s_t[threadIdx.x] += s_t[threadIdx.x + s]; } __syncthreads(); }
    if (threadIdx.x == 0) { part_area[block_base + local_block] = s_a[0]; part_tri2[block_base + local_block] = s_t[0]; }
    wave_reduce12(v12);
    if (lane_id() == 0 && v12[0] <= v12[6]) for (int a = 0; a < 3; ++a) { atomicMin(&ctr->bounds[a], enc(v12[a])); atomicMax(&ctr->bounds[6 + a], enc(v12[6 + a])); }
    const unsigned long long any = __ballot(err != 0);
    if (any && err) atomicOr(&ctr->err, err);
}

// ---- stage 2: pre-splitting ---------------------------------------------------------------------------------------------
struct SplitFrame { ClipPoly poly; PrimBounds box; int slot; int depth; };
enum { kModeHist = 1, kModeEmit = 2 }; // count the extra references of every triangle (+ the histogram of their empty areas) / write the boxes
// One thread walks the split tree of one triangle exactly like split_rec() of scene_build.cpp (low half first; the high half's
// reference slot is reserved when the split happens).  kModeHist counts and files the empty area of every split; kModeEmit writes boxes.
// One walk (round 6): a counting pass (kModeHist) also KEEPS the pieces it finds — the unsplit piece of triangle t in base_box[t], every extra piece as (box, t, slot) in this
// workgroup's region of an unordered list (`u`: an LDS cursor hands out the places) — so that once the prefix sum over the counts is known k_place_extra moves them to where
// kModeEmit's second walk would have written them (reference n + offsets[t] + slot: the same arrays, bit for bit) and the split trees are not walked again.  Only the LAST
// counting pass's list is used (every pass starts its regions afresh); a region that overflows sets u_count[gridDim.x] and the host falls back to the second walk.
struct PieceList { float* box; uint32_t* key; float* base_box; uint32_t* count; uint32_t region_cap; }; // box: 6 floats, key: (t, slot) per piece; count[wg], count[grid] = overflow
template <int MODE>
__global__ __launch_bounds__(256) void k_presplit(const TriRec* recs, const float* tbox, uint32_t n, double thr, int depth_cap, SplitFrame* frames, uint32_t* counts,
                                                  const uint32_t* offsets, uint32_t* hist, float* ref_box, uint32_t* ref_tri, uint32_t* capped, PieceList u) {
    __shared__ uint32_t u_cursor;
    if (MODE == kModeHist && u.box) { if (threadIdx.x == 0) u_cursor = 0u; __syncthreads(); }
    SplitFrame* stack = frames + (size_t)(blockIdx.x * 256u + threadIdx.x) * (kSplitDepthMax + 1);
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < n; t += gridDim.x * 256u) {
        ClipPoly cur; PrimBounds box; int slot = -1, depth = 0, sp = 0, next = 0;
        tri_poly(recs[t], cur);
        for (int a = 0; a < 3; ++a) { box.mn[a] = tbox[6 * (size_t)t + a]; box.mx[a] = tbox[6 * (size_t)t + 3 + a]; }
        for (;;) {
            const double ha = box_half_area(box);
            const double gain = ha - poly_area2(cur);
            ClipPoly lo, hi; PrimBounds bl, bh;
            // depth_cap < kSplitDepthMax: the survey pass of the host code below (a piece that would still be split there is reported through `capped`)
            if (MODE == kModeHist && depth >= depth_cap && depth < kSplitDepthMax && piece_qualifies(ha, gain, thr)) *capped = 1u;
            if (depth >= depth_cap || !piece_qualifies(ha, gain, thr) || !split_piece(cur, box, lo, hi, bl, bh)) {
                if (MODE == kModeEmit) {
                    const size_t r = slot < 0 ? (size_t)t : (size_t)n + offsets[t] + (uint32_t)slot;
                    for (int a = 0; a < 3; ++a) { ref_box[6 * r + a] = box.mn[a]; ref_box[6 * r + 3 + a] = box.mx[a]; }
                    ref_tri[r] = t;
                }
                if (MODE == kModeHist && u.box) { // keep the piece (see PieceList)
                    if (slot < 0) { for (int a = 0; a < 3; ++a) { u.base_box[6 * (size_t)t + a] = box.mn[a]; u.base_box[6 * (size_t)t + 3 + a] = box.mx[a]; } }
                    else {
                        const uint32_t pos = atomicAdd(&u_cursor, 1u);
                        if (pos < u.region_cap) {
                            const size_t q = (size_t)blockIdx.x * u.region_cap + pos;
                            for (int a = 0; a < 3; ++a) { u.box[6 * q + a] = box.mn[a]; u.box[6 * q + 3 + a] = box.mx[a]; }
                            u.key[2 * q] = t; u.key[2 * q + 1] = (uint32_t)slot;
                        }
                    }
                }
                if (sp == 0) break;
                --sp;
                const SplitFrame& f = stack[sp]; // only the vertices in use travel (a frame is 328 bytes, a piece has 3 - 5 vertices)
                cur.n = f.poly.n;
                for (int k = 0; k < cur.n; ++k) for (int d = 0; d < 3; ++d) cur.v[k][d] = f.poly.v[k][d];
                box = f.box; slot = f.slot; depth = f.depth;
                continue;
            }
            if (MODE == kModeHist) atomicAdd(&hist[__float_as_uint((float)gain) >> kHistShift], 1u);
            {
                SplitFrame& f = stack[sp];
                f.poly.n = hi.n;
                for (int k = 0; k < hi.n; ++k) for (int d = 0; d < 3; ++d) f.poly.v[k][d] = hi.v[k][d];
                f.box = bh; f.slot = next++; f.depth = depth + 1; ++sp;
            }
            cur.n = lo.n;
            for (int k = 0; k < lo.n; ++k) for (int d = 0; d < 3; ++d) cur.v[k][d] = lo.v[k][d];
            box = bl; ++depth;
        }
        if (MODE != kModeEmit) counts[t] = (uint32_t)next;
    }
    if (MODE == kModeHist && u.box) {
        __syncthreads();
        if (threadIdx.x == 0) { u.count[blockIdx.x] = u_cursor < u.region_cap ? u_cursor : u.region_cap; if (u_cursor > u.region_cap) u.count[gridDim.x] = 1u; }
    }
}
// The kept extra pieces to their final places: reference n + offsets[t] + slot (what kModeEmit computes while it walks).
__global__ __launch_bounds__(256) void k_place_extra(PieceList u, const uint32_t* __restrict__ offsets, uint32_t n, float* __restrict__ ref_box, uint32_t* __restrict__ ref_tri) {
    const uint32_t cnt = u.count[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < cnt; i += 256u) {
        const size_t q = (size_t)blockIdx.x * u.region_cap + i;
        const uint32_t t = u.key[2 * q], slot = u.key[2 * q + 1];
        const size_t r = (size_t)n + offsets[t] + slot;
        for (int a = 0; a < 6; ++a) ref_box[6 * r + a] = u.box[6 * q + a];
        ref_tri[r] = t;
    }
}
__global__ void k_iota_refs(uint32_t n, const float* tbox, float* ref_box, uint32_t* ref_tri) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    for (int a = 0; a < 6; ++a) ref_box[6 * (size_t)i + a] = tbox[6 * (size_t)i + a];
    ref_tri[i] = i;
}

// ---- stage 3: binary build ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_root_bounds(const float* ref_box, uint32_t nrefs, uint32_t* order0, Counters* ctr) {
    float v[12]; init12(v);
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < nrefs; i += gridDim.x * 256u) {
        float mn[3], mx[3]; load_box(ref_box, i, mn, mx);
        acc12(v, mn, mx);
        order0[i] = i;
    }
    wave_reduce12(v);
    if (lane_id() == 0 && v[0] <= v[6]) for (int a = 0; a < 6; ++a) { atomicMin(&ctr->bounds[a], enc(v[a])); atomicMax(&ctr->bounds[6 + a], enc(v[6 + a])); }
}
__global__ void k_root_task(Task* tasks, Task* small, uint32_t nrefs, Counters* ctr) {
    Task t; t.first = 0; t.count = nrefs; t.parent = -1; t.flags = 0;
    for (int a = 0; a < 3; ++a) { t.bmn[a] = dec_min(ctr->bounds[a]); t.cmn[a] = dec_min(ctr->bounds[3 + a]); t.bmx[a] = dec_max(ctr->bounds[6 + a]); t.cmx[a] = dec_max(ctr->bounds[9 + a]); }
    if (nrefs > kSmall) { tasks[0] = t; ctr->n_next = 1; ctr->n_small = 0; } else { small[0] = t; ctr->n_next = 0; ctr->n_small = 1; }
}

// chunk_base[t] = first chunk of task t (exclusive scan of ceil(count / kChunk)); one workgroup.
__global__ __launch_bounds__(1024) void k_chunk_scan(const Task* tasks, uint32_t nt, uint32_t* chunk_base) {
    __shared__ uint32_t s_sum[1024];
    const uint32_t per = (nt + 1023u) / 1024u, lo = threadIdx.x * per, hi = min(nt, lo + per);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += (tasks[i].count + kChunk - 1u) / kChunk;
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) { uint32_t v = threadIdx.x >= off ? s_sum[threadIdx.x - off] : 0u; __syncthreads(); s_sum[threadIdx.x] += v; __syncthreads(); }
    uint32_t run = s_sum[threadIdx.x] - sum;
    for (uint32_t i = lo; i < hi; ++i) { chunk_base[i] = run; run += (tasks[i].count + kChunk - 1u) / kChunk; }
    if (threadIdx.x == 1023u) chunk_base[nt] = s_sum[1023];
}

// The same for the NEXT level, whose task count the host does not know yet: read from the device (Counters::n_next), handed back with the chunk total in out2 — ONE blocking
// read-back per level instead of two (a level is ~110 us of which each read-back is ~25).  A count beyond the lists' capacity is only reported (the host fails the build).
__global__ __launch_bounds__(1024) void k_chunk_scan_next(const Task* tasks, const uint32_t* nt_ptr, uint32_t cap, uint32_t* chunk_base, uint32_t* out2) {
    __shared__ uint32_t s_sum[1024];
    const uint32_t nt = *nt_ptr;
    if (nt == 0u || nt > cap) { if (threadIdx.x == 0u) { out2[0] = nt; out2[1] = 0u; } return; } // (uniform)
    const uint32_t per = (nt + 1023u) / 1024u, lo = threadIdx.x * per, hi = min(nt, lo + per);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += (tasks[i].count + kChunk - 1u) / kChunk;
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) { uint32_t v = threadIdx.x >= off ? s_sum[threadIdx.x - off] : 0u; __syncthreads(); s_sum[threadIdx.x] += v; __syncthreads(); }
    uint32_t run = s_sum[threadIdx.x] - sum;
    for (uint32_t i = lo; i < hi; ++i) { chunk_base[i] = run; run += (tasks[i].count + kChunk - 1u) / kChunk; }
    if (threadIdx.x == 1023u) { chunk_base[nt] = s_sum[1023]; out2[0] = nt; out2[1] = s_sum[1023]; }
}

__global__ __launch_bounds__(256) void k_bin(const Task* tasks, uint32_t nt, const uint32_t* chunk_base, uint32_t* chunk_task, const uint32_t* order0, const uint32_t* order1,
                                             const float* ref_box, uint32_t* gmn, uint32_t* gmx, uint32_t* gcnt, uint32_t* chunk_cnt) {
    __shared__ uint32_t smn[kBinWords], smx[kBinWords], scnt[96];
    const uint32_t c = blockIdx.x;
    uint32_t lo_t = 0, hi_t = nt; // last task whose chunk_base <= c
    while (hi_t - lo_t > 1u) { const uint32_t m = (lo_t + hi_t) >> 1; if (chunk_base[m] <= c) lo_t = m; else hi_t = m; }
    const uint32_t t = lo_t;
    const Task tk = tasks[t];
    if (threadIdx.x == 0) chunk_task[c] = t;
    for (uint32_t i = threadIdx.x; i < (uint32_t)kBinWords; i += 256u) { smn[i] = 0xffffffffu; smx[i] = 0u; }
    if (threadIdx.x < 96u) scnt[threadIdx.x] = 0u;
    __syncthreads();
    float scale[3]; bool use[3];
    for (int a = 0; a < 3; ++a) { use[a] = tk.cmx[a] > tk.cmn[a]; scale[a] = use[a] ? (float)kSahBins / (tk.cmx[a] - tk.cmn[a]) : 0.0f; }
    const uint32_t* order = (tk.flags & kTaskBuf) ? order1 : order0;
    const uint32_t lo = tk.first + (c - chunk_base[t]) * kChunk, hi = min(tk.first + tk.count, lo + kChunk);
    for (uint32_t pos = lo + threadIdx.x; pos < hi; pos += 256u) {
        float mn[3], mx[3], ce[3]; load_box(ref_box, order[pos], mn, mx);
        for (int a = 0; a < 3; ++a) ce[a] = 0.5f * mn[a] + 0.5f * mx[a];
        for (int axis = 0; axis < 3; ++axis) {
            if (!use[axis]) continue;
            const int idx = axis * 32 + bin_of(ce[axis], tk.cmn[axis], scale[axis]);
            for (int a = 0; a < 3; ++a) {
                lds_min(&smn[idx * 6 + a], enc(mn[a])); lds_min(&smn[idx * 6 + 3 + a], enc(ce[a]));
                lds_max(&smx[idx * 6 + a], enc(mx[a])); lds_max(&smx[idx * 6 + 3 + a], enc(ce[a]));
            }
            atomicAdd(&scnt[idx], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 96u) {
        const uint32_t i = threadIdx.x, n = scnt[i];
        chunk_cnt[(size_t)c * 96u + i] = n;
        if (n) {
            for (int a = 0; a < 6; ++a) { atomicMin(&gmn[(size_t)t * kBinWords + i * 6 + a], smn[i * 6 + a]); atomicMax(&gmx[(size_t)t * kBinWords + i * 6 + a], smx[i * 6 + a]); }
            atomicAdd(&gcnt[(size_t)t * 96u + i], n);
        }
    }
}

// Bounds of order[lo, hi) by one wave (the split-by-index case: the halves are not unions of bins).
__device__ void range_bounds_global(const uint32_t* order, const float* ref_box, uint32_t lo, uint32_t hi, float v[12]) {
    init12(v);
    for (uint32_t p = lo + (uint32_t)lane_id(); p < hi; p += 64u) { float mn[3], mx[3]; load_box(ref_box, order[p], mn, mx); acc12(v, mn, mx); }
    wave_reduce12(v);
}

__global__ __launch_bounds__(256) void k_select(const Task* tasks, uint32_t nt, const uint32_t* gmn, const uint32_t* gmx, const uint32_t* gcnt, SplitInfo* split, Task* next, Task* small,
                                                uint32_t small_cap, Node2* node2, const uint32_t* order0, const uint32_t* order1, uint32_t* order_final, const float* ref_box, Counters* ctr,
                                                int max_leaf, float prim_cost) {
    const uint32_t t = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (t >= nt) return;
    const Task tk = tasks[t];
    SplitResult r;
    select_split(gmn + (size_t)t * kBinWords, gmx + (size_t)t * kBinWords, gcnt + (size_t)t * 96u, tk, max_leaf, prim_cost, r);
    const uint32_t* order = (tk.flags & kTaskBuf) ? order1 : order0;
    const int lane = lane_id();
    SplitInfo si; si.axis = -1; si.split = 0; si.mid = 0; si.lo = 0.f; si.scale = 0.f;
    if (r.leaf) { // (only when kSmall < max_leaf; kept for completeness)
        for (uint32_t i = (uint32_t)lane; i < tk.count; i += 64u) order_final[tk.first + i] = order[tk.first + i];
        if (lane == 0) { set_ref(node2, ctr, tk.parent, tk.flags, make_leaf_ref(tk.first, tk.count)); split[t] = si; }
        return;
    }
    uint32_t mid, child_buf = tk.flags & kTaskBuf;
    float L[12], R[12];
    if (r.axis < 0) { // all centroids coincide: split by index, nothing moves
        mid = tk.first + tk.count / 2u;
        range_bounds_global(order, ref_box, tk.first, mid, L);
        range_bounds_global(order, ref_box, mid, tk.first + tk.count, R);
    } else {
        mid = tk.first + r.nleft;
        for (int a = 0; a < 3; ++a) {
            L[a] = r.lb_mn[a]; L[3 + a] = r.lc_mn[a]; L[6 + a] = r.lb_mx[a]; L[9 + a] = r.lc_mx[a];
            R[a] = r.rb_mn[a]; R[3 + a] = r.rc_mn[a]; R[6 + a] = r.rb_mx[a]; R[9 + a] = r.rc_mx[a];
        }
        si.axis = r.axis; si.split = r.split; si.mid = mid; si.lo = sel3(tk.cmn, r.axis); si.scale = (float)kSahBins / (sel3(tk.cmx, r.axis) - sel3(tk.cmn, r.axis));
        child_buf ^= kTaskBuf;
    }
    if (lane != 0) return;
    split[t] = si;
    const int32_t me = (int32_t)(mid - 1u);
    Node2 n;
    for (int a = 0; a < 3; ++a) { n.lmin[a] = L[a]; n.lmax[a] = L[6 + a]; n.rmin[a] = R[a]; n.rmax[a] = R[6 + a]; }
    n.left = kEmptyChild; n.right = kEmptyChild;
    node2[me] = n;
    set_ref(node2, ctr, tk.parent, tk.flags, me);
    atomicAdd(&ctr->n_binary, 1u);
    for (int side = 0; side < 2; ++side) {
        const float* B = side ? R : L;
        Task c; make_task(c, side ? mid : tk.first, side ? tk.first + tk.count - mid : mid - tk.first, me, child_buf | (side ? kTaskSide : 0u), B, B + 6, B + 3, B + 9);
        if (c.count > kSmall) next[atomicAdd(&ctr->n_next, 1u)] = c;
        else { const uint32_t k = atomicAdd(&ctr->n_small, 1u); if (k < small_cap) small[k] = c; else ctr->overflow = 1u; }
    }
}

// Exclusive prefix sum of n 32-bit counts (the per-triangle reference counts of the pre-splitting passes) in three launches: k_scan_sums (one sum per block of
// kScanBlock items), k_scan_blocks (ONE workgroup scans the block sums in place), k_scan_apply (every block scans its items from its base).  A thread owns 16
// consecutive items; the block's 256 thread sums are scanned through LDS.  Wrap-around arithmetic like any u32 sum (the caller bounds the total).
constexpr uint32_t kScanItems = 16u, kScanBlock = 256u * kScanItems;
__global__ __launch_bounds__(256) void k_scan_sums(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ block_sum) {
    __shared__ uint32_t s_w[4];
    const uint32_t base = blockIdx.x * kScanBlock + threadIdx.x * kScanItems;
    uint32_t sum = 0;
    for (uint32_t k = 0; k < kScanItems; ++k) if (base + k < n) sum += in[base + k];
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if (lane_id() == 0) s_w[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) block_sum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(1024) void k_scan_blocks(uint32_t* block_sum, uint32_t nb) { // in place: block_sum[b] becomes the sum of the blocks before b
    __shared__ uint32_t s_sum[1024];
    const uint32_t per = (nb + 1023u) / 1024u, lo = threadIdx.x * per, hi = min(nb, lo + per);
    uint32_t sum = 0;
    for (uint32_t b = lo; b < hi; ++b) sum += block_sum[b];
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) { uint32_t v = threadIdx.x >= off ? s_sum[threadIdx.x - off] : 0u; __syncthreads(); s_sum[threadIdx.x] += v; __syncthreads(); }
    uint32_t run = s_sum[threadIdx.x] - sum;
    for (uint32_t b = lo; b < hi; ++b) { const uint32_t v = block_sum[b]; block_sum[b] = run; run += v; }
}
__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t* __restrict__ in, uint32_t n, const uint32_t* __restrict__ block_base, uint32_t* __restrict__ out) {
    __shared__ uint32_t s_t[256];
    const uint32_t base = blockIdx.x * kScanBlock + threadIdx.x * kScanItems;
    uint32_t v[kScanItems], sum = 0;
    for (uint32_t k = 0; k < kScanItems; ++k) { v[k] = base + k < n ? in[base + k] : 0u; sum += v[k]; }
    s_t[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 256u; off <<= 1) { uint32_t q = threadIdx.x >= off ? s_t[threadIdx.x - off] : 0u; __syncthreads(); s_t[threadIdx.x] += q; __syncthreads(); }
    uint32_t run = block_base[blockIdx.x] + s_t[threadIdx.x] - sum;
    for (uint32_t k = 0; k < kScanItems; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
}
static hipError_t exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* block_tmp) {
    const uint32_t nb = (n + kScanBlock - 1u) / kScanBlock;
    if (nb == 0u) return hipSuccess;
    hipLaunchKernelGGL(k_scan_sums, dim3(nb), dim3(256), 0, 0, in, n, block_tmp);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, 0, block_tmp, nb);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(256), 0, 0, in, n, block_tmp, out);
    return hipGetLastError();
}

// lefts[c] = references of chunk c that go left (from the chunk's own bin counts: no pass over the references).
__global__ __launch_bounds__(256) void k_chunk_lefts(const uint32_t* chunk_task, const uint32_t* chunk_cnt, const SplitInfo* split, uint32_t nchunks, uint32_t* lefts) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= nchunks) return;
    const SplitInfo si = split[chunk_task[c]];
    uint32_t s = 0;
    if (si.axis >= 0) for (int b = 0; b <= si.split; ++b) s += chunk_cnt[(size_t)c * 96u + (uint32_t)(si.axis * 32 + b)];
    lefts[c] = s;
}
// chunk_off[c] = references of chunk c's task that go left and sit in earlier chunks of that task; one workgroup.
__global__ __launch_bounds__(1024) void k_part_scan(const uint32_t* chunk_task, const uint32_t* chunk_base, const uint32_t* lefts, uint32_t nchunks, uint32_t* chunk_off, uint32_t* scratch) {
    __shared__ uint32_t s_sum[1024];
    const uint32_t per = (nchunks + 1023u) / 1024u, lo = threadIdx.x * per, hi = min(nchunks, lo + per);
    uint32_t sum = 0;
    for (uint32_t c = lo; c < hi; ++c) sum += lefts[c];
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) { uint32_t v = threadIdx.x >= off ? s_sum[threadIdx.x - off] : 0u; __syncthreads(); s_sum[threadIdx.x] += v; __syncthreads(); }
    uint32_t run = s_sum[threadIdx.x] - sum;
    for (uint32_t c = lo; c < hi; ++c) { scratch[c] = run; run += lefts[c]; }
    __threadfence_block();
    __syncthreads();
    for (uint32_t c = lo; c < hi; ++c) chunk_off[c] = scratch[c] - scratch[chunk_base[chunk_task[c]]];
}

__global__ __launch_bounds__(256) void k_scatter(const Task* tasks, const uint32_t* chunk_task, const uint32_t* chunk_base, const uint32_t* chunk_off, const SplitInfo* split,
                                                 uint32_t* order0, uint32_t* order1, const float* ref_box) {
    __shared__ uint32_t wl[4], wr[4];
    const uint32_t c = blockIdx.x, t = chunk_task[c];
    const SplitInfo si = split[t];
    if (si.axis < 0) return;
    const Task tk = tasks[t];
    const uint32_t k = c - chunk_base[t];
    const uint32_t lo = tk.first + k * kChunk, hi = min(tk.first + tk.count, lo + kChunk);
    const uint32_t* src = (tk.flags & kTaskBuf) ? order1 : order0;
    uint32_t* dst = (tk.flags & kTaskBuf) ? order0 : order1;
    uint32_t lbase = tk.first + chunk_off[c], rbase = si.mid + (k * kChunk - chunk_off[c]);
    const uint32_t w = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane_id()) - 1ull;
    for (uint32_t base = lo; base < hi; base += 256u) {
        const uint32_t pos = base + threadIdx.x;
        const bool valid = pos < hi;
        uint32_t id = 0; bool left = false;
        if (valid) {
            id = src[pos];
            const float* b = ref_box + 6 * (size_t)id;
            const float ce = 0.5f * b[si.axis] + 0.5f * b[3 + si.axis];
            left = bin_of(ce, si.lo, si.scale) <= si.split;
        }
        const unsigned long long ml = __ballot(valid && left), mr = __ballot(valid && !left);
        if (lane_id() == 0) { wl[w] = (uint32_t)__popcll(ml); wr[w] = (uint32_t)__popcll(mr); }
        __syncthreads();
        uint32_t lp = 0, rp = 0, lsum = 0, rsum = 0;
        for (uint32_t q = 0; q < 4u; ++q) { if (q < w) { lp += wl[q]; rp += wr[q]; } lsum += wl[q]; rsum += wr[q]; }
        if (valid) {
            if (left) dst[lbase + lp + (uint32_t)__popcll(ml & lt)] = id;
            else dst[rbase + rp + (uint32_t)__popcll(mr & lt)] = id;
        }
        lbase += lsum; rbase += rsum;
        __syncthreads();
    }
}

// One wave finishes the subtree of a node of <= kSmall references: the references' boxes live in LDS, `perm` holds their current
// order, every node of the subtree is binned (LDS atomics), decided (select_split) and partitioned (ballots) by the 64 lanes.
__global__ __launch_bounds__(256) void k_small(const Task* small, uint32_t ns, const uint32_t* order0, const uint32_t* order1, uint32_t* order_final, const float* ref_box,
                                               Node2* node2, Counters* ctr, int max_leaf, float prim_cost) {
    __shared__ uint32_t s_gid[4][kSmall];
    __shared__ float s_box[4][6][kSmall];
    __shared__ uint8_t s_perm[4][kSmall], s_perm2[4][kSmall]; // positions inside the node's <= 256 references
    static_assert(kSmall <= 256, "a reference's slot must fit a byte");
    __shared__ uint32_t s_mn[4][kBinWords], s_mx[4][kBinWords], s_cnt[4][96];
    __shared__ Task s_stack[4][9]; // the larger child waits: depth <= log2(kSmall)
    const uint32_t w = threadIdx.x >> 6;
    const uint32_t ti = blockIdx.x * 4u + w;
    if (ti >= ns) return;
    const int lane = lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t* gid = s_gid[w]; uint8_t* perm = s_perm[w]; uint8_t* perm2 = s_perm2[w];
    uint32_t* bmn = s_mn[w]; uint32_t* bmx = s_mx[w]; uint32_t* bcnt = s_cnt[w];
    Task cur = small[ti];
    const uint32_t gfirst = cur.first;
    PHASE_BEGIN();
    {
        const uint32_t* order = (cur.flags & kTaskBuf) ? order1 : order0;
        for (uint32_t i = (uint32_t)lane; i < cur.count; i += 64u) {
            const uint32_t id = order[gfirst + i];
            gid[i] = id; perm[i] = (uint8_t)i;
            float mn[3], mx[3]; load_box(ref_box, id, mn, mx);
            for (int a = 0; a < 3; ++a) { s_box[w][a][i] = mn[a]; s_box[w][3 + a][i] = mx[a]; }
        }
    }
    cur.first = 0; // local from here on
    wave_sync();
    PHASE(0);
    int sp = 0; uint32_t splits = 0;
    for (;;) {
        bool done = false;
        if (cur.count == 1u) {
            if (lane == 0) { order_final[gfirst + cur.first] = gid[perm[cur.first]]; set_ref(node2, ctr, cur.parent, cur.flags, make_leaf_ref(gfirst + cur.first, 1u)); }
            done = true;
        } else if (cur.count == 2u) {
            // Two references, decided without bins — the same decision select_split() takes: on every axis whose centroids differ the two
            // fall into bins 0 and 31, every candidate plane between them costs area(a) + area(b), and the first one in (axis, bin) order
            // wins: the first such axis, left = the reference with the smaller centroid there.  No such axis: the two coincide -> a leaf
            // (count <= max_leaf and no split to compare with), as the sequential builder decides.
            const uint32_t e0 = perm[cur.first], e1 = perm[cur.first + 1u];
            float m0[3], x0[3], m1[3], x1[3], c0[3], c1[3];
            for (int a = 0; a < 3; ++a) { m0[a] = s_box[w][a][e0]; x0[a] = s_box[w][3 + a][e0]; m1[a] = s_box[w][a][e1]; x1[a] = s_box[w][3 + a][e1];
                                          c0[a] = 0.5f * m0[a] + 0.5f * x0[a]; c1[a] = 0.5f * m1[a] + 0.5f * x1[a]; }
            int axis = -1;
            for (int a = 2; a >= 0; --a) if (cur.cmx[a] > cur.cmn[a]) axis = a;
            bool leaf = axis < 0 && 2 <= max_leaf;
            const float ha = half_area3(cur.bmn, cur.bmx);
            if (axis >= 0 && 2 <= max_leaf) {
                const float best_cost = half_area3(m0, x0) * 1.0f + half_area3(m1, x1) * 1.0f; // area * (float)1 + area * (float)1: the order of the operands does not matter
                const float leaf_cost = ha * 2.0f * prim_cost, split_cost = best_cost * prim_cost + ha * 1.0f;
                if (!(split_cost < leaf_cost)) leaf = true;
            }
            if (leaf) {
                if (lane < 2) order_final[gfirst + cur.first + (uint32_t)lane] = gid[lane ? e1 : e0];
                if (lane == 0) set_ref(node2, ctr, cur.parent, cur.flags, make_leaf_ref(gfirst + cur.first, 2u));
            } else {
                // (axis < 0 with max_leaf < 2: split by index — the order stays)
                const bool swap = axis >= 0 && sel3(c1, axis) < sel3(c0, axis); // the reference in bin 0 goes left; stable otherwise
                const uint32_t l = swap ? e1 : e0, r = swap ? e0 : e1;
                const int32_t me = (int32_t)(gfirst + cur.first);
                if (lane == 0) {
                    Node2 n;
                    for (int a = 0; a < 3; ++a) { n.lmin[a] = swap ? m1[a] : m0[a]; n.lmax[a] = swap ? x1[a] : x0[a]; n.rmin[a] = swap ? m0[a] : m1[a]; n.rmax[a] = swap ? x0[a] : x1[a]; }
                    n.left = make_leaf_ref(gfirst + cur.first, 1u); n.right = make_leaf_ref(gfirst + cur.first + 1u, 1u);
                    node2[me] = n;
                    set_ref(node2, ctr, cur.parent, cur.flags, me);
                    order_final[gfirst + cur.first] = gid[l]; order_final[gfirst + cur.first + 1u] = gid[r];
                }
                ++splits;
            }
            done = true;
        } else {
            for (uint32_t i = (uint32_t)lane; i < (uint32_t)kBinWords; i += 64u) { bmn[i] = 0xffffffffu; bmx[i] = 0u; }
            for (uint32_t i = (uint32_t)lane; i < 96u; i += 64u) bcnt[i] = 0u;
            wave_sync();
            PHASE(1);
            float scale[3]; bool use[3];
            for (int a = 0; a < 3; ++a) { use[a] = cur.cmx[a] > cur.cmn[a]; scale[a] = use[a] ? (float)kSahBins / (cur.cmx[a] - cur.cmn[a]) : 0.0f; }
            for (uint32_t i = (uint32_t)lane; i < cur.count; i += 64u) {
                const uint32_t e = perm[cur.first + i];
                float mn[3], mx[3], ce[3];
                for (int a = 0; a < 3; ++a) { mn[a] = s_box[w][a][e]; mx[a] = s_box[w][3 + a][e]; ce[a] = 0.5f * mn[a] + 0.5f * mx[a]; }
                for (int axis = 0; axis < 3; ++axis) {
                    if (!use[axis]) continue;
                    const int idx = axis * 32 + bin_of(ce[axis], cur.cmn[axis], scale[axis]);
                    for (int a = 0; a < 3; ++a) { // (no read-before-atomic filter here: a node's few references rarely repeat a bin — measured 37 -> 47 ms)
                        atomicMin(&bmn[idx * 6 + a], enc(mn[a])); atomicMin(&bmn[idx * 6 + 3 + a], enc(ce[a]));
                        atomicMax(&bmx[idx * 6 + a], enc(mx[a])); atomicMax(&bmx[idx * 6 + 3 + a], enc(ce[a]));
                    }
                    atomicAdd(&bcnt[idx], 1u);
                }
            }
            wave_sync();
            PHASE(2);
            SplitResult r;
            select_split(bmn, bmx, bcnt, cur, max_leaf, prim_cost, r);
            PHASE(3);
            if (r.leaf) {
                for (uint32_t i = (uint32_t)lane; i < cur.count; i += 64u) order_final[gfirst + cur.first + i] = gid[perm[cur.first + i]];
                if (lane == 0) set_ref(node2, ctr, cur.parent, cur.flags, make_leaf_ref(gfirst + cur.first, cur.count));
                done = true;
            } else {
                uint32_t mid; float L[12], R[12];
                if (r.axis < 0) {
                    mid = cur.first + cur.count / 2u;
                    init12(L); init12(R);
                    for (uint32_t i = (uint32_t)lane; i < cur.count; i += 64u) {
                        const uint32_t e = perm[cur.first + i];
                        float mn[3], mx[3];
                        for (int a = 0; a < 3; ++a) { mn[a] = s_box[w][a][e]; mx[a] = s_box[w][3 + a][e]; }
                        if (cur.first + i < mid) acc12(L, mn, mx); else acc12(R, mn, mx);
                    }
                    wave_reduce12(L); wave_reduce12(R);
                } else {
                    mid = cur.first + r.nleft;
                    for (int a = 0; a < 3; ++a) {
                        L[a] = r.lb_mn[a]; L[3 + a] = r.lc_mn[a]; L[6 + a] = r.lb_mx[a]; L[9 + a] = r.lc_mx[a];
                        R[a] = r.rb_mn[a]; R[3 + a] = r.rc_mn[a]; R[6 + a] = r.rb_mx[a]; R[9 + a] = r.rc_mx[a];
                    }
                    // stable partition of perm[first, first + count) by bin <= split
                    uint32_t l = 0, rr = 0;
                    const float plo = sel3(cur.cmn, r.axis), pscale = (float)kSahBins / (sel3(cur.cmx, r.axis) - sel3(cur.cmn, r.axis));
                    for (uint32_t base = 0; base < cur.count; base += 64u) {
                        const uint32_t i = base + (uint32_t)lane;
                        const bool valid = i < cur.count;
                        uint32_t e = 0; bool left = false;
                        if (valid) {
                            e = perm[cur.first + i];
                            const float ce = 0.5f * s_box[w][r.axis][e] + 0.5f * s_box[w][3 + r.axis][e];
                            left = bin_of(ce, plo, pscale) <= r.split;
                        }
                        const unsigned long long ml = __ballot(valid && left), mr = __ballot(valid && !left);
                        if (valid) {
                            if (left) perm2[cur.first + l + (uint32_t)__popcll(ml & lt)] = (uint8_t)e;
                            else perm2[mid + rr + (uint32_t)__popcll(mr & lt)] = (uint8_t)e;
                        }
                        l += (uint32_t)__popcll(ml); rr += (uint32_t)__popcll(mr);
                    }
                    wave_sync();
                    for (uint32_t i = (uint32_t)lane; i < cur.count; i += 64u) perm[cur.first + i] = perm2[cur.first + i];
                    wave_sync();
                }
                PHASE(4);
                const int32_t me = (int32_t)(gfirst + mid - 1u);
                if (lane == 0) {
                    Node2 n;
                    for (int a = 0; a < 3; ++a) { n.lmin[a] = L[a]; n.lmax[a] = L[6 + a]; n.rmin[a] = R[a]; n.rmax[a] = R[6 + a]; }
                    n.left = kEmptyChild; n.right = kEmptyChild;
                    node2[me] = n;
                    set_ref(node2, ctr, cur.parent, cur.flags, me);
                }
                ++splits;
                Task lc, rc;
                make_task(lc, cur.first, mid - cur.first, me, 0u, L, L + 6, L + 3, L + 9);
                make_task(rc, mid, cur.first + cur.count - mid, me, kTaskSide, R, R + 6, R + 3, R + 9);
                // the larger child waits on the stack (depth <= log2 kSmall), the smaller one is next
                const bool left_first = lc.count <= rc.count;
                if (lane == 0) s_stack[w][sp] = left_first ? rc : lc;
                ++sp;
                cur = left_first ? lc : rc;
                wave_sync();
            }
        }
        PHASE(5);
        if (done) {
            if (sp == 0) break;
            --sp;
            cur = s_stack[w][sp];
            wave_sync();
        }
        PHASE(6);
    }
    PHASE_END(ctr);
    if (lane == 0 && splits) atomicAdd(&ctr->n_binary, splits);
}

// ---- stage 4: collapse -------------------------------------------------------------------------------------------------
struct Tmp4 { float mn[4][3], mx[4][3]; int32_t ref[4]; uint32_t n, size, dfs, broot; };
static_assert(sizeof(Tmp4) == 128, "Tmp4 layout");

// Level by level: node `id` (rooted at binary node tmp[id].broot) takes its root's two children and opens the internal child with
// the largest box until four slots are used — Collapser::collapse of bvh_build.cpp — and hands out ids to its internal children.
__global__ __launch_bounds__(256) void k_collapse_level(const Node2* node2, Tmp4* tmp, uint32_t start, uint32_t end, Counters* ctr) {
    const uint32_t id = start + blockIdx.x * 256u + threadIdx.x;
    const bool active = id < end;
    float mn[4][3], mx[4][3]; int32_t ref[4]; int n = 0;
    auto push_children = [&](int32_t node) {
        const Node2 b = node2[node];
        for (int a = 0; a < 3; ++a) { mn[n][a] = b.lmin[a]; mx[n][a] = b.lmax[a]; mn[n + 1][a] = b.rmin[a]; mx[n + 1][a] = b.rmax[a]; }
        ref[n] = b.left; ref[n + 1] = b.right; n += 2;
    };
    uint32_t m = 0;
    if (active) {
        push_children((int32_t)tmp[id].broot);
        while (n < 4) {
            int best = -1; float ba = -1.f;
            for (int k = 0; k < n; ++k) {
                const float dx = mx[k][0] - mn[k][0], dy = mx[k][1] - mn[k][1], dz = mx[k][2] - mn[k][2];
                const float ar = dx * dy + dy * dz + dz * dx;
                if (ref[k] >= 0 && ar > ba) { ba = ar; best = k; }
            }
            if (best < 0) break;
            const int32_t node = ref[best];
            for (int k = best; k + 1 < n; ++k) { for (int a = 0; a < 3; ++a) { mn[k][a] = mn[k + 1][a]; mx[k][a] = mx[k + 1][a]; } ref[k] = ref[k + 1]; }
            --n;
            push_children(node);
        }
        for (int k = 0; k < n; ++k) m += ref[k] >= 0 ? 1u : 0u;
    }
    // ids of the internal children: one atomic per wave
    uint32_t incl = m;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64); if (lane_id() >= off) incl += o; }
    const uint32_t total = bcast_u(incl, 63);
    uint32_t base = 0;
    if (lane_id() == 63 && total) base = atomicAdd(&ctr->next_id, total);
    base = bcast_u(base, 63);
    if (!active) return;
    uint32_t child = base + incl - m;
    Tmp4 out = tmp[id];
    out.n = (uint32_t)n; out.size = 1u;
    for (int k = 0; k < 4; ++k) {
        if (k < n) {
            for (int a = 0; a < 3; ++a) { out.mn[k][a] = mn[k][a]; out.mx[k][a] = mx[k][a]; }
            if (ref[k] >= 0) { tmp[child].broot = (uint32_t)ref[k]; out.ref[k] = (int32_t)child; ++child; } else out.ref[k] = ref[k];
        } else out.ref[k] = kEmptyChild;
    }
    tmp[id] = out;
}
__global__ __launch_bounds__(256) void k_sizes_level(Tmp4* tmp, uint32_t start, uint32_t end) {
    const uint32_t id = start + blockIdx.x * 256u + threadIdx.x;
    if (id >= end) return;
    uint32_t s = 1u;
    for (uint32_t k = 0; k < tmp[id].n; ++k) if (tmp[id].ref[k] >= 0) s += tmp[tmp[id].ref[k]].size;
    tmp[id].size = s;
}
// Depth-first positions (parent, then the subtrees of its internal children in slot order: the host Collapser's order) and the final nodes.
__global__ __launch_bounds__(256) void k_emit_level(Tmp4* tmp, uint32_t start, uint32_t end, BvhNode* out, int32_t node_base, uint32_t prim_base) {
    const uint32_t id = start + blockIdx.x * 256u + threadIdx.x;
    if (id >= end) return;
    const Tmp4 t = tmp[id];
    uint32_t run = t.dfs + 1u;
    BvhNode nd;
    const int lo_slot[3] = {0, 4, 3}, hi_slot[3] = {1, 6, 7};
    const float big = 3.402823466e+38f;
    for (int k = 0; k < 4; ++k) {
        int32_t ref;
        if (k < (int)t.n) {
            for (int a = 0; a < 3; ++a) { nd.slot[lo_slot[a]][k] = t.mn[k][a]; nd.slot[hi_slot[a]][k] = t.mx[k][a]; }
            if (t.ref[k] >= 0) { tmp[t.ref[k]].dfs = run; ref = (int32_t)run + node_base; run += tmp[t.ref[k]].size; }
            else { const uint32_t v = (uint32_t)~t.ref[k]; ref = make_leaf_ref((v >> 3) + prim_base, (v & 7u) + 1u); }
        } else {
            for (int a = 0; a < 3; ++a) { nd.slot[lo_slot[a]][k] = big; nd.slot[hi_slot[a]][k] = -big; }
            ref = kEmptyChild;
        }
        nd.slot[2][k] = __int_as_float(ref);
        nd.slot[5][k] = 0.0f;
    }
    out[t.dfs] = nd;
}

// ---- stage 5 -------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather(const uint32_t* order, const uint32_t* ref_tri, const TriRec* recs, const TriUv* uvs, uint32_t nrefs, TriRec* out_tris, TriUv* out_uvs) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nrefs) return;
    const uint32_t t = ref_tri[order[i]];
    const float4* s = reinterpret_cast<const float4*>(recs + t);
    float4* d = reinterpret_cast<float4*>(out_tris + i);
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
    const float2* su = reinterpret_cast<const float2*>(uvs + t);
    float2* du = reinterpret_cast<float2*>(out_uvs + i);
    du[0] = su[0]; du[1] = su[1]; du[2] = su[2];
}

// ---- host side -------------------------------------------------------------------------------------------------------------
struct Arena { // one allocation carved into 256-byte aligned pieces
    char* base = nullptr; size_t size = 0, used = 0;
    hipError_t reserve(size_t bytes) { size = bytes; used = 0; return hipMalloc((void**)&base, bytes); }
    template <typename T> T* take(size_t n) { used = (used + 255u) & ~(size_t)255u; T* p = (T*)(base + used); used += n * sizeof(T); return used <= size ? p : nullptr; }
    void release() { if (base) (void)hipFree(base); base = nullptr; }
};
template <typename T> size_t padded(size_t n) { return ((n * sizeof(T)) + 255u + 256u) & ~(size_t)255u; }

struct Stopwatch {
    bool on; std::chrono::steady_clock::time_point t0;
    explicit Stopwatch(bool enabled) : on(enabled), t0(std::chrono::steady_clock::now()) {}
    void lap(const char* what) {
        if (!on) return;
        (void)hipDeviceSynchronize();
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "  device build: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

#define DB_TRY(expr)                                                                                                     \
    do {                                                                                                                 \
        hipError_t e_ = (expr);                                                                                          \
        if (e_ != hipSuccess) { err = std::string("device BLAS build: ") + #expr + ": " + hipGetErrorString(e_); return e_ == hipErrorOutOfMemory ? NRAYS_ERR_OOM : NRAYS_ERR_HIP; } \
    } while (0)

int build_impl(const std::vector<DeviceMeshPart>& parts, const DeviceBuildOptions& opt, int32_t node_base, uint32_t prim_base, DeviceBlas& out, std::string& err,
               Arena& a1, Arena& a2, Arena& a3) {
    const bool verbose = getenv("NRAYS_BUILD_TIMES") != nullptr;
    Stopwatch sw(verbose);
    size_t n = 0;
    for (const DeviceMeshPart& p : parts) n += p.num_triangles;
    if (n == 0 || n >= (1u << 28)) { err = "device BLAS build: bad triangle count"; return NRAYS_ERR_BAD_ARG; }
    out.num_triangles = n;
    const uint32_t nblocks_tri = [&] { uint32_t b = 0; for (const DeviceMeshPart& p : parts) b += (p.num_triangles + 255u) / 256u; return b; }();
    const uint32_t split_grid_max = getenv("NRAYS_SPLIT_GRID") ? (uint32_t)std::max(1, atoi(getenv("NRAYS_SPLIT_GRID"))) : 512u; // workgroups of k_presplit (each thread owns a 6.9 KB stack of frames)
    const uint32_t split_grid = (uint32_t)std::min<size_t>(split_grid_max, (n + 255) / 256);

    // ---- phase 1: inputs, triangle records, pre-split counts ----
    std::map<const void*, std::pair<size_t, void*>> uploads; // host array -> (bytes, device copy): meshes alias their vertex arrays
    for (const DeviceMeshPart& p : parts) {
        if (!p.vertices || !p.indices) { err = "device BLAS build: null mesh array"; return NRAYS_ERR_BAD_ARG; }
        auto note = [&](const void* h, size_t bytes) { if (h) { auto& u = uploads[h]; u.first = std::max(u.first, bytes); u.second = nullptr; } };
        note(p.vertices, (size_t)p.num_vertices * 24u); note(p.uvs, (size_t)p.num_vertices * 16u); note(p.indices, (size_t)p.num_triangles * 12u);
    }
    size_t bytes1 = 0;
    for (auto& kv : uploads) bytes1 += padded<char>(kv.second.first);
    // the per-thread stacks of k_presplit (0.9 GB from 131 k triangles on) exist only if pre-splitting can run at all
    const bool may_split = opt.presplit && n >= 64 && opt.budget > 0.0;
    const size_t frame_count = may_split ? (size_t)split_grid * 256u * (kSplitDepthMax + 1) : 1u;
    const size_t scan_bytes = ((n + 1 + kScanBlock - 1) / kScanBlock) * sizeof(uint32_t); // block sums of the prefix sum over the per-triangle reference counts (exclusive_scan_u32)
    // one-walk pre-splitting: the pieces of the last counting pass are kept (PieceList): 1.25 x the larger budget in all, dealt to the workgroups' regions (their triangles are
    // a uniform sample of the mesh); NRAYS_PRESPLIT_ONE_WALK=0: walk twice as before (A/B)
    const bool one_walk = may_split && !(getenv("NRAYS_PRESPLIT_ONE_WALK") && atoi(getenv("NRAYS_PRESPLIT_ONE_WALK")) == 0);
    // (NRAYS_DEBUG_PIECE_CAP=c: c places per region — tests: the regions overflow and the second walk takes over)
    const uint32_t region_cap = !one_walk ? 0u : getenv("NRAYS_DEBUG_PIECE_CAP") ? (uint32_t)std::max(1, atoi(getenv("NRAYS_DEBUG_PIECE_CAP")))
                                : (uint32_t)((size_t)(1.25 * std::max(opt.budget, opt.budget_hairy) * (double)n) / split_grid + 1024u);
    const size_t u_pieces = (size_t)region_cap * split_grid;
    bytes1 += padded<TriRec>(n) + padded<TriUv>(n) + padded<float>(6 * n) + 2 * padded<double>(nblocks_tri) + padded<Counters>(1) + 2 * padded<uint32_t>(n + 1) +
              padded<uint32_t>(kHistBins) + padded<SplitFrame>(frame_count) + padded<char>(scan_bytes) + padded<PartDev>(parts.size()) + 4096;
    if (one_walk) bytes1 += padded<float>(6 * u_pieces) + padded<uint32_t>(2 * u_pieces) + padded<float>(6 * n) + padded<uint32_t>(split_grid + 1u);
    DB_TRY(a1.reserve(bytes1));
    sw.lap("allocation (phase 1)");
    {   // A merged group (the sponza stand-in: 270 meshes under one isometry, ~800 small arrays) paid one synchronous hipMemcpy per array — 4-5 ms of
        // a 12 ms build.  Arrays below 1 MB are staged into one host buffer in the order the arena hands out their device pieces (consecutive, 256-byte
        // aligned) and go over in ONE copy per run; large arrays are copied from where they lie.
        std::vector<char> stage;
        char* run_dst = nullptr;
        auto flush = [&]() -> hipError_t {
            hipError_t e = hipSuccess;
            if (run_dst && !stage.empty()) e = hipMemcpy(run_dst, stage.data(), stage.size(), hipMemcpyHostToDevice);
            stage.clear(); run_dst = nullptr;
            return e;
        };
        size_t small_arrays = 0, small_bytes = 0;
        for (auto& kv : uploads) {
            const size_t bytes = kv.second.first;
            char* dst = a1.take<char>(bytes);
            kv.second.second = dst;
            if (!dst) { err = "device BLAS build: arena overflow (uploads)"; return NRAYS_ERR_OOM; }
            if (bytes >= (1u << 20)) {
                DB_TRY(flush());
                DB_TRY(hipMemcpy(dst, kv.first, bytes, hipMemcpyHostToDevice));
                if (verbose) { char what[64]; snprintf(what, sizeof what, "  upload of %.1f MB", bytes / 1048576.0); sw.lap(what); }
                continue;
            }
            if (!run_dst) run_dst = dst;
            const size_t at = (size_t)(dst - run_dst); // the arena's pieces are consecutive: the staging buffer mirrors their layout
            if (stage.size() < at + bytes) stage.resize(at + bytes);
            std::memcpy(stage.data() + at, kv.first, bytes);
            ++small_arrays; small_bytes += bytes;
        }
        DB_TRY(flush());
        if (verbose && small_arrays) { char what[96]; snprintf(what, sizeof what, "  upload of %zu small arrays (%.1f MB) in staged runs", small_arrays, small_bytes / 1048576.0); sw.lap(what); }
    }
    TriRec* recs = a1.take<TriRec>(n); TriUv* uvs = a1.take<TriUv>(n); float* tbox = a1.take<float>(6 * n);
    double* part_area = a1.take<double>(nblocks_tri); double* part_tri2 = a1.take<double>(nblocks_tri);
    Counters* ctr = a1.take<Counters>(1);
    uint32_t* counts = a1.take<uint32_t>(n + 1); uint32_t* offsets = a1.take<uint32_t>(n + 1); uint32_t* hist = a1.take<uint32_t>(kHistBins);
    SplitFrame* frames = a1.take<SplitFrame>(frame_count);
    void* scan_tmp1 = a1.take<char>(scan_bytes + 16);
    PartDev* dparts = a1.take<PartDev>(parts.size());
    if (!frames || !scan_tmp1 || !dparts) { err = "device BLAS build: arena overflow (phase 1)"; return NRAYS_ERR_OOM; }
    PieceList pieces; pieces.box = nullptr; pieces.key = nullptr; pieces.base_box = nullptr; pieces.count = nullptr; pieces.region_cap = region_cap;
    if (one_walk) {
        pieces.box = a1.take<float>(6 * u_pieces); pieces.key = a1.take<uint32_t>(2 * u_pieces); pieces.base_box = a1.take<float>(6 * n); pieces.count = a1.take<uint32_t>(split_grid + 1u);
        if (!pieces.count) { err = "device BLAS build: arena overflow (phase 1, piece list)"; return NRAYS_ERR_OOM; }
    }
    sw.lap("upload of the mesh arrays");
    Counters h_ctr; std::memset(&h_ctr, 0, sizeof h_ctr);
    for (int k = 0; k < 6; ++k) { h_ctr.bounds[k] = 0xffffffffu; h_ctr.bounds[6 + k] = 0u; }
    h_ctr.root_ref = kEmptyChild;
    DB_TRY(hipMemcpy(ctr, &h_ctr, sizeof h_ctr, hipMemcpyHostToDevice));
    {
        std::vector<PartDev> hp;
        uint32_t tri_base = 0, block_base = 0;
        for (const DeviceMeshPart& p : parts) {
            if (p.num_triangles == 0) continue;
            PartDev d; d.vertices = (const double*)uploads[p.vertices].second; d.uvs = p.uvs ? (const double*)uploads[p.uvs].second : nullptr;
            d.indices = (const uint32_t*)uploads[p.indices].second; d.num_vertices = p.num_vertices; d.num_triangles = p.num_triangles; d.node_id = p.node_id; d.tri_base = tri_base;
            d.block_base = block_base; d.pad = 0;
            hp.push_back(d);
            tri_base += p.num_triangles; block_base += (p.num_triangles + 255u) / 256u;
        }
        DB_TRY(hipMemcpy(dparts, hp.data(), hp.size() * sizeof(PartDev), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_tri_records, dim3(block_base), dim3(256), 0, 0, (const PartDev*)dparts, (uint32_t)hp.size(), recs, uvs, tbox, part_area, part_tri2, ctr);
    }
    DB_TRY(hipGetLastError());
    DB_TRY(hipMemcpy(&h_ctr, ctr, sizeof h_ctr, hipMemcpyDeviceToHost));
    if (h_ctr.err & 1u) { err = "triangle index out of range"; return NRAYS_ERR_BAD_ARG; }
    if (h_ctr.err & 2u) { err = "mesh vertex coordinate is not exactly representable in f32 (see DESIGN.md: f32-exact mesh storage)"; return NRAYS_ERR_UNSUPPORTED; }
    if (h_ctr.err & 4u) { err = "mesh uv is not exactly representable in f32"; return NRAYS_ERR_UNSUPPORTED; }
    auto dec_host = [](uint32_t e) { uint32_t b = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e; float f; std::memcpy(&f, &b, 4); return f; };
    for (int a = 0; a < 3; ++a) { out.mn[a] = dec_host(h_ctr.bounds[a]); out.mx[a] = dec_host(h_ctr.bounds[6 + a]); }
    sw.lap("triangle records");

    // pre-splitting: the rule and the constants of scene_build.cpp: presplit()
    size_t nrefs = n; bool hairy = false; double thr = 0.0; bool do_split = false;
    if (may_split) {
        std::vector<double> pa(nblocks_tri), pt(nblocks_tri);
        DB_TRY(hipMemcpy(pa.data(), part_area, nblocks_tri * sizeof(double), hipMemcpyDeviceToHost));
        DB_TRY(hipMemcpy(pt.data(), part_tri2, nblocks_tri * sizeof(double), hipMemcpyDeviceToHost));
        double total_area = 0.0, total_tri2 = 0.0;
        for (uint32_t b = 0; b < nblocks_tri; ++b) { total_area += pa[b]; total_tri2 += pt[b]; }
        hairy = total_area > 0.0 && (total_area - total_tri2) > opt.hairy_emptiness * total_area;
        const double min_gain = (hairy ? opt.min_gain_hairy : opt.min_gain) * total_area / (double)n;
        const size_t budget = (size_t)((hairy ? opt.budget_hairy : opt.budget) * (double)n);
        // Survey pass: every piece split against min_gain, but at most kSurveyDepth levels deep (a handful of huge degenerate triangles among small
        // ones would otherwise make single threads walk up to 2^20 pieces each before the budget is even known), with a histogram of the empty
        // areas of the splits.  If nothing was cut short and the count fits the budget, that is the answer.  Otherwise the threshold the budget
        // amounts to is read off the histogram and the split trees are walked again, full depth, against it — each such pass files its own
        // histogram, so a second correction (the first histogram missed what lay below the survey depth) lands within a bin of the budget.
        constexpr int kSurveyDepth = 10;
        uint32_t* capped = &ctr->pad;
        uint32_t extra = 0; thr = min_gain;
        auto pass = [&](double t, int cap) -> int {
            DB_TRY(hipMemset(hist, 0, kHistBins * sizeof(uint32_t)));
            DB_TRY(hipMemset(capped, 0, 4));
            if (pieces.count) DB_TRY(hipMemset(pieces.count + split_grid, 0, 4)); // the overflow mark of this pass's piece list
            hipLaunchKernelGGL(k_presplit<kModeHist>, dim3(split_grid), dim3(256), 0, 0, recs, tbox, (uint32_t)n, t, cap, frames, counts, offsets, hist, (float*)nullptr, (uint32_t*)nullptr, capped, pieces);
            DB_TRY(exclusive_scan_u32(counts, offsets, (uint32_t)(n + 1), (uint32_t*)scan_tmp1));
            DB_TRY(hipMemcpy(&extra, offsets + n, 4, hipMemcpyDeviceToHost));
            return NRAYS_OK;
        };
        { const int rc = pass(min_gain, kSurveyDepth); if (rc != NRAYS_OK) return rc; }
        uint32_t was_capped = 0;
        DB_TRY(hipMemcpy(&was_capped, capped, 4, hipMemcpyDeviceToHost));
        bool exact = !was_capped; // `counts` describe the full-depth walk against `thr`
        for (int round = 0; round < 6 && ((size_t)extra > budget || !exact); ++round) {
            if ((size_t)extra > budget) { // the budget binds: the threshold is the empty area below which the budget-th largest split falls
                std::vector<uint32_t> h(kHistBins);
                DB_TRY(hipMemcpy(h.data(), hist, kHistBins * sizeof(uint32_t), hipMemcpyDeviceToHost));
                size_t cum = 0; uint32_t b = kHistBins;
                while (b > 0 && cum + h[b - 1] <= budget) { cum += h[b - 1]; --b; } // bins >= b fit in the budget
                uint32_t bits = b << kHistShift; float edge; std::memcpy(&edge, &bits, 4);
                // (float)gain rounds to nearest: a piece just below the edge may have been filed above it; `gain > thr` is what every pass applies
                thr = std::max(std::max(min_gain, (double)edge), round ? thr * 1.05 : 0.0); // (strictly rising: the loop ends)
            }
            const int rc = pass(thr, kSplitDepthMax); if (rc != NRAYS_OK) return rc;
            exact = true;
        }
        if ((size_t)extra > 2 * budget + 1024) { err = "device BLAS build: pre-splitting does not settle on its budget"; return NRAYS_ERR_HIP; }
        nrefs = n + extra; do_split = extra > 0;
        if (pieces.count) { // did the last pass's pieces fit their regions?
            uint32_t over = 0;
            DB_TRY(hipMemcpy(&over, pieces.count + split_grid, 4, hipMemcpyDeviceToHost));
            if (over) { pieces.box = nullptr; if (verbose) fprintf(stderr, "  device build: piece list overflow, the split trees are walked a second time\n"); }
        }
    }
    out.hairy = hairy;
    if (nrefs + (size_t)prim_base >= (1u << 28)) { err = "too many triangles"; return NRAYS_ERR_UNSUPPORTED; }
    sw.lap("pre-split counts");

    // ---- phase 2: references, binary build ----
    const uint32_t R = (uint32_t)nrefs;
    const size_t cap_div = (size_t)std::max(1, opt.debug_cap_div);
    const size_t max_t = (R / (kSmall + 1u)) / cap_div + 2u, max_c = R / kChunk + max_t + 2u, small_cap = (R / 8u + 1024u) / cap_div + 1u;
    size_t bytes2 = padded<float>(6 * (size_t)R) + 3 * padded<uint32_t>(R) + padded<Node2>(R) + 2 * padded<Task>(max_t) + padded<Task>(small_cap) + 2 * padded<uint32_t>(max_t * kBinWords) +
                    padded<uint32_t>(max_t * 96u) + padded<SplitInfo>(max_t) + padded<uint32_t>(max_t + 1) + 4 * padded<uint32_t>(max_c) + padded<uint32_t>(max_c * 96u) + (1u << 16);
    DB_TRY(a2.reserve(bytes2));
    sw.lap("allocation (phase 2)");
    float* ref_box = a2.take<float>(6 * (size_t)R); uint32_t* ref_tri = a2.take<uint32_t>(R);
    uint32_t* order0 = a2.take<uint32_t>(R); uint32_t* order1 = a2.take<uint32_t>(R);
    Node2* node2 = a2.take<Node2>(R);
    Task* tasks[2] = {a2.take<Task>(max_t), a2.take<Task>(max_t)}; Task* small = a2.take<Task>(small_cap);
    uint32_t* gmn = a2.take<uint32_t>(max_t * kBinWords); uint32_t* gmx = a2.take<uint32_t>(max_t * kBinWords); uint32_t* gcnt = a2.take<uint32_t>(max_t * 96u);
    SplitInfo* split = a2.take<SplitInfo>(max_t); uint32_t* chunk_base = a2.take<uint32_t>(max_t + 1);
    uint32_t* chunk_task = a2.take<uint32_t>(max_c); uint32_t* chunk_off = a2.take<uint32_t>(max_c); uint32_t* scan_tmp = a2.take<uint32_t>(max_c); uint32_t* chunk_lefts = a2.take<uint32_t>(max_c);
    uint32_t* chunk_cnt = a2.take<uint32_t>(max_c * 96u);
    uint32_t* level2 = a2.take<uint32_t>(2); // k_chunk_scan_next: {tasks, chunks} of the next level
    if (!chunk_cnt || !level2) { err = "device BLAS build: arena overflow (phase 2)"; return NRAYS_ERR_OOM; }
    if (do_split && pieces.box) { // one walk: the kept pieces to their places (k_iota_refs: reference t = the unsplit piece of triangle t)
        hipLaunchKernelGGL(k_iota_refs, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, (uint32_t)n, (const float*)pieces.base_box, ref_box, ref_tri);
        hipLaunchKernelGGL(k_place_extra, dim3(split_grid), dim3(256), 0, 0, pieces, (const uint32_t*)offsets, (uint32_t)n, ref_box, ref_tri);
    } else if (do_split) { PieceList none; none.box = nullptr; none.key = nullptr; none.base_box = nullptr; none.count = nullptr; none.region_cap = 0u;
        hipLaunchKernelGGL(k_presplit<kModeEmit>, dim3(split_grid), dim3(256), 0, 0, recs, tbox, (uint32_t)n, thr, kSplitDepthMax, frames, counts, offsets, hist, ref_box, ref_tri, (uint32_t*)nullptr, none); }
    else hipLaunchKernelGGL(k_iota_refs, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, (uint32_t)n, tbox, ref_box, ref_tri);
    sw.lap("reference boxes");
    const float prim_cost = hairy ? opt.prim_cost_hairy : opt.prim_cost;
    const int max_leaf = std::min(std::max(opt.max_leaf, 1), 8);
    for (int k = 0; k < 6; ++k) { h_ctr.bounds[k] = 0xffffffffu; h_ctr.bounds[6 + k] = 0u; }
    h_ctr.n_next = h_ctr.n_small = h_ctr.overflow = h_ctr.n_binary = 0; h_ctr.root_ref = kEmptyChild;
    DB_TRY(hipMemcpy(ctr, &h_ctr, sizeof h_ctr, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_root_bounds, dim3(std::min<uint32_t>(1024u, (R + 255u) / 256u)), dim3(256), 0, 0, ref_box, R, order0, ctr);
    hipLaunchKernelGGL(k_root_task, dim3(1), dim3(1), 0, 0, tasks[0], small, R, ctr);
    uint32_t nt = R > kSmall ? 1u : 0u; int cur = 0, levels = 0;
    uint32_t nchunks = 0;
    if (nt > 0) { // the first level's chunks; every later level's come back with its task count (k_chunk_scan_next)
        hipLaunchKernelGGL(k_chunk_scan, dim3(1), dim3(1024), 0, 0, tasks[cur], nt, chunk_base);
        DB_TRY(hipMemcpy(&nchunks, chunk_base + nt, 4, hipMemcpyDeviceToHost));
    }
    while (nt > 0) {
        if (nt > max_t) { err = "device BLAS build: task list overflow"; return NRAYS_ERR_HIP; }
        if (nchunks == 0 || nchunks > max_c) { err = "device BLAS build: chunk list overflow"; return NRAYS_ERR_HIP; }
        DB_TRY(hipMemsetAsync(gmn, 0xff, (size_t)nt * kBinWords * 4u, 0));
        DB_TRY(hipMemsetAsync(gmx, 0, (size_t)nt * kBinWords * 4u, 0));
        DB_TRY(hipMemsetAsync(gcnt, 0, (size_t)nt * 96u * 4u, 0));
        DB_TRY(hipMemsetAsync(&ctr->n_next, 0, 4, 0));
        hipLaunchKernelGGL(k_bin, dim3(nchunks), dim3(256), 0, 0, tasks[cur], nt, chunk_base, chunk_task, order0, order1, ref_box, gmn, gmx, gcnt, chunk_cnt);
        hipLaunchKernelGGL(k_select, dim3((nt + 3u) / 4u), dim3(256), 0, 0, tasks[cur], nt, gmn, gmx, gcnt, split, tasks[cur ^ 1], small, (uint32_t)small_cap, node2, order0, order1, order0,
                           ref_box, ctr, max_leaf, prim_cost);
        hipLaunchKernelGGL(k_chunk_lefts, dim3((nchunks + 255u) / 256u), dim3(256), 0, 0, chunk_task, chunk_cnt, split, nchunks, chunk_lefts);
        hipLaunchKernelGGL(k_part_scan, dim3(1), dim3(1024), 0, 0, chunk_task, chunk_base, chunk_lefts, nchunks, chunk_off, scan_tmp);
        hipLaunchKernelGGL(k_scatter, dim3(nchunks), dim3(256), 0, 0, tasks[cur], chunk_task, chunk_base, chunk_off, split, order0, order1, ref_box);
        hipLaunchKernelGGL(k_chunk_scan_next, dim3(1), dim3(1024), 0, 0, tasks[cur ^ 1], (const uint32_t*)&ctr->n_next, (uint32_t)max_t, chunk_base, level2);
        uint32_t two[2] = {0u, 0u};
        DB_TRY(hipMemcpy(two, level2, 8, hipMemcpyDeviceToHost));
        nt = two[0]; nchunks = two[1];
        cur ^= 1; ++levels;
        if (levels > 4096) { err = "device BLAS build: the large-node phase does not end"; return NRAYS_ERR_HIP; }
    }
    DB_TRY(hipMemcpy(&h_ctr, ctr, sizeof h_ctr, hipMemcpyDeviceToHost));
    if (h_ctr.overflow || h_ctr.n_small > small_cap) { err = "device BLAS build: small-node list overflow"; return NRAYS_ERR_HIP; }
    if (verbose) fprintf(stderr, "  device build: %d large-node levels, %u small subtrees\n", levels, h_ctr.n_small);
    sw.lap("large nodes");
    if (h_ctr.n_small) hipLaunchKernelGGL(k_small, dim3((h_ctr.n_small + 3u) / 4u), dim3(256), 0, 0, small, h_ctr.n_small, order0, order1, order0, ref_box, node2, ctr, max_leaf, prim_cost);
    DB_TRY(hipGetLastError());
    DB_TRY(hipMemcpy(&h_ctr, ctr, sizeof h_ctr, hipMemcpyDeviceToHost));
#ifdef NR_BUILD_PHASES
    fprintf(stderr, "  k_small wave cycles: load %llu, clear %llu, bin %llu, select %llu, partition %llu, node+leaf writes %llu, pop %llu\n", h_ctr.phase[0], h_ctr.phase[1], h_ctr.phase[2],
            h_ctr.phase[3], h_ctr.phase[4], h_ctr.phase[5], h_ctr.phase[6]);
#endif
    sw.lap("small subtrees");

    // ---- phase 3: collapse, layout, gather ----
    DB_TRY(hipMalloc((void**)&out.tris, (size_t)R * sizeof(TriRec)));
    DB_TRY(hipMalloc((void**)&out.uvs, (size_t)R * sizeof(TriUv)));
    out.num_refs = R;
    hipLaunchKernelGGL(k_gather, dim3((R + 255u) / 256u), dim3(256), 0, 0, order0, ref_tri, recs, uvs, R, out.tris, out.uvs);
    if (h_ctr.root_ref < 0) { // a single leaf
        const uint32_t v = (uint32_t)~h_ctr.root_ref;
        out.root = h_ctr.root_ref == kEmptyChild ? kEmptyChild : make_leaf_ref((v >> 3) + prim_base, (v & 7u) + 1u);
        if (h_ctr.root_ref == kEmptyChild) { err = "device BLAS build: no root"; return NRAYS_ERR_HIP; }
        out.num_nodes = 0; out.max_depth = 0;
        DB_TRY(hipDeviceSynchronize());
        return NRAYS_OK;
    }
    const uint32_t nb = h_ctr.n_binary;
    DB_TRY(a3.reserve(padded<Tmp4>(nb) + 4096));
    Tmp4* tmp = a3.take<Tmp4>(nb);
    {
        Tmp4 root; std::memset(&root, 0, sizeof root); root.broot = (uint32_t)h_ctr.root_ref; root.dfs = 0;
        DB_TRY(hipMemcpy(tmp, &root, sizeof root, hipMemcpyHostToDevice));
        uint32_t one = 1; DB_TRY(hipMemcpy(&ctr->next_id, &one, 4, hipMemcpyHostToDevice));
    }
    std::vector<uint32_t> level_start{0u};
    uint32_t end = 1;
    while (level_start.back() < end) {
        const uint32_t s = level_start.back();
        hipLaunchKernelGGL(k_collapse_level, dim3((end - s + 255u) / 256u), dim3(256), 0, 0, node2, tmp, s, end, ctr);
        level_start.push_back(end);
        DB_TRY(hipMemcpy(&end, &ctr->next_id, 4, hipMemcpyDeviceToHost));
        if (end > nb) { err = "device BLAS build: more 4-wide nodes than binary nodes"; return NRAYS_ERR_HIP; }
        if (level_start.size() > 4096) { err = "device BLAS build: the collapse does not end"; return NRAYS_ERR_HIP; }
    }
    level_start.pop_back(); // the last entry opened an empty level
    const uint32_t n4 = end;
    for (size_t l = level_start.size(); l-- > 0;) {
        const uint32_t s = level_start[l], e = l + 1 < level_start.size() ? level_start[l + 1] : n4;
        if (e > s) hipLaunchKernelGGL(k_sizes_level, dim3((e - s + 255u) / 256u), dim3(256), 0, 0, tmp, s, e);
    }
    DB_TRY(hipMalloc((void**)&out.nodes, ((size_t)n4 + opt.node_tail) * sizeof(BvhNode)));
    out.num_nodes = n4; out.node_capacity = (size_t)n4 + opt.node_tail;
    for (size_t l = 0; l < level_start.size(); ++l) {
        const uint32_t s = level_start[l], e = l + 1 < level_start.size() ? level_start[l + 1] : n4;
        if (e > s) hipLaunchKernelGGL(k_emit_level, dim3((e - s + 255u) / 256u), dim3(256), 0, 0, tmp, s, e, out.nodes, node_base, prim_base);
    }
    out.root = node_base; // the root of the collapse is node 0 of this BLAS
    out.max_depth = (int)level_start.size() - 1;
    DB_TRY(hipGetLastError());
    DB_TRY(hipDeviceSynchronize());
    if (verbose) fprintf(stderr, "  device build: %zu triangles, %u refs%s, %u binary nodes, %u 4-wide nodes, depth %d\n", n, R, hairy ? ", hair-like" : "", nb, n4, out.max_depth);
    sw.lap("collapse + layout + gather");
    return NRAYS_OK;
}

} // namespace

int build_blas_device(const std::vector<DeviceMeshPart>& parts, const DeviceBuildOptions& opt, int32_t node_base, uint32_t prim_base, DeviceBlas& out, std::string& err) {
    Arena a1, a2, a3;
    int rc = build_impl(parts, opt, node_base, prim_base, out, err, a1, a2, a3);
    (void)hipDeviceSynchronize();
    a1.release(); a2.release(); a3.release();
    if (rc != NRAYS_OK) { free_device_blas(out); (void)hipGetLastError(); }
    return rc;
}

void free_device_blas(DeviceBlas& b) {
    if (b.nodes) (void)hipFree(b.nodes);
    if (b.tris) (void)hipFree(b.tris);
    if (b.uvs) (void)hipFree(b.uvs);
    b.nodes = nullptr; b.tris = nullptr; b.uvs = nullptr; b.num_nodes = 0; b.num_refs = 0; b.node_capacity = 0;
}

} // namespace nrays
