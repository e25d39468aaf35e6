// trace_device.h — gfx950 device code of the nrays trace loop: ray/shape intersectors, two-level
// BVH traversal with an LDS-resident stack, Phong shading with shadow rays, the reflection / refraction
// recursion as an in-register bounce loop (second children of double branches: wave ballot + prefix-sum
// compaction into an HBM queue).
//
// Reference functions replaced:
//   a1  render pixel/AA loop               src/scene.rs:67-95          -> generate_primary()
//   a4  ClosestRayTOICostFn + best_first   src/scene.rs:262-283        -> traverse<false>()
//   a5  SceneNode::cast + ncollide shapes  src/scene_node.rs:51-54     -> cast_*()
//   a6  ray/triangle                       ncollide (SURVEY B-8)       -> cast_triangle()
//   a7  TransparentShadowsRayTOICostFn     src/scene.rs:147-161,285-339-> traverse<true>()
//   a8-a10 Scene::trace / reflection / refraction  src/scene.rs:163-252-> shade_hit(), trace_chain()
//   a11 PhongMaterial                      src/phong_material.rs:39-151-> material_*()
//   a12 Light::sample                      src/light.rs:57-63          -> light loop in material_compute()
//   a13 Texture2d::sample                  src/texture2d.rs:203-256    -> tex_sample()
//   a14 Normal/UV materials                src/{normal,uv}_material.rs -> material_ambiant()
//
// Numerics: geometry in f64 (Scalar = f64, src/lib.rs:33), colour in f32, the same operation order
// as the reference / ncollide; the translation unit is compiled with -ffp-contract=off so that no
// a*b+c is fused (Rust never fuses), which keeps hit/miss decisions bit-identical to a strict IEEE
// evaluation.  BVH bounds are f32 rounded outward and tested with a relative slack, so box culling is
// a strict superset of the reference's and never changes a result.
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace nrays {

constexpr int kBlock = 256;     // threads per workgroup = 4 wave64 (8-wave workgroups measured slower: 100 vs 86 us on balls)
#ifndef NR_LDS_STACK
#define NR_LDS_STACK 24 // (32 until round 4: no measurable difference on any scene, profiles/r04_register_relief_ab.log; the LDS it frees holds Stack::park)
#endif
constexpr int kLdsStack = NR_LDS_STACK;   // traversal-stack entries kept in LDS per lane (then spills to HBM)
constexpr int32_t kSentinel = (int32_t)0x80000001; // marks "leave the BLAS" on the traversal stack
// Stack::park: dwords per lane a kFeatPark permutation keeps in LDS — what fits beside the traversal stacks of its resident workgroups
// (160 KiB per CU; three workgroups of 24 KiB of stack + 25 KiB: with 29 dwords a third workgroup no longer fitted and config 4 went from
// 10.8 to 14.9 ms): the three-wave alpha-shadow kernels 25 (several lights) / 24 (one light).  The four-wave opaque-mesh kernel gains
// nothing from 14 parked dwords (hairball 2.02 ms either way, 16 spp 19.0 -> 19.2): it has no kFeatPark permutation.
#ifndef NR_PARK_CAP
#define NR_PARK_CAP 25
#endif
constexpr int park_slots(int feat) {
    return !(feat & kFeatPark) || !(feat & kFeatAlphaShadow) ? 0
         : ((feat & kFeatMultiSample) ? (NR_PARK_CAP < 25 ? NR_PARK_CAP : 25) : (NR_PARK_CAP < 24 ? NR_PARK_CAP : 24));
}
static_assert(NR_PARK_CAP == 0 || NR_PARK_CAP >= 14, "shade_hit() parks 14 dwords unconditionally in a kFeatPark permutation: a tuning build needs NR_PARK_CAP = 0 (drop kFeatPark from the launch table) or >= 14");
constexpr int32_t kParked = (int32_t)0x80000002;   // a lane that yielded its node phase (traverse(): node_quorum); like kEmptyChild / kSentinel not a leaf ref
                                                   // that can occur (first = 2^28 - 1: scene_build.cpp refuses scenes that large)

#define NR_DEV __device__ __forceinline__
#ifndef NR_NODE_QUORUM_DEN
#define NR_NODE_QUORUM_DEN 3 // traverse(): node phases inside a hair-like mesh end below 1 / DEN of the query's lanes (0 = off)
#endif
#ifndef NR_ELIDE_DARK
#define NR_ELIDE_DARK 1 // light samples whose diffuse AND specular coefficients are exactly 0 (the light is behind the surface) are not traced (light_is_dark())
#endif
#ifndef NR_ELIDE_DARK_MATTE
#define NR_ELIDE_DARK_MATTE 1 // ... and for a material without a specular colour the sign of l.n alone decides (no_specular())
#endif
#ifndef NR_ELIDE_TRANSPARENT
#define NR_ELIDE_TRANSPARENT 1 // hits on fully transparent points skip their shadow rays and Phong (shade_hit())
#endif
#ifndef NR_SCALAR_NODES
#define NR_SCALAR_NODES 1 // wave-uniform node visits fetch the node with scalar loads (traverse())
#endif

// ---------------------------------------------------------------- vector algebra (f64) -------
struct d3 { double x, y, z; };
NR_DEV d3 D3(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
NR_DEV d3 operator+(d3 a, d3 b) { return D3(a.x + b.x, a.y + b.y, a.z + b.z); }
NR_DEV d3 operator-(d3 a, d3 b) { return D3(a.x - b.x, a.y - b.y, a.z - b.z); }
NR_DEV d3 operator*(d3 a, double s) { return D3(a.x * s, a.y * s, a.z * s); }
NR_DEV d3 operator/(d3 a, double s) { return D3(a.x / s, a.y / s, a.z / s); }
NR_DEV d3 operator-(d3 a) { return D3(-a.x, -a.y, -a.z); }
NR_DEV double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
NR_DEV d3 cross(d3 a, d3 b) { return D3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
NR_DEV double norm(d3 a) { return sqrt(dot(a, a)); }
NR_DEV d3 normalize(d3 a) { return a / norm(a); }
NR_DEV double comp(d3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

struct f3 { float x, y, z; };
NR_DEV f3 F3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
struct f4 { float x, y, z, w; };

// ---------------------------------------------------------------- RNG -------
NR_DEV unsigned long long rng_mix(unsigned long long z) {
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 27; z *= 0x94d049bb133111ebULL;
    z ^= z >> 31;
    return z;
}
NR_DEV unsigned long long rng_hash(unsigned long long key, unsigned long long salt) {
    return rng_mix((key ^ (salt * 0x9E3779B97F4A7C15ULL)) + 0xD1B54A32D192ED03ULL);
}
NR_DEV double rng_u01(unsigned long long key, unsigned long long dim) {
    return (double)(rng_hash(key, 0x1000ULL + dim) >> 11) * (1.0 / 9007199254740992.0);
}
constexpr unsigned long long kSaltPath = 2ULL, kSaltRefl = 0x100ULL, kSaltRefr = 0x101ULL, kSaltLight = 0x200ULL;

// ---------------------------------------------------------------- counters -------------------
#ifdef NR_PHASE_TIMING
// Tuning builds only (tools/kbench.py --libs ...): wave-level s_memtime attribution of the cycles of a
// wave to traversal phases; lane 0 accumulates, flush_counters sums over waves.
#define NR_TIC(var) unsigned long long var = __builtin_readcyclecounter()
#define NR_ITER(wv, ln) { cnt.ln++; if ((int)__lane_id() == __ffsll((long long)__ballot(1)) - 1) cnt.wv++; }
#define NR_INQ(field, n) { if ((int)__lane_id() == __ffsll((long long)__ballot(1)) - 1) cnt.field += (unsigned)(n); }
#define NR_INQ_COUNT(var) const int var = __popcll(__ballot(1))
#define NR_UNIFORM(curv) { const int f_ = __builtin_amdgcn_readfirstlane(curv); const unsigned long long a_ = __ballot(1); if (__ballot((curv) != f_) == 0ULL && (int)__lane_id() == __ffsll((long long)a_) - 1) cnt.wv_uni++; }
#define NR_TOC(cntfield, var) { unsigned long long now_ = __builtin_readcyclecounter(); cnt.cntfield += (unsigned)(now_ - var); var = now_; }
#else
#define NR_TIC(var)
#define NR_TOC(cntfield, var)
#define NR_ITER(wv, ln)
#define NR_INQ(field, n)
#define NR_INQ_COUNT(var)
#define NR_UNIFORM(curv)
#endif
struct Cnt {
    unsigned node, tri, prim, hit, tex;     // instrumented builds only
    unsigned fetch;                         // instrumented: 128-byte node records this lane fetched (a wave-uniform visit is counted by its leading lane only)
    unsigned shadow, refl, refr;            // ray classes, always counted
    unsigned max_depth;                     // deepest trace depth reached by this lane
    unsigned max_chain_nodes;               // instrumented: most AABB tests in one pixel's chain
    unsigned traced;                        // instrumented: primary rays that entered the trace loop
    unsigned elided;                        // shadow rays counted but not traced (shade_hit: hits that contribute nothing of their own)
#ifdef NR_PHASE_TIMING
    unsigned cyc_node, cyc_leaf, cyc_other, cyc_tri; // per-wave cycles (valid in lane 0); cyc_tri is part of cyc_leaf
    unsigned wv_node, ln_node, wv_tri, ln_tri;       // iterations of the node loop / triangle loop: per wave (counted by the leading active lane) and per lane
    unsigned wv_uni;                                 // node-loop wave iterations in which every active lane fetches the SAME node
    unsigned inq_node, inq_tri;                      // lanes still inside the query, summed over the wave iterations of the node / triangle loop
    unsigned cyc_x[8];                               // wave cycles outside the queries (DeviceCounters::dbg2)
    unsigned cyc_closest0, cyc_closestN, cyc_shadow; // wave cycles inside the closest-hit query of primary rays / of continuation rays / inside shadow queries
#endif
};

// ---------------------------------------------------------------- traversal stack ------------
// Slots 0..kLdsStack-1 live in LDS, laid out [slot][lane] so a wave's access is conflict-free (stride-1 dwords across
// lanes); deeper slots spill to a per-lane column in HBM.  Slot 0 holds a permanent kEmptyChild: popping an "empty"
// stack returns the end-of-traversal marker without a test.  `top` is the LDS ADDRESS of the next free slot, so the hot
// push is one ds_write + one add and the hot pop one add + one ds_read; whether ANY lane of the wave is near / beyond the
// LDS part is one ballot on that address (a scalar branch around the rare general code, no per-lane branch).
typedef __attribute__((address_space(3))) uint32_t lds_u32;    // explicitly LDS: ds_read / ds_write, never a generic flat access
typedef __attribute__((address_space(1))) uint32_t global_u32; // explicitly global memory
struct Stack {
    lds_u32* lds;      // &lds_stack[threadIdx.x] = slot 0 of this lane
    lds_u32* top;      // lds + n * kBlock, n = slots in use (the bottom marker included); beyond the LDS part only its value counts
    global_u32* spill; // &spill[global lane]   (may be null when the tree depth fits in LDS)
    uint32_t spill_stride;
    uint32_t lds0;     // uniform: LDS address of lds_stack[0]
    lds_u32* park;     // kFeatPark kernels: &lds_park[threadIdx.x], kParkSlots dwords per lane laid out [slot][lane] like the stack
    NR_DEV void park_d3(int slot, d3 v) const {
        volatile lds_u32* p = park + slot * kBlock;
        p[0] = (uint32_t)__double2loint(v.x); p[kBlock] = (uint32_t)__double2hiint(v.x); p[2 * kBlock] = (uint32_t)__double2loint(v.y);
        p[3 * kBlock] = (uint32_t)__double2hiint(v.y); p[4 * kBlock] = (uint32_t)__double2loint(v.z); p[5 * kBlock] = (uint32_t)__double2hiint(v.z);
    }
    NR_DEV void park_f(int slot, float v) const { *(volatile lds_u32*)(park + slot * kBlock) = __float_as_uint(v); }
    NR_DEV float unpark_f(int slot) const { return __uint_as_float(*(volatile lds_u32*)(park + slot * kBlock)); }
    NR_DEV void park_d(int slot, double v) const { volatile lds_u32* p = park + slot * kBlock; p[0] = (uint32_t)__double2loint(v); p[kBlock] = (uint32_t)__double2hiint(v); }
    NR_DEV double unpark_d(int slot) const { volatile lds_u32* p = park + slot * kBlock; const uint32_t a = p[0], b = p[kBlock]; return __hiloint2double((int)b, (int)a); }
    NR_DEV d3 unpark_d3(int slot) const {
        volatile lds_u32* p = park + slot * kBlock;
        const uint32_t a = p[0], b = p[kBlock], c = p[2 * kBlock], d = p[3 * kBlock], e = p[4 * kBlock], f = p[5 * kBlock];
        return D3(__hiloint2double((int)b, (int)a), __hiloint2double((int)d, (int)c), __hiloint2double((int)f, (int)e));
    }
    NR_DEV static uint32_t addr(const lds_u32* p) { return (uint32_t)(uintptr_t)p; }
    NR_DEV void init() { lds[0] = (uint32_t)kEmptyChild; top = lds + kBlock; } // once per kernel: nothing ever overwrites slot 0
    NR_DEV void reset() { top = lds + kBlock; }
    NR_DEV int slots() const { return (int)(top - lds) / kBlock; }
    // true iff every active lane of the wave can take `n` more slots inside the LDS part
    NR_DEV bool wave_has_room(int n) const { return __ballot(addr(top) >= lds0 + (uint32_t)(kLdsStack - n + 1) * kBlock * 4u) == 0; }
    NR_DEV void push(int32_t v) { // general form
        const int n = slots();
        if (n < kLdsStack) *top = (uint32_t)v;
        else spill[(size_t)(n - kLdsStack) * spill_stride] = (uint32_t)v;
        top += kBlock;
    }
    NR_DEV int32_t pop() {
        top -= kBlock;
        if (__builtin_expect(__ballot(addr(top) >= lds0 + (uint32_t)kLdsStack * kBlock * 4u) != 0, 0)) { // some lane is in its HBM column
            const int n = slots();
            // two explicit accesses: written as one conditional expression the compiler selects between the POINTERS and
            // issues a generic flat_load
            int32_t v = (int32_t)*(volatile lds_u32*)&lds[(n < kLdsStack ? n : kLdsStack - 1) * kBlock];
            if (n >= kLdsStack) v = (int32_t)spill[(size_t)(n - kLdsStack) * spill_stride];
            asm volatile("" : "+v"(v)); // the value is complete HERE: the callers' join points then carry no pending vector load (their wait would also cover the node prefetches)
            return v;
        }
        return (int32_t)*top;
    }
};

// ---------------------------------------------------------------- intersection record --------
struct Isect {
    double toi;
    d3 n;
    double u, v;
    bool has_uv;
    bool hit; // only meaningful on the value returned by cast_analytic
};

struct Xform { // Isometry3: R (row-major) and translation
    double r[9];
    d3 t;
};
NR_DEV d3 rot(const Xform& m, d3 v) {
    return D3(m.r[0] * v.x + m.r[1] * v.y + m.r[2] * v.z, m.r[3] * v.x + m.r[4] * v.y + m.r[5] * v.z, m.r[6] * v.x + m.r[7] * v.y + m.r[8] * v.z);
}
NR_DEV d3 inv_rot(const Xform& m, d3 v) {
    return D3(m.r[0] * v.x + m.r[3] * v.y + m.r[6] * v.z, m.r[1] * v.x + m.r[4] * v.y + m.r[7] * v.z, m.r[2] * v.x + m.r[5] * v.y + m.r[8] * v.z);
}

constexpr double kPi = 3.14159265358979323846;
constexpr double kDblMax = 1.7976931348623157e308;

// ncollide Ball (SURVEY B-4): centre = translation, rotation ignored.
// `record`: 0 = the caller only needs hit / toi (opaque shadow rays), 1 = and the normal (materials that never read the values of u, v:
// kInstNoUvValues), 2 = everything.  What is left out — a square root and three divisions, atan2 and asin — is never looked at.
NR_DEV bool cast_ball(double radius, d3 center, d3 o, d3 d, bool solid, int record, Isect& out) {
    d3 dc = o - center;
    double a = dot(d, d), b = dot(dc, d), c = dot(dc, dc) - radius * radius;
    if (c > 0.0 && b > 0.0) return false;
    double delta = b * b - a * c;
    if (delta < 0.0) return false;
    double sq = sqrt(delta);
    double t = (-b - sq) / a;
    bool inside = false;
    if (t <= 0.0) { inside = true; t = solid ? 0.0 : (-b + sq) / a; }
    out.toi = t;
    if (!record) return true;
    d3 pos = (o + d * t) - center;
    d3 n = normalize(pos);
    out.has_uv = true;
    out.n = inside ? -n : n;
    if (record < 2) return true;
    out.u = 0.5 + atan2(n.z, n.x) / (kPi * 2.0);
    out.v = 0.5 - asin(n.y) / kPi;
    return true;
}

// ncollide Cuboid via ray_aabb on [-he, he] (SURVEY B-5).
NR_DEV bool cast_cuboid(d3 he, const Xform& m, d3 o, d3 d, bool solid, Isect& out) {
    d3 lo = inv_rot(m, o - m.t), ld = inv_rot(m, d);
    double tmax = kDblMax, tmin = -kDblMax;
    int near_side = 0, far_side = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double dd = comp(ld, i), oo = comp(lo, i), mx = comp(he, i), mn = -mx;
        if (dd == 0.0) {
            if (oo < mn || oo > mx) return false;
        } else {
            double denom = 1.0 / dd;
            double tn = (mn - oo) * denom, tf = (mx - oo) * denom;
            bool flip = false;
            if (tn > tf) { double s = tn; tn = tf; tf = s; flip = true; }
            if (tn > tmin) { tmin = tn; near_side = flip ? -(i + 1) : (i + 1); }
            if (tf < tmax) { tmax = tf; far_side = flip ? (i + 1) : -(i + 1); }
            if (tmax < 0.0 || tmin > tmax) return false;
        }
    }
    d3 n = D3(0.0, 0.0, 0.0);
    double t; int side;
    if (tmin < 0.0) {
        side = far_side;
        if (solid) t = 0.0;
        else {
            t = tmax;
            double s = far_side < 0 ? -1.0 : 1.0;
            int ax = (far_side < 0 ? -far_side : far_side) - 1;
            if (far_side != 0) { if (ax == 0) n.x = s; else if (ax == 1) n.y = s; else n.z = s; }
        }
    } else {
        t = tmin; side = near_side;
        double s = near_side < 0 ? 1.0 : -1.0;
        int ax = (near_side < 0 ? -near_side : near_side) - 1;
        if (near_side != 0) { if (ax == 0) n.x = s; else if (ax == 1) n.y = s; else n.z = s; }
    }
    d3 pt = lo + ld * t;
    d3 dpt = pt - (-he);
    d3 scale = he - (-he);
    int id = side < 0 ? -side : side;
    out.toi = t; out.n = rot(m, n); out.has_uv = true;
    if (id == 1) { out.u = dpt.y / scale.y; out.v = dpt.z / scale.z; }
    else if (id == 2) { out.u = dpt.z / scale.z; out.v = dpt.x / scale.x; }
    else { out.u = dpt.x / scale.x; out.v = dpt.y / scale.y; }
    return true;
}

// ncollide Plane (SURVEY B-6).
NR_DEV bool cast_plane(d3 pn, const Xform& m, d3 o, d3 d, bool solid, Isect& out) {
    d3 lo = inv_rot(m, o - m.t), ld = inv_rot(m, d);
    double dot_normal_dpos = dot(pn, -lo);
    out.has_uv = false; out.u = 0.0; out.v = 0.0;
    if (solid && dot_normal_dpos > 0.0) { out.toi = 0.0; out.n = D3(0.0, 0.0, 0.0); return true; }
    double denom = dot(pn, ld);
    if (denom == 0.0) return false;
    double t = dot_normal_dpos / denom;
    if (t >= 0.0) {
        d3 n = dot_normal_dpos > 0.0 ? -pn : pn;
        out.toi = t; out.n = rot(m, n);
        return true;
    }
    return false;
}

// Shared tail of the closed-form convex casts (cylinder / cone / capsule).
NR_DEV bool convex_interval_hit(double t0, double t1, d3 n0, d3 n1, d3 ld, const Xform& m, bool solid, Isect& out) {
    if (!(t0 <= t1) || t1 < 0.0) return false;
    out.has_uv = false; out.u = 0.0; out.v = 0.0;
    if (t0 > 0.0) { out.toi = t0; out.n = rot(m, n0); return true; }
    if (solid) { out.toi = 0.0; out.n = rot(m, -ld); return true; }
    out.toi = t1; out.n = rot(m, n1);
    return true;
}

NR_DEV bool cast_cylinder(double hh, double r, const Xform& m, d3 o, d3 d, bool solid, Isect& out) {
    d3 lo = inv_rot(m, o - m.t), ld = inv_rot(m, d);
    double t0 = -kDblMax, t1 = kDblMax;
    bool enter_side = true, exit_side = true;
    double A = ld.x * ld.x + ld.z * ld.z;
    double B = lo.x * ld.x + lo.z * ld.z;
    double C = lo.x * lo.x + lo.z * lo.z - r * r;
    if (A == 0.0) { if (C > 0.0) return false; }
    else {
        double disc = B * B - A * C;
        if (disc < 0.0) return false;
        double sq = sqrt(disc);
        t0 = (-B - sq) / A; t1 = (-B + sq) / A;
    }
    if (ld.y == 0.0) { if (lo.y < -hh || lo.y > hh) return false; }
    else {
        double ta = (-hh - lo.y) / ld.y, tb = (hh - lo.y) / ld.y;
        if (ta > tb) { double s = ta; ta = tb; tb = s; }
        if (ta > t0) { t0 = ta; enter_side = false; }
        if (tb < t1) { t1 = tb; exit_side = false; }
    }
    if (!(t0 <= t1) || t1 < 0.0) return false;
    d3 n0, n1;
    if (enter_side) { d3 p = lo + ld * t0; double s = sqrt(p.x * p.x + p.z * p.z); n0 = D3(p.x / s, 0.0, p.z / s); }
    else n0 = D3(0.0, ld.y > 0.0 ? -1.0 : 1.0, 0.0);
    if (exit_side) { d3 p = lo + ld * t1; double s = sqrt(p.x * p.x + p.z * p.z); n1 = D3(p.x / s, 0.0, p.z / s); }
    else n1 = D3(0.0, ld.y > 0.0 ? 1.0 : -1.0, 0.0);
    return convex_interval_hit(t0, t1, n0, n1, ld, m, solid, out);
}

NR_DEV d3 cone_side_normal(d3 p, double hh, double k2) {
    d3 g = D3(p.x, k2 * (hh - p.y), p.z);
    double s = norm(g);
    if (s == 0.0) return D3(0.0, 1.0, 0.0);
    return g / s;
}
NR_DEV bool cast_cone(double hh, double r, const Xform& m, d3 o, d3 d, bool solid, Isect& out) {
    d3 lo = inv_rot(m, o - m.t), ld = inv_rot(m, d);
    double k = r / (2.0 * hh), k2 = k * k;
    double ow = hh - lo.y, dw = -ld.y;
    double s0 = -kDblMax, s1 = kDblMax;
    int s0_kind = 0, s1_kind = 0;
    if (dw == 0.0) { if (ow < 0.0 || ow > 2.0 * hh) return false; }
    else {
        double ta = (0.0 - ow) / dw, tb = (2.0 * hh - ow) / dw;
        if (ta <= tb) { s0 = ta; s0_kind = 1; s1 = tb; s1_kind = 2; }
        else { s0 = tb; s0_kind = 2; s1 = ta; s1_kind = 1; }
    }
    double A = ld.x * ld.x + ld.z * ld.z - k2 * dw * dw;
    double B = lo.x * ld.x + lo.z * ld.z - k2 * ow * dw;
    double C = lo.x * lo.x + lo.z * lo.z - k2 * ow * ow;
    double t0 = s0, t1 = s1;
    int k0 = s0_kind, k1 = s1_kind;
    if (A > 0.0) {
        double disc = B * B - A * C;
        if (disc < 0.0) return false;
        double sq = sqrt(disc);
        double ra = (-B - sq) / A, rb = (-B + sq) / A;
        if (ra > t0) { t0 = ra; k0 = 0; }
        if (rb < t1) { t1 = rb; k1 = 0; }
    } else if (A < 0.0) {
        double disc = B * B - A * C;
        if (disc > 0.0) {
            double sq = sqrt(disc);
            double lo_r = (-B + sq) / A, hi_r = (-B - sq) / A;
            double a1 = hi_r > s0 ? hi_r : s0;
            double b0 = lo_r < s1 ? lo_r : s1;
            if (a1 <= s1) { if (hi_r > s0) { t0 = hi_r; k0 = 0; } }
            else if (s0 <= b0) { if (lo_r < s1) { t1 = lo_r; k1 = 0; } }
            else return false;
        }
    } else {
        if (B == 0.0) { if (C > 0.0) return false; }
        else {
            double ts = -C / (2.0 * B);
            if (B > 0.0) { if (ts < t1) { t1 = ts; k1 = 0; } }
            else { if (ts > t0) { t0 = ts; k0 = 0; } }
        }
    }
    if (!(t0 <= t1) || t1 < 0.0) return false;
    d3 n0, n1;
    if (k0 == 0) n0 = cone_side_normal(lo + ld * t0, hh, k2); else n0 = D3(0.0, k0 == 1 ? 1.0 : -1.0, 0.0);
    if (k1 == 0) n1 = cone_side_normal(lo + ld * t1, hh, k2); else n1 = D3(0.0, k1 == 1 ? 1.0 : -1.0, 0.0);
    return convex_interval_hit(t0, t1, n0, n1, ld, m, solid, out);
}

NR_DEV bool cast_capsule(double hh, double r, const Xform& m, d3 o, d3 d, bool solid, Isect& out) {
    d3 lo = inv_rot(m, o - m.t), ld = inv_rot(m, d);
    double t0 = kDblMax, t1 = -kDblMax;
    d3 n0 = D3(0.0, 0.0, 0.0), n1 = D3(0.0, 0.0, 0.0);
    {
        double a0 = -kDblMax, a1 = kDblMax; bool ok = true;
        double A = ld.x * ld.x + ld.z * ld.z;
        double B = lo.x * ld.x + lo.z * ld.z;
        double C = lo.x * lo.x + lo.z * lo.z - r * r;
        if (A == 0.0) { if (C > 0.0) ok = false; }
        else {
            double disc = B * B - A * C;
            if (disc < 0.0) ok = false;
            else { double sq = sqrt(disc); a0 = (-B - sq) / A; a1 = (-B + sq) / A; }
        }
        bool e_side = true, x_side = true;
        if (ok) {
            if (ld.y == 0.0) { if (lo.y < -hh || lo.y > hh) ok = false; }
            else {
                double ta = (-hh - lo.y) / ld.y, tb = (hh - lo.y) / ld.y;
                if (ta > tb) { double s = ta; ta = tb; tb = s; }
                if (ta > a0) { a0 = ta; e_side = false; }
                if (tb < a1) { a1 = tb; x_side = false; }
            }
        }
        if (ok && a0 <= a1) {
            t0 = a0; t1 = a1;
            if (e_side) { d3 p = lo + ld * a0; n0 = D3(p.x / r, 0.0, p.z / r); }
            if (x_side) { d3 p = lo + ld * a1; n1 = D3(p.x / r, 0.0, p.z / r); }
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        d3 c = D3(0.0, s == 0 ? -hh : hh, 0.0);
        d3 dc = lo - c;
        double a = dot(ld, ld), b = dot(dc, ld), cc = dot(dc, dc) - r * r;
        double delta = b * b - a * cc;
        if (delta < 0.0) continue;
        double sq = sqrt(delta);
        double b0 = (-b - sq) / a, b1 = (-b + sq) / a;
        if (b0 < t0) { t0 = b0; n0 = ((lo + ld * b0) - c) / r; }
        if (b1 > t1) { t1 = b1; n1 = ((lo + ld * b1) - c) / r; }
    }
    return convex_interval_hit(t0, t1, n0, n1, ld, m, solid, out);
}

// ncollide triangle_ray_intersection (SURVEY B-8).  `full` also produces the normal and barycentrics.
NR_DEV bool cast_triangle(d3 a, d3 b, d3 c, d3 o, d3 d, double& toi, d3* normal, double* bary) {
    d3 ab = b - a, ac = c - a;
    d3 n = cross(ab, ac);
    double dn = dot(n, d);
    if (dn == 0.0) return false;
    d3 ap = o - a;
    double t = dot(ap, n);
    if ((t < 0.0 && dn < 0.0) || (t > 0.0 && dn > 0.0)) return false;
    double dabs = fabs(dn);
    d3 e = -cross(d, ap);
    double v, w, invd;
    if (t < 0.0) {
        v = -dot(ac, e);
        if (v < 0.0 || v > dabs) return false;
        w = dot(ab, e);
        if (w < 0.0 || v + w > dabs) return false;
        invd = 1.0 / dabs;
        toi = -t * invd;
        if (normal) *normal = -normalize(n);
    } else {
        v = dot(ac, e);
        if (v < 0.0 || v > dabs) return false;
        w = -dot(ab, e);
        if (w < 0.0 || v + w > dabs) return false;
        invd = 1.0 / dabs;
        toi = t * invd;
        if (normal) *normal = normalize(n);
    }
    if (bary) { v = v * invd; w = w * invd; bary[0] = -v - w + 1.0; bary[1] = v; bary[2] = w; }
    return true;
}

NR_DEV void load_xform(const Instance& in, Xform& m) {
#pragma unroll
    for (int k = 0; k < 9; ++k) m.r[k] = in.rot[k];
    m.t = D3(in.trans[0], in.trans[1], in.trans[2]);
}

// SceneNode::cast for the analytic shapes (scene_node.rs:51-54).
// One out-of-line copy shared by the closest-hit and shadow paths.  The record is RETURNED (56 bytes: the
// AMDGPU calling convention hands aggregates of up to 16 dwords back in VGPRs); an `Isect&` out-parameter
// would live in scratch memory and cost a store + load round trip per call.
// The instance comes as an explicitly GLOBAL pointer: across the call boundary a plain reference is a generic pointer
// and every field read a flat_load.
typedef const __attribute__((address_space(1))) Instance* GInstance; // instance record in global memory
typedef const __attribute__((address_space(3))) Instance* LInstance; // instance record in LDS (kFeatLdsScene kernels)
template <class P>
__device__ __forceinline__ Isect cast_analytic_at(P inp, d3 o, d3 d, bool record) {
    const auto& in = *inp;
    bool solid = (in.flags & kInstSolid) != 0;
    Xform m;
#pragma unroll
    for (int k = 0; k < 9; ++k) m.r[k] = in.rot[k];
    m.t = D3(in.trans[0], in.trans[1], in.trans[2]);
    Isect out;
    out.toi = 0.0; out.n = D3(0.0, 0.0, 0.0); out.u = 0.0; out.v = 0.0; out.has_uv = false; out.hit = false;
    switch (in.kind) {
    case NRAYS_SHAPE_BALL: out.hit = cast_ball(in.params[0], m.t, o, d, solid, !record ? 0 : ((in.flags & kInstNoUvValues) ? 1 : 2), out); break;
    case NRAYS_SHAPE_CUBOID: out.hit = cast_cuboid(D3(in.params[0], in.params[1], in.params[2]), m, o, d, solid, out); break;
    case NRAYS_SHAPE_CYLINDER: out.hit = cast_cylinder(in.params[0], in.params[1], m, o, d, solid, out); break;
    case NRAYS_SHAPE_CAPSULE: out.hit = cast_capsule(in.params[0], in.params[1], m, o, d, solid, out); break;
    case NRAYS_SHAPE_CONE: out.hit = cast_cone(in.params[0], in.params[1], m, o, d, solid, out); break;
    case NRAYS_SHAPE_PLANE: out.hit = cast_plane(D3(in.params[0], in.params[1], in.params[2]), m, o, d, solid, out); break;
    default: break;
    }
    return out;
}
__device__ __noinline__ Isect cast_analytic(GInstance inp, d3 o, d3 d, bool record) { return cast_analytic_at(inp, o, d, record); }
__device__ __noinline__ Isect cast_analytic_lds(LInstance inp, d3 o, d3 d, bool record) { return cast_analytic_at(inp, o, d, record); }
// The instance records of a kFeatLdsScene kernel live in the workgroup's LDS copy of the scene (k_primary).
template <int FEAT>
NR_DEV Isect cast_instance(const Instance& in, d3 o, d3 d, bool record = true) {
    if (FEAT & kFeatLdsScene) return cast_analytic_lds((LInstance)&in, o, d, record);
    return cast_analytic((GInstance)&in, o, d, record);
}

// ---------------------------------------------------------------- textures & materials -------
NR_DEV f4 tex_at(const ShadeTex& t, uint32_t x, uint32_t y) {
    size_t i = (size_t)y * t.width + x;
    f4 r;
    // the texel pointers live in device records: tell the compiler that they point to global memory (a generic
    // pointer loaded from memory makes every fetch a flat_load)
    if ((t.mode & 0xffu) == NRAYS_TEXEL_RGBA8) {
        const uint32_t p = ((const __attribute__((address_space(1))) uint32_t*)t.texels)[i]; // r | g << 8 | b << 16 | a << 24
        // `u8 as f32 / 255.0` (texture2d.rs:111-162) without the ten-instruction IEEE f32 division: for every byte value the
        // f64 product x * (1 / 255), rounded to f32, IS the correctly rounded f32 quotient (all 256 cases are compared in
        // tests/test_numerics_tables.py; the plain f32 product x * fl(1 / 255) is wrong for 126 of them)
        const double k = 1.0 / 255.0;
        r.x = (float)((double)(p & 0xffu) * k); r.y = (float)((double)((p >> 8) & 0xffu) * k);
        r.z = (float)((double)((p >> 16) & 0xffu) * k); r.w = (float)((double)(p >> 24) * k);
    } else {
        const __attribute__((address_space(1))) float* p = (const __attribute__((address_space(1))) float*)t.texels + 4 * i;
        r.x = p[0]; r.y = p[1]; r.z = p[2]; r.w = p[3];
    }
    return r;
}
// Texture2d::sample (texture2d.rs:207-256), taps clamped to the last row/column.
template <bool STATS>
NR_DEV f4 tex_sample(const ShadeTex& t, double u, double v, Cnt& cnt) {
    if (STATS) cnt.tex++;
    float ux = (float)u, uy = (float)v;
    if (((t.mode >> 16) & 0xffu) == NRAYS_OVERFLOW_CLAMP) {
        ux = ux < 0.0f ? 0.0f : (ux > 1.0f ? 1.0f : ux);
        uy = uy < 0.0f ? 0.0f : (uy > 1.0f ? 1.0f : uy);
    } else {
        // `% 1.0` (texture2d.rs:215-221) is fmodf(x, 1): the fractional part with the sign of x — x - trunc(x) is exact in f32, so
        // three instructions give the library routine's value bit for bit (checked over 2 M values incl. -0, 2^23 and denormals)
        ux = copysignf(ux - truncf(ux), ux); uy = copysignf(uy - truncf(uy), uy);
        if (ux < 0.0f) ux = 1.0f + ux;
        if (uy < 0.0f) uy = 1.0f + uy;
    }
    ux = ux * (float)(t.width - 1);
    uy = uy * (float)(t.height - 1);
    uint32_t wm = t.width - 1, hm = t.height - 1;
    if (((t.mode >> 8) & 0xffu) == NRAYS_INTERP_NEAREST) {
        uint32_t x = (uint32_t)roundf(ux), y = (uint32_t)roundf(uy);
        if (x > wm) x = wm;
        if (y > hm) y = hm;
        return tex_at(t, x, y);
    }
    uint32_t lx = (uint32_t)floorf(ux), ly = (uint32_t)floorf(uy);
    if (lx > wm) lx = wm;
    if (ly > hm) ly = hm;
    uint32_t hx = lx + 1, hy = ly + 1;
    float sx = ux - (float)lx, sy = uy - (float)ly;
    if (hx > wm) hx = wm;
    if (hy > hm) hy = hm;
    // the four taps under ONE test of the texel format: inside tex_at each tap is a branch with its own load and its own wait for it —
    // four memory round trips in a row where one is needed (the deep reflection tiles of the balls frame spent 40 % of their time in
    // this function; issuing the loads even earlier, before the shadow query of the hit, was measured and buys nothing more)
    f4 ul, ur, dr, dl;
    const size_t i_ul = (size_t)hy * t.width + lx, i_ur = (size_t)hy * t.width + hx, i_dr = (size_t)ly * t.width + hx, i_dl = (size_t)ly * t.width + lx;
    if ((t.mode & 0xffu) == NRAYS_TEXEL_RGBA8) {
        const __attribute__((address_space(1))) uint32_t* tp = (const __attribute__((address_space(1))) uint32_t*)t.texels;
        const uint32_t p0 = tp[i_ul], p1 = tp[i_ur], p2 = tp[i_dr], p3 = tp[i_dl];
        const double k = 1.0 / 255.0; // tex_at: the f64 product rounded to f32 is the correctly rounded `u8 as f32 / 255.0`
        auto unpack = [&](uint32_t p) { f4 r; r.x = (float)((double)(p & 0xffu) * k); r.y = (float)((double)((p >> 8) & 0xffu) * k);
                                         r.z = (float)((double)((p >> 16) & 0xffu) * k); r.w = (float)((double)(p >> 24) * k); return r; };
        ul = unpack(p0); ur = unpack(p1); dr = unpack(p2); dl = unpack(p3);
    } else {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const __attribute__((address_space(1))) f4v* tp = (const __attribute__((address_space(1))) f4v*)t.texels;
        const f4v q0 = tp[i_ul], q1 = tp[i_ur], q2 = tp[i_dr], q3 = tp[i_dl];
        ul.x = q0.x; ul.y = q0.y; ul.z = q0.z; ul.w = q0.w; ur.x = q1.x; ur.y = q1.y; ur.z = q1.z; ur.w = q1.w;
        dr.x = q2.x; dr.y = q2.y; dr.z = q2.z; dr.w = q2.w; dl.x = q3.x; dl.y = q3.y; dl.z = q3.z; dl.w = q3.w;
    }
    f4 ui, di, r;
    ui.x = ul.x * (1.0f - sx) + ur.x * sx; ui.y = ul.y * (1.0f - sx) + ur.y * sx; ui.z = ul.z * (1.0f - sx) + ur.z * sx; ui.w = ul.w * (1.0f - sx) + ur.w * sx;
    di.x = dl.x * (1.0f - sx) + dr.x * sx; di.y = dl.y * (1.0f - sx) + dr.y * sx; di.z = dl.z * (1.0f - sx) + dr.z * sx; di.w = dl.w * (1.0f - sx) + dr.w * sx;
    r.x = ui.x * sy + di.x * (1.0f - sy); r.y = ui.y * sy + di.y * (1.0f - sy); r.z = ui.z * sy + di.z * (1.0f - sy); r.w = ui.w * sy + di.w * (1.0f - sy);
    return r;
}

// Material::ambiant (phong_material.rs:39-70, normal_material.rs:9-14, uv_material.rs:10-20).
template <bool STATS>
NR_DEV f4 material_ambiant(const ShadeRec& m, const Isect& in, Cnt& cnt) {
    f4 r;
    const uint32_t kind = (m.flags >> 8) & 0xffu;
    if (kind == NRAYS_MAT_NORMAL) {
        r.x = (1.0f + (float)in.n.x) / 2.0f; r.y = (1.0f + (float)in.n.y) / 2.0f; r.z = (1.0f + (float)in.n.z) / 2.0f; r.w = 1.0f;
        return r;
    }
    if (kind == NRAYS_MAT_UV) {
        if (in.has_uv) { r.x = (float)in.u; r.y = (float)in.v; r.z = 0.0f; r.w = 1.0f; }
        else { r.x = r.y = r.z = r.w = 0.0f; }
        return r;
    }
    if (in.has_uv) {
        f4 tc; tc.x = tc.y = tc.z = tc.w = 1.0f;
        if (m.tex.texels) { tc = tex_sample<STATS>(m.tex, in.u, in.v, cnt); tc.w = 1.0f; }
        if (m.alpha_tex.texels) tc.w = tex_sample<STATS>(m.alpha_tex, in.u, in.v, cnt).w;
        r.x = m.ka[0] * tc.x; r.y = m.ka[1] * tc.y; r.z = m.ka[2] * tc.z; r.w = 1.0f * tc.w;
    } else { r.x = m.ka[0]; r.y = m.ka[1]; r.z = m.ka[2]; r.w = 1.0f; }
    return r;
}

// ---------------------------------------------------------------- BVH traversal --------------
struct Hit {
    double t;
    uint32_t inst; // index into the TLAS's instance array
    uint32_t prim; // global triangle slot (TRIMESH) else 0
};

// f32 view of a ray for box culling.  The BVH bounds are exact f32 supersets of the f64 geometry; the
// slab test runs in f32 with explicit error margins so that it is a SUPERSET of the f64 slab test
// (ncollide ray_aabb, called at src/scene.rs:276) and can therefore never change a result:
//   o32 = fl(o)              |o - o32| <= |o| 2^-24
//   inv32 = rcp(fl(d))       relative error <= 2^-24 (rounding of d) + 2^-23 (v_rcp_f32, 1 ulp)
//   p = fl(o32 * inv32)      one more rounding of |o inv|
//   e = |p| 2^-21            the absolute margin of the axis
//   n_near = fl(-p - e),  n_far = fl(-p + e)      (a third rounding of magnitude 2^-24 |p|)
//   t_near = fma(b_near, inv32, n_near),  t_far = fma(b_far, inv32, n_far)
// b inv32 and o32 inv32 share inv32, so its error stays RELATIVE to t (3 * 2^-24, + 2^-24 for the fma's own rounding ->
// relative slack on the final compare), while the roundings of o32, of p and of n_near / n_far are absolute, at most
// 3 * 2^-24 |o inv| together, well inside e = 8 * 2^-24 |p|.  The margin is part of the fma's addend, so a slab costs one
// instruction per plane.  b_near is the plane the ray enters through — the lower bound for a positive direction
// component, the upper one for a negative one — and the lane FETCHES it from there (BvhNode keeps the two planes of an
// axis one address bit apart and RayF::bits holds those bits), so no min / max separates near from far.
// A direction component that is zero in f32 (|d| below ~1e-30) is taken out of the slab arithmetic: inverse 0 and the
// addends -inf / +inf make that axis' interval (-inf, +inf) (the bounds are finite, so no NaN can arise), and
// zero_axis_cull() applies the reference's rule for a zero component — the ray misses iff its origin lies outside
// [mn, mx] — with the rounding of the origin as margin.  (Round 1 used the finite inverse 1e30 and let the margin
// decide; but the margin is ZERO for an origin coordinate of exactly 0, and a box with a face exactly in that
// coordinate plane — mx == o — then gets the interval [-huge, 0] and is culled, while the reference's test for a zero
// component, `o < mn || o > mx`, accepts the boundary.  Found by the full-size 3840x2160 comparison with the oracle: two
// pixels of the centre column, whose rays lie exactly in the stand-in's symmetry plane z = 0.)
struct RayF {
    float ix, iy, iz;       // inv32 per axis
    float nnx, nny, nnz;    // n_near
    float nfx, nfy, nfz;    // n_far
    uint32_t bits;          // bits 4 / 5 / 6: direction component x / y / z negative (= the address bit that selects the
                            // upper plane of that axis inside a BvhNode); bits 0..2: component a is zero in f32
                            // (zero_axis_cull() handles that axis)
};
NR_DEV float inv_f32(double d) { // 0 = "this axis does not constrain the ray" (see above)
    float x = (float)d;
    float r = __builtin_amdgcn_rcpf(x);
    return (fabsf(x) > 1e-30f && fabsf(r) < 1e30f) ? r : 0.0f;
}
NR_DEV RayF make_rayf(d3 o, d3 d) {
    RayF r;
    const float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z;
    r.ix = inv_f32(d.x); r.iy = inv_f32(d.y); r.iz = inv_f32(d.z);
    const float px = ox * r.ix, py = oy * r.iy, pz = oz * r.iz;
    const float kInf = __builtin_inff(), k21 = 4.76837158203125e-07f; // 2^-21
    const float ex = fabsf(px) * k21, ey = fabsf(py) * k21, ez = fabsf(pz) * k21;
    r.nnx = r.ix != 0.0f ? -px - ex : -kInf; r.nfx = r.ix != 0.0f ? -px + ex : kInf;
    r.nny = r.iy != 0.0f ? -py - ey : -kInf; r.nfy = r.iy != 0.0f ? -py + ey : kInf;
    r.nnz = r.iz != 0.0f ? -pz - ez : -kInf; r.nfz = r.iz != 0.0f ? -pz + ez : kInf;
    r.bits = (r.ix < 0.0f ? 16u : 0u) | (r.iy < 0.0f ? 32u : 0u) | (r.iz < 0.0f ? 64u : 0u) |
             (r.ix == 0.0f ? 1u : 0u) | (r.iy == 0.0f ? 2u : 0u) | (r.iz == 0.0f ? 4u : 0u);
    return r;
}
// Upper f32 bound of the current best distance (ties with it must still be visited).
NR_DEV float best_f32(double bt) { return (float)bt * 1.0000004f + 1e-37f; }

// The six planes of the four children of node `node`, entry / exit plane per axis as the ray's signs select them.
// The node array is addressed as uniform base + 32-bit byte offset (scene_build.cpp refuses scenes beyond 2^25 nodes).
struct NodePlanes { float4 xn, xf, yn, yf, zn, zf; };
NR_DEV NodePlanes load_planes(const BvhNode* nodes, int32_t node, const RayF& r) {
    const char* base = (const char*)nodes;
    const uint32_t at = (uint32_t)node << 7;
    const uint32_t ax = at | (r.bits & 16u), ay = at | (r.bits & 32u), az = at | (r.bits & 64u); // v_and_or_b32
    NodePlanes p;
    p.xn = *(const float4*)(base + ax); p.xf = *(const float4*)(base + (ax ^ 16u));
    p.yn = *(const float4*)(base + ay + 64); p.yf = *(const float4*)(base + (ay ^ 32u) + 64);
    p.zn = *(const float4*)(base + az + 48); p.zf = *(const float4*)(base + (az ^ 64u) + 48);
    return p;
}
NR_DEV int4 load_children(const BvhNode* nodes, int32_t node) { return *(const int4*)((const char*)nodes + (((uint32_t)node << 7) | 32u)); }

// The same fetch for a visit in which EVERY active lane of the wave sits on the same node with the same direction signs (`ukey` =
// node << 7 | RayF::bits, wave-uniform): seven scalar loads through the scalar data cache into SGPRs — no vector-memory request, no
// bytes through the vector L1's return path (the shared ceiling of the node loops: a vector fetch moves 112 bytes PER LANE), no VGPRs;
// the slab arithmetic then reads the planes as scalar operands.  8x8-pixel packets of primary rays, the shadow rays of a tile towards
// one point light and the samples of one pixel spend most of their visits like this.
typedef float fx4 __attribute__((ext_vector_type(4)));
typedef int ix4 __attribute__((ext_vector_type(4)));
NR_DEV float4 F4(fx4 v) { float4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r; }
NR_DEV NodePlanes load_planes_uniform(const BvhNode* nodes, uint32_t ukey) {
    const __attribute__((address_space(4))) char* base = (const __attribute__((address_space(4))) char*)(uintptr_t)nodes;
    const uint32_t at = ukey & ~127u;
    const uint32_t ax = at | (ukey & 16u), ay = at | (ukey & 32u), az = at | (ukey & 64u);
    NodePlanes p;
    p.xn = F4(*(const __attribute__((address_space(4))) fx4*)(base + ax)); p.xf = F4(*(const __attribute__((address_space(4))) fx4*)(base + (ax ^ 16u)));
    p.yn = F4(*(const __attribute__((address_space(4))) fx4*)(base + ay + 64)); p.yf = F4(*(const __attribute__((address_space(4))) fx4*)(base + (ay ^ 32u) + 64));
    p.zn = F4(*(const __attribute__((address_space(4))) fx4*)(base + az + 48)); p.zf = F4(*(const __attribute__((address_space(4))) fx4*)(base + (az ^ 64u) + 48));
    return p;
}
NR_DEV int4 load_children_uniform(const BvhNode* nodes, uint32_t ukey) {
    const __attribute__((address_space(4))) char* base = (const __attribute__((address_space(4))) char*)(uintptr_t)nodes;
    const ix4 c = *(const __attribute__((address_space(4))) ix4*)(base + ((ukey & ~127u) | 32u));
    int4 r; r.x = c.x; r.y = c.y; r.z = c.z; r.w = c.w;
    return r;
}

// Four boxes at once: sort keys k = entry distance (>= 0) of a child the ray may enter before `tbest`, +inf for a child
// it cannot (absent children hold an inverted box and always get +inf).  Two children per packed-f32 instruction.
typedef float f2 __attribute__((ext_vector_type(2)));
NR_DEV void box_keys4(const NodePlanes& p, const RayF& r, float tbest, float& k0, float& k1, float& k2, float& k3) {
    const f2 ix = {r.ix, r.ix}, iy = {r.iy, r.iy}, iz = {r.iz, r.iz};
    const f2 nnx = {r.nnx, r.nnx}, nny = {r.nny, r.nny}, nnz = {r.nnz, r.nnz}, nfx = {r.nfx, r.nfx}, nfy = {r.nfy, r.nfy}, nfz = {r.nfz, r.nfz};
    const f2 xn0 = __builtin_elementwise_fma(f2{p.xn.x, p.xn.y}, ix, nnx), xn1 = __builtin_elementwise_fma(f2{p.xn.z, p.xn.w}, ix, nnx);
    const f2 xf0 = __builtin_elementwise_fma(f2{p.xf.x, p.xf.y}, ix, nfx), xf1 = __builtin_elementwise_fma(f2{p.xf.z, p.xf.w}, ix, nfx);
    const f2 yn0 = __builtin_elementwise_fma(f2{p.yn.x, p.yn.y}, iy, nny), yn1 = __builtin_elementwise_fma(f2{p.yn.z, p.yn.w}, iy, nny);
    const f2 yf0 = __builtin_elementwise_fma(f2{p.yf.x, p.yf.y}, iy, nfy), yf1 = __builtin_elementwise_fma(f2{p.yf.z, p.yf.w}, iy, nfy);
    const f2 zn0 = __builtin_elementwise_fma(f2{p.zn.x, p.zn.y}, iz, nnz), zn1 = __builtin_elementwise_fma(f2{p.zn.z, p.zn.w}, iz, nnz);
    const f2 zf0 = __builtin_elementwise_fma(f2{p.zf.x, p.zf.y}, iz, nfz), zf1 = __builtin_elementwise_fma(f2{p.zf.z, p.zf.w}, iz, nfz);
    const float n0 = fmaxf(fmaxf(xn0.x, yn0.x), fmaxf(zn0.x, 0.0f)), f0 = fminf(fminf(xf0.x, yf0.x), fminf(zf0.x, tbest));
    const float n1 = fmaxf(fmaxf(xn0.y, yn0.y), fmaxf(zn0.y, 0.0f)), f1 = fminf(fminf(xf0.y, yf0.y), fminf(zf0.y, tbest));
    const float n2 = fmaxf(fmaxf(xn1.x, yn1.x), fmaxf(zn1.x, 0.0f)), f2_ = fminf(fminf(xf1.x, yf1.x), fminf(zf1.x, tbest));
    const float n3 = fmaxf(fmaxf(xn1.y, yn1.y), fmaxf(zn1.y, 0.0f)), f3 = fminf(fminf(xf1.y, yf1.y), fminf(zf1.y, tbest));
    // n <= f (1 + 1.4e-6): the relative slack of both distances on one side (n >= 0; a negative f never passes)
    const float kMiss = __builtin_inff(), kSlack = 1.0000014f;
    k0 = n0 <= f0 * kSlack ? n0 : kMiss;
    k1 = n1 <= f1 * kSlack ? n1 : kMiss;
    k2 = n2 <= f2_ * kSlack ? n2 : kMiss;
    k3 = n3 <= f3 * kSlack ? n3 : kMiss;
}

// The axes box_keys4 left unconstrained (the low bits of RayF::bits: rays exactly parallel to a coordinate plane — rare, but a
// camera on a symmetry plane of the scene produces a whole column of them): ncollide's ray_aabb rejects such a ray iff its
// origin coordinate lies outside [mn, mx] (SURVEY B-3).  o32 = fl(o) is off by at most 2^-24 |o| and the f32 bounds
// contain the f64 ones, so "o32 + m < mn or o32 - m > mx" with m = 2^-22 |o32| + 1e-20 implies the reference's rejection.
// (The 1e-20 covers components that are not exactly zero but below 1e-30, which the f32 view also treats as zero: the
// reference would divide by them, and an origin less than 1e-20 outside the slab could still reach it within 1e10 units.)
// For such an axis RayF::bits selects the lower bound as "entry" plane (inverse 0 is not negative).
NR_DEV void zero_axis_cull(uint32_t zero_axes, d3 o, const NodePlanes& p, float& k0, float& k1, float& k2, float& k3) {
    const float kMiss = __builtin_inff();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (!(zero_axes & (1u << a))) continue;
        const float o32 = (float)comp(o, a), m = fabsf(o32) * 2.384185791015625e-07f + 1e-20f;
        const float4 mn = a == 0 ? p.xn : (a == 1 ? p.yn : p.zn), mx = a == 0 ? p.xf : (a == 1 ? p.yf : p.zf);
        if (o32 + m < mn.x || o32 - m > mx.x) k0 = kMiss;
        if (o32 + m < mn.y || o32 - m > mx.y) k1 = kMiss;
        if (o32 + m < mn.z || o32 - m > mx.z) k2 = kMiss;
        if (o32 + m < mn.w || o32 - m > mx.w) k3 = kMiss;
    }
}

// Whether a primary ray can reach anything at all: exactly the first step of traverse<false> (same f32
// ray view, same root fetch, same conservative box test), so "false" means that traversal would return
// a miss and Scene::trace the background colour.  k_primary uses it to let wave tiles of empty screen
// skip the trace machinery altogether.
NR_DEV bool primary_may_hit(const DScene& S, d3 o, d3 d) {
    if (S.num_planes) return true; // planes are unbounded
    int32_t cur = S.closest_root;
    if (cur == kEmptyChild) return false;
    if (cur < 0) return true; // single leaf: no box above it
    RayF rf = make_rayf(o, d);
    const NodePlanes p = load_planes(S.nodes, cur, rf);
    float k0, k1, k2, k3;
    box_keys4(p, rf, best_f32(kDblMax), k0, k1, k2, k3);
    if ((rf.bits & 7u)) zero_axis_cull((rf.bits & 7u), o, p, k0, k1, k2, k3);
    return fminf(fminf(k0, k1), fminf(k2, k3)) < __builtin_inff();
}

// Number of boxes the root node holds (instrumented builds: the AABB tests primary_may_hit stands for).
NR_DEV unsigned root_children(const DScene& S) {
    if (S.closest_root < 0) return 0u;
    int4 ch = load_children(S.nodes, S.closest_root);
    return (unsigned)(ch.x != kEmptyChild) + (unsigned)(ch.y != kEmptyChild) + (unsigned)(ch.z != kEmptyChild) + (unsigned)(ch.w != kEmptyChild);
}

// ncollide ray_aabb (AABB::toi_with_ray, solid = true; SURVEY B-3) as a predicate, in f64 and in the
// reference's operation order.  The reference only casts a node / tests a triangle whose AABB this
// test accepts (src/scene.rs:276 and the TriMesh BVT), so it is applied to every ACCEPTED hit; the
// conservative f32 culling above only decides what is looked at.
NR_DEV bool aabb_pass(double mnx, double mny, double mnz, double mxx, double mxy, double mxz, d3 o, d3 d) {
    double tmax = kDblMax, tmin = -kDblMax;
    const double mn[3] = {mnx, mny, mnz}, mx[3] = {mxx, mxy, mxz};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double dd = comp(d, i), oo = comp(o, i);
        if (dd == 0.0) {
            if (oo < mn[i] || oo > mx[i]) return false;
        } else {
            double denom = 1.0 / dd;
            double tn = (mn[i] - oo) * denom, tf = (mx[i] - oo) * denom;
            if (tn > tf) { double s = tn; tn = tf; tf = s; }
            if (tn > tmin) tmin = tn;
            if (tf < tmax) tmax = tf;
            if (tmax < 0.0 || tmin > tmax) return false;
        }
    }
    return true;
}
NR_DEV bool node_aabb_pass(const DScene& S, uint32_t node_id, d3 o, d3 d) {
    const double* b = S.node_aabbs + 6 * (size_t)node_id;
    return aabb_pass(b[0], b[1], b[2], b[3], b[4], b[5], o, d);
}
NR_DEV bool tri_aabb_pass(d3 a, d3 b, d3 c, d3 o, d3 d) {
    return aabb_pass(fmin(a.x, fmin(b.x, c.x)), fmin(a.y, fmin(b.y, c.y)), fmin(a.z, fmin(b.z, c.z)),
                     fmax(a.x, fmax(b.x, c.x)), fmax(a.y, fmax(b.y, c.y)), fmax(a.z, fmax(b.z, c.z)), o, d);
}

// Reconstructs the full intersection record of a finished closest-hit query.
// Returns whether the hit passes the reference's AABB gates (only evaluated when CHECK is set).
template <bool SHADOW, int FEAT, bool CHECK = false>
NR_DEV bool resolve_hit(const DScene& S, d3 o, d3 d, const Hit& h, Isect& out, uint32_t& node_id) {
    const Instance& in = (SHADOW ? S.shadow_instances : S.instances)[h.inst];
    if ((FEAT & kFeatAnalytic) && (!(FEAT & kFeatMesh) || in.kind != NRAYS_SHAPE_TRIMESH)) {
        out = cast_instance<FEAT>(in, o, d);
        node_id = (uint32_t)in.node_id;
        return !CHECK || in.kind == NRAYS_SHAPE_PLANE || node_aabb_pass(S, node_id, o, d);
    }
    if (!(FEAT & kFeatMesh)) { node_id = 0; out.toi = 0.0; out.n = D3(0, 0, 0); out.u = out.v = 0.0; out.has_uv = false; return true; }
    Xform m;
    d3 lo = o, ld = d;
    constexpr bool kNoXf = (FEAT & kFeatNoXform) != 0;
    if (!kNoXf) {
        load_xform(in, m);
        if (!(in.flags & kInstIdentityRot)) { lo = inv_rot(m, o - m.t); ld = inv_rot(m, d); }
        else lo = o - m.t;
    }
    const TriRec& tr = S.tris[h.prim];
    d3 a = D3(tr.v0[0], tr.v0[1], tr.v0[2]), b = D3(tr.v1[0], tr.v1[1], tr.v1[2]), c = D3(tr.v2[0], tr.v2[1], tr.v2[2]);
    double bary[3]; d3 n; double toi;
    cast_triangle(a, b, c, lo, ld, toi, &n, bary);
    out.toi = toi;
    out.n = (kNoXf || (in.flags & kInstIdentityRot)) ? n : rot(m, n);
    node_id = tr.node_id;
    out.has_uv = (S.shade[node_id].flags & 1u) != 0;
    out.u = 0.0; out.v = 0.0;
    if (out.has_uv) {
        const TriUv& uv = S.triuvs[h.prim];
        out.u = (double)uv.uv[0] * bary[0] + (double)uv.uv[2] * bary[1] + (double)uv.uv[4] * bary[2];
        out.v = (double)uv.uv[1] * bary[0] + (double)uv.uv[3] * bary[1] + (double)uv.uv[5] * bary[2];
    }
    return !CHECK || (tri_aabb_pass(a, b, c, lo, ld) && node_aabb_pass(S, node_id, o, d));
}

// Shadow-ray bookkeeping of one node-closest hit (scene.rs:313-338): returns true if it blocks.
template <bool STATS>
NR_DEV bool shadow_node_hit(const DScene& S, uint32_t node_id, const Isect& is, f3& filter, Cnt& cnt) {
    if (STATS) cnt.hit++;
    const ShadeRec& nr = S.shade[node_id];
    f4 color = material_ambiant<STATS>(nr, is, cnt);
    float alpha = color.w * nr.alpha;
    if (alpha < 1.0f) {
        filter.x = (filter.x * color.x) * (1.0f - alpha);
        filter.y = (filter.y * color.y) * (1.0f - alpha);
        filter.z = (filter.z * color.z) * (1.0f - alpha);
        return false;
    }
    return true;
}

// Two-level traversal.
//   SHADOW == false: ClosestRayTOICostFn — global closest hit (ties: smallest node id, then
//                    smallest triangle id); returns true and `hit` if anything was hit.
//   SHADOW == true : TransparentShadowsRayTOICostFn — returns true if an opaque node-closest hit
//                    lies within `tlimit`; otherwise `filter` holds the product of the transparent
//                    node-closest hits' colour filters.
// GATED: apply the reference's exact AABB gates (node world AABB, triangle AABB; 6 f64 divisions) to every
// candidate.  Shadow rays always gate (a hit decides immediately).  Closest-hit rays run UNGATED and the
// caller verifies only the winner: the gated candidate set is a subset of the ungated one, so if the
// ungated minimum passes the gates it is also the gated minimum; if it fails (knife-edge rays only) the
// caller re-runs the same traversal with gated_closest = true.
template <bool SHADOW, bool STATS, int FEAT>
NR_DEV bool traverse(const DScene& S, Stack& st, d3 o, d3 d, double tlimit, Hit& hit, f3& filter, Cnt& cnt, bool gated_closest = false,
                     Isect* winner = nullptr) {
    // Analytic-only scenes keep the winner's full record (no second cast in resolve_hit); scenes with meshes
    // only carry (t, instance, triangle) through the loop and reconstruct the record afterwards.
    constexpr bool kKeepIsect = !SHADOW && !(FEAT & kFeatMesh);
    Isect bis;
    if (kKeepIsect) { bis.toi = 0.0; bis.n = D3(0.0, 0.0, 0.0); bis.u = 0.0; bis.v = 0.0; bis.has_uv = false; bis.hit = false; }
    const bool GATED = SHADOW || gated_closest;
    constexpr bool kAnalytic = (FEAT & kFeatAnalytic) != 0, kMesh = (FEAT & kFeatMesh) != 0;
    constexpr bool kAlpha = (FEAT & kFeatAlphaShadow) != 0; // shadow mode: otherwise every hit within tlimit blocks
    // the opaque-mesh kernels only (where hair lives): the three scalar instructions per visit cost the alpha-shadow kernels 2 %
    constexpr bool kQuorum = NR_NODE_QUORUM_DEN != 0 && kMesh && !kAnalytic && !kAlpha;
    constexpr bool kNoXf = (FEAT & kFeatNoXform) != 0; // no instance moves the ray: co / cd stay o / d (the compiler then keeps one copy)
    const Instance* insts = SHADOW ? S.shadow_instances : S.instances;
    const InstLink* links = SHADOW ? S.shadow_links : S.links;
    double bt = SHADOW ? tlimit : kDblMax;
    unsigned long long bkey = ~0ULL;
    bool bhit = false;
    uint32_t binst = 0, bprim = 0;
    // current (possibly instance-local) ray
    d3 co = o, cd = d;
    RayF rf = make_rayf(o, d);
    float btf = best_f32(bt);
    bool in_blas = false;
    uint32_t cur_inst = 0, cur_flags = 0;
    st.reset();
    // planes have infinite AABBs (ncollide Plane AABB = +-MAX): kept out of the BVH and visited as
    // pseudo-leaves, pushed first so that they are tested after the TLAS has tightened the bound.
    if (kAnalytic) {
        const int32_t* planes = SHADOW ? S.shadow_planes : S.planes;
        for (uint32_t p = 0; p < S.num_planes; ++p) st.push(~(int32_t)(((uint32_t)planes[p]) << 3));
    }
    int32_t cur = SHADOW ? S.shadow_root : S.closest_root;
    if (cur == kEmptyChild) cur = st.pop();

    // "while-while" traversal: the wave first runs internal-node steps only (one 64-byte fetch and two
    // f32 box tests per step) until every lane holds a leaf or is done, then runs the leaf code, instead
    // of serialising node / instance / triangle / sentinel code paths in every iteration.
    NR_TIC(tphase);
    for (;;) {
        NR_TOC(cyc_leaf, tphase);
        NR_INQ_COUNT(lanes_in_query);
        // Scenes with hair-like meshes (DScene::incoherent): the lanes of a wave walk different nodes and reach their leaves after very different
        // numbers of steps, and a node phase that lasts until the LAST lane holds a leaf leaves most lanes idle (hairball: 11 of 64
        // lanes active per node step).  There the node phase ends once fewer than a third of the query's lanes are still on internal
        // nodes; those lanes sit out the leaf phase and step on afterwards.  Results do not depend on the visiting order.  (Coherent
        // scenes lose: stragglers that fall behind the packet turn its wave-uniform visits into divergent ones; sponza +4 %.)
        const int node_quorum = kQuorum && S.incoherent ? __popcll(__ballot(1)) / NR_NODE_QUORUM_DEN : 0;
        while (cur >= 0) {
            NR_ITER(wv_node, ln_node);
            NR_INQ(inq_node, lanes_in_query);
            NR_UNIFORM(cur);
            // sort keys: entry distance, +inf for a child the ray cannot enter (absent children always: inverted boxes)
            const float kMiss = __builtin_inff();
            float k0, k1, k2, k3;
            int32_t c0, c1, c2, c3;
            const uint32_t nkey = ((uint32_t)cur << 7) | rf.bits;
            const uint32_t ukey = (uint32_t)__builtin_amdgcn_readfirstlane((int)nkey);
            if (NR_SCALAR_NODES && kMesh && __ballot(nkey != ukey) == 0ULL) { // wave-uniform: the active lanes share the node and the direction signs
                const int4 ch = load_children_uniform(S.nodes, ukey);
                const NodePlanes np = load_planes_uniform(S.nodes, ukey);
                // SURVEY 8d counts AABB tests: only the boxes that exist (an absent child slot is not a test)
                if (STATS) cnt.node += (unsigned)(ch.x != kEmptyChild) + (unsigned)(ch.y != kEmptyChild) + (unsigned)(ch.z != kEmptyChild) + (unsigned)(ch.w != kEmptyChild);
                if (STATS && (int)__lane_id() == __ffsll((long long)__ballot(1)) - 1) cnt.fetch++; // one scalar fetch serves the wave
                box_keys4(np, rf, btf, k0, k1, k2, k3);
                if ((rf.bits & 7u)) zero_axis_cull((rf.bits & 7u), co, np, k0, k1, k2, k3);
                c0 = ch.x; c1 = ch.y; c2 = ch.z; c3 = ch.w;
            } else {
                const int4 ch = load_children(S.nodes, cur);
                const NodePlanes np = load_planes(S.nodes, cur, rf);
                if (STATS) cnt.node += (unsigned)(ch.x != kEmptyChild) + (unsigned)(ch.y != kEmptyChild) + (unsigned)(ch.z != kEmptyChild) + (unsigned)(ch.w != kEmptyChild);
                if (STATS) cnt.fetch++;
                box_keys4(np, rf, btf, k0, k1, k2, k3);
                if ((rf.bits & 7u)) zero_axis_cull((rf.bits & 7u), co, np, k0, k1, k2, k3);
                c0 = ch.x; c1 = ch.y; c2 = ch.z; c3 = ch.w;
            }
            if (!SHADOW) {
                // closest hit: 5-comparator sorting network on (key, ref), ascending entry distance;
                // farthest first onto the stack, nearest becomes the next node
#define NR_CSWAP(ka, ca, kb, cb) { bool sw = kb < ka; float tk = sw ? kb : ka; kb = sw ? ka : kb; ka = tk; int32_t tc = sw ? cb : ca; cb = sw ? ca : cb; ca = tc; }
                NR_CSWAP(k0, c0, k1, c1) NR_CSWAP(k2, c2, k3, c3) NR_CSWAP(k0, c0, k2, c2) NR_CSWAP(k1, c1, k3, c3) NR_CSWAP(k1, c1, k2, c2)
#undef NR_CSWAP
                if (__builtin_expect(st.wave_has_room(3), 1)) {
                    // the common case, without a branch per child: every candidate is written, a miss is overwritten by
                    // the next one
                    lds_u32* a = st.top;
                    *a = (uint32_t)c3; a += k3 < kMiss ? kBlock : 0;
                    *a = (uint32_t)c2; a += k2 < kMiss ? kBlock : 0;
                    *a = (uint32_t)c1; a += k1 < kMiss ? kBlock : 0;
                    st.top = a;
                } else {
                    if (k3 < kMiss) st.push(c3);
                    if (k2 < kMiss) st.push(c2);
                    if (k1 < kMiss) st.push(c1);
                }
                if (k0 < kMiss) cur = c0;
                else cur = st.pop();
            } else {
                // shadow rays are any-hit (or per-node closest with a result that does not depend on the
                // visiting order): no sort, the last child hit is visited next, the others go onto the stack
                const bool h0 = k0 < kMiss, h1 = k1 < kMiss, h2 = k2 < kMiss, h3 = k3 < kMiss;
                cur = h3 ? c3 : (h2 ? c2 : (h1 ? c1 : (h0 ? c0 : kEmptyChild)));
                const bool p0 = h0 && (h1 || h2 || h3), p1 = h1 && (h2 || h3), p2 = h2 && h3;
                if (__builtin_expect(st.wave_has_room(3), 1)) {
                    lds_u32* a = st.top;
                    *a = (uint32_t)c0; a += p0 ? kBlock : 0;
                    *a = (uint32_t)c1; a += p1 ? kBlock : 0;
                    *a = (uint32_t)c2; a += p2 ? kBlock : 0;
                    st.top = a;
                } else {
                    if (p0) st.push(c0);
                    if (p1) st.push(c1);
                    if (p2) st.push(c2);
                }
                if (cur == kEmptyChild) cur = st.pop();
            }
            if (kQuorum && node_quorum && __popcll(__ballot(cur >= 0)) < node_quorum) { // wave-uniform
                if (cur >= 0) { st.push(cur); cur = kParked; } // leaves the node phase through the loop condition, like a lane with a leaf
            }
        }
        NR_TOC(cyc_node, tphase);
        if (kQuorum && cur == kParked) { cur = st.pop(); continue; } // sits out the leaf phase of the others
        if (cur == kEmptyChild) break;
        if (kMesh && cur == kSentinel) { // the BLAS of `cur_inst` is exhausted: back to world space
            in_blas = false;
            if (!kNoXf && !(cur_flags & kInstNoXform)) { co = o; cd = d; rf = make_rayf(o, d); }
            if (SHADOW && kAlpha && !(cur_flags & kInstAnyHit)) {
                if (bhit) {
                    Hit h; h.t = bt; h.inst = cur_inst; h.prim = bprim;
                    Isect is; uint32_t node_id;
                    resolve_hit<true, FEAT>(S, o, d, h, is, node_id);
                    if (shadow_node_hit<STATS>(S, node_id, is, filter, cnt)) return true;
                }
                bt = tlimit; bkey = ~0ULL; bhit = false; btf = best_f32(bt);
            }
            cur = st.pop();
            continue;
        }
        // leaf: ~cur = (first << 3) | bits.  Triangle leaves: bits = count - 1.  TLAS leaves: bits = kLeaf* flags.
        uint32_t lv = (uint32_t)~cur;
        uint32_t first = lv >> 3, bits = lv & 7u;
        if (kMesh && in_blas) { // triangle leaf
            NR_TIC(ttri);
            const float4* tq = (const float4*)(S.tris + first);
            float4 p0 = tq[0], p1 = tq[1], p2 = tq[2];
            const int32_t after_leaf = st.pop(); // popped now: the LDS latency hides behind the tests
#pragma nounroll
            for (uint32_t k = 0; k <= bits; ++k) {
                const float4 t0 = p0, t1 = p1, t2 = p2;
                if (k < bits) { tq += 3; p0 = tq[0]; p1 = tq[1]; p2 = tq[2]; }
                if (STATS) cnt.tri++;
                NR_ITER(wv_tri, ln_tri);
                NR_INQ(inq_tri, lanes_in_query);
                double toi;
                d3 va = D3(t0.x, t0.y, t0.z), vb = D3(t1.x, t1.y, t1.z), vc = D3(t2.x, t2.y, t2.z);
                if (cast_triangle(va, vb, vc, co, cd, toi, nullptr, nullptr) &&
                    (!GATED || (tri_aabb_pass(va, vb, vc, co, cd) && node_aabb_pass(S, __float_as_uint(t0.w), o, d)))) {
                    if (SHADOW && (!kAlpha || (cur_flags & kInstAnyHit))) { if (toi <= tlimit) return true; }
                    else {
                        unsigned long long key = SHADOW ? (unsigned long long)__float_as_uint(t1.w)
                                                        : (((unsigned long long)__float_as_uint(t0.w) << 32) | __float_as_uint(t1.w));
                        if (toi < bt || (toi == bt && key < bkey)) { bt = toi; bkey = key; bhit = true; binst = cur_inst; bprim = first + k; btf = best_f32(bt); }
                    }
                }
            }
            cur = after_leaf;
            NR_TOC(cyc_tri, ttri);
            continue;
        }
        if (kMesh && (!kAnalytic || (bits & kLeafMesh))) { // TLAS leaf: a BLAS
            cur_inst = first;
            InstLink link = links[first];
            cur_flags = link.flags;
            if (!kNoXf && !(bits & kLeafNoXform)) { // rotated / translated instance: move the ray into its local frame
                const Instance& in = insts[first];
                Xform m; load_xform(in, m);
                if (in.flags & kInstIdentityRot) { co = o - m.t; cd = d; }
                else { co = inv_rot(m, o - m.t); cd = inv_rot(m, d); }
                rf = make_rayf(co, cd);
            }
            in_blas = true;
            st.push(kSentinel);
            cur = link.blas_root;
            if (cur == kEmptyChild) cur = st.pop();
            continue;
        }
        if (kAnalytic) { // TLAS leaf: an analytic shape (or a plane pseudo-leaf)
            const Instance& in = insts[first];
            if (STATS) cnt.prim++;
            Isect is = cast_instance<FEAT>(in, o, d, !(SHADOW && !kAlpha)); // an opaque shadow hit only needs its distance
            if (is.hit && (!GATED || in.kind == NRAYS_SHAPE_PLANE || node_aabb_pass(S, (uint32_t)in.node_id, o, d))) {
                if (SHADOW) {
                    if (is.toi <= tlimit) {
                        if (!kAlpha) return true;
                        if (shadow_node_hit<STATS>(S, (uint32_t)in.node_id, is, filter, cnt)) return true;
                    }
                } else {
                    unsigned long long key = (unsigned long long)(uint32_t)in.node_id << 32;
                    if (is.toi < bt || (is.toi == bt && key < bkey)) {
                        bt = is.toi; bkey = key; bhit = true; binst = first; bprim = 0; btf = best_f32(bt);
                        if (kKeepIsect) bis = is;
                    }
                }
            }
        }
        cur = st.pop();
    }

    if (SHADOW) return false;
    hit.t = bt; hit.inst = binst; hit.prim = bprim;
    if (kKeepIsect && winner) *winner = bis;
    return bhit;
}

// ---------------------------------------------------------------- shading --------------------
struct RayState { // RayWithEnergy (ray_with_energy.rs:4-8) + bookkeeping of the iterative formulation
    d3 o, d;
    double refr;
    float energy;
    float weight;
    unsigned long long key;
    uint32_t pixel;
};

// PhongMaterial::compute (phong_material.rs:72-151); other materials fall back to ambiant (material.rs:8-16).
#ifdef NR_MAT_NOINLINE
#define NR_MAT_ATTR __device__ __noinline__
#else
#define NR_MAT_ATTR NR_DEV
#endif
#ifdef NR_SHADOW_NOINLINE
template <bool STATS, int FEAT>
__device__ __noinline__ bool shadow_query(const DScene& S, Stack& st, d3 o, d3 d, double tlimit, f3& filter, Cnt& cnt) {
    Hit dummy;
    return traverse<true, STATS, FEAT>(S, st, o, d, tlimit, dummy, filter, cnt);
}
#else
template <bool STATS, int FEAT>
NR_DEV bool shadow_query(const DScene& S, Stack& st, d3 o, d3 d, double tlimit, f3& filter, Cnt& cnt) {
    Hit dummy;
    return traverse<true, STATS, FEAT>(S, st, o, d, tlimit, dummy, filter, cnt);
}
#endif
// A light sample on the far side of the surface: phong_material.rs:109-141 traces its shadow ray first and THEN multiplies the filter by
// diffuse = kd * max(dot(l, n) as f32, 0) and, only `if scoeff > 0`, a specular term with scoeff = -dot(normalize(2 n (l.n) - l), dir) as f32.
// Where (l.n as f32) <= 0 and scoeff <= 0 the sample adds light.color * (filter * 0) = +-0 to a sum that is never -0: the pixel does not depend on the
// shadow ray (finite colours), so plain renders count it (rays_shadow stays the reference's number; rays_shadow_elided) and do not trace it.  dcoeff is
// tested on the very expression the shading uses; scoeff's sign is decided WITHOUT the normalisation — the mirrored direction has length 1 up to
// rounding (l, n unit), so the normalised dot differs from dot(ru, dir) by < 1e-14: beyond the 1e-9 margin the sign is certain, inside it the ray is traced.
// A material without a specular colour (Ks 0 0 0, most of an OBJ scene's materials): specular = ks * scoeff^shininess = 0 for every scoeff in (0, 1], and
// diffuse + 0 = diffuse — the sample is dark as soon as the light is behind the surface.
// Both elisions rest on x * 0 == 0 for every skipped x: nrays_scene_create checks that every light (position, radius, colour), material colour and RGBA32F texel
// is finite and no shininess negative, and sets DScene::no_elide otherwise — such a scene is rendered by the instrumented (STATS) kernel with stats_elide = 0, which
// traces and shades everything the reference does, NaN for NaN (tests/test_elision_gpu.py); the plain kernels test nothing at run time (a scalar load and a branch in the
// light loop cost the 8-light sponza frame 4 %).  `normal` and `dir` are unit vectors by construction (the casts return normalised normals, ncollide's contract; rays are normalised
// where they are made), never scene input.
NR_DEV bool light_is_dark(d3 ldir, d3 normal, d3 dir, bool no_specular) {
    const double dln = dot(ldir, normal);
    if ((float)dln > 0.0f) return false;
    if (no_specular) return true;
    const d3 ru = (-ldir) + (normal * dln) * 2.0;
    return dot(ru, dir) > 1e-9;
}
NR_DEV bool no_specular(const ShadeRec& m) { return NR_ELIDE_DARK_MATTE && m.ks[0] == 0.0f && m.ks[1] == 0.0f && m.ks[2] == 0.0f && m.shininess >= 0.0f; } // (a negative exponent could make 0 * inf)
// `lsl` > 0 (light-parallel wave tiles, k_primary): 2^lsl consecutive lanes hold the SAME hit — they traced the same ray — and
// share its light loop: lane slot j of the group traces the shadow rays of lights j, j + 2^lsl, ..., and the per-light sums are
// then folded into `res` in light order by every lane of the group (`__shfl` from the lane that holds light l), i.e. exactly
// `res = res + acc_l / n_l` light after light (phong_material.rs:106-147): the pixel is bit-identical to the one-lane loop.
template <bool STATS, int FEAT>
NR_MAT_ATTR f4 material_compute(const DScene& S, Stack& st, const ShadeRec& m, RayState& ray, d3& point, Isect& in, Cnt& cnt,
                                bool pre, bool pre_lit, f3 pre_filter, uint32_t lsl = 0u, float alpha_in = -1.0f) {
    if (((m.flags >> 8) & 0xffu) != NRAYS_MAT_PHONG) return material_ambiant<STATS>(m, in, cnt);
    f4 tex; tex.x = tex.y = tex.z = tex.w = 1.0f;
    float alpha = 1.0f;
    if (in.has_uv && m.tex.texels) tex = tex_sample<STATS>(m.tex, in.u, in.v, cnt);
    if (alpha_in >= 0.0f) alpha = alpha_in; // (shade_hit() sampled the opacity map already)
    else if (in.has_uv && m.alpha_tex.texels) alpha = tex_sample<STATS>(m.alpha_tex, in.u, in.v, cnt).w;
    f3 res = F3(m.ka[0] * tex.x, m.ka[1] * tex.y, m.ka[2] * tex.z);
    d3 normal = in.n;
    const bool no_spec = no_specular(m);
    // one light: the sum over its samples (light.rs:57-63 + phong_material.rs:108-146)
    auto light_sum = [&](uint32_t li) -> f3 {
        const LightRec& light = S.lights[li];
        f3 acc = F3(0.0f, 0.0f, 0.0f);
        uint32_t ns = light.racsample * light.racsample;
        unsigned long long lkey = rng_hash(ray.key, kSaltLight + li);
#pragma nounroll
        for (uint32_t k = 0; k < ns; ++k) {
            d3 pos = D3(light.pos[0], light.pos[1], light.pos[2]);
            if (light.radius != 0.0) { // light.rs:59-61 (cube-octant jitter)
                unsigned long long sk = rng_hash(lkey, k);
                d3 rnd = D3(rng_u01(sk, 0), rng_u01(sk, 1), rng_u01(sk, 2));
                pos = pos + rnd * light.radius;
            }
            d3 ldir = pos - point;
            double nrm = norm(ldir);
            ldir = ldir / nrm;
            double dist = nrm - 0.001;
            d3 so = point + ldir * 0.001;
            f3 filter = F3(1.0f, 1.0f, 1.0f);
            if (!(FEAT & kFeatMultiSample)) { // the single shadow ray of this hit was traced before the shading state existed
                if (!pre || !pre_lit) continue;
                filter = pre_filter;
            } else {
                cnt.shadow++;
                if ((!STATS || (S.stats_elide & 1u)) && NR_ELIDE_DARK && light_is_dark(ldir, normal, ray.d, no_spec)) { cnt.elided++; continue; }
                NR_TIC(tsq);
                // kFeatPark: what the Phong terms below need of this hit waits in LDS while the shadow ray is traced (the values are the same
                // bits afterwards; `in.n`, `point` and `ray.d` are the caller's objects, so its later uses read the reloaded registers too)
                if ((FEAT & kFeatPark) && (FEAT & kFeatAlphaShadow) && park_slots(FEAT) >= 12) {
                    st.park_d3(0, normal); st.park_d3(6, point); if (park_slots(FEAT) >= 18) st.park_d3(12, ray.d);
                    if (park_slots(FEAT) >= 25) { st.park_f(18, tex.x); st.park_f(19, tex.y); st.park_f(20, tex.z); st.park_f(21, tex.w); st.park_f(22, res.x); st.park_f(23, res.y); st.park_f(24, res.z); }
                }
                const bool blocked = shadow_query<STATS, FEAT>(S, st, so, ldir, dist, filter, cnt);
                if ((FEAT & kFeatPark) && (FEAT & kFeatAlphaShadow) && park_slots(FEAT) >= 12) {
                    normal = st.unpark_d3(0); in.n = normal; point = st.unpark_d3(6); if (park_slots(FEAT) >= 18) ray.d = st.unpark_d3(12);
                    if (park_slots(FEAT) >= 25) { tex.x = st.unpark_f(18); tex.y = st.unpark_f(19); tex.z = st.unpark_f(20); tex.w = st.unpark_f(21); res.x = st.unpark_f(22); res.y = st.unpark_f(23); res.z = st.unpark_f(24); }
                }
                NR_TOC(cyc_shadow, tsq);
                if (blocked) continue; // shadowed
            }
            double dot_ldir_norm = dot(ldir, normal);
            float dcoeff = (float)dot_ldir_norm;
            dcoeff = dcoeff > 0.0f ? dcoeff : 0.0f;
            f3 diffuse_color = F3(m.kd[0] * tex.x, m.kd[1] * tex.y, m.kd[2] * tex.z);
            f3 diffuse = F3(diffuse_color.x * dcoeff, diffuse_color.y * dcoeff, diffuse_color.z * dcoeff);
            d3 lproj = normal * dot_ldir_norm;
            d3 rldir = normalize((-ldir) + lproj * 2.0);
            float scoeff = (float)(-dot(rldir, ray.d));
            if (scoeff > 0.0f) {
                scoeff = powf(scoeff, m.shininess);
                f3 sp = F3(m.ks[0] * scoeff, m.ks[1] * scoeff, m.ks[2] * scoeff);
                acc.x = acc.x + light.color[0] * (filter.x * (diffuse.x + sp.x));
                acc.y = acc.y + light.color[1] * (filter.y * (diffuse.y + sp.y));
                acc.z = acc.z + light.color[2] * (filter.z * (diffuse.z + sp.z));
            } else {
                acc.x = acc.x + light.color[0] * (filter.x * diffuse.x);
                acc.y = acc.y + light.color[1] * (filter.y * diffuse.y);
                acc.z = acc.z + light.color[2] * (filter.z * diffuse.z);
            }
        }
        return acc;
    };
    constexpr bool kSplit = (FEAT & kFeatMultiSample) && (FEAT & kFeatMesh) && !(FEAT & kFeatDouble);
    if (kSplit && lsl != 0u) { // wave-uniform
        const uint32_t nslot = 1u << lsl, lane = __lane_id(), slot = lane & (nslot - 1u), gbase = lane & ~(nslot - 1u);
#pragma nounroll
        for (uint32_t base = 0; base < S.num_lights; base += nslot) {
            const uint32_t li = base + slot;
            f3 acc = F3(0.0f, 0.0f, 0.0f);
            if (li < S.num_lights) acc = light_sum(li);
            for (uint32_t j = 0; j < nslot && base + j < S.num_lights; ++j) { // wave-uniform bounds
                const uint32_t rs = S.lights[base + j].racsample;
                const float inv = 1.0f / (float)(rs * rs);
                const float ax = __shfl(acc.x, (int)(gbase + j)), ay = __shfl(acc.y, (int)(gbase + j)), az = __shfl(acc.z, (int)(gbase + j));
                res.x = inv * ax + res.x; res.y = inv * ay + res.y; res.z = inv * az + res.z;
            }
        }
    } else {
#pragma nounroll
        for (uint32_t li = 0; li < S.num_lights; ++li) {
            const f3 acc = light_sum(li);
            const uint32_t rs = S.lights[li].racsample;
            float inv = 1.0f / (float)(rs * rs);
            res.x = inv * acc.x + res.x; res.y = inv * acc.y + res.y; res.z = inv * acc.z + res.z;
        }
    }
    f4 out; out.x = res.x; out.y = res.y; out.z = res.z; out.w = alpha;
    return out;
}

// Continuation rays of one wave are appended to the next generation's queue with one atomic per
// wave: ballot -> popcount prefix -> base offset broadcast.
struct QueueOut {
    RayQueue q;
    uint32_t capacity;
    uint32_t* count;          // count of the generation being produced
    unsigned int* overflow;
};

NR_DEV void queue_store(const RayQueue& q, uint32_t i, const RayState& r, uint32_t depth) {
    q.o[0][i] = r.o.x; q.o[1][i] = r.o.y; q.o[2][i] = r.o.z;
    q.d[0][i] = r.d.x; q.d[1][i] = r.d.y; q.d[2][i] = r.d.z;
    q.refr[i] = r.refr; q.energy[i] = r.energy; q.weight[i] = r.weight; q.pixel[i] = r.pixel; q.key[i] = r.key;
    q.depth[i] = depth;
}
NR_DEV void queue_load(const RayQueue& q, uint32_t i, RayState& r, uint32_t& depth) {
    r.o = D3(q.o[0][i], q.o[1][i], q.o[2][i]);
    r.d = D3(q.d[0][i], q.d[1][i], q.d[2][i]);
    r.refr = q.refr[i]; r.energy = q.energy[i]; r.weight = q.weight[i]; r.pixel = q.pixel[i]; r.key = q.key[i];
    depth = q.depth[i];
}

// Appends the flagged lanes' rays to the continuation queue with ONE atomic per wave:
// ballot -> popcount prefix -> base offset broadcast.  Must be reached by every lane of the wave.
NR_DEV void emit_rays(const QueueOut& qo, bool has, const RayState& r, uint32_t depth) {
    unsigned long long m = __ballot(has);
    uint32_t n = (uint32_t)__popcll(m);
    if (n == 0) return; // wave-uniform
    uint32_t lane = __lane_id();
    int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(qo.count, n);
    base = __shfl(base, leader);
    if (has) {
        uint32_t i = base + (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL));
        if (i < qo.capacity) queue_store(qo.q, i, r, depth); else atomicOr(qo.overflow, 1u);
    }
}

// One step of the trace recursion (Scene::trace, scene.rs:163-193): closest hit, shade, weight
// algebra.  Returns this ray's own weighted contribution to its pixel and REPLACES `ray` by its
// continuation (has_next); a second continuation (only possible in kFeatDouble scenes) goes to `extra`.
template <bool STATS, int FEAT>
NR_DEV f3 shade_hit(const DScene& S, Stack& st, RayState& ray, uint32_t depth, uint32_t max_depth,
                    bool& has_next, bool& has_extra, RayState& extra, Cnt& cnt, bool keyed, uint32_t lsl = 0u) {
    // light-parallel wave tiles: the 2^lsl lanes of a pixel trace the SAME chain; only the group's first lane counts its rays
    const bool count_me = lsl == 0u || (__lane_id() & ((1u << lsl) - 1u)) == 0u;
    has_next = false; has_extra = false;
    Hit hit; f3 nofilter = F3(1.0f, 1.0f, 1.0f);
    Isect is; uint32_t node_id;
    bool gated = false;
    bool pre = false, pre_lit = false; f3 pre_filter = F3(1.0f, 1.0f, 1.0f);
    for (;;) { // second iteration only when the ungated winner fails the reference's AABB gates (knife-edge rays)
        NR_TIC(tq);
        const bool any_hit = traverse<false, STATS, FEAT>(S, st, ray.o, ray.d, kDblMax, hit, nofilter, cnt, gated, &is);
#ifdef NR_PHASE_TIMING
        if (depth == 0u) { NR_TOC(cyc_closest0, tq); } else { NR_TOC(cyc_closestN, tq); }
#endif
        if (!any_hit)
            return F3(S.background[0] * ray.weight, S.background[1] * ray.weight, S.background[2] * ray.weight);
        NR_TIC(trs);
        if (!(FEAT & kFeatMesh)) { // `is` is the winner's record already; only the deferred AABB gate is left
            const Instance& in = S.instances[hit.inst];
            node_id = (uint32_t)in.node_id;
            if (gated || in.kind == NRAYS_SHAPE_PLANE || node_aabb_pass(S, node_id, ray.o, ray.d)) break;
        } else { const bool ok_ = resolve_hit<false, FEAT, true>(S, ray.o, ray.d, hit, is, node_id); NR_TOC(cyc_x[2], trs); if (ok_ || gated) break; }
        gated = true;
    }
    NR_TIC(tsh);
    // A hit on a FULLY TRANSPARENT point (opacity-map texel 0, or node alpha 0: the holes of alpha-tested foliage, lace, chains) — or on a perfect mirror — contributes
    // obj.rgb * (weight * 0 * (1 - mix)) = 0 to its pixel whatever its shading is (scene.rs:179-190) — the reference still traces its shadow rays and
    // evaluates Phong.  Here the opacity is sampled first and such a hit goes straight to its refraction ray: the pixel is bit-identical (x + 0 = x;
    // colours are finite), the shadow rays are COUNTED as the reference traces them (DeviceCounters::rays_shadow keeps the oracle's value) but not
    // traced.  Not in the instrumented kernel, whose test / sample counts stay the reference algorithm's.  Sponza stand-in: half of the
    // alpha-mapped hits, all of them on the frame's longest chains (profiles/r05_transparent_hit_elision_ab.log).  Only in the kernels of mesh scenes
    // that hold a transparent node (kFeatMesh + kFeatAlphaShadow): in the opaque-mesh kernels the test alone cost 1 % of the hairball frames and in the
    // analytic ones 1.5 % of the primitives frame (same log); a perfect mirror in such a scene is shaded as the reference shades it.
    float alpha_known = -1.0f;
    bool elide = false;
    if ((!STATS || (S.stats_elide & 2u)) && NR_ELIDE_TRANSPARENT && (FEAT & kFeatAlphaShadow) && (FEAT & kFeatMesh)) {
        const ShadeRec& sm = S.shade[node_id];
        if (((sm.flags >> 8) & 0xffu) == NRAYS_MAT_PHONG) {
            const bool mirror = sm.refl_mix == 1.0f; // (the same for a perfect mirror: its own term is obj.rgb * (weight * alpha * (1 - 1)))
            if ((is.has_uv && sm.alpha_tex.texels) || sm.alpha == 0.0f || mirror) {
                alpha_known = (is.has_uv && sm.alpha_tex.texels) ? tex_sample<STATS>(sm.alpha_tex, is.u, is.v, cnt).w : 1.0f;
                elide = mirror || alpha_known * sm.alpha == 0.0f;
            }
        }
    }
#ifdef NR_PT_SPLIT_MATERIAL
    NR_TOC(cyc_x[6], tsh); // shading record + opacity sample
#endif
    if (elide) {
        if (count_me) { // the shadow rays the reference traces from this hit: one per light sample (light.rs:57-63)
            unsigned n = 0u;
            if (!(FEAT & kFeatMultiSample)) n = (S.num_lights == 1 && S.lights[0].racsample == 1u) ? 1u : 0u;
            else for (uint32_t li = 0; li < S.num_lights; ++li) n += S.lights[li].racsample * S.lights[li].racsample;
            cnt.shadow += n; cnt.elided += n;
        }
    }
    // Single-sample lighting (one point light, or one area light with racsample 1): trace the shadow ray NOW,
    // while only the ray, the hit distance and the chain state are live, and hand the result to the Phong
    // evaluation below — the normal / uv / texture state then never has to survive a traversal.  Same ray,
    // same result as phong_material.rs:109-112; only the evaluation order differs.
    if (!elide && !(FEAT & kFeatMultiSample) && S.num_lights == 1 && ((S.shade[node_id].flags >> 8) & 0xffu) == NRAYS_MAT_PHONG) {
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
            if (count_me) cnt.shadow++;
            pre = true;
            NR_TOC(cyc_x[3], tsh);
            NR_TIC(tsq);
            if ((!STATS || (S.stats_elide & 1u)) && NR_ELIDE_DARK && light_is_dark(ldir, is.n, ray.d, no_specular(S.shade[node_id]))) { if (count_me) cnt.elided++; pre_lit = false; } // (as if shadowed: the sample's term is 0 either way)
            else {
            // kFeatPark: the hit's record and the ray wait in LDS while the shadow ray is traced (the same bits come back)
            if ((FEAT & kFeatPark) && (FEAT & kFeatAlphaShadow)) {
                st.park_d3(0, is.n); st.park_d3(6, ray.d); st.park_d(12, hit.t);
                if (park_slots(FEAT) >= 24) { st.park_d3(14, ray.o); st.park_d(20, is.u); st.park_d(22, is.v); }
            }
            pre_lit = !shadow_query<STATS, FEAT>(S, st, point + ldir * 0.001, ldir, nrm - 0.001, pre_filter, cnt);
            if ((FEAT & kFeatPark) && (FEAT & kFeatAlphaShadow)) {
                is.n = st.unpark_d3(0); ray.d = st.unpark_d3(6); hit.t = st.unpark_d(12);
                if (park_slots(FEAT) >= 24) { ray.o = st.unpark_d3(14); is.u = st.unpark_d(20); is.v = st.unpark_d(22); }
            }
            }
            NR_TOC(cyc_shadow, tsq);
#ifdef NR_PHASE_TIMING
            tsh = __builtin_readcyclecounter();
#endif
        }
    }
    is.toi = hit.t;
    if (STATS) cnt.hit++;
    const ShadeRec& sn = S.shade[node_id];
    d3 pt = ray.o + ray.d * hit.t;
    f4 obj;
#ifdef NR_PT_SPLIT_MATERIAL
    NR_TOC(cyc_x[4], tsh); // between the opacity sample / the shadow query and the material: in a divergent wave, the lanes that skip the shadow query WAIT here for the lanes that run it
#endif
    if (elide) { obj.x = obj.y = obj.z = 0.0f; obj.w = alpha_known; }
    else obj = material_compute<STATS, FEAT>(S, st, sn, ray, pt, is, cnt, pre, pre_lit, pre_filter, lsl, alpha_known);
#ifdef NR_PT_SPLIT_MATERIAL
    NR_TOC(cyc_x[7], tsh); // material_compute
#else
    NR_TOC(cyc_x[4], tsh);
#endif
    bool may_recurse = depth < (uint32_t)kMaxGenerations && (max_depth == 0 || depth < max_depth);
    float mix = sn.refl_mix;
    float alpha = obj.w * sn.alpha;
    float wa = alpha == 1.0f ? ray.weight : ray.weight * alpha; // scene.rs:183-190
    // own term: obj.rgb * (1 - mix), scene.rs:179-180 (applied even when reflection is gated off)
    float wo = wa * (1.0f - mix);
    f3 contrib = F3(obj.x * wo, obj.y * wo, obj.z * wo);
    bool do_refl = mix != 0.0f && ray.energy > 0.1f && may_recurse; // trace_reflection gate, scene.rs:204
    bool do_refr = alpha != 1.0f && may_recurse;                    // trace_refraction gate, scene.rs:229
    d3 dirn = is.n * dot(ray.d, is.n); // normal * dot(dir, normal): shared by both formulas
    if (do_refr) { // scene.rs:229-248
        double n1, n2;
        if (ray.refr == 1.0) { n1 = 1.0; n2 = sn.refr_coeff; } else { n1 = sn.refr_coeff; n2 = 1.0; }
        d3 tangent = ray.d - dirn;
        d3 new_dir = normalize(dirn + tangent * (n2 / n1));
        RayState rt;
        rt.o = pt + new_dir * 0.001; rt.d = new_dir; rt.refr = n2; rt.energy = ray.energy;
        rt.weight = ray.weight * (1.0f - alpha); rt.key = keyed ? rng_hash(ray.key, kSaltRefr) : 0ULL; rt.pixel = ray.pixel;
        if (count_me) cnt.refr++;
        if (do_refl) { if (FEAT & kFeatDouble) { extra = rt; has_extra = true; } }
        else { ray = rt; has_next = true; NR_TOC(cyc_x[5], tsh); return contrib; }
    }
    if (do_refl) { // scene.rs:204-214
        d3 rdir = ray.d - dirn * 2.0;
        ray.o = pt + rdir * 0.001; ray.d = rdir; ray.energy = ray.energy - sn.refl_atenuation;
        ray.weight = wa * mix; ray.key = keyed ? rng_hash(ray.key, kSaltRefl) : 0ULL;
        has_next = true; if (count_me) cnt.refl++;
    }
    NR_TOC(cyc_x[5], tsh);
    return contrib;
}

// The recursion of Scene::trace unrolled into an iterative bounce loop.  A hit that spawns ONE
// continuation (reflection or refraction: every shipped scene) keeps it in registers and loops;
// when a hit spawns both, the refraction ray goes to the compacted HBM queue and is picked up by a
// k_bounce launch.  Returns the sum of the chain's weighted contributions to ray.pixel.
// Must be called by every lane of the wave (inactive lanes pass alive = false).
// `keyed`: the frame consumes RNG keys (AA jitter or an area light); otherwise the per-bounce key hashes are skipped.
template <bool STATS, int FEAT>
NR_DEV f3 trace_chain(const DScene& S, Stack& st, bool alive, RayState ray, uint32_t depth, uint32_t max_depth,
                      const QueueOut& qo, Cnt& cnt, bool keyed, uint32_t lsl = 0u) {
    f3 sum = F3(0.0f, 0.0f, 0.0f);
    while (__ballot(alive) != 0ULL) { // wave-uniform
        bool has_extra = false;
        RayState extra;
        if (FEAT & kFeatDouble) extra = ray;
        if (alive) {
            f3 c = shade_hit<STATS, FEAT>(S, st, ray, depth, max_depth, alive, has_extra, extra, cnt, keyed, lsl);
            sum.x = sum.x + c.x; sum.y = sum.y + c.y; sum.z = sum.z + c.z;
            if (depth > cnt.max_depth) cnt.max_depth = depth;
        }
        if (FEAT & kFeatDouble) emit_rays(qo, has_extra, extra, depth + 1);
        ++depth;
    }
    return sum;
}

// scene.rs:74-89: jitter, NDC, unproject by (P V)^-1, normalise.
// PLAIN: a frame known (on the host) to use no RNG keys — one sample per pixel, no AA jitter, no area light; the generic
// code and the uniforms it needs drop out of the kernel.  (Rounds 2-5 tabulated M[:,0] dx_i and M[:,1] dy_j per column / row in
// a kernel of their own; computing them here costs the same — balls 0.0479 ms either way, primitives 0.202 -> 0.199 — and a
// camera's first frame loses a launch: profiles/r06_raygen_tables_ab.log.)
template <bool PLAIN = false>
NR_DEV void generate_primary(const DRender& R, uint32_t i, uint32_t j, uint32_t s, uint32_t pixel_out, RayState& ray, const double* m_mem = nullptr) {
    if (PLAIN) { // jitter-free, no RNG keys: dx, dy straight from the pixel (scene.rs:81-83)
        double h[4];
        // m_mem: the address of DRender::m in the kernel-argument segment (k_primary).  The first two columns of the matrix are then fetched by VECTOR loads, once per
        // tile, instead of living in 16 more SGPRs across the whole tile loop (the kernels already spill 140 - 250 SGPRs: sponza + 0.8 %, balls + 1 % with them)
        double m0[4], m1[4];
        if (m_mem) {
            unsigned long long addr = (unsigned long long)m_mem;
            asm volatile("" : "+v"(addr)); // a VGPR address: global_load, not s_load
            const __attribute__((address_space(1))) double* mp = (const __attribute__((address_space(1))) double*)addr;
#pragma unroll
            for (int r = 0; r < 4; ++r) { m0[r] = mp[r]; m1[r] = mp[4 + r]; }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) { m0[r] = R.m[r]; m1[r] = R.m[4 + r]; }
        }
        // i / width and j / height, correctly rounded, in three instructions each instead of the ~30 of an f64 division: with y = RN(1 / b) from the
        // host, q0 = RN(a y), r = a - q0 b (exact in an fma), q = RN(q0 + r y) == RN(a / b) (Markstein's correction step; checked for EVERY pixel index of
        // every resolution up to 16384 by tools/probe/div_markstein.c, tests/test_numerics_tables.py; the host only launches a PLAIN kernel below that)
        const double ai = (double)i, aj = (double)j, bw = (double)R.width, bh = (double)R.height;
        const double qi0 = ai * R.inv_width, qj0 = aj * R.inv_height;
        const double qi = __builtin_fma(__builtin_fma(-qi0, bw, ai), R.inv_width, qi0), qj = __builtin_fma(__builtin_fma(-qj0, bh, aj), R.inv_height, qj0);
        const double dx = (qi - 0.5) * 2.0;
        const double dy = -(qj - 0.5) * 2.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = m0[r] * dx + m1[r] * dy + R.m[8 + r] * -1.0 + R.m[12 + r] * 1.0;
        d3 eye = D3(h[0] / h[3], h[1] / h[3], h[2] / h[3]);
        d3 e0 = D3(R.eye[0], R.eye[1], R.eye[2]);
        ray.o = e0; ray.d = normalize(eye - e0); ray.refr = 1.0; ray.energy = 1.0f; ray.weight = 1.0f;
        ray.key = 0ULL; ray.pixel = pixel_out;
        return;
    }
    unsigned long long skey = 0;
    if (R.use_rng) { // keys are only ever consumed by AA jitter and area-light sampling
        unsigned long long pix = (unsigned long long)i + (unsigned long long)j * R.width;
        asm volatile("" : "+v"(pix)); // keeps the (sample-invariant) pixel hash from being hoisted out of this branch
        unsigned long long pkey = rng_hash(R.seed, pix);
        skey = rng_hash(pkey, s);
    }
    double h[4];
    double ox = (double)i, oy = (double)j;
    if (R.window_width != 0.0) {
        ox = ox + (rng_u01(skey, 0) - 0.5) * R.window_width;
        oy = oy + (rng_u01(skey, 1) - 0.5) * R.window_width;
    }
    double dx = (ox / (double)R.width - 0.5) * 2.0;
    double dy = -(oy / (double)R.height - 0.5) * 2.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = R.m[r] * dx + R.m[4 + r] * dy + R.m[8 + r] * -1.0 + R.m[12 + r] * 1.0;
    d3 eye = D3(h[0] / h[3], h[1] / h[3], h[2] / h[3]);
    d3 e0 = D3(R.eye[0], R.eye[1], R.eye[2]);
    ray.o = e0; ray.d = normalize(eye - e0); ray.refr = 1.0; ray.energy = 1.0f; ray.weight = 1.0f;
    ray.key = R.use_rng ? rng_hash(skey, kSaltPath) : 0ULL; ray.pixel = pixel_out;
}

} // namespace nrays
