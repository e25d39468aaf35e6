/*
 * nrays_oracle.c — CPU ORACLE for the nrays trace loop.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * library.  The product (nrays_amd/csrc, libnrays_hip.so) never links, loads or calls it.
 *
 * PARITY UNPINNED: the reference (sebcrozet/nrays, Rust) ships no tests, golden vectors or
 * images, cannot be built here (no cargo/rustc, un-vendored crates, F6) and its
 * intersection arithmetic lives in the third-party crates ncollide3d 0.16.x / nalgebra 0.15.x
 * (Cargo.toml:12-13, no Cargo.lock), whose sources are absent.  This file restates
 *   - nrays-owned arithmetic line by line (citations below are relative to /root/reference), and
 *   - the published ncollide3d/nalgebra algorithms as recalled in SURVEY.md Appendix B,
 * and is pinned only by analytic known-answer tests (tests/test_oracle_kat.py), by fixtures derived
 * independently of this file (60-digit membership bisection + support-map certificates,
 * tests/golden/make_kat_independent.py, tests/test_kat_independent.py) and by brute-force-vs-BVT
 * equivalence.  Explicit deviations:
 *   D-1  RNG is counter-based (the reference uses an OS-seeded thread RNG, scene.rs:75, light.rs:60);
 *   D-2  equal-toi ties are broken by smallest node index, then smallest triangle index
 *        (reference order is BinaryHeap dependent);
 *   D-3  cone / cylinder / capsule ray casts are closed-form (ncollide uses a GJK ray cast);
 *        for an origin inside a non-solid shape they return the exit point with the OUTWARD
 *        normal, as ncollide's reversed GJK re-cast does;
 *   D-4  recursion is capped at 64 generations (unbounded refraction recursion, scene.rs:246);
 *   D-6  bilinear taps are clamped to the last row/column (texture2d.rs:239-248 reads out of bounds);
 *   rotations are applied as 3x3 matrices built from the axis-angle (nalgebra applies the unit
 *        quaternion directly; identical for angle = 0, 1e-16 apart otherwise).
 *
 * Build: gcc -O3 -std=c11 -ffp-contract=off (Rust never fuses a*b+c; neither does this file).  No -march=native:
 * the library is built in one container and timed on another host (bench.py's cpu_baseline).
 */
#define _GNU_SOURCE
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/nrays_abi.h"

/* ------------------------------------------------------------------------------------------ */
/* small vector algebra (nalgebra Vector3<f64>, evaluation order as nalgebra's unrolled loops) */
/* ------------------------------------------------------------------------------------------ */
typedef struct { double x, y, z; } v3;

static inline v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, double s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdiv(v3 a, double s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline double vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 vcross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline double vnorm(v3 a) { return sqrt(vdot(a, a)); }
static inline v3 vnormalize(v3 a) { return vdiv(a, vnorm(a)); } /* na::normalize = v / |v| */
static inline double vget(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
static inline void vset(v3* a, int i, double v) { if (i == 0) a->x = v; else if (i == 1) a->y = v; else a->z = v; }

typedef struct { float x, y, z; } c3;    /* Vector3<f32> colour */
typedef struct { float x, y, z, w; } c4; /* Point4<f32> */
static inline c3 C3(float x, float y, float z) { c3 r = {x, y, z}; return r; }

/* ------------------------------------------------------------------------------------------ */
/* counter-based RNG (deviation D-1; shared specification with the HIP kernels) */
/* ------------------------------------------------------------------------------------------ */
static inline uint64_t rng_mix(uint64_t z) {
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 27; z *= 0x94d049bb133111ebULL;
    z ^= z >> 31;
    return z;
}
static inline uint64_t rng_hash(uint64_t key, uint64_t salt) {
    return rng_mix((key ^ (salt * 0x9E3779B97F4A7C15ULL)) + 0xD1B54A32D192ED03ULL);
}
static inline double rng_u01(uint64_t key, uint64_t dim) {
    return (double)(rng_hash(key, 0x1000ULL + dim) >> 11) * (1.0 / 9007199254740992.0);
}
#define RNG_SALT_PATH 2ULL
#define RNG_SALT_REFL 0x100ULL
#define RNG_SALT_REFR 0x101ULL
#define RNG_SALT_LIGHT 0x200ULL

/* ------------------------------------------------------------------------------------------ */
/* rays, AABBs                                                                                */
/* ------------------------------------------------------------------------------------------ */
typedef struct { v3 o, d; } Ray;
typedef struct { v3 mins, maxs; } AABB;

typedef struct {
    double toi;
    v3 normal;
    int has_uv;
    double u, v;
    int prim; /* triangle index inside a TriMesh (tie-break D-2), else 0 */
} Inter;

typedef struct {
    uint64_t node_tests, tri_tests, prim_tests, hit_records, tex_samples;
    uint64_t rays_primary, rays_reflection, rays_refraction, rays_shadow;
} Counters;

/* ncollide ray_aabb clip, as used by AABB::toi_with_ray(identity, ray, solid=true)
 * (called at src/scene.rs:276,309; SURVEY B-3).  Returns 1 and *toi (0 if the origin is inside). */
static int aabb_toi(const AABB* bv, const Ray* r, double* toi) {
    double tmax = DBL_MAX, tmin = -DBL_MAX;
    for (int i = 0; i < 3; ++i) {
        double d = vget(r->d, i), o = vget(r->o, i), mn = vget(bv->mins, i), mx = vget(bv->maxs, i);
        if (d == 0.0) {
            if (o < mn || o > mx) return 0;
        } else {
            double denom = 1.0 / d;
            double tn = (mn - o) * denom, tf = (mx - o) * denom;
            if (tn > tf) { double s = tn; tn = tf; tf = s; }
            if (tn > tmin) tmin = tn;
            if (tf < tmax) tmax = tf;
            if (tmax < 0.0 || tmin > tmax) return 0;
        }
    }
    *toi = tmin < 0.0 ? 0.0 : tmin;
    return 1;
}

static inline void aabb_merge(AABB* a, const AABB* b) {
    a->mins = V(fmin(a->mins.x, b->mins.x), fmin(a->mins.y, b->mins.y), fmin(a->mins.z, b->mins.z));
    a->maxs = V(fmax(a->maxs.x, b->maxs.x), fmax(a->maxs.y, b->maxs.y), fmax(a->maxs.z, b->maxs.z));
}
static inline v3 aabb_center(const AABB* a) { return vmul(vadd(a->mins, a->maxs), 0.5); }

/* ------------------------------------------------------------------------------------------ */
/* BVT: ncollide BVT::new_balanced (median partition, SURVEY B-1) + best_first_search (B-2)    */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    AABB bv;
    int left, right; /* children (internal) */
    int leaf;        /* leaf payload index or -1 */
} BvtNode;

typedef struct {
    BvtNode* nodes;
    int count, cap, root;
} Bvt;

static int bvt_push(Bvt* t, BvtNode n) {
    if (t->count == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 64;
        t->nodes = (BvtNode*)realloc(t->nodes, (size_t)t->cap * sizeof(BvtNode));
    }
    t->nodes[t->count] = n;
    return t->count++;
}

static int cmp_double(const void* a, const void* b) {
    double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}

/* `ids` are leaf payloads, `bvs[id]` their AABBs.  depth selects the axis (depth % 3). */
static int bvt_build_rec(Bvt* t, int* ids, int n, const AABB* bvs, int depth) {
    if (n == 1) {
        BvtNode leaf; leaf.bv = bvs[ids[0]]; leaf.left = leaf.right = -1; leaf.leaf = ids[0];
        return bvt_push(t, leaf);
    }
    int axis = depth % 3;
    double* med = (double*)malloc((size_t)n * sizeof(double));
    for (int i = 0; i < n; ++i) { v3 c = aabb_center(&bvs[ids[i]]); med[i] = vget(c, axis); }
    qsort(med, (size_t)n, sizeof(double), cmp_double);
    double median = (n % 2 == 0) ? (med[n / 2 - 1] + med[n / 2]) / 2.0 : med[n / 2];
    free(med);

    int* left = (int*)malloc((size_t)n * sizeof(int));
    int* right = (int*)malloc((size_t)n * sizeof(int));
    int nl = 0, nr = 0, insert_left = 0;
    AABB bb = bvs[ids[0]];
    for (int i = 0; i < n; ++i) {
        aabb_merge(&bb, &bvs[ids[i]]);
        v3 c = aabb_center(&bvs[ids[i]]);
        double pos = vget(c, axis);
        if (pos < median || (pos == median && insert_left)) { left[nl++] = ids[i]; insert_left = 0; }
        else { right[nr++] = ids[i]; insert_left = 1; }
    }
    if (nl == 0) left[nl++] = right[--nr];
    else if (nr == 0) right[nr++] = left[--nl];

    BvtNode in; in.bv = bb; in.left = in.right = -1; in.leaf = -1;
    int me = bvt_push(t, in);
    int l = bvt_build_rec(t, left, nl, bvs, depth + 1);
    int r = bvt_build_rec(t, right, nr, bvs, depth + 1);
    t->nodes[me].left = l; t->nodes[me].right = r;
    free(left); free(right);
    return me;
}

static void bvt_build(Bvt* t, int n, const AABB* bvs) {
    memset(t, 0, sizeof(*t));
    t->root = -1;
    if (n <= 0) return;
    int* ids = (int*)malloc((size_t)n * sizeof(int));
    for (int i = 0; i < n; ++i) ids[i] = i;
    t->root = bvt_build_rec(t, ids, n, bvs, 0);
    free(ids);
}

/* binary min-heap on cost (ncollide uses BinaryHeap keyed on -cost) */
typedef struct { double cost; int node; } HeapItem;
typedef struct { HeapItem* a; int n, cap; } Heap;

static void heap_push(Heap* h, double cost, int node) {
    if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 64; h->a = (HeapItem*)realloc(h->a, (size_t)h->cap * sizeof(HeapItem)); }
    int i = h->n++;
    while (i > 0) {
        int p = (i - 1) / 2;
        if (h->a[p].cost <= cost) break;
        h->a[i] = h->a[p]; i = p;
    }
    h->a[i].cost = cost; h->a[i].node = node;
}
static HeapItem heap_pop(Heap* h) {
    HeapItem top = h->a[0];
    HeapItem last = h->a[--h->n];
    int i = 0;
    for (;;) {
        int c = 2 * i + 1;
        if (c >= h->n) break;
        if (c + 1 < h->n && h->a[c + 1].cost < h->a[c].cost) c++;
        if (last.cost <= h->a[c].cost) break;
        h->a[i] = h->a[c]; i = c;
    }
    if (h->n > 0) h->a[i] = last;
    return top;
}

/* Cost-function interface of best_first_search (BVTCostFn, src/scene.rs:270-283,303-338). */
typedef struct CostFn {
    /* returns 1 and *cost if the bounding volume may contain a solution */
    int (*bv_cost)(struct CostFn*, const AABB*, double* cost);
    /* returns 1, *cost and fills *inter if the leaf is a candidate */
    int (*b_cost)(struct CostFn*, int leaf, double* cost, Inter* inter);
} CostFn;

/* ncollide best_first_search (SURVEY B-2) with the deterministic tie-break D-2: a candidate with
 * cost == best replaces the incumbent iff its leaf index is smaller.  The reference prunes a volume
 * when its cost >= best, which makes the winner among equal-toi candidates depend on BinaryHeap
 * order and on last-bit rounding of the AABB entry distance; here volumes are pruned only when
 * cost > best * (1 + 1e-12), so the result is exactly
 *     lexicographic min (toi, leaf index) over { leaves whose AABB test passes and whose cast hits },
 * a set that does not depend on the tree shape (the fp slab test is monotone under box inclusion). */
#define PRUNE_SLACK(best) ((best) < DBL_MAX / 2 ? (best) * (1.0 + 1e-12) : (best))
static int bvt_best_first(const Bvt* t, CostFn* fn, int* out_leaf, Inter* out_inter, double* out_cost) {
    if (t->root < 0) return 0;
    double best = DBL_MAX;
    int best_leaf = -1;
    Inter best_inter; memset(&best_inter, 0, sizeof(best_inter));
    Heap h = {0};
    double c;
    if (fn->bv_cost(fn, &t->nodes[t->root].bv, &c)) heap_push(&h, c, t->root);
    while (h.n > 0) {
        HeapItem it = heap_pop(&h);
        if (it.cost > PRUNE_SLACK(best)) break;
        const BvtNode* nd = &t->nodes[it.node];
        if (nd->leaf < 0) {
            int ch[2] = {nd->left, nd->right};
            for (int k = 0; k < 2; ++k)
                if (fn->bv_cost(fn, &t->nodes[ch[k]].bv, &c) && c <= PRUNE_SLACK(best)) heap_push(&h, c, ch[k]);
        } else {
            Inter in; double cost;
            if (fn->b_cost(fn, nd->leaf, &cost, &in)) {
                if (cost < best || (cost == best && best_leaf >= 0 && nd->leaf < best_leaf)) {
                    best = cost; best_leaf = nd->leaf; best_inter = in;
                }
            }
        }
    }
    free(h.a);
    if (best_leaf < 0) return 0;
    *out_leaf = best_leaf; *out_inter = best_inter; if (out_cost) *out_cost = best;
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* Isometry3 (translation + rotation), built as Isometry3::new(t, axis_angle) loader3d.rs:552  */
/* ------------------------------------------------------------------------------------------ */
typedef struct { double m[3][3]; v3 t; } Iso; /* m = rotation matrix R (local -> world) */

static double (*volatile libm_sin)(double) = sin;
static double (*volatile libm_cos)(double) = cos;
static void iso_from_axis_angle(Iso* iso, const double t[3], const double w[3]) {
    double angle = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    iso->t = V(t[0], t[1], t[2]);
    if (angle == 0.0) {
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) iso->m[i][j] = i == j ? 1.0 : 0.0;
        return;
    }
    /* unit quaternion (w, i, j, k) = (cos(a/2), axis*sin(a/2)), then its rotation matrix */
    /* nalgebra: (angle / 2).sin_cos() = two separate libm calls; a compiler that merges them into sincos() gets
     * a different last bit for ~0.16 % of the angles (glibc), hence the volatile function pointers */
    double s = libm_sin(angle / 2.0), qw = libm_cos(angle / 2.0);
    double qi = w[0] / angle * s, qj = w[1] / angle * s, qk = w[2] / angle * s;
    double ww = qw * qw, ii = qi * qi, jj = qj * qj, kk = qk * qk;
    double ij = qi * qj * 2.0, wk = qw * qk * 2.0, wj = qw * qj * 2.0, ik = qi * qk * 2.0, jk = qj * qk * 2.0, wi = qw * qi * 2.0;
    iso->m[0][0] = ww + ii - jj - kk; iso->m[0][1] = ij - wk;           iso->m[0][2] = wj + ik;
    iso->m[1][0] = wk + ij;           iso->m[1][1] = ww - ii + jj - kk; iso->m[1][2] = jk - wi;
    iso->m[2][0] = ik - wj;           iso->m[2][1] = wi + jk;           iso->m[2][2] = ww - ii - jj + kk;
}
static inline v3 iso_rot(const Iso* s, v3 v) { /* R v */
    return V(s->m[0][0] * v.x + s->m[0][1] * v.y + s->m[0][2] * v.z,
             s->m[1][0] * v.x + s->m[1][1] * v.y + s->m[1][2] * v.z,
             s->m[2][0] * v.x + s->m[2][1] * v.y + s->m[2][2] * v.z);
}
static inline v3 iso_inv_rot(const Iso* s, v3 v) { /* R^T v */
    return V(s->m[0][0] * v.x + s->m[1][0] * v.y + s->m[2][0] * v.z,
             s->m[0][1] * v.x + s->m[1][1] * v.y + s->m[2][1] * v.z,
             s->m[0][2] * v.x + s->m[1][2] * v.y + s->m[2][2] * v.z);
}
static inline Ray iso_inv_ray(const Iso* s, const Ray* r) { /* ray.inverse_transform_by(m) */
    Ray l; l.o = iso_inv_rot(s, vsub(r->o, s->t)); l.d = iso_inv_rot(s, r->d); return l;
}
static inline v3 iso_abs_rot(const Iso* s, v3 v) { /* m.absolute_transform_vector */
    return V(fabs(s->m[0][0]) * v.x + fabs(s->m[0][1]) * v.y + fabs(s->m[0][2]) * v.z,
             fabs(s->m[1][0]) * v.x + fabs(s->m[1][1]) * v.y + fabs(s->m[1][2]) * v.z,
             fabs(s->m[2][0]) * v.x + fabs(s->m[2][1]) * v.y + fabs(s->m[2][2]) * v.z);
}

/* ------------------------------------------------------------------------------------------ */
/* primitive ray casts (SURVEY Appendix B-4 .. B-8; cone/cylinder/capsule: deviation D-3)       */
/* ------------------------------------------------------------------------------------------ */
#define PI_D 3.14159265358979323846

/* B-4 Ball(r) at the isometry's translation (rotation ignored). */
static int cast_ball(double radius, v3 center, const Ray* ray, int solid, Inter* out) {
    v3 dc = vsub(ray->o, center);
    double a = vdot(ray->d, ray->d), b = vdot(dc, ray->d), c = vdot(dc, dc) - radius * radius;
    if (c > 0.0 && b > 0.0) return 0;
    double delta = b * b - a * c;
    if (delta < 0.0) return 0;
    double sq = sqrt(delta);
    double t = (-b - sq) / a;
    int inside = 0;
    if (t <= 0.0) { inside = 1; t = solid ? 0.0 : (-b + sq) / a; }
    v3 pos = vsub(vadd(ray->o, vmul(ray->d, t)), center);
    v3 n = vnormalize(pos);
    out->toi = t;
    out->has_uv = 1; /* ball_uv of the outward normal */
    out->u = 0.5 + atan2(n.z, n.x) / (PI_D * 2.0);
    out->v = 0.5 - asin(n.y) / PI_D;
    out->normal = inside ? vneg(n) : n;
    out->prim = 0;
    return 1;
}

/* B-5 Cuboid(half_extents): ncollide ray_aabb on [-he, he] in local space, with face ids. */
static int cast_cuboid(v3 he, const Iso* iso, const Ray* ray, int solid, Inter* out) {
    Ray l = iso_inv_ray(iso, ray);
    double tmax = DBL_MAX, tmin = -DBL_MAX;
    int near_side = 0, far_side = 0;
    for (int i = 0; i < 3; ++i) {
        double d = vget(l.d, i), o = vget(l.o, i), mn = -vget(he, i), mx = vget(he, i);
        if (d == 0.0) {
            if (o < mn || o > mx) return 0;
        } else {
            double denom = 1.0 / d;
            double tn = (mn - o) * denom, tf = (mx - o) * denom;
            int flip = 0;
            if (tn > tf) { double s = tn; tn = tf; tf = s; flip = 1; }
            if (tn > tmin) { tmin = tn; near_side = flip ? -(i + 1) : (i + 1); }
            if (tf < tmax) { tmax = tf; far_side = flip ? (i + 1) : -(i + 1); }
            if (tmax < 0.0 || tmin > tmax) return 0;
        }
    }
    v3 n = V(0, 0, 0);
    double t; int side;
    if (tmin < 0.0) { /* origin inside */
        side = far_side;
        if (solid) t = 0.0;
        else {
            t = tmax;
            if (far_side < 0) vset(&n, -far_side - 1, -1.0); else if (far_side > 0) vset(&n, far_side - 1, 1.0);
        }
    } else {
        t = tmin; side = near_side;
        if (near_side < 0) vset(&n, -near_side - 1, 1.0); else if (near_side > 0) vset(&n, near_side - 1, -1.0);
    }
    v3 pt = vadd(l.o, vmul(l.d, t));
    v3 dpt = vsub(pt, vneg(he));
    v3 scale = vsub(he, vneg(he));
    int id = side < 0 ? -side : side;
    out->toi = t; out->normal = iso_rot(iso, n); out->has_uv = 1; out->prim = 0;
    if (id == 1) { out->u = dpt.y / scale.y; out->v = dpt.z / scale.z; }
    else if (id == 2) { out->u = dpt.z / scale.z; out->v = dpt.x / scale.x; }
    else { out->u = dpt.x / scale.x; out->v = dpt.y / scale.y; }
    return 1;
}

/* B-6 Plane(n) through the local origin. */
static int cast_plane(v3 pn, const Iso* iso, const Ray* ray, int solid, Inter* out) {
    Ray l = iso_inv_ray(iso, ray);
    double dot_normal_dpos = vdot(pn, vneg(l.o));
    out->has_uv = 0; out->prim = 0;
    if (solid && dot_normal_dpos > 0.0) { out->toi = 0.0; out->normal = V(0, 0, 0); return 1; }
    double denom = vdot(pn, l.d);
    if (denom == 0.0) return 0;
    double t = dot_normal_dpos / denom;
    if (t >= 0.0) {
        v3 n = dot_normal_dpos > 0.0 ? vneg(pn) : pn;
        out->toi = t; out->normal = iso_rot(iso, n);
        return 1;
    }
    return 0;
}

/* Shared tail of the closed-form convex casts: [t0,t1] is the parametric interval of the ray
 * inside the solid, n0/n1 the OUTWARD normals where it enters/leaves. */
static int convex_interval_hit(double t0, double t1, v3 n0, v3 n1, const Ray* l, const Iso* iso, int solid, Inter* out) {
    if (!(t0 <= t1) || t1 < 0.0) return 0;
    out->has_uv = 0; out->prim = 0;
    if (t0 > 0.0) { out->toi = t0; out->normal = iso_rot(iso, n0); return 1; }
    if (solid) { out->toi = 0.0; out->normal = iso_rot(iso, vneg(l->d)); return 1; }
    out->toi = t1; out->normal = iso_rot(iso, n1);
    return 1;
}

/* D-3 Cylinder(half_height, radius), axis = local Y. */
static int cast_cylinder(double hh, double r, const Iso* iso, const Ray* ray, int solid, Inter* out) {
    Ray l = iso_inv_ray(iso, ray);
    double t0 = -DBL_MAX, t1 = DBL_MAX;
    int enter_side = 1, exit_side = 1; /* 1 = lateral surface, 0 = cap */
    double A = l.d.x * l.d.x + l.d.z * l.d.z;
    double B = l.o.x * l.d.x + l.o.z * l.d.z;
    double C = l.o.x * l.o.x + l.o.z * l.o.z - r * r;
    if (A == 0.0) { if (C > 0.0) return 0; }
    else {
        double disc = B * B - A * C;
        if (disc < 0.0) return 0;
        double sq = sqrt(disc);
        t0 = (-B - sq) / A; t1 = (-B + sq) / A;
    }
    if (l.d.y == 0.0) { if (l.o.y < -hh || l.o.y > hh) return 0; }
    else {
        double ta = (-hh - l.o.y) / l.d.y, tb = (hh - l.o.y) / l.d.y;
        if (ta > tb) { double s = ta; ta = tb; tb = s; }
        if (ta > t0) { t0 = ta; enter_side = 0; }
        if (tb < t1) { t1 = tb; exit_side = 0; }
    }
    if (!(t0 <= t1) || t1 < 0.0) return 0;
    v3 n0, n1;
    if (enter_side) { v3 p = vadd(l.o, vmul(l.d, t0)); double s = sqrt(p.x * p.x + p.z * p.z); n0 = V(p.x / s, 0.0, p.z / s); }
    else n0 = V(0.0, l.d.y > 0.0 ? -1.0 : 1.0, 0.0);
    if (exit_side) { v3 p = vadd(l.o, vmul(l.d, t1)); double s = sqrt(p.x * p.x + p.z * p.z); n1 = V(p.x / s, 0.0, p.z / s); }
    else n1 = V(0.0, l.d.y > 0.0 ? 1.0 : -1.0, 0.0);
    return convex_interval_hit(t0, t1, n0, n1, &l, iso, solid, out);
}

/* D-3 Cone(half_height, radius): apex at (0,+hh,0), base disc at y = -hh. */
static v3 cone_side_normal(v3 p, double hh, double k2) {
    v3 g = V(p.x, k2 * (hh - p.y), p.z);
    double s = vnorm(g);
    if (s == 0.0) return V(0.0, 1.0, 0.0);
    return vdiv(g, s);
}
static int cast_cone(double hh, double r, const Iso* iso, const Ray* ray, int solid, Inter* out) {
    Ray l = iso_inv_ray(iso, ray);
    double k = r / (2.0 * hh), k2 = k * k;
    double ow = hh - l.o.y, dw = -l.d.y; /* w(t) = ow + t dw = depth below the apex, in [0, 2hh] */
    double s0 = -DBL_MAX, s1 = DBL_MAX;
    int s0_kind = 0, s1_kind = 0; /* which slab plane bounds: 1 = apex plane (w=0), 2 = base (w=2hh) */
    if (dw == 0.0) { if (ow < 0.0 || ow > 2.0 * hh) return 0; }
    else {
        double ta = (0.0 - ow) / dw, tb = (2.0 * hh - ow) / dw;
        if (ta <= tb) { s0 = ta; s0_kind = 1; s1 = tb; s1_kind = 2; }
        else { s0 = tb; s0_kind = 2; s1 = ta; s1_kind = 1; }
    }
    double A = l.d.x * l.d.x + l.d.z * l.d.z - k2 * dw * dw;
    double B = l.o.x * l.d.x + l.o.z * l.d.z - k2 * ow * dw;
    double C = l.o.x * l.o.x + l.o.z * l.o.z - k2 * ow * ow;
    double t0 = s0, t1 = s1;
    int k0 = s0_kind, k1 = s1_kind; /* 0 = lateral surface */
    if (A > 0.0) {
        double disc = B * B - A * C;
        if (disc < 0.0) return 0;
        double sq = sqrt(disc);
        double ra = (-B - sq) / A, rb = (-B + sq) / A;
        if (ra > t0) { t0 = ra; k0 = 0; }
        if (rb < t1) { t1 = rb; k1 = 0; }
    } else if (A < 0.0) {
        double disc = B * B - A * C;
        if (disc > 0.0) {
            double sq = sqrt(disc);
            double lo = (-B + sq) / A, hi = (-B - sq) / A; /* lo < hi; inside the double cone for t<=lo or t>=hi */
            double a1 = hi > s0 ? hi : s0; /* candidate [max(s0,hi), s1] */
            double b0 = lo < s1 ? lo : s1; /* candidate [s0, min(s1,lo)] */
            if (a1 <= s1) { if (hi > s0) { t0 = hi; k0 = 0; } }
            else if (s0 <= b0) { if (lo < s1) { t1 = lo; k1 = 0; } }
            else return 0;
        }
    } else {
        if (B == 0.0) { if (C > 0.0) return 0; }
        else {
            double ts = -C / (2.0 * B);
            if (B > 0.0) { if (ts < t1) { t1 = ts; k1 = 0; } }
            else { if (ts > t0) { t0 = ts; k0 = 0; } }
        }
    }
    if (!(t0 <= t1) || t1 < 0.0) return 0;
    v3 n0, n1;
    if (k0 == 0) n0 = cone_side_normal(vadd(l.o, vmul(l.d, t0)), hh, k2); else n0 = V(0.0, k0 == 1 ? 1.0 : -1.0, 0.0);
    if (k1 == 0) n1 = cone_side_normal(vadd(l.o, vmul(l.d, t1)), hh, k2); else n1 = V(0.0, k1 == 1 ? 1.0 : -1.0, 0.0);
    return convex_interval_hit(t0, t1, n0, n1, &l, iso, solid, out);
}

/* D-3 Capsule(half_height, radius): segment [-hh,hh] on Y swept by a ball. */
static int cast_capsule(double hh, double r, const Iso* iso, const Ray* ray, int solid, Inter* out) {
    Ray l = iso_inv_ray(iso, ray);
    double t0 = DBL_MAX, t1 = -DBL_MAX;
    v3 n0 = V(0, 0, 0), n1 = V(0, 0, 0);
    /* lateral part: infinite cylinder clipped to |y| <= hh */
    {
        double a0 = -DBL_MAX, a1 = DBL_MAX; int ok = 1;
        double A = l.d.x * l.d.x + l.d.z * l.d.z;
        double B = l.o.x * l.d.x + l.o.z * l.d.z;
        double C = l.o.x * l.o.x + l.o.z * l.o.z - r * r;
        if (A == 0.0) { if (C > 0.0) ok = 0; }
        else {
            double disc = B * B - A * C;
            if (disc < 0.0) ok = 0;
            else { double sq = sqrt(disc); a0 = (-B - sq) / A; a1 = (-B + sq) / A; }
        }
        int e_side = 1, x_side = 1;
        if (ok) {
            if (l.d.y == 0.0) { if (l.o.y < -hh || l.o.y > hh) ok = 0; }
            else {
                double ta = (-hh - l.o.y) / l.d.y, tb = (hh - l.o.y) / l.d.y;
                if (ta > tb) { double s = ta; ta = tb; tb = s; }
                if (ta > a0) { a0 = ta; e_side = 0; }
                if (tb < a1) { a1 = tb; x_side = 0; }
            }
        }
        if (ok && a0 <= a1) {
            t0 = a0; t1 = a1;
            if (e_side) { v3 p = vadd(l.o, vmul(l.d, a0)); n0 = V(p.x / r, 0.0, p.z / r); }
            if (x_side) { v3 p = vadd(l.o, vmul(l.d, a1)); n1 = V(p.x / r, 0.0, p.z / r); }
        }
    }
    for (int s = 0; s < 2; ++s) { /* the two end balls; they own the bound on a strict improvement */
        v3 c = V(0.0, s == 0 ? -hh : hh, 0.0);
        v3 dc = vsub(l.o, c);
        double a = vdot(l.d, l.d), b = vdot(dc, l.d), cc = vdot(dc, dc) - r * r;
        double delta = b * b - a * cc;
        if (delta < 0.0) continue;
        double sq = sqrt(delta);
        double b0 = (-b - sq) / a, b1 = (-b + sq) / a;
        if (b0 < t0) { t0 = b0; n0 = vdiv(vsub(vadd(l.o, vmul(l.d, b0)), c), r); }
        if (b1 > t1) { t1 = b1; n1 = vdiv(vsub(vadd(l.o, vmul(l.d, b1)), c), r); }
    }
    return convex_interval_hit(t0, t1, n0, n1, &l, iso, solid, out);
}

/* B-8 ncollide triangle_ray_intersection.  bary = (1-v-w, v, w). */
static int cast_triangle(v3 a, v3 b, v3 c, const Ray* ray, double* toi, v3* normal, double bary[3]) {
    v3 ab = vsub(b, a), ac = vsub(c, a);
    v3 n = vcross(ab, ac);
    double d = vdot(n, ray->d);
    if (d == 0.0) return 0;
    v3 ap = vsub(ray->o, a);
    double t = vdot(ap, n);
    if ((t < 0.0 && d < 0.0) || (t > 0.0 && d > 0.0)) return 0;
    double dabs = fabs(d);
    v3 e = vneg(vcross(ray->d, ap));
    double v, w, invd;
    if (t < 0.0) {
        v = -vdot(ac, e);
        if (v < 0.0 || v > dabs) return 0;
        w = vdot(ab, e);
        if (w < 0.0 || v + w > dabs) return 0;
        invd = 1.0 / dabs;
        *toi = -t * invd;
        *normal = vneg(vnormalize(n));
    } else {
        v = vdot(ac, e);
        if (v < 0.0 || v > dabs) return 0;
        w = -vdot(ab, e);
        if (w < 0.0 || v + w > dabs) return 0;
        invd = 1.0 / dabs;
        *toi = t * invd;
        *normal = vnormalize(n);
    }
    v = v * invd; w = w * invd;
    bary[0] = -v - w + 1.0; bary[1] = v; bary[2] = w;
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* scene model                                                                                 */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int nverts, ntris;
    const double* verts; /* borrowed from the descriptor */
    const double* uvs;
    const uint32_t* idx;
    Bvt bvt;     /* B-9: TriMesh's own BVT over per-triangle AABBs */
    AABB aabb;   /* local-space AABB (root BV) */
} OMesh;

typedef struct {
    NraysNode d;
    Iso iso;
    AABB aabb; /* world AABB = geometry.bounding_volume(&transform), scene_node.rs:41 */
    const OMesh* mesh;
} ONode;

typedef struct {
    const NraysSceneDesc* desc;
    OMesh* meshes;
    ONode* nodes;
    int nnodes;
    Bvt world; /* scene.rs:126 */
    c3 background;
} OScene;

static inline v3 mesh_vert(const OMesh* m, uint32_t i) { return V(m->verts[3 * i], m->verts[3 * i + 1], m->verts[3 * i + 2]); }

/* TriMesh ray cast = nested best-first search (B-9). */
typedef struct {
    CostFn base;
    const OMesh* mesh;
    Ray ray; /* local space */
    Counters* cnt;
} MeshCostFn;

static int mesh_bv_cost(CostFn* f, const AABB* bv, double* cost) {
    MeshCostFn* m = (MeshCostFn*)f;
    m->cnt->node_tests++;
    return aabb_toi(bv, &m->ray, cost);
}
static int mesh_b_cost(CostFn* f, int leaf, double* cost, Inter* in) {
    MeshCostFn* m = (MeshCostFn*)f;
    const OMesh* me = m->mesh;
    m->cnt->tri_tests++;
    uint32_t i0 = me->idx[3 * leaf], i1 = me->idx[3 * leaf + 1], i2 = me->idx[3 * leaf + 2];
    double bary[3];
    if (!cast_triangle(mesh_vert(me, i0), mesh_vert(me, i1), mesh_vert(me, i2), &m->ray, &in->toi, &in->normal, bary)) return 0;
    in->prim = leaf;
    if (me->uvs) {
        in->has_uv = 1;
        in->u = me->uvs[2 * i0] * bary[0] + me->uvs[2 * i1] * bary[1] + me->uvs[2 * i2] * bary[2];
        in->v = me->uvs[2 * i0 + 1] * bary[0] + me->uvs[2 * i1 + 1] * bary[1] + me->uvs[2 * i2 + 1] * bary[2];
    } else in->has_uv = 0;
    *cost = in->toi;
    return 1;
}
static int cast_trimesh(const OMesh* mesh, const Iso* iso, const Ray* ray, Inter* out, Counters* cnt) {
    MeshCostFn fn; fn.base.bv_cost = mesh_bv_cost; fn.base.b_cost = mesh_b_cost;
    fn.mesh = mesh; fn.ray = iso_inv_ray(iso, ray); fn.cnt = cnt;
    int leaf;
    if (!bvt_best_first(&mesh->bvt, &fn.base, &leaf, out, NULL)) return 0;
    out->normal = iso_rot(iso, out->normal);
    return 1;
}

/* SceneNode::cast (src/scene_node.rs:51-54; the nmap branch :60-74 is never constructed). */
static int node_cast(const ONode* n, const Ray* ray, Inter* out, Counters* cnt) {
    int solid = n->d.solid != 0;
    const double* p = n->d.params;
    if (n->d.shape_kind != NRAYS_SHAPE_TRIMESH) cnt->prim_tests++;
    switch (n->d.shape_kind) {
    case NRAYS_SHAPE_BALL: return cast_ball(p[0], n->iso.t, ray, solid, out);
    case NRAYS_SHAPE_CUBOID: return cast_cuboid(V(p[0], p[1], p[2]), &n->iso, ray, solid, out);
    case NRAYS_SHAPE_CYLINDER: return cast_cylinder(p[0], p[1], &n->iso, ray, solid, out);
    case NRAYS_SHAPE_CAPSULE: return cast_capsule(p[0], p[1], &n->iso, ray, solid, out);
    case NRAYS_SHAPE_CONE: return cast_cone(p[0], p[1], &n->iso, ray, solid, out);
    case NRAYS_SHAPE_PLANE: return cast_plane(V(p[0], p[1], p[2]), &n->iso, ray, solid, out);
    case NRAYS_SHAPE_TRIMESH: return cast_trimesh(n->mesh, &n->iso, ray, out, cnt);
    default: return 0;
    }
}

/* World AABBs: ncollide HasBoundingVolume<AABB> of each shape under the isometry. */
static v3 support_local(const NraysNode* d, v3 dir) {
    double hh = d->params[0], r = d->params[1];
    v3 res = V(dir.x, 0.0, dir.z);
    double n = vnorm(res);
    if (d->shape_kind == NRAYS_SHAPE_CYLINDER) {
        res = n == 0.0 ? V(0, 0, 0) : vmul(vdiv(res, n), r);
        res.y = copysign(hh, dir.y);
        return res;
    }
    if (d->shape_kind == NRAYS_SHAPE_CONE) {
        if (n == 0.0) return V(0.0, copysign(hh, dir.y), 0.0);
        res = vmul(vdiv(res, n), r); res.y = -hh;
        if (vdot(dir, res) < dir.y * hh) return V(0.0, hh, 0.0);
        return res;
    }
    /* capsule */
    return vadd(V(0.0, copysign(hh, dir.y), 0.0), vmul(dir, r));
}
static AABB node_world_aabb(const ONode* n) {
    AABB bb;
    const NraysNode* d = &n->d;
    switch (d->shape_kind) {
    case NRAYS_SHAPE_BALL: {
        v3 r = V(d->params[0], d->params[0], d->params[0]);
        bb.mins = vsub(n->iso.t, r); bb.maxs = vadd(n->iso.t, r); break;
    }
    case NRAYS_SHAPE_CUBOID: {
        v3 h = iso_abs_rot(&n->iso, V(d->params[0], d->params[1], d->params[2]));
        bb.mins = vsub(n->iso.t, h); bb.maxs = vadd(n->iso.t, h); break;
    }
    case NRAYS_SHAPE_PLANE:
        bb.mins = V(-DBL_MAX, -DBL_MAX, -DBL_MAX); bb.maxs = V(DBL_MAX, DBL_MAX, DBL_MAX); break;
    case NRAYS_SHAPE_TRIMESH: {
        v3 c = aabb_center(&n->mesh->aabb);
        v3 h = vmul(vsub(n->mesh->aabb.maxs, n->mesh->aabb.mins), 0.5);
        v3 wc = vadd(iso_rot(&n->iso, c), n->iso.t);
        v3 wh = iso_abs_rot(&n->iso, h);
        bb.mins = vsub(wc, wh); bb.maxs = vadd(wc, wh); break;
    }
    default: { /* support-mapped shapes: implicit_shape_aabb */
        for (int i = 0; i < 3; ++i) {
            v3 e = V(i == 0, i == 1, i == 2);
            v3 ld = iso_inv_rot(&n->iso, e);
            v3 sp = vadd(iso_rot(&n->iso, support_local(d, ld)), n->iso.t);
            v3 sn = vadd(iso_rot(&n->iso, support_local(d, vneg(ld))), n->iso.t);
            vset(&bb.maxs, i, vget(sp, i)); vset(&bb.mins, i, vget(sn, i));
        }
        break;
    }
    }
    return bb;
}

/* ------------------------------------------------------------------------------------------ */
/* textures and materials                                                                      */
/* ------------------------------------------------------------------------------------------ */
static inline c4 tex_at(const NraysTexture* t, uint32_t x, uint32_t y) { /* texture2d.rs:203-205 */
    size_t i = (size_t)y * t->width + x;
    c4 r;
    if (t->format == NRAYS_TEXEL_RGBA8) {
        const uint8_t* p = (const uint8_t*)t->texels + 4 * i;
        r.x = (float)p[0] / 255.0f; r.y = (float)p[1] / 255.0f; r.z = (float)p[2] / 255.0f; r.w = (float)p[3] / 255.0f;
    } else {
        const float* p = (const float*)t->texels + 4 * i;
        r.x = p[0]; r.y = p[1]; r.z = p[2]; r.w = p[3];
    }
    return r;
}
/* Texture2d::sample, src/texture2d.rs:207-256 (taps clamped: D-6). */
static c4 tex_sample(const NraysTexture* t, double u, double v, Counters* cnt) {
    cnt->tex_samples++;
    float ux = (float)u, uy = (float)v;
    if (t->overflow == NRAYS_OVERFLOW_CLAMP) {
        ux = ux < 0.0f ? 0.0f : (ux > 1.0f ? 1.0f : ux);
        uy = uy < 0.0f ? 0.0f : (uy > 1.0f ? 1.0f : uy);
    } else {
        ux = fmodf(ux, 1.0f); uy = fmodf(uy, 1.0f);
        if (ux < 0.0f) ux = 1.0f + ux;
        if (uy < 0.0f) uy = 1.0f + uy;
    }
    ux = ux * (float)(t->width - 1);
    uy = uy * (float)(t->height - 1);
    uint32_t wm = t->width - 1, hm = t->height - 1;
    if (t->interp == NRAYS_INTERP_NEAREST) {
        uint32_t x = (uint32_t)roundf(ux), y = (uint32_t)roundf(uy);
        if (x > wm) x = wm; if (y > hm) y = hm;
        return tex_at(t, x, y);
    }
    uint32_t lx = (uint32_t)floorf(ux), ly = (uint32_t)floorf(uy);
    if (lx > wm) lx = wm; if (ly > hm) ly = hm;
    uint32_t hx = lx + 1, hy = ly + 1;
    float sx = ux - (float)lx, sy = uy - (float)ly;
    if (hx > wm) hx = wm; if (hy > hm) hy = hm;
    c4 ul = tex_at(t, lx, hy), ur = tex_at(t, hx, hy), dr = tex_at(t, hx, ly), dl = tex_at(t, lx, ly);
    c4 ui, di, r;
    ui.x = ul.x * (1.0f - sx) + ur.x * sx; ui.y = ul.y * (1.0f - sx) + ur.y * sx; ui.z = ul.z * (1.0f - sx) + ur.z * sx; ui.w = ul.w * (1.0f - sx) + ur.w * sx;
    di.x = dl.x * (1.0f - sx) + dr.x * sx; di.y = dl.y * (1.0f - sx) + dr.y * sx; di.z = dl.z * (1.0f - sx) + dr.z * sx; di.w = dl.w * (1.0f - sx) + dr.w * sx;
    r.x = ui.x * sy + di.x * (1.0f - sy); r.y = ui.y * sy + di.y * (1.0f - sy); r.z = ui.z * sy + di.z * (1.0f - sy); r.w = ui.w * sy + di.w * (1.0f - sy);
    return r;
}

/* Material::ambiant for the three implementations (phong_material.rs:39-70,
 * normal_material.rs:9-14, uv_material.rs:10-20). */
static c4 material_ambiant(const OScene* s, const NraysMaterial* m, const Inter* in, Counters* cnt) {
    c4 r;
    if (m->kind == NRAYS_MAT_NORMAL) {
        r.x = (1.0f + (float)in->normal.x) / 2.0f; r.y = (1.0f + (float)in->normal.y) / 2.0f;
        r.z = (1.0f + (float)in->normal.z) / 2.0f; r.w = 1.0f;
        return r;
    }
    if (m->kind == NRAYS_MAT_UV) {
        if (in->has_uv) { r.x = (float)in->u; r.y = (float)in->v; r.z = 0.0f; r.w = 1.0f; }
        else { r.x = r.y = r.z = r.w = 0.0f; }
        return r;
    }
    if (in->has_uv) {
        c4 tc = {1.0f, 1.0f, 1.0f, 1.0f};
        if (m->texture_id >= 0) { tc = tex_sample(&s->desc->textures[m->texture_id], in->u, in->v, cnt); tc.w = 1.0f; }
        if (m->alpha_texture_id >= 0) tc.w = tex_sample(&s->desc->textures[m->alpha_texture_id], in->u, in->v, cnt).w;
        r.x = m->ambiant[0] * tc.x; r.y = m->ambiant[1] * tc.y; r.z = m->ambiant[2] * tc.z; r.w = 1.0f * tc.w;
    } else { r.x = m->ambiant[0]; r.y = m->ambiant[1]; r.z = m->ambiant[2]; r.w = 1.0f; }
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* Scene::intersects_ray + TransparentShadowsRayTOICostFn (src/scene.rs:147-161,285-339)        */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    CostFn base;
    const OScene* scene;
    Ray ray;
    double maxtoi;
    c3 filter;
    Counters* cnt;
} ShadowCostFn;

static int world_bv_cost_shadow(CostFn* f, const AABB* bv, double* cost) {
    ShadowCostFn* s = (ShadowCostFn*)f;
    s->cnt->node_tests++;
    return aabb_toi(bv, &s->ray, cost);
}
static int world_b_cost_shadow(CostFn* f, int leaf, double* cost, Inter* in) {
    ShadowCostFn* s = (ShadowCostFn*)f;
    const ONode* n = &s->scene->nodes[leaf];
    if (!node_cast(n, &s->ray, in, s->cnt)) return 0;
    if (in->toi <= s->maxtoi) {
        s->cnt->hit_records++;
        c4 color = material_ambiant(s->scene, &s->scene->desc->materials[n->d.material_id], in, s->cnt);
        float alpha = color.w * n->d.alpha;
        if (alpha < 1.0f) {
            s->filter.x = (s->filter.x * color.x) * (1.0f - alpha);
            s->filter.y = (s->filter.y * color.y) * (1.0f - alpha);
            s->filter.z = (s->filter.z * color.z) * (1.0f - alpha);
            return 0;
        }
        *cost = in->toi;
        return 1;
    }
    return 0;
}
/* returns 1 (lit) and the colour filter, or 0 (an opaque node blocks the ray) */
static int scene_intersects_ray(const OScene* sc, const Ray* ray, double maxtoi, c3* filter, Counters* cnt) {
    ShadowCostFn fn; fn.base.bv_cost = world_bv_cost_shadow; fn.base.b_cost = world_b_cost_shadow;
    fn.scene = sc; fn.ray = *ray; fn.maxtoi = maxtoi; fn.filter = C3(1.0f, 1.0f, 1.0f); fn.cnt = cnt;
    cnt->rays_shadow++;
    int leaf; Inter in;
    int blocked = bvt_best_first(&sc->world, &fn.base, &leaf, &in, NULL);
    *filter = fn.filter;
    return !blocked;
}

/* ------------------------------------------------------------------------------------------ */
/* PhongMaterial::compute (src/phong_material.rs:72-151) and Light::sample (src/light.rs:57-63) */
/* ------------------------------------------------------------------------------------------ */
typedef struct { Ray ray; double refr; float energy; uint64_t key; } RayWE; /* ray_with_energy.rs:4-8 */

static c4 material_compute(const OScene* s, const NraysMaterial* m, const RayWE* ray, v3 point, const Inter* in, Counters* cnt) {
    if (m->kind != NRAYS_MAT_PHONG) return material_ambiant(s, m, in, cnt); /* material.rs:8-16 */
    c4 tex = {1.0f, 1.0f, 1.0f, 1.0f};
    float alpha = 1.0f;
    if (in->has_uv && m->texture_id >= 0) tex = tex_sample(&s->desc->textures[m->texture_id], in->u, in->v, cnt);
    if (in->has_uv && m->alpha_texture_id >= 0) alpha = tex_sample(&s->desc->textures[m->alpha_texture_id], in->u, in->v, cnt).w;
    c3 res = C3(m->ambiant[0] * tex.x, m->ambiant[1] * tex.y, m->ambiant[2] * tex.z);
    v3 normal = in->normal;
    for (uint32_t li = 0; li < s->desc->num_lights; ++li) {
        const NraysLight* light = &s->desc->lights[li];
        c3 acc = C3(0.0f, 0.0f, 0.0f);
        uint32_t ns = light->racsample * light->racsample;
        uint64_t lkey = rng_hash(ray->key, RNG_SALT_LIGHT + li);
        for (uint32_t k = 0; k < ns; ++k) {
            v3 pos = V(light->pos[0], light->pos[1], light->pos[2]);
            if (light->radius != 0.0) {
                uint64_t sk = rng_hash(lkey, k);
                v3 rnd = V(rng_u01(sk, 0), rng_u01(sk, 1), rng_u01(sk, 2));
                pos = vadd(pos, vmul(rnd, light->radius));
            }
            v3 ldir = vsub(pos, point);
            double norm = vnorm(ldir);
            ldir = vdiv(ldir, norm);
            double dist = norm - 0.001;
            Ray sray; sray.o = vadd(point, vmul(ldir, 0.001)); sray.d = ldir;
            c3 filter;
            if (!scene_intersects_ray(s, &sray, dist, &filter, cnt)) continue;
            double dot_ldir_norm = vdot(ldir, normal);
            float dcoeff = (float)dot_ldir_norm;
            dcoeff = dcoeff > 0.0f ? dcoeff : 0.0f; /* f32::max(0.0) */
            c3 diffuse_color = C3(m->diffuse[0] * tex.x, m->diffuse[1] * tex.y, m->diffuse[2] * tex.z);
            c3 diffuse = C3(diffuse_color.x * dcoeff, diffuse_color.y * dcoeff, diffuse_color.z * dcoeff);
            v3 lproj = vmul(normal, dot_ldir_norm);
            v3 rldir = vnormalize(vadd(vneg(ldir), vmul(lproj, 2.0)));
            float scoeff = (float)(-vdot(rldir, ray->ray.d));
            if (scoeff > 0.0f) {
                scoeff = powf(scoeff, m->shininess);
                c3 sp = C3(m->specular[0] * scoeff, m->specular[1] * scoeff, m->specular[2] * scoeff);
                acc.x = acc.x + light->color[0] * (filter.x * (diffuse.x + sp.x));
                acc.y = acc.y + light->color[1] * (filter.y * (diffuse.y + sp.y));
                acc.z = acc.z + light->color[2] * (filter.z * (diffuse.z + sp.z));
            } else {
                acc.x = acc.x + light->color[0] * (filter.x * diffuse.x);
                acc.y = acc.y + light->color[1] * (filter.y * diffuse.y);
                acc.z = acc.z + light->color[2] * (filter.z * diffuse.z);
            }
        }
        float inv = 1.0f / (float)(light->racsample * light->racsample); /* axpy(a, acc, 1.0): a*acc + res */
        res.x = inv * acc.x + res.x; res.y = inv * acc.y + res.y; res.z = inv * acc.z + res.z;
    }
    c4 out = {res.x, res.y, res.z, alpha};
    return out;
}

/* ------------------------------------------------------------------------------------------ */
/* Scene::trace + ClosestRayTOICostFn (src/scene.rs:163-252,262-283)                            */
/* ------------------------------------------------------------------------------------------ */
typedef struct { CostFn base; const OScene* scene; Ray ray; Counters* cnt; } ClosestCostFn;
static int world_bv_cost(CostFn* f, const AABB* bv, double* cost) {
    ClosestCostFn* s = (ClosestCostFn*)f;
    s->cnt->node_tests++;
    return aabb_toi(bv, &s->ray, cost);
}
static int world_b_cost(CostFn* f, int leaf, double* cost, Inter* in) {
    ClosestCostFn* s = (ClosestCostFn*)f;
    if (!node_cast(&s->scene->nodes[leaf], &s->ray, in, s->cnt)) return 0;
    *cost = in->toi;
    return 1;
}

#define HARD_DEPTH_CAP 64

static c3 scene_trace(const OScene* sc, const RayWE* ray, uint32_t depth, uint32_t max_depth, Counters* cnt) {
    ClosestCostFn fn; fn.base.bv_cost = world_bv_cost; fn.base.b_cost = world_b_cost;
    fn.scene = sc; fn.ray = ray->ray; fn.cnt = cnt;
    int leaf; Inter inter;
    if (!bvt_best_first(&sc->world, &fn.base, &leaf, &inter, NULL)) return sc->background;
    const ONode* sn = &sc->nodes[leaf];
    cnt->hit_records++;
    v3 pt = vadd(ray->ray.o, vmul(ray->ray.d, inter.toi));
    c4 obj = material_compute(sc, &sc->desc->materials[sn->d.material_id], ray, pt, &inter, cnt);
    int may_recurse = depth < HARD_DEPTH_CAP && (max_depth == 0 || depth < max_depth);

    /* trace_reflection, scene.rs:196-218 */
    c3 refl = C3(0.0f, 0.0f, 0.0f);
    float mix = sn->d.refl_mix;
    if (mix != 0.0f && ray->energy > 0.1f && may_recurse) {
        v3 nproj = vmul(inter.normal, vdot(ray->ray.d, inter.normal));
        v3 rdir = vsub(ray->ray.d, vmul(nproj, 2.0));
        RayWE r2; r2.ray.o = vadd(pt, vmul(rdir, 0.001)); r2.ray.d = rdir; r2.refr = ray->refr;
        r2.energy = ray->energy - sn->d.refl_atenuation; r2.key = rng_hash(ray->key, RNG_SALT_REFL);
        cnt->rays_reflection++;
        refl = scene_trace(sc, &r2, depth + 1, max_depth, cnt);
    }
    float alpha = obj.w * sn->d.alpha;
    c3 obj_color = C3(obj.x * (1.0f - mix) + refl.x * mix, obj.y * (1.0f - mix) + refl.y * mix, obj.z * (1.0f - mix) + refl.z * mix);

    /* trace_refraction, scene.rs:221-252 */
    c3 refr = C3(0.0f, 0.0f, 0.0f);
    if (alpha != 1.0f && may_recurse) {
        double n1, n2;
        if (ray->refr == 1.0) { n1 = 1.0; n2 = sn->d.refr_coeff; } else { n1 = sn->d.refr_coeff; n2 = 1.0; }
        v3 dir_along_normal = vmul(inter.normal, vdot(ray->ray.d, inter.normal));
        v3 tangent = vsub(ray->ray.d, dir_along_normal);
        v3 new_dir = vnormalize(vadd(dir_along_normal, vmul(tangent, n2 / n1)));
        RayWE r2; r2.ray.o = vadd(pt, vmul(new_dir, 0.001)); r2.ray.d = new_dir; r2.refr = n2;
        r2.energy = ray->energy; r2.key = rng_hash(ray->key, RNG_SALT_REFR);
        cnt->rays_refraction++;
        refr = scene_trace(sc, &r2, depth + 1, max_depth, cnt);
    }
    if (alpha == 1.0f) return obj_color;
    return C3(obj_color.x * alpha + refr.x * (1.0f - alpha), obj_color.y * alpha + refr.y * (1.0f - alpha),
              obj_color.z * alpha + refr.z * (1.0f - alpha));
}

/* ------------------------------------------------------------------------------------------ */
/* scene construction (Scene::new src/scene.rs:119-133; SceneNode::new src/scene_node.rs:22-47) */
/* ------------------------------------------------------------------------------------------ */
static int oscene_build(OScene* s, const NraysSceneDesc* d) {
    memset(s, 0, sizeof(*s));
    s->desc = d;
    s->background = C3(d->background[0], d->background[1], d->background[2]);
    s->meshes = (OMesh*)calloc(d->num_meshes ? d->num_meshes : 1, sizeof(OMesh));
    for (uint32_t i = 0; i < d->num_meshes; ++i) {
        OMesh* m = &s->meshes[i];
        const NraysMesh* dm = &d->meshes[i];
        m->nverts = (int)dm->num_vertices; m->ntris = (int)dm->num_triangles;
        m->verts = dm->vertices; m->uvs = dm->uvs; m->idx = dm->indices;
        if (m->ntris == 0) { memset(&m->bvt, 0, sizeof(m->bvt)); m->bvt.root = -1; m->aabb.mins = m->aabb.maxs = V(0, 0, 0); continue; }
        AABB* bvs = (AABB*)malloc((size_t)m->ntris * sizeof(AABB));
        for (int t = 0; t < m->ntris; ++t) {
            v3 a = mesh_vert(m, m->idx[3 * t]), b = mesh_vert(m, m->idx[3 * t + 1]), c = mesh_vert(m, m->idx[3 * t + 2]);
            bvs[t].mins = V(fmin(a.x, fmin(b.x, c.x)), fmin(a.y, fmin(b.y, c.y)), fmin(a.z, fmin(b.z, c.z)));
            bvs[t].maxs = V(fmax(a.x, fmax(b.x, c.x)), fmax(a.y, fmax(b.y, c.y)), fmax(a.z, fmax(b.z, c.z)));
        }
        bvt_build(&m->bvt, m->ntris, bvs);
        m->aabb = m->bvt.nodes[m->bvt.root].bv;
        free(bvs);
    }
    s->nnodes = (int)d->num_nodes;
    s->nodes = (ONode*)calloc(d->num_nodes ? d->num_nodes : 1, sizeof(ONode));
    AABB* nbvs = (AABB*)malloc((d->num_nodes ? d->num_nodes : 1) * sizeof(AABB));
    for (uint32_t i = 0; i < d->num_nodes; ++i) {
        ONode* n = &s->nodes[i];
        n->d = d->nodes[i];
        if (n->d.material_id >= d->num_materials) return NRAYS_ERR_BAD_ARG;
        iso_from_axis_angle(&n->iso, n->d.translation, n->d.axis_angle);
        n->mesh = NULL;
        if (n->d.shape_kind == NRAYS_SHAPE_TRIMESH) {
            if (n->d.mesh_id < 0 || (uint32_t)n->d.mesh_id >= d->num_meshes) return NRAYS_ERR_BAD_ARG;
            n->mesh = &s->meshes[n->d.mesh_id];
        }
        n->aabb = node_world_aabb(n);
        nbvs[i] = n->aabb;
    }
    bvt_build(&s->world, s->nnodes, nbvs);
    free(nbvs);
    return NRAYS_OK;
}
static void oscene_free(OScene* s) {
    if (s->meshes) for (uint32_t i = 0; i < s->desc->num_meshes; ++i) free(s->meshes[i].bvt.nodes);
    free(s->meshes); free(s->nodes); free(s->world.nodes);
}

/* ------------------------------------------------------------------------------------------ */
/* scene::render (src/scene.rs:29-116)                                                         */
/* ------------------------------------------------------------------------------------------ */
static inline int row_owned(const NraysRenderParams* p, uint32_t j) {
    if (p->band_rows == 0 || p->band_owners <= 1) return 1;
    return ((j / p->band_rows) % p->band_owners) == p->band_owner;
}
static inline uint32_t local_row(const NraysRenderParams* p, uint32_t j) {
    if (p->band_rows == 0 || p->band_owners <= 1) return j;
    uint32_t band = j / p->band_rows;
    return (band / p->band_owners) * p->band_rows + (j % p->band_rows);
}

static c3 render_pixel(const OScene* sc, const NraysRenderParams* p, uint32_t i, uint32_t j, Counters* cnt) {
    c3 tot = C3(0.0f, 0.0f, 0.0f);
    double resx = (double)p->width, resy = (double)p->height;
    const double* M = p->inv_proj_view; /* column-major */
    v3 eye0 = V(p->camera_eye[0], p->camera_eye[1], p->camera_eye[2]);
    uint64_t pkey = rng_hash(p->seed, (uint64_t)i + (uint64_t)j * p->width);
    for (uint32_t s = 0; s < p->ray_per_pixel; ++s) {
        uint64_t skey = rng_hash(pkey, s);
        double ox = (double)i, oy = (double)j;
        if (p->window_width != 0.0) { /* scene.rs:74-76 */
            ox = ox + (rng_u01(skey, 0) - 0.5) * p->window_width;
            oy = oy + (rng_u01(skey, 1) - 0.5) * p->window_width;
        }
        double dx = (ox / resx - 0.5) * 2.0;
        double dy = -(oy / resy - 0.5) * 2.0;
        double sv[4] = {dx, dy, -1.0, 1.0};
        double h[4];
        for (int r = 0; r < 4; ++r) h[r] = M[r] * sv[0] + M[4 + r] * sv[1] + M[8 + r] * sv[2] + M[12 + r] * sv[3];
        v3 eye = V(h[0] / h[3], h[1] / h[3], h[2] / h[3]);
        RayWE ray; ray.ray.o = eye0; ray.ray.d = vnormalize(vsub(eye, eye0)); ray.refr = 1.0; ray.energy = 1.0f;
        ray.key = rng_hash(skey, RNG_SALT_PATH);
        cnt->rays_primary++;
        c3 c = scene_trace(sc, &ray, 0, p->max_depth, cnt);
        tot.x = tot.x + c.x; tot.y = tot.y + c.y; tot.z = tot.z + c.z;
    }
    float n = (float)p->ray_per_pixel;
    return C3(tot.x / n, tot.y / n, tot.z / n);
}

typedef struct {
    const OScene* sc; const NraysRenderParams* p; float* out;
    size_t low, up; Counters cnt;
    int reps;                  /* the thread renders its range this many times (1 for a plain render) */
    pthread_barrier_t* start;  /* timed runs only */
    double cpu_s; /* CPU time of this thread between the gate and its last pixel (timed runs) */
    double t_begin, t_end; /* CLOCK_MONOTONIC when this thread left the gate / wrote its last pixel (timed runs) */
} Job;

static void* render_job(void* arg) {
    Job* jb = (Job*)arg;
    const NraysRenderParams* p = jb->p;
    struct timespec c0, c1;
    if (jb->start) { pthread_barrier_wait(jb->start); clock_gettime(CLOCK_THREAD_CPUTIME_ID, &c0); clock_gettime(CLOCK_MONOTONIC, &c1); jb->t_begin = (double)c1.tv_sec + 1e-9 * (double)c1.tv_nsec; } /* timed runs: all threads leave the gate together */
    for (int rep = 0; rep < jb->reps; ++rep)
    for (size_t ipt = jb->low; ipt < jb->up; ++ipt) {
        uint32_t j = (uint32_t)(ipt / p->width), i = (uint32_t)(ipt - (size_t)j * p->width);
        if (!row_owned(p, j)) continue;
        c3 c = render_pixel(jb->sc, p, i, j, &jb->cnt);
        size_t o = ((size_t)local_row(p, j) * p->width + i) * 3;
        jb->out[o] = c.x; jb->out[o + 1] = c.y; jb->out[o + 2] = c.z;
    }
    if (jb->start) {
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &c1); jb->cpu_s = (double)(c1.tv_sec - c0.tv_sec) + 1e-9 * (double)(c1.tv_nsec - c0.tv_nsec);
        clock_gettime(CLOCK_MONOTONIC, &c1); jb->t_end = (double)c1.tv_sec + 1e-9 * (double)c1.tv_nsec;
    }
    return NULL;
}

static __thread char g_err[256];
const char* nrays_oracle_last_error(void) { return g_err; }

/* The oracle entry point.  Same descriptors and output layout as nrays_render.  `num_threads`
 * follows scene.rs:49-66: contiguous static ranges of parts = npixels/num_threads + 1 pixels. */
static int oracle_render_impl(const NraysSceneDesc* desc, const NraysRenderParams* params, float* out_rgb,
                              int num_threads, NraysStats* stats, int reps, double* seconds);
/* CPU seconds of the threads of this thread's last timed call: {sum, min, max} — the static contiguous partition of
 * scene.rs:61-63 gives every thread the same number of pixels, not the same work. */
static __thread double g_thread_cpu[3];
void nrays_oracle_last_thread_cpu(double out[3]) { out[0] = g_thread_cpu[0]; out[1] = g_thread_cpu[1]; out[2] = g_thread_cpu[2]; }

int nrays_oracle_render(const NraysSceneDesc* desc, const NraysRenderParams* params, float* out_rgb,
                        int num_threads, NraysStats* stats) {
    return oracle_render_impl(desc, params, out_rgb, num_threads, stats, 1, NULL);
}

/* The CPU-baseline form (bench.py): the scene (BVTs) is built BEFORE the clock starts, as Scene::new is outside
 * scene::render in the reference (loader3d.rs:57-93); the threads are created once, wait at a barrier, and each renders
 * its static pixel range (scene.rs:49-66) `reps` times; `*seconds` is the wall time from the barrier to the last join.
 * `stats` holds the counts of all `reps` frames. */
int nrays_oracle_render_timed(const NraysSceneDesc* desc, const NraysRenderParams* params, float* out_rgb,
                              int num_threads, int reps, NraysStats* stats, double* seconds) {
    if (reps < 1 || !seconds) { snprintf(g_err, sizeof g_err, "bad timing arguments"); return NRAYS_ERR_BAD_ARG; }
    return oracle_render_impl(desc, params, out_rgb, num_threads, stats, reps, seconds);
}

static int oracle_render_impl(const NraysSceneDesc* desc, const NraysRenderParams* params, float* out_rgb,
                              int num_threads, NraysStats* stats, int reps, double* seconds) {
    if (!desc || !params || !out_rgb) { snprintf(g_err, sizeof g_err, "null argument"); return NRAYS_ERR_BAD_ARG; }
    if (params->ray_per_pixel == 0 || params->width == 0 || params->height == 0) {
        snprintf(g_err, sizeof g_err, "ray_per_pixel, width and height must be > 0"); return NRAYS_ERR_BAD_ARG;
    }
    OScene sc;
    int rc = oscene_build(&sc, desc);
    if (rc != NRAYS_OK) { snprintf(g_err, sizeof g_err, "bad scene descriptor"); oscene_free(&sc); return rc; }
    if (num_threads < 1) num_threads = 1;
    size_t npixels = (size_t)params->width * params->height;
    Job* jobs = (Job*)calloc((size_t)num_threads, sizeof(Job));
    pthread_t* th = (pthread_t*)calloc((size_t)num_threads, sizeof(pthread_t));
    size_t parts = npixels / (size_t)num_threads + 1;
    for (int t = 0; t < num_threads; ++t) {
        jobs[t].sc = &sc; jobs[t].p = params; jobs[t].out = out_rgb; jobs[t].reps = reps; jobs[t].start = NULL;
        jobs[t].low = parts * (size_t)t;
        jobs[t].up = parts * (size_t)(t + 1) < npixels ? parts * (size_t)(t + 1) : npixels;
        if (jobs[t].low > npixels) jobs[t].low = npixels;
    }
    if (seconds) { /* timed: threads + the caller meet at a barrier, the clock runs from there to the last join */
        pthread_barrier_t gate;
        struct timespec t0, t1;
        pthread_barrier_init(&gate, NULL, (unsigned)num_threads + 1u);
        for (int t = 0; t < num_threads; ++t) { jobs[t].start = &gate; pthread_create(&th[t], NULL, render_job, &jobs[t]); }
        pthread_barrier_wait(&gate);
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int t = 0; t < num_threads; ++t) pthread_join(th[t], NULL);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        pthread_barrier_destroy(&gate);
        /* wall time of the threaded region = first thread out of the gate .. last thread done, from the threads' own clocks: the caller
         * shares the cores with them and may be scheduled long after the gate opened (its own t0 then starts late) */
        (void)t0; (void)t1;
        { double b = jobs[0].t_begin, e = jobs[0].t_end;
          for (int t = 1; t < num_threads; ++t) { if (jobs[t].t_begin < b) b = jobs[t].t_begin; if (jobs[t].t_end > e) e = jobs[t].t_end; }
          *seconds = e - b; }
        g_thread_cpu[0] = 0.0; g_thread_cpu[1] = jobs[0].cpu_s; g_thread_cpu[2] = jobs[0].cpu_s;
        for (int t = 0; t < num_threads; ++t) {
            g_thread_cpu[0] += jobs[t].cpu_s;
            if (jobs[t].cpu_s < g_thread_cpu[1]) g_thread_cpu[1] = jobs[t].cpu_s;
            if (jobs[t].cpu_s > g_thread_cpu[2]) g_thread_cpu[2] = jobs[t].cpu_s;
        }
    } else if (num_threads == 1) render_job(&jobs[0]);
    else {
        for (int t = 0; t < num_threads; ++t) pthread_create(&th[t], NULL, render_job, &jobs[t]);
        for (int t = 0; t < num_threads; ++t) pthread_join(th[t], NULL);
    }
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        for (int t = 0; t < num_threads; ++t) {
            stats->rays_primary += jobs[t].cnt.rays_primary; stats->rays_reflection += jobs[t].cnt.rays_reflection;
            stats->rays_refraction += jobs[t].cnt.rays_refraction; stats->rays_shadow += jobs[t].cnt.rays_shadow;
            stats->node_tests += jobs[t].cnt.node_tests; stats->tri_tests += jobs[t].cnt.tri_tests;
            stats->prim_tests += jobs[t].cnt.prim_tests; stats->hit_records += jobs[t].cnt.hit_records;
            stats->tex_samples += jobs[t].cnt.tex_samples;
            stats->rays_primary_traced += jobs[t].cnt.rays_primary; /* the reference queries the BVT for every primary ray */
        }
        stats->instrumented = 1;
    }
    free(jobs); free(th);
    oscene_free(&sc);
    return NRAYS_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Probes used by the known-answer tests: one query, no rendering.                             */
/* ------------------------------------------------------------------------------------------ */
/* Closest hits of `n` rays against the scene (ClosestRayTOICostFn), or by brute force over every
 * node and triangle (no BVT) when `bruteforce` != 0: the BVT-equivalence property test.
 * Per ray: out[8] = toi, n[3], has_uv, u, v, node; hit[i] = 0/1. */
static int brute_cast(const OScene* sc, const Ray* ray, Inter* bi, int* node, Counters* cnt) {
    double best = DBL_MAX; int best_node = -1;
    for (int i = 0; i < sc->nnodes; ++i) {
        const ONode* n = &sc->nodes[i];
        Inter in; int hit = 0;
        double dummy;
        memset(&in, 0, sizeof in);
        if (!aabb_toi(&n->aabb, ray, &dummy)) continue; /* the node's world AABB gates its cast (scene.rs:276) */
        if (n->d.shape_kind == NRAYS_SHAPE_TRIMESH) {
            Ray l = iso_inv_ray(&n->iso, ray);
            double bt = DBL_MAX;
            const OMesh* me = n->mesh;
            for (int t = 0; t < me->ntris; ++t) {
                double toi, bary[3]; v3 nn;
                uint32_t i0 = me->idx[3 * t], i1 = me->idx[3 * t + 1], i2 = me->idx[3 * t + 2];
                v3 pa = mesh_vert(me, i0), pb = mesh_vert(me, i1), pc = mesh_vert(me, i2);
                AABB tb;
                tb.mins = V(fmin(pa.x, fmin(pb.x, pc.x)), fmin(pa.y, fmin(pb.y, pc.y)), fmin(pa.z, fmin(pb.z, pc.z)));
                tb.maxs = V(fmax(pa.x, fmax(pb.x, pc.x)), fmax(pa.y, fmax(pb.y, pc.y)), fmax(pa.z, fmax(pb.z, pc.z)));
                if (!aabb_toi(&tb, &l, &dummy)) continue; /* the triangle's own AABB gates its test (TriMesh BVT leaf) */
                if (cast_triangle(pa, pb, pc, &l, &toi, &nn, bary) && toi < bt) {
                    bt = toi; hit = 1; in.toi = toi; in.normal = iso_rot(&n->iso, nn); in.prim = t;
                    in.has_uv = me->uvs != NULL;
                    if (me->uvs) {
                        in.u = me->uvs[2 * i0] * bary[0] + me->uvs[2 * i1] * bary[1] + me->uvs[2 * i2] * bary[2];
                        in.v = me->uvs[2 * i0 + 1] * bary[0] + me->uvs[2 * i1 + 1] * bary[1] + me->uvs[2 * i2 + 1] * bary[2];
                    }
                }
            }
        } else hit = node_cast(n, ray, &in, cnt);
        if (hit && in.toi < best) { best = in.toi; best_node = i; *bi = in; }
    }
    *node = best_node;
    return best_node >= 0;
}
int nrays_oracle_cast_batch(const NraysSceneDesc* desc, uint32_t n, const double* origins, const double* dirs,
                            int bruteforce, double* out, int32_t* hit) {
    OScene sc; int rc = oscene_build(&sc, desc);
    if (rc != NRAYS_OK) { oscene_free(&sc); return rc; }
    Counters cnt; memset(&cnt, 0, sizeof cnt);
    for (uint32_t r = 0; r < n; ++r) {
        Ray ray; ray.o = V(origins[3 * r], origins[3 * r + 1], origins[3 * r + 2]); ray.d = V(dirs[3 * r], dirs[3 * r + 1], dirs[3 * r + 2]);
        int leaf = -1; Inter in; memset(&in, 0, sizeof in);
        int h;
        if (bruteforce) h = brute_cast(&sc, &ray, &in, &leaf, &cnt);
        else {
            ClosestCostFn fn; fn.base.bv_cost = world_bv_cost; fn.base.b_cost = world_b_cost; fn.scene = &sc; fn.cnt = &cnt; fn.ray = ray;
            h = bvt_best_first(&sc.world, &fn.base, &leaf, &in, NULL);
        }
        hit[r] = h;
        double* o = out + 8 * (size_t)r;
        if (h) { o[0] = in.toi; o[1] = in.normal.x; o[2] = in.normal.y; o[3] = in.normal.z; o[4] = in.has_uv; o[5] = in.has_uv ? in.u : 0.0; o[6] = in.has_uv ? in.v : 0.0; o[7] = leaf; }
        else memset(o, 0, 8 * sizeof(double));
    }
    oscene_free(&sc);
    return NRAYS_OK;
}
/* Scene::intersects_ray: returns 1 and filter[3] if lit, 0 if blocked. */
int nrays_oracle_shadow(const NraysSceneDesc* desc, const double origin[3], const double dir[3], double maxtoi, float filter[3]) {
    OScene sc; int rc = oscene_build(&sc, desc);
    if (rc != NRAYS_OK) { oscene_free(&sc); return rc; }
    Counters cnt; memset(&cnt, 0, sizeof cnt);
    Ray ray; ray.o = V(origin[0], origin[1], origin[2]); ray.d = V(dir[0], dir[1], dir[2]);
    c3 f; int lit = scene_intersects_ray(&sc, &ray, maxtoi, &f, &cnt);
    filter[0] = f.x; filter[1] = f.y; filter[2] = f.z;
    oscene_free(&sc);
    return lit;
}
/* Texture2d::sample probe. */
void nrays_oracle_tex_sample(const NraysTexture* tex, double u, double v, float out[4]) {
    Counters cnt; memset(&cnt, 0, sizeof cnt);
    c4 r = tex_sample(tex, u, v, &cnt);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
/* World AABB of node `i` (geometry.bounding_volume(&transform), scene_node.rs:41). */
int nrays_oracle_node_aabb(const NraysSceneDesc* desc, uint32_t i, double out[6]) {
    OScene sc; int rc = oscene_build(&sc, desc);
    if (rc != NRAYS_OK || i >= desc->num_nodes) { oscene_free(&sc); return NRAYS_ERR_BAD_ARG; }
    out[0] = sc.nodes[i].aabb.mins.x; out[1] = sc.nodes[i].aabb.mins.y; out[2] = sc.nodes[i].aabb.mins.z;
    out[3] = sc.nodes[i].aabb.maxs.x; out[4] = sc.nodes[i].aabb.maxs.y; out[5] = sc.nodes[i].aabb.maxs.z;
    oscene_free(&sc);
    return NRAYS_OK;
}
/* RNG probe (shared spec with the HIP side). */
double nrays_oracle_rng_u01(uint64_t seed, uint64_t pixel, uint64_t sample, uint64_t dim) {
    return rng_u01(rng_hash(rng_hash(seed, pixel), sample), dim);
}
