// scene_build.cpp — NraysSceneDesc -> HostScene (see scene_build.h).
#include "scene_build.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <future>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>

#include "bvh_build.h"
#include "bvh_device.h"
#include "presplit_clip.h"

namespace nrays {
namespace {

// Isometry3::new(translation, axis_angle) -> rotation matrix, via the unit quaternion
// (cos(a/2), axis*sin(a/2)) like nalgebra's UnitQuaternion::from_scaled_axis (loader3d.rs:552).
void rotation_from_axis_angle(const double w[3], double R[9], bool& identity) {
    double angle = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (angle == 0.0) {
        for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        identity = true;
        return;
    }
    identity = false;
    // two separate libm calls, like Rust's f64::sin_cos: a merged sincos() differs in the last bit for ~0.16 % of
    // the angles (glibc), which is enough to flip knife-edge hits several bounces later
    static double (*volatile libm_sin)(double) = std::sin;
    static double (*volatile libm_cos)(double) = std::cos;
    double s = libm_sin(angle / 2.0), qw = libm_cos(angle / 2.0);
    double qi = w[0] / angle * s, qj = w[1] / angle * s, qk = w[2] / angle * s;
    double ww = qw * qw, ii = qi * qi, jj = qj * qj, kk = qk * qk;
    double ij = qi * qj * 2.0, wk = qw * qk * 2.0, wj = qw * qj * 2.0, ik = qi * qk * 2.0, jk = qj * qk * 2.0, wi = qw * qi * 2.0;
    R[0] = ww + ii - jj - kk; R[1] = ij - wk;           R[2] = wj + ik;
    R[3] = wk + ij;           R[4] = ww - ii + jj - kk; R[5] = jk - wi;
    R[6] = ik - wj;           R[7] = wi + jk;           R[8] = ww - ii - jj + kk;
}

// Conservative f32 world box of a local box [c - h, c + h] under (R, t).
PrimBounds world_box(const double R[9], const double t[3], const double c[3], const double h[3]) {
    PrimBounds b;
    for (int i = 0; i < 3; ++i) {
        double wc = R[3 * i] * c[0] + R[3 * i + 1] * c[1] + R[3 * i + 2] * c[2] + t[i];
        double wh = std::fabs(R[3 * i]) * h[0] + std::fabs(R[3 * i + 1]) * h[1] + std::fabs(R[3 * i + 2]) * h[2];
        // outward f32 rounding plus one more ulp to absorb the f64 rounding of the transform itself
        b.mn[i] = std::nextafterf(round_down_f32(wc - wh), -std::numeric_limits<float>::infinity());
        b.mx[i] = std::nextafterf(round_up_f32(wc + wh), std::numeric_limits<float>::infinity());
    }
    return b;
}

bool shape_has_uv(uint32_t kind) { return kind == NRAYS_SHAPE_BALL || kind == NRAYS_SHAPE_CUBOID; }

// ---- world AABB of a node, in the reference's own arithmetic --------------------------------
// SceneNode::new stores geometry.bounding_volume(&transform) (scene_node.rs:41) and Scene::trace only
// casts a node whose AABB the ray hits (scene.rs:276).  The kernels cull with conservative f32 boxes
// and then apply THIS box, with ncollide's exact slab test, to every accepted hit, so the set of
// candidate hits is the reference's.  Formulas: ncollide3d bounding_volume impls (SURVEY B-1..B-9).
struct V3 { double x, y, z; };
V3 mat_vec(const double R[9], V3 v) {
    return {R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z, R[6] * v.x + R[7] * v.y + R[8] * v.z};
}
V3 mat_t_vec(const double R[9], V3 v) {
    return {R[0] * v.x + R[3] * v.y + R[6] * v.z, R[1] * v.x + R[4] * v.y + R[7] * v.z, R[2] * v.x + R[5] * v.y + R[8] * v.z};
}
V3 mat_abs_vec(const double R[9], V3 v) {
    return {std::fabs(R[0]) * v.x + std::fabs(R[1]) * v.y + std::fabs(R[2]) * v.z,
            std::fabs(R[3]) * v.x + std::fabs(R[4]) * v.y + std::fabs(R[5]) * v.z,
            std::fabs(R[6]) * v.x + std::fabs(R[7]) * v.y + std::fabs(R[8]) * v.z};
}
V3 support_local(const NraysNode& d, V3 dir) { // SupportMap of Cylinder / Cone / Capsule (axis = local Y)
    double hh = d.params[0], r = d.params[1];
    V3 res = {dir.x, 0.0, dir.z};
    double n = std::sqrt(res.x * res.x + res.y * res.y + res.z * res.z);
    if (d.shape_kind == NRAYS_SHAPE_CYLINDER) {
        if (n == 0.0) res = {0, 0, 0}; else res = {res.x / n * r, res.y / n * r, res.z / n * r};
        res.y = std::copysign(hh, dir.y);
        return res;
    }
    if (d.shape_kind == NRAYS_SHAPE_CONE) {
        if (n == 0.0) return {0.0, std::copysign(hh, dir.y), 0.0};
        res = {res.x / n * r, res.y / n * r, res.z / n * r}; res.y = -hh;
        if (dir.x * res.x + dir.y * res.y + dir.z * res.z < dir.y * hh) return {0.0, hh, 0.0};
        return res;
    }
    return {0.0 + dir.x * r, std::copysign(hh, dir.y) + dir.y * r, 0.0 + dir.z * r};
}
void node_world_aabb(const NraysNode& n, const double R[9], const float mesh_mn[3], const float mesh_mx[3], double out[6]) {
    const double* t = n.translation;
    switch (n.shape_kind) {
    case NRAYS_SHAPE_BALL:
        for (int a = 0; a < 3; ++a) { out[a] = t[a] - n.params[0]; out[3 + a] = t[a] + n.params[0]; }
        return;
    case NRAYS_SHAPE_CUBOID: {
        V3 h = mat_abs_vec(R, {n.params[0], n.params[1], n.params[2]});
        out[0] = t[0] - h.x; out[1] = t[1] - h.y; out[2] = t[2] - h.z; out[3] = t[0] + h.x; out[4] = t[1] + h.y; out[5] = t[2] + h.z;
        return;
    }
    case NRAYS_SHAPE_PLANE:
        for (int a = 0; a < 3; ++a) { out[a] = -std::numeric_limits<double>::max(); out[3 + a] = std::numeric_limits<double>::max(); }
        return;
    case NRAYS_SHAPE_TRIMESH: {
        V3 c = {((double)mesh_mn[0] + (double)mesh_mx[0]) * 0.5, ((double)mesh_mn[1] + (double)mesh_mx[1]) * 0.5, ((double)mesh_mn[2] + (double)mesh_mx[2]) * 0.5};
        V3 h = {((double)mesh_mx[0] - (double)mesh_mn[0]) * 0.5, ((double)mesh_mx[1] - (double)mesh_mn[1]) * 0.5, ((double)mesh_mx[2] - (double)mesh_mn[2]) * 0.5};
        V3 rc = mat_vec(R, c);
        V3 wc = {rc.x + t[0], rc.y + t[1], rc.z + t[2]};
        V3 wh = mat_abs_vec(R, h);
        out[0] = wc.x - wh.x; out[1] = wc.y - wh.y; out[2] = wc.z - wh.z; out[3] = wc.x + wh.x; out[4] = wc.y + wh.y; out[5] = wc.z + wh.z;
        return;
    }
    default:
        for (int i = 0; i < 3; ++i) {
            V3 e = {i == 0 ? 1.0 : 0.0, i == 1 ? 1.0 : 0.0, i == 2 ? 1.0 : 0.0};
            V3 ld = mat_t_vec(R, e);
            V3 sp = mat_vec(R, support_local(n, ld));
            V3 sn = mat_vec(R, support_local(n, {-ld.x, -ld.y, -ld.z}));
            double spv[3] = {sp.x + t[0], sp.y + t[1], sp.z + t[2]}, snv[3] = {sn.x + t[0], sn.y + t[1], sn.z + t[2]};
            out[3 + i] = spv[i]; out[i] = snv[i];
        }
        return;
    }
}

struct Blas {
    int32_t root;
    float mn[3], mx[3]; // local bounds
    bool hairy;         // thin diagonal triangles throughout (presplit())
    bool device = false; // built by bvh_device.hip: root / leaf refs already address the scene's final arrays
};

// ---- triangle pre-splitting ----------------------------------------------------------------------
// Long thin triangles that run diagonally (hair, cables, banisters) have boxes that are almost entirely empty:
// every ray crossing the box pays a full f64 triangle test (~170 VALU instructions, the dominant cost of the
// hairball kernel).  Before the BVH is built, the references with the largest "empty" box area are split at the
// midpoint of their longest axis: the triangle is clipped against the plane and each half gets the box of its
// piece (rounded outward to f32, plus one ulp for the f64 rounding of the clip), so the pieces' boxes cover the
// triangle.  A reference still points at the WHOLE triangle: the leaf test, the tie-break key and the exact AABB
// gates are unchanged, a triangle reached through two references is simply found twice with the same
// (toi, node, triangle) key.  Measured (MI355X, 1080p): budget 1 / min gain 0.5: hairball 5.94 -> 4.57 ms (54 -> 27
// triangle tests per ray), sponza 1.85 -> 1.81 ms, +3.6 s of scene build on 2.88 M triangles; budget 2 / 0.2: hairball
// 4.22 ms but sponza +1 %; budget 3 / 0.1: hairball 3.96 ms, sponza +6 %, build +20 s — hence the per-mesh choice
// below: the aggressive setting only for hair-like meshes (hairball 4.20 -> 3.71 ms on top of the later leaf-loop work).
#ifndef NR_PRESPLIT_BUDGET
#define NR_PRESPLIT_BUDGET 1.0  // at most this many extra references per triangle on average
#endif
#ifndef NR_PRESPLIT_MINGAIN
#define NR_PRESPLIT_MINGAIN 1.0 // a piece is split while its empty box area exceeds this fraction of the average box area (round 5: 0.5 -> 1.0, see NR_PRIM_COST)
#endif
// refs_box / refs_tri: one entry per reference (initially one per triangle), grown in place.  Candidates are pieces
//   (a) whose box is at least NR_PRESPLIT_EMPTY empty (1 - 2 area / half box area: thin diagonal primitives; a large
//       axis-aligned wall triangle has emptiness 0 and is never split, whatever its size), and
//   (b) whose empty box area exceeds NR_PRESPLIT_MINGAIN times the average box area of the mesh;
// they are split in order of decreasing empty area (a heap) until the budget of NR_PRESPLIT_BUDGET extra references
// per triangle is used up, so a tight budget goes to the worst offenders first.
#ifndef NR_PRESPLIT_HAIRY
#define NR_PRESPLIT_HAIRY 0.9        // area-weighted emptiness of the mesh above which it counts as hair-like
#endif
#ifndef NR_PRIM_COST_HAIRY
#define NR_PRIM_COST_HAIRY 0.7f // re-tuned with the quorum-ended node phases of round 3 (0.35 before): hairball 2.36 -> 2.27 ms, 14.6 -> 9.0 triangle tests per ray
#endif
#ifndef NR_PRESPLIT_BUDGET_HAIRY
#define NR_PRESPLIT_BUDGET_HAIRY 8.0 // round 4 (builds on the device cost milliseconds): 5.0 -> 8.0 with the finer rule below
#endif
#ifndef NR_PRESPLIT_MINGAIN_HAIRY
#define NR_PRESPLIT_MINGAIN_HAIRY 0.02 // round 4, tools/build_sweep.py (profiles/r04_build_sweep.log): 0.05 -> 0.02: hairball 2.04 -> 1.95 ms (9.0 -> 6.1 triangle tests per ray, 1.5 -> 2.3 GB); below it the frame stays at 1.95 (the node loop, not the triangle tests, bounds it)
#endif
// The exact procedure: pieces split in order of decreasing empty area until the budget is used up.  `tris` lists the
// triangles taking part (all of them, or a sample); boxes[k] / owner[k] describe reference k (k < tris.size(): the
// k-th listed triangle).  Returns the empty area of the last piece split (the threshold the budget amounts to), or
// `min_gain` when every candidate was split before the budget ran out.
static double presplit_heap(const std::vector<TriRec>& recs, const std::vector<uint32_t>& tris, std::vector<PrimBounds>& boxes,
                            std::vector<uint32_t>& owner, size_t budget, double min_gain) {
    struct Cand { double gain; uint32_t ref; uint32_t poly; };
    auto cmp = [](const Cand& a, const Cand& b) { return a.gain < b.gain; };
    std::vector<Cand> heap;
    std::vector<ClipPoly> polys; // pool; slots of split pieces are reused by their low halves
    auto consider = [&](uint32_t ref, uint32_t poly_slot) {
        const double ha = box_half_area(boxes[ref]);
        const double gain = ha - poly_area2(polys[poly_slot]);
        if (gain > min_gain && gain > NR_PRESPLIT_EMPTY * ha) { heap.push_back(Cand{gain, ref, poly_slot}); std::push_heap(heap.begin(), heap.end(), cmp); return true; }
        return false;
    };
    for (size_t k = 0; k < tris.size(); ++k) {
        ClipPoly p; tri_poly(recs[tris[k]], p);
        polys.push_back(p);
        if (!consider((uint32_t)k, (uint32_t)polys.size() - 1)) polys.pop_back();
    }
    double last = min_gain;
    while (budget > 0 && !heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), cmp);
        const Cand c = heap.back(); heap.pop_back();
        ClipPoly lo, hi; PrimBounds bl, bh;
        if (!split_piece(polys[c.poly], boxes[c.ref], lo, hi, bl, bh)) continue;
        last = c.gain;
        boxes[c.ref] = bl;
        const uint32_t ref_hi = (uint32_t)boxes.size();
        boxes.push_back(bh); owner.push_back(owner[c.ref]);
        polys[c.poly] = lo;
        consider(c.ref, c.poly);
        polys.push_back(hi);
        if (!consider(ref_hi, (uint32_t)polys.size() - 1)) polys.pop_back();
        --budget;
    }
    return heap.empty() ? min_gain : last;
}
// The same rule as a threshold: a piece is split while its empty area exceeds `thr` (children have less empty area than
// their parent, so this is what the heap does once the threshold its budget amounts to is known) — no global state,
// hence one task per range of triangles.
struct SplitOut { std::vector<PrimBounds> box; std::vector<uint32_t> tri; };
static void split_rec(const ClipPoly& poly, const PrimBounds& box, uint32_t tri, double thr, int depth, PrimBounds* root_slot, long slot, SplitOut& out) {
    auto store = [&](const PrimBounds& b) { if (slot < 0) *root_slot = b; else out.box[(size_t)slot] = b; };
    const double ha = box_half_area(box);
    const double gain = ha - poly_area2(poly);
    ClipPoly lo, hi; PrimBounds bl, bh;
    if (depth >= kSplitDepthMax || !(gain > thr && gain > NR_PRESPLIT_EMPTY * ha) || !split_piece(poly, box, lo, hi, bl, bh)) { store(box); return; }
    const long hi_slot = (long)out.box.size();
    out.box.push_back(bh); out.tri.push_back(tri);
    split_rec(lo, bl, tri, thr, depth + 1, root_slot, slot, out);
    split_rec(hi, bh, tri, thr, depth + 1, root_slot, hi_slot, out);
}
// Returns whether the mesh is hair-like (see below).
static bool presplit(const std::vector<TriRec>& recs, std::vector<PrimBounds>& refs_box, std::vector<uint32_t>& refs_tri) {
    const size_t n = recs.size();
    if (n < 64 || NR_PRESPLIT_BUDGET <= 0.0) return false;
    double total_area = 0.0;
    for (size_t i = 0; i < n; ++i) total_area += box_half_area(refs_box[i]);
    // Hair-like meshes (most triangles are thin and diagonal: the average box is more than NR_PRESPLIT_HAIRY empty)
    // get the aggressive setting; architectural meshes, where only a few curved parts qualify, the mild one.
    double total_tri2 = 0.0;
    for (size_t i = 0; i < n; ++i) { ClipPoly p; tri_poly(recs[i], p); total_tri2 += poly_area2(p); }
    const bool hairy = total_area > 0.0 && (total_area - total_tri2) > NR_PRESPLIT_HAIRY * total_area;
    const double budget_per_tri = hairy ? NR_PRESPLIT_BUDGET_HAIRY : NR_PRESPLIT_BUDGET;
    const double min_gain = (hairy ? NR_PRESPLIT_MINGAIN_HAIRY : NR_PRESPLIT_MINGAIN) * total_area / (double)n;
    const size_t budget = (size_t)(budget_per_tri * (double)n);
    constexpr size_t kExactBelow = 400000; // triangles: below this the heap runs on the whole mesh
    if (n <= kExactBelow) {
        std::vector<uint32_t> all(n);
        for (size_t i = 0; i < n; ++i) all[i] = (uint32_t)i;
        presplit_heap(recs, all, refs_box, refs_tri, budget, min_gain);
        return hairy;
    }
    // Large meshes: the threshold is measured on every s-th triangle with 1/s of the budget, then all triangles are
    // split against it in parallel.
    const size_t stride = (n + kExactBelow / 4 - 1) / (kExactBelow / 4);
    std::vector<uint32_t> sample; std::vector<PrimBounds> sbox; std::vector<uint32_t> sowner;
    for (size_t i = 0; i < n; i += stride) { sample.push_back((uint32_t)i); sbox.push_back(refs_box[i]); sowner.push_back((uint32_t)i); }
    const double thr = presplit_heap(recs, sample, sbox, sowner, budget / stride, min_gain);
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t tasks = std::min<size_t>(std::max<unsigned>(hw, 1u), 64);
    std::vector<SplitOut> outs(tasks);
    std::vector<std::future<void>> futs;
    for (size_t t = 0; t < tasks; ++t) {
        const size_t lo = n * t / tasks, hi = n * (t + 1) / tasks;
        futs.push_back(std::async(std::launch::async, [&, lo, hi, t]() {
            for (size_t i = lo; i < hi; ++i) {
                ClipPoly p; tri_poly(recs[i], p);
                const PrimBounds box = refs_box[i];
                split_rec(p, box, (uint32_t)i, thr, 0, &refs_box[i], -1, outs[t]);
            }
        }));
    }
    for (auto& f : futs) f.get();
    size_t extra = 0;
    for (const SplitOut& o : outs) extra += o.box.size();
    refs_box.reserve(n + extra); refs_tri.reserve(n + extra);
    for (const SplitOut& o : outs) { refs_box.insert(refs_box.end(), o.box.begin(), o.box.end()); refs_tri.insert(refs_tri.end(), o.tri.begin(), o.tri.end()); }
    return hairy;
}

// Appends a BLAS over the triangles of `node_ids` (TriMesh nodes sharing one isometry).
#ifndef NR_MAX_LEAF
#define NR_MAX_LEAF 8
#endif
// Triangle count from which a BLAS is built on the GPU (bvh_device.hip); NRAYS_GPU_BUILD=0: never, NRAYS_GPU_BUILD_MIN=n: from n triangles.
// Crossover on MI355X (tools/build_crossover.py, profiles/r04_build_crossover.log): 1 000 triangles host 1.6 / device 2.1 ms, 3 000: 3.9 / 2.5, 50 000: 65 / 3.6, 1 M: 337 / 15.
static size_t device_build_min() { // read per scene (not cached): tests and A/B runs flip it between two nrays_scene_create calls
    if (const char* e = getenv("NRAYS_GPU_BUILD")) if (atoi(e) == 0) return std::numeric_limits<size_t>::max();
    if (const char* e = getenv("NRAYS_GPU_BUILD_MIN")) return (size_t)std::max(1ll, atoll(e));
    return (size_t)2000;
}
static DeviceBuildOptions device_options(bool presplit_on) {
    DeviceBuildOptions o;
    o.max_leaf = NR_MAX_LEAF; o.prim_cost = NR_PRIM_COST; o.prim_cost_hairy = NR_PRIM_COST_HAIRY;
    o.budget = NR_PRESPLIT_BUDGET; o.budget_hairy = NR_PRESPLIT_BUDGET_HAIRY; o.min_gain = NR_PRESPLIT_MINGAIN; o.min_gain_hairy = NR_PRESPLIT_MINGAIN_HAIRY;
    o.hairy_emptiness = NR_PRESPLIT_HAIRY; o.presplit = presplit_on;
    // tuning overrides, read per scene (tools/build_sweep.py): a device build costs milliseconds, so the quality knobs can be swept in one process
    auto envd = [](const char* name, double& v) { if (const char* e = getenv(name)) v = atof(e); };
    auto envf = [](const char* name, float& v) { if (const char* e = getenv(name)) v = (float)atof(e); };
    envd("NRAYS_PRESPLIT_BUDGET", o.budget); envd("NRAYS_PRESPLIT_BUDGET_HAIRY", o.budget_hairy);
    envd("NRAYS_PRESPLIT_MINGAIN", o.min_gain); envd("NRAYS_PRESPLIT_MINGAIN_HAIRY", o.min_gain_hairy);
    envf("NRAYS_PRIM_COST", o.prim_cost); envf("NRAYS_PRIM_COST_HAIRY", o.prim_cost_hairy);
    if (const char* e = getenv("NRAYS_MAX_LEAF")) o.max_leaf = atoi(e);
    if (const char* e = getenv("NRAYS_DEBUG_BUILD_CAPS")) o.debug_cap_div = std::max(1, atoi(e));
    return o;
}

// which: 0 = by size (device_build_min), 1 = host builder, 2 = device builder
int append_blas(const NraysSceneDesc* d, const std::vector<uint32_t>& node_ids, HostScene& out, Blas& blas, std::string& err, int which = 0, bool presplit_on = true) {
    const auto TA = std::chrono::steady_clock::now();
    {
        size_t total = 0;
        for (uint32_t ni : node_ids) total += d->meshes[d->nodes[ni].mesh_id].num_triangles;
        if (which == 2 || (which == 0 && total >= device_build_min())) {
            std::vector<DeviceMeshPart> parts;
            for (uint32_t ni : node_ids) {
                const NraysMesh& m = d->meshes[d->nodes[ni].mesh_id];
                parts.push_back(DeviceMeshPart{m.vertices, m.uvs, m.indices, m.num_vertices, m.num_triangles, ni});
            }
            DeviceBlas db;
            const int rc = build_blas_device(parts, device_options(presplit_on), (int32_t)out.dev_nodes, (uint32_t)out.dev_tris, db, err);
            if (getenv("NRAYS_BUILD_TIMES")) fprintf(stderr, "device BLAS build %.3f s (%zu triangles, %zu refs%s)%s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - TA).count(),
                                                     total, db.num_refs, db.hairy ? ", hair-like" : "", rc == NRAYS_OK ? "" : " FAILED");
            if (rc == NRAYS_OK) {
                blas.root = db.root; blas.hairy = db.hairy; blas.device = true;
                for (int a = 0; a < 3; ++a) { blas.mn[a] = db.mn[a]; blas.mx[a] = db.mx[a]; }
                out.max_bvh_depth = std::max(out.max_bvh_depth, db.max_depth);
                out.dev_nodes += db.num_nodes; out.dev_tris += db.num_refs;
                out.dev_blas.push_back(db);
                return NRAYS_OK;
            }
            if (rc == NRAYS_ERR_BAD_ARG || rc == NRAYS_ERR_UNSUPPORTED || which == 2) return rc; // the caller's data, or a probe of the device builder itself
            fprintf(stderr, "nrays: %s; building this BLAS on the host\n", err.c_str()); // an internal limit of the device builder: the host builder takes over
            err.clear();
        }
    }
    std::vector<PrimBounds> pb;
    std::vector<TriRec> recs;
    std::vector<TriUv> uvs;
    for (uint32_t ni : node_ids) {
        const NraysMesh& m = d->meshes[d->nodes[ni].mesh_id];
        for (uint32_t t = 0; t < m.num_triangles; ++t) {
            TriRec r; TriUv uv; PrimBounds b;
            std::memset(&uv, 0, sizeof uv);
            float* vs[3] = {r.v0, r.v1, r.v2};
            for (int a = 0; a < 3; ++a) { b.mn[a] = std::numeric_limits<float>::infinity(); b.mx[a] = -std::numeric_limits<float>::infinity(); }
            for (int k = 0; k < 3; ++k) {
                uint32_t vi = m.indices[3 * t + k];
                if (vi >= m.num_vertices) { err = "triangle index out of range"; return NRAYS_ERR_BAD_ARG; }
                for (int a = 0; a < 3; ++a) {
                    double x = m.vertices[3 * (size_t)vi + a];
                    float f = (float)x;
                    if (!((double)f == x)) { // also rejects NaN
                        err = "mesh vertex coordinate is not exactly representable in f32 (see DESIGN.md: f32-exact mesh storage)";
                        return NRAYS_ERR_UNSUPPORTED;
                    }
                    vs[k][a] = f;
                    b.mn[a] = std::min(b.mn[a], f); b.mx[a] = std::max(b.mx[a], f);
                }
                if (m.uvs) {
                    for (int a = 0; a < 2; ++a) {
                        double x = m.uvs[2 * (size_t)vi + a];
                        float f = (float)x;
                        if (!((double)f == x)) { err = "mesh uv is not exactly representable in f32"; return NRAYS_ERR_UNSUPPORTED; }
                        uv.uv[2 * k + a] = f;
                    }
                }
            }
            r.node_id = ni; r.tri_id = t; r.pad = 0;
            recs.push_back(r); uvs.push_back(uv); pb.push_back(b);
        }
    }
    for (int a = 0; a < 3; ++a) { blas.mn[a] = std::numeric_limits<float>::infinity(); blas.mx[a] = -std::numeric_limits<float>::infinity(); }
    for (const PrimBounds& b : pb) for (int a = 0; a < 3; ++a) { blas.mn[a] = std::min(blas.mn[a], b.mn[a]); blas.mx[a] = std::max(blas.mx[a], b.mx[a]); }
    std::vector<uint32_t> ref_tri(recs.size());
    for (size_t k = 0; k < ref_tri.size(); ++k) ref_tri[k] = (uint32_t)k;
    auto T0 = std::chrono::steady_clock::now();
    const bool hairy = presplit_on && presplit(recs, pb, ref_tri); // pb becomes one box per REFERENCE
    auto T1 = std::chrono::steady_clock::now();
    // thin tubes: a leaf's triangles mostly miss, and a node visit is cheap — leaves split sooner (hairball 2.96 -> 2.91 ms; the
    // architectural stand-in is 4 % slower with this value, profiles/r02_nodeloop_ab.log)
    BuiltBvh bvh = build_bvh(pb, NR_MAX_LEAF, hairy ? NR_PRIM_COST_HAIRY : 0.0f);
    auto T2 = std::chrono::steady_clock::now();
    if (getenv("NRAYS_BUILD_TIMES") && recs.size() > 1000000) fprintf(stderr, "  append_blas: triangle records %.2f s\n", std::chrono::duration<double>(T0 - TA).count());
    if (getenv("NRAYS_BUILD_TIMES")) fprintf(stderr, "presplit %.2f s, build_bvh %.2f s (%zu triangles, %zu refs%s)\n", std::chrono::duration<double>(T1 - T0).count(), std::chrono::duration<double>(T2 - T1).count(), recs.size(), pb.size(), hairy ? ", hair-like" : "");
    if (out.tris.size() + pb.size() >= (1u << 28)) { err = "too many triangles"; return NRAYS_ERR_UNSUPPORTED; }
    rebase_bvh(bvh, (int32_t)out.nodes.size(), (uint32_t)out.tris.size());
    out.max_bvh_depth = std::max(out.max_bvh_depth, bvh.max_depth);
    {   // leaf-ordered copies of the triangle records and their uvs, by several threads (15 M references for the hairball)
        const size_t base = out.tris.size(), n = bvh.order.size();
        out.tris.resize(base + n); out.triuvs.resize(base + n);
        const size_t tasks = n >= 400000 ? std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), 32) : 1;
        auto gather = [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) { const uint32_t t = ref_tri[bvh.order[i]]; out.tris[base + i] = recs[t]; out.triuvs[base + i] = uvs[t]; } };
        std::vector<std::future<void>> futs;
        for (size_t t = 1; t < tasks; ++t) futs.push_back(std::async(std::launch::async, gather, n * t / tasks, n * (t + 1) / tasks));
        gather(0, n / tasks);
        for (auto& f : futs) f.get();
    }
    out.nodes.insert(out.nodes.end(), bvh.nodes.begin(), bvh.nodes.end());
    blas.root = bvh.root;
    blas.hairy = hairy;
    blas.device = false; // (the caller may hand in the Blas of an earlier, device-built group)
    if (getenv("NRAYS_BUILD_TIMES") && recs.size() > 1000000) fprintf(stderr, "  append_blas: after the build (rebase, gather, node copy) %.2f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - T2).count());
    return NRAYS_OK;
}

// Builds a TLAS over `insts` (reordering them in place) and appends its nodes.
int32_t append_tlas(std::vector<Instance>& insts, const std::vector<PrimBounds>& boxes, HostScene& out) {
    BuiltBvh bvh = build_bvh(boxes, 1);
    std::vector<Instance> re(insts.size());
    for (size_t k = 0; k < bvh.order.size(); ++k) re[k] = insts[bvh.order[k]];
    insts.swap(re);
    rebase_bvh(bvh, (int32_t)out.nodes.size(), 0);
    // the 3 low bits of a TLAS leaf ref carry the instance's LeafBits
    auto tag = [&](int32_t ref) -> int32_t {
        if (ref >= 0 || ref == kEmptyChild) return ref;
        uint32_t first = ((uint32_t)~ref) >> 3;
        const Instance& in = insts[first];
        uint32_t bits = (in.kind == NRAYS_SHAPE_TRIMESH ? kLeafMesh : 0u) | ((in.flags & kInstAnyHit) ? kLeafAnyHit : 0u) |
                        ((in.kind == NRAYS_SHAPE_TRIMESH && (in.flags & kInstNoXform)) ? kLeafNoXform : 0u);
        return ~(int32_t)((first << 3) | bits);
    };
    for (BvhNode& n : bvh.nodes) for (int k = 0; k < 4; ++k) n.children()[k] = tag(n.children()[k]);
    bvh.root = tag(bvh.root);
    out.max_bvh_depth = std::max(out.max_bvh_depth, bvh.max_depth);
    out.nodes.insert(out.nodes.end(), bvh.nodes.begin(), bvh.nodes.end());
    return bvh.root;
}

} // namespace

int build_blas_probe(const NraysMesh* mesh, bool device, bool presplit_on, BlasProbe& out, std::string& err) {
    if (!mesh || !mesh->vertices || !mesh->indices || mesh->num_triangles == 0) { err = "bad mesh"; return NRAYS_ERR_BAD_ARG; }
    NraysNode node; std::memset(&node, 0, sizeof node);
    node.shape_kind = NRAYS_SHAPE_TRIMESH; node.mesh_id = 0;
    NraysSceneDesc d; std::memset(&d, 0, sizeof d);
    d.meshes = mesh; d.num_meshes = 1; d.nodes = &node; d.num_nodes = 1;
    HostScene hs; Blas blas;
    const int rc = append_blas(&d, std::vector<uint32_t>{0u}, hs, blas, err, device ? 2 : 1, presplit_on);
    if (rc != NRAYS_OK) { for (DeviceBlas& b : hs.dev_blas) free_device_blas(b); return rc; }
    out.root = blas.root; out.hairy = blas.hairy; out.max_depth = hs.max_bvh_depth;
    if (device) {
        DeviceBlas& b = hs.dev_blas[0];
        out.nodes.resize(b.num_nodes);
        std::vector<TriRec> tris(b.num_refs);
        bool ok = (b.num_nodes == 0 || hipMemcpy(out.nodes.data(), b.nodes, b.num_nodes * sizeof(BvhNode), hipMemcpyDeviceToHost) == hipSuccess) &&
                  hipMemcpy(tris.data(), b.tris, b.num_refs * sizeof(TriRec), hipMemcpyDeviceToHost) == hipSuccess;
        free_device_blas(b);
        if (!ok) { err = "copy of the device-built BLAS failed"; return NRAYS_ERR_HIP; }
        for (const TriRec& r : tris) out.tri_ids.push_back(r.tri_id);
    } else {
        out.nodes = hs.nodes;
        for (const TriRec& r : hs.tris) out.tri_ids.push_back(r.tri_id);
    }
    return NRAYS_OK;
}

int build_host_scene(const NraysSceneDesc* d, HostScene& out, std::string& err) {
    if (!d) { err = "null scene descriptor"; return NRAYS_ERR_BAD_ARG; }
    if ((d->num_lights && !d->lights) || (d->num_materials && !d->materials) || (d->num_textures && !d->textures) ||
        (d->num_meshes && !d->meshes) || (d->num_nodes && !d->nodes)) { err = "null array with non-zero count"; return NRAYS_ERR_BAD_ARG; }
    for (int a = 0; a < 3; ++a) out.background[a] = d->background[a];

    for (uint32_t i = 0; i < d->num_lights; ++i) {
        const NraysLight& l = d->lights[i];
        LightRec r;
        for (int a = 0; a < 3; ++a) { r.pos[a] = l.pos[a]; r.color[a] = l.color[a]; }
        r.radius = l.radius; r.racsample = l.racsample;
        if (l.radius != 0.0) out.any_area_light = true;
        out.lights.push_back(r);
    }
    for (uint32_t i = 0; i < d->num_textures; ++i) {
        const NraysTexture& t = d->textures[i];
        if (t.width < 1 || t.height < 1 || !t.texels || t.format > NRAYS_TEXEL_RGBA32F) { err = "bad texture"; return NRAYS_ERR_BAD_ARG; }
        HostTexture h;
        h.rec.width = t.width; h.rec.height = t.height; h.rec.format = t.format; h.rec.interp = t.interp;
        h.rec.overflow = t.overflow; h.rec.pad = 0; h.rec.texels = nullptr;
        size_t nbytes = (size_t)t.width * t.height * (t.format == NRAYS_TEXEL_RGBA8 ? 4 : 16);
        h.bytes.assign((const uint8_t*)t.texels, (const uint8_t*)t.texels + nbytes);
        out.textures.push_back(std::move(h));
    }
    for (uint32_t i = 0; i < d->num_materials; ++i) {
        const NraysMaterial& m = d->materials[i];
        if (m.kind > NRAYS_MAT_UV) { err = "bad material kind"; return NRAYS_ERR_BAD_ARG; }
        if (m.texture_id >= (int32_t)d->num_textures || m.alpha_texture_id >= (int32_t)d->num_textures) { err = "bad texture id"; return NRAYS_ERR_BAD_ARG; }
        MaterialRec r; std::memset(&r, 0, sizeof r);
        r.kind = m.kind;
        for (int a = 0; a < 3; ++a) { r.ka[a] = m.ambiant[a]; r.kd[a] = m.diffuse[a]; r.ks[a] = m.specular[a]; }
        r.shininess = m.shininess;
        r.tex = m.texture_id < 0 ? -1 : m.texture_id;
        r.alpha_tex = m.alpha_texture_id < 0 ? -1 : m.alpha_texture_id;
        out.materials.push_back(r);
    }

    // ---- per-node records, opacity classification, transform groups ----------------------
    struct NodeInfo { double R[9]; bool identity; bool opaque; bool has_uv; bool no_uv_values; };
    std::vector<NodeInfo> info(d->num_nodes);
    std::vector<uint32_t> pending_aabb;                       // TriMesh nodes whose local AABB comes from their device-built BLAS (below)
    std::map<uint32_t, std::array<float, 6>> single_bounds;   // node -> local bounds of a device-built BLAS that holds this node alone
    float att_min = std::numeric_limits<float>::infinity();
    for (uint32_t i = 0; i < d->num_nodes; ++i) {
        const NraysNode& n = d->nodes[i];
        if (n.shape_kind > NRAYS_SHAPE_TRIMESH) { err = "bad shape kind"; return NRAYS_ERR_BAD_ARG; }
        if (n.material_id >= d->num_materials) { err = "bad material id"; return NRAYS_ERR_BAD_ARG; }
        if (n.shape_kind == NRAYS_SHAPE_TRIMESH && (n.mesh_id < 0 || (uint32_t)n.mesh_id >= d->num_meshes)) { err = "bad mesh id"; return NRAYS_ERR_BAD_ARG; }
        rotation_from_axis_angle(n.axis_angle, info[i].R, info[i].identity);
        const NraysMaterial& m = d->materials[n.material_id];
        bool has_uv = n.shape_kind == NRAYS_SHAPE_TRIMESH ? d->meshes[n.mesh_id].uvs != nullptr : shape_has_uv(n.shape_kind);
        info[i].has_uv = has_uv;
        info[i].no_uv_values = m.kind == NRAYS_MAT_NORMAL || (m.kind == NRAYS_MAT_PHONG && m.texture_id < 0 && m.alpha_texture_id < 0);
        // ambiant().w is 1 for NormalMaterial, 1/0 for UVMaterial with/without uvs, and the alpha
        // map's w (or 1) for PhongMaterial; a node blocks shadow rays iff w * node.alpha >= 1 (scene.rs:322-331).
        bool w_is_one = m.kind == NRAYS_MAT_NORMAL || (m.kind == NRAYS_MAT_UV && has_uv) ||
                        (m.kind == NRAYS_MAT_PHONG && (m.alpha_texture_id < 0 || !has_uv));
        info[i].opaque = w_is_one && n.alpha >= 1.0f;
        if (!(w_is_one && n.alpha == 1.0f)) out.any_transparent = true;
        if (!(w_is_one && n.alpha == 1.0f) && n.refl_mix != 0.0f) out.any_double_branch = true;
        if (n.refl_mix != 0.0f) { out.any_reflective = true; att_min = std::min(att_min, n.refl_atenuation); }
        NodeRec r;
        r.refl_mix = n.refl_mix; r.refl_atenuation = n.refl_atenuation; r.alpha = n.alpha; r.material_id = n.material_id;
        r.refr_coeff = n.refr_coeff; r.pad[0] = has_uv ? 1u : 0u; r.pad[1] = 0;
        out.node_recs.push_back(r);
        {
            ShadeRec sr; std::memset(&sr, 0, sizeof sr);
            sr.refl_mix = n.refl_mix; sr.refl_atenuation = n.refl_atenuation; sr.alpha = n.alpha; sr.refr_coeff = n.refr_coeff;
            sr.flags = (has_uv ? 1u : 0u) | (m.kind << 8);
            for (int a = 0; a < 3; ++a) { sr.ka[a] = m.ambiant[a]; sr.kd[a] = m.diffuse[a]; sr.ks[a] = m.specular[a]; }
            sr.shininess = m.shininess;
            auto fill = [&](ShadeTex& t, int32_t id) {
                if (id < 0) return;
                const NraysTexture& x = d->textures[id];
                t.width = x.width; t.height = x.height; t.mode = x.format | (x.interp << 8) | (x.overflow << 16);
            };
            fill(sr.tex, m.texture_id); fill(sr.alpha_tex, m.alpha_texture_id);
            out.shade.push_back(sr);
            out.shade_tex.push_back(m.texture_id < 0 ? -1 : m.texture_id);
            out.shade_alpha_tex.push_back(m.alpha_texture_id < 0 ? -1 : m.alpha_texture_id);
        }
        {
            float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
            // A mesh large enough for the device builder gets its local AABB from that build (k_tri_records reduces the same f32 min / max over the faces' vertices and checks
            // the indices): the host scan of the hairball stand-in's 8.6 M indices — random vertex reads — took 8 of its 100 ms.  Filled in below, once its BLAS exists.
            if (n.shape_kind == NRAYS_SHAPE_TRIMESH && d->meshes[n.mesh_id].num_triangles >= device_build_min()) { // (NRAYS_GPU_BUILD=0: the minimum is SIZE_MAX)
                pending_aabb.push_back(i);
                out.node_aabbs.insert(out.node_aabbs.end(), 6, 0.0);
                continue;
            }
            if (n.shape_kind == NRAYS_SHAPE_TRIMESH) { // local AABB = union of the faces' vertices (TriMesh BVT root)
                const NraysMesh& m = d->meshes[n.mesh_id];
                for (int a = 0; a < 3; ++a) { mn[a] = std::numeric_limits<float>::infinity(); mx[a] = -std::numeric_limits<float>::infinity(); }
                for (uint32_t t = 0; t < m.num_triangles * 3u; ++t) {
                    uint32_t vi = m.indices[t];
                    if (vi >= m.num_vertices) { err = "triangle index out of range"; return NRAYS_ERR_BAD_ARG; }
                    for (int a = 0; a < 3; ++a) { float f = (float)m.vertices[3 * (size_t)vi + a]; mn[a] = std::min(mn[a], f); mx[a] = std::max(mx[a], f); }
                }
            }
            double bb[6];
            node_world_aabb(n, info[i].R, mn, mx, bb);
            out.node_aabbs.insert(out.node_aabbs.end(), bb, bb + 6);
        }
    }
    if (out.any_reflective) {
        uint32_t k = 0; float e = 1.0f;
        if (!(att_min > 0.0f)) k = kMaxGenerations;
        else while (e > 0.1f && k < (uint32_t)kMaxGenerations) { ++k; e = e - att_min; }
        out.reflection_generations = k;
    }

    std::vector<Instance> cinst, sinst, planes_c, planes_s;
    std::vector<PrimBounds> cbox, sbox;

    auto base_instance = [&](uint32_t ni) {
        const NraysNode& n = d->nodes[ni];
        Instance in; std::memset(&in, 0, sizeof in);
        for (int k = 0; k < 9; ++k) in.rot[k] = info[ni].R[k];
        for (int k = 0; k < 3; ++k) { in.trans[k] = n.translation[k]; in.params[k] = n.params[k]; }
        in.kind = n.shape_kind;
        in.flags = (n.solid ? kInstSolid : 0u) | (info[ni].identity ? kInstIdentityRot : 0u) | (info[ni].has_uv ? kInstHasUv : 0u) | (info[ni].no_uv_values ? kInstNoUvValues : 0u);
        if (info[ni].identity && n.translation[0] == 0.0 && n.translation[1] == 0.0 && n.translation[2] == 0.0) in.flags |= kInstNoXform;
        in.node_id = (int32_t)ni; in.blas_root = kEmptyChild;
        return in;
    };

    // analytic shapes
    for (uint32_t i = 0; i < d->num_nodes; ++i) {
        const NraysNode& n = d->nodes[i];
        if (n.shape_kind == NRAYS_SHAPE_TRIMESH) continue;
        Instance in = base_instance(i);
        Instance si = in;
        if (info[i].opaque) si.flags |= kInstAnyHit;
        if (n.shape_kind == NRAYS_SHAPE_PLANE) { planes_c.push_back(in); planes_s.push_back(si); continue; }
        double c[3] = {0, 0, 0}, h[3];
        double Rid[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        const double* R = info[i].R;
        switch (n.shape_kind) {
        case NRAYS_SHAPE_BALL: h[0] = h[1] = h[2] = n.params[0]; R = Rid; break; // rotation ignored (SURVEY B-4)
        case NRAYS_SHAPE_CUBOID: h[0] = n.params[0]; h[1] = n.params[1]; h[2] = n.params[2]; break;
        case NRAYS_SHAPE_CAPSULE: h[0] = h[2] = n.params[1]; h[1] = n.params[0] + n.params[1]; break;
        default: h[0] = h[2] = n.params[1]; h[1] = n.params[0]; break; // cylinder, cone
        }
        for (int a = 0; a < 3; ++a) h[a] = std::fabs(h[a]);
        PrimBounds b = world_box(R, n.translation, c, h);
        cinst.push_back(in); cbox.push_back(b);
        sinst.push_back(si); sbox.push_back(b);
    }

    // TriMesh nodes grouped by identical isometry
    std::map<std::vector<uint64_t>, std::vector<uint32_t>> groups;
    std::vector<std::vector<uint64_t>> group_order;
    for (uint32_t i = 0; i < d->num_nodes; ++i) {
        const NraysNode& n = d->nodes[i];
        if (n.shape_kind != NRAYS_SHAPE_TRIMESH) continue;
        if (d->meshes[n.mesh_id].num_triangles == 0) continue;
        std::vector<uint64_t> key(6);
        std::memcpy(&key[0], n.translation, 24); std::memcpy(&key[3], n.axis_angle, 24);
        if (!groups.count(key)) group_order.push_back(key);
        groups[key].push_back(i);
    }
    for (const auto& key : group_order) {
        const std::vector<uint32_t>& ids = groups[key];
        uint32_t n0 = ids[0];
        auto add = [&](const std::vector<uint32_t>& sub, bool closest, bool shadow, bool anyhit, const Blas* reuse, Blas& blas) -> int {
            if (!reuse) { int rc = append_blas(d, sub, out, blas, err); if (rc != NRAYS_OK) return rc; } else blas = *reuse;
            if (!reuse && sub.size() == 1 && blas.device) single_bounds[sub[0]] = std::array<float, 6>{blas.mn[0], blas.mn[1], blas.mn[2], blas.mx[0], blas.mx[1], blas.mx[2]};
            Instance in = base_instance(n0);
            in.flags &= ~(uint32_t)kInstSolid; // TriMesh ignores `solid` (SURVEY B-8)
            in.node_id = sub.size() == 1 ? (int32_t)sub[0] : -1;
            in.blas_root = blas.root;
            if (blas.device) in.flags |= kInstDeviceTmp;
            if (blas.hairy) { in.flags |= kInstIncoherent; out.any_incoherent = true; }
            double c[3], h[3];
            for (int a = 0; a < 3; ++a) { c[a] = 0.5 * ((double)blas.mn[a] + (double)blas.mx[a]); h[a] = 0.5 * ((double)blas.mx[a] - (double)blas.mn[a]); }
            PrimBounds b = world_box(info[n0].R, d->nodes[n0].translation, c, h);
            if (closest) { cinst.push_back(in); cbox.push_back(b); }
            if (shadow) { Instance si = in; if (anyhit) si.flags |= kInstAnyHit; sinst.push_back(si); sbox.push_back(b); }
            return NRAYS_OK;
        };
        std::vector<uint32_t> opaque, transp;
        for (uint32_t i : ids) (info[i].opaque ? opaque : transp).push_back(i);
        Blas all; int rc;
        if (transp.empty()) {
            rc = add(ids, true, true, true, nullptr, all); if (rc != NRAYS_OK) return rc;
        } else {
            rc = add(ids, true, false, false, nullptr, all); if (rc != NRAYS_OK) return rc;
            Blas tmp;
            if (!opaque.empty()) { rc = add(opaque, false, true, true, nullptr, tmp); if (rc != NRAYS_OK) return rc; }
            for (uint32_t i : transp) { rc = add(std::vector<uint32_t>{i}, false, true, false, nullptr, tmp); if (rc != NRAYS_OK) return rc; }
        }
    }

    for (uint32_t i : pending_aabb) { // the local AABBs left open above: from the node's own device-built BLAS, else (merged groups, host fall-back) by the scan
        const NraysNode& n = d->nodes[i];
        const NraysMesh& m = d->meshes[n.mesh_id];
        float mn[3], mx[3];
        const auto it = single_bounds.find(i);
        if (it != single_bounds.end()) { for (int a = 0; a < 3; ++a) { mn[a] = it->second[a]; mx[a] = it->second[3 + a]; } }
        else {
            for (int a = 0; a < 3; ++a) { mn[a] = std::numeric_limits<float>::infinity(); mx[a] = -std::numeric_limits<float>::infinity(); }
            for (uint32_t t = 0; t < m.num_triangles * 3u; ++t) {
                const uint32_t vi = m.indices[t];
                if (vi >= m.num_vertices) { err = "triangle index out of range"; return NRAYS_ERR_BAD_ARG; }
                for (int a = 0; a < 3; ++a) { const float f = (float)m.vertices[3 * (size_t)vi + a]; mn[a] = std::min(mn[a], f); mx[a] = std::max(mx[a], f); }
            }
        }
        double bb[6];
        node_world_aabb(n, info[i].R, mn, mx, bb);
        for (int a = 0; a < 6; ++a) out.node_aabbs[6 * (size_t)i + a] = bb[a];
    }

    {   // kernel permutation: 1 = analytic shapes, 2 = meshes, 4 = some node may be non-opaque to shadow rays
        int f = 0;
        for (uint32_t i = 0; i < d->num_nodes; ++i) {
            const NraysNode& n = d->nodes[i];
            if (n.shape_kind == NRAYS_SHAPE_TRIMESH) { if (d->meshes[n.mesh_id].num_triangles) f |= 2; } else f |= 1;
            if (!info[i].opaque) f |= 4;
        }
        if (f == 4 || f == 0) f |= 1; // empty scenes take the lightest kernel
        out.any_mesh = (f & 2) != 0;
        if (out.any_double_branch) f = 15; // reflection + refraction at one hit: full kernel with the HBM queue
        if (!(d->num_lights == 1 && d->lights[0].racsample == 1)) f |= 16; // more than one light sample per hit
        out.features = f;
    }
    const size_t host_blas_nodes = out.nodes.size(); // what follows are TLAS nodes
    out.bounded = planes_c.empty();
    for (int a = 0; a < 3; ++a) { out.bounds_mn[a] = std::numeric_limits<float>::infinity(); out.bounds_mx[a] = -std::numeric_limits<float>::infinity(); }
    for (const PrimBounds& b : cbox) for (int a = 0; a < 3; ++a) { out.bounds_mn[a] = std::min(out.bounds_mn[a], b.mn[a]); out.bounds_mx[a] = std::max(out.bounds_mx[a], b.mx[a]); }
    out.closest_root = append_tlas(cinst, cbox, out);
    out.shadow_root = append_tlas(sinst, sbox, out);
    for (size_t k = 0; k < planes_c.size(); ++k) {
        out.planes.push_back((int32_t)cinst.size()); cinst.push_back(planes_c[k]);
        out.shadow_planes.push_back((int32_t)sinst.size()); sinst.push_back(planes_s[k]);
    }
    if (out.dev_nodes || out.dev_tris) {
        // Device-built BLASes come FIRST in the scene's node / triangle arrays (their refs were absolute from the start); everything the
        // host built moves behind them: BLAS nodes (child and leaf refs), TLAS nodes (child refs; their leaves are instance indices), roots.
        const int32_t dn = (int32_t)out.dev_nodes; const uint32_t dt = (uint32_t)out.dev_tris;
        for (size_t i = 0; i < out.nodes.size(); ++i)
            for (int k = 0; k < 4; ++k) { int32_t& c = out.nodes[i].children()[k]; c = i < host_blas_nodes ? rebase_ref(c, dn, dt) : (c >= 0 ? c + dn : c); }
        if (out.closest_root >= 0) out.closest_root += dn;
        if (out.shadow_root >= 0) out.shadow_root += dn;
        for (std::vector<Instance>* list : {&cinst, &sinst})
            for (Instance& in : *list) {
                if (in.kind != NRAYS_SHAPE_TRIMESH) continue;
                if (in.flags & kInstDeviceTmp) in.flags &= ~(uint32_t)kInstDeviceTmp; else in.blas_root = rebase_ref(in.blas_root, dn, dt);
            }
        if (out.tris.size() + out.dev_tris >= (1u << 28)) { err = "too many triangles"; return NRAYS_ERR_UNSUPPORTED; }
    }
    out.instances.swap(cinst);
    out.shadow_instances.swap(sinst);
    for (const Instance& in : out.instances) out.links.push_back(InstLink{in.blas_root, in.flags});
    for (const Instance& in : out.shadow_instances) out.shadow_links.push_back(InstLink{in.blas_root, in.flags});
    // the traversal addresses a node as base + 32-bit byte offset (trace_device.h: load_planes)
    if (out.nodes.size() + out.dev_nodes >= ((size_t)1 << 25)) { err = "scene too large: more than 2^25 BVH nodes"; return NRAYS_ERR_BAD_ARG; }
    return NRAYS_OK;
}

} // namespace nrays
