// bvh_build.cpp — binned-SAH BVH2 builder (host).  See bvh_build.h.
#include "bvh_build.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <future>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <thread>

namespace nrays {

float round_down_f32(double v) {
    float f = (float)v;
    if ((double)f > v) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
    return f;
}
float round_up_f32(double v) {
    float f = (float)v;
    if ((double)f < v) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
    return f;
}

namespace {

struct Box {
    float mn[3], mx[3];
    void reset() {
        for (int a = 0; a < 3; ++a) { mn[a] = std::numeric_limits<float>::infinity(); mx[a] = -std::numeric_limits<float>::infinity(); }
    }
    void grow(const float* lo, const float* hi) {
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], lo[a]); mx[a] = std::max(mx[a], hi[a]); }
    }
    void grow(const Box& b) { grow(b.mn, b.mx); }
    float half_area() const {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (!(dx >= 0.f) || !(dy >= 0.f) || !(dz >= 0.f)) return 0.f;
        return dx * dy + dy * dz + dz * dx;
    }
};

constexpr int kBins = kSahBins;
constexpr float kPrimCost = NR_PRIM_COST;

struct Node2 { // binary node produced by the SAH build, collapsed into 4-wide BvhNodes afterwards
    float lmin[3], lmax[3], rmin[3], rmax[3];
    int32_t left, right;
};

// Large subtrees of the top `kParLevels` levels are built by their own thread into a local node array and
// spliced into the parent's array afterwards (child indices shifted by the splice offset): the tree is the one the
// sequential build produces, only the order of the nodes in memory differs.  The partition step works on disjoint
// ranges of `order`, everything else a task touches is read-only or its own.
constexpr int kParLevels = 6;          // up to 64 concurrent subtrees (8 / 30 000 measured: the splices cost more than the extra tasks save, 1.11 -> 1.36 s)
constexpr uint32_t kParMinCount = 100000;
constexpr uint32_t kParScanMin = 400000; // nodes above this many primitives scan them with several threads

struct Builder {
    const std::vector<PrimBounds>& prims;
    std::vector<uint32_t>& order;
    std::vector<Node2>& nodes;
    const std::vector<float>* cent_ptr; // 3 per prim (owned by the root builder)
    std::vector<float> cent_own;
    int max_leaf;
    float prim_cost = kPrimCost; // an exact f64 ray / triangle test in units of one node visit (SAH leaf criterion)
    int max_depth = 0;

    Builder(const std::vector<PrimBounds>& p, std::vector<uint32_t>& o, std::vector<Node2>& n, int ml)
        : prims(p), order(o), nodes(n), max_leaf(ml) {
        cent_own.resize(p.size() * 3);
        for (size_t i = 0; i < p.size(); ++i)
            for (int a = 0; a < 3; ++a) cent_own[3 * i + a] = 0.5f * p[i].mn[a] + 0.5f * p[i].mx[a];
        cent_ptr = &cent_own;
    }
    Builder(const Builder& parent, std::vector<Node2>& local) // a task's builder: shares everything but the node array
        : prims(parent.prims), order(parent.order), nodes(local), cent_ptr(parent.cent_ptr), max_leaf(parent.max_leaf) { prim_cost = parent.prim_cost; }

    // Builds the subtree over order[first, first+count); returns its ref and bounds.
    int32_t build(uint32_t first, uint32_t count, Box& bounds, int depth) {
        const std::vector<float>& cent = *cent_ptr;
        max_depth = std::max(max_depth, depth);
        // Large nodes scan their primitives with several threads (bounds, then the bins of the three axes: minima, maxima and counts
        // merge exactly, so the tree is the sequential one): the top of a 15 M-reference hair tree was 1.4 s of single-threaded binning.
        static const unsigned hw = std::max(1u, std::thread::hardware_concurrency()); // (a system call: once, not per node)
        const uint32_t chunks = (count >= kParScanMin && depth < 8) ? std::min<uint32_t>(std::max(1u, (hw >> depth)), 32u) : 1u;
        auto for_chunks = [&](auto&& fn) { // fn(chunk index, first, last)
            if (chunks <= 1u) { fn(0u, first, first + count); return; }
            std::vector<std::future<void>> futs;
            for (uint32_t c = 1; c < chunks; ++c) {
                const uint32_t lo = first + (uint32_t)((uint64_t)count * c / chunks), hi = first + (uint32_t)((uint64_t)count * (c + 1) / chunks);
                futs.push_back(std::async(std::launch::async, [&fn, c, lo, hi]() { fn(c, lo, hi); }));
            }
            fn(0u, first, first + (uint32_t)((uint64_t)count / chunks));
            for (auto& f : futs) f.get();
        };
        bounds.reset();
        Box cb; cb.reset();
        {
            Box pb0, pc0; std::vector<Box> pbx(chunks - 1u), pcx(chunks - 1u); // (no heap traffic in the millions of small nodes)
            for_chunks([&](uint32_t c, uint32_t lo, uint32_t hi) {
                Box b, cc; b.reset(); cc.reset();
                for (uint32_t i = lo; i < hi; ++i) {
                    const PrimBounds& p = prims[order[i]];
                    b.grow(p.mn, p.mx);
                    const float* ctr = &cent[3 * order[i]];
                    cc.grow(ctr, ctr);
                }
                (c ? pbx[c - 1u] : pb0) = b; (c ? pcx[c - 1u] : pc0) = cc;
            });
            bounds.grow(pb0); cb.grow(pc0);
            for (uint32_t c = 1; c < chunks; ++c) { bounds.grow(pbx[c - 1u]); cb.grow(pcx[c - 1u]); }
        }
        if (count <= 1) return make_leaf_ref(first, count);

        // binned SAH over the three axes
        float best_cost = std::numeric_limits<float>::infinity();
        int best_axis = -1, best_split = -1;
        struct Bins { Box bb[3][kBins]; uint32_t bc[3][kBins]; };
        Bins bins0; std::vector<Bins> binsx(chunks - 1u);
        auto bins_of = [&](uint32_t c) -> Bins& { return c ? binsx[c - 1u] : bins0; };
        float lo3[3], scale3[3]; bool use3[3];
        for (int axis = 0; axis < 3; ++axis) { lo3[axis] = cb.mn[axis]; use3[axis] = cb.mx[axis] > cb.mn[axis]; scale3[axis] = use3[axis] ? (float)kBins / (cb.mx[axis] - cb.mn[axis]) : 0.0f; }
        for_chunks([&](uint32_t c, uint32_t lo, uint32_t hi) {
            Bins& B = bins_of(c);
            for (int axis = 0; axis < 3; ++axis) for (int b = 0; b < kBins; ++b) { B.bb[axis][b].reset(); B.bc[axis][b] = 0; }
            for (uint32_t i = lo; i < hi; ++i) {
                const uint32_t id = order[i];
                for (int axis = 0; axis < 3; ++axis) {
                    if (!use3[axis]) continue;
                    int b = (int)((cent[3 * id + axis] - lo3[axis]) * scale3[axis]);
                    b = std::min(std::max(b, 0), kBins - 1);
                    B.bb[axis][b].grow(prims[id].mn, prims[id].mx); B.bc[axis][b]++;
                }
            }
        });
        for (int axis = 0; axis < 3; ++axis) {
            if (!use3[axis]) continue;
            Box bb[kBins]; uint32_t bc[kBins];
            for (int b = 0; b < kBins; ++b) { bb[b] = bins0.bb[axis][b]; bc[b] = bins0.bc[axis][b]; }
            for (uint32_t c = 1; c < chunks; ++c) for (int b = 0; b < kBins; ++b) { bb[b].grow(binsx[c - 1u].bb[axis][b]); bc[b] += binsx[c - 1u].bc[axis][b]; }
            float right_area[kBins]; uint32_t right_cnt[kBins];
            Box acc; acc.reset(); uint32_t cnt = 0;
            for (int b = kBins - 1; b > 0; --b) { acc.grow(bb[b]); cnt += bc[b]; right_area[b] = acc.half_area(); right_cnt[b] = cnt; }
            acc.reset(); cnt = 0;
            for (int b = 0; b < kBins - 1; ++b) {
                acc.grow(bb[b]); cnt += bc[b];
                if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                float cost = acc.half_area() * (float)cnt + right_area[b + 1] * (float)right_cnt[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = b; }
            }
        }
        if ((int)count <= max_leaf) {
            // SAH: an exact f64 ray/triangle test costs about kPrimCost times a (two-box, f32) node visit
            float leaf_cost = bounds.half_area() * (float)count * prim_cost;
            float split_cost = best_axis < 0 ? std::numeric_limits<float>::infinity() : best_cost * prim_cost + bounds.half_area() * 1.0f;
            if (!(split_cost < leaf_cost)) return make_leaf_ref(first, count);
        }
        uint32_t mid;
        if (best_axis < 0) {
            mid = first + count / 2; // all centroids coincide: split by index
        } else {
            float lo = cb.mn[best_axis], hi = cb.mx[best_axis];
            float scale = (float)kBins / (hi - lo);
            auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t id) {
                int b = (int)((cent[3 * id + best_axis] - lo) * scale);
                b = std::min(std::max(b, 0), kBins - 1);
                return b <= best_split;
            });
            mid = (uint32_t)(it - order.begin());
            if (mid == first || mid == first + count) mid = first + count / 2;
        }
        int32_t me = (int32_t)nodes.size();
        nodes.emplace_back();
        Box lb, rb;
        int32_t l, r;
        if (depth < kParLevels && mid - first >= kParMinCount && first + count - mid >= kParMinCount) {
            std::vector<Node2> local;
            Builder lbuilder(*this, local);
            auto fut = std::async(std::launch::async, [&]() { return lbuilder.build(first, mid - first, lb, depth + 1); });
            r = build(mid, first + count - mid, rb, depth + 1);
            l = fut.get();
            max_depth = std::max(max_depth, lbuilder.max_depth);
            const int32_t off = (int32_t)nodes.size();
            for (Node2 ln : local) {
                if (ln.left >= 0) ln.left += off;
                if (ln.right >= 0) ln.right += off;
                nodes.push_back(ln);
            }
            if (l >= 0) l += off;
        } else {
            l = build(first, mid - first, lb, depth + 1);
            r = build(mid, first + count - mid, rb, depth + 1);
        }
        Node2& n = nodes[me];
        for (int a = 0; a < 3; ++a) { n.lmin[a] = lb.mn[a]; n.lmax[a] = lb.mx[a]; n.rmin[a] = rb.mn[a]; n.rmax[a] = rb.mx[a]; }
        n.left = l; n.right = r;
        return me;
    }
};

// Collapses the binary tree rooted at binary node `n2` into 4-wide nodes; returns the new node index.
struct Collapser {
    const std::vector<Node2>& n2;
    std::vector<BvhNode>& out;
    int max_depth = 0;
    struct Slot { float mn[3], mx[3]; int32_t ref; };
    static float area(const Slot& s) {
        float dx = s.mx[0] - s.mn[0], dy = s.mx[1] - s.mn[1], dz = s.mx[2] - s.mn[2];
        return dx * dy + dy * dz + dz * dx;
    }
    int32_t collapse(int32_t root, int depth) {
        max_depth = std::max(max_depth, depth);
        std::vector<Slot> slots;
        auto push_children = [&](int32_t node) {
            const Node2& b = n2[node];
            Slot l, r;
            for (int a = 0; a < 3; ++a) { l.mn[a] = b.lmin[a]; l.mx[a] = b.lmax[a]; r.mn[a] = b.rmin[a]; r.mx[a] = b.rmax[a]; }
            l.ref = b.left; r.ref = b.right;
            slots.push_back(l); slots.push_back(r);
        };
        push_children(root);
        while (slots.size() < 4) { // open the internal child with the largest box
            int best = -1; float ba = -1.f;
            for (size_t k = 0; k < slots.size(); ++k) if (slots[k].ref >= 0 && area(slots[k]) > ba) { ba = area(slots[k]); best = (int)k; }
            if (best < 0) break;
            int32_t node = slots[best].ref;
            slots.erase(slots.begin() + best);
            push_children(node);
        }
        int32_t me = (int32_t)out.size();
        out.emplace_back();
        std::vector<int32_t> refs(slots.size());
        for (size_t k = 0; k < slots.size(); ++k) refs[k] = slots[k].ref >= 0 ? collapse(slots[k].ref, depth + 1) : slots[k].ref;
        BvhNode& n = out[me];
        for (int k = 0; k < 4; ++k) {
            if (k < (int)slots.size()) {
                n.set_box(k, slots[k].mn, slots[k].mx);
                n.children()[k] = refs[k];
            } else { // finite (a zero inverse direction times infinity would be a NaN) and inverted: no ray enters it
                const float big = std::numeric_limits<float>::max();
                const float mn[3] = {big, big, big}, mx[3] = {-big, -big, -big};
                n.set_box(k, mn, mx);
                n.children()[k] = kEmptyChild;
            }
            n.slot[5][k] = 0.0f;
        }
        return me;
    }
};

} // namespace

BuiltBvh build_bvh(const std::vector<PrimBounds>& prims, int max_leaf, float prim_cost) {
    BuiltBvh out;
    out.max_depth = 0;
    out.order.resize(prims.size());
    for (size_t i = 0; i < prims.size(); ++i) out.order[i] = (uint32_t)i;
    if (prims.empty()) { out.root = kEmptyChild; return out; }
    max_leaf = std::min(std::max(max_leaf, 1), 8);
    const bool verbose = prims.size() > 1000000 && getenv("NRAYS_BUILD_TIMES");
    auto T0 = std::chrono::steady_clock::now();
    std::vector<Node2> binary;
    binary.reserve(prims.size());
    Builder b(prims, out.order, binary, max_leaf);
    if (prim_cost > 0.0f) b.prim_cost = prim_cost;
    Box bounds;
    auto T1 = std::chrono::steady_clock::now();
    int32_t root2 = b.build(0, (uint32_t)prims.size(), bounds, 0);
    auto T2 = std::chrono::steady_clock::now();
    if (root2 < 0) { out.root = root2; return out; } // a single leaf
    Collapser c{binary, out.nodes};
    out.root = c.collapse(root2, 0);
    out.max_depth = c.max_depth;
    auto T3 = std::chrono::steady_clock::now();
    if (verbose) fprintf(stderr, "  build_bvh: setup %.2f s, binary build %.2f s (%zu nodes), collapse %.2f s (%zu nodes)\n", std::chrono::duration<double>(T1 - T0).count(),
                         std::chrono::duration<double>(T2 - T1).count(), binary.size(), std::chrono::duration<double>(T3 - T2).count(), out.nodes.size());
    return out;
}

void rebase_bvh(BuiltBvh& bvh, int32_t node_base, uint32_t prim_base) {
    for (BvhNode& n : bvh.nodes)
        for (int k = 0; k < 4; ++k) n.children()[k] = rebase_ref(n.children()[k], node_base, prim_base);
    bvh.root = rebase_ref(bvh.root, node_base, prim_base);
}

} // namespace nrays
