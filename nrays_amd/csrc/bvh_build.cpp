// bvh_build.cpp — binned-SAH BVH2 builder (host).  See bvh_build.h.
#include "bvh_build.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

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

constexpr int kBins = 16;
#ifndef NR_PRIM_COST
#define NR_PRIM_COST 0.7f
#endif
constexpr float kPrimCost = NR_PRIM_COST;

struct Builder {
    const std::vector<PrimBounds>& prims;
    std::vector<uint32_t>& order;
    std::vector<BvhNode>& nodes;
    std::vector<float> cent; // 3 per prim
    int max_leaf;
    int max_depth = 0;

    Builder(const std::vector<PrimBounds>& p, std::vector<uint32_t>& o, std::vector<BvhNode>& n, int ml)
        : prims(p), order(o), nodes(n), max_leaf(ml) {
        cent.resize(p.size() * 3);
        for (size_t i = 0; i < p.size(); ++i)
            for (int a = 0; a < 3; ++a) cent[3 * i + a] = 0.5f * p[i].mn[a] + 0.5f * p[i].mx[a];
    }

    // Builds the subtree over order[first, first+count); returns its ref and bounds.
    int32_t build(uint32_t first, uint32_t count, Box& bounds, int depth) {
        max_depth = std::max(max_depth, depth);
        bounds.reset();
        Box cb; cb.reset();
        for (uint32_t i = first; i < first + count; ++i) {
            const PrimBounds& p = prims[order[i]];
            bounds.grow(p.mn, p.mx);
            const float* c = &cent[3 * order[i]];
            cb.grow(c, c);
        }
        if (count <= 1) return make_leaf_ref(first, count);

        // binned SAH over the three axes
        float best_cost = std::numeric_limits<float>::infinity();
        int best_axis = -1, best_split = -1;
        for (int axis = 0; axis < 3; ++axis) {
            float lo = cb.mn[axis], hi = cb.mx[axis];
            if (!(hi > lo)) continue;
            Box bb[kBins]; uint32_t bc[kBins];
            for (int b = 0; b < kBins; ++b) { bb[b].reset(); bc[b] = 0; }
            float scale = (float)kBins / (hi - lo);
            for (uint32_t i = first; i < first + count; ++i) {
                uint32_t id = order[i];
                int b = (int)((cent[3 * id + axis] - lo) * scale);
                b = std::min(std::max(b, 0), kBins - 1);
                bb[b].grow(prims[id].mn, prims[id].mx); bc[b]++;
            }
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
            float leaf_cost = bounds.half_area() * (float)count * kPrimCost;
            float split_cost = best_axis < 0 ? std::numeric_limits<float>::infinity() : best_cost * kPrimCost + bounds.half_area() * 1.0f;
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
        int32_t l = build(first, mid - first, lb, depth + 1);
        int32_t r = build(mid, first + count - mid, rb, depth + 1);
        BvhNode& n = nodes[me];
        for (int a = 0; a < 3; ++a) { n.lmin[a] = lb.mn[a]; n.lmax[a] = lb.mx[a]; n.rmin[a] = rb.mn[a]; n.rmax[a] = rb.mx[a]; }
        n.left = l; n.right = r; n.pad[0] = n.pad[1] = 0;
        return me;
    }
};

} // namespace

BuiltBvh build_bvh(const std::vector<PrimBounds>& prims, int max_leaf) {
    BuiltBvh out;
    out.max_depth = 0;
    out.order.resize(prims.size());
    for (size_t i = 0; i < prims.size(); ++i) out.order[i] = (uint32_t)i;
    if (prims.empty()) { out.root = kEmptyChild; return out; }
    max_leaf = std::min(std::max(max_leaf, 1), 8);
    out.nodes.reserve(prims.size());
    Builder b(prims, out.order, out.nodes, max_leaf);
    Box bounds;
    out.root = b.build(0, (uint32_t)prims.size(), bounds, 0);
    out.max_depth = b.max_depth;
    return out;
}

void rebase_bvh(BuiltBvh& bvh, int32_t node_base, uint32_t prim_base) {
    for (BvhNode& n : bvh.nodes) {
        n.left = rebase_ref(n.left, node_base, prim_base);
        n.right = rebase_ref(n.right, node_base, prim_base);
    }
    bvh.root = rebase_ref(bvh.root, node_base, prim_base);
}

} // namespace nrays
