// bvh_build.h — host-side binned-SAH builder (binary build, then collapse) producing the 128-byte
// 4-wide node layout of device_types.h.  Replaces ncollide's BVT::new_balanced (called at src/scene.rs:126 and inside
// TriMesh::new, examples/loader3d.rs:695).  The tree shape differs from the reference's median
// split on purpose: closest-hit / shadow results do not depend on the tree, only node-visit
// counts do (DESIGN.md reports both trees' counts).
#pragma once
#include <cstdint>
#include <vector>

#include "device_types.h"

#ifndef NR_SAH_BINS
#define NR_SAH_BINS 32 // 16 -> 32: hairball -1 %, sponza -0.4 %, same build time
#endif
#ifndef NR_PRIM_COST
#define NR_PRIM_COST 1.0f // round 5, three waves per SIMD and the frames bound by their sum of tile cycles (0.5 before, 0.7 before that; hair-like meshes have their own): with NR_PRESPLIT_MINGAIN 1.0 sponza 1.052 -> 1.039 ms, 8 lights 2.61 -> 2.56 (3.7 / 4.0 instead of 4.9 triangle tests per ray; profiles/r05_build_knobs_sweep.log)
#endif

#ifdef __HIPCC__
#define NR_HD __host__ __device__
#else
#define NR_HD
#endif

namespace nrays {

constexpr int kSahBins = NR_SAH_BINS; // shared with the device builder (bvh_device.hip), which is written for 32

struct PrimBounds {
    float mn[3], mx[3];
};

struct BuiltBvh {
    std::vector<BvhNode> nodes;  // local indices (0-based); caller rebases when concatenating
    std::vector<uint32_t> order; // order[k] = original primitive index stored at leaf slot k
    int32_t root;                // >= 0 node, < 0 leaf ref, kEmptyChild when there are no primitives
    int max_depth;
};

// Leaf ref encoding: ~((first << 3) | (count - 1)), count in [1, 8].
NR_HD inline int32_t make_leaf_ref(uint32_t first, uint32_t count) { return ~(int32_t)((first << 3) | (count - 1)); }

// prim_cost: cost of testing one primitive in units of a node visit (SAH leaf criterion); 0 = the default (NR_PRIM_COST)
BuiltBvh build_bvh(const std::vector<PrimBounds>& prims, int max_leaf, float prim_cost = 0.0f);

// Adds `node_base` to internal child indices and `prim_base` to leaf `first` fields.
void rebase_bvh(BuiltBvh& bvh, int32_t node_base, uint32_t prim_base);
inline int32_t rebase_ref(int32_t ref, int32_t node_base, uint32_t prim_base) {
    if (ref == kEmptyChild) return ref;
    if (ref >= 0) return ref + node_base;
    uint32_t v = (uint32_t)~ref;
    return make_leaf_ref((v >> 3) + prim_base, (v & 7u) + 1u);
}

float round_down_f32(double v); // largest f32 <= v
float round_up_f32(double v);   // smallest f32 >= v

} // namespace nrays
