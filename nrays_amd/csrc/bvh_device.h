// bvh_device.h — device-side BLAS build for large TriMesh groups (bvh_device.hip): triangle records, pre-splitting, the binned-SAH
// binary build, the 4-wide collapse in the host builder's depth-first layout and the leaf-ordered triangle copies all run on the
// GPU.  Same split rule, same arithmetic as bvh_build.cpp / scene_build.cpp (the tree is the one the host builder produces from
// the same references); replaces ncollide's BVT::new_balanced inside TriMesh::new (examples/loader3d.rs:695) like they do.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "device_types.h"

namespace nrays {

struct DeviceMeshPart { // one TriMesh node of the group sharing a BLAS
    const double* vertices; // host, 3 * num_vertices
    const double* uvs;      // host, 2 * num_vertices, or null
    const uint32_t* indices; // host, 3 * num_triangles
    uint32_t num_vertices, num_triangles;
    uint32_t node_id;
};

struct DeviceBlas {
    BvhNode* nodes = nullptr; size_t num_nodes = 0; // device memory; child refs already absolute (node_base / prim_base applied)
    size_t node_capacity = 0;                        // nodes the allocation holds (>= num_nodes: room for the scene's host-built nodes behind them)
    TriRec* tris = nullptr; TriUv* uvs = nullptr; size_t num_refs = 0; // device memory, leaf order
    int32_t root = kEmptyChild;
    float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0}; // local bounds of the triangles
    bool hairy = false;
    int max_depth = 0;
    size_t num_triangles = 0;
};

struct DeviceBuildOptions {
    int max_leaf = 8;
    float prim_cost = 1.0f, prim_cost_hairy = 0.7f;
    double budget = 1.0, budget_hairy = 5.0, min_gain = 1.0, min_gain_hairy = 0.02, hairy_emptiness = 0.9;
    bool presplit = true;
    int debug_cap_div = 1;    // NRAYS_DEBUG_BUILD_CAPS=n (tests): the builder's internal lists get 1 / n of their capacity, so that their overflow path — an error the caller answers with the host builder — runs
    size_t node_tail = 16384; // spare node slots behind the BLAS: a scene with ONE device-built BLAS adopts its arrays as they are (no second copy of GBs)
};

// Builds the BLAS of `parts` on the current device.  Returns NRAYS_OK or a negative NraysStatus with `err` set.
// node_base / prim_base: position of this BLAS's first node / first triangle slot in the scene's final arrays.
int build_blas_device(const std::vector<DeviceMeshPart>& parts, const DeviceBuildOptions& opt, int32_t node_base, uint32_t prim_base,
                      DeviceBlas& out, std::string& err);
void free_device_blas(DeviceBlas& b);

// Test probe (nrays_debug_blas_build): copies a device-built BLAS back to the host.
struct HostBlasCopy { std::vector<BvhNode> nodes; std::vector<TriRec> tris; int32_t root; int max_depth; bool hairy; };

} // namespace nrays
