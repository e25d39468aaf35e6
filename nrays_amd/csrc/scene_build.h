// scene_build.h — flattens an NraysSceneDesc (the data Scene::new / SceneNode::new receive,
// src/scene.rs:119, src/scene_node.rs:22) into the HBM layout of device_types.h.  Host only.
#pragma once
#include <string>
#include <vector>

#include "../../include/nrays_abi.h"
#include "bvh_device.h"
#include "device_types.h"

namespace nrays {

struct HostTexture {
    TextureRec rec;               // texels pointer filled at upload
    std::vector<uint8_t> bytes;   // copy of the texel data
};

struct HostScene {
    // BLASes built on the device (bvh_device.hip) occupy the FIRST dev_nodes / dev_tris entries of the scene's node / triangle arrays,
    // in the order of dev_blas; the host-built arrays below follow them (their refs are shifted when the scene is complete).
    std::vector<DeviceBlas> dev_blas;
    size_t dev_nodes = 0, dev_tris = 0;
    std::vector<BvhNode> nodes;
    std::vector<TriRec> tris;
    std::vector<TriUv> triuvs;
    std::vector<Instance> instances;         // closest TLAS order, planes appended at the end
    std::vector<Instance> shadow_instances;  // shadow TLAS order, planes appended at the end
    std::vector<InstLink> links, shadow_links; // parallel to instances / shadow_instances
    std::vector<int32_t> planes, shadow_planes;
    std::vector<NodeRec> node_recs;
    std::vector<ShadeRec> shade;             // per node; texel pointers patched after the texture upload
    std::vector<int32_t> shade_tex, shade_alpha_tex; // texture indices behind shade[i].tex / alpha_tex (-1 = none)
    std::vector<double> node_aabbs;          // 6 per node (mins, maxs), reference arithmetic
    std::vector<MaterialRec> materials;
    std::vector<HostTexture> textures;
    std::vector<LightRec> lights;
    int32_t closest_root = kEmptyChild, shadow_root = kEmptyChild;
    float background[3] = {1.f, 1.f, 1.f};
    // generation control (see nrays_hip.hip)
    bool any_reflective = false;   // some node has refl_mix != 0 (scene.rs:204)
    bool any_transparent = false;  // some node can produce alpha != 1 (scene.rs:229)
    bool any_area_light = false;   // some light has radius != 0 (light.rs:60 consumes random numbers)
    bool any_double_branch = false; // some node can spawn BOTH a reflection and a refraction at one hit
    bool any_incoherent = false;    // some BLAS holds a hair-like mesh (DScene::incoherent)
    bool any_mesh = false;          // some TriMesh node has triangles (decides the tile scheduling path)
    uint32_t reflection_generations = 0; // upper bound from the energy rule (scene.rs:204-206)
    int max_bvh_depth = 0;
    // f32 box (rounded outward) around every bounded node = the boxes of the closest-hit TLAS; `bounded` is false when
    // a plane (unbounded) is present.  Used to find the pixels whose primary rays cannot hit anything (screen_bounds()).
    bool bounded = false;
    float bounds_mn[3] = {0, 0, 0}, bounds_mx[3] = {0, 0, 0};
    int features = 0;              // Features bits of trace_device.h: which kernel permutation renders this scene
};

// Returns NRAYS_OK or a negative NraysStatus with `err` set.
int build_host_scene(const NraysSceneDesc* desc, HostScene& out, std::string& err);

// Test probe behind nrays_debug_blas_build: the BLAS of ONE mesh from the host builder or the device builder, copied to the host.
// nodes: local indices (root = node 0 unless the BLAS is a single leaf), tri_ids: TriRec::tri_id per leaf slot.
struct BlasProbe { std::vector<BvhNode> nodes; std::vector<uint32_t> tri_ids; int32_t root = kEmptyChild; int max_depth = 0; bool hairy = false; };
int build_blas_probe(const NraysMesh* mesh, bool device, bool presplit_on, BlasProbe& out, std::string& err);

} // namespace nrays
