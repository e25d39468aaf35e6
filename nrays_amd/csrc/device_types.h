// device_types.h — HBM-resident layout of a flattened nrays scene (shared by host flattening code
// and the gfx950 kernels).  See DESIGN.md §"Data layout in HBM".
#pragma once
#include <stdint.h>

namespace nrays {

// One 4-wide BVH node (128 B = 8 x dwordx4): the boxes of up to four children followed by their refs.
// The tree is built as a binned-SAH binary tree and collapsed (the child with the largest box is
// replaced by its own two children until four slots are used): the traversal is bound by the chain of
// dependent node fetches, and a 4-wide node halves that chain.  One fetch pays for four AABB tests
// (32 B each, the SURVEY 8d unit).  Bounds are f32 rounded OUTWARD from the f64 geometry, so the
// conservative f32 slab test against them is a superset of the reference's f64 test (ncollide
// AABB::toi_with_ray, src/scene.rs:276).
// child >= 0: index of an internal node; child < 0: leaf, ~child = (first << 3) | bits, where bits =
// count - 1 for triangle leaves (BLAS) and LeafBits for TLAS leaves (first = instance index).  An
// absent child has an inverted box (min = +FLT_MAX, max = -FLT_MAX: finite, so that a zero inverse
// direction cannot turn it into a NaN) that no ray can enter, and child = kEmptyChild.
struct BvhNode {
    // Eight 16-byte slots.  One slot = the same plane of all four children (the slab arithmetic runs two children per
    // packed-f32 instruction).  The lower and upper plane of an axis sit at byte offsets that differ in ONE address bit —
    // x: 0 / 16, y: 64 / 96, z: 48 / 112 — so a lane fetches the plane its ray ENTERS through at
    // `lower offset | (direction negative ? that bit : 0)` and the exit plane at that address ^ bit: the near / far
    // selection of the slab test costs no instruction (trace_device.h: load_planes).  Children at 32, slot 80 unused.
    float slot[8][4];
    static constexpr int kLoSlot[3] = {0, 4, 3}, kHiSlot[3] = {1, 6, 7}, kChildSlot = 2;
    void set_box(int k, const float mn[3], const float mx[3]) { for (int a = 0; a < 3; ++a) { slot[kLoSlot[a]][k] = mn[a]; slot[kHiSlot[a]][k] = mx[a]; } }
    int32_t* children() { return reinterpret_cast<int32_t*>(slot[kChildSlot]); }
    const int32_t* children() const { return reinterpret_cast<const int32_t*>(slot[kChildSlot]); }
};
static_assert(sizeof(BvhNode) == 128, "BvhNode must be 128 bytes");
constexpr int32_t kEmptyChild = (int32_t)0x80000000;

// Triangle record, 48 B = 3 x dwordx4 (36 B of vertex data + ids riding in the .w lanes):
//   q0 = (v0.xyz, bits(scene node id)), q1 = (v1.xyz, bits(triangle index inside its TriMesh)),
//   q2 = (v2.xyz, 0).  Vertices are in mesh-local space and f32-EXACT (obj.rs:197-205 parses f32;
//   loader3d.rs:669 divides by 4.0, which is exact), so widening to f64 reproduces the reference's
//   f64 vertex exactly.
struct TriRec {
    float v0[3]; uint32_t node_id;
    float v1[3]; uint32_t tri_id;
    float v2[3]; uint32_t pad;
};
static_assert(sizeof(TriRec) == 48, "TriRec must be 48 bytes");

// Per-triangle-corner texture coordinates (f32-exact, obj.rs parse_vt), fetched only at accepted hits.
struct TriUv { float uv[6]; };

enum InstanceFlags : uint32_t {
    kInstSolid = 1u,         // SceneNode::solid (scene_node.rs:13)
    kInstIdentityRot = 2u,   // rotation is exactly the identity: skip the ray transform
    kInstAnyHit = 4u,        // shadow TLAS only: every node behind this BLAS is opaque (alpha == 1
                             // for every hit), so the first hit within maxtoi blocks (scene.rs:328-330)
    kInstHasUv = 8u,         // the mesh(es) behind this BLAS carry uvs
    kInstNoXform = 16u,      // identity rotation AND zero translation: local space == world space
    kInstNoUvValues = 32u,   // the node's material never reads the VALUES of u, v (NormalMaterial, untextured Phong): a ball skips atan2 / asin
    kInstIncoherent = 64u,   // hair-like mesh (scene_build.cpp: presplit): neighbouring rays walk different nodes (DScene::incoherent)
    kInstDeviceTmp = 0x80000000u // host-side only, while a scene is being flattened: the BLAS was built on the device (bvh_device.hip) and
                             // its root already addresses the final arrays; cleared before the instance is uploaded
};

// Flags carried in the 3 low bits of a TLAS leaf ref (triangle leaves use them as count - 1).
enum LeafBits : uint32_t { kLeafNoXform = 1u, kLeafAnyHit = 2u, kLeafMesh = 4u };

// Compact per-instance link read when a ray enters a BLAS (8 B instead of the 136-B Instance, which is
// only needed for rotated / translated instances and at accepted hits).
struct InstLink {
    int32_t blas_root;
    uint32_t flags; // InstanceFlags
};

// TLAS leaf payload: one analytic shape, or one BLAS (a single TriMesh node, or several TriMesh
// nodes sharing one isometry merged into one BLAS — node ids then ride in TriRec::node_id).
struct Instance {
    double rot[9];   // R, local -> world, row-major (Isometry3::new, loader3d.rs:552)
    double trans[3];
    double params[3];
    uint32_t kind;   // NraysShapeKind
    uint32_t flags;  // InstanceFlags
    int32_t node_id; // scene node, or -1 when the BLAS merges several nodes
    int32_t blas_root; // root BvhNode index (TRIMESH) — or a leaf ref (< 0) for a 1-leaf BLAS
};
static_assert(sizeof(Instance) == 136, "Instance layout");

// Shading record of a SceneNode (src/scene_node.rs:8-19).
struct NodeRec {
    float refl_mix, refl_atenuation, alpha;
    uint32_t material_id;
    double refr_coeff;
    uint32_t pad[2];
};
static_assert(sizeof(NodeRec) == 32, "NodeRec layout");

struct MaterialRec { // src/phong_material.rs:9-16
    uint32_t kind;
    float ka[3], kd[3], ks[3];
    float shininess;
    int32_t tex, alpha_tex;
    uint32_t pad[3];
};
static_assert(sizeof(MaterialRec) == 64, "MaterialRec layout");

struct TextureRec { // src/texture2d.rs:62-66
    uint32_t width, height, format, interp, overflow, pad;
    const void* texels;
};

// Everything shading needs about a SceneNode in ONE record (scene_node.rs:8-19 + its PhongMaterial,
// phong_material.rs:9-16, + the two texture descriptors, texture2d.rs:62-66): after a hit the chain of
// dependent fetches is triangle -> ShadeRec -> texels instead of node -> material -> texture -> texels.
struct ShadeTex {
    const void* texels; // null = no texture
    uint32_t width, height;
    uint32_t mode;      // format | interp << 8 | overflow << 16
    uint32_t pad;
};
struct ShadeRec {
    float refl_mix, refl_atenuation, alpha;
    uint32_t flags;     // bit 0: the shape / mesh carries uvs; bits 8..15: NraysMaterialKind
    double refr_coeff;
    float ka[3], kd[3], ks[3], shininess;
    ShadeTex tex, alpha_tex;
    uint32_t pad[2];
};
static_assert(sizeof(ShadeRec) == 120 || sizeof(ShadeRec) == 128, "ShadeRec layout");

struct LightRec { // src/light.rs:8-13
    double pos[3];
    double radius;
    float color[3];
    uint32_t racsample;
};

// Counters kept in HBM (one 64-bit atomic per wave per kernel for the ray classes; the traversal
// counters are only touched by the instrumented kernel variants).
struct DeviceCounters {
    unsigned long long rays_reflection, rays_refraction, rays_shadow;
    unsigned long long rays_primary_traced; // primary rays whose wave tile entered the trace loop (not decided by the screen bounds / the root test)
    unsigned long long node_tests, tri_tests, prim_tests, hit_records, tex_samples;
    unsigned int overflow;  // set when a continuation queue ran out of capacity
    unsigned int max_depth; // deepest trace depth reached (= number of continuation generations)
    unsigned int max_chain_nodes; // instrumented: most AABB tests spent on one pixel's chain
    unsigned int pad0;
    unsigned long long shadow_elided; // shadow rays counted in rays_shadow but not traced (64 bits like rays_shadow, which contains them: a 4K frame with 256 samples and 8 lights passes 2^32): light samples
                                      // behind the surface, hits on fully transparent / perfectly mirroring points (trace_device.h: light_is_dark, shade_hit)
    unsigned long long node_fetches; // instrumented: 128-byte node records fetched, one per wave for a wave-uniform visit, one per lane otherwise
    unsigned long long dbg2[8];   // tuning builds (NR_PHASE_TIMING): wave cycles outside the queries — dequeue wait, raygen + root test, hit reconstruction + gates, shadow-ray set-up, material, weights + continuation, pixel write
    unsigned long long dbg[8];    // tuning builds (NR_PHASE_TIMING): wave / lane iteration counts of the node loops and triangle leaves, cycles per query class, wave-uniform node iterations
};

constexpr int kMaxGenerations = 64; // hard cap on trace depth (reference recursion is unbounded, scene.rs:246)

// Continuation-ray queue (SoA in HBM).  A hit that spawns both a reflection and a refraction keeps
// the reflection in registers and appends the refraction here through wave ballot + prefix-sum
// compaction; round r of k_bounce reads queue[r & 1] with count[r] entries and appends to
// queue[(r + 1) & 1] / count[r + 1].
struct RayQueue {
    double* o[3];
    double* d[3];
    double* refr;            // RayWithEnergy::refr
    float* energy;           // RayWithEnergy::energy
    float* weight;           // product of blend factors from the primary ray down to this ray
    uint32_t* pixel;         // index into the (tile-compact) output buffer
    uint32_t* depth;         // trace depth of the ray (for max_depth and the hard cap)
    unsigned long long* key; // RNG path key
};

struct DScene {
    const BvhNode* nodes;     // all BVH nodes (both TLASes and every BLAS)
    const TriRec* tris;       // leaf-ordered triangles of every BLAS
    const TriUv* triuvs;      // parallel to tris (may be null if no mesh has uvs)
    const Instance* instances;        // closest-hit TLAS leaves (+ planes at the end)
    const Instance* shadow_instances; // shadow TLAS leaves (+ planes at the end)
    const InstLink* links;            // parallel to instances
    const InstLink* shadow_links;     // parallel to shadow_instances
    const ShadeRec* shade;            // one per scene node
    const double* node_aabbs;         // 6 f64 per scene node: world AABB exactly as the reference computes
                                      // geometry.bounding_volume(&transform) (scene_node.rs:41); gates accepted hits
    const LightRec* lights;
    const int32_t* planes;        // indices into `instances` of planes (infinite AABB: tested linearly)
    const int32_t* shadow_planes; // indices into `shadow_instances` of the same planes
    int32_t closest_root;         // TLAS for Scene::trace (>= 0 node, < 0 leaf ref, kEmptyChild = empty)
    int32_t shadow_root;          // TLAS for Scene::intersects_ray
    uint32_t num_planes;
    uint32_t num_lights;
    float background[3];
    uint32_t incoherent;          // some mesh is hair-like (kInstIncoherent): traverse() ends node phases by quorum
    uint32_t stats_elide;         // instrumented renders (set per launch): bit 0 = skip dark light samples, bit 1 = skip the shading of hits without a term of their own, as the
                                  // scene's PLAIN kernel does (NRAYS_COUNT_AS_TIMED); 0 = trace everything the reference traces
    uint32_t no_elide;            // some light / material / RGBA32F texel of the scene is not finite (or a shininess is negative): x * 0 is then not 0 for every x the
                                  // reference multiplies, so NO shadow ray or Phong evaluation may be skipped: nrays_scene_create decides, and the host renders such a
                                  // scene's plain frames with the instrumented kernel (stats_elide = 0), which skips nothing
    // Small analytic scenes (no meshes; all records below within kLdsSceneBytes): one packed copy of nodes, instances,
    // links, shading records, node AABBs, lights and plane lists, which the kFeatLdsScene kernels stage into LDS once per
    // workgroup — a dependent record fetch then costs an LDS access instead of a trip through the vector memory path.
    // lds_off[k] = byte offset of section k (LdsSection) inside the blob; null when the scene does not qualify.
    const uint32_t* lds_blob;
    uint32_t lds_bytes; // multiple of 16
    uint32_t lds_off[10];
};
enum LdsSection { kLdsNodes = 0, kLdsInstances, kLdsShadowInstances, kLdsLinks, kLdsShadowLinks, kLdsShade, kLdsNodeAabbs, kLdsLights, kLdsPlanes, kLdsShadowPlanes };
constexpr uint32_t kLdsSceneBytes = 8192;

struct DRender {
    uint32_t width, height;      // full frame
    uint32_t rows_local;         // rows in the compact (tile) buffer
    uint32_t spp;                // ray_per_pixel
    uint32_t sample_begin, sample_end; // samples handled by this launch
    uint32_t max_depth;
    uint32_t band_rows, band_owner, band_owners;
    uint32_t first_batch;        // 1: the frame starts here, 0: a later sample batch continues the running sums in `out`
    uint32_t use_rng;            // 0 when no random number can be consumed (window == 0, no area light)
    uint32_t lane_log2;          // log2 of the lanes that share one pixel (sample-major mapping of AA frames; 0 = one lane per pixel)
    // Pixels outside [cull_i0, cull_i1] x [cull_j0, cull_j1] (global column / row) cannot reach the scene's bounding box
    // with any of their primary rays: a wave tile without a pixel inside writes the background without generating a ray
    // (nrays_hip.hip: screen_bounds()).  INT_MIN / INT_MAX = every pixel may hit (camera inside the box, planes, ...).
    int32_t cull_i0, cull_i1, cull_j0, cull_j1;
    // The same bounds in units of scheduling blocks (16 x 16 pixels at one lane per pixel, the wave's pixel block of an
    // anti-aliased frame), local rows: only the blocks [win_x0, win_x0 + win_nx) x [win_y0, win_y0 + win_ny) enter the
    // work lists; the pixels of all other blocks are filled with the background by the kernel's prologue.  Equal to the
    // whole frame when nothing can be decided (or when the frame is rendered in several sample batches).
    uint32_t win_x0, win_nx, win_y0, win_ny;
    // Cost-ordered workgroup lists (tile_order != null, analytic scenes): the first lead_wgs workgroups — one per CU —
    // own the 4 * lead_wgs most expensive entries, one per wave, so every long tile starts at once on a SIMD of its own;
    // the remaining entries (and the rows outside the window) are dealt to ALL workgroups.  0 = one uniform list.
    uint32_t lead_wgs;
    uint32_t lead_entries; // how many entries the lead workgroups own (lead_wgs * 1..4)
    double window_width;
    double inv_width, inv_height; // RN(1 / width), RN(1 / height): the PLAIN kernels' raygen divides by them exactly (trace_device.h: generate_primary)
    double eye[3];
    double m[16];                // (P V)^-1 column-major
    unsigned long long seed;
    // Mesh scenes (dynamic dequeue): per-wave-tile cost of this frame (written) and the wave tiles in
    // descending order of the previous frame's cost (read; null = image order).  Scheduling only.
    uint32_t* tile_cost;
    // The shader clock under this scene's load, measured by the handle's INSTRUMENTED launches: [2] shader cycles (s_memtime) and [3] 100 MHz ticks (s_memrealtime) over the
    // lifetimes of a sample of their waves, summed (never cleared: the ratio is the clock, weighted by wave time); [0], [1] unused
    unsigned long long* cost_meta;
#ifdef NR_DEBUG_TILE_COSTS
    uint32_t dbg_mode;    // tuning builds: 1 = wave_times[1] holds the wave's work-tile cycles / 16 (26 bits) and its number of work tiles (6 bits)
    uint32_t* wave_times; // tuning builds: per wave {kernel entry, first tile, exit} in 10 ns ticks (s_memrealtime) and its tile count
#endif
    const uint32_t* tile_order;
    // Light-parallel wave tiles (multi-light mesh frames): an entry of tile_order with bit 31 set stands for ONE of the 2^light_lsl
    // parts of a wave tile (bits 28..30: which) — 64 >> light_lsl pixels, 2^light_lsl lanes per pixel, one light each in the
    // shadow phase (material_compute) — so that the frame's few longest tiles (a closest-hit traversal + one shadow traversal PER
    // LIGHT per layer, in sequence) are spread over 2^light_lsl waves and their lights run side by side.  order_len[x] = entries of
    // XCD x's list (k_tile_order expands the tiles above its cost threshold).  Scheduling only: pixels do not depend on it.
    const uint32_t* order_len;
    uint32_t light_lsl;
};
constexpr uint32_t kEntrySplit = 0x80000000u, kEntryTileMask = 0x0fffffffu;
// A recorded tile cost (DRender::tile_cost, 16-cycle units): bit 31 = the tile ran as light-parallel / pixel-split parts and the value is its most expensive part's, scaled to the tile.
constexpr uint32_t kCostSplit = 0x80000000u, kCostMask = 0x7fffffffu;

// Kernel permutations by scene content (decided once per scene on the host): a scene only pays, in
// registers and instructions, for the code paths it can reach.
enum Features : int {
    kFeatAnalytic = 1,     // balls / cuboids / cylinders / capsules / cones / planes exist
    kFeatMesh = 2,         // TriMesh nodes exist (BLAS traversal, ray/triangle)
    kFeatAlphaShadow = 4,  // some node may be non-opaque to shadow rays (per-node closest hit + colour filter)
    kFeatDouble = 8,       // some node can spawn a reflection AND a refraction at one hit (second child -> HBM queue)
    kFeatMultiSample = 16, // more than one light sample per hit: shadow rays are traced inside the light loop;
                           // otherwise the single shadow ray is traced before the shading state exists
    kFeatAll = 31,
    kFeatLdsScene = 32,    // analytic-only scenes whose records fit DScene::lds_blob: the kernel reads them from LDS
    kFeatNoXform = 64,     // every TLAS leaf is an untransformed BLAS (identity rotation, zero translation: local space == world space):
                           // the traversal keeps ONE ray instead of a world and a local one — 12 VGPRs and the two ray set-ups per BLAS
                           // visit (the three-wave multi-light kernel: 170 -> 123 spilled dwords, config 4 12.5 -> 11.3 ms)
    kFeatPark = 128        // the three-wave multi-light permutations park a hit's shading state (normal, point, ray direction: 18 dwords
                           // per lane) in LDS across each of its shadow traversals, so that it does not sit in — or get spilled around —
                           // the traversal's inner loops (Stack::park; trace_device.h: material_compute)
};

} // namespace nrays
