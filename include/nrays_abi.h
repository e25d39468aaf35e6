/*
 * nrays_abi.h — C ABI of the MI355X-native replacement for nrays' per-pixel trace loop.
 *
 * This is the drop-in boundary for `scene::render` (reference src/scene.rs:29-36) and for the
 * construction surface that feeds it (`Scene::new` src/scene.rs:119, `SceneNode::new`
 * src/scene_node.rs:22-47, `Light::new` src/light.rs:16, `PhongMaterial::new`
 * src/phong_material.rs:19-26).  A host written in any language (the reference is Rust) fills
 * the POD descriptors below once per scene and calls `nrays_scene_create`; every later
 * `scene::render` becomes one `nrays_render` call.  INTEGRATION.md shows the Rust `extern "C"`
 * binding a maintainer would add.
 *
 * Rules of the boundary
 *   - plain C, `extern "C"`, POD structs only (`#[repr(C)]` on the Rust side), no exceptions;
 *   - the library copies everything it needs inside `nrays_scene_create`; the caller keeps
 *     ownership of every pointer it passes;
 *   - every entry point returns 0 (NRAYS_OK) or a negative NraysStatus; it never aborts
 *     (the reference panics / silently swallows thread panics, src/scene.rs:111 — not reproduced);
 *   - `nrays_last_error()` returns a thread-local, NUL-terminated description of the last failure.
 *
 * The same descriptors are consumed by the CPU oracle (oracle/nrays_oracle.c), which is test
 * infrastructure and is NOT part of this library.
 */
#ifndef NRAYS_ABI_H
#define NRAYS_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever the exported surface grows or a struct changes: 3 = + nrays_debug_blas_build / NraysBlasDump, nrays_multi_get_timings / NraysMultiTimings (round 4); 4 = NraysStats::rays_shadow_elided (round 5); 5 = NraysStats::node_fetches, nrays_render_device_counted, NraysTileCosts::shader_clock_hz / kernel_ms (round 6). */
#define NRAYS_ABI_VERSION 5

typedef enum NraysStatus {
    NRAYS_OK = 0,
    NRAYS_ERR_BAD_ARG = -1,        /* NULL pointer, ray_per_pixel == 0 (src/scene.rs:37), bad index */
    NRAYS_ERR_HIP = -2,            /* a HIP runtime call failed */
    NRAYS_ERR_OOM = -3,            /* host or device allocation failed */
    NRAYS_ERR_UNSUPPORTED = -4,    /* e.g. mesh vertices that are not f32-exact */
    NRAYS_ERR_NO_DEVICE = -5,      /* no gfx950 device visible to the process */
    NRAYS_ERR_QUEUE_OVERFLOW = -6, /* continuation-ray queue capacity exceeded */
    NRAYS_ERR_RCCL = -7            /* an RCCL call failed (multi-GPU entry points) */
} NraysStatus;

/* Shapes the loader can construct (examples/loader3d.rs:593-695). */
typedef enum NraysShapeKind {
    NRAYS_SHAPE_BALL = 0,     /* params[0] = radius                      (loader3d.rs:601) */
    NRAYS_SHAPE_CUBOID = 1,   /* params[0..2] = half extents             (loader3d.rs:612) */
    NRAYS_SHAPE_CYLINDER = 2, /* params[0] = half height, params[1] = radius, axis = local Y (:623) */
    NRAYS_SHAPE_CAPSULE = 3,  /* params[0] = half height, params[1] = radius               (:634) */
    NRAYS_SHAPE_CONE = 4,     /* params[0] = half height, params[1] = radius, apex at +Y   (:645) */
    NRAYS_SHAPE_PLANE = 5,    /* params[0..2] = unit normal, through the local origin      (:656) */
    NRAYS_SHAPE_TRIMESH = 6   /* mesh_id selects an NraysMesh                              (:695) */
} NraysShapeKind;

/* Material implementations that can cross the boundary (src/material.rs:6-17). */
typedef enum NraysMaterialKind {
    NRAYS_MAT_PHONG = 0,  /* src/phong_material.rs:9-152 */
    NRAYS_MAT_NORMAL = 1, /* src/normal_material.rs:5-22 */
    NRAYS_MAT_UV = 2      /* src/uv_material.rs:6-28 */
} NraysMaterialKind;

typedef enum NraysTexelFormat {
    NRAYS_TEXEL_RGBA8 = 0,  /* 4 x u8; sampled as `u8 as f32 / 255.0` (src/texture2d.rs:111-162) */
    NRAYS_TEXEL_RGBA32F = 1 /* 4 x f32, the reference's own in-memory form (Point4<f32>) */
} NraysTexelFormat;

typedef enum NraysInterpolation { NRAYS_INTERP_BILINEAR = 0, NRAYS_INTERP_NEAREST = 1 } NraysInterpolation;
typedef enum NraysOverflow { NRAYS_OVERFLOW_WRAP = 0, NRAYS_OVERFLOW_CLAMP = 1 } NraysOverflow;

/* src/light.rs:8-23.  `racsample` is already floor(sqrt(nsample)) (light.rs:20). */
typedef struct NraysLight {
    double pos[3];
    double radius;
    uint32_t racsample;
    float color[3];
} NraysLight;

/* src/texture2d.rs:10-76.  Row 0 is the BOTTOM row of the image: the Y flip of
 * texture2d.rs:99-107 has already been applied by whoever decoded the file, and so has the
 * depth/opacity decode of :109-177 (opaque -> (r,g,b,1); opacity map -> (1,1,1,a)). */
typedef struct NraysTexture {
    uint32_t width;
    uint32_t height;
    uint32_t format;   /* NraysTexelFormat */
    uint32_t interp;   /* NraysInterpolation */
    uint32_t overflow; /* NraysOverflow */
    uint32_t reserved;
    const void* texels; /* width*height texels, row-major, index y*width + x (texture2d.rs:204) */
} NraysTexture;

/* src/phong_material.rs:9-26 for PHONG; the colour fields are ignored for NORMAL / UV. */
typedef struct NraysMaterial {
    uint32_t kind; /* NraysMaterialKind */
    float ambiant[3];
    float diffuse[3];
    float specular[3];
    float shininess;
    int32_t texture_id;       /* index into textures, or -1 */
    int32_t alpha_texture_id; /* index into textures, or -1 */
} NraysMaterial;

/* ncollide3d TriMesh::new(points, indices, uvs) as called at examples/loader3d.rs:695.
 * Several meshes may alias the same `vertices` / `uvs` arrays (one SceneNode per OBJ group,
 * each holding the whole vertex array and only its faces, loader3d.rs:690-695). */
typedef struct NraysMesh {
    uint32_t num_vertices;
    uint32_t num_triangles;
    const double* vertices;  /* 3*num_vertices, local space; must be f32-exact (obj.rs:197-205 parses f32) */
    const double* uvs;       /* 2*num_vertices or NULL */
    const uint32_t* indices; /* 3*num_triangles */
} NraysMesh;

/* One SceneNode (src/scene_node.rs:8-47).  The transform is given as the loader builds it:
 * Isometry3::new(translation, axis_angle) (examples/loader3d.rs:546-552), i.e. `axis_angle` is a
 * scaled-axis rotation in RADIANS (|axis_angle| = angle), not Euler angles. */
typedef struct NraysNode {
    uint32_t shape_kind; /* NraysShapeKind */
    uint32_t solid;      /* 0/1, scene_node.rs:13 */
    double params[3];
    double translation[3];
    double axis_angle[3];
    float refl_mix;
    float refl_atenuation;
    float alpha;
    float reserved0;
    double refr_coeff;
    uint32_t material_id;
    int32_t mesh_id; /* for NRAYS_SHAPE_TRIMESH, else -1 */
} NraysNode;

/* Everything `Scene::new(nodes, lights, background)` receives (src/scene.rs:119-133). */
typedef struct NraysSceneDesc {
    float background[3];
    uint32_t num_lights;
    const NraysLight* lights;
    uint32_t num_materials;
    const NraysMaterial* materials;
    uint32_t num_textures;
    const NraysTexture* textures;
    uint32_t num_meshes;
    const NraysMesh* meshes;
    uint32_t num_nodes;
    const NraysNode* nodes;
} NraysSceneDesc;

/* Arguments of scene::render (src/scene.rs:29-36) plus extensions whose zero value preserves the
 * reference behaviour. */
typedef struct NraysRenderParams {
    uint32_t width;           /* resolution.x */
    uint32_t height;          /* resolution.y */
    uint32_t ray_per_pixel;   /* must be > 0 (scene.rs:37) */
    uint32_t max_depth;       /* 0 = energy rule only (scene.rs:204); else extra cap on trace depth.
                                 A hard safety cap of 64 generations always applies (unbounded
                                 refraction recursion in the reference, scene.rs:246). */
    double window_width;      /* AA jitter window in pixels (scene.rs:75) */
    double camera_eye[3];
    double inv_proj_view[16]; /* (P*V)^-1, COLUMN-major as nalgebra stores Matrix4 (loader3d.rs:77-79) */
    uint64_t seed;            /* counter-based RNG seed (reference RNG is OS-seeded, scene.rs:75) */
    /* Framebuffer tiling (multi-GPU): rows are grouped in bands of `band_rows`; band b is rendered
     * iff b % band_owners == band_owner.  band_rows == 0 renders the whole frame.  The output of a
     * tiled render is the compact buffer of the owner's bands in increasing order. */
    uint32_t band_rows;
    uint32_t band_owner;
    uint32_t band_owners;
    uint32_t reserved;
} NraysRenderParams;

/* Counters of the last render of a scene (or of an oracle render).  A "ray" is one BVT query
 * (`world.best_first_search`, src/scene.rs:153,166). */
typedef struct NraysStats {
    uint64_t rays_primary;    /* scene.rs:89 */
    uint64_t rays_reflection; /* scene.rs:209 */
    uint64_t rays_refraction; /* scene.rs:246 */
    uint64_t rays_shadow;     /* scene.rs:153 */
    uint64_t node_tests;      /* AABB tests (TLAS + BLAS); filled by instrumented renders only */
    uint64_t tri_tests;       /* ray/triangle tests */
    uint64_t prim_tests;      /* analytic primitive / instance tests */
    uint64_t hit_records;     /* node + material records fetched at accepted hits */
    uint64_t tex_samples;     /* texture samples (4 taps each when bilinear) */
    uint32_t generations;     /* continuation generations executed */
    uint32_t instrumented;    /* 1 if the traversal counters above are valid */
    double kernel_ms_primary; /* mean GPU time of the primary kernel launch (HIP events on the render stream) */
    double kernel_ms_total;   /* mean GPU time of a whole render, first launch to last */
    uint32_t frames_timed;    /* renders averaged in the two figures above (since the previous get_stats): the events
                                 are recorded on every 4th render of a handle and on every instrumented one */
    uint32_t reserved;
    uint64_t rays_primary_traced; /* instrumented renders only: primary rays that went through a BVT query: rays_primary minus the ones whose wave tile was
                                     decided without one (outside the scene's screen bounds, or no ray of the tile passes the
                                     root of the BVT) — the pixels are the same, the reference would have queried for them */
    uint64_t rays_shadow_elided;  /* part of rays_shadow: shadow rays the reference traces although their result is multiplied by exactly 0 — light samples
                                     behind the surface (diffuse and specular coefficients both 0: phong_material.rs:109-141) and the samples of hits that
                                     contribute nothing of their own to the pixel (a fully transparent point: opacity-map texel 0 or node alpha 0; a perfect
                                     mirror: scene.rs:179-190).  Counted, so that rays_shadow stays the reference's number, but not traced by plain
                                     renders; instrumented renders trace them: 0 (NRAYS_COUNT_AS_TIMED: they skip them like a plain render and report them here) */
    uint64_t node_fetches;        /* instrumented renders only: 128-byte BVH node records fetched — ONE per wave for a visit in which every active lane sits on the same
                                     node with the same direction signs (the kernels read it through the scalar unit and broadcast), one per lane otherwise.
                                     node_fetches * 128 is the traversal's unique node traffic; node_tests * 32 counts a box per lane whoever fetched it */
} NraysStats;

/* Threading contract of a scene handle: the library is re-entrant on DISTINCT handles (any threads, any streams).
 * ONE handle must not be used by two threads at the same time (its calls must be serialised by the caller); its
 * renders execute in call order — a render enqueued on another stream than its predecessor is ordered behind it by
 * the library — because the handle owns per-frame device state (counters, continuation queues, per-camera scheduling state, tile
 * costs).  The environment switches NRAYS_MAX_PRIMARY / NRAYS_GRAB / NRAYS_LPT (tests and A/B runs) are read once,
 * by nrays_scene_create. */
typedef struct NraysScene NraysScene; /* opaque */

/* Builds the device-resident scene on the CURRENT HIP device of the calling thread: flattens the
 * nodes, builds the BVHs, uploads.  Replaces Scene::new + BVT::new_balanced (src/scene.rs:119-133). */
int nrays_scene_create(const NraysSceneDesc* desc, NraysScene** out_scene);

/* Replaces scene::render (src/scene.rs:29-116).  `out_rgb` is HOST memory, caller-allocated,
 * rows*width*3 floats, row-major, index (i + j*width)*3 (scene.rs:104), where rows = height for an
 * untiled render and nrays_tile_rows(params) for a tiled one.  Blocking. */
int nrays_render(NraysScene* scene, const NraysRenderParams* params, float* out_rgb);

/* The same frame as 8-bit RGB, quantised on the device exactly as Image::to_png does on the host (src/image.rs:66-76:
 * c * 255, clamped to [0, 255], truncated; NaN -> 0): what the loader3d front-end writes into its PNG, at a quarter of
 * the device-to-host bytes of nrays_render.  `out_rgb8` is HOST memory, rows*width*3 bytes, same indexing.  Blocking. */
int nrays_render_rgb8(NraysScene* scene, const NraysRenderParams* params, uint8_t* out_rgb8);

/* Same, but `out_rgb_device` is DEVICE memory on the scene's device and the work is enqueued on
 * `hip_stream` (a hipStream_t, NULL = default stream) without a final synchronisation unless the
 * scene needs host-side generation control (transparent scenes). */
int nrays_render_device(NraysScene* scene, const NraysRenderParams* params, float* out_rgb_device,
                        void* hip_stream);

/* As nrays_render_device, with the traversal counters of NraysStats collected (slower). */
int nrays_render_device_instrumented(NraysScene* scene, const NraysRenderParams* params,
                                     float* out_rgb_device, void* hip_stream);

/* As nrays_render_device_instrumented, with a choice of WHAT is counted.  flags = 0: the reference algorithm — every ray scene.rs / phong_material.rs trace is traced
 * and counted.  NRAYS_COUNT_AS_TIMED: the work of the PLAIN (timed) render of this scene — the shadow rays whose result is multiplied by exactly 0 are skipped as the
 * plain kernels skip them (and reported in rays_shadow_elided), so that node / triangle / hit / texture counts are those of the kernel whose time a roofline divides by
 * (bench.py: roofline_block).  Pixels are the same either way. */
#define NRAYS_COUNT_AS_TIMED 1u
int nrays_render_device_counted(NraysScene* scene, const NraysRenderParams* params, float* out_rgb_device, void* hip_stream, uint32_t flags);

/* Number of rows in the compact output buffer of a (possibly tiled) render. */
uint32_t nrays_tile_rows(const NraysRenderParams* params);

/* Un-permutes `band_owners` gathered compact tile buffers (concatenated in owner order, each
 * nrays_tile_rows*width*3 floats) into one row-major frame.  Device pointers. */
int nrays_untile_device(const float* gathered, float* out_rgb_device, uint32_t width, uint32_t height,
                        uint32_t band_rows, uint32_t band_owners, void* hip_stream);

/* Synchronises with the last render of `scene` and returns its counters; the kernel timings are
 * averaged over the renders issued since the previous call (at most 256). */
int nrays_get_stats(NraysScene* scene, NraysStats* out_stats);

/* Counters of the PRIMARY kernel alone (first sample batch) of the last instrumented render: its
 * primary rays, the shadow rays they spawned and the traversal work of both — the per-launch units
 * behind bench.py's roofline figure. */
int nrays_get_primary_kernel_stats(NraysScene* scene, NraysStats* out_stats);

/* Per-wave-tile cost of the last frame that recorded it (the first frames of a camera record the shader cycles every 8x8-pixel
 * wave tile took, for the cost-ordered work lists): the two numbers that bound a frame of the persistent kernel — it cannot end
 * before its LONGEST tile does (a pixel's chain of dependent traversals), nor before sum / resident_waves cycles have passed. */
typedef struct NraysTileCosts {
    uint64_t tiles;          /* wave tiles recorded */
    uint64_t sum_cycles;     /* shader cycles (s_memtime) the waves spend on them (every part of a tile the cost-ordered lists split counted) */
    uint64_t max_cycles;     /* the longest unit the schedule deals: a tile, or ONE PART of a split tile (light-parallel / pixel-split parts) */
    uint64_t resident_waves; /* waves of the persistent grid that rendered the frame */
    double shader_clock_hz;  /* the shader clock under this scene's load, MEASURED by the handle's instrumented launches (nrays_render_device_instrumented / _counted):
                                s_memtime over s_memrealtime (100 MHz), summed over the lifetimes of a sample of their waves; 0 before the first such launch */
    double kernel_ms;        /* duration of that launch (HIP events of its own around it).  max_cycles / shader_clock_hz and sum_cycles / resident_waves /
                                shader_clock_hz are fractions of kernel_ms: units and time come from the same launch */
} NraysTileCosts;
int nrays_get_tile_costs(NraysScene* scene, NraysTileCosts* out);

/* Probe of the two BVT queries of the path on caller-supplied rays — the device intersectors and traversals WITHOUT raygen and
 * shading around them, so that fixtures derived independently of this code base (tests/golden/kat_independent.npz) can be
 * checked against the HIP path directly.
 *   mode 0  Scene::trace's closest-hit query (src/scene.rs:164-166) + SceneNode::cast's record (src/scene_node.rs:51-54):
 *           flags bit 0 = hit, bit 1 = the record carries uvs; toi, world normal, uv, scene-node index.
 *   mode 1  Scene::intersects_ray (src/scene.rs:147-161) with `max_toi[i]`: flags bit 0 = blocked by an opaque node;
 *           normal[0..2] = the colour filter of the transparent nodes crossed (1, 1, 1 if none).
 * `origins` / `dirs`: n x 3 doubles, `max_toi`: n doubles (mode 1 only), `out`: n records; all HOST memory.  Blocking. */
typedef struct NraysCastResult {
    double toi;
    double normal[3];
    double uv[2];
    int32_t node_id;
    uint32_t flags;
} NraysCastResult;
int nrays_debug_cast_batch(NraysScene* scene, uint32_t mode, uint32_t n, const double* origins, const double* dirs,
                           const double* max_toi, NraysCastResult* out);

/* World AABB of scene node `node` as the device holds it: geometry.bounding_volume(&transform) of src/scene_node.rs:41 in the
 * reference's arithmetic; the kernels apply ncollide's exact ray / AABB test to it for every accepted hit (the reference only casts
 * a node whose AABB the ray passes, src/scene.rs:276).  out = {min x, y, z, max x, y, z}. */
int nrays_debug_node_aabb(NraysScene* scene, uint32_t node, double out[6]);

/* How the library classified the scene (test probe): out[0] = kernel permutation it renders with (1 analytic shapes, 2 meshes,
 * 4 some node may be non-opaque to shadow rays, 16 more than one light sample per hit), out[1] = 1 when a hair-like mesh makes the
 * BVT queries end their node phases by quorum (NRAYS_NODE_QUORUM=0 in the environment of nrays_scene_create turns that off). */
int nrays_debug_scene_flags(const NraysScene* scene, uint32_t out[2]);

/* Test probe of `Scene::new`'s BVT construction for one TriMesh (src/scene.rs:119-133; ncollide's BVT::new_balanced inside TriMesh::new,
 * examples/loader3d.rs:695): builds the BLAS of `mesh` with the host builder (flags bit 0 clear) or the device builder (bit 0 set;
 * nrays_scene_create picks it for meshes from NRAYS_GPU_BUILD_MIN triangles, default 2 000), bit 1 = without triangle pre-splitting,
 * and copies it out: `nodes` = num_nodes x 32 floats (the 128-byte 4-wide node: planes by slot, child refs in slot 2, local
 * indices, depth-first order), `tri_ids` = the triangle index behind each of the num_refs leaf slots.  Both builders apply the same
 * split rule with the same arithmetic, so from the same references they return the same nodes.  The caller provides the buffers
 * (node_capacity / ref_capacity entries); all HOST memory.  Blocking. */
typedef struct NraysBlasDump {
    uint32_t num_nodes;
    uint32_t num_refs;
    int32_t root;      /* >= 0: node index, < 0: a single leaf */
    int32_t max_depth;
    uint32_t hairy;    /* the mesh was classified hair-like (aggressive pre-splitting, quorum-ended node phases) */
    uint32_t node_capacity;
    uint32_t ref_capacity;
    uint32_t pad;
    float* nodes;
    uint32_t* tri_ids;
} NraysBlasDump;
int nrays_debug_blas_build(const NraysMesh* mesh, uint32_t flags, NraysBlasDump* out);

/* Device bytes of the flattened scene (BVH nodes, triangle records, instance / shading records, textures): what
 * a frame must read at least once — the compulsory part of bench.py's roofline block. */
uint64_t nrays_scene_device_bytes(const NraysScene* scene);

void nrays_scene_destroy(NraysScene* scene);

/* ---------------------------------------------------------------------------------------------------------------
 * Multi-GPU: the framebuffer tiled over the GPUs of one node (replaces the thread partition of src/scene.rs:49-66).
 * The scene is replicated on every GPU, bands of 16 rows are dealt round-robin to the owners, every owner renders its
 * compact tile, ONE exchange (grouped RCCL send / receive over xGMI, every peer straight to owner 0) brings the tiles
 * to owner 0, which un-permutes them.  The frame is bit-identical for any number of owners.
 *
 *   one process, all GPUs (what a Rust caller of scene::render uses):
 *       nrays_comm_create_local(n, NULL, &comm); nrays_scene_set_create(&desc, comm, &set);
 *       nrays_render_multi(set, &params, frame);            // = scene::render on n GPUs
 *   one process per GPU (torch.distributed.run, MPI, ...): rank 0 calls nrays_comm_unique_id, ships the 128 bytes to
 *       the other ranks by its own means, every rank calls nrays_comm_create(id, n, rank, &comm) on ITS device, then
 *       nrays_scene_set_create / nrays_render_multi[_device] collectively; rank 0 receives the frame.
 * The band fields of NraysRenderParams are ignored (the set owns the partition). */
#define NRAYS_UNIQUE_ID_BYTES 128
typedef struct NraysComm NraysComm;         /* opaque */
typedef struct NraysSceneSet NraysSceneSet; /* opaque */

int nrays_comm_unique_id(uint8_t out_id[NRAYS_UNIQUE_ID_BYTES]);
int nrays_comm_create(const uint8_t id[NRAYS_UNIQUE_ID_BYTES], uint32_t num_ranks, uint32_t rank, NraysComm** out_comm);
/* `devices`: HIP device index of every owner, or NULL for owner o on device o % device_count.  Owners may share a
 * device (their tiles then move by device-to-device copies): a 1-GPU box can run the N-owner path. */
int nrays_comm_create_local(uint32_t num_owners, const int32_t* devices, NraysComm** out_comm);
uint32_t nrays_comm_owners(const NraysComm* comm);
void nrays_comm_destroy(NraysComm* comm); /* after every scene set that uses it */

int nrays_scene_set_create(const NraysSceneDesc* desc, NraysComm* comm, NraysSceneSet** out_set);
void nrays_scene_set_destroy(NraysSceneSet* set);
/* The per-GPU scene handles behind a set (owned by the set): for instrumented renders and per-GPU statistics of one
 * owner's tile (nrays_render_device_instrumented / nrays_get_stats with that owner's band parameters). */
uint32_t nrays_scene_set_num_local(const NraysSceneSet* set);
NraysScene* nrays_scene_set_local_scene(NraysSceneSet* set, uint32_t k, uint32_t* out_owner);

/* scene::render on the group; `out_rgb` is HOST memory (height*width*3 floats), filled on the process that drives
 * owner 0 (NULL elsewhere).  Blocking. */
int nrays_render_multi(NraysSceneSet* set, const NraysRenderParams* params, float* out_rgb);
/* Same with DEVICE memory on owner 0's GPU and no final synchronisation: consecutive calls form a depth-1 pipeline
 * (the tile render of frame k + 1 overlaps the exchange of frame k).  nrays_multi_sync waits for everything enqueued. */
int nrays_render_multi_device(NraysSceneSet* set, const NraysRenderParams* params, float* out_rgb_device);
int nrays_multi_sync(NraysSceneSet* set);
/* Counters of the last frame summed over the owners this process drives. */
int nrays_multi_get_stats(NraysSceneSet* set, NraysStats* out_stats);
/* Where a frame of the group spends its time on THIS process's first owner, averaged over the frames since the previous call
 * (HIP events on the owner's render / communication streams): the tile render (nrays_render_device), the exchange (owner 0: its own
 * tile copy + every receive; other owners: their send) and, on owner 0, the un-permute (k_untile).  The reference's analogue is the
 * join of its render threads, src/scene.rs:97-112. */
typedef struct NraysMultiTimings {
    double render_ms;
    double exchange_ms;
    double untile_ms;
    uint32_t frames;   /* frames averaged */
    uint32_t owner;    /* band owner the figures belong to */
} NraysMultiTimings;
int nrays_multi_get_timings(NraysSceneSet* set, NraysMultiTimings* out_timings);

const char* nrays_last_error(void);

uint32_t nrays_abi_version(void);

#ifdef __cplusplus
}
#endif

#endif /* NRAYS_ABI_H */
