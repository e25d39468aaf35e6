// scene_handle.h — the scene handle behind the C ABI (NraysScene) and what the library's translation units share
// around it.  Host only; nrays_hip.hip owns the megakernel path (k_primary), wavefront.hip the staged path.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/nrays_abi.h"
#include "device_types.h"
#include "scene_build.h"

namespace nrays {

int set_last_error(int status, const std::string& msg); // nrays_hip.hip: sets nrays_last_error(), returns `status`
struct WavefrontState;                                   // wavefront.hip: buffers of the staged path, created on first use

struct QueueMem {
    RayQueue q;
    void* block = nullptr;
};

// A camera as the scheduler sees it (nrays_hip.hip: cam_snapshot): eye, unit rays through the four corners of the frame, the angle of a pixel,
// the distance to the scene's bounding box.
struct CamSnap { double eye[3] = {0, 0, 0}; double dir[4][3] = {}; double pix_angle = 0.0, depth = 1.0; bool valid = false; };

constexpr int kNumCounts = kMaxGenerations + 2 + 8; // queue round counters + 8 per-XCD work counters
constexpr int kMaxGrid = 2048;     // upper bound of the persistent grid (the launch uses CUs x waves/SIMD workgroups)

} // namespace nrays

using nrays::DeviceCounters; using nrays::DScene; using nrays::HostScene; using nrays::QueueMem;

struct NraysScene {
    int device = 0;
    HostScene host;          // kept for counts only; bulk arrays are released after upload
    DScene d;
    std::vector<void*> allocs;
    uint64_t scene_bytes = 0; // device bytes of the uploaded scene arrays (BVH nodes, triangles, records, textures)
    // per-scene transient state, grown on demand
    QueueMem queue[2];
    uint32_t queue_capacity = 0;
    uint32_t* d_counts_set[2] = {nullptr, nullptr};         // double-buffered, kNumCounts each
    DeviceCounters* d_counters_set[2] = {nullptr, nullptr}; // double-buffered per frame
    uint32_t* d_counts = nullptr;         // set used by the last launch
    DeviceCounters* d_counters = nullptr; // set used by the last frame
    uint64_t launch_index = 0, frame_index = 0;
    uint32_t* d_spill = nullptr;
    long long* d_fixed = nullptr; size_t fixed_slots = 0; // per-pixel fixed-point sums of the queued chains (double-branching scenes)
    bool fixed_dirty = false; // k_bounce rounds were enqueued and their k_fold_fixed was not (an error in between): cleared at the next frame's start
    // previous frame's wave-tile costs (k_primary) and the order derived from them (k_tile_order); valid for one
    // (width, rows, band) geometry at a time
    uint32_t* d_tile_cost = nullptr; uint32_t* d_tile_order = nullptr; uint32_t tile_slots = 0;
    hipEvent_t ev_rec[2] = {nullptr, nullptr}; bool rec_events_valid = false; int rec_slot = -1; // around the last primary launch that recorded tile costs (NraysTileCosts::kernel_ms)
    unsigned long long* d_cost_meta = nullptr; // DRender::cost_meta: start / end ticks and the measured clock of the launch that recorded d_tile_cost
    // light-parallel tiles: log2 of the lanes per pixel (0 = the scene is not eligible), the split threshold in units of the frame's
    // work per resident wave (NRAYS_LIGHT_SPLIT: 0 = never, < 0 = every tile, default 1), the lengths of the eight lists
    uint32_t light_lsl = 0; float light_split_factor = 1.0f; uint32_t* d_order_len = nullptr;
    const float* d_seed_boxes = nullptr; uint32_t seed_boxes = 0; bool seed_enabled = true; uint32_t seed_rays = 1; // k_seed_costs: first guess of a cold camera's tile costs (NRAYS_COST_SEED=0: none)
    uint64_t cost_key = 0; bool cost_valid = false;
    uint32_t cost_tiles = 0, cost_grid = 0, cost_split_lsl = 0; // wave tiles / workgroups / log2 of a split tile's parts of the frame that recorded d_tile_cost last (nrays_get_tile_costs)
    // analytic scenes (workgroup lists): costs are recorded on the first frame of a camera, sorted once on the second, and
    // the order is then reused as long as the camera stays (the scene of a handle never changes)
    uint64_t cost_cam = 0, order_key = 0, order_cam = 0; bool order_valid = false; uint32_t order_age = 0;
    // ... and by cameras NEAR the one whose costs it was sorted from (nrays_hip.hip: cam_shift_px) for up to kMaxOrderAge frames, so that a moving camera does not
    // record and sort on every frame.  order_seeded: the order comes from k_seed_costs' guess, the next frame replaces it.
    nrays::CamSnap cost_snap, order_snap; bool order_seeded = false;
    bool host_times = false;                        // NRAYS_HOST_TIMES: render_impl prints where the host time of a handle's first frames goes
    bool near_reuse = true;                         // NRAYS_NEAR_REUSE=0: only the very same camera reuses an order (A/B)
    double near_pixels = 16.0; uint32_t max_order_age = 8; // NRAYS_NEAR_PIXELS / NRAYS_ORDER_AGE
    float split_hyst = 0.5f;                        // NRAYS_SPLIT_HYST: a tile that ran in parts stays split down to this fraction of the split threshold (k_tile_order)
    bool lone_known = false; uint64_t lone_key = 0, stats_key = 0; // the lead / second decision of the last sort that reported, and the geometry it belongs to
    // ... and the sort also reports the sum and the maximum of the costs: their ratio is the frame's parallelism, which
    // decides between cost-ordered lists with the long tiles on the first workgroup of each CU (few long tiles) and image-order
    // lists (many tiles: throughput)
#ifdef NR_DEBUG_TILE_COSTS
    uint32_t* d_wave_times = nullptr; uint32_t dbg_grid = 0; uint32_t* d_seed_copy = nullptr;
#endif
    unsigned long long* d_cost_stats = nullptr; unsigned long long* h_cost_stats = nullptr; hipEvent_t ev_stats = nullptr;
    bool stats_pending = false, lone_waves = false;
    uint32_t spill_entries = 0; // HBM stack entries per lane beyond the kLdsStack entries kept in LDS (0 = never needed)
    int num_cus = 256;
    int features = nrays::kFeatAll;
    bool park = true;     // kFeatPark permutations for the three-wave multi-light kernels (NRAYS_PARK=0: off)
    bool noxform = false; // every BLAS untransformed: the kFeatNoXform permutations of the mesh kernels render this scene
    float* d_frame = nullptr; size_t frame_floats = 0;
    uint8_t* d_rgb8 = nullptr; size_t rgb8_bytes = 0; // nrays_render_rgb8
    hipStream_t own_stream = nullptr;
    // ring of HIP event triples (frame begin, primary kernel begin/end, frame end) recorded on the render
    // stream; nrays_get_stats averages the frames recorded since its previous call.
    static constexpr int kRing = 256;
    hipEvent_t ev_begin[kRing] = {}, ev_pbegin[kRing] = {}, ev_pend[kRing] = {}, ev_end[kRing] = {};
    bool single_launch[kRing] = {};
    bool has_prepass[kRing] = {}; // the frame started with k_tile_order: ev_begin was recorded before it
    uint64_t frames_recorded = 0, frames_reported = 0;
    DeviceCounters* d_counters_primary = nullptr; // snapshot taken right after the primary kernel
    hipStream_t last_stream = nullptr;
    hipEvent_t last_done = nullptr; // last event recorded by the previous render (one of the ring's events)
    hipEvent_t ev_switch = nullptr; // recorded on the previous render's stream when a render arrives on another one
    bool have_last = false;
    // A/B and test switches, read ONCE when the handle is created (never in the frame path)
    uint64_t max_primary_per_launch = 32ull << 20; // NRAYS_MAX_PRIMARY: sample batching threshold (tests force several launches)
    bool max_primary_forced = false;
    int lane_log2_override = -1;                    // NRAYS_LANE_LOG2: cap of the lanes per pixel of AA frames (A/B)
    // The HIP events behind NraysStats::kernel_ms_* are recorded on every 4th frame of a handle (and on every instrumented
    // one): three event records per frame cost ~6 us of a 85 us frame (balls: 0.0849 -> 0.0789 ms per step); the averages
    // nrays_get_stats reports are over the sampled frames.  NRAYS_EVENT_STRIDE overrides it (1 = every frame).
    uint32_t event_stride = 4;
    uint64_t frames_total = 0;
    bool last_timed = true;
    int grab_override = -1;                         // NRAYS_GRAB
    bool lpt_enabled = true;                        // NRAYS_LPT=0 restores image order
    bool lpt_reuse = true;                          // NRAYS_LPT_REUSE=0: mesh scenes re-sort their tiles every frame even when the camera rests
    bool lpt_analytic = true;                       // NRAYS_LPT_ANALYTIC=0: analytic scenes never switch to cost-ordered lists
    int grid_wg_per_cu = 0;                         // NRAYS_GRID_WG_PER_CU=n caps the persistent grid at n workgroups per CU (tuning)
    double lone_factor = 1.5;                       // NRAYS_LONE_FACTOR: cost-ordered lead / second lists when sum / max of the tile costs < factor * SIMDs
    int lead_per_wg = 4;                            // NRAYS_LEAD_PER_WG=1..4: long entries per lead workgroup
    bool lead_mode = true;                          // NRAYS_LEAD_WGS=0: cost-ordered lists run on one workgroup per CU instead of lead + second workgroups
    int occ_override = -1;                          // NRAYS_OCC=2|3: waves per SIMD of the alpha-shadow mesh kernels (A/B)
    bool cull_enabled = true;                       // NRAYS_SCREEN_CULL=0: no wave tile is decided from the scene's screen bounds
    nrays::WavefrontState* wf = nullptr;            // staged (wavefront) path: queues, chunk tables, sums (wavefront.hip)
    int wavefront_mode = -1;                        // NRAYS_WAVEFRONT: 0 = never, 1 = whenever the scene is eligible, -1 = the library's rule (wavefront.hip)
    NraysStats last;
    uint64_t last_primary = 0, last_primary_first_batch = 0;
    bool last_instrumented = false;
};

