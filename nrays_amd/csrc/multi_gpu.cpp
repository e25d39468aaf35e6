// multi_gpu.cpp — framebuffer tiling over the GPUs of one node, INSIDE the library.
//
// The reference's only parallelism is a contiguous static split of the pixel range over CPU threads
// (src/scene.rs:49-66).  Here the scene is replicated on every GPU, the frame is cut into bands of 16 rows dealt
// round-robin to the owners (a GPU each), every owner renders its compact tile (nrays_render_device with the band
// fields of NraysRenderParams), ONE exchange step brings the tiles to owner 0 over RCCL / xGMI — grouped
// ncclSend / ncclRecv: every peer has its own direct link to GPU 0, so a flat gather, no ring — and k_untile
// un-permutes the bands there.  The RNG is keyed by the global pixel index, so the frame does not depend on the
// number of owners.
//
// Two ways to form the group, one code path:
//   nrays_comm_create_local   ONE process drives all owners (what a Rust caller of scene::render wants): one host
//                             thread, hipSetDevice per owner, ncclCommInitAll over the distinct devices.  Owners that
//                             share owner 0's device exchange by a device-to-device copy (this is how a 1-GPU box
//                             runs the N-owner path in the tests).
//   nrays_comm_create         one process per GPU (torch.distributed.run launches bench.py this way): rank r is owner r,
//                             the ncclUniqueId of rank 0 reaches the others through the caller's own channel.
//
// A step is a depth-1 pipeline: the tile render of frame k + 1 is enqueued on the render stream while the exchange of
// frame k is still in flight on the communication stream (two tile / gather buffers alternate; events order the reuse).
// Everything here goes through the public C ABI of nrays_hip.hip (nrays_scene_create, nrays_render_device,
// nrays_untile_device, nrays_get_stats); no Python, no torch.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nrays_abi.h"

namespace nrays { int set_last_error(int status, const std::string& msg); }

namespace {

constexpr uint32_t kBandRows = 16; // one 16x16 workgroup tile high

#define MG_HIP(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess)                                                                                     \
            return nrays::set_last_error(e_ == hipErrorOutOfMemory ? NRAYS_ERR_OOM : NRAYS_ERR_HIP,               \
                                         std::string(#expr) + ": " + hipGetErrorString(e_));                      \
    } while (0)
#define MG_NCCL(expr)                                                                                             \
    do {                                                                                                          \
        ncclResult_t r_ = (expr);                                                                                 \
        if (r_ != ncclSuccess)                                                                                    \
            return nrays::set_last_error(NRAYS_ERR_RCCL, std::string(#expr) + ": " + ncclGetErrorString(r_));     \
    } while (0)

// Restores the calling thread's current HIP device on every exit path (the entry points switch devices per owner).
struct DeviceGuard {
    int device = 0;
    DeviceGuard() { (void)hipGetDevice(&device); }
    ~DeviceGuard() { (void)hipSetDevice(device); }
};

} // namespace

struct NraysComm {
    bool ranked = false;          // one process per GPU
    uint32_t owners = 1;          // band owners of a frame (= ranks, or local owners)
    uint32_t rank = 0;            // ranked mode: this process's owner index
    std::vector<int> devices;     // local mode: HIP device of every owner; ranked mode: {current device}
    // one communicator per DISTINCT local device (local mode) or the process's single communicator (ranked mode)
    std::vector<int> comm_devices;
    std::vector<ncclComm_t> comms;
    // A failure inside a grouped RCCL call leaves the peers' matching operations without a partner: the communicators are aborted
    // and every later call on this group fails fast with NRAYS_ERR_RCCL instead of hanging.
    bool poisoned = false;
    int comm_index_of_device(int dev) const {
        for (size_t i = 0; i < comm_devices.size(); ++i) if (comm_devices[i] == dev) return (int)i;
        return -1;
    }
};

struct NraysSceneSet {
    NraysComm* comm = nullptr;
    struct Owner {
        uint32_t index = 0;       // band owner
        int device = 0;
        NraysScene* scene = nullptr;
        float* tile[2] = {nullptr, nullptr};
        hipStream_t render_stream = nullptr, comm_stream = nullptr;
        hipEvent_t rendered[2] = {nullptr, nullptr};  // tile[slot] is complete
        hipEvent_t sent[2] = {nullptr, nullptr};      // tile[slot] has left (may be rendered into again)
    };
    // nrays_multi_get_timings: timing-enabled events around the first local owner's stages, a ring of the last frames
    static constexpr int kTimed = 64;
    hipEvent_t t_render0[kTimed] = {}, t_render1[kTimed] = {}, t_exch0[kTimed] = {}, t_exch1[kTimed] = {}, t_untile1[kTimed] = {};
    bool t_has_exchange[kTimed] = {}, t_has_untile[kTimed] = {};
    uint64_t t_recorded = 0, t_reported = 0;
    bool t_ready = false;
    std::vector<Owner> local;     // the owners this process drives
    // owner 0's side (present iff this process drives owner 0)
    float* gathered[2] = {nullptr, nullptr};
    float* frame = nullptr; size_t frame_floats = 0;
    size_t tile_floats = 0, gathered_floats = 0;
    uint32_t width = 0, height = 0;
    uint64_t step = 0;
    bool direct = false; // NRAYS_MULTI_DIRECT=1: every band travels straight into its rows of the frame (no gather buffer, no k_untile pass)
    bool has_root() const { return !local.empty() && local[0].index == 0; }
};

extern "C" {

int nrays_comm_unique_id(uint8_t out_id[NRAYS_UNIQUE_ID_BYTES]) {
    if (!out_id) return nrays::set_last_error(NRAYS_ERR_BAD_ARG, "null argument");
    static_assert(NRAYS_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    MG_NCCL(ncclGetUniqueId(&id));
    std::memcpy(out_id, id.internal, NCCL_UNIQUE_ID_BYTES);
    return NRAYS_OK;
}

int nrays_comm_create(const uint8_t id_bytes[NRAYS_UNIQUE_ID_BYTES], uint32_t num_ranks, uint32_t rank, NraysComm** out_comm) {
    if (!id_bytes || !out_comm || num_ranks == 0 || rank >= num_ranks) return nrays::set_last_error(NRAYS_ERR_BAD_ARG, "bad communicator arguments");
    *out_comm = nullptr;
    int dev = 0;
    MG_HIP(hipGetDevice(&dev));
    NraysComm* c = new NraysComm();
    c->ranked = true; c->owners = num_ranks; c->rank = rank; c->devices = {dev};
    if (num_ranks > 1) {
        ncclUniqueId id;
        std::memcpy(id.internal, id_bytes, NCCL_UNIQUE_ID_BYTES);
        ncclComm_t comm;
        ncclResult_t r = ncclCommInitRank(&comm, (int)num_ranks, id, (int)rank);
        if (r != ncclSuccess) { delete c; return nrays::set_last_error(NRAYS_ERR_RCCL, std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); }
        c->comm_devices = {dev}; c->comms = {comm};
    }
    *out_comm = c;
    return NRAYS_OK;
}

int nrays_comm_create_local(uint32_t num_owners, const int32_t* devices, NraysComm** out_comm) {
    if (!out_comm || num_owners == 0) return nrays::set_last_error(NRAYS_ERR_BAD_ARG, "bad communicator arguments");
    *out_comm = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return nrays::set_last_error(NRAYS_ERR_NO_DEVICE, "no HIP device visible");
    NraysComm* c = new NraysComm();
    c->owners = num_owners;
    for (uint32_t o = 0; o < num_owners; ++o) {
        int d = devices ? devices[o] : (int)(o % (uint32_t)ndev);
        if (d < 0 || d >= ndev) { delete c; return nrays::set_last_error(NRAYS_ERR_BAD_ARG, "device index out of range"); }
        c->devices.push_back(d);
        if (c->comm_index_of_device(d) < 0) c->comm_devices.push_back(d);
    }
    if (c->comm_devices.size() > 1) { // a real exchange between GPUs: one RCCL rank per distinct device
        c->comms.resize(c->comm_devices.size());
        ncclResult_t r = ncclCommInitAll(c->comms.data(), (int)c->comm_devices.size(), c->comm_devices.data());
        if (r != ncclSuccess) { delete c; return nrays::set_last_error(NRAYS_ERR_RCCL, std::string("ncclCommInitAll: ") + ncclGetErrorString(r)); }
    }
    *out_comm = c;
    return NRAYS_OK;
}

// ncclCommAbort releases THIS process's communicators (nothing is left to destroy afterwards) and makes later calls on the group fail fast.
// It does not release a peer process that already sits in its own grouped call or stream kernel: peers have to time out or be aborted by
// their own process (in one-process mode every owner's communicator is aborted here).
static void poison(NraysComm* c) {
    if (c->poisoned) return;
    c->poisoned = true;
    for (size_t i = 0; i < c->comms.size(); ++i) { (void)hipSetDevice(c->comm_devices[i]); (void)ncclCommAbort(c->comms[i]); }
    c->comms.clear();
}

void nrays_comm_destroy(NraysComm* c) {
    if (!c) return;
    DeviceGuard guard;
    for (size_t i = 0; i < c->comms.size(); ++i) { (void)hipSetDevice(c->comm_devices[i]); (void)ncclCommDestroy(c->comms[i]); }
    delete c;
}

uint32_t nrays_comm_owners(const NraysComm* c) { return c ? c->owners : 0; }

uint32_t nrays_scene_set_num_local(const NraysSceneSet* s) { return s ? (uint32_t)s->local.size() : 0; }
NraysScene* nrays_scene_set_local_scene(NraysSceneSet* s, uint32_t k, uint32_t* out_owner) {
    if (!s || k >= s->local.size()) return nullptr;
    if (out_owner) *out_owner = s->local[k].index;
    return s->local[k].scene;
}

void nrays_scene_set_destroy(NraysSceneSet* s) {
    if (!s) return;
    for (auto& o : s->local) {
        (void)hipSetDevice(o.device);
        if (o.render_stream) (void)hipStreamSynchronize(o.render_stream);
        if (o.comm_stream) (void)hipStreamSynchronize(o.comm_stream);
        if (o.scene) nrays_scene_destroy(o.scene);
        for (int k = 0; k < 2; ++k) {
            if (o.tile[k]) (void)hipFree(o.tile[k]);
            if (o.rendered[k]) (void)hipEventDestroy(o.rendered[k]);
            if (o.sent[k]) (void)hipEventDestroy(o.sent[k]);
        }
        if (o.render_stream) (void)hipStreamDestroy(o.render_stream);
        if (o.comm_stream) (void)hipStreamDestroy(o.comm_stream);
    }
    if (s->t_ready && !s->local.empty()) {
        (void)hipSetDevice(s->local[0].device);
        for (int k = 0; k < NraysSceneSet::kTimed; ++k) {
            (void)hipEventDestroy(s->t_render0[k]); (void)hipEventDestroy(s->t_render1[k]); (void)hipEventDestroy(s->t_exch0[k]);
            (void)hipEventDestroy(s->t_exch1[k]); (void)hipEventDestroy(s->t_untile1[k]);
        }
    }
    if (s->has_root()) {
        (void)hipSetDevice(s->local[0].device);
        for (int k = 0; k < 2; ++k) if (s->gathered[k]) (void)hipFree(s->gathered[k]);
        if (s->frame) (void)hipFree(s->frame);
    }
    delete s;
}

// Replaces Scene::new (src/scene.rs:119-133) for a group: flattens and uploads the scene on every GPU this process drives.
int nrays_scene_set_create(const NraysSceneDesc* desc, NraysComm* comm, NraysSceneSet** out_set) {
    if (!desc || !comm || !out_set) return nrays::set_last_error(NRAYS_ERR_BAD_ARG, "null argument");
    *out_set = nullptr;
    NraysSceneSet* s = new NraysSceneSet();
    s->comm = comm;
    if (const char* e = getenv("NRAYS_MULTI_DIRECT")) s->direct = atoi(e) != 0;
    DeviceGuard guard;
    auto bail = [&](int rc) { nrays_scene_set_destroy(s); return rc; };
    const uint32_t first = comm->ranked ? comm->rank : 0u, count = comm->ranked ? 1u : comm->owners;
    for (uint32_t k = 0; k < count; ++k) {
        NraysSceneSet::Owner o;
        o.index = first + k;
        o.device = comm->ranked ? comm->devices[0] : comm->devices[o.index];
        if (hipSetDevice(o.device) != hipSuccess) return bail(nrays::set_last_error(NRAYS_ERR_HIP, "hipSetDevice failed"));
        int rc = nrays_scene_create(desc, &o.scene);
        if (rc != NRAYS_OK) { s->local.push_back(o); return bail(rc); }
        if (hipStreamCreateWithFlags(&o.render_stream, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&o.comm_stream, hipStreamNonBlocking) != hipSuccess) { s->local.push_back(o); return bail(nrays::set_last_error(NRAYS_ERR_HIP, "stream creation failed")); }
        for (int b = 0; b < 2; ++b)
            if (hipEventCreateWithFlags(&o.rendered[b], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&o.sent[b], hipEventDisableTiming) != hipSuccess) { s->local.push_back(o); return bail(nrays::set_last_error(NRAYS_ERR_HIP, "event creation failed")); }
        s->local.push_back(o);
    }
    if (!s->local.empty()) { // timing events of the first local owner (nrays_multi_get_timings)
        if (hipSetDevice(s->local[0].device) != hipSuccess) return bail(nrays::set_last_error(NRAYS_ERR_HIP, "hipSetDevice failed"));
        bool ok = true;
        for (int k = 0; k < NraysSceneSet::kTimed && ok; ++k)
            ok = hipEventCreate(&s->t_render0[k]) == hipSuccess && hipEventCreate(&s->t_render1[k]) == hipSuccess && hipEventCreate(&s->t_exch0[k]) == hipSuccess &&
                 hipEventCreate(&s->t_exch1[k]) == hipSuccess && hipEventCreate(&s->t_untile1[k]) == hipSuccess;
        if (!ok) return bail(nrays::set_last_error(NRAYS_ERR_HIP, "event creation failed"));
        s->t_ready = true;
    }
    *out_set = s;
    return NRAYS_OK;
}

// Floats of one owner's compact tile for this frame (every owner's tile has the same, padded, size).
static size_t tile_floats_of(const NraysSceneSet* s, const NraysRenderParams* p) {
    NraysRenderParams q = *p;
    q.band_rows = s->comm->owners > 1 ? kBandRows : 0; q.band_owner = 0; q.band_owners = s->comm->owners;
    return (size_t)nrays_tile_rows(&q) * p->width * 3;
}

static int ensure_buffers(NraysSceneSet* s, const NraysRenderParams* p) {
    const uint32_t owners = s->comm->owners;
    const size_t tile_floats = tile_floats_of(s, p);
    if (tile_floats > s->tile_floats) {
        for (auto& o : s->local) {
            MG_HIP(hipSetDevice(o.device));
            MG_HIP(hipStreamSynchronize(o.render_stream)); MG_HIP(hipStreamSynchronize(o.comm_stream));
            for (int k = 0; k < 2; ++k) {
                if (o.tile[k]) { (void)hipFree(o.tile[k]); o.tile[k] = nullptr; }
                MG_HIP(hipMalloc((void**)&o.tile[k], tile_floats * sizeof(float)));
            }
        }
        s->tile_floats = tile_floats;
    }
    if (s->has_root() && owners > 1) {
        const size_t need = tile_floats * owners;
        if (need > s->gathered_floats) {
            MG_HIP(hipSetDevice(s->local[0].device));
            for (int k = 0; k < 2; ++k) {
                if (s->gathered[k]) { (void)hipFree(s->gathered[k]); s->gathered[k] = nullptr; }
                MG_HIP(hipMalloc((void**)&s->gathered[k], need * sizeof(float)));
            }
            s->gathered_floats = need;
        }
    }
    return NRAYS_OK;
}

// One frame: every local owner renders its tile, the tiles travel to owner 0, k_untile writes the row-major frame into
// `out_rgb_device` (device memory on owner 0's GPU; ignored by processes that do not drive owner 0).  Asynchronous:
// nrays_multi_sync waits.  The caller must not touch `out_rgb_device` of frame k before that, nor reuse it for frame
// k + 1 unless it is read on owner 0's communication stream order (nrays_render_multi does exactly this).
int nrays_render_multi_device(NraysSceneSet* s, const NraysRenderParams* params, float* out_rgb_device) {
    if (!s || !params) return nrays::set_last_error(NRAYS_ERR_BAD_ARG, "null argument");
    if (s->has_root() && !out_rgb_device) return nrays::set_last_error(NRAYS_ERR_BAD_ARG, "owner 0 needs an output buffer");
    if (params->width == 0 || params->height == 0 || params->ray_per_pixel == 0) return nrays::set_last_error(NRAYS_ERR_BAD_ARG, "bad render parameters");
    NraysComm* c = s->comm;
    if (c->poisoned) return nrays::set_last_error(NRAYS_ERR_RCCL, "the communicator was aborted after a failed exchange: destroy the scene set and the communicator");
    const uint32_t owners = c->owners;
    DeviceGuard guard;
    int rc = ensure_buffers(s, params);
    if (rc != NRAYS_OK) return rc;
    const int slot = (int)(s->step & 1u);
    const size_t count = tile_floats_of(s, params); // this frame's tile size (the buffers may be larger: they only grow)
    const int tslot = (int)(s->t_recorded % NraysSceneSet::kTimed);
    s->t_has_exchange[tslot] = false; s->t_has_untile[tslot] = false;

    // 1. tile renders, each on its owner's render stream
    for (auto& o : s->local) {
        MG_HIP(hipSetDevice(o.device));
        NraysRenderParams q = *params;
        q.band_rows = owners > 1 ? kBandRows : 0; q.band_owner = o.index; q.band_owners = owners;
        if (s->step >= 2) MG_HIP(hipStreamWaitEvent(o.render_stream, o.sent[slot], 0)); // tile[slot] of frame k - 2 has left
        float* dst = (owners == 1) ? out_rgb_device : o.tile[slot];
        const bool timed_owner = s->t_ready && &o == &s->local[0];
        if (timed_owner) MG_HIP(hipEventRecord(s->t_render0[tslot], o.render_stream));
        rc = nrays_render_device(o.scene, &q, dst, (void*)o.render_stream);
        if (rc != NRAYS_OK) return rc;
        if (timed_owner) MG_HIP(hipEventRecord(s->t_render1[tslot], o.render_stream));
        MG_HIP(hipEventRecord(o.rendered[slot], o.render_stream));
        MG_HIP(hipStreamWaitEvent(o.comm_stream, o.rendered[slot], 0));
    }
    if (owners > 1 && s->direct) {
        // 2'. the exchange WITHOUT the gather buffer: local band lb of owner o IS rows (lb * owners + o) * 16 ... of the frame, contiguous on both
        // sides, so it can land where it belongs — same-device owners with one strided 2-D copy each, the others with one ncclSend / ncclRecv
        // per band in ONE group (RCCL fuses the operations of a peer) — and there is no un-permute pass.  A/B against the gather form:
        // profiles/r05_multi_direct_ab.log.
        const uint32_t W = params->width, H = params->height;
        const size_t band_floats = (size_t)kBandRows * W * 3;
        auto bands_of = [&](uint32_t o) -> uint32_t { const uint32_t nb = (H + kBandRows - 1) / kBandRows; return nb > o ? (nb - o + owners - 1) / owners : 0u; };
        auto rows_of = [&](uint32_t o, uint32_t lb) -> uint32_t { const uint32_t r0 = (lb * owners + o) * kBandRows; return r0 >= H ? 0u : (H - r0 < kBandRows ? H - r0 : kBandRows); };
        const int root_dev = c->ranked ? -1 : c->devices[0];
        NraysSceneSet::Owner* root = s->has_root() ? &s->local[0] : nullptr;
        if (s->t_ready) { MG_HIP(hipSetDevice(s->local[0].device)); MG_HIP(hipEventRecord(s->t_exch0[tslot], s->local[0].comm_stream)); s->t_has_exchange[tslot] = true; }
        bool any_rccl = c->ranked;
        for (auto& o : s->local) { // owners on owner 0's device (owner 0 itself among them): copies on its communication stream
            if (c->ranked ? o.index != 0 : o.device != root_dev) { any_rccl = true; continue; }
            if (!root) continue;
            MG_HIP(hipSetDevice(root->device));
            MG_HIP(hipStreamWaitEvent(root->comm_stream, o.rendered[slot], 0));
            const uint32_t nb = bands_of(o.index);
            const uint32_t full = nb && rows_of(o.index, nb - 1) == kBandRows ? nb : (nb ? nb - 1 : 0u);
            float* dst0 = out_rgb_device + (size_t)o.index * band_floats;
            if (full) MG_HIP(hipMemcpy2DAsync(dst0, (size_t)owners * band_floats * sizeof(float), o.tile[slot], band_floats * sizeof(float), band_floats * sizeof(float), full, hipMemcpyDeviceToDevice, root->comm_stream));
            if (full < nb) MG_HIP(hipMemcpyAsync(dst0 + (size_t)full * owners * band_floats, o.tile[slot] + (size_t)full * band_floats, (size_t)rows_of(o.index, full) * W * 3 * sizeof(float), hipMemcpyDeviceToDevice, root->comm_stream));
        }
        if (any_rccl && !c->comms.empty()) {
            MG_NCCL(ncclGroupStart());
            int group_rc = NRAYS_OK; std::string group_msg;
            auto hip_ok = [&](hipError_t e, const char* what) { if (e != hipSuccess && group_rc == NRAYS_OK) { group_rc = NRAYS_ERR_HIP; group_msg = std::string(what) + ": " + hipGetErrorString(e); } return e == hipSuccess; };
            auto nccl_ok = [&](ncclResult_t r, const char* what) { if (r != ncclSuccess && group_rc == NRAYS_OK) { group_rc = NRAYS_ERR_RCCL; group_msg = std::string(what) + ": " + ncclGetErrorString(r); } return r == ncclSuccess; };
            auto recv_bands = [&](uint32_t o, int peer, ncclComm_t comm) {
                for (uint32_t lb = 0; lb < bands_of(o) && group_rc == NRAYS_OK; ++lb)
                    nccl_ok(ncclRecv(out_rgb_device + ((size_t)lb * owners + o) * band_floats, (size_t)rows_of(o, lb) * W * 3, ncclFloat, peer, comm, root->comm_stream), "ncclRecv");
            };
            auto send_bands = [&](NraysSceneSet::Owner& o, int peer, ncclComm_t comm) {
                for (uint32_t lb = 0; lb < bands_of(o.index) && group_rc == NRAYS_OK; ++lb)
                    nccl_ok(ncclSend(o.tile[slot] + (size_t)lb * band_floats, (size_t)rows_of(o.index, lb) * W * 3, ncclFloat, peer, comm, o.comm_stream), "ncclSend");
            };
            if (c->ranked) {
                if (c->rank == 0) { for (uint32_t r = 1; r < owners && group_rc == NRAYS_OK; ++r) recv_bands(r, (int)r, c->comms[0]); }
                else send_bands(s->local[0], 0, c->comms[0]);
            } else {
                const int root_ci = c->comm_index_of_device(root_dev);
                for (auto& o : s->local) {
                    if (o.index == 0 || o.device == root_dev) continue;
                    if (group_rc != NRAYS_OK) break;
                    const int ci = c->comm_index_of_device(o.device);
                    if (!hip_ok(hipSetDevice(o.device), "hipSetDevice")) break;
                    send_bands(o, root_ci, c->comms[ci]);
                    if (!hip_ok(hipSetDevice(root_dev), "hipSetDevice")) break;
                    recv_bands(o.index, ci, c->comms[root_ci]);
                }
            }
            nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
            if (group_rc != NRAYS_OK) {
                poison(c);
                return nrays::set_last_error(group_rc, group_msg + " (exchange aborted; the communicator is no longer usable)");
            }
        }
        for (auto& o : s->local) { // tile[slot] may be rendered into again once its copy / send is done
            MG_HIP(hipSetDevice(o.device));
            const bool on_root = root && (c->ranked ? o.index == 0 : o.device == root_dev);
            MG_HIP(hipEventRecord(o.sent[slot], on_root ? root->comm_stream : o.comm_stream));
        }
        if (s->t_ready) { MG_HIP(hipSetDevice(s->local[0].device)); MG_HIP(hipEventRecord(s->t_exch1[tslot], s->local[0].comm_stream)); }
    } else if (owners > 1) {
        // 2. the exchange: same-device owners copy, the others send / owner 0 receives (one grouped RCCL call)
        const int root_dev = c->ranked ? -1 : c->devices[0];
        NraysSceneSet::Owner* root = s->has_root() ? &s->local[0] : nullptr;
        if (s->t_ready) { MG_HIP(hipSetDevice(s->local[0].device)); MG_HIP(hipEventRecord(s->t_exch0[tslot], s->local[0].comm_stream)); s->t_has_exchange[tslot] = true; }
        bool any_rccl = false;
        for (auto& o : s->local) {
            if (o.index == 0) continue;
            if (!c->ranked && o.device == root_dev) {
                MG_HIP(hipSetDevice(root_dev));
                // ordered on owner 0's communication stream after this owner's render
                MG_HIP(hipStreamWaitEvent(root->comm_stream, o.rendered[slot], 0));
                MG_HIP(hipMemcpyAsync(s->gathered[slot] + (size_t)o.index * count, o.tile[slot], count * sizeof(float), hipMemcpyDeviceToDevice, root->comm_stream));
            } else any_rccl = true;
        }
        if (c->ranked) any_rccl = true;
        if (root) { // owner 0's own tile
            MG_HIP(hipSetDevice(root->device));
            MG_HIP(hipMemcpyAsync(s->gathered[slot], root->tile[slot], count * sizeof(float), hipMemcpyDeviceToDevice, root->comm_stream));
        }
        if (any_rccl && !c->comms.empty()) {
            MG_NCCL(ncclGroupStart());
            // Between ncclGroupStart and ncclGroupEnd nothing may return: the first failure is remembered, the group is ALWAYS
            // closed, and a failed exchange aborts THIS process's communicators (poison): a retry fails fast; peer processes time out or are aborted by their own process.
            int group_rc = NRAYS_OK; std::string group_msg;
            auto hip_ok = [&](hipError_t e, const char* what) { if (e != hipSuccess && group_rc == NRAYS_OK) { group_rc = NRAYS_ERR_HIP; group_msg = std::string(what) + ": " + hipGetErrorString(e); } return e == hipSuccess; };
            auto nccl_ok = [&](ncclResult_t r, const char* what) { if (r != ncclSuccess && group_rc == NRAYS_OK) { group_rc = NRAYS_ERR_RCCL; group_msg = std::string(what) + ": " + ncclGetErrorString(r); } return r == ncclSuccess; };
            if (c->ranked) {
                if (c->rank == 0) {
                    for (uint32_t r = 1; r < owners && group_rc == NRAYS_OK; ++r)
                        nccl_ok(ncclRecv(s->gathered[slot] + (size_t)r * count, count, ncclFloat, (int)r, c->comms[0], root->comm_stream), "ncclRecv");
                } else {
                    nccl_ok(ncclSend(s->local[0].tile[slot], count, ncclFloat, 0, c->comms[0], s->local[0].comm_stream), "ncclSend");
                }
            } else {
                const int root_ci = c->comm_index_of_device(root_dev);
                for (auto& o : s->local) {
                    if (o.index == 0 || o.device == root_dev) continue;
                    if (group_rc != NRAYS_OK) break;
                    const int ci = c->comm_index_of_device(o.device);
                    if (!hip_ok(hipSetDevice(o.device), "hipSetDevice")) break;
                    if (!nccl_ok(ncclSend(o.tile[slot], count, ncclFloat, root_ci, c->comms[ci], o.comm_stream), "ncclSend")) break;
                    if (!hip_ok(hipSetDevice(root_dev), "hipSetDevice")) break;
                    if (!nccl_ok(ncclRecv(s->gathered[slot] + (size_t)o.index * count, count, ncclFloat, ci, c->comms[root_ci], root->comm_stream), "ncclRecv")) break;
                }
            }
            nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
            if (group_rc != NRAYS_OK) {
                poison(c);
                return nrays::set_last_error(group_rc, group_msg + " (exchange aborted; the communicator is no longer usable)");
            }
        }
        for (auto& o : s->local) { // tile[slot] may be rendered into again once its copy / send is done
            MG_HIP(hipSetDevice(o.device));
            hipStream_t done_on = (!c->ranked && o.device == root_dev && root) ? root->comm_stream : o.comm_stream;
            if (o.index == 0 && root) done_on = root->comm_stream;
            MG_HIP(hipEventRecord(o.sent[slot], done_on));
        }
        if (s->t_ready) { MG_HIP(hipSetDevice(s->local[0].device)); MG_HIP(hipEventRecord(s->t_exch1[tslot], s->local[0].comm_stream)); }
        // 3. un-permute the bands on owner 0
        if (root) {
            MG_HIP(hipSetDevice(root->device));
            rc = nrays_untile_device(s->gathered[slot], out_rgb_device, params->width, params->height, kBandRows, owners, (void*)root->comm_stream);
            if (rc != NRAYS_OK) return rc;
            if (s->t_ready) { MG_HIP(hipEventRecord(s->t_untile1[tslot], root->comm_stream)); s->t_has_untile[tslot] = true; }
        }
    }
    s->width = params->width; s->height = params->height;
    s->step++;
    if (s->t_ready) s->t_recorded++;
    return NRAYS_OK;
}

int nrays_multi_sync(NraysSceneSet* s) {
    if (!s) return nrays::set_last_error(NRAYS_ERR_BAD_ARG, "null argument");
    DeviceGuard guard;
    for (auto& o : s->local) {
        MG_HIP(hipSetDevice(o.device));
        MG_HIP(hipStreamSynchronize(o.render_stream));
        MG_HIP(hipStreamSynchronize(o.comm_stream));
    }
    return NRAYS_OK;
}

// Replaces scene::render (src/scene.rs:29-116) on the whole group: `out_rgb` is HOST memory (height*width*3 floats) and
// is filled on the process that drives owner 0 (it may be NULL elsewhere).  Blocking.
int nrays_render_multi(NraysSceneSet* s, const NraysRenderParams* params, float* out_rgb) {
    if (!s || !params) return nrays::set_last_error(NRAYS_ERR_BAD_ARG, "null argument");
    float* dev_out = nullptr;
    if (s->has_root()) {
        if (!out_rgb) return nrays::set_last_error(NRAYS_ERR_BAD_ARG, "owner 0 needs an output buffer");
        const size_t floats = (size_t)params->width * params->height * 3;
        DeviceGuard guard;
        MG_HIP(hipSetDevice(s->local[0].device));
        if (floats > s->frame_floats) {
            if (s->frame) { (void)hipFree(s->frame); s->frame = nullptr; s->frame_floats = 0; }
            MG_HIP(hipMalloc((void**)&s->frame, floats * sizeof(float)));
            s->frame_floats = floats;
        }
        dev_out = s->frame;
    }
    int rc = nrays_render_multi_device(s, params, dev_out);
    if (rc != NRAYS_OK) return rc;
    if (s->has_root()) {
        NraysSceneSet::Owner& root = s->local[0];
        DeviceGuard guard;
        MG_HIP(hipSetDevice(root.device));
        hipStream_t last = s->comm->owners > 1 ? root.comm_stream : root.render_stream;
        MG_HIP(hipMemcpyAsync(out_rgb, s->frame, (size_t)params->width * params->height * 3 * sizeof(float), hipMemcpyDeviceToHost, last));
    }
    return nrays_multi_sync(s);
}

int nrays_multi_get_timings(NraysSceneSet* s, NraysMultiTimings* out) {
    if (!s || !out) return nrays::set_last_error(NRAYS_ERR_BAD_ARG, "null argument");
    std::memset(out, 0, sizeof *out);
    if (!s->t_ready || s->local.empty()) return NRAYS_OK;
    int rc = nrays_multi_sync(s);
    if (rc != NRAYS_OK) return rc;
    DeviceGuard guard;
    MG_HIP(hipSetDevice(s->local[0].device));
    uint64_t first = s->t_reported;
    if (s->t_recorded - first > (uint64_t)NraysSceneSet::kTimed) first = s->t_recorded - NraysSceneSet::kTimed;
    double r = 0.0, e = 0.0, u = 0.0; uint32_t n = 0;
    for (uint64_t f = first; f < s->t_recorded; ++f) {
        const int k = (int)(f % NraysSceneSet::kTimed);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s->t_render0[k], s->t_render1[k]) != hipSuccess) continue;
        r += ms;
        if (s->t_has_exchange[k] && hipEventElapsedTime(&ms, s->t_exch0[k], s->t_exch1[k]) == hipSuccess) e += ms;
        if (s->t_has_untile[k] && hipEventElapsedTime(&ms, s->t_exch1[k], s->t_untile1[k]) == hipSuccess) u += ms;
        ++n;
    }
    s->t_reported = s->t_recorded;
    if (n) { out->render_ms = r / n; out->exchange_ms = e / n; out->untile_ms = u / n; }
    out->frames = n; out->owner = s->local[0].index;
    return NRAYS_OK;
}

// Ray-class counters of the last frame, summed over the owners this process drives (kernel timings: owner-0-local
// averages, see nrays_get_stats).
int nrays_multi_get_stats(NraysSceneSet* s, NraysStats* out) {
    if (!s || !out) return nrays::set_last_error(NRAYS_ERR_BAD_ARG, "null argument");
    std::memset(out, 0, sizeof *out);
    DeviceGuard guard;
    for (size_t k = 0; k < s->local.size(); ++k) {
        NraysStats st;
        (void)hipSetDevice(s->local[k].device);
        int rc = nrays_get_stats(s->local[k].scene, &st);
        if (rc != NRAYS_OK) return rc;
        out->rays_primary += st.rays_primary; out->rays_reflection += st.rays_reflection; out->rays_refraction += st.rays_refraction;
        out->rays_shadow += st.rays_shadow; out->node_tests += st.node_tests; out->tri_tests += st.tri_tests; out->prim_tests += st.prim_tests;
        out->hit_records += st.hit_records; out->tex_samples += st.tex_samples; out->rays_primary_traced += st.rays_primary_traced; out->rays_shadow_elided += st.rays_shadow_elided; out->node_fetches += st.node_fetches;
        out->generations = std::max(out->generations, st.generations);
        if (k == 0) { out->kernel_ms_primary = st.kernel_ms_primary; out->kernel_ms_total = st.kernel_ms_total; out->frames_timed = st.frames_timed; out->instrumented = st.instrumented; }
    }
    return NRAYS_OK;
}

} // extern "C"
