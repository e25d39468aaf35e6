// stream_device.h — the two BVT queries of the trace loop (ClosestRayTOICostFn, src/scene.rs:262-283;
// TransparentShadowsRayTOICostFn, src/scene.rs:285-339) over a STREAM of rays with lane refill: every lane of a wave holds
// one ray's traversal; a lane whose ray is finished hands its result over and takes the next ray of the stream, while the
// other lanes keep their place in their own traversals (a resumable per-lane traversal inside a per-lane state machine).
//
// traverse() (trace_device.h) runs 64 rays from start to end together: a wave iterates until its SLOWEST lane is done —
// in a hair-like mesh 17 - 31 % of the lanes are active per node step, the rest have finished or missed.  Here the node /
// leaf phases are the same code ("while-while", quorum-ended node phases, LDS stack, deferred exact gates: see traverse()),
// but between two rounds of phases the wave looks at its lanes: finished ones deliver (a wave-uniform point: results are
// compacted by ballot -> prefix into the next queue), and once kRefillMin of them are free the stream fills them again.
// Results do not depend on which rays share a wave or on the visiting order (ties: D-2), so the rays' results — and the
// frames — are bit-identical to traverse()'s.  The wave-uniform scalar node fetch survives where the lanes still agree.
//
// The stream is a policy object `Source` (wavefront.hip):
//   bool more()                                   wave-uniform: rays may still come
//   void refill(bool idle, bool& got, d3& o, d3& d, double& tlimit)
//                                                 whole wave; an idle lane may receive a ray (got = true)
//   bool deliver(bool fin, d3 o, d3 d, bool hit, const Hit& h, bool blocked, f3 filter, bool gated)
//                                                 whole wave; `fin` lanes hand over their result; returns true for a lane whose
//                                                 closest hit failed the reference's exact AABB gates and must run again, gated
#pragma once
#include "trace_device.h"

#ifndef NR_WF_REFILL_MIN
#define NR_WF_REFILL_MIN 20 // lanes that must be free before the stream fills them (a refill costs the wave a ray set-up)
#endif

namespace nrays {

constexpr int32_t kIdle = (int32_t)0x80000003; // the lane holds no ray
constexpr int32_t kFin = (int32_t)0x80000004;  // the lane's ray is finished, its result not yet delivered

template <bool SHADOW, int FEAT, class Source>
NR_DEV void traverse_stream(const DScene& S, Stack& st, Source& src, Cnt& cnt) {
    constexpr bool kAnalytic = (FEAT & kFeatAnalytic) != 0, kMesh = (FEAT & kFeatMesh) != 0;
    constexpr bool kAlpha = (FEAT & kFeatAlphaShadow) != 0; // shadow mode: otherwise every hit within tlimit blocks
    constexpr bool kQuorum = NR_NODE_QUORUM_DEN != 0 && kMesh && !kAnalytic && !kAlpha;
    const Instance* insts = SHADOW ? S.shadow_instances : S.instances;
    const InstLink* links = SHADOW ? S.shadow_links : S.links;
    // per-lane state of one ray's traversal (traverse(): the locals of one call)
    d3 o = D3(0.0, 0.0, 0.0), d = D3(0.0, 0.0, 1.0), co = o, cd = d;
    RayF rf = make_rayf(o, d);
    double tlimit = 0.0, bt = 0.0;
    float btf = 0.0f;
    unsigned long long bkey = ~0ULL;
    bool bhit = false, in_blas = false, gated = false, blocked = false;
    uint32_t binst = 0, bprim = 0, cur_inst = 0, cur_flags = 0;
    f3 filter = F3(1.0f, 1.0f, 1.0f);
    int32_t cur = kIdle;
    auto start = [&]() { // o, d, tlimit, gated are set
        co = o; cd = d; rf = make_rayf(o, d);
        bt = SHADOW ? tlimit : kDblMax; btf = best_f32(bt); bkey = ~0ULL; bhit = false; binst = 0; bprim = 0;
        in_blas = false; cur_inst = 0; cur_flags = 0; blocked = false; filter = F3(1.0f, 1.0f, 1.0f);
        st.reset();
        if (kAnalytic) {
            const int32_t* planes = SHADOW ? S.shadow_planes : S.planes;
            for (uint32_t p = 0; p < S.num_planes; ++p) st.push(~(int32_t)(((uint32_t)planes[p]) << 3));
        }
        cur = SHADOW ? S.shadow_root : S.closest_root;
        if (cur == kEmptyChild) cur = st.pop();
    };
    for (;;) {
        // ---- wave-uniform point: results out, new rays in — only once kRefillMin lanes are free (finished or empty) or nothing else
        // is left to do: the hand-over (exact gates of a closest hit, ray set-up) is f64 code that should run on many lanes at once
        const unsigned long long fin_m = __ballot(cur == kFin), idle_m0 = __ballot(cur == kIdle), all_m = __ballot(1);
        const bool more = src.more();
        if ((fin_m | idle_m0) != 0ULL && (__popcll(fin_m | idle_m0) >= NR_WF_REFILL_MIN || (fin_m | idle_m0) == all_m || (!more && fin_m != 0ULL && __popcll(fin_m) >= NR_WF_REFILL_MIN / 2))) {
            if (fin_m != 0ULL) {
                const bool fin = cur == kFin;
                Hit h; h.t = bt; h.inst = binst; h.prim = bprim;
                const bool again = src.deliver(fin, o, d, bhit, h, blocked, filter, gated);
                if (fin) { if (again) { gated = true; start(); } else cur = kIdle; }
            }
            if (more) {
                bool got = false;
                d3 no = o, nd = d; double ntl = tlimit;
                src.refill(cur == kIdle, got, no, nd, ntl);
                if (got) { o = no; d = nd; tlimit = ntl; gated = false; start(); }
            }
        }
        if (__ballot(cur != kIdle) == 0ULL) { if (src.more()) continue; else break; } // wave-uniform
        const bool GATED = SHADOW || gated;
        // ---- node phase (traverse(): "while-while")
        const int node_quorum = kQuorum && S.incoherent ? __popcll(__ballot(cur != kIdle && cur != kFin)) / NR_NODE_QUORUM_DEN : 0;
        while (cur >= 0) {
            const float kMiss = __builtin_inff();
            float k0, k1, k2, k3;
            int32_t c0, c1, c2, c3;
            const uint32_t nkey = ((uint32_t)cur << 7) | rf.bits;
            const uint32_t ukey = (uint32_t)__builtin_amdgcn_readfirstlane((int)nkey);
            if (NR_SCALAR_NODES && kMesh && __ballot(nkey != ukey) == 0ULL) { // wave-uniform: the active lanes share the node and the direction signs
                const int4 ch = load_children_uniform(S.nodes, ukey);
                const NodePlanes np = load_planes_uniform(S.nodes, ukey);
                box_keys4(np, rf, btf, k0, k1, k2, k3);
                if ((rf.bits & 7u)) zero_axis_cull((rf.bits & 7u), co, np, k0, k1, k2, k3);
                c0 = ch.x; c1 = ch.y; c2 = ch.z; c3 = ch.w;
            } else {
                const int4 ch = load_children(S.nodes, cur);
                const NodePlanes np = load_planes(S.nodes, cur, rf);
                box_keys4(np, rf, btf, k0, k1, k2, k3);
                if ((rf.bits & 7u)) zero_axis_cull((rf.bits & 7u), co, np, k0, k1, k2, k3);
                c0 = ch.x; c1 = ch.y; c2 = ch.z; c3 = ch.w;
            }
            if (!SHADOW) {
#define NR_CSWAP(ka, ca, kb, cb) { bool sw = kb < ka; float tk = sw ? kb : ka; kb = sw ? ka : kb; ka = tk; int32_t tc = sw ? cb : ca; cb = sw ? ca : cb; ca = tc; }
                NR_CSWAP(k0, c0, k1, c1) NR_CSWAP(k2, c2, k3, c3) NR_CSWAP(k0, c0, k2, c2) NR_CSWAP(k1, c1, k3, c3) NR_CSWAP(k1, c1, k2, c2)
#undef NR_CSWAP
                if (__builtin_expect(st.wave_has_room(3), 1)) {
                    lds_u32* a = st.top;
                    *a = (uint32_t)c3; a += k3 < kMiss ? kBlock : 0;
                    *a = (uint32_t)c2; a += k2 < kMiss ? kBlock : 0;
                    *a = (uint32_t)c1; a += k1 < kMiss ? kBlock : 0;
                    st.top = a;
                } else {
                    if (k3 < kMiss) st.push(c3);
                    if (k2 < kMiss) st.push(c2);
                    if (k1 < kMiss) st.push(c1);
                }
                if (k0 < kMiss) cur = c0;
                else cur = st.pop();
            } else {
                const bool h0 = k0 < kMiss, h1 = k1 < kMiss, h2 = k2 < kMiss, h3 = k3 < kMiss;
                cur = h3 ? c3 : (h2 ? c2 : (h1 ? c1 : (h0 ? c0 : kEmptyChild)));
                const bool p0 = h0 && (h1 || h2 || h3), p1 = h1 && (h2 || h3), p2 = h2 && h3;
                if (__builtin_expect(st.wave_has_room(3), 1)) {
                    lds_u32* a = st.top;
                    *a = (uint32_t)c0; a += p0 ? kBlock : 0;
                    *a = (uint32_t)c1; a += p1 ? kBlock : 0;
                    *a = (uint32_t)c2; a += p2 ? kBlock : 0;
                    st.top = a;
                } else {
                    if (p0) st.push(c0);
                    if (p1) st.push(c1);
                    if (p2) st.push(c2);
                }
                if (cur == kEmptyChild) cur = st.pop();
            }
            if (kQuorum && node_quorum && __popcll(__ballot(cur >= 0)) < node_quorum) { // wave-uniform
                if (cur >= 0) { st.push(cur); cur = kParked; }
            }
        }
        // ---- leaf phase
        if (cur == kIdle || cur == kFin) continue;
        if (kQuorum && cur == kParked) { cur = st.pop(); continue; } // sat out the leaf phase of the others
        if (cur == kEmptyChild) { cur = kFin; continue; }            // the stack is empty: this ray is done
        if (kMesh && cur == kSentinel) { // the BLAS of `cur_inst` is exhausted: back to world space
            in_blas = false;
            if (!(cur_flags & kInstNoXform)) { co = o; cd = d; rf = make_rayf(o, d); }
            bool stop = false;
            if (SHADOW && kAlpha && !(cur_flags & kInstAnyHit)) {
                if (bhit) {
                    Hit h; h.t = bt; h.inst = cur_inst; h.prim = bprim;
                    Isect is; uint32_t node_id;
                    resolve_hit<true, FEAT>(S, o, d, h, is, node_id);
                    if (shadow_node_hit<false>(S, node_id, is, filter, cnt)) stop = true;
                }
                bt = tlimit; bkey = ~0ULL; bhit = false; btf = best_f32(bt);
            }
            if (stop) { blocked = true; cur = kFin; } else cur = st.pop();
            continue;
        }
        const uint32_t lv = (uint32_t)~cur;
        const uint32_t first = lv >> 3, bits = lv & 7u;
        if (kMesh && in_blas) { // triangle leaf
            const float4* tq = (const float4*)(S.tris + first);
            float4 p0 = tq[0], p1 = tq[1], p2 = tq[2];
            const int32_t after_leaf = st.pop();
            bool stop = false;
#pragma nounroll
            for (uint32_t k = 0; k <= bits; ++k) {
                const float4 t0 = p0, t1 = p1, t2 = p2;
                if (k < bits) { tq += 3; p0 = tq[0]; p1 = tq[1]; p2 = tq[2]; }
                double toi;
                d3 va = D3(t0.x, t0.y, t0.z), vb = D3(t1.x, t1.y, t1.z), vc = D3(t2.x, t2.y, t2.z);
                if (cast_triangle(va, vb, vc, co, cd, toi, nullptr, nullptr) &&
                    (!GATED || (tri_aabb_pass(va, vb, vc, co, cd) && node_aabb_pass(S, __float_as_uint(t0.w), o, d)))) {
                    if (SHADOW && (!kAlpha || (cur_flags & kInstAnyHit))) { if (toi <= tlimit) { stop = true; break; } }
                    else {
                        unsigned long long key = SHADOW ? (unsigned long long)__float_as_uint(t1.w)
                                                        : (((unsigned long long)__float_as_uint(t0.w) << 32) | __float_as_uint(t1.w));
                        if (toi < bt || (toi == bt && key < bkey)) { bt = toi; bkey = key; bhit = true; binst = cur_inst; bprim = first + k; btf = best_f32(bt); }
                    }
                }
            }
            if (stop) { blocked = true; cur = kFin; } else cur = after_leaf;
            continue;
        }
        if (kMesh && (!kAnalytic || (bits & kLeafMesh))) { // TLAS leaf: a BLAS
            cur_inst = first;
            InstLink link = links[first];
            cur_flags = link.flags;
            if (!(bits & kLeafNoXform)) {
                const Instance& in = insts[first];
                Xform m; load_xform(in, m);
                if (in.flags & kInstIdentityRot) { co = o - m.t; cd = d; }
                else { co = inv_rot(m, o - m.t); cd = inv_rot(m, d); }
                rf = make_rayf(co, cd);
            }
            in_blas = true;
            st.push(kSentinel);
            cur = link.blas_root;
            if (cur == kEmptyChild) cur = st.pop();
            continue;
        }
        if (kAnalytic) { // TLAS leaf: an analytic shape (or a plane pseudo-leaf)
            const Instance& in = insts[first];
            bool stop = false;
            Isect is = cast_instance<FEAT>(in, o, d, !(SHADOW && !kAlpha));
            if (is.hit && (!GATED || in.kind == NRAYS_SHAPE_PLANE || node_aabb_pass(S, (uint32_t)in.node_id, o, d))) {
                if (SHADOW) {
                    if (is.toi <= tlimit) {
                        if (!kAlpha) stop = true;
                        else if (shadow_node_hit<false>(S, (uint32_t)in.node_id, is, filter, cnt)) stop = true;
                    }
                } else {
                    unsigned long long key = (unsigned long long)(uint32_t)in.node_id << 32;
                    if (is.toi < bt || (is.toi == bt && key < bkey)) { bt = is.toi; bkey = key; bhit = true; binst = first; bprim = 0; btf = best_f32(bt); }
                }
            }
            if (stop) { blocked = true; cur = kFin; continue; }
        }
        cur = st.pop();
    }
}

} // namespace nrays
