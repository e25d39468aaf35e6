// presplit_clip.h — the arithmetic of triangle pre-splitting (scene_build.cpp: presplit(); bvh_device.hip: k_presplit), shared by the
// host builder and the device builder so that both clip, box and rate a piece with the same f64 operations in the same order
// (f64 +, -, *, /, sqrt are correctly rounded on gfx950 and -ffp-contract=off holds on both sides: identical boxes).
#pragma once
#include <cmath>
#include <cstdint>

#include "bvh_build.h"

#ifndef NR_PRESPLIT_EMPTY
#define NR_PRESPLIT_EMPTY 0.5 // a piece qualifies when its box is at least this empty (1 - 2 area / half box area)
#endif

namespace nrays {

// A triangle clipped to a box is a convex polygon of at most 9 vertices (3 + one per box face).
constexpr int kClipMax = 12;
constexpr int kSplitDepthMax = 20;
struct ClipPoly { double v[kClipMax][3]; int n; };

NR_HD inline float clip_next_down(float f) { // nextafterf(f, -inf) for finite f
    uint32_t b; __builtin_memcpy(&b, &f, 4);
    if (f == 0.0f) b = 0x80000001u; else if (b & 0x80000000u) ++b; else --b;
    float r; __builtin_memcpy(&r, &b, 4); return r;
}
NR_HD inline float clip_next_up(float f) { // nextafterf(f, +inf) for finite f
    uint32_t b; __builtin_memcpy(&b, &f, 4);
    if (f == 0.0f) b = 0x00000001u; else if (b & 0x80000000u) --b; else ++b;
    float r; __builtin_memcpy(&r, &b, 4); return r;
}
NR_HD inline float clip_round_down(double v) { float f = (float)v; if ((double)f > v) f = clip_next_down(f); return f; } // largest f32 <= v
NR_HD inline float clip_round_up(double v) { float f = (float)v; if ((double)f < v) f = clip_next_up(f); return f; }     // smallest f32 >= v

// Sutherland-Hodgman against one axis plane; returns false (and an unusable polygon) if the vertex budget is exceeded.
NR_HD inline bool clip_half(const ClipPoly& in, int axis, double c, bool keep_low, ClipPoly& out) {
    out.n = 0;
    for (int k = 0; k < in.n; ++k) {
        const double* a = in.v[k];
        const double* b = in.v[(k + 1) % in.n];
        bool ia = keep_low ? a[axis] <= c : a[axis] >= c, ib = keep_low ? b[axis] <= c : b[axis] >= c;
        if (ia) {
            if (out.n >= kClipMax) return false;
            for (int d = 0; d < 3; ++d) out.v[out.n][d] = a[d];
            ++out.n;
        }
        if (ia != ib) {
            if (out.n >= kClipMax) return false;
            double t = (c - a[axis]) / (b[axis] - a[axis]);
            for (int d = 0; d < 3; ++d) out.v[out.n][d] = d == axis ? c : a[d] + (b[d] - a[d]) * t;
            ++out.n;
        }
    }
    return true;
}
NR_HD inline PrimBounds poly_box(const ClipPoly& p, const PrimBounds& within) {
    PrimBounds b;
    for (int a = 0; a < 3; ++a) {
        double lo = HUGE_VAL, hi = -HUGE_VAL;
        for (int k = 0; k < p.n; ++k) { lo = p.v[k][a] < lo ? p.v[k][a] : lo; hi = hi < p.v[k][a] ? p.v[k][a] : hi; }
        // outward f32 rounding + one ulp for the rounding of the clip itself; never larger than the box being split
        const float dn = clip_next_down(clip_round_down(lo)), up = clip_next_up(clip_round_up(hi));
        b.mn[a] = within.mn[a] < dn ? dn : within.mn[a];
        b.mx[a] = up < within.mx[a] ? up : within.mx[a];
    }
    return b;
}
NR_HD inline double box_half_area(const PrimBounds& b) {
    double dx = (double)b.mx[0] - b.mn[0], dy = (double)b.mx[1] - b.mn[1], dz = (double)b.mx[2] - b.mn[2];
    return dx * dy + dy * dz + dz * dx;
}
NR_HD inline double poly_area2(const ClipPoly& p) { // twice the area of a planar convex polygon (fan from vertex 0)
    double s = 0.0;
    for (int k = 1; k + 1 < p.n; ++k) {
        double e1[3], e2[3];
        for (int d = 0; d < 3; ++d) { e1[d] = p.v[k][d] - p.v[0][d]; e2[d] = p.v[k + 1][d] - p.v[0][d]; }
        double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
        s += sqrt(cx * cx + cy * cy + cz * cz);
    }
    return s;
}
NR_HD inline void tri_poly(const TriRec& r, ClipPoly& p) {
    p.n = 3;
    for (int d = 0; d < 3; ++d) { p.v[0][d] = r.v0[d]; p.v[1][d] = r.v1[d]; p.v[2][d] = r.v2[d]; }
}
// One midpoint split of a piece; false if the piece cannot be split (degenerate clip, vertex budget, empty box).
NR_HD inline bool split_piece(const ClipPoly& poly, const PrimBounds& box, ClipPoly& lo, ClipPoly& hi, PrimBounds& bl, PrimBounds& bh) {
    int axis = 0; float ext = box.mx[0] - box.mn[0];
    for (int a = 1; a < 3; ++a) if (box.mx[a] - box.mn[a] > ext) { ext = box.mx[a] - box.mn[a]; axis = a; }
    const double mid = 0.5 * ((double)box.mn[axis] + (double)box.mx[axis]);
    if (!clip_half(poly, axis, mid, true, lo) || !clip_half(poly, axis, mid, false, hi)) return false;
    if (lo.n < 3 || hi.n < 3) return false; // the plane misses the piece (degenerate): leave it alone
    bl = poly_box(lo, box); bh = poly_box(hi, box);
    for (int a = 0; a < 3; ++a) if (!(bl.mn[a] <= bl.mx[a]) || !(bh.mn[a] <= bh.mx[a])) return false;
    return true;
}
// The rule both builders apply to a piece: it is split while its empty box area exceeds `thr` and half of its box.
NR_HD inline bool piece_qualifies(double half_area, double gain, double thr) { return gain > thr && gain > NR_PRESPLIT_EMPTY * half_area; }

} // namespace nrays
