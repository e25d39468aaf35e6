"""presplit_clip.h is compiled into BOTH builders (host: scene_build.cpp, device: bvh_device.hip) so that they clip, box and rate a piece
of a triangle with the same arithmetic.  Its bit-twiddled nextafter / directed roundings replace the libm calls the host builder used
before the code was shared: this test compiles the header with g++ and compares them with <cmath> over random and special values
(reference side: the boxes BVT::new_balanced receives, examples/loader3d.rs:695 — here only their outward f32 rounding is at stake)."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROGRAM = r'''
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <random>
#include "presplit_clip.h"
using namespace nrays;
static float down_ref(double v) { float f = (float)v; if ((double)f > v) f = std::nextafterf(f, -std::numeric_limits<float>::infinity()); return f; }
static float up_ref(double v) { float f = (float)v; if ((double)f < v) f = std::nextafterf(f, std::numeric_limits<float>::infinity()); return f; }
static bool same(float a, float b) { return std::memcmp(&a, &b, 4) == 0; }
int main() {
    std::mt19937_64 rng(12345);
    long bad = 0, n = 0;
    auto check_f = [&](float f) {
        if (!std::isfinite(f)) return;
        ++n;
        if (!same(clip_next_down(f), std::nextafterf(f, -std::numeric_limits<float>::infinity()))) ++bad;
        if (!same(clip_next_up(f), std::nextafterf(f, std::numeric_limits<float>::infinity()))) ++bad;
    };
    auto check_d = [&](double v) { ++n; if (!same(clip_round_down(v), down_ref(v))) ++bad; if (!same(clip_round_up(v), up_ref(v))) ++bad; };
    const float specials[] = {0.0f, -0.0f, 1.0f, -1.0f, std::numeric_limits<float>::min(), -std::numeric_limits<float>::min(), std::numeric_limits<float>::denorm_min(),
                              -std::numeric_limits<float>::denorm_min(), std::numeric_limits<float>::max(), -std::numeric_limits<float>::max(), 1.17549421e-38f, 16777216.0f};
    for (float f : specials) { check_f(f); check_d((double)f); check_d((double)f * (1.0 + 1e-12)); check_d((double)f * (1.0 - 1e-12)); }
    for (int i = 0; i < 2000000; ++i) {
        uint32_t b = (uint32_t)rng(); float f; std::memcpy(&f, &b, 4); check_f(f);
        uint64_t q = rng(); double d; std::memcpy(&d, &q, 8);
        if (std::isfinite(d) && std::fabs(d) < 3e38 && std::fabs(d) > 1e-44) check_d(d);
        check_d(std::ldexp((double)(int64_t)(rng() >> 11) / 9007199254740992.0 - 0.5, (int)(rng() % 60) - 30));
    }
    // a clip: the two halves of a triangle cover it and their boxes stay inside the box that was split
    TriRec r; float v[9] = {0.f, 0.f, 0.f, 4.f, 1.f, 0.25f, 1.f, 3.f, 2.f};
    std::memcpy(r.v0, v, 12); std::memcpy(r.v1, v + 3, 12); std::memcpy(r.v2, v + 6, 12);
    ClipPoly p, lo, hi; tri_poly(r, p);
    PrimBounds box{{0.f, 0.f, 0.f}, {4.f, 3.f, 2.f}}, bl, bh;
    if (!split_piece(p, box, lo, hi, bl, bh)) ++bad;
    for (int a = 0; a < 3; ++a) if (bl.mn[a] < box.mn[a] || bl.mx[a] > box.mx[a] || bh.mn[a] < box.mn[a] || bh.mx[a] > box.mx[a]) ++bad;
    if (!(bl.mx[0] >= 2.f && bh.mn[0] <= 2.f)) ++bad; // the longest axis is x, split at 2
    if (std::fabs(poly_area2(lo) + poly_area2(hi) - poly_area2(p)) > 1e-12 * poly_area2(p)) ++bad;
    std::printf("%ld checks, %ld mismatches\n", n, bad);
    return bad != 0;
}
'''


def test_shared_clip_arithmetic_matches_libm():
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.cpp"), os.path.join(d, "t")
        open(src, "w").write(PROGRAM)
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "nrays_amd", "csrc"), "-o", exe, src])
        out = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
        assert out.returncode == 0, out.stdout
        assert " 0 mismatches" in out.stdout, out.stdout
