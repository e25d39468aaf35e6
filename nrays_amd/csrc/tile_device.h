// tile_device.h — device code shared by the frame kernels of the megakernel path (nrays_hip.hip: k_primary) and of the
// staged path (wavefront.hip: k_wf_primary): the background rows outside the window of blocks that can see the scene.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef NR_NT_STORES
#define NR_NT_STORES 1 // frame-buffer stores carry the non-temporal hint: the 25 MB of a 1080p frame do not sweep the scene out of the L2s
#endif

namespace nrays {

// One row of the compact frame buffer outside the window of blocks that can see the scene (k_primary): background sums
// (padding rows of the last band: zero) for the floats t0, t0 + tstep, ... of the row.  Out of line: its registers and
// uniforms stay out of the tile loop's allocation.
// INL: the copy inside the tile loop of the workgroup-list kernels is inlined — a call returns through `s_waitcnt vmcnt(0)`, i.e. waits
// for the row's stores to be acknowledged (1.3 us per quarter row, 6 us at the end of a frame: 20 us of the 45 us balls launch were
// workgroups finishing their rows one acknowledged call after the other).
__device__ __attribute__((always_inline)) inline void fill_background_row_body(float bg0, float bg1, float bg2, uint32_t spp, float* out, uint32_t width, uint32_t height,
                                                 uint32_t band_rows, uint32_t band_owner, uint32_t band_owners, uint32_t win_x0, uint32_t win_nx,
                                                 uint32_t win_y0, uint32_t win_ny, uint32_t lane_log2, uint32_t rl, uint32_t t0, uint32_t tstep) {
    const uint32_t bwl = lane_log2 ? (7u - lane_log2) >> 1 : 4u, bhl = lane_log2 ? (6u - lane_log2) >> 1 : 4u;
    const uint32_t wi0 = win_x0 << bwl, wi1 = (win_x0 + win_nx) << bwl, wr0 = win_y0 << bhl, wr1 = (win_y0 + win_ny) << bhl;
    float b0 = 0.0f, b1 = 0.0f, b2 = 0.0f;
    for (uint32_t s = 0; s < spp; ++s) { b0 = b0 + bg0; b1 = b1 + bg1; b2 = b2 + bg2; }
    uint32_t j = rl;
    if (band_rows != 0 && band_owners > 1) j = ((rl / band_rows) * band_owners + band_owner) * band_rows + (rl % band_rows);
    const bool real = j < height;
    const bool split = rl >= wr0 && rl < wr1 && win_nx != 0u; // this row crosses the window: skip its columns
    __attribute__((address_space(1))) float* row = (__attribute__((address_space(1))) float*)(out + (size_t)rl * width * 3);
    if (!real) { b0 = 0.0f; b1 = 0.0f; b2 = 0.0f; }
    if (((width * 3u) & 3u) == 0u && (((uintptr_t)out) & 15u) == 0u && (tstep % 3u) == 1u) {
        // 16-byte stores: chunk q holds the floats 4q .. 4q + 3, i.e. the channels (q mod 3), (q + 1) mod 3, ... — three patterns, and
        // q mod 3 advances by one per step because tstep = 1 (mod 3).  (The scalar loop below spent a division and a 4-byte store per
        // float: ~1.5 us per call, and the rows of a workgroup whose waves sit on long tiles were the tail of the balls frame.)
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v pat[3] = {f4v{b0, b1, b2, b0}, f4v{b1, b2, b0, b1}, f4v{b2, b0, b1, b2}};
        const uint32_t nq = width * 3u / 4u, f_lo = 3u * wi0, f_hi = 3u * wi1; // floats [f_lo, f_hi) belong to the window
        uint32_t ph = t0 % 3u;
        for (uint32_t q = t0; q < nq; q += tstep, ph = ph == 2u ? 0u : ph + 1u) {
            const uint32_t f = 4u * q;
            const f4v v = ph == 0u ? pat[0] : (ph == 1u ? pat[1] : pat[2]);
#if NR_NT_STORES
            if (!split || f + 4u <= f_lo || f >= f_hi) { __builtin_nontemporal_store(v, (__attribute__((address_space(1))) f4v*)(row + f)); continue; }
#else
            if (!split || f + 4u <= f_lo || f >= f_hi) { *(__attribute__((address_space(1))) f4v*)(row + f) = v; continue; }
#endif
            for (uint32_t k = 0; k < 4u; ++k) if (f + k < f_lo || f + k >= f_hi) row[f + k] = v[k]; // a chunk across the window's edge
        }
        return;
    }
    for (uint32_t f = t0; f < width * 3u; f += tstep) {
        const uint32_t i = f / 3u, c = f - i * 3u;
        if (split && i >= wi0 && i < wi1) continue;
        row[f] = c == 0u ? b0 : (c == 1u ? b1 : b2);
    }
}

__device__ __noinline__ void fill_background_row(float bg0, float bg1, float bg2, uint32_t spp, float* out, uint32_t width, uint32_t height,
                                                 uint32_t band_rows, uint32_t band_owner, uint32_t band_owners, uint32_t win_x0, uint32_t win_nx,
                                                 uint32_t win_y0, uint32_t win_ny, uint32_t lane_log2, uint32_t rl, uint32_t t0, uint32_t tstep) {
    fill_background_row_body(bg0, bg1, bg2, spp, out, width, height, band_rows, band_owner, band_owners, win_x0, win_nx, win_y0, win_ny, lane_log2, rl, t0, tstep);
}

} // namespace nrays
