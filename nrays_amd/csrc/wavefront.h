// wavefront.h — entry points of the staged ("wavefront") form of the trace loop (wavefront.hip), called by
// render_impl (nrays_hip.hip).  Host only.
#pragma once
#include <hip/hip_runtime.h>

#include "scene_handle.h"

namespace nrays {

// Whether a plain (not instrumented) frame of this handle is rendered by the staged path: scenes of TriMesh nodes only,
// no double branching, point-like lights (racsample 1), and NRAYS_WAVEFRONT / the library's rule say so.
bool wavefront_wanted(const NraysScene* sc, const NraysRenderParams* p, uint32_t lane_log2);

// Renders the frame described by `R` (window, cull bounds, raygen tables, lane mapping already decided by render_impl)
// into d_out: primary stage, then closest / shadow / shade stages generation after generation, every sample of a pixel
// in one pass over sub-ranges of the wave tiles.  Records [ev_pbegin .. ev_pend] around the first sub-range's stages when
// `timed`.  Pixels are bit-identical to the megakernel's.
int wavefront_render(NraysScene* sc, const NraysRenderParams* p, DRender R, float* d_out, hipStream_t stream, uint32_t tiles_x,
                     uint32_t tiles_y, bool timed, int slot, DeviceCounters* next_ctr, uint32_t* next_counts);

void wavefront_release(NraysScene* sc);

} // namespace nrays
