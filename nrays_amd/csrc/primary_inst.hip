// primary_inst.hip — the k_primary permutations of ONE group (primary_kernel.h: NR_PRIMARY_PERMUTATIONS), compiled once per
// group with -DNR_PRIMARY_GROUP=g (__graft_entry__.py: build_hip): the translation units compile side by side.
#include "primary_kernel.h"

#ifndef NR_PRIMARY_GROUP
#error "compile with -DNR_PRIMARY_GROUP=<0..6>"
#endif

namespace nrays {

#ifdef NR_ONLY
constexpr int kOnly[] = {NR_ONLY};
constexpr bool wanted(int feat) { // the FEAT codes of a tuning build (the two full kernels are always there)
    if (feat == kFeatAll) return true;
    for (int f : kOnly) if (f == feat) return true;
    return false;
}
#else
constexpr bool wanted(int) { return true; }
#endif

template <int GROUP, bool STATS, int FEAT, bool PLAIN, int OCC>
static bool launch_if(const PrimaryLaunch& a, bool stats, int feat, bool plain, int occ) {
    if constexpr (GROUP == NR_PRIMARY_GROUP && wanted(FEAT)) {
        if (stats != STATS || feat != FEAT || plain != PLAIN || occ != OCC) return false;
        hipLaunchKernelGGL((k_primary<STATS, FEAT, PLAIN, OCC>), dim3(a.grid), dim3(kBlock), 0, a.stream, *a.d, *a.R, *a.qo, a.out, a.ctr, a.spill,
                           a.tiles_x, a.tiles_y, a.work, a.grab, a.zero_counts, a.zero_ctr);
        return true;
    } else {
        return false;
    }
}

#define NR_CAT_(a, b) a##b
#define NR_CAT(a, b) NR_CAT_(a, b)
bool NR_CAT(launch_primary_group, NR_PRIMARY_GROUP)(const PrimaryLaunch& a, bool stats, int feat, bool plain, int occ) {
#define X(G, S, F, P, O) if (launch_if<G, S, F, P, O>(a, stats, feat, plain, occ)) return true;
    NR_PRIMARY_PERMUTATIONS(X)
#undef X
    return false;
}

} // namespace nrays
