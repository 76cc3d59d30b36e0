// Optional per-launch timing (bench.py roofline): HIP events recorded on the launch stream right around a kernel family.
// Off unless vqw_profile_begin() was called (conv.hip holds the state).  Families: 0 = MFMA fwd/dgrad, 1 = MFMA wgrad (incl.
// slab reduce), 2 = generic fwd, 3 = generic wgrad, 4 = Winograd-form fwd/dgrad/wgrad (FLOPs = the 4/9 the matrix cores
// execute), 5 = HBM-bound normalisation / element-wise kernels (bytes = tensor passes x 4 B x elements, as launched).
#pragma once
#include "common.h"

#define PROF_FAMILIES 6
#define PROF_FAMILY_HBM 5
int vqw_prof_open(int family, double flops, double bytes, hipStream_t st);      // -> record index or -1
void vqw_prof_close(int idx, hipStream_t st);

struct ProfScope {
    int idx;
    hipStream_t st;
    ProfScope(int family, double flops, hipStream_t s, double bytes = 0.0) : idx(vqw_prof_open(family, flops, bytes, s)), st(s) {}
    ~ProfScope() {
        if (idx >= 0) vqw_prof_close(idx, st);
    }
};
// an HBM-bound launch over `passes` tensor passes of `elems` fp32 elements
#define VQW_PROF_HBM(stream, passes, elems) ProfScope prof_scope_(PROF_FAMILY_HBM, 0.0, (hipStream_t)(stream), 4.0 * (double)(passes) * (double)(elems))
