// Device helpers shared by the MFMA convolution kernels (conv_mfma.hip, conv_halo.hip).
#pragma once
#include "common.h"

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Bijective XCD-aware remap: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a contiguous run
// of tiles (neighbouring pixel tiles share halo rows and all N tiles of a pixel tile share the A operand in L2).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7;
    int xcd = bid & 7, idx = bid >> 3;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Raw buffer loads: 32-bit byte offsets against a wave-uniform descriptor; an offset >= num_records returns 0 in
// hardware, which is the conv zero padding / tile masking for free (no exec-mask branches, no 64-bit address math).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, unsigned nbytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    float4 f;     // (index, then convert: __builtin_bit_cast on a vector element mis-reads element 0 with this clang)
    unsigned a = v[0], b = v[1], c = v[2], d = v[3];
    f.x = __uint_as_float(a); f.y = __uint_as_float(b); f.z = __uint_as_float(c); f.w = __uint_as_float(d);
    return f;
}

__device__ __forceinline__ void buf_st4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float4 f) {
    u32x4 v;
    v[0] = __float_as_uint(f.x); v[1] = __float_as_uint(f.y); v[2] = __float_as_uint(f.z); v[3] = __float_as_uint(f.w);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)byte_off, 0, 0);
}

// A select between two values that are both evaluated.  Written as `c ? f(x) : k` the address arithmetic of f lands in
// an exec-masked block of its own; basic-block boundaries inside an MFMA loop keep the scheduler from spreading the
// prefetch / store instructions between the MFMAs (they end up in one run during which the matrix pipe idles).
__device__ __forceinline__ unsigned sel_u32(bool c, unsigned a, unsigned b) { return c ? a : b; }

// Per-tile statistics the conv epilogues leave for the following normalisation: (sum, M2) with M2 = sum (x - tile mean)^2,
// merged pairwise (Chan et al.) so that no fp32 quantity ever holds sum(x^2): var = E[x^2] - mean^2 from fp32 sums loses
// (mean / sigma)^2 * 6e-8 - 3e-4 of the output on a plane that sits 60 sigma off zero (profiles/r02_stats_precision.txt).
// lane_stats: m values held by one lane.  stat_merge_eq: two groups of n elements EACH (inv2n = 1 / (2 n)).
// stat_merge: groups of na and nb elements.
template <int M>
__device__ __forceinline__ void lane_stats(const float* v, float& s, float& m2) {
    s = 0.f;
#pragma unroll
    for (int i = 0; i < M; ++i) s += v[i];
    const float mean = s * (1.f / M);
    m2 = 0.f;
#pragma unroll
    for (int i = 0; i < M; ++i) { const float d = v[i] - mean; m2 = fmaf(d, d, m2); }
}
__device__ __forceinline__ void stat_merge_eq(float& s, float& m2, float so, float m2o, float inv2n) {
    const float d = so - s;
    m2 = (m2 + m2o) + d * d * inv2n;
    s += so;
}
__device__ __forceinline__ void stat_merge(float& s, float& m2, float na, float so, float m2o, float nb) {
    const float d = so / nb - s / na;
    m2 = (m2 + m2o) + d * d * (na * nb / (na + nb));
    s += so;
}
