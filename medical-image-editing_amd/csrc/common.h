// Shared helpers for the VQ-W-Net HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#define VQW_OK 0
#define VQW_ERR_ARG (-1)
#define VQW_ERR_HIP (-2)

// internal helper: not part of the C ABI (hidden visibility keeps it out of the dynamic symbol table)
extern "C" __attribute__((visibility("hidden"))) void vqw_set_error(const char* fmt, ...);

#define VQW_CHECK(cond, ...)                                                   \
    do {                                                                       \
        if (!(cond)) {                                                         \
            vqw_set_error(__VA_ARGS__);                                        \
            return VQW_ERR_ARG;                                                \
        }                                                                      \
    } while (0)

#define VQW_LAUNCH_CHECK(name)                                                 \
    do {                                                                       \
        hipError_t e_ = hipGetLastError();                                     \
        if (e_ != hipSuccess) {                                                \
            vqw_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return VQW_ERR_HIP;                                                \
        }                                                                      \
    } while (0)

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

// Memory-bound grid sizing rule: cap at ~8 blocks per CU and grid-stride the rest.
static inline int stream_grid(long work_items, int block) {
    long g = (work_items + block - 1) / block;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// wave64 butterfly sum
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {       // fixed butterfly: the same bits every run
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
