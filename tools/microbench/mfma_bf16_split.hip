// Rate side of the split-precision experiment (tools/split_precision.py is the accuracy side): what does the bf16 matrix core
// deliver for an fp32-class product built from six bf16 products, with the 3-way split of the activation operand done in the
// loop by the VALU?
//   v_mfma_f32_32x32x16_bf16: 32768 FLOP per instruction.  Per K = 16 step and per 32-row operand fragment the wave splits 8 fp32
//   values per lane into 3 x 8 bf16 (NV VALU instructions, emulated here by that many dependent fp32 fmas / converts) and issues
//   6 NB MFMAs (NB = 32-column blocks that share the fragment; the weights arrive pre-split).
// Prints achieved TFLOP/s of bf16 MFMA and the fp32-class rate = bf16 rate / 6, for NB = 1, 2, 4, with and without the split.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_bf16_split.hip -o tools/microbench/mfma_bf16_split.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// 3-way split of 8 fp32 values into bf16 pieces (round-to-nearest-even through the hardware convert)
__device__ __forceinline__ void split8(const float* x, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];
        const float r1 = x[i] - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        p0[i] = h; p1[i] = m; p2[i] = (__bf16)r2;
    }
}

template <int NB, int SPLIT, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_run(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = in[(t * 8 + i) & 0xffff];
    bf16x8 w[3];
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 8; ++i) w[j][i] = (__bf16)in[(t + 64 * j + i) & 0xffff];
    f32x16 c[NB];
    for (int n = 0; n < NB; ++n)
        for (int i = 0; i < 16; ++i) c[n][i] = 0.f;
    bf16x8 a0, a1, a2;
    split8(x, a0, a1, a2);
    for (int it = 0; it < iters; ++it) {
        if (SPLIT) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = x[i] * 1.0001f + c[0][i];      // a fresh fragment every step (keeps the split live)
            split8(x, a0, a1, a2);
        }
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            c[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, w[0], c[n], 0, 0, 0);
            c[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, w[1], c[n], 0, 0, 0);
            c[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, w[2], c[n], 0, 0, 0);
            c[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, w[0], c[n], 0, 0, 0);
            c[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, w[1], c[n], 0, 0, 0);
            c[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, w[0], c[n], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int n = 0; n < NB; ++n)
        for (int i = 0; i < 16; ++i) s += c[n][i];
    if (s == 123.456f) out[t] = s;
}

template <int NB, int SPLIT, int WAVES>
static void run(const float* in, float* out) {
    const int iters = 4000, blocks = 256 * 4 * 2 / WAVES * (WAVES == 8 ? 1 : 1);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        k_run<NB, SPLIT, WAVES><<<256 * (8 / WAVES), 64 * WAVES>>>(in, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    (void)blocks;
    const double flop = 256.0 * 8 * iters * NB * 6 * 32768.0;       // 8 waves per CU
    printf("  NB %d  split in loop %d  waves/WG %d: %.3f ms  bf16 MFMA %.0f TFLOP/s  fp32-class (six products) %.0f TFLOP/s\n", NB, SPLIT, WAVES,
           best, flop / best / 1e9, flop / 6 / best / 1e9);
}

int main() {
    float *in, *out;
    hipMalloc(&in, 1 << 20); hipMalloc(&out, 1 << 24);
    hipMemset(in, 0, 1 << 20);
    printf("256 CUs x 8 waves (2 per SIMD), v_mfma_f32_32x32x16_bf16, six products per K step and N block:\n");
    run<1, 0, 4>(in, out); run<2, 0, 4>(in, out); run<4, 0, 4>(in, out);
    run<1, 1, 4>(in, out); run<2, 1, 4>(in, out); run<4, 1, 4>(in, out);
    return 0;
}
