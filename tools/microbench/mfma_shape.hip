// Sustained fp32 MFMA rate by instruction shape (the chip lowers its clock under load; MI355X_MICROARCH.md, DVFS (7)):
// v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32, operands in registers, random data, one or two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_shape.hip -o /tmp/mfma_shape && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ void __launch_bounds__(256) k_mfma(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(t * 8 + i) & 0xffff]; b[i] = in[(t * 8 + i + 4096) & 0xffff]; }
    if (SHAPE == 32) {
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + 1) & 7], c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + 1) & 7], b[i], c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + 2) & 7], b[(i + 3) & 7], c3, 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
        out[t] = s;
    } else if (SHAPE == 321 || SHAPE == 322) {      // one (321) or two (322) dependent accumulation chains per wave
        f32x16 c0 = {0}, c1 = {0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                if (SHAPE == 321 || (i & 1) == 0) c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i & 7], b[(i + 1) & 7], c0, 0, 0, 0);
                else c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i & 7], b[(i + 1) & 7], c1, 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
        out[t] = s;
    } else {
        f32x4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {       // 8 x 16x16x4 = 2 x the FLOPs of... (16*16*4*2 = 2048 each): 64 per iteration to match 32 x 4096
#pragma unroll
                for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i + j) & 7], b[j], c[j], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
        out[t] = s;
    }
}

template <int SHAPE>
static void run(const float* din, float* dout, int blocks, int iters, const char* tag) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) k_mfma<SHAPE><<<blocks, 256>>>(din, dout, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) k_mfma<SHAPE><<<blocks, 256>>>(din, dout, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flops = (double)blocks * 4 /*waves*/ * iters * (SHAPE == 16 ? 64 * 2048.0 : 32 * 4096.0);
    printf("%-34s blocks %5d  %.3f ms  %.1f TFLOP/s\n", tag, blocks, ms, flops / ms / 1e9);
}

int main() {
    std::vector<float> h(65536);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    float *din, *dout;
    hipMalloc(&din, h.size() * 4);
    hipMalloc(&dout, 4 * 256 * 4096);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        run<32>(din, dout, 256, iters, "32x32x2 f32, 1 wave/SIMD");
        run<16>(din, dout, 256, iters, "16x16x4 f32, 1 wave/SIMD");
        run<32>(din, dout, 512, iters, "32x32x2 f32, 2 waves/SIMD");
        run<321>(din, dout, 256, iters, "32x32x2 ONE chain, 1 wave/SIMD");
        run<321>(din, dout, 512, iters, "32x32x2 ONE chain, 2 waves/SIMD");
        run<322>(din, dout, 256, iters, "32x32x2 TWO chains, 1 wave/SIMD");
        run<322>(din, dout, 512, iters, "32x32x2 TWO chains, 2 waves/SIMD");
        run<16>(din, dout, 512, iters, "16x16x4 f32, 2 waves/SIMD");
    }
    return 0;
}
