// What an LDS-fed fp32 MFMA loop sustains (the conv / VQ kernels' inner loop): A fragments come from LDS by ds_read_b128
// (one read feeds four 32x32x2 MFMAs), B operands sit in registers.  Variants: how far the reads run ahead of the MFMAs,
// one or two waves per SIMD, s_setprio around the MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_lds.hip -o mfma_lds.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// AHEAD: number of float4 fragment reads in flight before the MFMAs that use them (1 or 2 or 4); PRIO: s_setprio 1 around MFMAs
template <int AHEAD, int NACC>
__global__ void __launch_bounds__(256, 2) k_loop(const float* __restrict__ in, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // 32 rows x 260 floats
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, h = lane >> 5;
    for (int i = tid; i < 32 * 260; i += 256) smem[i] = in[i & 0xffff];
    float xr[128];
#pragma unroll
    for (int i = 0; i < 128; ++i) xr[i] = in[(tid * 128 + i) & 0xffff];
    __syncthreads();
    const float* arow = smem + n * 260 + 4 * h;
    f32x16 acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = f32x16{0};
    for (int it = 0; it < iters; ++it) {
        float4 a[AHEAD + 1];
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) a[k] = *(const float4*)(arow + 8 * k);
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            if (j + AHEAD < 32) a[(j + AHEAD) % (AHEAD + 1)] = *(const float4*)(arow + 8 * (j + AHEAD));
            __builtin_amdgcn_sched_barrier(0);
            const float4 f = a[j % (AHEAD + 1)];
            acc[j % NACC] = MFMA32(f.x, xr[4 * j], acc[j % NACC]);
            acc[j % NACC] = MFMA32(f.y, xr[4 * j + 1], acc[j % NACC]);
            acc[j % NACC] = MFMA32(f.z, xr[4 * j + 2], acc[j % NACC]);
            acc[j % NACC] = MFMA32(f.w, xr[4 * j + 3], acc[j % NACC]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NACC; ++k)
        for (int i = 0; i < 16; ++i) s += acc[k][i];
    out[blockIdx.x * 256 + tid] = s;
}

template <int AHEAD, int NACC>
static void run(const float* din, float* dout, int blocks, int iters, const char* tag) {
    hipFuncSetAttribute((const void*)k_loop<AHEAD, NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, 66816);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) k_loop<AHEAD, NACC><<<blocks, 256, 66816>>>(din, dout, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) k_loop<AHEAD, NACC><<<blocks, 256, 66816>>>(din, dout, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flops = (double)blocks * 4 * iters * 128 * 4096.0;
    printf("%-52s blocks %5d  %.3f ms  %.1f TFLOP/s\n", tag, blocks, ms, flops / ms / 1e9);
}

int main() {
    std::vector<float> h(65536);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    float *din, *dout;
    hipMalloc(&din, h.size() * 4);
    hipMalloc(&dout, 4 * 256 * 4096);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int iters = 2000;
    run<1, 1>(din, dout, 256, iters, "reads 1 group ahead, 1 chain, 1 WG/CU");
    run<1, 1>(din, dout, 512, iters, "reads 1 group ahead, 1 chain, 2 WG/CU");
    run<2, 1>(din, dout, 512, iters, "reads 2 groups ahead, 1 chain, 2 WG/CU");
    run<4, 1>(din, dout, 512, iters, "reads 4 groups ahead, 1 chain, 2 WG/CU");
    run<1, 2>(din, dout, 512, iters, "reads 1 group ahead, 2 chains, 2 WG/CU");
    run<2, 2>(din, dout, 512, iters, "reads 2 groups ahead, 2 chains, 2 WG/CU");
    run<2, 2>(din, dout, 256, iters, "reads 2 groups ahead, 2 chains, 1 WG/CU");
    run<4, 4>(din, dout, 512, iters, "reads 4 groups ahead, 4 chains, 2 WG/CU");
    return 0;
}
