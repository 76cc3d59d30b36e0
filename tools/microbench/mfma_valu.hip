// What does a vector instruction cost beside fp32 MFMAs?  v_mfma_f32_16x16x4_f32 / 32x32x2_f32 loops with NF filler
// instructions of one kind behind every MFMA, one or two waves per SIMD.  Prints the time per MFMA in pipe cycles' worth
// (relative to the bare loop) so that "hidden" (ratio 1.0) and "additive" (ratio 1 + NF * cost / 32) can be told apart.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_valu.hip -o tools/microbench/mfma_valu.bin && tools/microbench/mfma_valu.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { F_NONE = 0, F_ADDF, F_ADDI, F_MOV, F_DSR64, F_FMA, F_CNDMASK, F_DSW64, F_PKADD, F_PKFMA, F_SADD, F_SNOP, F_BUFLD, F_ADDF_S, F_DSR128, F_DPP };

template <int KIND>
__device__ __forceinline__ void filler(float& x, float& y, int& i, const float* lds, int lane) {
    if (KIND == F_ADDF) asm volatile("v_add_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y));
    if (KIND == F_FMA) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(y));
    if (KIND == F_ADDI) asm volatile("v_add_u32 %0, %1, %2" : "=v"(i) : "v"(i), "v"(lane));
    if (KIND == F_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(y));
    if (KIND == F_CNDMASK) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x) : "v"(x), "v"(y));
    if (KIND == F_PKADD) { f32x2 p = {x, y}; asm volatile("v_pk_add_f32 %0, %1, %1" : "=v"(p) : "v"(p)); x = p.x; }
    if (KIND == F_PKFMA) { f32x2 p = {x, y}; asm volatile("v_pk_fma_f32 %0, %1, %1, %1" : "=v"(p) : "v"(p)); x = p.x; }
    if (KIND == F_SADD) asm volatile("s_add_u32 s20, s20, 3" ::: "s20", "scc");
    if (KIND == F_SNOP) asm volatile("s_nop 0");
    if (KIND == F_BUFLD) { float t; asm volatile("global_load_dword %0, %1, off" : "=v"(t) : "v"(lds));  /* lds here = a global pointer, see the call */ }
    if (KIND == F_ADDF_S) asm volatile("v_add_f32 %0, s20, %1" : "=v"(x) : "v"(y) : "s20");
    if (KIND == F_DPP) asm volatile("v_add_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(x) : "v"(x), "v"(y));
    if (KIND == F_DSR128) {
        f32x4 t;
        asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(lane * 16));
    }
    if (KIND == F_DSR64) {
        f32x2 t;
        asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(lane * 8));
        // the value is never waited for inside the loop (a final s_waitcnt at the end)
    }
    if (KIND == F_DSW64) {
        f32x2 t = {x, y};
        asm volatile("ds_write_b64 %0, %1" : : "v"(lane * 8), "v"(t));
    }
}

template <int SHAPE, int KIND, int NF>
__global__ void __launch_bounds__(256) k_loop(const float* __restrict__ in, float* __restrict__ out, int iters) {
    __shared__ float lds[4096];
    const int t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    lds[threadIdx.x] = in[t & 0xffff];
    __syncthreads();
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(t * 8 + i) & 0xffff]; b[i] = in[(t * 8 + i + 4096) & 0xffff]; }
    float x = a[0], y = b[1];
    int iv = t;
    float s = 0.f;
    if (SHAPE == 16) {
        f32x4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i + j) & 7], b[j], c[j], 0, 0, 0);
#pragma unroll
                    for (int f = 0; f < NF; ++f) filler<KIND>(x, y, iv, KIND == F_BUFLD ? in + lane : lds, lane);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    } else {
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], c0, 0, 0, 0);
#pragma unroll
                for (int f = 0; f < 2 * NF; ++f) filler<KIND>(x, y, iv, KIND == F_BUFLD ? in + lane : lds, lane);
                __builtin_amdgcn_sched_barrier(0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + 1) & 7], c1, 0, 0, 0);
#pragma unroll
                for (int f = 0; f < 2 * NF; ++f) filler<KIND>(x, y, iv, KIND == F_BUFLD ? in + lane : lds, lane);
                __builtin_amdgcn_sched_barrier(0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + 1) & 7], b[i], c2, 0, 0, 0);
#pragma unroll
                for (int f = 0; f < 2 * NF; ++f) filler<KIND>(x, y, iv, KIND == F_BUFLD ? in + lane : lds, lane);
                __builtin_amdgcn_sched_barrier(0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + 2) & 7], b[(i + 3) & 7], c3, 0, 0, 0);
#pragma unroll
                for (int f = 0; f < 2 * NF; ++f) filler<KIND>(x, y, iv, KIND == F_BUFLD ? in + lane : lds, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
    out[t] = s + x + (float)iv;
}

static double g_base[2][2];

template <int SHAPE, int KIND, int NF>
static void run(const float* din, float* dout, int wps, const char* tag) {
    const int blocks = 256 * wps, iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) k_loop<SHAPE, KIND, NF><<<blocks, 256>>>(din, dout, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) k_loop<SHAPE, KIND, NF><<<blocks, 256>>>(din, dout, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double nm = (double)iters * (SHAPE == 16 ? 64 : 32);           // MFMAs per wave
    const double flops = (double)blocks * 4 * nm * (SHAPE == 16 ? 2048.0 : 4096.0);
    double& base = g_base[SHAPE == 16 ? 0 : 1][wps - 1];
    if (KIND == F_NONE) base = ms;
    // extra time per filler, in units of the bare loop's time per 32 pipe cycles (one 16x16x4 MFMA), times 32 -> "cycles"
    const double per_pipe32 = base / (nm * wps * (SHAPE == 16 ? 1 : 2));
    const double extra = NF ? (ms - base) / (nm * wps * (SHAPE == 16 ? 1 : 2) * NF) / per_pipe32 * 32.0 : 0.0;
    printf("%-10s %-8s NF=%d  %d wave/SIMD  %.3f ms  %.1f TFLOP/s  ratio %.3f  extra pipe-cycles per filler %.2f\n",
           SHAPE == 16 ? "16x16x4" : "32x32x2", tag, NF, wps, ms, flops / ms / 1e9, ms / base, extra);
}

template <int SHAPE>
static void sweep(const float* din, float* dout) {
    for (int wps = 1; wps <= 2; ++wps) {
        run<SHAPE, F_NONE, 0>(din, dout, wps, "bare");
        run<SHAPE, F_ADDF, 1>(din, dout, wps, "v_add_f32");
        run<SHAPE, F_ADDF, 2>(din, dout, wps, "v_add_f32");
        run<SHAPE, F_ADDF, 4>(din, dout, wps, "v_add_f32");
        run<SHAPE, F_FMA, 2>(din, dout, wps, "v_fma_f32");
        run<SHAPE, F_ADDI, 1>(din, dout, wps, "v_add_u32");
        run<SHAPE, F_ADDI, 2>(din, dout, wps, "v_add_u32");
        run<SHAPE, F_ADDI, 4>(din, dout, wps, "v_add_u32");
        run<SHAPE, F_MOV, 2>(din, dout, wps, "v_mov");
        run<SHAPE, F_CNDMASK, 2>(din, dout, wps, "v_cndmask");
        run<SHAPE, F_DSR64, 1>(din, dout, wps, "ds_rd_b64");
        run<SHAPE, F_DSR64, 2>(din, dout, wps, "ds_rd_b64");
        run<SHAPE, F_DSW64, 1>(din, dout, wps, "ds_wr_b64");
        run<SHAPE, F_DSR128, 1>(din, dout, wps, "ds_rd_b128");
        run<SHAPE, F_PKADD, 1>(din, dout, wps, "v_pk_add");
        run<SHAPE, F_PKADD, 2>(din, dout, wps, "v_pk_add");
        run<SHAPE, F_PKFMA, 2>(din, dout, wps, "v_pk_fma");
        run<SHAPE, F_SADD, 2>(din, dout, wps, "s_add");
        run<SHAPE, F_SADD, 4>(din, dout, wps, "s_add");
        run<SHAPE, F_SNOP, 2>(din, dout, wps, "s_nop");
        run<SHAPE, F_BUFLD, 1>(din, dout, wps, "glb_load");
        run<SHAPE, F_ADDF_S, 2>(din, dout, wps, "v_add_sgpr");
        run<SHAPE, F_DPP, 2>(din, dout, wps, "v_add_dpp");
    }
}

int main() {
    std::vector<float> h(65536);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    float *din, *dout;
    hipMalloc(&din, h.size() * 4);
    hipMalloc(&dout, 4 * 256 * 4096);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    sweep<16>(din, dout);
    sweep<32>(din, dout);
    return 0;
}
