// What do a region epilogue's global stores cost beside fp32 MFMAs, by store pattern and by placement?
// 256 workgroups x 512 threads (one per CU, as the Winograd kernels), each wave: R regions of NM MFMAs (16x16x4 fp32) and
// 8 KB of output per wave and region (64 KB per workgroup: a 64-tile x 64-cout region).  Patterns (64 lanes, pixel stride PS):
//   A  dword   : 16 lanes = 16 consecutive floats (64 B) of one pixel, 4 pixels per instruction       (32 instr / wave-region)
//   B  dwordx4 : 4 lanes = 64 B of one pixel, 16 pixels per instruction                               ( 8 instr)
//   C  dwordx4 : 16 lanes = 256 B of one pixel, 4 pixels per instruction                              ( 8 instr)
//   D  dwordx4 : 8 lanes = 128 B of one pixel, 8 pixels per instruction                               ( 8 instr)
//   E  dwordx4 : 1 KB contiguous                                                                      ( 8 instr)
// Placement: burst = all stores of a region behind its last MFMA and a barrier; spread = one store every NM / n MFMAs of
// the NEXT region.  Prints ms and the cost over the no-store run per pattern.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/store_patterns.hip -o tools/microbench/store_patterns.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, unsigned nbytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
}

// byte offset of store k (of NS) of a wave inside the workgroup-region's 64 KB block laid out [256 pixels][PS bytes]
template <int PAT>
__device__ __forceinline__ unsigned st_off(int wave, int lane, int k, int PS) {
    // the wave owns 32 pixels x 64 couts?  Keep it simple: the wave's 8 KB = pixels [32 wave, 32 wave + 32) x 256 B when PS = 256
    const int m = lane & 15, q = lane >> 4;
    if (PAT == 0) {            // A: k = 0..31: pixel = 4 (k >> 2) + q ... (k & 3) selects the 64 B quarter of the pixel's 256 B
        const int px = 32 * wave + 4 * (k >> 2) + q;
        return (unsigned)px * PS + (k & 3) * 64 + m * 4;
    }
    if (PAT == 1) {            // B: k = 0..7: 16 pixels x 64 B
        const int px = 32 * wave + 16 * (k >> 2) + m;
        return (unsigned)px * PS + (k & 3) * 64 + q * 16;
    }
    if (PAT == 2) {            // C: 4 pixels x 256 B
        const int px = 32 * wave + 4 * k + q;
        return (unsigned)px * PS + m * 16;
    }
    if (PAT == 3) {            // D: 8 pixels x 128 B
        const int px = 32 * wave + 8 * (k >> 1) + (lane >> 3);
        return (unsigned)px * PS + (k & 1) * 128 + (lane & 7) * 16;
    }
    return (unsigned)(32 * wave) * PS + k * 1024 + lane * 16;      // E (PS = 256: contiguous 8 KB)
}

template <int PAT, int MODE>   // MODE 0: no stores, 1: burst, 2: spread
__global__ void __launch_bounds__(512, 1) k_run(const float* __restrict__ in, float* __restrict__ out, unsigned nbytes, int R, int NM, int PS) {
    extern __shared__ float lds[];
    constexpr int NS = PAT == 0 ? 32 : 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    lds[tid] = in[tid];
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(out, nbytes);
    f32x4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = in[tid & 255], b = in[(tid + 64) & 255];
    unsigned off[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) off[k] = st_off<PAT>(wave, lane, k, PS);
    const int every = NM / NS;
    for (int r = 0; r < R; ++r) {
        const unsigned base = (unsigned)(blockIdx.x * R + r) * (256u * PS);          // region block
        const unsigned pbase = base - 256u * PS;                                      // previous region (spread)
        if (MODE == 2) {
            // NS groups of `every` MFMAs, one store of the previous region behind the first MFMA of each group
#pragma unroll
            for (int kk = 0; kk < NS; ++kk) {
                for (int it = 0; it < every; it += 8) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
                    if (it == 0 && r > 0) {
                        if (PAT == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(c[(kk + 3) & 7][0]), rs, off[kk], pbase, 0);
                        else { u32x4 v = {__float_as_uint(c[(kk + 3) & 7][0]), __float_as_uint(c[(kk + 3) & 7][1]), __float_as_uint(c[(kk + 3) & 7][2]), __float_as_uint(c[(kk + 3) & 7][3])};
                               __builtin_amdgcn_raw_buffer_store_b128(v, rs, off[kk], pbase, 0); }
                    }
                }
            }
        } else {
            for (int it = 0; it < NM; it += 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
            }
        }
        if (MODE == 1) {
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < NS; ++kk) {
                if (PAT == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(c[kk & 7][kk >> 3 & 3]), rs, off[kk], base, 0);
                else { u32x4 v = {__float_as_uint(c[kk][0]), __float_as_uint(c[kk][1]), __float_as_uint(c[kk][2]), __float_as_uint(c[kk][3])};
                       __builtin_amdgcn_raw_buffer_store_b128(v, rs, off[kk], base, 0); }
            }
            __syncthreads();
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    if (s == 123.456f) out[tid] = s;
}

template <int PAT, int MODE>
static float run(const float* in, float* out, unsigned nbytes, int R, int NM, int PS) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k_run<PAT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        k_run<PAT, MODE><<<256, 512, 120 * 1024>>>(in, out, nbytes, R, NM, PS);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    const int R = 16;
    float *in, *out;
    const int PSmax = 512;
    const size_t nbytes = (size_t)256 * R * 256 * PSmax;
    hipMalloc(&in, 1 << 20); hipMalloc(&out, nbytes);
    hipMemset(in, 0, 1 << 20); hipMemset(out, 0, nbytes);
    const char* names[5] = {"A dword 4px x 64B", "B x4 16px x 64B", "C x4 4px x 256B", "D x4 8px x 128B", "E x4 1KB contiguous"};
    for (int NM : {256, 512, 1024}) {
        for (int PS : {256, 512}) {
            const float t0 = run<0, 0>(in, out, (unsigned)nbytes, R, NM, PS);
            printf("NM %4d MFMAs / wave-region, pixel stride %3d B: no stores %.3f ms (%.0f cycles / region at 2.4 GHz)\n", NM, PS, t0, t0 * 2.4e6 / R);
            float tb[5], ts[5];
            tb[0] = run<0, 1>(in, out, (unsigned)nbytes, R, NM, PS); ts[0] = run<0, 2>(in, out, (unsigned)nbytes, R, NM, PS);
            tb[1] = run<1, 1>(in, out, (unsigned)nbytes, R, NM, PS); ts[1] = run<1, 2>(in, out, (unsigned)nbytes, R, NM, PS);
            tb[2] = run<2, 1>(in, out, (unsigned)nbytes, R, NM, PS); ts[2] = run<2, 2>(in, out, (unsigned)nbytes, R, NM, PS);
            tb[3] = run<3, 1>(in, out, (unsigned)nbytes, R, NM, PS); ts[3] = run<3, 2>(in, out, (unsigned)nbytes, R, NM, PS);
            tb[4] = run<4, 1>(in, out, (unsigned)nbytes, R, NM, PS); ts[4] = run<4, 2>(in, out, (unsigned)nbytes, R, NM, PS);
            for (int p = 0; p < 5; ++p)
                printf("   %-22s burst %.3f ms (+%5.0f cycles / region)   spread %.3f ms (+%5.0f)\n", names[p], tb[p], (tb[p] - t0) * 2.4e6 / R,
                       ts[p], (ts[p] - t0) * 2.4e6 / R);
        }
    }
    return 0;
}
