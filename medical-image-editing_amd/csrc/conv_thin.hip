// Thin-channel convolutions: the 1-channel stem (Cin == 1, k in {1,3}) and the 1-channel head (Cout == 1, 1x1).
// These are HBM-bound streams, not GEMMs: one thread per pixel, weights in registers/LDS, float4 I/O.
#include "common.h"
#include "conv_common.h"
#include "mfma_util.h"
#include <cstdlib>

// ---------------------------------------------------------------------------------------------
// stem forward: y[p][0..Cout) = b + sum_t x[p + shift(t)] * w[co][t]          (Cin == 1, Cout % 4 == 0, Cout <= 64)
template <int CO, bool STAGED = false>
__global__ void __launch_bounds__(256) k_stem_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                  float* __restrict__ y, int N, int H, int W, int ks, int dil, int relu) {
    __shared__ float sw[CO * 9 + CO];
    __shared__ float4 stage[STAGED ? 4 * 64 * (CO / 4 + 1) : 1];
    const int taps = ks * ks, half = ks >> 1;
    for (int i = threadIdx.x; i < CO * taps; i += 256) sw[i] = w[i];
    for (int i = threadIdx.x; i < CO; i += 256) sw[CO * 9 + i] = bias ? bias[i] : 0.f;
    __syncthreads();
    const long P = (long)N * H * W;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
        int xw = (int)(p % W);
        long q = p / W;
        int yh = (int)(q % H);
        float v[9];
        for (int t = 0; t < taps; ++t) {
            int hy = yh + (t / ks - half) * dil, wx = xw + (t % ks - half) * dil;
            v[t] = (hy >= 0 && hy < H && wx >= 0 && wx < W) ? x[p + (long)(hy - yh) * W + (wx - xw)] : 0.f;
        }
        if (STAGED) {
            // A thread's CO floats are CO / 4 stores 4 CO bytes apart: every store instruction of the wave would touch 64 separate
            // 16-byte pieces.  The wave's 64 x CO tile goes through its own LDS rows instead (row = one pixel, padded by one
            // float4) and leaves as CO / 4 stores of 1 KB in a row.  (P % 64 == 0: a wave's pixels are all valid.)
            constexpr int C4 = CO / 4, RS = C4 + 1;
            float4* tile = stage + (threadIdx.x >> 6) * 64 * RS;
            const int lane = threadIdx.x & 63;
#pragma unroll
            for (int c4 = 0; c4 < C4; ++c4) {
                float r[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int co = c4 * 4 + k;
                    float a = sw[CO * 9 + co];
                    for (int t = 0; t < taps; ++t) a = fmaf(v[t], sw[co * taps + t], a);
                    r[k] = relu ? fmaxf(a, 0.f) : a;
                }
                float4 o;
                o.x = r[0]; o.y = r[1]; o.z = r[2]; o.w = r[3];
                tile[lane * RS + c4] = o;
            }
            __builtin_amdgcn_wave_barrier();
            float4* out = (float4*)(y + (p - lane) * CO);          // the wave's first pixel
#pragma unroll
            for (int k = 0; k < C4; ++k) {
                const int f = k * 64 + lane;                       // float4 index inside the wave's tile
                out[f] = tile[(f / C4) * RS + (f % C4)];
            }
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        float4* out = (float4*)(y + p * CO);
#pragma unroll
        for (int c4 = 0; c4 < CO / 4; ++c4) {
            float r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int co = c4 * 4 + k;
                float a = sw[CO * 9 + co];
                for (int t = 0; t < taps; ++t) a = fmaf(v[t], sw[co * taps + t], a);
                r[k] = relu ? fmaxf(a, 0.f) : a;
            }
            float4 o;
            o.x = r[0]; o.y = r[1]; o.z = r[2]; o.w = r[3];
            out[c4] = o;
        }
    }
}

// stem wgrad: dW[co][t] = sum_p dy[p][co] * x[p + shift(t)], db[co] = sum_p dy[p][co].  Thread keeps CO x (taps+1)
// partial sums over its pixels; block tree-reduces through LDS; per-block partials are summed by reduce_rows.
template <int CO>
__global__ void __launch_bounds__(256) k_stem_wgrad(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                    int N, int H, int W, int ks, int dil) {
    constexpr int NA = 10;                     // 9 taps + bias
    __shared__ float sred[4][CO * NA];
    const int taps = ks * ks, half = ks >> 1;
    float acc[CO][NA];
#pragma unroll
    for (int c = 0; c < CO; ++c)
#pragma unroll
        for (int t = 0; t < NA; ++t) acc[c][t] = 0.f;
    const long P = (long)N * H * W;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
        int xw = (int)(p % W);
        long q = p / W;
        int yh = (int)(q % H);
        float v[NA];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            int hy = yh + (t / ks - half) * dil, wx = xw + (t % ks - half) * dil;
            v[t] = (t < taps && hy >= 0 && hy < H && wx >= 0 && wx < W) ? x[p + (long)(hy - yh) * W + (wx - xw)] : 0.f;
        }
        v[9] = 1.f;
        const float4* g = (const float4*)(dy + p * CO);
#pragma unroll
        for (int c4 = 0; c4 < CO / 4; ++c4) {
            float4 d = g[c4];
            float dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int t = 0; t < NA; ++t) acc[c4 * 4 + k][t] = fmaf(dv[k], v[t], acc[c4 * 4 + k][t]);
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < CO; ++c)
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            float s = wave_sum(acc[c][t]);
            if (lane == 0) sred[wv][c * NA + t] = s;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < CO * NA; i += 256)
        part[(long)blockIdx.x * CO * NA + i] = (sred[0][i] + sred[1][i]) + (sred[2][i] + sred[3][i]);
}
// sum over rows of column `col` of part[nb][ncols]; 256 threads, the same association every run
__device__ __forceinline__ float block_colsum(const float* __restrict__ part, int nb, int ncols, int col) {
    __shared__ float sred_c[256];
    float s = 0.f;
    for (int b = threadIdx.x; b < nb; b += 256) s += part[(long)b * ncols + col];
    sred_c[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sred_c[threadIdx.x] += sred_c[threadIdx.x + w];
        __syncthreads();
    }
    return sred_c[0];
}
// partial rows [nb][CO][10] -> dw [CO][taps] (+ db [CO])
__global__ void k_stem_wgrad_finalize(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db, int nb, int CO,
                                      int taps, int acc) {
    // one workgroup per output: 256 threads take the rows round-robin, fixed-order tree through LDS (one thread walking
    // all rows made this launch, not the streaming kernel in front of it, the longer of the two)
    const int i = blockIdx.x;
    const float s = block_colsum(part, nb, CO * 10, i);
    if (threadIdx.x != 0) return;
    const int c = i / 10, t = i % 10;
    if (t < taps) dw[c * taps + t] = acc ? dw[c * taps + t] + s : s;
    else if (t == 9 && db) db[c] = acc ? db[c] + s : s;
}

// ---------------------------------------------------------------------------------------------
// wide stem (Cin == 1, Cout a multiple of 64: BASELINE config 4's 1 -> 256 first block).  Lanes run over the output
// channels, so one pixel's Cout floats are one coalesced row; a thread owns 4 channels (float4) for every pixel of its
// pixel lane and keeps their 9 tap weights (forward) or 10 partial sums (weight gradient: 9 taps + bias) in registers.
// Forward writes / weight gradient reads N*H*W*Cout floats once: HBM-bound streams.
__device__ __forceinline__ void stem_taps(const float* __restrict__ x, long p, int H, int W, int ks, int dil, float* v) {
    const int xw = (int)(p % W), yh = (int)((p / W) % H), half = ks >> 1, taps = ks * ks;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int hy = yh + (t / ks - half) * dil, wx = xw + (t % ks - half) * dil;
        v[t] = (t < taps && hy >= 0 && hy < H && wx >= 0 && wx < W) ? x[p + (long)(hy - yh) * W + (wx - xw)] : 0.f;
    }
}
__global__ void __launch_bounds__(256) k_stem_wide_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ y, long P, int H, int W, int Cout, int ks, int dil, int relu) {
    const int C4 = Cout >> 2, taps = ks * ks;
    const int tc = threadIdx.x % C4, tr = threadIdx.x / C4, rows = 256 / C4;     // Cout % 64 == 0 -> C4 in {16, 32, 64}
    float wr[4][9], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        b[k] = bias ? bias[4 * tc + k] : 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[k][t] = t < taps ? w[(4 * tc + k) * taps + t] : 0.f;
    }
    for (long p = (long)blockIdx.x * rows + tr; p < P; p += (long)gridDim.x * rows) {
        float v[9];
        stem_taps(x, p, H, W, ks, dil, v);
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float a = b[k];
#pragma unroll
            for (int t = 0; t < 9; ++t) a = fmaf(v[t], wr[k][t], a);
            r[k] = relu ? fmaxf(a, 0.f) : a;
        }
        ((float4*)(y + p * Cout))[tc] = float4{r[0], r[1], r[2], r[3]};
    }
}
// part[block][Cout][10]: per-block sums in pixel order (a thread's pixels strided by the grid), the block's pixel lanes
// folded through LDS in lane order; k_stem_wgrad_finalize sums the blocks.
__global__ void __launch_bounds__(256) k_stem_wide_wgrad(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                         long P, int H, int W, int Cout, int ks, int dil) {
    extern __shared__ float s_acc[];             // [rows][Cout][10]
    const int C4 = Cout >> 2;
    const int tc = threadIdx.x % C4, tr = threadIdx.x / C4, rows = 256 / C4;
    float acc[4][10];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int t = 0; t < 10; ++t) acc[k][t] = 0.f;
    for (long p = (long)blockIdx.x * rows + tr; p < P; p += (long)gridDim.x * rows) {
        float v[10];
        stem_taps(x, p, H, W, ks, dil, v);
        v[9] = 1.f;
        const float4 d = ((const float4*)(dy + p * Cout))[tc];
        const float dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int t = 0; t < 10; ++t) acc[k][t] = fmaf(dv[k], v[t], acc[k][t]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int t = 0; t < 10; ++t) s_acc[(tr * Cout + 4 * tc + k) * 10 + t] = acc[k][t];
    __syncthreads();
    for (int i = threadIdx.x; i < Cout * 10; i += 256) {
        float a = s_acc[i];
        for (int r = 1; r < rows; ++r) a += s_acc[r * Cout * 10 + i];
        part[(long)blockIdx.x * Cout * 10 + i] = a;
    }
}
static inline bool stem_wide(int Cout) { return Cout % 64 == 0 && Cout >= 128 && Cout <= 256; }
bool conv_stem_wgrad_ok(const ConvIn& in, int Cout, int ks) {       // the weight gradient: 16 channels or the wide form
    return in.C0 == 1 && in.C1 == 0 && !in.up0 && (ks == 1 || ks == 3) && (Cout == 16 || (Cout % 64 == 0 && Cout >= 64 && Cout <= 256));
}

bool conv_stem_ok(const ConvIn& in, int Cout, int ks) {
    return in.C0 == 1 && in.C1 == 0 && !in.up0 && (ks == 1 || ks == 3) && (Cout == 16 || Cout == 32 || Cout == 64 || stem_wide(Cout));
}
int conv_stem_fwd(const ConvIn& in, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int ks, int dil,
                  int relu, hipStream_t st) {
    int g = stream_grid((long)N * H * W, 256);
    if (stem_wide(Cout)) {
        const int rows = 256 / (Cout / 4);
        k_stem_wide_fwd<<<stream_grid((long)N * H * W, rows), 256, 0, st>>>(in.src0, w, bias, y, (long)N * H * W, H, W, Cout, ks, dil, relu);
        VQW_LAUNCH_CHECK("conv_stem_fwd(wide)");
        return VQW_OK;
    }
    static const bool staged_env = []{ const char* e = getenv("VQW_STEM_STAGED"); return !e || atoi(e) != 0; }();
    const bool staged = staged_env && ((long)N * H * W) % 64 == 0;      // whole waves of valid pixels
    if (Cout == 16 && staged) k_stem_fwd<16, true><<<g, 256, 0, st>>>(in.src0, w, bias, y, N, H, W, ks, dil, relu);
    else if (Cout == 32 && staged) k_stem_fwd<32, true><<<g, 256, 0, st>>>(in.src0, w, bias, y, N, H, W, ks, dil, relu);
    else if (Cout == 16) k_stem_fwd<16><<<g, 256, 0, st>>>(in.src0, w, bias, y, N, H, W, ks, dil, relu);
    else if (Cout == 32) k_stem_fwd<32><<<g, 256, 0, st>>>(in.src0, w, bias, y, N, H, W, ks, dil, relu);
    else k_stem_fwd<64><<<g, 256, 0, st>>>(in.src0, w, bias, y, N, H, W, ks, dil, relu);
    VQW_LAUNCH_CHECK("conv_stem_fwd");
    return VQW_OK;
}
#define STEM_WG_BLOCKS 512
size_t conv_stem_wgrad_ws_floats(int Cout) { return (size_t)STEM_WG_BLOCKS * Cout * 10; }
int conv_stem_wgrad(const ConvIn& in, const float* dy, float* dw, float* dbias, float* ws, int N, int H, int W, int Cout, int ks,
                    int dil, int acc, hipStream_t st) {
    int nb = imin(STEM_WG_BLOCKS, imax(1, (int)(((long)N * H * W + 255) / 256)));
    if (Cout != 16) {
        const int rows = 256 / (Cout / 4);
        k_stem_wide_wgrad<<<nb, 256, (size_t)rows * Cout * 10 * sizeof(float), st>>>(in.src0, dy, ws, (long)N * H * W, H, W, Cout, ks, dil);
        k_stem_wgrad_finalize<<<Cout * 10, 256, 0, st>>>(ws, dw, dbias, nb, Cout, ks * ks, acc);
        VQW_LAUNCH_CHECK("conv_stem_wgrad(wide)");
        return VQW_OK;
    }
    if (Cout != 16) { vqw_set_error("conv_stem_wgrad: only Cout == 16"); return VQW_ERR_ARG; }
    k_stem_wgrad<16><<<nb, 256, 0, st>>>(in.src0, dy, ws, N, H, W, ks, dil);
    k_stem_wgrad_finalize<<<Cout * 10, 256, 0, st>>>(ws, dw, dbias, nb, Cout, ks * ks, acc);
    VQW_LAUNCH_CHECK("conv_stem_wgrad");
    return VQW_OK;
}

// ---------------------------------------------------------------------------------------------
// head: Cout == 1, 1x1, Cin % 4 == 0 (unet_decoder.py:105 conv1x1).  Also serves its dgrad (Cin == 1 -> Cout, 1x1).
__global__ void __launch_bounds__(256) k_head_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                  float* __restrict__ y, long P, int Cin, int relu) {
    extern __shared__ __attribute__((aligned(16))) float swh[];
    for (int i = threadIdx.x; i < Cin; i += 256) swh[i] = w[i];
    __syncthreads();
    const float b = bias ? bias[0] : 0.f;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
        const float4* xr = (const float4*)(x + p * Cin);
        float a = b;
        for (int c4 = 0; c4 < Cin / 4; ++c4) {
            float4 v = xr[c4];
            a = fmaf(v.x, swh[4 * c4], a); a = fmaf(v.y, swh[4 * c4 + 1], a);
            a = fmaf(v.z, swh[4 * c4 + 2], a); a = fmaf(v.w, swh[4 * c4 + 3], a);
        }
        y[p] = relu ? fmaxf(a, 0.f) : a;
    }
}
// dw[ci] = sum_p dy[p] * x[p][ci]; db = sum_p dy[p]
__global__ void __launch_bounds__(256) k_head_wgrad(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                    long P, int Cin) {
    // thread = (pixel row group, channel quad): C4 = Cin/4 lanes per pixel
    __shared__ float sacc[256 * 4 + 256];
    const int C4 = Cin >> 2;
    const int tc = threadIdx.x % C4, tr = threadIdx.x / C4, rows = 256 / C4;
    float4 a;
    a.x = a.y = a.z = a.w = 0.f;
    float bs = 0.f;
    const long per = (P + gridDim.x - 1) / gridDim.x;
    const long p0 = blockIdx.x * per, p1 = p0 + per < P ? p0 + per : P;
    if (tr < rows) {
        // four pixel rows per trip: four independent loads in flight per thread (one load per trip left the stream
        // latency-bound at ~1 TB/s); the partial sums are folded in a fixed order
        float4 b1, b2, b3;
        b1.x = b1.y = b1.z = b1.w = b2.x = b2.y = b2.z = b2.w = b3.x = b3.y = b3.z = b3.w = 0.f;
        float bs1 = 0.f, bs2 = 0.f, bs3 = 0.f;
        long p = p0 + tr;
        for (; p + 3L * rows < p1; p += 4L * rows) {
            const float g0 = dy[p], g1 = dy[p + rows], g2 = dy[p + 2L * rows], g3 = dy[p + 3L * rows];
            const float4 v0 = ((const float4*)(x + p * Cin))[tc];
            const float4 v1 = ((const float4*)(x + (p + rows) * Cin))[tc];
            const float4 v2 = ((const float4*)(x + (p + 2L * rows) * Cin))[tc];
            const float4 v3 = ((const float4*)(x + (p + 3L * rows) * Cin))[tc];
            a.x = fmaf(g0, v0.x, a.x); a.y = fmaf(g0, v0.y, a.y); a.z = fmaf(g0, v0.z, a.z); a.w = fmaf(g0, v0.w, a.w);
            b1.x = fmaf(g1, v1.x, b1.x); b1.y = fmaf(g1, v1.y, b1.y); b1.z = fmaf(g1, v1.z, b1.z); b1.w = fmaf(g1, v1.w, b1.w);
            b2.x = fmaf(g2, v2.x, b2.x); b2.y = fmaf(g2, v2.y, b2.y); b2.z = fmaf(g2, v2.z, b2.z); b2.w = fmaf(g2, v2.w, b2.w);
            b3.x = fmaf(g3, v3.x, b3.x); b3.y = fmaf(g3, v3.y, b3.y); b3.z = fmaf(g3, v3.z, b3.z); b3.w = fmaf(g3, v3.w, b3.w);
            if (tc == 0) { bs += g0; bs1 += g1; bs2 += g2; bs3 += g3; }
        }
        for (; p < p1; p += rows) {
            float g = dy[p];
            float4 v = ((const float4*)(x + p * Cin))[tc];
            a.x = fmaf(g, v.x, a.x); a.y = fmaf(g, v.y, a.y); a.z = fmaf(g, v.z, a.z); a.w = fmaf(g, v.w, a.w);
            if (tc == 0) bs += g;
        }
        a.x = (a.x + b1.x) + (b2.x + b3.x); a.y = (a.y + b1.y) + (b2.y + b3.y);
        a.z = (a.z + b1.z) + (b2.z + b3.z); a.w = (a.w + b1.w) + (b2.w + b3.w);
        bs = (bs + bs1) + (bs2 + bs3);
    }
    sacc[threadIdx.x * 4] = a.x; sacc[threadIdx.x * 4 + 1] = a.y; sacc[threadIdx.x * 4 + 2] = a.z; sacc[threadIdx.x * 4 + 3] = a.w;
    sacc[1024 + threadIdx.x] = bs;
    __syncthreads();
    if (tr == 0) {
        for (int r = 1; r < rows; ++r) {
            int t = r * C4 + tc;
            a.x += sacc[t * 4]; a.y += sacc[t * 4 + 1]; a.z += sacc[t * 4 + 2]; a.w += sacc[t * 4 + 3];
            if (tc == 0) bs += sacc[1024 + t];
        }
        float* o = part + (long)blockIdx.x * (Cin + 1);
        o[tc * 4] = a.x; o[tc * 4 + 1] = a.y; o[tc * 4 + 2] = a.z; o[tc * 4 + 3] = a.w;
        if (tc == 0) o[Cin] = bs;
    }
}
__global__ void k_head_wgrad_finalize(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db, int nb, int Cin, int acc) {
    const int i = blockIdx.x;          // one workgroup per output (see k_stem_wgrad_finalize)
    const float s = block_colsum(part, nb, Cin + 1, i);
    if (threadIdx.x != 0) return;
    if (i < Cin) dw[i] = acc ? dw[i] + s : s;
    else if (db) db[0] = acc ? db[0] + s : s;
}
bool conv_head_ok(const ConvIn& in, int Cout, int ks) {
    return Cout == 1 && ks == 1 && in.C1 == 0 && !in.up0 && (in.C0 % 4 == 0) && in.C0 >= 4 && in.C0 <= 256 && (256 % (in.C0 / 4) == 0);
}
int conv_head_fwd(const ConvIn& in, const float* w, const float* bias, float* y, long P, int relu, hipStream_t st) {
    k_head_fwd<<<stream_grid(P, 256), 256, in.C0 * sizeof(float), st>>>(in.src0, w, bias, y, P, in.C0, relu);
    VQW_LAUNCH_CHECK("conv_head_fwd");
    return VQW_OK;
}
#define HEAD_WG_BLOCKS 1024
size_t conv_head_wgrad_ws_floats(int Cin) { return (size_t)HEAD_WG_BLOCKS * (Cin + 1); }
int conv_head_wgrad(const ConvIn& in, const float* dy, float* dw, float* dbias, float* ws, long P, int acc, hipStream_t st) {
    int nb = (int)imin(HEAD_WG_BLOCKS, imax(1, (int)(P / 2048)));
    k_head_wgrad<<<nb, 256, 0, st>>>(in.src0, dy, ws, P, in.C0);
    k_head_wgrad_finalize<<<in.C0 + 1, 256, 0, st>>>(ws, dw, dbias, nb, in.C0, acc);
    VQW_LAUNCH_CHECK("conv_head_wgrad");
    return VQW_OK;
}

// ---------------------------------------------------------------------------------------------
// streaming 1x1 weight gradient for thin layers on large maps (ResBlock downsample / ASPP 1x1 branch at the full-resolution
// levels: 16->32, 32->32 @256x256 ...): dW[co][ci] = sum_p dY[p][co] * X[p][ci] is a (Cout x Cin) result over millions of
// pixels - an HBM stream (N*H*W*(Cin+Cout)*4 bytes once), not a GEMM to tile.  Every wave walks its own share of pixel
// pairs: one fp32 MFMA per pair with dY as the A operand (M = co) and X as B (N = ci), K = the two pixels (half-wave
// each), operands loaded straight from global (a wave instruction = the two pixels' contiguous rows), 8 pairs = 8 x
// (TM + TN) loads in flight per wave.  Workgroup slabs folded in wave order, slabs summed by reduce_rows: deterministic.
// (The split-K implicit-GEMM kernel reached 1.1-1.4 TB/s on these shapes.)
#define PW_BLOCKS 1024
template <int TM, int TN>
__global__ void __launch_bounds__(256) k_pw_wgrad(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                  long P, int Cin, int Cout) {
    extern __shared__ float s_slab[];             // [4 waves][Cout][Cin]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l5 = lane & 31, h = lane >> 5;
    const long gw = (long)blockIdx.x * 4 + wv, nw = (long)gridDim.x * 4;
    const long chunks = (P / 2 + 7) / 8;          // 8 pixel pairs per chunk
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // raw buffer loads: an offset past the tensor (pixels beyond P, channels beyond Cin) returns 0 - no exec-masked branches
    const __amdgpu_buffer_rsrc_t rdy = make_rsrc(dy, (unsigned)(P * Cout * 4)), rx = make_rsrc(x, (unsigned)(P * Cin * 4));
    const unsigned OOB = 0xffffff00u;
    unsigned xo[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) xo[j] = 32 * j + l5 < Cin ? (unsigned)(32 * j + l5) * 4u : OOB;
    auto load = [&](long c, float (&a)[8][TM], float (&b)[8][TN]) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long p = 2 * (c * 8 + u) + h;
            const unsigned pa = p < P ? (unsigned)(p * Cout) * 4u : OOB, pb = p < P ? (unsigned)(p * Cin) * 4u : OOB;
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[u][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rdy, (int)sel_u32(pa != OOB, pa + (32 * i + l5) * 4u, OOB), 0, 0));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[u][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, (int)sel_u32((pb != OOB) & (xo[j] != OOB), pb + xo[j], OOB), 0, 0));
        }
    };
    float a0[8][TM], b0[8][TN], a1[8][TM], b1[8][TN];
    long c = gw;
    if (c < chunks) load(c, a0, b0);
    for (; c < chunks; c += 2 * nw) {        // two register sets: the next chunk's loads are in flight under this chunk's MFMAs
        if (c + nw < chunks) load(c + nw, a1, b1);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = MFMA32(a0[u][i], b0[u][j], acc[i][j]);
        if (c + nw < chunks) {
            if (c + 2 * nw < chunks) load(c + 2 * nw, a0, b0);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = MFMA32(a1[u][i], b1[u][j], acc[i][j]);
        }
    }
    float* slab = s_slab + (size_t)wv * Cout * Cin;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * i + 8 * (r >> 2) + 4 * h + (r & 3), ci = 32 * j + l5;
                if (ci < Cin) slab[co * Cin + ci] = acc[i][j][r];
            }
    __syncthreads();
    const int n = Cout * Cin;
    for (int k = threadIdx.x; k < n; k += 256)
        part[(size_t)blockIdx.x * n + k] = (s_slab[k] + s_slab[n + k]) + (s_slab[2 * n + k] + s_slab[3 * n + k]);
}
bool conv_pw_wgrad_ok(const ConvIn& in, int Cout, int ks, long P) {
    const int Cin = in.C0;
    return ks == 1 && in.C1 == 0 && !in.up0 && (Cin == 16 || Cin == 32 || Cin == 64) && (Cout == 32 || Cout == 64 || Cout == 128) &&
           (long)Cin * Cout <= 4096 && (P & 1) == 0 && P >= 131072 && P * (long)(Cin > Cout ? Cin : Cout) * 4 < 0xffffff00L;
}
size_t conv_pw_wgrad_ws_floats(int Cin, int Cout) { return (size_t)PW_BLOCKS * Cin * Cout; }
int conv_pw_wgrad(const ConvIn& in, const float* dy, float* dw, float* ws, long P, int Cout, int acc, hipStream_t st) {
    const int Cin = in.C0, TM = Cout / 32, TN = (Cin + 31) / 32;
    const long chunks = (P / 2 + 7) / 8;
    const int nb = (int)(chunks / 4 < PW_BLOCKS ? (chunks + 3) / 4 : PW_BLOCKS);
    const size_t lds = (size_t)4 * Cin * Cout * sizeof(float);       // <= 64 KB by conv_pw_wgrad_ok
#define PW_CASE(TM_, TN_) if (TM == TM_ && TN == TN_) k_pw_wgrad<TM_, TN_><<<nb, 256, lds, st>>>(in.src0, dy, ws, P, Cin, Cout)
    PW_CASE(1, 1); else PW_CASE(2, 1); else PW_CASE(4, 1); else PW_CASE(1, 2); else PW_CASE(2, 2);
    else { vqw_set_error("conv_pw_wgrad: unsupported tile %d x %d", TM, TN); return VQW_ERR_ARG; }
#undef PW_CASE
    VQW_LAUNCH_CHECK("conv_pw_wgrad");
    return reduce_rows(ws, dw, (long)Cin * Cout, nb, st, acc);
}

// =====================================================================================================================
// Streaming 1 x 1 convolution (round 4): forward and input gradient of the channel-mixing layers on the large maps
// (ResBlock projections 16->32 @256 / 32->64 @128, the pyramid's 1 x 1 branch 32->32 @256; blocks.py:25, aspp.py:25)
// =====================================================================================================================
// These layers are HBM-bound (2-5 FLOP per byte) and ran on the implicit-GEMM kernel at 3.0-4.0 TB/s: a 128-pixel MFMA tile
// with LDS staging, barriers and dword stores per 24 KB of traffic.  Here: y[p][n] = sum_k x[p][k] B[k][n] on the vector lanes.
// A 256-thread workgroup covers R = 256 / (NOUT / 4) whole pixels per pass; thread t keeps its output quad q4 = t % (NOUT / 4)
// for the whole walk (four passes in flight), B (KIN x NOUT, <= 16 KB) sits in LDS and is read as broadcast float4 rows, the
// pixel's KIN inputs arrive as KIN / 4 float4 loads that the NOUT / 4 threads of the pixel share in the texture cache, the
// result leaves as one float4 store: 16-byte accesses end to end.  MODE 0: y = result (+ bias); 1: y += result (a later
// member of a gradient group).  With `part` the workgroup also leaves the (sum, M2 about the tile mean) statistics partials
// of its TILE pixels per output channel for the InstanceNorm that follows (shifted sums per thread, Chan merges across the
// threads of a channel in a fixed order).
template <int KIN, int NOUT, int MODE>
__global__ void __launch_bounds__(256) k_pw_stream(const float4* __restrict__ x, const float* __restrict__ Bm, const float* __restrict__ bias,
                                                   float4* __restrict__ y, float* __restrict__ part, long P, int tile_px) {
    constexpr int Q = NOUT / 4, R = 256 / Q, K4 = KIN / 4;
    __shared__ float4 sB[KIN * Q];
    __shared__ float sred[4][2][NOUT];
    const int t = threadIdx.x, q4 = t % Q, r = t / Q;
    // Bm arrives as the layer's OHWI weights [NOUT][KIN]; the mixing matrix is its transpose [KIN][NOUT]
    for (int i = t; i < KIN * NOUT; i += 256) ((float*)sB)[i] = Bm[(i % NOUT) * KIN + i / NOUT];
    __syncthreads();
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = ((const float4*)bias)[q4];
    const long p_begin = (long)blockIdx.x * tile_px;
    const long p_end = p_begin + tile_px < P ? p_begin + tile_px : P;
    // statistics: shifted sums about the thread's first value per channel (n_thr values per thread, equal for every thread)
    float sh[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    bool have_shift = false;
    for (long p0 = p_begin + r; p0 < p_end; p0 += 2 * R) {
        float4 acc[2];
        float4 xin[2][K4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long p = p0 + u * R;
            if (p < p_end) {
#pragma unroll
                for (int j = 0; j < K4; ++j) xin[u][j] = x[p * K4 + j];
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long p = p0 + u * R;
            if (p >= p_end) break;
            float4 a = bv;
            // B's rows at q4 do not change from pixel to pixel: the compiler keeps them in registers (KIN float4: fine up to
            // KIN = 32, 422 registers at 64) - there an index it cannot see through keeps them in LDS
            int qo = q4;
#ifdef PW_NOHOIST32
            if (KIN >= 32) asm volatile("" : "+v"(qo));
#else
            if (KIN > 32) asm volatile("" : "+v"(qo));
#endif
#pragma unroll
            for (int j = 0; j < K4; ++j) {
                const float xv[4] = {xin[u][j].x, xin[u][j].y, xin[u][j].z, xin[u][j].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 b = sB[(4 * j + k) * Q + qo];
                    a.x = fmaf(xv[k], b.x, a.x); a.y = fmaf(xv[k], b.y, a.y); a.z = fmaf(xv[k], b.z, a.z); a.w = fmaf(xv[k], b.w, a.w);
                }
            }
            if (MODE == 1) {
                const float4 o = y[p * Q + q4];
                a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
            }
            y[p * Q + q4] = a;
            acc[u] = a;
            if (part) {
                const float av[4] = {a.x, a.y, a.z, a.w};
                if (!have_shift) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) sh[k] = av[k];
                    have_shift = true;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float d = av[k] - sh[k]; s1[k] += d; s2[k] = fmaf(d, d, s2[k]); }
            }
        }
        (void)acc;
    }
    if (part) {        // uniform; every thread has seen n_thr = tile_px / R pixels (the launcher guarantees whole tiles)
        const float n_thr = (float)(tile_px / R);
        float sum[4], m2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sum[k] = fmaf(n_thr, sh[k], s1[k]);                    // sum of the thread's values
            m2[k] = s2[k] - s1[k] * s1[k] / n_thr;                 // about the thread's own mean
        }
        // threads that share q4: inside a wave the lanes q4 + Q i (xor offsets Q, 2 Q, ... < 64), then the four waves in order
        float n = n_thr;
        for (int off = Q; off < 64; off <<= 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) stat_merge_eq(sum[k], m2[k], __shfl_xor(sum[k], off, 64), __shfl_xor(m2[k], off, 64), 0.5f / n);
            n *= 2.f;
        }
        const int lane = t & 63, wv = t >> 6;
        if (lane < Q) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { sred[wv][0][4 * lane + k] = sum[k]; sred[wv][1][4 * lane + k] = m2[k]; }
        }
        __syncthreads();
        if (t < NOUT) {
            // Q >= 64 cannot happen (NOUT <= 64 -> Q <= 16): every wave holds every quad
            float a = sred[0][0][t], b = sred[0][1][t];
            stat_merge_eq(a, b, sred[1][0][t], sred[1][1][t], 0.5f / n);
            float c = sred[2][0][t], d = sred[2][1][t];
            stat_merge_eq(c, d, sred[3][0][t], sred[3][1][t], 0.5f / n);
            stat_merge_eq(a, b, c, d, 0.25f / n);
            float* o = part + ((size_t)blockIdx.x * NOUT + t) * 2;
            o[0] = a;
            o[1] = b;
        }
    }
}

static const int g_pw_stream = []{ const char* e = getenv("VQW_PW_STREAM"); return e ? atoi(e) : 1; }();      // 0: implicit-GEMM kernel (A/B)
static inline int pw_tile_px(int HW) { return HW % 1024 == 0 ? 1024 : (HW % 512 == 0 ? 512 : 0); }
// K -> Nout channel mixing on a map of at least 128 x 128 pixels per image, both channel counts in {16, 32, 64}
bool conv_pw_stream_ok(int K, int Nout, int N, int HW) {
    auto okc = [](int c) { return c == 16 || c == 32 || c == 64; };
    return g_pw_stream && okc(K) && okc(Nout) && HW >= 16384 && pw_tile_px(HW) > 0 && (long)N * HW * (K > Nout ? K : Nout) * 4 < 0xffffff00L;
}
int conv_pw_stream_stat_tiles(int HW) { const int tp = pw_tile_px(HW); return tp ? HW / tp : 0; }
// Bm: the OHWI weights [Nout][K] of the 1 x 1 layer that maps K -> Nout channels (transposed while they are staged in LDS)
int conv_pw_stream(const float* x, const float* Bm, const float* bias, float* y, float* part, int N, int HW, int K, int Nout, int mode,
                   hipStream_t st) {
    const long P = (long)N * HW;
    const int tp = pw_tile_px(HW);
    const int grid = (int)((P + tp - 1) / tp);
#define PWS(K_, N_)                                                                                                                 \
    if (K == K_ && Nout == N_) {                                                                                                    \
        if (mode == 1) k_pw_stream<K_, N_, 1><<<grid, 256, 0, st>>>((const float4*)x, Bm, bias, (float4*)y, part, P, tp);             \
        else k_pw_stream<K_, N_, 0><<<grid, 256, 0, st>>>((const float4*)x, Bm, bias, (float4*)y, part, P, tp);                       \
    } else
    PWS(16, 16) PWS(16, 32) PWS(16, 64) PWS(32, 16) PWS(32, 32) PWS(32, 64) PWS(64, 16) PWS(64, 32) PWS(64, 64)
    { vqw_set_error("conv_pw_stream: channel counts %d -> %d not served", K, Nout); return VQW_ERR_ARG; }
#undef PWS
    VQW_LAUNCH_CHECK("conv_pw_stream");
    return VQW_OK;
}
