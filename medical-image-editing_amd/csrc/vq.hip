// Vector-quantisation kernels: nearest-codebook search, gather, commitment loss, EMA statistics.
// Replaces networks/vq/vq_module.py:45-62 (_torch_knn), :168-202 (_quantize), :204-207 (lookup) and
// networks/vq/grad_approximation.py:7-29 of the reference.  Rows are NHWC pixels: x[Npix][D].
//
// Score (same association order as the reference): s_k = ((2 * (e_k . x)) - |e_k|^2) - |x|^2, fp32 FMA
// chain over d ascending; arg-max over k with ties to the LOWEST index.  The K x N score matrix, the N x K
// one-hot and the D x N x K "embed_sum" GEMM of the reference are never materialised.
#include "common.h"
#include "conv_common.h"
#include "../../include/vqwnet_hip.h"

#define VQ_BLOCK 256
#define VQ_LDS_FLOATS 15360  // 60 KiB of LDS for codebook + norms (+ privatised stats)

static inline int vq_blocks(long Npix) { return (int)imin(2048, ceil_div(Npix, VQ_BLOCK)); }

static inline bool vq_lds_codebook(int D, int K) { return (long)K * D + K <= VQ_LDS_FLOATS; }
static inline bool vq_lds_stats(int D, int K) { return (long)K * D + K + 4L * K * (D + 1) <= VQ_LDS_FLOATS; }   // one statistics slab per wave

// ---- large codebooks (BASELINE config 4: K = 1024, D = 256): score GEMM on the matrix cores + wave-per-pixel select
#define VQ_GEMM_CHUNK 65536L          // pixels per score chunk (the K x N matrix is never held whole)
static inline bool vq_use_gemm(int D, int K) { return !vq_lds_codebook(D, K) && K >= 64 && (K % 4 == 0) && (D % 4 == 0) && D >= 8; }
static inline long vq_gemm_chunk(long Npix) { return Npix < VQ_GEMM_CHUNK ? Npix : VQ_GEMM_CHUNK; }
static inline size_t vq_gemm_ws_bytes(long Npix, int D, int K) {
    size_t cpart = (((size_t)(Npix + 3) / 4) * sizeof(double) + 255) / 256 * 256;
    size_t accum = ((size_t)K * (D + 1) + K) * sizeof(float) + 256;
    return cpart + accum + (size_t)vq_gemm_chunk(Npix) * K * sizeof(float) + 256;
}

extern "C" size_t vqw_vq_ws_bytes(long Npix, int D, int K) {
    if (vq_use_gemm(D, K)) return vq_gemm_ws_bytes(Npix, D, K);
    size_t nb = (size_t)vq_blocks(Npix);
    // per-block commit partial (double) + per-block stats partials (float) or ONE global float accumulator
    size_t rows = vq_lds_stats(D, K) ? nb : 1;
    return ((nb * sizeof(double) + 255) / 256) * 256 + rows * (size_t)K * (D + 1) * sizeof(float) + 256;
}

// DT > 0: query row held in DT registers.  DT == 0: generic D (row re-read from global/L1).
// LDS_CB: codebook + norms staged in LDS.  LDS_ST: EMA statistics privatised in LDS (per-block partials).
template <int DT, bool LDS_CB, bool LDS_ST>
__global__ void __launch_bounds__(VQ_BLOCK) k_vq_fwd(const float* __restrict__ x, const float* __restrict__ embed,
                                                     int64_t* __restrict__ ids, float* __restrict__ q,
                                                     double* __restrict__ commit_part, float* __restrict__ stat_part,
                                                     float* __restrict__ stat_global, long Npix, int D, int K, int want_stats,
                                                     int id_base) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_cb = smem;                                   // [K][D] if LDS_CB
    float* s_nrm = smem + (LDS_CB ? K * D : 0);           // [K]
    float* s_st = s_nrm + K;                              // [waves][K*(D+1)] if LDS_ST: counts[K] then sum[d][k], one slab per wave
    __shared__ double s_red[VQ_BLOCK / 64];
    const int t = threadIdx.x;
    if (LDS_CB)
        for (int i = t; i < K * D; i += VQ_BLOCK) s_cb[i] = embed[i];
    if (LDS_ST && want_stats)
        for (int i = t; i < (VQ_BLOCK / 64) * K * (D + 1); i += VQ_BLOCK) s_st[i] = 0.f;
    for (int k = t; k < K; k += VQ_BLOCK) {
        float n2 = 0.f;
        for (int d = 0; d < D; ++d) { float e = embed[k * D + d]; n2 = fmaf(e, e, n2); }
        s_nrm[k] = n2;
    }
    __syncthreads();
    const float* cb = LDS_CB ? s_cb : embed;
    double csum = 0.0;
    // wave-uniform trip count: the statistics below are reduced with wave shuffles, so every lane stays in the loop
    // (lanes past the end compute on the last pixel and contribute nothing)
    for (long base = (long)blockIdx.x * VQ_BLOCK; base < Npix; base += (long)gridDim.x * VQ_BLOCK) {
        const bool active = base + t < Npix;
        const long p = active ? base + t : Npix - 1;
        const float* xr = x + p * D;
        float xv[DT > 0 ? DT : 1];
        float x2 = 0.f;
        if (DT > 0) {
#pragma unroll
            for (int d4 = 0; d4 < DT / 4; ++d4) {
                float4 v = ((const float4*)xr)[d4];
                xv[4 * d4] = v.x; xv[4 * d4 + 1] = v.y; xv[4 * d4 + 2] = v.z; xv[4 * d4 + 3] = v.w;
            }
#pragma unroll
            for (int d = 0; d < DT; ++d) x2 = fmaf(xv[d], xv[d], x2);
        } else {
            for (int d = 0; d < D; ++d) { float v = xr[d]; x2 = fmaf(v, v, x2); }
        }
        float best = -INFINITY;
        int bi = 0;
        for (int k = 0; k < K; ++k) {
            const float* e = cb + k * D;
            float dot = 0.f;
            if (DT > 0) {
#pragma unroll
                for (int d = 0; d < DT; ++d) dot = fmaf(e[d], xv[d], dot);
            } else {
                for (int d = 0; d < D; ++d) dot = fmaf(e[d], xr[d], dot);
            }
            float s = (2.f * dot - s_nrm[k]) - x2;
            if (s > best) { best = s; bi = k; }
        }
        if (active) ids[p] = (int64_t)(bi + id_base);
        const float* e = cb + bi * D;
        float* qr = q + p * D;
        float c = 0.f;
        if (DT > 0) {
#pragma unroll
            for (int d4 = 0; d4 < DT / 4; ++d4) {
                float4 o;
                o.x = e[4 * d4]; o.y = e[4 * d4 + 1]; o.z = e[4 * d4 + 2]; o.w = e[4 * d4 + 3];
                if (active) ((float4*)qr)[d4] = o;
                float a0 = xv[4 * d4] - o.x, a1 = xv[4 * d4 + 1] - o.y, a2 = xv[4 * d4 + 2] - o.z, a3 = xv[4 * d4 + 3] - o.w;
                c = fmaf(a0, a0, c); c = fmaf(a1, a1, c); c = fmaf(a2, a2, c); c = fmaf(a3, a3, c);
            }
        } else {
            for (int d = 0; d < D; ++d) { float ev = e[d]; if (active) qr[d] = ev; float a = xr[d] - ev; c = fmaf(a, a, c); }
        }
        if (active) csum += (double)c;
        if (want_stats && LDS_ST) {
            // deterministic: per code, a fixed-order wave butterfly of the member lanes' values, added by ONE lane into
            // the wave's own LDS slab (no atomics: the summation order never depends on scheduling)
            float* slab = s_st + (t >> 6) * K * (D + 1);
            const int code = active ? bi : -1;
            for (int k = 0; k < K; ++k) {
                const unsigned long long members = __ballot(code == k);
                if (members == 0ull) continue;
                const bool mine = code == k;
                if ((t & 63) == 0) slab[k] += (float)__popcll(members);
                if (DT > 0) {
#pragma unroll
                    for (int d = 0; d < DT; ++d) {
                        float v = wave_sum_f(mine ? xv[d] : 0.f);
                        if ((t & 63) == 0) slab[K + d * K + k] += v;
                    }
                } else {
                    for (int d = 0; d < D; ++d) {
                        float v = wave_sum_f(mine ? xr[d] : 0.f);
                        if ((t & 63) == 0) slab[K + d * K + k] += v;
                    }
                }
            }
        } else if (want_stats && active) {
            atomicAdd(stat_global + bi, 1.f);
            if (DT > 0) {
#pragma unroll
                for (int d = 0; d < DT; ++d) atomicAdd(stat_global + K + d * K + bi, xv[d]);
            } else {
                for (int d = 0; d < D; ++d) atomicAdd(stat_global + K + d * K + bi, xr[d]);
            }
        }
    }
    csum = wave_sum_d(csum);
    if ((t & 63) == 0) s_red[t >> 6] = csum;
    __syncthreads();
    if (t == 0) {
        double a = 0.0;
        for (int w = 0; w < VQ_BLOCK / 64; ++w) a += s_red[w];
        commit_part[blockIdx.x] = a;
    }
    if (LDS_ST && want_stats) {          // fold the wave slabs in wave order
        float* o = stat_part + (long)blockIdx.x * K * (D + 1);
        const int KD1 = K * (D + 1);
        for (int i = t; i < KD1; i += VQ_BLOCK) {
            float a = s_st[i];
            for (int w = 1; w < VQ_BLOCK / 64; ++w) a += s_st[w * KD1 + i];
            o[i] = a;
        }
    }
}

__global__ void k_vq_finalize(const double* __restrict__ commit_part, const float* __restrict__ stat_part, int nblocks,
                              int nstat_rows, float* __restrict__ commit, double* __restrict__ stats, int KD1, double inv_numel) {
    __shared__ double s_red[4];
    const int t = threadIdx.x;
    if (blockIdx.x == 0) {
        double a = 0.0;
        for (int i = t; i < nblocks; i += blockDim.x) a += commit_part[i];
        a = wave_sum_d(a);
        if ((t & 63) == 0) s_red[t >> 6] = a;
        __syncthreads();
        if (t == 0) commit[0] = (float)((s_red[0] + s_red[1] + s_red[2] + s_red[3]) * inv_numel);
    }
    if (stats) {      // column sums of the per-block partial rows: 16 columns x 16 row groups per workgroup, fixed order
        __shared__ double sm[16][17];
        const int cx = t & 15, g = t >> 4;
        for (int base = blockIdx.x * 16; base < KD1; base += gridDim.x * 16) {
            const int i = base + cx;
            double a = 0.0;
            if (i < KD1)
                for (int b = g; b < nstat_rows; b += 16) a += (double)stat_part[(long)b * KD1 + i];
            sm[g][cx] = a;
            __syncthreads();
            if (g == 0 && i < KD1) {
                double s2 = 0.0;
                for (int k = 0; k < 16; ++k) s2 += sm[k][cx];
                stats[i] = s2;
            }
            __syncthreads();
        }
    }
}

__global__ void k_vq_enorm(const float* __restrict__ embed, float* __restrict__ enorm, int D, int K) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    float n2 = 0.f;
    for (int d = 0; d < D; ++d) { float e = embed[(long)k * D + d]; n2 = fmaf(e, e, n2); }
    enorm[k] = n2;
}

// One wave per pixel: final scores ((2*dot - |e|^2) - |x|^2), arg-max with ties to the lowest index, gather, commitment
// partial, EMA statistics by float atomics into a [counts K | sums K x D] accumulator (contiguous 4*D bytes per code).
__global__ void __launch_bounds__(256) k_vq_select(const float* __restrict__ scores, const float* __restrict__ x,
                                                   const float* __restrict__ embed, const float* __restrict__ enorm,
                                                   int64_t* __restrict__ ids, float* __restrict__ q, double* __restrict__ cpart,
                                                   float* __restrict__ accum, long p0, long pn, int D, int K, int id_base) {
    __shared__ float s_c[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long pl = (long)blockIdx.x * 4 + wv;          // pixel inside the chunk
    float csum = 0.f;
    if (pl < pn) {
        const long p = p0 + pl;
        const float* xr = x + p * D;
        float x2 = 0.f;
        for (int d = lane * 4; d < D; d += 256) {
            float4 v = *(const float4*)(xr + d);
            x2 = fmaf(v.x, v.x, x2); x2 = fmaf(v.y, v.y, x2); x2 = fmaf(v.z, v.z, x2); x2 = fmaf(v.w, v.w, x2);
        }
        x2 = wave_sum(x2);
        const float* sr = scores + pl * K;
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int k = lane * 4; k < K; k += 256) {
            float4 sv = *(const float4*)(sr + k);
            float4 en = *(const float4*)(enorm + k);
            float c0 = (2.f * sv.x - en.x) - x2, c1 = (2.f * sv.y - en.y) - x2, c2 = (2.f * sv.z - en.z) - x2, c3 = (2.f * sv.w - en.w) - x2;
            if (c0 > best) { best = c0; bi = k; }
            if (c1 > best) { best = c1; bi = k + 1; }
            if (c2 > best) { best = c2; bi = k + 2; }
            if (c3 > best) { best = c3; bi = k + 3; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ob = __shfl_xor(best, o, 64);
            int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) ids[p] = (int64_t)(bi + id_base);
        const float* e = embed + (long)bi * D;
        float* qr = q + p * D;
        float c = 0.f;
        for (int d = lane * 4; d < D; d += 256) {
            float4 ev = *(const float4*)(e + d), xv = *(const float4*)(xr + d);
            *(float4*)(qr + d) = ev;
            float a0 = xv.x - ev.x, a1 = xv.y - ev.y, a2 = xv.z - ev.z, a3 = xv.w - ev.w;
            c = fmaf(a0, a0, c); c = fmaf(a1, a1, c); c = fmaf(a2, a2, c); c = fmaf(a3, a3, c);
            if (accum) {
                float* ar = accum + K + (long)bi * D + d;
                atomicAdd(ar, xv.x); atomicAdd(ar + 1, xv.y); atomicAdd(ar + 2, xv.z); atomicAdd(ar + 3, xv.w);
            }
        }
        if (accum && lane == 0) atomicAdd(accum + bi, 1.f);
        csum = wave_sum(c);
    }
    if (lane == 0) s_c[wv] = csum;
    __syncthreads();
    if (threadIdx.x == 0) cpart[p0 / 4 + blockIdx.x] = (double)s_c[0] + (double)s_c[1] + (double)s_c[2] + (double)s_c[3];
}

// commit = sum(cpart)/numel; stats[k] = counts, stats[K + d*K + k] = sums[k][d]  (the reference's embed_avg layout)
__global__ void k_vq_gemm_finalize(const double* __restrict__ cpart, long ncp, const float* __restrict__ accum, float* __restrict__ commit,
                                   double* __restrict__ stats, int D, int K, double inv_numel) {
    __shared__ double s_red[4];
    const int t = threadIdx.x;
    if (blockIdx.x == 0) {
        double a = 0.0;
        for (long i = t; i < ncp; i += blockDim.x) a += cpart[i];
        a = wave_sum_d(a);
        if ((t & 63) == 0) s_red[t >> 6] = a;
        __syncthreads();
        if (t == 0) commit[0] = (float)((s_red[0] + s_red[1] + s_red[2] + s_red[3]) * inv_numel);
    }
    if (stats) {
        const long n = (long)K * (D + 1);
        for (long i = (long)blockIdx.x * blockDim.x + t; i < n; i += (long)gridDim.x * blockDim.x) {
            if (i < K) stats[i] = (double)accum[i];
            else {
                long r = i - K;
                int d = (int)(r / K), k = (int)(r % K);
                stats[i] = (double)accum[K + (long)k * D + d];
            }
        }
    }
}

static int vq_fwd_gemm(const float* x, const float* embed, int64_t* ids, int id_base, float* q, float* commit, double* stats,
                       void* ws, long Npix, int D, int K, hipStream_t st) {
    const size_t cp_bytes = (((size_t)(Npix + 3) / 4) * sizeof(double) + 255) / 256 * 256;
    double* cpart = (double*)ws;
    float* accum = (float*)((char*)ws + cp_bytes);                       // [K] counts, [K][D] sums, then [K] code norms
    float* enorm = accum + (size_t)K * (D + 1);
    float* scores = (float*)((char*)accum + (((size_t)K * (D + 1) + K) * sizeof(float) + 255) / 256 * 256);
    if (stats && hipMemsetAsync(accum, 0, (size_t)K * (D + 1) * sizeof(float), st) != hipSuccess) {
        vqw_set_error("vqw_vq_fwd: memset failed");
        return VQW_ERR_HIP;
    }
    k_vq_enorm<<<ceil_div(K, 256), 256, 0, st>>>(embed, enorm, D, K);
    const long PC = vq_gemm_chunk(Npix);
    for (long p0 = 0; p0 < Npix; p0 += PC) {
        const long pn = Npix - p0 < PC ? Npix - p0 : PC;
        // scores[p][k] = x_p . e_k  == a 1x1 convolution with Cout = K over the pixel chunk (fp32 MFMA implicit GEMM)
        ConvIn in{x + p0 * D, nullptr, D, 0, 0};
        int rc = conv_mfma_fwd(in, embed, nullptr, scores, 1, 1, (int)pn, K, 1, 1, 0, st);
        if (rc) return rc;
        k_vq_select<<<(unsigned)((pn + 3) / 4), 256, 0, st>>>(scores, x, embed, enorm, ids, q, cpart, stats ? accum : nullptr, p0, pn, D,
                                                              K, id_base);
    }
    VQW_LAUNCH_CHECK("vqw_vq_fwd(gemm)");
    k_vq_gemm_finalize<<<stats ? 256 : 1, 256, 0, st>>>(cpart, (Npix + 3) / 4, accum, commit, stats, D, K, 1.0 / ((double)Npix * D));
    VQW_LAUNCH_CHECK("vqw_vq_gemm_finalize");
    return VQW_OK;
}

template <int DT>
static int launch_vq(const float* x, const float* embed, int64_t* ids, float* q, double* cpart, float* spart, float* sglob,
                     long Npix, int D, int K, int want, int id_base, bool lds_cb, bool lds_st, int nb, size_t lds_bytes,
                     hipStream_t st) {
    if (lds_cb && lds_st) k_vq_fwd<DT, true, true><<<nb, VQ_BLOCK, lds_bytes, st>>>(x, embed, ids, q, cpart, spart, sglob, Npix, D, K, want, id_base);
    else if (lds_cb) k_vq_fwd<DT, true, false><<<nb, VQ_BLOCK, lds_bytes, st>>>(x, embed, ids, q, cpart, spart, sglob, Npix, D, K, want, id_base);
    else k_vq_fwd<DT, false, false><<<nb, VQ_BLOCK, lds_bytes, st>>>(x, embed, ids, q, cpart, spart, sglob, Npix, D, K, want, id_base);
    return 0;
}

extern "C" int vqw_vq_fwd(const float* x, const float* embed, int64_t* ids, int id_base, float* q, float* commit, double* stats,
                          void* ws, size_t ws_bytes, long Npix, int D, int K, void* stream) {
    VQW_CHECK(x && embed && ids && q && commit && ws && Npix > 0 && D > 0 && K > 0, "vqw_vq_fwd: bad arguments");
    VQW_CHECK(ws_bytes >= vqw_vq_ws_bytes(Npix, D, K), "vqw_vq_fwd: workspace too small");
    VQW_CHECK(K <= 8192, "vqw_vq_fwd: dict_size %d exceeds the supported 8192", K);
    hipStream_t st = (hipStream_t)stream;
    VQW_CHECK((((uintptr_t)x | (uintptr_t)q) & 15) == 0, "vqw_vq_fwd: x and q must be 16-byte aligned");
    if (vq_use_gemm(D, K)) return vq_fwd_gemm(x, embed, ids, id_base, q, commit, stats, ws, Npix, D, K, st);
    const int nb = vq_blocks(Npix);
    const int KD1 = K * (D + 1);
    double* cpart = (double*)ws;
    float* spart = (float*)((char*)ws + (((size_t)nb * sizeof(double) + 255) / 256) * 256);
    VQW_CHECK((((uintptr_t)x | (uintptr_t)q) & 15) == 0, "vqw_vq_fwd: x and q must be 16-byte aligned");
    bool lds_cb = vq_lds_codebook(D, K);
    bool lds_st = vq_lds_stats(D, K);
    size_t lds_floats = (size_t)K + (lds_cb ? (size_t)K * D : 0) + (lds_st ? (size_t)(VQ_BLOCK / 64) * KD1 : 0);
    int want = stats != nullptr;
    float* sglob = spart;  // global accumulator (row 0) when not privatised
    if (want && !lds_st) {
        hipError_t e = hipMemsetAsync(sglob, 0, (size_t)KD1 * sizeof(float), st);
        if (e != hipSuccess) { vqw_set_error("vqw_vq_fwd: memset failed"); return VQW_ERR_HIP; }
    }
    size_t lb = lds_floats * sizeof(float);
    if (D == 16) launch_vq<16>(x, embed, ids, q, cpart, spart, sglob, Npix, D, K, want, id_base, lds_cb, lds_st, nb, lb, st);
    else if (D == 32) launch_vq<32>(x, embed, ids, q, cpart, spart, sglob, Npix, D, K, want, id_base, lds_cb, lds_st, nb, lb, st);
    else if (D == 64) launch_vq<64>(x, embed, ids, q, cpart, spart, sglob, Npix, D, K, want, id_base, lds_cb, lds_st, nb, lb, st);
    else launch_vq<0>(x, embed, ids, q, cpart, spart, sglob, Npix, D, K, want, id_base, lds_cb, lds_st, nb, lb, st);
    VQW_LAUNCH_CHECK("vqw_vq_fwd");
    k_vq_finalize<<<imax(1, imin(256, ceil_div(KD1, 16))), 256, 0, st>>>(cpart, spart, nb, lds_st ? nb : 1, commit,
                                                                         want ? stats : nullptr, KD1,
                                                                         1.0 / ((double)Npix * D));
    VQW_LAUNCH_CHECK("vqw_vq_finalize");
    return VQW_OK;
}

// EMA (vq_module.py:132-136,195-196) + Laplace-smoothed renormalisation (:198-200); single small block.
__global__ void k_vq_ema(const double* __restrict__ stats, float* __restrict__ embed, float* __restrict__ cs,
                         float* __restrict__ ea, float m, float eps, float sum_scale, int D, int K) {
    __shared__ double s_red[4];
    __shared__ float s_n;
    const int t = threadIdx.x;
    const float om = 1.f - m;
    double part = 0.0;
    for (int k = t; k < K; k += blockDim.x) {
        float c = cs[k] * m + om * (float)stats[k];
        cs[k] = c;
        part += (double)c;
    }
    for (int i = t; i < D * K; i += blockDim.x) ea[i] = ea[i] * m + om * ((float)stats[K + i] * sum_scale);
    part = wave_sum_d(part);
    if ((t & 63) == 0) s_red[t >> 6] = part;
    __syncthreads();
    if (t == 0) s_n = (float)(s_red[0] + s_red[1] + s_red[2] + s_red[3]);
    __syncthreads();
    const float n = s_n;
    for (int i = t; i < D * K; i += blockDim.x) {
        int k = i / D, d = i % D;
        float csn = n * (cs[k] + eps) / (n + (float)K * eps);
        embed[i] = ea[d * K + k] / csn;
    }
}
extern "C" int vqw_vq_ema_update(const double* stats, float* embed, float* cluster_size, float* embed_avg, float momentum,
                                 float eps, float sum_scale, int D, int K, void* stream) {
    VQW_CHECK(stats && embed && cluster_size && embed_avg && D > 0 && K > 0, "vqw_vq_ema_update: bad arguments");
    k_vq_ema<<<1, 256, 0, (hipStream_t)stream>>>(stats, embed, cluster_size, embed_avg, momentum, eps, sum_scale, D, K);
    VQW_LAUNCH_CHECK("vqw_vq_ema_update");
    return VQW_OK;
}

__global__ void k_vq_lookup(const int64_t* __restrict__ ids, const float* __restrict__ embed, const uint8_t* __restrict__ mask,
                            const float* __restrict__ scale, float* __restrict__ out, long total, int D, int K) {
    long stride = (long)gridDim.x * blockDim.x;
    float sc = scale ? scale[0] : 1.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        long p = i / D;
        int d = (int)(i % D);
        int64_t k = ids[p];
        float v = (k >= 0 && k < K) ? embed[k * D + d] : 0.f;
        if (mask) v = mask[p] ? v * sc : 0.f;
        out[i] = v;
    }
}
extern "C" int vqw_vq_lookup(const int64_t* ids, const float* embed, const uint8_t* mask, const float* scale_dev, float* out,
                             long Npix, int D, int K, void* stream) {
    VQW_CHECK(ids && embed && out && Npix > 0 && D > 0 && K > 0, "vqw_vq_lookup: bad arguments");
    long total = Npix * D;
    k_vq_lookup<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(ids, embed, mask, scale_dev, out, total, D, K);
    VQW_LAUNCH_CHECK("vqw_vq_lookup");
    return VQW_OK;
}

__global__ void k_vq_bwd(const float* __restrict__ x, const float* __restrict__ q, const float* __restrict__ gq,
                         const float* __restrict__ gc, float* __restrict__ gx, long n, float two_over_n) {
    float s = gc ? gc[0] * two_over_n : 0.f;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float g = gq ? gq[i] : 0.f;
        gx[i] = fmaf(s, x[i] - q[i], g);
    }
}
extern "C" int vqw_vq_bwd(const float* x, const float* q, const float* g_q, const float* g_commit, float* gx, long numel,
                          void* stream) {
    VQW_CHECK(x && q && gx && numel > 0, "vqw_vq_bwd: bad arguments");
    k_vq_bwd<<<stream_grid(numel, 256), 256, 0, (hipStream_t)stream>>>(x, q, g_q, g_commit, gx, numel, 2.0f / (float)numel);
    VQW_LAUNCH_CHECK("vqw_vq_bwd");
    return VQW_OK;
}

// run_recon.py:179-192: mask = (map != 0); ids0 = max(map,1)-1; scale = numel / sum(mask).  One block.
__global__ void k_mask_scale(const int64_t* __restrict__ lab, uint8_t* __restrict__ mask, int64_t* __restrict__ ids0,
                             float* __restrict__ scale, long n) {
    __shared__ unsigned long long s_red[16];
    unsigned long long cnt = 0;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
        int64_t l = lab[i];
        uint8_t m = l != 0;
        mask[i] = m;
        ids0[i] = (l > 1 ? l : 1) - 1;
        cnt += m;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long a = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) a += s_red[w];
        scale[0] = (float)n / (float)a;
    }
}
extern "C" int vqw_mask_scale(const int64_t* label_map, uint8_t* mask, int64_t* ids0, float* scale_dev, long n, void* stream) {
    VQW_CHECK(label_map && mask && ids0 && scale_dev && n > 0, "vqw_mask_scale: bad arguments");
    k_mask_scale<<<1, 1024, 0, (hipStream_t)stream>>>(label_map, mask, ids0, scale_dev, n);
    VQW_LAUNCH_CHECK("vqw_mask_scale");
    return VQW_OK;
}
