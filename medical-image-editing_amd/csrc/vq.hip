// Vector-quantisation kernels: nearest-codebook search, gather, commitment loss, EMA statistics.
// Replaces networks/vq/vq_module.py:45-62 (_torch_knn), :168-202 (_quantize), :204-207 (lookup) and
// networks/vq/grad_approximation.py:7-29 of the reference.  Rows are NHWC pixels: x[Npix][D].
//
// Score (same association order as the reference): s_k = ((2 * (e_k . x)) - |e_k|^2) - |x|^2 in fp32; arg-max over k
// with ties to the LOWEST index.  The K x N score matrix, the N x K one-hot and the D x N x K "embed_sum" GEMM of the
// reference are never materialised.
//
// Three search routes (vq_plan):
//   SMALL   codebook (+ norms) resident in LDS, query row in registers, scalar FMA chains over d ascending.  EMA
//           statistics  X^T . onehot  (vq_module.py:186) on the matrix cores: per wave a 64-pixel tile of X goes through
//           LDS as the A operand of v_mfma_f32_16x16x4_f32, the one-hot of the wave's ids is built in registers as B;
//           counts come from a row of ones appended to X.  Exact fp32, fixed order.
//   MFMA    any codebook that does not fit LDS (BASELINE config 4: K = 1024, D = 256): the score GEMM runs on
//           v_mfma_f32_32x32x2_f32 with codes on the M axis and pixels on the N axis, so one lane owns one pixel and
//           keeps its running (max, arg-max) in two registers across all code tiles; S never exists.
//   GENERIC odd D / very wide D: scalar kernel, codebook from LDS or global.
// EMA statistics of the MFMA / GENERIC routes: deterministic counting sort of the pixels by code (wave-private
// histograms, stable ranks) + segmented row sums in sorted order.  No float atomics anywhere.
#include "common.h"
#include "mfma_util.h"
#include "../../include/vqwnet_hip.h"

#define VQ_BLOCK 256
#ifndef VQ_EXP
#define VQ_EXP 0       // timing-only A/B builds (tools/vq_ab.sh): 1 = no arg-max epilogue, 2 = no stage refill / barrier, 4 = one MFMA chain per two blocks
#endif
#define VQ_LDS_FLOATS 15360  // 60 KiB of LDS for codebook + norms
#define VQ_MAX_D 1024

static inline int vq_blocks(long Npix) { return (int)imin(1024, ceil_div(Npix, VQ_BLOCK)); }   // one resident round (4 per CU)
static inline bool vq_lds_codebook(int D, int K) { return (long)K * D + K <= VQ_LDS_FLOATS; }
static inline int vq_pow2_ge(int v) { int p = 1; while (p < v) p <<= 1; return p; }
// matrix-core statistics in the small kernel: (D/16 + 1) x NT accumulator tiles of 16 x 16, NT = pow2 >= ceil(K/16)
static inline int vq_small_nt(int D, int K) {
    if (!(D == 16 || D == 32 || D == 64) || !vq_lds_codebook(D, K)) return 0;
    int nt = vq_pow2_ge(ceil_div(K, 16));
    return (D / 16 + 1) * nt <= 16 ? nt : 0;
}
static inline int vq_mfma_dt(int D) { return D <= 32 ? 32 : D <= 64 ? 64 : D <= 128 ? 128 : 256; }
static inline bool vq_use_mfma(int D, int K) { return (D % 4 == 0) && D >= 8 && D <= 256 && K >= 32; }

enum { VQ_PLAN_SMALL_MFMA_STATS = 0, VQ_PLAN_SMALL_SORTED = 1, VQ_PLAN_MFMA = 2, VQ_PLAN_GENERIC = 3 };
static inline int vq_plan(int D, int K) {
    if (vq_small_nt(D, K) > 0) return VQ_PLAN_SMALL_MFMA_STATS;
    if (vq_lds_codebook(D, K) && (D == 16 || D == 32 || D == 64)) return VQ_PLAN_SMALL_SORTED;
    if (vq_use_mfma(D, K)) return VQ_PLAN_MFMA;
    return VQ_PLAN_GENERIC;
}
extern "C" int vqw_vq_plan(int D, int K) { return vq_plan(D, K); }

// ---- workspace layout ------------------------------------------------------------------------------------------------
#define VQ_SB 2048          // pixels per sort block (one wave each)
#define VQ_CH 128           // sorted rows per segment-sum work item
static inline size_t al256(size_t b) { return (b + 255) / 256 * 256; }
struct VqWs {
    double* cpart;      // commit partials, one per workgroup
    float* enorm;       // |e_k|^2 (+inf padding) for the MFMA route
    float* spart;       // per-workgroup statistics rows of the SMALL route
    double* spart2;     // stage-one column sums (32 rows each)
    int *hist, *total, *base, *cstart, *lrank, *perm;
    float* partial;
    size_t bytes;
};
static inline long vq_ncpart(long Npix, int plan) { return plan == VQ_PLAN_MFMA ? (Npix + 127) / 128 : vq_blocks(Npix); }
static VqWs vq_carve(void* ws, long Npix, int D, int K) {
    const int plan = vq_plan(D, K);
    VqWs w;
    char* p = (char*)ws;
    size_t o = 0;
    w.cpart = (double*)(p + o); o += al256((size_t)vq_ncpart(Npix, plan) * sizeof(double));
    w.enorm = (float*)(p + o); o += al256(((size_t)K + 512) * sizeof(float));
    w.spart = (float*)(p + o);
    if (plan == VQ_PLAN_SMALL_MFMA_STATS) o += al256((size_t)vq_blocks(Npix) * K * (D + 1) * sizeof(float));
    w.spart2 = (double*)(p + o);
    if (plan == VQ_PLAN_SMALL_MFMA_STATS) o += al256((size_t)ceil_div(vq_blocks(Npix), 32) * K * (D + 1) * sizeof(double));
    const long nsb = (Npix + VQ_SB - 1) / VQ_SB;
    const long tmax = (Npix + VQ_CH - 1) / VQ_CH + K;
    w.hist = (int*)(p + o);
    if (plan != VQ_PLAN_SMALL_MFMA_STATS) {
        o += al256((size_t)nsb * K * sizeof(int));
        w.total = (int*)(p + o); o += al256((size_t)(K + 1) * sizeof(int));
        w.base = (int*)(p + o); o += al256((size_t)(K + 1) * sizeof(int));
        w.cstart = (int*)(p + o); o += al256((size_t)(K + 1) * sizeof(int));
        w.lrank = (int*)(p + o); o += al256((size_t)Npix * sizeof(int));
        w.perm = (int*)(p + o); o += al256((size_t)Npix * sizeof(int));
        w.partial = (float*)(p + o); o += al256((size_t)tmax * D * sizeof(float));
    } else {
        w.total = w.base = w.cstart = w.lrank = w.perm = nullptr;
        w.partial = nullptr;
    }
    w.bytes = o + 256;
    return w;
}
extern "C" size_t vqw_vq_ws_bytes(long Npix, int D, int K) { return vq_carve(nullptr, Npix, D, K).bytes; }

// ======================================================================================================================
// SMALL / GENERIC route
// ======================================================================================================================
// DT > 0: query row held in DT registers.  DT == 0: generic D (row re-read from global/L1).
// LDS_CB: codebook staged in LDS.  NT > 0: EMA statistics on the matrix cores (DT > 0 only), NT column tiles of 16 codes.
template <int DT, bool LDS_CB, int NT>
__global__ void __launch_bounds__(VQ_BLOCK) k_vq_fwd(const float* __restrict__ x, const float* __restrict__ embed,
                                                     int64_t* __restrict__ ids, float* __restrict__ q,
                                                     double* __restrict__ commit_part, float* __restrict__ stat_part,
                                                     long Npix, int D, int K, int want_stats, int id_base) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_cb = smem;                                   // [K][D] if LDS_CB
    float* s_nrm = smem + (LDS_CB ? K * D : 0);           // [K]
    float* s_x = s_nrm + ((K + 3) & ~3);                  // NT > 0: [waves][64][DT + 4] X tiles, later [waves][K*(D+1)] slabs
    __shared__ double s_red[VQ_BLOCK / 64];
    constexpr int MT = NT > 0 ? DT / 16 + 1 : 1;
    constexpr int XS = DT + 4;                            // padded tile row: conflict-free b128 stores, 2-way worst b32 reads
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (LDS_CB)
        for (int i = t; i < K * D; i += VQ_BLOCK) s_cb[i] = embed[i];
    for (int k = t; k < K; k += VQ_BLOCK) {
        float n2 = 0.f;
        for (int d = 0; d < D; ++d) { float e = embed[k * D + d]; n2 = fmaf(e, e, n2); }
        s_nrm[k] = n2;
    }
    __syncthreads();
    const float* cb = LDS_CB ? s_cb : embed;
    f32x4 acc[MT][NT > 0 ? NT : 1];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < (NT > 0 ? NT : 1); ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    double csum = 0.0;
    // wave-uniform trip count (the statistics tile is a wave-wide operation); lanes past the end compute on the last pixel
    // and contribute nothing
    float4 nxt[DT > 0 ? DT / 4 : 1];         // the next iteration's row is in flight while this one is scored
    if (DT > 0) {
        const long p0 = (long)blockIdx.x * VQ_BLOCK + t < Npix ? (long)blockIdx.x * VQ_BLOCK + t : Npix - 1;
#pragma unroll
        for (int d4 = 0; d4 < DT / 4; ++d4) nxt[d4] = ((const float4*)(x + p0 * D))[d4];
    }
    for (long base = (long)blockIdx.x * VQ_BLOCK; base < Npix; base += (long)gridDim.x * VQ_BLOCK) {
        const bool active = base + t < Npix;
        const long p = active ? base + t : Npix - 1;
        const float* xr = x + p * D;
        float xv[DT > 0 ? DT : 1];
        float x2 = 0.f;
        if (DT > 0) {
#pragma unroll
            for (int d4 = 0; d4 < DT / 4; ++d4) {
                const float4 v = nxt[d4];
                xv[4 * d4] = v.x; xv[4 * d4 + 1] = v.y; xv[4 * d4 + 2] = v.z; xv[4 * d4 + 3] = v.w;
            }
            const long nb_ = base + (long)gridDim.x * VQ_BLOCK;
            if (nb_ < Npix) {
                const long pn = nb_ + t < Npix ? nb_ + t : Npix - 1;
#pragma unroll
                for (int d4 = 0; d4 < DT / 4; ++d4) nxt[d4] = ((const float4*)(x + pn * D))[d4];
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
        if constexpr (NT > 0 && DT > 0) if (want_stats) {
            // sums[d][k] += sum_p X[p][d] * [id_p == k]  as  A (16 rows of d x 4 pixels) . B (4 pixels x 16 codes), 16 steps of
            // 4 pixels per 64-pixel tile; row d == D of A is all ones -> counts.  One accumulation chain per tile, in pixel order.
            float* xt = s_x + wv * 64 * XS;
#pragma unroll
            for (int d4 = 0; d4 < DT / 4; ++d4)
                *(float4*)(xt + lane * XS + 4 * d4) = float4{xv[4 * d4], xv[4 * d4 + 1], xv[4 * d4 + 2], xv[4 * d4 + 3]};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int code = active ? bi : -1;
            const int lm = lane & 15, lg = lane >> 4;
            const float ones = lm == 0 ? 1.f : 0.f;
#pragma unroll 4
            for (int s = 0; s < 16; ++s) {
                const int pid = __shfl(code, 4 * s + lg, 64);
                float bm[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) bm[j] = pid == j * 16 + lm ? 1.f : 0.f;
                const float* ar = xt + (4 * s + lg) * XS + lm;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const float a = i < MT - 1 ? ar[i * 16] : ones;
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = MFMA16(a, bm[j], acc[i][j]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    csum = wave_sum_d(csum);
    if (lane == 0) s_red[wv] = csum;
    __syncthreads();
    if (t == 0) {
        double a = 0.0;
        for (int w = 0; w < VQ_BLOCK / 64; ++w) a += s_red[w];
        commit_part[blockIdx.x] = a;
    }
    if constexpr (NT > 0) if (want_stats) {
        // accumulator tile (i, j), register r: row m = 16 i + 4 (lane / 16) + r, column n = 16 j + lane % 16
        const int KD1 = K * (D + 1);
        float* slab = s_x + wv * KD1;          // layout of `stats`: counts[K] then sums[d][k]
        const int lm = lane & 15, lg = lane >> 4;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 16 * i + 4 * lg + r, n = 16 * j + lm;
                    if (n < K) {
                        if (m < D) slab[K + m * K + n] = acc[i][j][r];
                        else if (m == D) slab[n] = acc[i][j][r];
                    }
                }
        __syncthreads();
        float* o = stat_part + (long)blockIdx.x * KD1;
        for (int i = t; i < KD1; i += VQ_BLOCK) {      // fold the wave slabs in wave order
            float a = s_x[i];
            for (int w = 1; w < VQ_BLOCK / 64; ++w) a += s_x[w * KD1 + i];
            o[i] = a;
        }
    }
}

// Tile form of the SMALL route (D in {16, 32, 64}, codebook in LDS): a wave owns 64 consecutive pixels at a time.  Their rows
// are loaded with fully coalesced float4 loads (lane i <-> float4 i of the 64 x D chunk; the next tile's loads are in
// flight while this one is scored), committed to a wave-private padded LDS tile, and every lane reads its own row back
// (conflict-free with the DT + 4 stride).  The same tile is the A operand of the statistics MFMAs, then receives the
// gathered codes and is written out with coalesced float4 stores.  The row-per-lane global accesses of k_vq_fwd (64-byte
// lane stride: 32 cache lines per wave instruction) held it at 3.6 TB/s.
template <int DT, int NT>
__global__ void __launch_bounds__(VQ_BLOCK) k_vq_tile(const float* __restrict__ x, const float* __restrict__ embed,
                                                      int64_t* __restrict__ ids, float* __restrict__ q,
                                                      double* __restrict__ commit_part, float* __restrict__ stat_part,
                                                      long Npix, int K, int want_stats, int id_base) {
    constexpr int D = DT, C4 = DT / 4, XS = DT + 4;
    constexpr int MT = NT > 0 ? DT / 16 + 1 : 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_cb = smem;                                   // [K][D]
    float* s_nrm = smem + K * D;                          // [K]
    float* s_x = s_nrm + ((K + 3) & ~3);                  // [waves][64][XS] tiles, later [waves][K*(D+1)] slabs
    __shared__ double s_red[VQ_BLOCK / 64];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (int i = t; i < K * D; i += VQ_BLOCK) s_cb[i] = embed[i];
    for (int k = t; k < K; k += VQ_BLOCK) {
        float n2 = 0.f;
        for (int d = 0; d < D; ++d) { const float e = embed[k * D + d]; n2 = fmaf(e, e, n2); }
        s_nrm[k] = n2;
    }
    __syncthreads();
    f32x4 acc[MT][NT > 0 ? NT : 1];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < (NT > 0 ? NT : 1); ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float* xt = s_x + wv * 64 * XS;
    const long ntiles = (Npix + 63) / 64, tstride = (long)gridDim.x * (VQ_BLOCK / 64);
    const long nf4 = Npix * C4;                           // float4 elements of x / q
    const float4* x4 = (const float4*)x;
    float4* q4 = (float4*)q;
    float4 pre[C4];
    long tile = (long)blockIdx.x * (VQ_BLOCK / 64) + wv;
    if (tile < ntiles) {
#pragma unroll
        for (int r = 0; r < C4; ++r) { const long f = tile * 64 * C4 + r * 64 + lane; pre[r] = f < nf4 ? x4[f] : float4{0.f, 0.f, 0.f, 0.f}; }
    }
    double csum = 0.0;
    for (; tile < ntiles; tile += tstride) {
#pragma unroll
        for (int r = 0; r < C4; ++r) {
            const int f = r * 64 + lane;
            *(float4*)(xt + (f / C4) * XS + 4 * (f % C4)) = pre[r];
        }
        const long nt_ = tile + tstride;
        if (nt_ < ntiles) {
#pragma unroll
            for (int r = 0; r < C4; ++r) { const long f = nt_ * 64 * C4 + r * 64 + lane; pre[r] = f < nf4 ? x4[f] : float4{0.f, 0.f, 0.f, 0.f}; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const long p = tile * 64 + lane;
        const bool active = p < Npix;
        float xv[DT];
#pragma unroll
        for (int d4 = 0; d4 < C4; ++d4) {
            const float4 v = *(const float4*)(xt + lane * XS + 4 * d4);
            xv[4 * d4] = v.x; xv[4 * d4 + 1] = v.y; xv[4 * d4 + 2] = v.z; xv[4 * d4 + 3] = v.w;
        }
        float x2 = 0.f;
#pragma unroll
        for (int d = 0; d < DT; ++d) x2 = fmaf(xv[d], xv[d], x2);
        float best = -INFINITY;
        int bi = 0;
        for (int k = 0; k < K; ++k) {
            const float* e = s_cb + k * D;
            float dot = 0.f;
#pragma unroll
            for (int d = 0; d < DT; ++d) dot = fmaf(e[d], xv[d], dot);
            const float sc = (2.f * dot - s_nrm[k]) - x2;
            if (sc > best) { best = sc; bi = k; }
        }
        if (active) ids[p] = (int64_t)(bi + id_base);
        if constexpr (NT > 0) if (want_stats) {
            // sums[d][k] += sum_p X[p][d] * [id_p == k]: A = 16 channels x 4 pixels from the tile, B = 4 pixels x 16 codes
            // (one-hot built in registers), 16 steps of 4 pixels; channel row D of A is all ones -> counts
            const int code = active ? bi : -1;
            const int lm = lane & 15, lg = lane >> 4;
            const float ones = lm == 0 ? 1.f : 0.f;
#pragma unroll 4
            for (int s = 0; s < 16; ++s) {
                const int pid = __shfl(code, 4 * s + lg, 64);
                float bm[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) bm[j] = pid == j * 16 + lm ? 1.f : 0.f;
                const float* ar = xt + (4 * s + lg) * XS + lm;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const float a = i < MT - 1 ? ar[i * 16] : ones;
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = MFMA16(a, bm[j], acc[i][j]);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float* e = s_cb + bi * D;                     // gather into the lane's own row of the tile
        float c = 0.f;
#pragma unroll
        for (int d4 = 0; d4 < C4; ++d4) {
            const float4 o = *(const float4*)(e + 4 * d4);
            *(float4*)(xt + lane * XS + 4 * d4) = o;
            const float a0 = xv[4 * d4] - o.x, a1 = xv[4 * d4 + 1] - o.y, a2 = xv[4 * d4 + 2] - o.z, a3 = xv[4 * d4 + 3] - o.w;
            c = fmaf(a0, a0, c); c = fmaf(a1, a1, c); c = fmaf(a2, a2, c); c = fmaf(a3, a3, c);
        }
        if (active) csum += (double)c;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < C4; ++r) {
            const int f = r * 64 + lane;
            const long gf = tile * 64 * C4 + f;
            const float4 v = *(const float4*)(xt + (f / C4) * XS + 4 * (f % C4));
            if (gf < nf4) q4[gf] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    csum = wave_sum_d(csum);
    if (lane == 0) s_red[wv] = csum;
    __syncthreads();
    if (t == 0) {
        double a = 0.0;
        for (int w = 0; w < VQ_BLOCK / 64; ++w) a += s_red[w];
        commit_part[blockIdx.x] = a;
    }
    if constexpr (NT > 0) if (want_stats) {
        const int KD1 = K * (D + 1);
        float* slab = s_x + wv * KD1;          // layout of `stats`: counts[K] then sums[d][k]
        const int lm = lane & 15, lg = lane >> 4;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 16 * i + 4 * lg + r, n = 16 * j + lm;
                    if (n < K) {
                        if (m < D) slab[K + m * K + n] = acc[i][j][r];
                        else if (m == D) slab[n] = acc[i][j][r];
                    }
                }
        __syncthreads();
        float* o = stat_part + (long)blockIdx.x * KD1;
        for (int i = t; i < KD1; i += VQ_BLOCK) {      // fold the wave slabs in wave order
            float a = s_x[i];
            for (int w = 1; w < VQ_BLOCK / 64; ++w) a += s_x[w * KD1 + i];
            o[i] = a;
        }
    }
}

// commit = sum(commit_part) / numel
__global__ void k_vq_finalize(const double* __restrict__ commit_part, long nblocks, float* __restrict__ commit, double inv_numel) {
    __shared__ double s_red[4];
    const int t = threadIdx.x;
    double a = 0.0;
    for (long i = t; i < nblocks; i += blockDim.x) a += commit_part[i];
    a = wave_sum_d(a);
    if ((t & 63) == 0) s_red[t >> 6] = a;
    __syncthreads();
    if (t == 0) commit[0] = (float)((s_red[0] + s_red[1] + s_red[2] + s_red[3]) * inv_numel);
}
// stats (SMALL route) = column sums of the per-workgroup rows, in double, fixed order, two stages: 32 rows per
// workgroup (a thread per column, coalesced), then the <= 32 stage-one rows
__global__ void __launch_bounds__(256) k_vq_stat_rows(const float* __restrict__ stat_part, int nrows, double* __restrict__ part2, int KD1) {
    const int r0 = blockIdx.x * 32, r1 = min(r0 + 32, nrows);
    for (int i = threadIdx.x; i < KD1; i += 256) {
        double a = 0.0;
        for (int r = r0; r < r1; ++r) a += (double)stat_part[(long)r * KD1 + i];
        part2[(long)blockIdx.x * KD1 + i] = a;
    }
}
__global__ void __launch_bounds__(256) k_vq_stat_final(const double* __restrict__ part2, int nrows2, double* __restrict__ stats, int KD1) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= KD1) return;
    double a = 0.0;
    for (int r = 0; r < nrows2; ++r) a += part2[(long)r * KD1 + i];
    stats[i] = a;
}

// ======================================================================================================================
// MFMA route: fused score GEMM + running arg-max
// ======================================================================================================================
// one wave per code: |e_k|^2 (fixed butterfly order); padding rows get +inf and can never win: (2*0 - inf) - |x|^2 = -inf
__global__ void __launch_bounds__(256) k_vq_enorm(const float* __restrict__ embed, float* __restrict__ enorm, int D, int K, int Kpad) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= Kpad) return;
    float n2 = 0.f;
    if (k < K)
        for (int d = lane; d < D; d += 64) { const float e = embed[(long)k * D + d]; n2 = fmaf(e, e, n2); }
    n2 = wave_sum_f(n2);
    if (lane == 0) enorm[k] = k < K ? n2 : INFINITY;
}

// One workgroup = 4 waves x 32 pixels.  A lane (n = lane % 32, h = lane / 32) keeps half of pixel n's row in DT/2
// registers: MFMA step 4j+i contracts the channel pair {8j+i (h=0), 8j+4+i (h=1)}, so both operands are float4 loads.
// Code tiles of TR = 32 * (256 / DT) rows stream through a double-buffered LDS stage (next tile prefetched into
// registers under the MFMAs, one barrier per stage).  Accumulator register r of a 32-code block holds code row
// 8 (r/4) + 4 h + r%4 for the lane's pixel: the arg-max is 16 compare/selects per 128 MFMAs, in ascending code order.
// FULL: D == DT (no channel padding: the per-chunk range predicates compile away).
template <int DT, bool FULL>
__global__ void __launch_bounds__(256, 2) k_vq_mfma(const float* __restrict__ x, const float* __restrict__ embed,
                                                    const float* __restrict__ enorm, int64_t* __restrict__ ids,
                                                    float* __restrict__ q, double* __restrict__ cpart, unsigned Npix, int D, int K,
                                                    int id_base) {
    constexpr int NBLK = 256 / DT, TR = 32 * NBLK, LS = DT + 4, STAGE = TR * LS + TR, C4 = DT / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ double s_red[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 31, h = lane >> 5;
    const unsigned p = blockIdx.x * 128u + wv * 32u + n;
    const bool pvalid = p < Npix;
    const unsigned OOB = 0xffffff00u;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, Npix * (unsigned)D * 4u);
    const __amdgpu_buffer_rsrc_t re = make_rsrc(embed, (unsigned)K * (unsigned)D * 4u);
    float xr[DT / 2];
    float x2 = 0.f;
#pragma unroll
    for (int j = 0; j < DT / 8; ++j) {
        const unsigned d0 = 8 * j + 4 * h;
        const float4 v = buf_ld4(rx, sel_u32(pvalid & (FULL || d0 < (unsigned)D), (p * (unsigned)D + d0) * 4u, OOB));
        xr[4 * j] = v.x; xr[4 * j + 1] = v.y; xr[4 * j + 2] = v.z; xr[4 * j + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < DT / 2; ++i) x2 = fmaf(xr[i], xr[i], x2);
    x2 += __shfl_xor(x2, 32, 64);

    // stage loader: 2048 float4 per tile, 8 per thread (thread -> column chunk c4 of rows r0 + i * RSTEP); rows past K and
    // columns past D read as zero (buffer range check)
    constexpr int RSTEP = 256 / C4;
    const int r0 = tid / C4, c4 = tid % C4;
    const bool cvalid = FULL || 4 * c4 < D;
    const unsigned goff0 = ((unsigned)r0 * (unsigned)D + 4u * c4) * 4u, gstep = (unsigned)RSTEP * (unsigned)D * 4u;
    const int loff0 = r0 * LS + 4 * c4;
    const int nstage = (K + TR - 1) / TR;
    const unsigned stage_bytes = (unsigned)TR * (unsigned)D * 4u;
    float4 pf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) pf[i] = buf_ld4(re, sel_u32(cvalid, goff0 + i * gstep, OOB));
#pragma unroll
    for (int i = 0; i < 8; ++i) *(float4*)(smem + loff0 + i * RSTEP * LS) = pf[i];
    if (tid < TR) smem[TR * LS + tid] = enorm[tid];
    __syncthreads();

    float best = -INFINITY;
    int bi = 0;
    for (int s = 0; s < nstage; ++s) {
        const float* buf = smem + ((VQ_EXP & 2) ? 0 : (s & 1)) * STAGE;
        float* nbuf = smem + ((s + 1) & 1) * STAGE;
        const bool more = s + 1 < nstage;
        if (more && !(VQ_EXP & 2)) {
#pragma unroll
            for (int i = 0; i < 8; ++i) pf[i] = buf_ld4(re, sel_u32(cvalid, goff0 + i * gstep + (unsigned)(s + 1) * stage_bytes, OOB));
        }
        const float en_next = (more && tid < TR) ? enorm[(s + 1) * TR + tid] : 0.f;
#pragma unroll 1
        for (int b = 0; b < NBLK; ++b) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* arow = buf + (b * 32 + n) * LS + 4 * h;
            // two fragment register sets: the LDS read of group j+1 is issued before the four MFMAs of group j
            float4 a0 = *(const float4*)arow, a1;
#pragma unroll
            for (int j = 0; j < DT / 8; j += 2) {
                a1 = *(const float4*)(arow + 8 * (j + 1));
                __builtin_amdgcn_sched_barrier(0);
                acc = MFMA32(a0.x, xr[4 * j], acc);
                acc = MFMA32(a0.y, xr[4 * j + 1], acc);
                acc = MFMA32(a0.z, xr[4 * j + 2], acc);
                acc = MFMA32(a0.w, xr[4 * j + 3], acc);
                __builtin_amdgcn_sched_barrier(0);
                if (j + 2 < DT / 8) a0 = *(const float4*)(arow + 8 * (j + 2));
                __builtin_amdgcn_sched_barrier(0);
                acc = MFMA32(a1.x, xr[4 * j + 4], acc);
                acc = MFMA32(a1.y, xr[4 * j + 5], acc);
                acc = MFMA32(a1.z, xr[4 * j + 6], acc);
                acc = MFMA32(a1.w, xr[4 * j + 7], acc);
                __builtin_amdgcn_sched_barrier(0);
            }
            const float* en = buf + TR * LS + b * 32 + 4 * h;
            const int code0 = s * TR + b * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < ((VQ_EXP & 1) ? 1 : 4); ++g) {
                const float4 e4 = *(const float4*)(en + 8 * g);
                const float s0 = (2.f * acc[4 * g] - e4.x) - x2, s1 = (2.f * acc[4 * g + 1] - e4.y) - x2;
                const float s2 = (2.f * acc[4 * g + 2] - e4.z) - x2, s3 = (2.f * acc[4 * g + 3] - e4.w) - x2;
                if (s0 > best) { best = s0; bi = code0 + 8 * g; }
                if (s1 > best) { best = s1; bi = code0 + 8 * g + 1; }
                if (s2 > best) { best = s2; bi = code0 + 8 * g + 2; }
                if (s3 > best) { best = s3; bi = code0 + 8 * g + 3; }
            }
        }
        if (more && !(VQ_EXP & 2)) {
#pragma unroll
            for (int i = 0; i < 8; ++i) *(float4*)(nbuf + loff0 + i * RSTEP * LS) = pf[i];
            if (tid < TR) nbuf[TR * LS + tid] = en_next;
        }
        if (!(VQ_EXP & 2)) __syncthreads();
    }
    {   // the two halves of a pixel hold disjoint code rows: keep the larger score, ties to the lower code
        const float ob = __shfl_xor(best, 32, 64);
        const int oi = __shfl_xor(bi, 32, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (pvalid && h == 0) ids[p] = (int64_t)(bi + id_base);
    float c = 0.f;
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(q, Npix * (unsigned)D * 4u);
#pragma unroll
    for (int j = 0; j < DT / 8; ++j) {
        const unsigned d0 = 8 * j + 4 * h;
        const bool dv = FULL || d0 < (unsigned)D;
        const float4 ev = buf_ld4(re, sel_u32(dv, ((unsigned)bi * (unsigned)D + d0) * 4u, OOB));
        buf_st4(rq, sel_u32(pvalid & dv, (p * (unsigned)D + d0) * 4u, OOB), ev);      // out-of-range offsets are dropped
        const float a0 = xr[4 * j] - ev.x, a1 = xr[4 * j + 1] - ev.y, a2 = xr[4 * j + 2] - ev.z, a3 = xr[4 * j + 3] - ev.w;
        c = fmaf(a0, a0, c); c = fmaf(a1, a1, c); c = fmaf(a2, a2, c); c = fmaf(a3, a3, c);
    }
    double cs = wave_sum_d(pvalid ? (double)c : 0.0);
    if (lane == 0) s_red[wv] = cs;
    __syncthreads();
    if (tid == 0) cpart[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// ======================================================================================================================
// deterministic EMA statistics for the MFMA / GENERIC routes: counting sort by code + segmented sums in sorted order
// ======================================================================================================================
// One wave per sort block of VQ_SB pixels: rank of every pixel among the pixels of its code inside the block (in pixel
// order) and the block's histogram.  Peers of a lane = lanes of the round with the same code (one ballot per code bit).
__global__ void __launch_bounds__(64) k_vq_rank(const int64_t* __restrict__ ids, int id_base, int* __restrict__ lrank,
                                                int* __restrict__ hist, long Npix, int K, int nbits) {
    extern __shared__ int s_cnt[];
    const int lane = threadIdx.x;
    for (int k = lane; k < K; k += 64) s_cnt[k] = 0;
    __syncthreads();
    const long p0 = (long)blockIdx.x * VQ_SB;
    const long pend = p0 + VQ_SB < Npix ? p0 + VQ_SB : Npix;
    for (long pb = p0; pb < pend; pb += 64) {
        const long p = pb + lane;
        const bool valid = p < pend;
        int code = valid ? (int)(ids[p] - id_base) : 0;
        code = code < 0 ? 0 : (code >= K ? K - 1 : code);
        unsigned long long peers = __ballot(valid);
        for (int b = 0; b < nbits; ++b) {
            const bool bit = (code >> b) & 1;
            const unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const int rin = __popcll(peers & ((1ull << lane) - 1ull));
        const int basec = s_cnt[code];
        __syncthreads();
        if (valid && rin == 0) s_cnt[code] = basec + __popcll(peers);
        __syncthreads();
        if (valid) lrank[p] = basec + rin;
    }
    for (int k = lane; k < K; k += 64) hist[(long)blockIdx.x * K + k] = s_cnt[k];
}

// per code: exclusive scan of the block histograms over the sort blocks (in place) and the code's total
__global__ void __launch_bounds__(256) k_vq_hscan(int* __restrict__ hist, int* __restrict__ total, int nsb, int K) {
    __shared__ int s_sum[4][64];
    const int t = threadIdx.x, kk = blockIdx.x * 64 + (t & 63), sl = t >> 6;
    const int lo = (int)((long)nsb * sl / 4), hi = (int)((long)nsb * (sl + 1) / 4);
    int sum = 0;
    if (kk < K)
        for (int sb = lo; sb < hi; ++sb) sum += hist[(long)sb * K + kk];
    s_sum[sl][t & 63] = sum;
    __syncthreads();
    int run = 0, tot = 0;
    for (int s = 0; s < 4; ++s) { const int v = s_sum[s][t & 63]; if (s < sl) run += v; tot += v; }
    if (kk < K) {
        for (int sb = lo; sb < hi; ++sb) { const long i = (long)sb * K + kk; const int v = hist[i]; hist[i] = run; run += v; }
        if (sl == 0) total[kk] = tot;
    }
}

// base[k] = first sorted position of code k; cstart[k] = first segment-sum work item of code k; counts -> stats[0..K)
__global__ void __launch_bounds__(1024) k_vq_bases(const int* __restrict__ total, int* __restrict__ base, int* __restrict__ cstart,
                                                   double* __restrict__ stats, int K) {
    __shared__ int s_a[1024], s_b[1024];
    const int t = threadIdx.x, it = (K + 1023) / 1024;
    int la = 0, lb = 0;
    for (int i = 0; i < it; ++i) { const int k = t * it + i; if (k < K) { la += total[k]; lb += (total[k] + VQ_CH - 1) / VQ_CH; } }
    s_a[t] = la; s_b[t] = lb;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {      // Hillis-Steele inclusive scan
        const int va = t >= o ? s_a[t - o] : 0, vb = t >= o ? s_b[t - o] : 0;
        __syncthreads();
        s_a[t] += va; s_b[t] += vb;
        __syncthreads();
    }
    int ra = s_a[t] - la, rb = s_b[t] - lb;
    for (int i = 0; i < it; ++i) {
        const int k = t * it + i;
        if (k < K) {
            base[k] = ra; cstart[k] = rb;
            stats[k] = (double)total[k];
            ra += total[k]; rb += (total[k] + VQ_CH - 1) / VQ_CH;
        }
    }
    if (t == 1023) { base[K] = s_a[1023]; cstart[K] = s_b[1023]; }
}

__global__ void k_vq_scatter(const int64_t* __restrict__ ids, int id_base, const int* __restrict__ lrank,
                             const int* __restrict__ hist, const int* __restrict__ base, int* __restrict__ perm, long Npix, int K) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < Npix; p += stride) {
        int code = (int)(ids[p] - id_base);
        code = code < 0 ? 0 : (code >= K ? K - 1 : code);
        perm[base[code] + hist[(p / VQ_SB) * K + code] + lrank[p]] = (int)p;
    }
}

// One wave per work item: the sum, in sorted order, of up to VQ_CH member rows of one code.  V4: D % 4 == 0.
template <bool V4>
__global__ void __launch_bounds__(256) k_vq_segsum(const float* __restrict__ x, const int* __restrict__ perm,
                                                   const int* __restrict__ total, const int* __restrict__ base,
                                                   const int* __restrict__ cstart, float* __restrict__ partial, int D, int K) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= cstart[K]) return;
    int lo = 0, hi = K;                       // largest k with cstart[k] <= w and a non-empty range
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (cstart[mid] <= w) lo = mid; else hi = mid; }
    const int k = lo, j = w - cstart[k];
    const int r0 = base[k] + j * VQ_CH;
    const int cnt = min(VQ_CH, base[k] + total[k] - r0);
    constexpr int NA = V4 ? VQ_MAX_D / 256 : VQ_MAX_D / 64;
    float4 a4[V4 ? NA : 1];
    float a1[V4 ? 1 : NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) { if constexpr (V4) a4[i] = float4{0.f, 0.f, 0.f, 0.f}; else a1[i] = 0.f; }
    for (int rb = 0; rb < cnt; rb += 64) {
        const int mine = rb + lane < cnt ? perm[r0 + rb + lane] : 0;
        const int nr = min(64, cnt - rb);
        int r = 0;
        for (; r + 4 <= nr; r += 4) {          // four rows in flight, added in sorted order
            const float* row[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) row[u] = x + (size_t)__shfl(mine, r + u, 64) * D;
            if constexpr (V4) {
                float4 v[4][NA];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < NA; ++i) {
                        const int d = 4 * (lane + 64 * i);
                        v[u][i] = d < D ? *(const float4*)(row[u] + d) : float4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < NA; ++i) { a4[i].x += v[u][i].x; a4[i].y += v[u][i].y; a4[i].z += v[u][i].z; a4[i].w += v[u][i].w; }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < NA; ++i) { const int d = lane + 64 * i; if (d < D) a1[i] += row[u][d]; }
            }
        }
        for (; r < nr; ++r) {
            const float* row = x + (size_t)__shfl(mine, r, 64) * D;
            if constexpr (V4) {
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const int d = 4 * (lane + 64 * i);
                    if (d < D) { const float4 v = *(const float4*)(row + d); a4[i].x += v.x; a4[i].y += v.y; a4[i].z += v.z; a4[i].w += v.w; }
                }
            } else {
#pragma unroll
                for (int i = 0; i < NA; ++i) { const int d = lane + 64 * i; if (d < D) a1[i] += row[d]; }
            }
        }
    }
    float* o = partial + (size_t)w * D;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        if constexpr (V4) { const int d = 4 * (lane + 64 * i); if (d < D) *(float4*)(o + d) = a4[i]; }
        else { const int d = lane + 64 * i; if (d < D) o[d] = a1[i]; }
    }
}

// One workgroup per code: its work-item partials summed in order (16 waves take contiguous sixteenths, folded in wave
// order) -> stats[K + d*K + k] (the reference's embed_sum layout, as double)
__global__ void __launch_bounds__(1024) k_vq_segfinal(const float* __restrict__ partial, const int* __restrict__ cstart,
                                                      double* __restrict__ stats, int D, int K) {
    extern __shared__ float s_p[];           // [16][D]
    const int k = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c0 = cstart[k], np = cstart[k + 1] - c0;
    const int lo = (int)((long)np * wv / 16), hi = (int)((long)np * (wv + 1) / 16);
    for (int d = lane; d < D; d += 64) {
        float a = 0.f;
#pragma unroll 4
        for (int j = lo; j < hi; ++j) a += partial[(size_t)(c0 + j) * D + d];
        s_p[wv * D + d] = a;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 1024) {
        float a = s_p[d];
        for (int w = 1; w < 16; ++w) a += s_p[w * D + d];
        stats[K + (size_t)d * K + k] = (double)a;
    }
}

static int vq_sorted_stats(const float* x, const int64_t* ids, int id_base, double* stats, const VqWs& w, long Npix, int D, int K,
                           hipStream_t st) {
    const int nsb = (int)((Npix + VQ_SB - 1) / VQ_SB);
    int nbits = 0;
    while ((1 << nbits) < K) ++nbits;
    k_vq_rank<<<nsb, 64, (size_t)K * sizeof(int), st>>>(ids, id_base, w.lrank, w.hist, Npix, K, nbits);
    k_vq_hscan<<<ceil_div(K, 64), 256, 0, st>>>(w.hist, w.total, nsb, K);
    k_vq_bases<<<1, 1024, 0, st>>>(w.total, w.base, w.cstart, stats, K);
    k_vq_scatter<<<stream_grid(Npix, 256), 256, 0, st>>>(ids, id_base, w.lrank, w.hist, w.base, w.perm, Npix, K);
    const long tmax = (Npix + VQ_CH - 1) / VQ_CH + K;
    if (D % 4 == 0) k_vq_segsum<true><<<(unsigned)((tmax + 3) / 4), 256, 0, st>>>(x, w.perm, w.total, w.base, w.cstart, w.partial, D, K);
    else k_vq_segsum<false><<<(unsigned)((tmax + 3) / 4), 256, 0, st>>>(x, w.perm, w.total, w.base, w.cstart, w.partial, D, K);
    k_vq_segfinal<<<K, 1024, (size_t)16 * D * sizeof(float), st>>>(w.partial, w.cstart, stats, D, K);
    VQW_LAUNCH_CHECK("vqw_vq_fwd(sorted statistics)");
    return VQW_OK;
}

template <int DT, bool FULL>
static int launch_vq_mfma(const float* x, const float* embed, const float* enorm, int64_t* ids, float* q, double* cpart, long Npix,
                          int D, int K, int id_base, hipStream_t st) {
    constexpr int NBLK = 256 / DT, TR = 32 * NBLK, LS = DT + 4, STAGE = TR * LS + TR;
    const size_t lds = 2 * (size_t)STAGE * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)k_vq_mfma<DT, FULL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            vqw_set_error("vqw_vq_fwd: cannot reserve %zu bytes of LDS", lds);
            return VQW_ERR_HIP;
        }
        attr_done = true;
    }
    // 32-bit buffer offsets: pixel chunks whose rows stay below 4 GiB (a multiple of the 128-pixel workgroup tile)
    const long pch = (long)((0xfffffe00ull / ((unsigned long long)D * 4ull)) / 128ull) * 128;
    for (long p0 = 0; p0 < Npix; p0 += pch) {
        const long pn = Npix - p0 < pch ? Npix - p0 : pch;
        k_vq_mfma<DT, FULL><<<(unsigned)((pn + 127) / 128), 256, lds, st>>>(x + p0 * D, embed, enorm, ids + p0, q + p0 * D, cpart + p0 / 128,
                                                                      (unsigned)pn, D, K, id_base);
    }
    return VQW_OK;
}

template <int DT, int NT>
static int launch_vq_tile_nt(int nb, size_t lds, hipStream_t st, const float* x, const float* embed, int64_t* ids, float* q,
                             double* cpart, float* spart, long Npix, int K, int want, int id_base) {
    if (lds > 65536 && hipFuncSetAttribute((const void*)k_vq_tile<DT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        vqw_set_error("vqw_vq_fwd: cannot reserve %zu bytes of LDS", lds);      // D = 64 tiles next to a large codebook
        return VQW_ERR_HIP;
    }
    k_vq_tile<DT, NT><<<nb, VQ_BLOCK, lds, st>>>(x, embed, ids, q, cpart, spart, Npix, K, want, id_base);
    return VQW_OK;
}
template <int DT>
static int launch_vq_tile(int nt, int nb, size_t lds, hipStream_t st, const float* x, const float* embed, int64_t* ids, float* q,
                          double* cpart, float* spart, long Npix, int K, int want, int id_base) {
    if (nt == 1) return launch_vq_tile_nt<DT, 1>(nb, lds, st, x, embed, ids, q, cpart, spart, Npix, K, want, id_base);
    if (nt == 2) return launch_vq_tile_nt<DT, 2>(nb, lds, st, x, embed, ids, q, cpart, spart, Npix, K, want, id_base);
    if (DT <= 32 && nt == 4) return launch_vq_tile_nt<DT, (DT <= 32 ? 4 : 0)>(nb, lds, st, x, embed, ids, q, cpart, spart, Npix, K, want, id_base);
    if (DT == 16 && nt == 8) return launch_vq_tile_nt<DT, (DT == 16 ? 8 : 0)>(nb, lds, st, x, embed, ids, q, cpart, spart, Npix, K, want, id_base);
    return launch_vq_tile_nt<DT, 0>(nb, lds, st, x, embed, ids, q, cpart, spart, Npix, K, want, id_base);
}

extern "C" int vqw_vq_fwd(const float* x, const float* embed, int64_t* ids, int id_base, float* q, float* commit, double* stats,
                          void* ws, size_t ws_bytes, long Npix, int D, int K, void* stream) {
    VQW_CHECK(x && embed && ids && q && commit && ws && Npix > 0 && D > 0 && K > 0, "vqw_vq_fwd: bad arguments");
    VQW_CHECK(ws_bytes >= vqw_vq_ws_bytes(Npix, D, K), "vqw_vq_fwd: workspace too small");
    VQW_CHECK(K <= 8192, "vqw_vq_fwd: dict_size %d exceeds the supported 8192", K);
    VQW_CHECK(D <= VQ_MAX_D, "vqw_vq_fwd: emb_dim %d exceeds the supported %d", D, VQ_MAX_D);
    VQW_CHECK(Npix < (1L << 31), "vqw_vq_fwd: %ld pixels per call exceed the supported 2^31", Npix);
    hipStream_t st = (hipStream_t)stream;
    VQW_CHECK((((uintptr_t)x | (uintptr_t)q | (uintptr_t)embed) & 15) == 0, "vqw_vq_fwd: x, q and embed must be 16-byte aligned");
    const int plan = vq_plan(D, K);
    const VqWs w = vq_carve(ws, Npix, D, K);
    const int want = stats != nullptr;
    const double inv_numel = 1.0 / ((double)Npix * D);
    if (plan == VQ_PLAN_MFMA) {
        const int dt = vq_mfma_dt(D);
        const int tr = 32 * (256 / dt);
        const int kpad = (K + tr - 1) / tr * tr;
        k_vq_enorm<<<ceil_div(kpad, 4), 256, 0, st>>>(embed, w.enorm, D, K, kpad);
        int rc;
#define VQ_MFMA_CASE(DT_) (D == DT_ ? launch_vq_mfma<DT_, true>(x, embed, w.enorm, ids, q, w.cpart, Npix, D, K, id_base, st) \
                                    : launch_vq_mfma<DT_, false>(x, embed, w.enorm, ids, q, w.cpart, Npix, D, K, id_base, st))
        if (dt == 32) rc = VQ_MFMA_CASE(32);
        else if (dt == 64) rc = VQ_MFMA_CASE(64);
        else if (dt == 128) rc = VQ_MFMA_CASE(128);
        else rc = VQ_MFMA_CASE(256);
#undef VQ_MFMA_CASE
        if (rc) return rc;
        VQW_LAUNCH_CHECK("vqw_vq_fwd(mfma)");
        if (want) { rc = vq_sorted_stats(x, ids, id_base, stats, w, Npix, D, K, st); if (rc) return rc; }
        k_vq_finalize<<<1, 256, 0, st>>>(w.cpart, vq_ncpart(Npix, plan), commit, inv_numel);
        VQW_LAUNCH_CHECK("vqw_vq_finalize");
        return VQW_OK;
    }
    const int nb = vq_blocks(Npix);
    const int KD1 = K * (D + 1);
    const bool lds_cb = vq_lds_codebook(D, K);
    const bool tiled = lds_cb && (D == 16 || D == 32 || D == 64);
    const int nt = plan == VQ_PLAN_SMALL_MFMA_STATS ? vq_small_nt(D, K) : 0;
    size_t lds_floats = (size_t)((K + 3) & ~3) + (lds_cb ? (size_t)K * D : 0);
    if (tiled) {
        const size_t tiles = (size_t)(VQ_BLOCK / 64) * 64 * (D + 4), slabs = nt > 0 ? (size_t)(VQ_BLOCK / 64) * KD1 : 0;
        lds_floats += tiles > slabs ? tiles : slabs;
    }
    const size_t lb = lds_floats * sizeof(float);
    int rc = VQW_OK;
    if (tiled && D == 16) rc = launch_vq_tile<16>(nt, nb, lb, st, x, embed, ids, q, w.cpart, w.spart, Npix, K, want, id_base);
    else if (tiled && D == 32) rc = launch_vq_tile<32>(nt, nb, lb, st, x, embed, ids, q, w.cpart, w.spart, Npix, K, want, id_base);
    else if (tiled) rc = launch_vq_tile<64>(nt, nb, lb, st, x, embed, ids, q, w.cpart, w.spart, Npix, K, want, id_base);
    else if (lds_cb) k_vq_fwd<0, true, 0><<<nb, VQ_BLOCK, lb, st>>>(x, embed, ids, q, w.cpart, w.spart, Npix, D, K, want, id_base);
    else k_vq_fwd<0, false, 0><<<nb, VQ_BLOCK, lb, st>>>(x, embed, ids, q, w.cpart, w.spart, Npix, D, K, want, id_base);
    if (rc) return rc;
    VQW_LAUNCH_CHECK("vqw_vq_fwd");
    if (want && nt == 0) { int rc = vq_sorted_stats(x, ids, id_base, stats, w, Npix, D, K, st); if (rc) return rc; }
    k_vq_finalize<<<1, 256, 0, st>>>(w.cpart, nb, commit, inv_numel);
    if (want && nt > 0) {
        const int nr2 = ceil_div(nb, 32);
        k_vq_stat_rows<<<nr2, 256, 0, st>>>(w.spart, nb, w.spart2, KD1);
        k_vq_stat_final<<<ceil_div(KD1, 256), 256, 0, st>>>(w.spart2, nr2, stats, KD1);
    }
    VQW_LAUNCH_CHECK("vqw_vq_finalize");
    return VQW_OK;
}

// EMA (vq_module.py:132-136,195-196) + Laplace-smoothed renormalisation (:198-200).  One workgroup per 32 (d) x 32 (k)
// tile of embed_avg; every workgroup derives n = sum_k cluster_size_new[k] itself from the OLD cluster sizes and the
// counts (same order everywhere), k_vq_ema_cs then writes the new cluster sizes.
__global__ void __launch_bounds__(256) k_vq_ema(const double* __restrict__ stats, float* __restrict__ embed, const float* __restrict__ cs,
                                                float* __restrict__ ea, float m, float eps, float sum_scale, int D, int K) {
    __shared__ double s_red[4];
    __shared__ float s_n;
    __shared__ float s_t[32][33];
    const int t = threadIdx.x;
    const float om = 1.f - m;
    double part = 0.0;
    for (int k = t; k < K; k += 256) part += (double)(cs[k] * m + om * (float)stats[k]);
    part = wave_sum_d(part);
    if ((t & 63) == 0) s_red[t >> 6] = part;
    __syncthreads();
    if (t == 0) s_n = (float)(s_red[0] + s_red[1] + s_red[2] + s_red[3]);
    __syncthreads();
    const float n = s_n;
    const int tk = (K + 31) / 32;
    const int d0 = (blockIdx.x / tk) * 32, k0 = (blockIdx.x % tk) * 32;
    const int tx = t & 31, ty = t >> 5;              // 32 x 8 threads, 4 rows each
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int d = d0 + ty + 8 * r, k = k0 + tx;
        if (d < D && k < K) {
            const long i = (long)d * K + k;
            const float a = ea[i] * m + om * ((float)stats[K + i] * sum_scale);
            ea[i] = a;
            const float c = cs[k] * m + om * (float)stats[k];
            const float csn = n * (c + eps) / (n + (float)K * eps);
            s_t[ty + 8 * r][tx] = a / csn;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = k0 + ty + 8 * r, d = d0 + tx;
        if (d < D && k < K) embed[(long)k * D + d] = s_t[tx][ty + 8 * r];
    }
}
__global__ void k_vq_ema_cs(const double* __restrict__ stats, float* __restrict__ cs, float m, int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) cs[k] = cs[k] * m + (1.f - m) * (float)stats[k];
}
extern "C" int vqw_vq_ema_update(const double* stats, float* embed, float* cluster_size, float* embed_avg, float momentum,
                                 float eps, float sum_scale, int D, int K, void* stream) {
    VQW_CHECK(stats && embed && cluster_size && embed_avg && D > 0 && K > 0, "vqw_vq_ema_update: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    k_vq_ema<<<ceil_div(D, 32) * ceil_div(K, 32), 256, 0, st>>>(stats, embed, cluster_size, embed_avg, momentum, eps, sum_scale, D, K);
    k_vq_ema_cs<<<ceil_div(K, 256), 256, 0, st>>>(stats, cluster_size, momentum, K);
    VQW_LAUNCH_CHECK("vqw_vq_ema_update");
    return VQW_OK;
}

// One Lloyd update of the k-means codebook initialisation (unet_encoder.py:66-91; kmeans_pytorch's loop): centre k
// becomes the mean of its members (statistics of vqw_vq_fwd); a code without members keeps its centre (kmeans_pytorch
// 0.3.0 writes NaN there).  shift[0] = sum_k |new_k - old_k|_2 (its convergence measure), shift[1] = #empty codes.
__global__ void __launch_bounds__(256) k_kmeans_update(const double* __restrict__ stats, float* __restrict__ centres,
                                                       double* __restrict__ part, int D, int K) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= K) return;
    const double cnt = stats[k];
    double d2 = 0.0;
    if (cnt > 0.0)
        for (int d = lane; d < D; d += 64) {
            const float nv = (float)(stats[K + (size_t)d * K + k] / cnt), ov = centres[(size_t)k * D + d];
            centres[(size_t)k * D + d] = nv;
            d2 += ((double)nv - ov) * ((double)nv - ov);
        }
    d2 = wave_sum_d(d2);
    if (lane == 0) { part[2 * k] = sqrt(d2); part[2 * k + 1] = cnt > 0.0 ? 0.0 : 1.0; }
}
__global__ void k_kmeans_shift(const double* __restrict__ part, double* __restrict__ shift, int K) {
    __shared__ double sa[256], sb[256];
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < K; k += 256) { a += part[2 * k]; b += part[2 * k + 1]; }
    sa[threadIdx.x] = a; sb[threadIdx.x] = b;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) { sa[threadIdx.x] += sa[threadIdx.x + w]; sb[threadIdx.x] += sb[threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { shift[0] = sa[0]; shift[1] = sb[0]; }
}
extern "C" int vqw_kmeans_update(const double* stats, float* centres, double* shift, void* ws, size_t ws_bytes, int D, int K,
                                 void* stream) {
    VQW_CHECK(stats && centres && shift && ws && D > 0 && K > 0, "vqw_kmeans_update: bad arguments");
    VQW_CHECK(ws_bytes >= (size_t)2 * K * sizeof(double), "vqw_kmeans_update: workspace too small (needs 16 K bytes)");
    hipStream_t st = (hipStream_t)stream;
    k_kmeans_update<<<ceil_div(K, 4), 256, 0, st>>>(stats, centres, (double*)ws, D, K);
    k_kmeans_shift<<<1, 256, 0, st>>>((const double*)ws, shift, K);
    VQW_LAUNCH_CHECK("vqw_kmeans_update");
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
