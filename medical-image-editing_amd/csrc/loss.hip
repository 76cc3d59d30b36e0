// Embedding (cross-view cluster consistency) loss, codebook margin / norm losses, one-hot and label flip.
// Replaces functions/embed_loss.py:22-88 and functions/onehot.py:11-20 of the reference without ever building
// the (b, D, K, n_loc) broadcast or the N x (K+1) one-hot.
#include "common.h"
#include "../../include/vqwnet_hip.h"

#define CL_BLOCK 256
#define CL_MAX_SPLITS 64
#define CL_EPS 1e-6f  // EmbeddingLoss.epsilon (embed_loss.py:8)

static inline int cl_splits(int B, long HW) {
    int s = ceil_div(1024, B);
    long cap = HW / 512 > 1 ? HW / 512 : 1;
    if (s > cap) s = (int)cap;
    if (s > CL_MAX_SPLITS) s = CL_MAX_SPLITS;
    return s < 1 ? 1 : s;
}
extern "C" size_t vqw_cross_ws_bytes(int B, int K, long HW) {
    (void)HW;
    return (size_t)B * CL_MAX_SPLITS * K * 2 * sizeof(float);
}

// labels variant.  part[b][split][k][2] = (sum of squared distances, count)
__global__ void __launch_bounds__(CL_BLOCK) k_cross_partial(const float* __restrict__ embed, const int32_t* __restrict__ labels,
                                                            const float* __restrict__ cb, float* __restrict__ part, long HW,
                                                            int D, int K, int splits, int cb_lds) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_acc = smem;  // [waves][K][2]: one slab per wave, single writer, folded in wave order (deterministic)
    const int b = blockIdx.y, s = blockIdx.x, t = threadIdx.x;
    constexpr int NWV = CL_BLOCK / 64;
    for (int i = t; i < NWV * 2 * K; i += CL_BLOCK) s_acc[i] = 0.f;
    // a small codebook (cb_lds: K * D floats behind the slabs) is read from LDS, the pixel's row with 16-byte loads: the same
    // differences summed in the same order as the scalar walk below
    float* s_cb = s_acc + NWV * 2 * K;
    if (cb_lds)
        for (int i = t; i < K * D; i += CL_BLOCK) s_cb[i] = cb[i];
    __syncthreads();
    float* slab = s_acc + (t >> 6) * 2 * K;
    long per = (HW + splits - 1) / splits;
    long p0 = s * per, p1 = p0 + per < HW ? p0 + per : HW;
    for (long base = p0; base < p1; base += CL_BLOCK) {            // wave-uniform trip count (shuffles below)
        const long p = base + t;
        int l = p < p1 ? labels[(long)b * HW + p] : 0;
        float d2 = 0.f;
        if (l >= 1 && l <= K && cb_lds) {
            const float4* e4 = (const float4*)(embed + ((long)b * HW + p) * D);
            const float* c = s_cb + (l - 1) * D;
            for (int d = 0; d < D; d += 4) {
                const float4 ev = e4[d >> 2];
                float a = ev.x - c[d]; d2 = fmaf(a, a, d2);
                a = ev.y - c[d + 1]; d2 = fmaf(a, a, d2);
                a = ev.z - c[d + 2]; d2 = fmaf(a, a, d2);
                a = ev.w - c[d + 3]; d2 = fmaf(a, a, d2);
            }
        } else if (l >= 1 && l <= K) {
            const float* e = embed + ((long)b * HW + p) * D;
            const float* c = cb + (long)(l - 1) * D;
            for (int d = 0; d < D; ++d) { float a = e[d] - c[d]; d2 = fmaf(a, a, d2); }
        } else {
            l = 0;
        }
        for (int k = 1; k <= K; ++k) {
            const unsigned long long members = __ballot(l == k);
            if (members == 0ull) continue;
            float v = wave_sum_f(l == k ? d2 : 0.f);
            if ((t & 63) == 0) {
                slab[2 * (k - 1)] += v;
                slab[2 * (k - 1) + 1] += (float)__popcll(members);
            }
        }
    }
    __syncthreads();
    float* o = part + ((long)b * splits + s) * K * 2;
    for (int i = t; i < 2 * K; i += CL_BLOCK) {
        float a = s_acc[i];
        for (int w = 1; w < NWV; ++w) a += s_acc[w * 2 * K + i];
        o[i] = a;
    }
}

// dense variant: r[b][k][p] (NCHW float weights)
__global__ void __launch_bounds__(CL_BLOCK) k_cross_partial_dense(const float* __restrict__ embed, const float* __restrict__ r,
                                                                  const float* __restrict__ cb, float* __restrict__ part,
                                                                  long HW, int D, int K, int splits) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_acc = smem;  // [waves][K][2], as above
    const int b = blockIdx.y, s = blockIdx.x, t = threadIdx.x;
    constexpr int NWV = CL_BLOCK / 64;
    for (int i = t; i < NWV * 2 * K; i += CL_BLOCK) s_acc[i] = 0.f;
    __syncthreads();
    float* slab = s_acc + (t >> 6) * 2 * K;
    long per = (HW + splits - 1) / splits;
    long p0 = s * per, p1 = p0 + per < HW ? p0 + per : HW;
    for (long base = p0; base < p1; base += CL_BLOCK) {
        const long p = base + t;
        const bool active = p < p1;
        const float* e = embed + ((long)b * HW + (active ? p : p0)) * D;
        for (int k = 0; k < K; ++k) {
            float w = active ? r[((long)b * K + k) * HW + p] : 0.f;
            float num = 0.f;
            if (w != 0.f) {
                const float* c = cb + (long)k * D;
                float d2 = 0.f;
                for (int d = 0; d < D; ++d) { float a = e[d] - c[d]; d2 = fmaf(a, a, d2); }
                num = d2 * w;
            }
            if (__ballot(w != 0.f) == 0ull) continue;
            float sn = wave_sum_f(num), sw = wave_sum_f(w);
            if ((t & 63) == 0) {
                slab[2 * k] += sn;
                slab[2 * k + 1] += sw;
            }
        }
    }
    __syncthreads();
    float* o = part + ((long)b * splits + s) * K * 2;
    for (int i = t; i < 2 * K; i += CL_BLOCK) {
        float a = s_acc[i];
        for (int w = 1; w < NWV; ++w) a += s_acc[w * 2 * K + i];
        o[i] = a;
    }
}

// loss = mean over (b,k) with cnt != 0 of num/(cnt+eps); coef[b][k] = 1/((cnt+eps) * n_present) or 0.
__global__ void k_cross_finalize(const float* __restrict__ part, float* __restrict__ loss, float* __restrict__ coef, int BK,
                                 int K, int splits) {
    __shared__ double s_l[4];
    __shared__ int s_c[4];
    __shared__ int s_np;
    double lsum = 0.0;
    int np = 0;
    for (int i = threadIdx.x; i < BK; i += blockDim.x) {
        int b = i / K, k = i % K;
        double num = 0.0, cnt = 0.0;
        for (int s = 0; s < splits; ++s) {
            const float* o = part + (((long)b * splits + s) * K + k) * 2;
            num += (double)o[0];
            cnt += (double)o[1];
        }
        if (cnt != 0.0) {
            float per = (float)num / ((float)cnt + CL_EPS);
            lsum += (double)per;
            np += 1;
            coef[i] = 1.f / ((float)cnt + CL_EPS);
        } else {
            coef[i] = 0.f;
        }
    }
    lsum = wave_sum_d(lsum);
    for (int o = 32; o > 0; o >>= 1) np += __shfl_xor(np, o, 64);
    if ((threadIdx.x & 63) == 0) { s_l[threadIdx.x >> 6] = lsum; s_c[threadIdx.x >> 6] = np; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int n = s_c[0] + s_c[1] + s_c[2] + s_c[3];
        s_np = n;
        // torch: mean of an empty selection is NaN
        loss[0] = n > 0 ? (float)((s_l[0] + s_l[1] + s_l[2] + s_l[3]) / (double)n) : __builtin_nanf("");
    }
    __syncthreads();
    float inv = s_np > 0 ? 1.f / (float)s_np : 0.f;
    for (int i = threadIdx.x; i < BK; i += blockDim.x) coef[i] *= inv;
}

extern "C" int vqw_cross_loss_fwd(const float* embed, const int32_t* labels, const float* codebook_kd, float* loss, float* coef,
                                  void* ws, size_t ws_bytes, int B, long HW, int D, int K, void* stream) {
    VQW_CHECK(embed && labels && codebook_kd && loss && coef && ws && B > 0 && HW > 0 && D > 0 && K > 0,
              "vqw_cross_loss_fwd: bad arguments");
    VQW_CHECK(ws_bytes >= vqw_cross_ws_bytes(B, K, HW), "vqw_cross_loss_fwd: workspace too small");
    VQW_CHECK(K <= 8192, "vqw_cross_loss_fwd: K too large");
    hipStream_t st = (hipStream_t)stream;
    int splits = cl_splits(B, HW);
    const int cb_lds = (D % 4 == 0 && (long)K * D <= 4096 && (((uintptr_t)embed) & 15) == 0) ? 1 : 0;
    k_cross_partial<<<dim3(splits, B), CL_BLOCK, ((CL_BLOCK / 64) * 2 * K + (cb_lds ? K * D : 0)) * sizeof(float), st>>>(embed, labels, codebook_kd, (float*)ws, HW, D, K, splits, cb_lds);
    k_cross_finalize<<<1, 256, 0, st>>>((const float*)ws, loss, coef, B * K, K, splits);
    VQW_LAUNCH_CHECK("vqw_cross_loss_fwd");
    return VQW_OK;
}
extern "C" int vqw_cross_loss_dense_fwd(const float* embed, const float* r_nchw, const float* codebook_kd, float* loss,
                                        float* coef, void* ws, size_t ws_bytes, int B, long HW, int D, int K, void* stream) {
    VQW_CHECK(embed && r_nchw && codebook_kd && loss && coef && ws && B > 0 && HW > 0 && D > 0 && K > 0,
              "vqw_cross_loss_dense_fwd: bad arguments");
    VQW_CHECK(ws_bytes >= vqw_cross_ws_bytes(B, K, HW), "vqw_cross_loss_dense_fwd: workspace too small");
    VQW_CHECK(K <= 8192, "vqw_cross_loss_dense_fwd: K too large");
    hipStream_t st = (hipStream_t)stream;
    int splits = cl_splits(B, HW);
    k_cross_partial_dense<<<dim3(splits, B), CL_BLOCK, (CL_BLOCK / 64) * 2 * K * sizeof(float), st>>>(embed, r_nchw, codebook_kd, (float*)ws, HW, D, K, splits);
    k_cross_finalize<<<1, 256, 0, st>>>((const float*)ws, loss, coef, B * K, K, splits);
    VQW_LAUNCH_CHECK("vqw_cross_loss_dense_fwd");
    return VQW_OK;
}

__global__ void k_cross_bwd(const float* __restrict__ embed, const int32_t* __restrict__ labels, const float* __restrict__ cb,
                            const float* __restrict__ coef, const float* __restrict__ gl, float* __restrict__ ge, long total,
                            long HW, int D, int K) {
    float g2 = 2.f * gl[0];
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        long p = i / D;
        int d = (int)(i % D);
        int l = labels[p];
        float v = 0.f;
        if (l >= 1 && l <= K) {
            int b = (int)(p / HW);
            v = g2 * coef[b * K + (l - 1)] * (embed[i] - cb[(long)(l - 1) * D + d]);
        }
        ge[i] = v;
    }
}
// float4 form for D = 4 * 2^lg (no integer divisions: the flat kernel spends most of its time in i / D, i % D, p / HW)
__global__ void __launch_bounds__(256) k_cross_bwd4(const float4* __restrict__ embed, const int32_t* __restrict__ labels, const float* __restrict__ cb,
                                                    const float* __restrict__ coef, const float* __restrict__ gl, float4* __restrict__ ge,
                                                    unsigned per_image4, int lg, int D, int K) {
    const float g2 = 2.f * gl[0];
    const unsigned b = blockIdx.y;
    const unsigned d4m = (1u << lg) - 1u;
    for (unsigned j = blockIdx.x * 256u + threadIdx.x; j < per_image4; j += gridDim.x * 256u) {
        const size_t i4 = (size_t)b * per_image4 + j;
        const size_t p = i4 >> lg;
        const int l = labels[p];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (l >= 1 && l <= K) {
            const float s = g2 * coef[b * K + (l - 1)];
            const float4 e = embed[i4];
            const float* c = cb + (long)(l - 1) * D + 4 * (j & d4m);
            v.x = s * (e.x - c[0]); v.y = s * (e.y - c[1]); v.z = s * (e.z - c[2]); v.w = s * (e.w - c[3]);
        }
        ge[i4] = v;
    }
}
extern "C" int vqw_cross_loss_bwd(const float* embed, const int32_t* labels, const float* codebook_kd, const float* coef,
                                  const float* gloss, float* gembed, int B, long HW, int D, int K, void* stream) {
    VQW_CHECK(embed && labels && codebook_kd && coef && gloss && gembed && B > 0 && HW > 0 && D > 0 && K > 0,
              "vqw_cross_loss_bwd: bad arguments");
    long total = (long)B * HW * D;
    const int D4 = D / 4;
    if (D % 4 == 0 && (D4 & (D4 - 1)) == 0 && HW * D4 < (1L << 31) && B <= 65535 && ((((uintptr_t)embed | (uintptr_t)gembed)) & 15) == 0) {
        int lg = 0;
        while ((1 << lg) < D4) ++lg;
        const unsigned per = (unsigned)(HW * D4);
        unsigned gx = (per + 255u) / 256u;
        if (gx > 2048u) gx = 2048u;
        k_cross_bwd4<<<dim3(gx, B), 256, 0, (hipStream_t)stream>>>((const float4*)embed, labels, codebook_kd, coef, gloss, (float4*)gembed, per, lg, D, K);
        VQW_LAUNCH_CHECK("vqw_cross_loss_bwd");
        return VQW_OK;
    }
    k_cross_bwd<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(embed, labels, codebook_kd, coef, gloss, gembed, total, HW, D, K);
    VQW_LAUNCH_CHECK("vqw_cross_loss_bwd");
    return VQW_OK;
}
__global__ void k_cross_bwd_dense(const float* __restrict__ embed, const float* __restrict__ r, const float* __restrict__ cb,
                                  const float* __restrict__ coef, const float* __restrict__ gl, float* __restrict__ ge,
                                  long total, long HW, int D, int K) {
    float g2 = 2.f * gl[0];
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        long p = i / D;
        int d = (int)(i % D);
        int b = (int)(p / HW);
        long pp = p % HW;
        float e = embed[i];
        float v = 0.f;
        for (int k = 0; k < K; ++k) {
            float w = r[((long)b * K + k) * HW + pp];
            if (w != 0.f) v += w * coef[b * K + k] * (e - cb[(long)k * D + d]);
        }
        ge[i] = g2 * v;
    }
}
extern "C" int vqw_cross_loss_dense_bwd(const float* embed, const float* r_nchw, const float* codebook_kd, const float* coef,
                                        const float* gloss, float* gembed, int B, long HW, int D, int K, void* stream) {
    VQW_CHECK(embed && r_nchw && codebook_kd && coef && gloss && gembed && B > 0 && HW > 0 && D > 0 && K > 0,
              "vqw_cross_loss_dense_bwd: bad arguments");
    long total = (long)B * HW * D;
    k_cross_bwd_dense<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(embed, r_nchw, codebook_kd, coef, gloss, gembed, total, HW, D, K);
    VQW_LAUNCH_CHECK("vqw_cross_loss_dense_bwd");
    return VQW_OK;
}

// embed_loss.py:68-88: l_dist = sum_{i,j} clamp(2 margin - |c_i - c_j|, 0)^2 / (2 K (K-1)) over all K^2 pairs (i == j
// included, each contributing (2 margin)^2, as upstream), l_reg = mean_k |c_k|.  One workgroup per row i: its four waves
// take the partners j round-robin, lanes run over the channels (coalesced rows from L2), wave butterfly per pair, pair
// terms summed in double in a fixed order; a second launch folds the K row partials.  (A single workgroup over the K^2
// pairs took 123 ms at K = 1024, D = 256 - a third of BASELINE config 4's step.)
__global__ void __launch_bounds__(256) k_codebook_rows(const float* __restrict__ cb, float margin, double* __restrict__ part, int D, int K) {
    __shared__ double s_a[4];
    const int i = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float* ci = cb + (long)i * D;
    double dsum = 0.0;
    for (int j = wv; j < K; j += 4) {
        const float* cj = cb + (long)j * D;
        float s = 0.f;
        for (int d = lane; d < D; d += 64) { const float a = ci[d] - cj[d]; s = fmaf(a, a, s); }
        s = wave_sum_f(s);
        float h = 2.f * margin - sqrtf(s);
        h = h > 0.f ? h : 0.f;
        dsum += (double)(h * h);
    }
    float n2 = 0.f;
    for (int d = lane; d < D; d += 64) n2 = fmaf(ci[d], ci[d], n2);
    n2 = wave_sum_f(n2);
    if (lane == 0) s_a[wv] = dsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * i] = (s_a[0] + s_a[1]) + (s_a[2] + s_a[3]);
        part[2 * i + 1] = (double)sqrtf(n2);
    }
}
__global__ void __launch_bounds__(256) k_codebook_fold(const double* __restrict__ part, float* __restrict__ l_dist, float* __restrict__ l_reg, int K) {
    __shared__ double s_a[256], s_b[256];
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < K; k += 256) { a += part[2 * k]; b += part[2 * k + 1]; }
    s_a[threadIdx.x] = a; s_b[threadIdx.x] = b;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) { s_a[threadIdx.x] += s_a[threadIdx.x + w]; s_b[threadIdx.x] += s_b[threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        l_dist[0] = (float)(s_a[0] / (2.0 * K * (K - 1)));
        l_reg[0] = (float)(s_b[0] / K);
    }
}
extern "C" int vqw_codebook_losses(const float* codebook_kd, float margin, float* l_dist, float* l_reg, void* ws, size_t ws_bytes, int D,
                                   int K, void* stream) {
    VQW_CHECK(codebook_kd && l_dist && l_reg && ws && D > 0 && K > 1, "vqw_codebook_losses: bad arguments (K must be > 1)");
    VQW_CHECK(ws_bytes >= (size_t)16 * K, "vqw_codebook_losses: workspace too small (needs 16 K bytes)");
    hipStream_t st = (hipStream_t)stream;
    k_codebook_rows<<<K, 256, 0, st>>>(codebook_kd, margin, (double*)ws, D, K);
    k_codebook_fold<<<1, 256, 0, st>>>((const double*)ws, l_dist, l_reg, K);
    VQW_LAUNCH_CHECK("vqw_codebook_losses");
    return VQW_OK;
}

// onehot.py:11-20: labels [B][HW] -> out[B][n_classes][HW] float (NCHW, as the reference returns it)
__global__ void k_onehot(const int32_t* __restrict__ labels, float* __restrict__ out, long total, long HW, int nc) {
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        long p = i % HW;
        long bc = i / HW;
        int c = (int)(bc % nc);
        long b = bc / nc;
        out[i] = labels[b * HW + p] == c ? 1.f : 0.f;
    }
}
extern "C" int vqw_onehot(const int32_t* labels, float* out_nchw, int B, long HW, int n_classes, void* stream) {
    VQW_CHECK(labels && out_nchw && B > 0 && HW > 0 && n_classes > 0, "vqw_onehot: bad arguments");
    long total = (long)B * n_classes * HW;
    k_onehot<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(labels, out_nchw, total, HW, n_classes);
    VQW_LAUNCH_CHECK("vqw_onehot");
    return VQW_OK;
}

// Exact-integer cross-view id map for identity / horizontal-flip views (single_window_trainer.py:91-96 with
// flip transforms): out[b][h][w] = ids[b][h][W-1-w], zero inside `border` pixels of the frame.
__global__ void k_flip_labels(const int64_t* __restrict__ ids, int32_t* __restrict__ out, int border, long total, int H, int W) {
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int w = (int)(i % W);
        long r = i / W;
        int h = (int)(r % H);
        int v = (int)ids[r * W + (W - 1 - w)];
        if (h < border || h >= H - border || w < border || w >= W - border) v = 0;
        out[i] = v;
    }
}
extern "C" int vqw_flip_labels(const int64_t* ids, int32_t* out, int border, int B, int H, int W, void* stream) {
    VQW_CHECK(ids && out && B > 0 && H > 0 && W > 0 && border >= 0, "vqw_flip_labels: bad arguments");
    long total = (long)B * H * W;
    k_flip_labels<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(ids, out, border, total, H, W);
    VQW_LAUNCH_CHECK("vqw_flip_labels");
    return VQW_OK;
}
