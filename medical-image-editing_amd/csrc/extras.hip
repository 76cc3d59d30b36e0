// Optional-path kernels: PixelShuffle(2) (blocks.py:100-104, unet_decoder.py:78-86), DropBlock (dropblock.py:47-94),
// SoftDice / Focal losses (functions/seg_loss.py:15-62).  All HBM-bound element-wise / reduction work.
#include "common.h"
#include "../../include/vqwnet_hip.h"

// ---------------------------------------------------------------------------------------------
// PixelShuffle(r=2), NHWC: y[n, 2h+i, 2w+j, c] = x[n, h, w, c*4 + i*2 + j].  INVERSE=1 is the backward map.
template <int INVERSE>
__global__ void k_pixel_shuffle2(const float* __restrict__ src, float* __restrict__ dst, int N, int H, int W, int C) {
    // (H, W, C) describe the OUTPUT of the forward map (high-res, C channels); low-res tensor is (H/2, W/2, 4C)
    long total = (long)N * H * W * C;
    long stride = (long)gridDim.x * blockDim.x;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
        int c = (int)(o % C);
        long p = o / C;
        int x = (int)(p % W);
        long q = p / W;
        int y = (int)(q % H);
        int n = (int)(q / H);
        long lo = ((((long)n * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) * (4L * C)) + c * 4 + (y & 1) * 2 + (x & 1);
        if (INVERSE) dst[lo] = src[o];
        else dst[o] = src[lo];
    }
}
extern "C" int vqw_pixel_shuffle2(const float* src, float* dst, int N, int H, int W, int C, int inverse, void* stream) {
    VQW_CHECK(src && dst && N > 0 && H > 0 && W > 0 && C > 0 && (H % 2 == 0) && (W % 2 == 0), "vqw_pixel_shuffle2: bad arguments");
    long total = (long)N * H * W * C;
    if (inverse) k_pixel_shuffle2<1><<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(src, dst, N, H, W, C);
    else k_pixel_shuffle2<0><<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(src, dst, N, H, W, C);
    VQW_LAUNCH_CHECK("vqw_pixel_shuffle2");
    return VQW_OK;
}

// ---------------------------------------------------------------------------------------------
// DropBlock: keep[n,h,w] = 1 - max over the block window of seed (stride 1, pad block/2, even sizes cropped at the end);
// scale = numel(keep) / sum(keep).  One workgroup per launch keeps the count deterministic (masks are B*H*W small).
__global__ void k_dropblock_mask(const float* __restrict__ seed, float* __restrict__ keep, float* __restrict__ scale, int N,
                                 int H, int W, int block) {
    __shared__ double s_red[16];
    const int pad = block / 2;
    long total = (long)N * H * W;
    double cnt = 0.0;
    for (long i = threadIdx.x; i < total; i += blockDim.x) {
        int x = (int)(i % W);
        long q = i / W;
        int y = (int)(q % H);
        int n = (int)(q / H);
        float m = 0.f;
        // output (y,x) of the padded stride-1 pooling covers input rows y-pad .. y-pad+block-1
        for (int dy = 0; dy < block; ++dy) {
            int yy = y - pad + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = 0; dx < block; ++dx) {
                int xx = x - pad + dx;
                if (xx < 0 || xx >= W) continue;
                m = fmaxf(m, seed[((long)n * H + yy) * W + xx]);
            }
        }
        float k = 1.f - m;
        keep[i] = k;
        cnt += (double)k;
    }
    cnt = wave_sum_d(cnt);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) a += s_red[w];
        scale[0] = (float)((double)total / a);
    }
}
extern "C" int vqw_dropblock_mask(const float* seed, float* keep, float* scale_dev, int N, int H, int W, int block_size,
                                  void* stream) {
    VQW_CHECK(seed && keep && scale_dev && N > 0 && H > 0 && W > 0 && block_size >= 1, "vqw_dropblock_mask: bad arguments");
    k_dropblock_mask<<<1, 1024, 0, (hipStream_t)stream>>>(seed, keep, scale_dev, N, H, W, block_size);
    VQW_LAUNCH_CHECK("vqw_dropblock_mask");
    return VQW_OK;
}
// y = x * keep[n,h,w] * scale   (also the backward: gx = gy * keep * scale)
__global__ void k_dropblock_apply(const float* __restrict__ x, const float* __restrict__ keep, const float* __restrict__ scale,
                                  float* __restrict__ y, long total, int C) {
    const float s = scale[0];
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) y[i] = x[i] * keep[i / C] * s;
}
extern "C" int vqw_dropblock_apply(const float* x, const float* keep, const float* scale_dev, float* y, long P, int C, void* stream) {
    VQW_CHECK(x && keep && scale_dev && y && P > 0 && C > 0, "vqw_dropblock_apply: bad arguments");
    long total = P * C;
    k_dropblock_apply<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(x, keep, scale_dev, y, total, C);
    VQW_LAUNCH_CHECK("vqw_dropblock_apply");
    return VQW_OK;
}

// ---------------------------------------------------------------------------------------------
// Segmentation losses on NCHW logits / one-hot targets (the layout the reference's flatten() consumes), C <= 64.
// Pass 1 accumulates per-class sums (dice) or the pixel sum (focal); pass 2 (backward) recomputes the softmax.
#define SEG_MAXC 64
#define SEG_BLOCKS 512
extern "C" size_t vqw_seg_ws_bytes(int C) { return (size_t)SEG_BLOCKS * (2 * C + 1) * sizeof(double) + (2 * C + 2) * sizeof(double); }

__device__ __forceinline__ void seg_softmax(const float* __restrict__ z, long base, long HW, int C, float* p, float& lse) {
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) { p[c] = z[base + c * HW]; m = fmaxf(m, p[c]); }
    float s = 0.f;
    for (int c = 0; c < C; ++c) { p[c] = __expf(p[c] - m); s += p[c]; }
    float inv = 1.f / s;
    for (int c = 0; c < C; ++c) p[c] *= inv;
    lse = m + __logf(s);
}

// part[block][0..C) = sum p*t, [C..2C) = sum p + sum t, [2C] = focal pixel sum
__global__ void __launch_bounds__(256) k_seg_partial(const float* __restrict__ z, const float* __restrict__ t, double* __restrict__ part,
                                                     int B, long HW, int C, float gamma, float eps) {
    extern __shared__ __attribute__((aligned(16))) double sacc[];   // [2C+1]
    for (int i = threadIdx.x; i < 2 * C + 1; i += 256) sacc[i] = 0.0;
    __syncthreads();
    float p[SEG_MAXC];
    long total = (long)B * HW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long b = i / HW, px = i % HW;
        long base = b * C * HW + px;
        float lse;
        seg_softmax(z, base, HW, C, p, lse);
        float f = 0.f;
        for (int c = 0; c < C; ++c) {
            float tc = t[base + c * HW];
            atomicAdd(&sacc[c], (double)(p[c] * tc));
            atomicAdd(&sacc[C + c], (double)(p[c] + tc));
            float pc = fminf(fmaxf(p[c], eps), 1.f - eps);
            float lp = z[base + c * HW] - lse;
            f += -tc * lp * __powf(1.f - pc, gamma);
        }
        atomicAdd(&sacc[2 * C], (double)f);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C + 1; i += 256) part[(long)blockIdx.x * (2 * C + 1) + i] = sacc[i];
}
// sums[0..C) inter, [C..2C) denom, [2C] focal sum, [2C+1] unused;  loss_out[0] = dice, loss_out[1] = focal
__global__ void k_seg_finalize(const double* __restrict__ part, int nblocks, double* __restrict__ sums, float* __restrict__ loss_out,
                               int C, int ignore_index, float smooth, double inv_pixels) {
    __shared__ double s[2 * SEG_MAXC + 1];
    for (int i = threadIdx.x; i < 2 * C + 1; i += blockDim.x) {
        double a = 0.0;
        for (int b = 0; b < nblocks; ++b) a += part[(long)b * (2 * C + 1) + i];
        s[i] = a;
        sums[i] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double I = 0.0, D = 0.0;
        for (int c = 0; c < C; ++c)
            if (c != ignore_index) { I += s[c]; D += s[C + c]; }
        double Dc = D > (double)smooth ? D : (double)smooth;
        loss_out[0] = (float)(1.0 - 2.0 * I / Dc);
        loss_out[1] = (float)(s[2 * C] * inv_pixels);
        sums[2 * C + 1] = D;
    }
}
extern "C" int vqw_seg_losses_fwd(const float* logits_nchw, const float* target_nchw, float* loss_out /*[2]: dice, focal*/,
                                  double* sums /*[2C+2]*/, void* ws, size_t ws_bytes, int B, long HW, int C, int ignore_index,
                                  float smooth, float gamma, float eps, void* stream) {
    VQW_CHECK(logits_nchw && target_nchw && loss_out && sums && ws && B > 0 && HW > 0 && C > 0 && C <= SEG_MAXC,
              "vqw_seg_losses_fwd: bad arguments (C <= %d)", SEG_MAXC);
    VQW_CHECK(ws_bytes >= vqw_seg_ws_bytes(C), "vqw_seg_losses_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int nb = imin(SEG_BLOCKS, imax(1, (int)(((long)B * HW + 255) / 256)));
    k_seg_partial<<<nb, 256, (2 * C + 1) * sizeof(double), st>>>(logits_nchw, target_nchw, (double*)ws, B, HW, C, gamma, eps);
    k_seg_finalize<<<1, 128, 0, st>>>((const double*)ws, nb, sums, loss_out, C, ignore_index, smooth, 1.0 / ((double)B * HW));
    VQW_LAUNCH_CHECK("vqw_seg_losses_fwd");
    return VQW_OK;
}
// gz = g_dice * d(dice)/dz + g_focal * d(focal)/dz
__global__ void k_seg_bwd(const float* __restrict__ z, const float* __restrict__ t, const double* __restrict__ sums,
                          const float* __restrict__ g_dice, const float* __restrict__ g_focal, float* __restrict__ gz, int B, long HW,
                          int C, int ignore_index, float smooth, float gamma, float eps, float inv_pixels) {
    float p[SEG_MAXC], a[SEG_MAXC];
    double I = 0.0, D = 0.0;
    for (int c = 0; c < C; ++c)
        if (c != ignore_index) { I += sums[c]; D += sums[C + c]; }
    const bool live = D > (double)smooth;
    const float gd = g_dice ? g_dice[0] : 0.f, gf = g_focal ? g_focal[0] * inv_pixels : 0.f;
    // d dice / d p_c = -2 (D - I) / D^2 for kept classes (t enters D only additively)  [clamped denominator: -2/smooth]
    const float dpk = live ? (float)(-2.0 * (D - I) / (D * D)) : 0.f;
    const float dpt = live ? (float)(-2.0 / D) : (float)(-2.0 / (double)smooth);
    long total = (long)B * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long b = i / HW, px = i % HW;
        long base = b * C * HW + px;
        float lse;
        seg_softmax(z, base, HW, C, p, lse);
        // dice: g_c = dL/dp_c = t_c * dpt + (live ? 2I/D^2 : 0)  ->  written as t_c*(-2/D) + 2I/D^2
        float sd = 0.f, sf = 0.f;
        for (int c = 0; c < C; ++c) {
            float tc = t[base + c * HW];
            float gdc = (c != ignore_index) ? (tc * dpt + (live ? (float)(2.0 * I / (D * D)) : 0.f)) : 0.f;
            float pc = fminf(fmaxf(p[c], eps), 1.f - eps);
            float mk = (p[c] > eps && p[c] < 1.f - eps) ? 1.f : 0.f;
            float lp = z[base + c * HW] - lse;
            float om = 1.f - pc;
            float A = -tc * (__powf(om, gamma) - lp * gamma * __powf(om, gamma - 1.f) * mk * p[c]);
            a[c] = gdc;          // reuse: dice upstream wrt p
            sd += gdc * p[c];
            sf += A;
            p[c] = p[c];
            gz[base + c * HW] = A;   // stash A (focal) in the output for the second loop
        }
        (void)dpk;
        for (int c = 0; c < C; ++c) {
            float A = gz[base + c * HW];
            float dice_z = p[c] * (a[c] - sd);
            float focal_z = A - p[c] * sf;
            gz[base + c * HW] = gd * dice_z + gf * focal_z;
        }
    }
}
extern "C" int vqw_seg_losses_bwd(const float* logits_nchw, const float* target_nchw, const double* sums, const float* g_dice,
                                  const float* g_focal, float* glogits, int B, long HW, int C, int ignore_index, float smooth,
                                  float gamma, float eps, void* stream) {
    VQW_CHECK(logits_nchw && target_nchw && sums && glogits && B > 0 && HW > 0 && C > 0 && C <= SEG_MAXC,
              "vqw_seg_losses_bwd: bad arguments");
    long total = (long)B * HW;
    k_seg_bwd<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(logits_nchw, target_nchw, sums, g_dice, g_focal, glogits, B,
                                                                         HW, C, ignore_index, smooth, gamma, eps,
                                                                         1.0f / (float)total);
    VQW_LAUNCH_CHECK("vqw_seg_losses_bwd");
    return VQW_OK;
}
