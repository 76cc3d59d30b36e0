// Two-view augmentation and id-map warps on the device (reference: networks/random_transform.py:10-112, which builds
// them from kornia 0.5.1 — absent offline; the exact arithmetic implemented here is the one oracle/augment_ref.py
// states, parity unpinned).  All HBM-bound streaming kernels over (B, C, H, W) fp32 planes / (B, H, W) id maps:
//   warp_image   bilinear resample through a per-sample 3x3 matrix (destination pixel -> source pixel), zero padding
//   warp_labels  nearest resample of an id map through the same kind of matrix, 0 = out of frame
//   photometric  brightness add, contrast multiply (both clamped to [0,1]), posterize, additive Gaussian noise
//   gauss_blur   separable Gaussian, reflect border, per-sample on/off
#include "common.h"

namespace {

// Source coordinates are evaluated in double: exact for the integer transforms (flips, whole-pixel shifts) and far
// enough from the nearest-neighbour rounding boundary otherwise that the oracle's float64 arithmetic picks the same pixel.
__device__ __forceinline__ void map_point(const float* __restrict__ m, int xd, int yd, double& xs, double& ys) {
    double x = (double)xd, y = (double)yd;
    double u = (double)m[0] * x + (double)m[1] * y + (double)m[2];
    double v = (double)m[3] * x + (double)m[4] * y + (double)m[5];
    double w = (double)m[6] * x + (double)m[7] * y + (double)m[8];
    xs = u / w;
    ys = v / w;
}

__global__ void k_warp_image(const float* __restrict__ src, const float* __restrict__ minv, float* __restrict__ dst, long total,
                             int C, int H, int W) {
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int x = (int)(i % W);
        long r = i / W;
        int y = (int)(r % H);
        long plane = r / H;                 // b * C + c
        int b = (int)(plane / C);
        double xs, ys;
        map_point(minv + 9L * b, x, y, xs, ys);
        double x0f = floor(xs), y0f = floor(ys);
        float fx = (float)(xs - x0f), fy = (float)(ys - y0f);
        int x0 = (int)x0f, y0 = (int)y0f;
        const float* p = src + plane * (long)H * W;
        auto at = [&](int yy, int xx) -> float {
            return ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? p[(long)yy * W + xx] : 0.f;
        };
        // finite check: a degenerate matrix row (w = 0) gives inf / nan coordinates -> out of frame
        float v = 0.f;
        if (xs == xs && ys == ys && fabs(xs) < 1e9 && fabs(ys) < 1e9) {
            float top = at(y0, x0) * (1.f - fx) + at(y0, x0 + 1) * fx;
            float bot = at(y0 + 1, x0) * (1.f - fx) + at(y0 + 1, x0 + 1) * fx;
            v = top * (1.f - fy) + bot * fy;
        }
        dst[i] = v;
    }
}

template <typename TI>
__global__ void k_warp_labels(const TI* __restrict__ ids, const float* __restrict__ minv, int32_t* __restrict__ out, long total,
                              int H, int W) {
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int x = (int)(i % W);
        long r = i / W;
        int y = (int)(r % H);
        int b = (int)(r / H);
        double xs, ys;
        map_point(minv + 9L * b, x, y, xs, ys);
        int32_t v = 0;
        if (xs == xs && ys == ys && fabs(xs) < 1e9 && fabs(ys) < 1e9) {
            int xi = (int)rint(xs), yi = (int)rint(ys);      // round half to even, as torch's nearest grid sampling
            if ((unsigned)yi < (unsigned)H && (unsigned)xi < (unsigned)W) v = (int32_t)ids[((long)b * H + yi) * W + xi];
        }
        out[i] = v;
    }
}

// params per sample: {brightness add, contrast multiplier, posterize bits (8 = off), noise std}
__global__ void k_photometric(const float* __restrict__ x, const float* __restrict__ params, const float* __restrict__ noise,
                              float* __restrict__ y, long total, long per_sample) {
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const float* q = params + 4 * (i / per_sample);
        float v = x[i];
        v = fminf(fmaxf(v + q[0], 0.f), 1.f);
        v = fminf(fmaxf(v * q[1], 0.f), 1.f);
        int bits = (int)q[2];
        if (bits < 8) {
            int u = (int)(v * 255.f);                 // truncation, as a cast to uint8
            u &= (0xFF << (8 - bits)) & 0xFF;
            v = (float)u / 255.f;
        }
        if (noise) v += q[3] * noise[i];
        y[i] = v;
    }
}

__device__ __forceinline__ int reflect(int i, int n) {      // torch 'reflect' padding (no edge repeat); n >= 2
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}
template <int HORIZONTAL>
__global__ void k_blur_pass(const float* __restrict__ x, const float* __restrict__ taps, const unsigned char* __restrict__ apply,
                            float* __restrict__ y, long total, int planes_per_sample, int H, int W, int K) {
    long stride = (long)gridDim.x * blockDim.x;
    const int half = K / 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int xx = (int)(i % W);
        long r = i / W;
        int yy = (int)(r % H);
        long plane = r / H;
        if (apply && !apply[plane / planes_per_sample]) {
            y[i] = x[i];
            continue;
        }
        const float* p = x + plane * (long)H * W;
        float acc = 0.f;
        for (int t = 0; t < K; ++t) {
            int xs = HORIZONTAL ? reflect(xx + t - half, W) : xx;
            int ys = HORIZONTAL ? yy : reflect(yy + t - half, H);
            acc += taps[t] * p[(long)ys * W + xs];
        }
        y[i] = acc;
    }
}

}  // namespace

extern "C" int vqw_warp_image(const float* src, const float* minv, float* dst, int B, int C, int H, int W, void* stream) {
    VQW_CHECK(src && minv && dst && src != dst && B > 0 && C > 0 && H > 0 && W > 0, "vqw_warp_image: bad arguments");
    long total = (long)B * C * H * W;
    k_warp_image<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(src, minv, dst, total, C, H, W);
    VQW_LAUNCH_CHECK("vqw_warp_image");
    return VQW_OK;
}

extern "C" int vqw_warp_labels(const void* ids, int ids_are_int64, const float* minv, int32_t* out, int B, int H, int W,
                               void* stream) {
    VQW_CHECK(ids && minv && out && (const void*)out != ids && B > 0 && H > 0 && W > 0, "vqw_warp_labels: bad arguments");
    long total = (long)B * H * W;
    hipStream_t st = (hipStream_t)stream;
    if (ids_are_int64) k_warp_labels<int64_t><<<stream_grid(total, 256), 256, 0, st>>>((const int64_t*)ids, minv, out, total, H, W);
    else k_warp_labels<int32_t><<<stream_grid(total, 256), 256, 0, st>>>((const int32_t*)ids, minv, out, total, H, W);
    VQW_LAUNCH_CHECK("vqw_warp_labels");
    return VQW_OK;
}

extern "C" int vqw_photometric(const float* x, const float* params, const float* noise, float* y, int B, long per_sample,
                               void* stream) {
    VQW_CHECK(x && params && y && B > 0 && per_sample > 0, "vqw_photometric: bad arguments");
    long total = (long)B * per_sample;
    k_photometric<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(x, params, noise, y, total, per_sample);
    VQW_LAUNCH_CHECK("vqw_photometric");
    return VQW_OK;
}

extern "C" int vqw_gauss_blur(const float* x, const float* taps, const unsigned char* apply, float* tmp, float* y, int B, int C,
                              int H, int W, int K, void* stream) {
    VQW_CHECK(x && taps && tmp && y && x != tmp && tmp != y && B > 0 && C > 0 && H > 1 && W > 1, "vqw_gauss_blur: bad arguments");
    VQW_CHECK(K >= 1 && (K & 1) && K / 2 < H && K / 2 < W, "vqw_gauss_blur: kernel size %d must be odd and smaller than 2x the image", K);
    long total = (long)B * C * H * W;
    hipStream_t st = (hipStream_t)stream;
    k_blur_pass<1><<<stream_grid(total, 256), 256, 0, st>>>(x, taps, apply, tmp, total, C, H, W, K);
    k_blur_pass<0><<<stream_grid(total, 256), 256, 0, st>>>(tmp, taps, apply, y, total, C, H, W, K);
    VQW_LAUNCH_CHECK("vqw_gauss_blur");
    return VQW_OK;
}
