// PatchGAN discriminator path of the second training step (reference: networks/discriminator.py:18-87,
// functions/gan_loss.py:6-10, trainers/single_window_trainer.py:434-488): 4x4 convolutions with stride 1 or 2 and
// padding 1, BatchNorm2d (affine) + LeakyReLU(0.2), hinge / generator losses.  NHWC activations, OHWI weights.
//
// The wide layers (64->128, 128->256 stride 2; 256->512 stride 1) run on the exact-fp32 MFMA kernels of conv_mfma.hip:
// the stride-2 gather is the geometry of the collapsed dgrad, its transpose the collapsed forward, its weight gradient
// the parity/tap matrices of the collapsed wgrad with the roles swapped; the stride-1 layers run on a common H x W grid
// with a 16-tap table (this file pads / crops the (H-1) x (W-1) side).  The 1-channel ends (1->64, 512->1) and any
// other shape use the direct VALU kernels below (any kernel size / stride / padding).
#include "common.h"
#include "conv_common.h"
#include "../../include/vqwnet_hip.h"

namespace {

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// one thread per output element (pixel, co); adjacent threads = adjacent co
__global__ void __launch_bounds__(256) k_sconv_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ y, int N, int H, int W,
                                                   int Ho, int Wo, int Cin, int Cout, int ks, int stride, int pad, float slope) {
    const int taps = ks * ks;
    long total = (long)N * Ho * Wo * Cout;
    long gstride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
        int co = (int)(i % Cout);
        long p = i / Cout;
        int xo = (int)(p % Wo);
        long q = p / Wo;
        int yo = (int)(q % Ho);
        int n = (int)(q / Ho);
        float acc = bias ? bias[co] : 0.f;
        for (int t = 0; t < taps; ++t) {
            int hy = yo * stride - pad + t / ks, wx = xo * stride - pad + t % ks;
            if ((unsigned)hy >= (unsigned)H || (unsigned)wx >= (unsigned)W) continue;
            const float* wr = w + ((long)co * taps + t) * Cin;
            const float* s = x + (((long)n * H + hy) * W + wx) * Cin;
            if ((Cin & 3) == 0) {
                for (int c = 0; c < Cin; c += 4) {
                    float4 a = *(const float4*)(s + c), b = *(const float4*)(wr + c);
                    acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
                }
            } else {
                for (int c = 0; c < Cin; ++c) acc = fmaf(s[c], wr[c], acc);
            }
        }
        y[i] = lrelu(acc, slope);
    }
}

// input gradient: one thread per (input pixel, ci); gx = sum over the output pixels / taps that touched it
__global__ void __launch_bounds__(256) k_sconv_dgrad(const float* __restrict__ gy, const float* __restrict__ w,
                                                     float* __restrict__ gx, int N, int H, int W, int Ho, int Wo, int Cin,
                                                     int Cout, int ks, int stride, int pad) {
    const int taps = ks * ks;
    long total = (long)N * H * W * Cin;
    long gstride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
        int ci = (int)(i % Cin);
        long p = i / Cin;
        int xx = (int)(p % W);
        long q = p / W;
        int yy = (int)(q % H);
        int n = (int)(q / H);
        float acc = 0.f;
        for (int t = 0; t < taps; ++t) {
            int ny = yy + pad - t / ks, nx = xx + pad - t % ks;
            if (ny < 0 || nx < 0 || ny % stride || nx % stride) continue;
            int yo = ny / stride, xo = nx / stride;
            if (yo >= Ho || xo >= Wo) continue;
            const float* g = gy + (((long)n * Ho + yo) * Wo + xo) * Cout;
            const float* wr = w + (long)t * Cin + ci;                 // w[co][t][ci], co stride taps*Cin
            for (int co = 0; co < Cout; ++co) acc = fmaf(g[co], wr[(long)co * taps * Cin], acc);
        }
        gx[i] = acc;
    }
}

// ---- the discriminator's 1-channel ends (1 -> 64 stride 2 in front, 512 -> 1 stride 1 at the back), 4x4 taps -----------
// forward, Cin = 1: a thread owns one output pixel and four couts; the 16 input taps are read once into registers
// (neighbouring threads of a pixel read the same addresses: broadcast), the 4 x 16 weights come from LDS.
__global__ void __launch_bounds__(256) k_sconv_fwd_c1(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y, int N, int H, int W,
                                                      int Ho, int Wo, int Cout, int stride, int pad, float slope) {
    extern __shared__ float sw[];                          // [Cout][16]
    for (int i = threadIdx.x; i < Cout * 16; i += 256) sw[i] = w[i];
    __syncthreads();
    const int C4 = Cout >> 2;
    long total = (long)N * Ho * Wo * C4;
    long gstride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
        int c4 = (int)(i % C4);
        long p = i / C4;
        int xo = (int)(p % Wo);
        long q = p / Wo;
        int yo = (int)(q % Ho);
        int n = (int)(q / Ho);
        const float* img = x + (long)n * H * W;
        float v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            int hy = yo * stride - pad + (t >> 2), wx = xo * stride - pad + (t & 3);
            v[t] = ((unsigned)hy < (unsigned)H && (unsigned)wx < (unsigned)W) ? img[(long)hy * W + wx] : 0.f;
        }
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int co = c4 * 4 + k;
            float acc = bias ? bias[co] : 0.f;
            const float* wr = sw + co * 16;
#pragma unroll
            for (int t = 0; t < 16; ++t) acc = fmaf(v[t], wr[t], acc);
            o[k] = lrelu(acc, slope);
        }
        *(float4*)(y + p * Cout + c4 * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
// input gradient, Cin = 1: a wave per input pixel group ... one thread per input pixel; the (at most 16) contributing
// taps each read a contiguous Cout-vector of gy; weights w[co][t] from LDS transposed to [t][co]
__global__ void __launch_bounds__(256) k_sconv_dgrad_c1(const float* __restrict__ gy, const float* __restrict__ w,
                                                        float* __restrict__ gx, int N, int H, int W, int Ho, int Wo, int Cout,
                                                        int stride, int pad) {
    extern __shared__ float sw[];                          // [16][Cout]
    for (int i = threadIdx.x; i < Cout * 16; i += 256) sw[(i & 15) * Cout + (i >> 4)] = w[i];
    __syncthreads();
    long total = (long)N * H * W;
    long gstride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
        int xx = (int)(i % W);
        long q = i / W;
        int yy = (int)(q % H);
        int n = (int)(q / H);
        float acc = 0.f;
        for (int t = 0; t < 16; ++t) {
            int ny = yy + pad - (t >> 2), nx = xx + pad - (t & 3);
            if (ny < 0 || nx < 0 || ny % stride || nx % stride) continue;
            int yo = ny / stride, xo = nx / stride;
            if (yo >= Ho || xo >= Wo) continue;
            const float4* g = (const float4*)(gy + (((long)n * Ho + yo) * Wo + xo) * Cout);
            const float4* wr = (const float4*)(sw + t * Cout);
            for (int c = 0; c < (Cout >> 2); ++c) {
                float4 a = g[c], b = wr[c];
                acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
            }
        }
        gx[i] = acc;
    }
}
// forward, Cout = 1: one wave per output pixel, lanes over the input channels (float4 each), butterfly sum
__global__ void __launch_bounds__(256) k_sconv_fwd_o1(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y, int N, int H, int W,
                                                      int Ho, int Wo, int Cin, int stride, int pad, float slope) {
    const int lane = threadIdx.x & 63;
    long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nw = ((long)gridDim.x * blockDim.x) >> 6;
    const long total = (long)N * Ho * Wo;
    for (long p = wave; p < total; p += nw) {
        int xo = (int)(p % Wo);
        long q = p / Wo;
        int yo = (int)(q % Ho);
        int n = (int)(q / Ho);
        float acc = 0.f;
        for (int t = 0; t < 16; ++t) {
            int hy = yo * stride - pad + (t >> 2), wx = xo * stride - pad + (t & 3);
            if ((unsigned)hy >= (unsigned)H || (unsigned)wx >= (unsigned)W) continue;      // wave-uniform
            const float* s = x + (((long)n * H + hy) * W + wx) * Cin;
            const float* wr = w + (long)t * Cin;
            for (int c = lane * 4; c < Cin; c += 256) {
                float4 a = *(const float4*)(s + c), b = *(const float4*)(wr + c);
                acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) y[p] = lrelu(acc + (bias ? bias[0] : 0.f), slope);
    }
}
// input gradient, Cout = 1: gx[p][ci] = sum_t w[t][ci] * gy[out pixel of tap t]; thread per (pixel, 4 channels)
__global__ void __launch_bounds__(256) k_sconv_dgrad_o1(const float* __restrict__ gy, const float* __restrict__ w,
                                                        float* __restrict__ gx, int N, int H, int W, int Ho, int Wo, int Cin,
                                                        int stride, int pad) {
    const int C4 = Cin >> 2;
    long total = (long)N * H * W * C4;
    long gstride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
        int c4 = (int)(i % C4);
        long p = i / C4;
        int xx = (int)(p % W);
        long q = p / W;
        int yy = (int)(q % H);
        int n = (int)(q / H);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = 0; t < 16; ++t) {
            int ny = yy + pad - (t >> 2), nx = xx + pad - (t & 3);
            if (ny < 0 || nx < 0 || ny % stride || nx % stride) continue;
            int yo = ny / stride, xo = nx / stride;
            if (yo >= Ho || xo >= Wo) continue;
            const float g = gy[((long)n * Ho + yo) * Wo + xo];
            const float4 b = *(const float4*)(w + (long)t * Cin + c4 * 4);
            acc.x = fmaf(g, b.x, acc.x); acc.y = fmaf(g, b.y, acc.y); acc.z = fmaf(g, b.z, acc.z); acc.w = fmaf(g, b.w, acc.w);
        }
        *(float4*)(gx + p * Cin + c4 * 4) = acc;
    }
}

// weight gradient: workgroup = (co, tap, pixel split); threads stride over (pixel, ci); partial[split][co][tap][ci]
__global__ void __launch_bounds__(256) k_sconv_wgrad(const float* __restrict__ x, const float* __restrict__ gy,
                                                     float* __restrict__ part, int N, int H, int W, int Ho, int Wo, int Cin,
                                                     int Cout, int ks, int stride, int pad, int nsplit) {
    extern __shared__ float red[];                 // [256 / cl][cl] partial sums, cl = lanes over ci
    const int taps = ks * ks;
    int b = blockIdx.x;
    const int split = b % nsplit; b /= nsplit;
    const int t = b % taps;
    const int co = b / taps;
    const int cl = Cin >= 256 ? 256 : (Cin >= 64 ? 64 : (Cin >= 16 ? 16 : (Cin >= 4 ? 4 : 1)));
    const int lanes_p = 256 / cl;
    const int lc = threadIdx.x % cl, lp = threadIdx.x / cl;
    const long Po = (long)N * Ho * Wo;
    const long per = (Po + nsplit - 1) / nsplit;
    const long p0 = split * per, p1 = p0 + per < Po ? p0 + per : Po;
    const int ky = t / ks, kx = t % ks;
    for (int c0 = 0; c0 < Cin; c0 += cl) {
        const int ci = c0 + lc;
        float acc = 0.f;
        if (ci < Cin) {
            for (long p = p0 + lp; p < p1; p += lanes_p) {
                int xo = (int)(p % Wo);
                long q = p / Wo;
                int yo = (int)(q % Ho);
                int n = (int)(q / Ho);
                int hy = yo * stride - pad + ky, wx = xo * stride - pad + kx;
                if ((unsigned)hy >= (unsigned)H || (unsigned)wx >= (unsigned)W) continue;
                acc = fmaf(gy[p * Cout + co], x[(((long)n * H + hy) * W + wx) * Cin + ci], acc);
            }
        }
        red[lp * cl + lc] = acc;
        __syncthreads();
        if (lp == 0 && ci < Cin) {
            float s = 0.f;
            for (int k = 0; k < lanes_p; ++k) s += red[k * cl + lc];
            part[(((long)split * Cout + co) * taps + t) * Cin + ci] = s;
        }
        __syncthreads();
    }
}

// weight gradient of a 1-input-channel layer (the discriminator's first conv): lanes over (co, 4 pixel lanes); per pixel
// the Cout gradients are one coalesced row and the k*k input taps are wave-uniform scalars.  partial[split][co][tap]
template <int TAPS>
__global__ void __launch_bounds__(256) k_sconv_wgrad_c1(const float* __restrict__ x, const float* __restrict__ gy,
                                                        float* __restrict__ part, int N, int H, int W, int Ho, int Wo, int Cout,
                                                        int ks, int stride, int pad, int nsplit) {
    __shared__ float red[256 * TAPS];
    const int cl = Cout < 256 ? Cout : 256;            // lanes over co (Cout <= 256 and a divisor of 256)
    const int lanes_p = 256 / cl;
    const int co = threadIdx.x % cl, lp = threadIdx.x / cl;
    const long Po = (long)N * Ho * Wo;
    const long per = (Po + nsplit - 1) / nsplit;
    const long p0 = blockIdx.x * per, p1 = p0 + per < Po ? p0 + per : Po;
    float acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc[t] = 0.f;
    for (long p = p0 + lp; p < p1; p += lanes_p) {
        int xo = (int)(p % Wo);
        long q = p / Wo;
        int yo = (int)(q % Ho);
        int n = (int)(q / Ho);
        const float g = gy[p * Cout + co];
        const float* img = x + (long)n * H * W;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            int hy = yo * stride - pad + t / ks, wx = xo * stride - pad + t % ks;
            float v = ((unsigned)hy < (unsigned)H && (unsigned)wx < (unsigned)W) ? img[(long)hy * W + wx] : 0.f;
            acc[t] = fmaf(g, v, acc[t]);
        }
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t) red[(lp * TAPS + t) * cl + co] = acc[t];
    __syncthreads();
    for (int i = threadIdx.x; i < cl * TAPS; i += 256) {
        int c = i % cl, t = i / cl;
        float s = 0.f;
        for (int k = 0; k < lanes_p; ++k) s += red[(k * TAPS + t) * cl + c];
        part[((long)blockIdx.x * Cout + c) * TAPS + t] = s;
    }
}

__global__ void k_leaky_bwd(const float* __restrict__ y, const float* __restrict__ gy, float* __restrict__ gx, float slope, long n) {
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) gx[i] = y[i] > 0.f ? gy[i] : gy[i] * slope;
}

// mode 0: mean(relu(1 - x)); 1: mean(relu(1 + x)); 2: -mean(x).  One workgroup, fixed order, double accumulation.
__global__ void __launch_bounds__(1024) k_hinge_fwd(const float* __restrict__ x, long n, int mode, float* __restrict__ loss) {
    __shared__ double sm[1024];
    double a = 0.0;
    for (long i = threadIdx.x; i < n; i += 1024) {
        float v = x[i];
        a += mode == 0 ? (double)fmaxf(1.f - v, 0.f) : (mode == 1 ? (double)fmaxf(1.f + v, 0.f) : -(double)v);
    }
    sm[threadIdx.x] = a;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(sm[0] / (double)n);
}
__global__ void k_hinge_bwd(const float* __restrict__ x, long n, int mode, const float* __restrict__ gloss, float* __restrict__ gx) {
    long stride = (long)gridDim.x * blockDim.x;
    const float g = gloss[0] / (float)n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = x[i];
        gx[i] = mode == 0 ? (1.f - v > 0.f ? -g : 0.f) : (mode == 1 ? (1.f + v > 0.f ? g : 0.f) : -g);
    }
}

// dst[n, y, x, :] = src[n, y, x, :] inside the source frame, 0 outside: crops (Hd < Hs) or zero-pads (Hd > Hs)
__global__ void k_crop_pad(const float* __restrict__ src, float* __restrict__ dst, long total, int Hs, int Ws, int Hd, int Wd, int C) {
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int c = (int)(i % C);
        long p = i / C;
        int x = (int)(p % Wd);
        long q = p / Wd;
        int y = (int)(q % Hd);
        long n = q / Hd;
        dst[i] = (y < Hs && x < Ws) ? src[((n * Hs + y) * Ws + x) * C + c] : 0.f;
    }
}
int crop_pad(const float* src, float* dst, int N, int Hs, int Ws, int Hd, int Wd, int C, hipStream_t st) {
    long total = (long)N * Hd * Wd * C;
    k_crop_pad<<<stream_grid(total, 256), 256, 0, st>>>(src, dst, total, Hs, Ws, Hd, Wd, C);
    VQW_LAUNCH_CHECK("crop_pad");
    return VQW_OK;
}

int out_dim(int n, int ks, int stride, int pad) { return (n + 2 * pad - ks) / stride + 1; }
// which 4x4 layers go to the MFMA kernels
bool k4_mfma(int N, int H, int W, int Cin, int Cout, int ks, int stride, int pad) {
    if (ks != 4 || pad != 1 || !conv_k4_mfma_ok(Cin, Cout, (long)N * H * W)) return false;
    return stride == 1 ? (H >= 4 && W >= 4) : (H % 2 == 0 && W % 2 == 0);
}
int wgrad_splits(long Po) {
    long s = Po / 512;
    return (int)(s < 1 ? 1 : (s > 256 ? 256 : s));
}
bool wgrad_c1_ok(int Cin, int Cout, int ks) { return Cin == 1 && ks == 4 && Cout <= 256 && 256 % Cout == 0; }
int wgrad_c1_splits(long Po) {
    long s = Po / 1024;
    return (int)(s < 1 ? 1 : (s > 1024 ? 1024 : s));
}

}  // namespace

static int check_sconv(const char* who, int N, int H, int W, int Cin, int Cout, int ks, int stride, int pad) {
    VQW_CHECK(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && ks >= 1 && ks <= 7 && (stride == 1 || stride == 2) && pad >= 0 &&
                  pad < ks && H + 2 * pad >= ks && W + 2 * pad >= ks,
              "%s: bad geometry N=%d H=%d W=%d Cin=%d Cout=%d k=%d stride=%d pad=%d", who, N, H, W, Cin, Cout, ks, stride, pad);
    return VQW_OK;
}

extern "C" size_t vqw_sconv_fwd_ws_bytes(int N, int H, int W, int Cin, int Cout, int ks, int stride, int pad) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    return (k4_mfma(N, H, W, Cin, Cout, ks, stride, pad) && stride == 1) ? (size_t)N * H * W * Cout * sizeof(float) : 0;
}
extern "C" int vqw_sconv_fwd(const float* x, const float* w_ohwi, const float* bias, float* y, void* ws, size_t ws_bytes, int N,
                             int H, int W, int Cin, int Cout, int ks, int stride, int pad, float slope, void* stream) {
    int rc = check_sconv("vqw_sconv_fwd", N, H, W, Cin, Cout, ks, stride, pad);
    if (rc) return rc;
    VQW_CHECK(x && w_ohwi && y, "vqw_sconv_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int Ho = out_dim(H, ks, stride, pad), Wo = out_dim(W, ks, stride, pad);
    if (slope == 1.f && k4_mfma(N, H, W, Cin, Cout, ks, stride, pad)) {
        if (stride == 2) return conv_k4s2_fwd(x, w_ohwi, bias, y, N, Ho, Wo, Cin, Cout, st);
        VQW_CHECK(ws && ws_bytes >= vqw_sconv_fwd_ws_bytes(N, H, W, Cin, Cout, ks, stride, pad), "vqw_sconv_fwd: workspace too small");
        rc = conv_k4s1_grid(x, w_ohwi, bias, (float*)ws, N, H, W, Cin, Cout, 1, st);       // on the H x W grid ...
        if (rc) return rc;
        return crop_pad((const float*)ws, y, N, H, W, Ho, Wo, Cout, st);                  // ... cropped to (H-1) x (W-1)
    }
    long total = (long)N * Ho * Wo * Cout;
    if (ks == 4 && Cin == 1 && Cout % 4 == 0 && Cout <= 1024 && (((uintptr_t)y) & 15) == 0) {
        k_sconv_fwd_c1<<<stream_grid(total / 4, 256), 256, Cout * 16 * sizeof(float), st>>>(x, w_ohwi, bias, y, N, H, W, Ho, Wo, Cout,
                                                                                             stride, pad, slope);
    } else if (ks == 4 && Cout == 1 && Cin % 4 == 0 && (((uintptr_t)x | (uintptr_t)w_ohwi) & 15) == 0) {
        k_sconv_fwd_o1<<<stream_grid(total * 64, 256), 256, 0, st>>>(x, w_ohwi, bias, y, N, H, W, Ho, Wo, Cin, stride, pad, slope);
    } else {
        k_sconv_fwd<<<stream_grid(total, 256), 256, 0, st>>>(x, w_ohwi, bias, y, N, H, W, Ho, Wo, Cin, Cout, ks, stride, pad, slope);
    }
    VQW_LAUNCH_CHECK("vqw_sconv_fwd");
    return VQW_OK;
}

extern "C" size_t vqw_sconv_dgrad_ws_bytes(int N, int H, int W, int Cin, int Cout, int ks, int stride, int pad) {
    if (N <= 0 || H <= 0 || W <= 0 || !k4_mfma(N, H, W, Cin, Cout, ks, stride, pad)) return 0;
    size_t fl = (size_t)16 * Cin * Cout;                             // re-ordered weights
    if (stride == 1) fl += (size_t)N * H * W * Cout;                 // gy zero-padded to the H x W grid
    return fl * sizeof(float);
}
extern "C" int vqw_sconv_dgrad(const float* gy, const float* w_ohwi, float* gx, void* ws, size_t ws_bytes, int N, int H, int W,
                               int Cin, int Cout, int ks, int stride, int pad, void* stream) {
    int rc = check_sconv("vqw_sconv_dgrad", N, H, W, Cin, Cout, ks, stride, pad);
    if (rc) return rc;
    VQW_CHECK(gy && w_ohwi && gx, "vqw_sconv_dgrad: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int Ho = out_dim(H, ks, stride, pad), Wo = out_dim(W, ks, stride, pad);
    if (k4_mfma(N, H, W, Cin, Cout, ks, stride, pad)) {
        VQW_CHECK(ws && ws_bytes >= vqw_sconv_dgrad_ws_bytes(N, H, W, Cin, Cout, ks, stride, pad), "vqw_sconv_dgrad: workspace too small");
        float* wsf = (float*)ws;
        if (stride == 2) return conv_k4s2_dgrad(gy, w_ohwi, wsf, gx, N, Ho, Wo, Cin, Cout, st);
        float* gpad = wsf + (size_t)16 * Cin * Cout;
        rc = vqw_pack_dgrad_weights(w_ohwi, wsf, Cout, Cin, 4, stream);                   // wt[ci][15 - t][co]
        if (rc) return rc;
        rc = crop_pad(gy, gpad, N, Ho, Wo, H, W, Cout, st);
        if (rc) return rc;
        return conv_k4s1_grid(gpad, wsf, nullptr, gx, N, H, W, Cout, Cin, 2, st);
    }
    long total = (long)N * H * W * Cin;
    if (ks == 4 && Cin == 1 && Cout % 4 == 0 && Cout <= 1024 && (((uintptr_t)gy) & 15) == 0) {
        k_sconv_dgrad_c1<<<stream_grid(total, 256), 256, Cout * 16 * sizeof(float), st>>>(gy, w_ohwi, gx, N, H, W, Ho, Wo, Cout, stride, pad);
    } else if (ks == 4 && Cout == 1 && Cin % 4 == 0 && (((uintptr_t)gx | (uintptr_t)w_ohwi) & 15) == 0) {
        k_sconv_dgrad_o1<<<stream_grid(total / 4, 256), 256, 0, st>>>(gy, w_ohwi, gx, N, H, W, Ho, Wo, Cin, stride, pad);
    } else {
        k_sconv_dgrad<<<stream_grid(total, 256), 256, 0, st>>>(gy, w_ohwi, gx, N, H, W, Ho, Wo, Cin, Cout, ks, stride, pad);
    }
    VQW_LAUNCH_CHECK("vqw_sconv_dgrad");
    return VQW_OK;
}

static bool k4_mfma_wgrad(int N, int H, int W, int Cin, int Cout, int ks, int stride, int pad) {
    if (!k4_mfma(N, H, W, Cin, Cout, ks, stride, pad)) return false;
    return stride == 1 || conv_k4s2_wgrad_ok(Cin, Cout, N, H / 2, W / 2);
}
extern "C" size_t vqw_sconv_wgrad_ws_bytes(int Cin, int Cout, int ks, int N, int H, int W, int stride, int pad) {
    if (N <= 0 || H <= 0 || W <= 0 || ks <= 0 || stride <= 0) return 0;
    const long Po = (long)N * out_dim(H, ks, stride, pad) * out_dim(W, ks, stride, pad);
    size_t fl = bias_grad_ws_floats(Cout);
    if (k4_mfma_wgrad(N, H, W, Cin, Cout, ks, stride, pad)) {
        if (stride == 2) fl += conv_k4s2_wgrad_ws_floats(Cin, Cout, N, H / 2, W / 2);
        else fl += (size_t)N * H * W * Cout + conv_k4s1_wgrad_ws_floats(Cin, Cout, (long)N * H * W);
    } else if (wgrad_c1_ok(Cin, Cout, ks)) {
        fl += (size_t)wgrad_c1_splits(Po) * Cout * ks * ks;
    } else {
        fl += (size_t)wgrad_splits(Po) * Cout * ks * ks * Cin;
    }
    return fl * sizeof(float);
}

extern "C" int vqw_sconv_wgrad(const float* x, const float* gy, float* dw_ohwi, float* dbias, void* ws, size_t ws_bytes, int N,
                               int H, int W, int Cin, int Cout, int ks, int stride, int pad, int accumulate, void* stream) {
    int rc = check_sconv("vqw_sconv_wgrad", N, H, W, Cin, Cout, ks, stride, pad);
    if (rc) return rc;
    VQW_CHECK(x && gy && dw_ohwi && ws, "vqw_sconv_wgrad: null pointer");
    VQW_CHECK(ws_bytes >= vqw_sconv_wgrad_ws_bytes(Cin, Cout, ks, N, H, W, stride, pad), "vqw_sconv_wgrad: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int Ho = out_dim(H, ks, stride, pad), Wo = out_dim(W, ks, stride, pad);
    const long Po = (long)N * Ho * Wo;
    float* wsf = (float*)ws;
    if (dbias) {
        rc = bias_grad(gy, dbias, wsf, Po, Cout, st, accumulate);
        if (rc) return rc;
    }
    float* part = wsf + bias_grad_ws_floats(Cout);
    if (k4_mfma_wgrad(N, H, W, Cin, Cout, ks, stride, pad)) {
        if (stride == 2) return conv_k4s2_wgrad(x, gy, dw_ohwi, part, N, Ho, Wo, Cin, Cout, accumulate, st);
        float* gpad = part;
        rc = crop_pad(gy, gpad, N, Ho, Wo, H, W, Cout, st);
        if (rc) return rc;
        return conv_k4s1_wgrad_grid(x, gpad, dw_ohwi, part + (size_t)N * H * W * Cout, N, H, W, Cin, Cout, accumulate, st);
    }
    if (wgrad_c1_ok(Cin, Cout, ks)) {
        const int ns = wgrad_c1_splits(Po);
        k_sconv_wgrad_c1<16><<<ns, 256, 0, st>>>(x, gy, part, N, H, W, Ho, Wo, Cout, ks, stride, pad, ns);
        VQW_LAUNCH_CHECK("vqw_sconv_wgrad(c1)");
        return reduce_rows(part, dw_ohwi, (long)Cout * 16, ns, st, accumulate);
    }
    const int nsplit = wgrad_splits(Po);
    k_sconv_wgrad<<<Cout * ks * ks * nsplit, 256, 256 * sizeof(float), st>>>(x, gy, part, N, H, W, Ho, Wo, Cin, Cout, ks, stride, pad,
                                                                               nsplit);
    VQW_LAUNCH_CHECK("vqw_sconv_wgrad");
    return reduce_rows(part, dw_ohwi, (long)Cout * ks * ks * Cin, nsplit, st, accumulate);
}

extern "C" int vqw_leaky_relu_bwd(const float* y, const float* gy, float* gx, float slope, long n, void* stream) {
    VQW_CHECK(y && gy && gx && n > 0, "vqw_leaky_relu_bwd: bad arguments");
    k_leaky_bwd<<<stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(y, gy, gx, slope, n);
    VQW_LAUNCH_CHECK("vqw_leaky_relu_bwd");
    return VQW_OK;
}

extern "C" int vqw_hinge_fwd(const float* x, long n, int mode, float* loss, void* stream) {
    VQW_CHECK(x && loss && n > 0 && mode >= 0 && mode <= 2, "vqw_hinge_fwd: bad arguments");
    k_hinge_fwd<<<1, 1024, 0, (hipStream_t)stream>>>(x, n, mode, loss);
    VQW_LAUNCH_CHECK("vqw_hinge_fwd");
    return VQW_OK;
}
extern "C" int vqw_hinge_bwd(const float* x, long n, int mode, const float* gloss, float* gx, void* stream) {
    VQW_CHECK(x && gloss && gx && n > 0 && mode >= 0 && mode <= 2, "vqw_hinge_bwd: bad arguments");
    k_hinge_bwd<<<stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(x, n, mode, gloss, gx);
    VQW_LAUNCH_CHECK("vqw_hinge_bwd");
    return VQW_OK;
}
