// Element-wise, pooling, scalar-loss and optimiser kernels (HBM-bound; float4 streams).
#include "common.h"
#include "prof.h"
#include "../../include/vqwnet_hip.h"

static thread_local char g_err[512] = "";
extern "C" __attribute__((visibility("hidden"))) void vqw_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* vqw_last_error(void) { return g_err; }
extern "C" int vqw_abi_version(void) { return 8; }

// ---------------------------------------------------------------------------------------------
template <int RELU>
__global__ void k_add(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long n) {
    long n4 = n >> 2;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 u = ((const float4*)a)[i], v = ((const float4*)b)[i], r;
        r.x = u.x + v.x; r.y = u.y + v.y; r.z = u.z + v.z; r.w = u.w + v.w;
        if (RELU) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
        ((float4*)y)[i] = r;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float r = a[i] + b[i];
        y[i] = RELU ? fmaxf(r, 0.f) : r;
    }
}

extern "C" int vqw_add(const float* a, const float* b, float* y, long n, int relu, void* stream) {
    VQW_PROF_HBM(stream, 3, n);
    VQW_CHECK(a && b && y && n > 0, "vqw_add: bad arguments");
    VQW_CHECK((((uintptr_t)a | (uintptr_t)b | (uintptr_t)y) & 15) == 0, "vqw_add: pointers must be 16-byte aligned");
    int g = stream_grid((n + 3) / 4, 256);
    if (relu) k_add<1><<<g, 256, 0, (hipStream_t)stream>>>(a, b, y, n);
    else k_add<0><<<g, 256, 0, (hipStream_t)stream>>>(a, b, y, n);
    VQW_LAUNCH_CHECK("vqw_add");
    return VQW_OK;
}

__global__ void k_relu_bwd(const float* __restrict__ y, const float* __restrict__ gy, float* __restrict__ gx, long n) {
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        gx[i] = y[i] > 0.f ? gy[i] : 0.f;
}
extern "C" int vqw_relu_bwd(const float* y, const float* gy, float* gx, long n, void* stream) {
    VQW_PROF_HBM(stream, 3, n);
    VQW_CHECK(y && gy && gx && n > 0, "vqw_relu_bwd: bad arguments");
    k_relu_bwd<<<stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(y, gy, gx, n);
    VQW_LAUNCH_CHECK("vqw_relu_bwd");
    return VQW_OK;
}

// ---------------------------------------------------------------------------------------------
// MaxPool2d(2), NHWC.  Ties resolve to the first maximum in row-major window order, as ATen does.
__global__ void k_maxpool2_fwd(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
    int Ho = H >> 1, Wo = W >> 1;
    long total = (long)N * Ho * Wo * C;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int c = (int)(i % C);
        long p = i / C;
        int wo = (int)(p % Wo);
        long q = p / Wo;
        int ho = (int)(q % Ho);
        int n = (int)(q / Ho);
        const float* b = x + (((long)n * H + 2 * ho) * W + 2 * wo) * C + c;
        float m = b[0];
        float v = b[C];
        m = v > m ? v : m;
        v = b[(long)W * C];
        m = v > m ? v : m;
        v = b[(long)W * C + C];
        m = v > m ? v : m;
        y[i] = m;
    }
}
extern "C" int vqw_maxpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    VQW_PROF_HBM(stream, 1.25, (double)N * H * W * C);
    VQW_CHECK(x && y && N > 0 && C > 0 && H >= 2 && W >= 2, "vqw_maxpool2_fwd: bad arguments (N=%d H=%d W=%d C=%d)", N, H, W, C);
    long total = (long)N * (H / 2) * (W / 2) * C;
    k_maxpool2_fwd<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(x, y, N, H, W, C);
    VQW_LAUNCH_CHECK("vqw_maxpool2_fwd");
    return VQW_OK;
}

// gx (full resolution) = g_skip (optional) + gy routed to the arg-max of each 2x2 window.
// Rows/columns beyond 2*(H/2) (odd sizes) only receive g_skip.
__global__ void k_maxpool2_bwd(const float* __restrict__ x, const float* __restrict__ gy,
                               const float* __restrict__ gs, float* __restrict__ gx, int N, int H, int W, int C) {
    int Ho = H >> 1, Wo = W >> 1;
    long total = (long)N * H * W * C;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int c = (int)(i % C);
        long p = i / C;
        int w = (int)(p % W);
        long q = p / W;
        int h = (int)(q % H);
        int n = (int)(q / H);
        float g = gs ? gs[i] : 0.f;
        int ho = h >> 1, wo = w >> 1;
        if (ho < Ho && wo < Wo) {
            const float* b = x + (((long)n * H + 2 * ho) * W + 2 * wo) * C + c;
            float m = b[0];
            int am = 0;
            float v = b[C];
            if (v > m) { m = v; am = 1; }
            v = b[(long)W * C];
            if (v > m) { m = v; am = 2; }
            v = b[(long)W * C + C];
            if (v > m) { m = v; am = 3; }
            int me = ((h & 1) << 1) | (w & 1);
            if (am == me) g += gy[(((long)n * Ho + ho) * Wo + wo) * C + c];
        }
        gx[i] = g;
    }
}
extern "C" int vqw_maxpool2_bwd(const float* x, const float* gy, const float* g_skip, float* gx,
                                int N, int H, int W, int C, void* stream) {
    VQW_PROF_HBM(stream, 2.25, (double)N * H * W * C);
    VQW_CHECK(x && gy && gx && N > 0 && C > 0 && H >= 2 && W >= 2, "vqw_maxpool2_bwd: bad arguments");
    long total = (long)N * H * W * C;
    k_maxpool2_bwd<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(x, gy, g_skip, gx, N, H, W, C);
    VQW_LAUNCH_CHECK("vqw_maxpool2_bwd");
    return VQW_OK;
}

// Backward of the ResBlock tail  out = ReLU(a + b); pooled = MaxPool2d(2)(out)  (blocks.py:29-36) in one pass:
//   g = [out > 0] * (g_out + g_pooled routed to the arg-max of its 2x2 window),  da = db = g.
// Replaces max-pool backward + autograd's add of the two gradients of `out` + ReLU backward (three kernels, eight
// full-resolution tensor passes) by one kernel with three and a quarter.  Thread = one 2x2 window x 4 channels.
// Ties go to the first maximum in row-major window order like ATen (and k_maxpool2_bwd).  H, W even, C % 4 == 0.
__global__ void __launch_bounds__(256) k_res_tail_bwd4(const float4* __restrict__ out, const float4* __restrict__ gp,
                                                       const float4* __restrict__ go, float4* __restrict__ gx, int N, int H,
                                                       int W, int C4) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long total = (long)N * Ho * Wo * C4;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % C4);
        long p = i / C4;
        const int wo = (int)(p % Wo);
        p /= Wo;
        const int ho = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const long b = (((long)n * H + 2 * ho) * W + 2 * wo) * C4 + c;
        const long idx[4] = {b, b + C4, b + (long)W * C4, b + (long)W * C4 + C4};
        float4 v[4], g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = out[idx[k]];
        const float4 gy = gp ? gp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = go ? go[idx[k]] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float* vf = (const float*)v;
        float* gf = (float*)g;
        const float gyf[4] = {gy.x, gy.y, gy.z, gy.w};
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            float m = vf[ch];
            int am = 0;
#pragma unroll
            for (int k = 1; k < 4; ++k)
                if (vf[4 * k + ch] > m) { m = vf[4 * k + ch]; am = k; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float t = gf[4 * k + ch] + (k == am ? gyf[ch] : 0.f);
                gf[4 * k + ch] = vf[4 * k + ch] > 0.f ? t : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) gx[idx[k]] = g[k];
    }
}
// forward of the same tail: out = ReLU(a + b) and pooled = max over each 2x2 window, one pass (the separate operators
// read `out` back for the pooling)
__global__ void __launch_bounds__(256) k_res_tail_fwd4(const float4* __restrict__ a, const float4* __restrict__ b,
                                                       float4* __restrict__ out, float4* __restrict__ pooled, int N, int H, int W,
                                                       int C4) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long total = (long)N * Ho * Wo * C4;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % C4);
        long p = i / C4;
        const int wo = (int)(p % Wo);
        p /= Wo;
        const int ho = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const long base = (((long)n * H + 2 * ho) * W + 2 * wo) * C4 + c;
        const long idx[4] = {base, base + C4, base + (long)W * C4, base + (long)W * C4 + C4};
        float4 m;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 x = a[idx[k]], y = b[idx[k]];
            float4 o;
            o.x = fmaxf(x.x + y.x, 0.f); o.y = fmaxf(x.y + y.y, 0.f); o.z = fmaxf(x.z + y.z, 0.f); o.w = fmaxf(x.w + y.w, 0.f);
            out[idx[k]] = o;
            if (k == 0) m = o;
            else { m.x = fmaxf(m.x, o.x); m.y = fmaxf(m.y, o.y); m.z = fmaxf(m.z, o.z); m.w = fmaxf(m.w, o.w); }
        }
        pooled[i] = m;
    }
}
extern "C" int vqw_res_tail_fwd(const float* a, const float* b, float* out, float* pooled, int N, int H, int W, int C,
                                void* stream) {
    VQW_PROF_HBM(stream, 3.25, (double)N * H * W * C);
    VQW_CHECK(a && b && out && pooled && N > 0 && C > 0, "vqw_res_tail_fwd: bad arguments");
    VQW_CHECK((H & 1) == 0 && (W & 1) == 0 && (C & 3) == 0 && H >= 2 && W >= 2, "vqw_res_tail_fwd: needs even H, W and C %% 4 == 0");
    VQW_CHECK(((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out | (uintptr_t)pooled) & 15) == 0), "vqw_res_tail_fwd: 16-byte alignment");
    const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
    k_res_tail_fwd4<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>((const float4*)a, (const float4*)b, (float4*)out,
                                                                               (float4*)pooled, N, H, W, C / 4);
    VQW_LAUNCH_CHECK("vqw_res_tail_fwd");
    return VQW_OK;
}
// The ResBlock tail reading the two RAW conv outputs and their InstanceNorm statistics (blocks.py:25-36):
//   a = ReLU(IN(x2)), b = IN(xid), out = ReLU(a + b), pooled = MaxPool2d(2)(out)
// -- the two normalisation apply passes (read + write of a full-resolution tensor each) are gone.
__global__ void __launch_bounds__(256) k_res_tail_norm_fwd4(const float4* __restrict__ x2, const float* __restrict__ mr2,
                                                            const float4* __restrict__ xid, const float* __restrict__ mrid,
                                                            float4* __restrict__ out, float4* __restrict__ pooled, int N, int H,
                                                            int W, int C4) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long total = (long)N * Ho * Wo * C4;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % C4);
        long p = i / C4;
        const int wo = (int)(p % Wo);
        p /= Wo;
        const int ho = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const long base = (((long)n * H + 2 * ho) * W + 2 * wo) * C4 + c;
        const long idx[4] = {base, base + C4, base + (long)W * C4, base + (long)W * C4 + C4};
        const float4* ma = (const float4*)(mr2 + 2 * ((long)n * C4 * 4 + c * 4));
        const float4* mb = (const float4*)(mrid + 2 * ((long)n * C4 * 4 + c * 4));
        const float4 a0 = ma[0], a1 = ma[1], b0 = mb[0], b1 = mb[1];       // (mean, rstd) pairs of channels 4c..4c+3
        float4 m;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 x = x2[idx[k]], y = xid[idx[k]];
            float4 u, v, o;
            u.x = fmaxf((x.x - a0.x) * a0.y, 0.f); u.y = fmaxf((x.y - a0.z) * a0.w, 0.f);
            u.z = fmaxf((x.z - a1.x) * a1.y, 0.f); u.w = fmaxf((x.w - a1.z) * a1.w, 0.f);
            v.x = (y.x - b0.x) * b0.y; v.y = (y.y - b0.z) * b0.w; v.z = (y.z - b1.x) * b1.y; v.w = (y.w - b1.z) * b1.w;
            o.x = fmaxf(u.x + v.x, 0.f); o.y = fmaxf(u.y + v.y, 0.f); o.z = fmaxf(u.z + v.z, 0.f); o.w = fmaxf(u.w + v.w, 0.f);
            out[idx[k]] = o;
            if (k == 0) m = o;
            else { m.x = fmaxf(m.x, o.x); m.y = fmaxf(m.y, o.y); m.z = fmaxf(m.z, o.z); m.w = fmaxf(m.w, o.w); }
        }
        pooled[i] = m;
    }
}
extern "C" int vqw_res_tail_norm_fwd(const float* x2, const float* mr2, const float* xid, const float* mrid, float* out,
                                     float* pooled, int N, int H, int W, int C, void* stream) {
    VQW_PROF_HBM(stream, 3.25, (double)N * H * W * C);
    VQW_CHECK(x2 && mr2 && xid && mrid && out && pooled && N > 0 && C > 0, "vqw_res_tail_norm_fwd: bad arguments");
    VQW_CHECK((H & 1) == 0 && (W & 1) == 0 && (C & 3) == 0 && H >= 2 && W >= 2, "vqw_res_tail_norm_fwd: needs even H, W and C %% 4 == 0");
    VQW_CHECK(((((uintptr_t)x2 | (uintptr_t)xid | (uintptr_t)out | (uintptr_t)pooled | (uintptr_t)mr2 | (uintptr_t)mrid) & 15) == 0),
              "vqw_res_tail_norm_fwd: 16-byte alignment");
    const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
    k_res_tail_norm_fwd4<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>((const float4*)x2, mr2, (const float4*)xid, mrid,
                                                                                    (float4*)out, (float4*)pooled, N, H, W, C / 4);
    VQW_LAUNCH_CHECK("vqw_res_tail_norm_fwd");
    return VQW_OK;
}
extern "C" int vqw_res_tail_bwd(const float* out, const float* g_pooled, const float* g_out, float* gx, int N, int H, int W,
                                int C, void* stream) {
    VQW_PROF_HBM(stream, 3.25, (double)N * H * W * C);
    VQW_CHECK(out && gx && (g_pooled || g_out) && N > 0 && C > 0, "vqw_res_tail_bwd: bad arguments");
    VQW_CHECK((H & 1) == 0 && (W & 1) == 0 && (C & 3) == 0 && H >= 2 && W >= 2, "vqw_res_tail_bwd: needs even H, W and C %% 4 == 0");
    VQW_CHECK(((((uintptr_t)out | (uintptr_t)g_pooled | (uintptr_t)g_out | (uintptr_t)gx) & 15) == 0), "vqw_res_tail_bwd: 16-byte alignment");
    const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
    k_res_tail_bwd4<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>((const float4*)out, (const float4*)g_pooled,
                                                                               (const float4*)g_out, (float4*)gx, N, H, W, C / 4);
    VQW_LAUNCH_CHECK("vqw_res_tail_bwd");
    return VQW_OK;
}

// ---------------------------------------------------------------------------------------------
__global__ void k_tanh_fwd(const float* __restrict__ x, float* __restrict__ y, long n) {
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = tanhf(x[i]);
}
__global__ void k_tanh_bwd(const float* __restrict__ y, const float* __restrict__ gy, float* __restrict__ gx, long n) {
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float t = y[i];
        gx[i] = gy[i] * (1.f - t * t);
    }
}
__global__ void k_affine(const float* __restrict__ x, float* __restrict__ y, float a, float b, long n) {
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = x[i] * a + b;
}
extern "C" int vqw_tanh_fwd(const float* x, float* y, long n, void* stream) {
    VQW_CHECK(x && y && n > 0, "vqw_tanh_fwd: bad arguments");
    k_tanh_fwd<<<stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(x, y, n);
    VQW_LAUNCH_CHECK("vqw_tanh_fwd");
    return VQW_OK;
}
extern "C" int vqw_tanh_bwd(const float* y, const float* gy, float* gx, long n, void* stream) {
    VQW_CHECK(y && gy && gx && n > 0, "vqw_tanh_bwd: bad arguments");
    k_tanh_bwd<<<stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(y, gy, gx, n);
    VQW_LAUNCH_CHECK("vqw_tanh_bwd");
    return VQW_OK;
}
extern "C" int vqw_affine(const float* x, float* y, float scale, float shift, long n, void* stream) {
    VQW_CHECK(x && y && n > 0, "vqw_affine: bad arguments");
    k_affine<<<stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(x, y, scale, shift, n);
    VQW_LAUNCH_CHECK("vqw_affine");
    return VQW_OK;
}

// ---------------------------------------------------------------------------------------------
// mean squared error, deterministic two-stage reduction (double partials)
#define RED_BLOCKS 1024
extern "C" size_t vqw_reduce_ws_bytes(long n) { (void)n; return RED_BLOCKS * sizeof(double); }

__global__ void k_mse_partial(const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ part, long n) {
    __shared__ double sm[4];
    double acc = 0.0;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float d = a[i] - b[i];
        acc += (double)(d * d);
    }
    acc = wave_sum_d(acc);
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sm[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
__global__ void k_sum_finalize(const double* __restrict__ part, int nparts, double scale, float* __restrict__ out) {
    __shared__ double sm[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) acc += part[i];
    acc = wave_sum_d(acc);
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sm[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((sm[0] + sm[1] + sm[2] + sm[3]) * scale);
}
extern "C" int vqw_mse_fwd(const float* a, const float* b, float* loss, void* ws, size_t ws_bytes, long n, void* stream) {
    VQW_CHECK(a && b && loss && ws && n > 0, "vqw_mse_fwd: bad arguments");
    VQW_CHECK(ws_bytes >= vqw_reduce_ws_bytes(n), "vqw_mse_fwd: workspace too small");
    int g = imin(RED_BLOCKS, stream_grid(n, 256));
    k_mse_partial<<<g, 256, 0, (hipStream_t)stream>>>(a, b, (double*)ws, n);
    k_sum_finalize<<<1, 256, 0, (hipStream_t)stream>>>((const double*)ws, g, 1.0 / (double)n, loss);
    VQW_LAUNCH_CHECK("vqw_mse_fwd");
    return VQW_OK;
}
__global__ void k_mse_bwd(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gl,
                          float* __restrict__ ga, long n, float inv_n2) {
    float s = gl[0] * inv_n2;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) ga[i] = (a[i] - b[i]) * s;
}
extern "C" int vqw_mse_bwd(const float* a, const float* b, const float* gloss, float* ga, long n, void* stream) {
    VQW_CHECK(a && b && gloss && ga && n > 0, "vqw_mse_bwd: bad arguments");
    k_mse_bwd<<<stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(a, b, gloss, ga, n, 2.0f / (float)n);
    VQW_LAUNCH_CHECK("vqw_mse_bwd");
    return VQW_OK;
}

// Windowed MSE (trainers/multi_window_trainer.py:93-109 with base.py:290-314): both tensors are re-windowed before the
// squared error, w(x) = clamp(alpha * x + beta, lo, hi) — the reference's denormalize (dataset window) followed by
// normalize (lung / mediastinal window) folded into one affine map and one clamp.  Gradient is zero where clamped.
__device__ __forceinline__ float win_map(float x, float alpha, float beta, float lo, float hi) {
    return fminf(fmaxf(alpha * x + beta, lo), hi);
}
__global__ void k_window_mse_partial(const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ part, long n,
                                     float alpha, float beta, float lo, float hi) {
    __shared__ double sm[256];
    double acc = 0.0;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float d = win_map(a[i], alpha, beta, lo, hi) - win_map(b[i], alpha, beta, lo, hi);
        acc += (double)(d * d);
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}
extern "C" int vqw_window_mse_fwd(const float* a, const float* b, float* loss, void* ws, size_t ws_bytes, long n, float alpha,
                                  float beta, float lo, float hi, void* stream) {
    VQW_CHECK(a && b && loss && ws && n > 0 && lo <= hi, "vqw_window_mse_fwd: bad arguments");
    VQW_CHECK(ws_bytes >= vqw_reduce_ws_bytes(n), "vqw_window_mse_fwd: workspace too small");
    int g = imin(RED_BLOCKS, stream_grid(n, 256));
    k_window_mse_partial<<<g, 256, 0, (hipStream_t)stream>>>(a, b, (double*)ws, n, alpha, beta, lo, hi);
    k_sum_finalize<<<1, 256, 0, (hipStream_t)stream>>>((const double*)ws, g, 1.0 / (double)n, loss);
    VQW_LAUNCH_CHECK("vqw_window_mse_fwd");
    return VQW_OK;
}
__global__ void k_window_mse_bwd(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gl,
                                 float* __restrict__ ga, long n, float inv_n2, float alpha, float beta, float lo, float hi) {
    float s = gl[0] * inv_n2 * alpha;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float z = alpha * a[i] + beta;
        float d = fminf(fmaxf(z, lo), hi) - win_map(b[i], alpha, beta, lo, hi);
        ga[i] = (z > lo && z < hi) ? d * s : 0.f;
    }
}
extern "C" int vqw_window_mse_bwd(const float* a, const float* b, const float* gloss, float* ga, long n, float alpha, float beta,
                                  float lo, float hi, void* stream) {
    VQW_CHECK(a && b && gloss && ga && n > 0, "vqw_window_mse_bwd: bad arguments");
    k_window_mse_bwd<<<stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(a, b, gloss, ga, n, 2.0f / (float)n, alpha, beta, lo, hi);
    VQW_LAUNCH_CHECK("vqw_window_mse_bwd");
    return VQW_OK;
}

__global__ void k_weighted_sum(const float* const* __restrict__ terms, const float* __restrict__ w, int n, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += w[i] * terms[i][0];
        out[0] = s;
    }
}
extern "C" int vqw_weighted_sum(const float* const* terms_dev, const float* weights_dev, int n_terms, float* out, void* stream) {
    VQW_CHECK(terms_dev && weights_dev && out && n_terms > 0 && n_terms <= 64, "vqw_weighted_sum: bad arguments");
    k_weighted_sum<<<1, 64, 0, (hipStream_t)stream>>>(terms_dev, weights_dev, n_terms, out);
    VQW_LAUNCH_CHECK("vqw_weighted_sum");
    return VQW_OK;
}

// The same with the term pointers and weights as KERNEL ARGUMENTS (host arrays, at most 16 terms): no device-side table,
// so no host-to-device copy per call (a pinned staging buffer per call made the host allocator — and now and then the
// whole step — wait for the device).
struct WsumArgs {
    const float* t[16];
    float w[16];
    int n;
};
__global__ void k_weighted_sum_args(WsumArgs a, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < a.n; ++i) s += a.w[i] * a.t[i][0];
        out[0] = s;
    }
}
extern "C" int vqw_weighted_sum_host(const float* const* terms, const float* weights, int n_terms, float* out, void* stream) {
    VQW_CHECK(terms && weights && out && n_terms > 0 && n_terms <= 16, "vqw_weighted_sum_host: 1..16 terms");
    WsumArgs a;
    for (int i = 0; i < 16; ++i) { a.t[i] = i < n_terms ? terms[i] : nullptr; a.w[i] = i < n_terms ? weights[i] : 0.f; }
    a.n = n_terms;
    k_weighted_sum_args<<<1, 64, 0, (hipStream_t)stream>>>(a, out);
    VQW_LAUNCH_CHECK("vqw_weighted_sum_host");
    return VQW_OK;
}

// ---------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam semantics: L2 decay folded into the gradient; eps added after the
// bias-corrected sqrt).  28 B/param of HBM traffic: read p,g,m,v; write p,m,v.
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                       long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    long stride = (long)gridDim.x * blockDim.x;
    float step = lr / bc1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float gi = g[i], pi = p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        float mi = m[i] * b1 + (1.f - b1) * gi;
        float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step * (mi / denom);
        m[i] = mi;
        v[i] = vi;
    }
}
// One launch for a whole parameter group (all tensors share the step count, hence the bias corrections): the table holds
// one entry per chunk of a tensor; same per-element arithmetic as k_adam.
struct AdamChunk {
    float* p;
    const float* g;
    float* m;
    float* v;
    long n;
};
__global__ void __launch_bounds__(256) k_adam_multi(const AdamChunk* __restrict__ chunks, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2_sqrt) {
    const AdamChunk c = chunks[blockIdx.x];
    const float step = lr / bc1;
    // four elements per lane where the chunk allows (same per-element arithmetic): 16-byte loads / stores, a quarter of the
    // memory instructions of the scalar walk
    long i0 = 0;
    if ((((uintptr_t)c.p | (uintptr_t)c.g | (uintptr_t)c.m | (uintptr_t)c.v) & 15) == 0) {
        const long n4 = c.n >> 2;
        float4* p4 = (float4*)c.p; const float4* g4 = (const float4*)c.g; float4* m4 = (float4*)c.m; float4* v4 = (float4*)c.v;
        for (long i = threadIdx.x; i < n4; i += 256) {
            const float4 gq = g4[i], pq = p4[i], mq = m4[i], vq = v4[i];
            float gi[4] = {gq.x, gq.y, gq.z, gq.w}, pi[4] = {pq.x, pq.y, pq.z, pq.w}, mi[4] = {mq.x, mq.y, mq.z, mq.w}, vi[4] = {vq.x, vq.y, vq.z, vq.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (wd != 0.f) gi[k] = fmaf(wd, pi[k], gi[k]);
                mi[k] = mi[k] * b1 + (1.f - b1) * gi[k];
                vi[k] = vi[k] * b2 + (1.f - b2) * gi[k] * gi[k];
                const float denom = sqrtf(vi[k]) / bc2_sqrt + eps;
                pi[k] = pi[k] - step * (mi[k] / denom);
            }
            p4[i] = make_float4(pi[0], pi[1], pi[2], pi[3]);
            m4[i] = make_float4(mi[0], mi[1], mi[2], mi[3]);
            v4[i] = make_float4(vi[0], vi[1], vi[2], vi[3]);
        }
        i0 = n4 << 2;
    }
    for (long i = i0 + threadIdx.x; i < c.n; i += 256) {
        float gi = c.g[i], pi = c.p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        float mi = c.m[i] * b1 + (1.f - b1) * gi;
        float vi = c.v[i] * b2 + (1.f - b2) * gi * gi;
        float denom = sqrtf(vi) / bc2_sqrt + eps;
        c.p[i] = pi - step * (mi / denom);
        c.m[i] = mi;
        c.v[i] = vi;
    }
}
extern "C" int vqw_adam_multi(const void* chunks_dev, int n_chunks, float lr, float beta1, float beta2, float eps,
                              float weight_decay, float bias_corr1, float bias_corr2, void* stream) {
    VQW_CHECK(chunks_dev && n_chunks > 0, "vqw_adam_multi: bad arguments");
    VQW_CHECK(bias_corr1 > 0.f && bias_corr2 > 0.f, "vqw_adam_multi: bias corrections must be positive");
    static_assert(sizeof(AdamChunk) == 40, "chunk table layout is part of the ABI: 4 pointers + int64 count");
    k_adam_multi<<<n_chunks, 256, 0, (hipStream_t)stream>>>((const AdamChunk*)chunks_dev, lr, beta1, beta2, eps, weight_decay,
                                                            bias_corr1, sqrtf(bias_corr2));
    VQW_LAUNCH_CHECK("vqw_adam_multi");
    return VQW_OK;
}

extern "C" int vqw_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, float bias_corr1, float bias_corr2, void* stream) {
    VQW_PROF_HBM(stream, 7, n);
    VQW_CHECK(p && g && m && v && n > 0, "vqw_adam_step: bad arguments");
    VQW_CHECK(bias_corr1 > 0.f && bias_corr2 > 0.f, "vqw_adam_step: bias corrections must be positive");
    k_adam<<<stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                                                                 bias_corr1, sqrtf(bias_corr2));
    VQW_LAUNCH_CHECK("vqw_adam_step");
    return VQW_OK;
}
