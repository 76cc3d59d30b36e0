// InstanceNorm2d(+ReLU) and StyledDenorm (BatchNorm2d(affine=False) * (1+gamma) + beta) kernels, NHWC fp32.
//
// All statistics use a deterministic two-stage reduction: stage 1 writes double partials
// part[n][split][c][2] (fixed pixel ranges per block, fixed tree inside the block), stage 2 sums the
// partials in index order.  No float atomics -> bitwise reproducible run to run.
#include "common.h"
#include "prof.h"
#include "../../include/vqwnet_hip.h"

#define PLANE_MAX_SPLITS 64

static inline int plane_splits(int N, int HW) {
    // enough blocks to fill 256 CUs a few times over, but >= 256 pixels per block
    int s = ceil_div(2048, N);
    int cap = imax(1, HW / 256);
    s = imin(imin(s, cap), PLANE_MAX_SPLITS);
    return imax(s, 1);
}

static inline size_t plane_part_bytes(int N, int C) { return (size_t)N * PLANE_MAX_SPLITS * C * 2 * sizeof(double); }
// partial sums [N][MAX_SPLITS][C][2] doubles, followed by [N][C][2] floats of finalised means
extern "C" size_t vqw_plane_ws_bytes(int N, int C, int HW) {
    (void)HW;
    return plane_part_bytes(N, C) + (size_t)N * C * 2 * sizeof(float);
}

// ---------------------------------------------------------------------------------------------
// Generic per-(n,c) two-quantity plane reduction.  Functor F: (x-element index i, n, c) -> (a, b).
// Block layout: tc = threads along channels (min(C,256)), rows = 256/tc pixel rows per pass.
template <class F>
__global__ void __launch_bounds__(256) k_plane_reduce(F f, double* __restrict__ part, int HW, int C, int splits) {
    __shared__ double sa[256], sb[256];
    const int n = blockIdx.y, s = blockIdx.x;
    const int tcn = C < 256 ? C : 256;
    const int rows = 256 / tcn;
    const int t = threadIdx.x;
    const int tc = t % tcn, tr = t / tcn;
    const int per = (HW + splits - 1) / splits;
    const int p0 = s * per;
    const int p1 = (p0 + per < HW) ? p0 + per : HW;
    const bool active = tr < rows;
    for (int cb = 0; cb < C; cb += tcn) {
        const int c = cb + tc;
        double a = 0.0, b = 0.0;
        if (active && c < C) {
            for (int p = p0 + tr; p < p1; p += rows) {
                long i = ((long)n * HW + p) * C + c;
                float va, vb;
                f(i, n, c, va, vb);
                a += (double)va;
                b += (double)vb;
            }
        }
        sa[t] = a;
        sb[t] = b;
        __syncthreads();
        if (tr == 0 && c < C) {
            for (int r = 1; r < rows; ++r) { a += sa[r * tcn + tc]; b += sb[r * tcn + tc]; }
            double* o = part + (((long)n * splits + s) * C + c) * 2;
            o[0] = a;
            o[1] = b;
        }
        __syncthreads();
    }
}

// float4 variant for C % 4 == 0: a lane owns 4 consecutive channels, C/4 lanes span a pixel, 256/(C/4) pixel rows per
// pass (1 KiB contiguous per wave-instruction).  Functor F4: (float4 index i4, n, channel c) -> a[4], b[4].
// (round 4: at least four waves per SIMD - the compiler's 214-register schedule left two, ~64 KB of loads in flight per CU,
// and the pure-read passes ran at 3.6-4.1 TB/s; a group of four elements is summed in fp32 before it joins the double
// accumulators unless the functor asks for doubles throughout (F4::kDoubleTree: the forward statistics, where E[x^2] - mean^2
// cancels))
template <class F4>
__global__ void __launch_bounds__(256, F4::kMinWaves) k_plane_reduce4(F4 f, double* __restrict__ part, int HW, int C, int splits) {
    __shared__ double sa[4][256], sb[4][256];
    const int n = blockIdx.y, s = blockIdx.x;
    const int C4 = C >> 2;
    const int tcn = C4 < 256 ? C4 : 256;
    const int rows = 256 / tcn;
    const int t = threadIdx.x;
    const int tc = t % tcn, tr = t / tcn;
    const int per = (HW + splits - 1) / splits;
    const int p0 = s * per;
    const int p1 = (p0 + per < HW) ? p0 + per : HW;
    const bool active = tr < rows;
    for (int cb = 0; cb < C4; cb += tcn) {
        const int c4 = cb + tc;
        double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
        if (active && c4 < C4) {
            int p = p0 + tr;
            for (; p + 3 * rows < p1; p += 4 * rows) {  // four independent element loads in flight per lane
                float va[4], vb[4], wa[4], wb[4], xa[4], xb[4], ya[4], yb[4];
                const long q = (long)n * HW + p;
                f(q * C4 + c4, q, n, c4 * 4, va, vb);
                f((q + rows) * C4 + c4, q + rows, n, c4 * 4, wa, wb);
                f((q + 2 * rows) * C4 + c4, q + 2 * rows, n, c4 * 4, xa, xb);
                f((q + 3 * rows) * C4 + c4, q + 3 * rows, n, c4 * 4, ya, yb);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (F4::kDoubleTree) {
                        a[k] += ((double)va[k] + (double)wa[k]) + ((double)xa[k] + (double)ya[k]);
                        b[k] += ((double)vb[k] + (double)wb[k]) + ((double)xb[k] + (double)yb[k]);
                    } else {
                        a[k] += (double)((va[k] + wa[k]) + (xa[k] + ya[k]));
                        b[k] += (double)((vb[k] + wb[k]) + (xb[k] + yb[k]));
                    }
                }
            }
            for (; p + rows < p1; p += 2 * rows) {
                float va[4], vb[4], wa[4], wb[4];
                const long q = (long)n * HW + p;
                f(q * C4 + c4, q, n, c4 * 4, va, vb);
                f((q + rows) * C4 + c4, q + rows, n, c4 * 4, wa, wb);
#pragma unroll
                for (int k = 0; k < 4; ++k) { a[k] += (double)va[k] + (double)wa[k]; b[k] += (double)vb[k] + (double)wb[k]; }
            }
            for (; p < p1; p += rows) {
                float va[4], vb[4];
                const long q = (long)n * HW + p;
                f(q * C4 + c4, q, n, c4 * 4, va, vb);
#pragma unroll
                for (int k = 0; k < 4; ++k) { a[k] += (double)va[k]; b[k] += (double)vb[k]; }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { sa[k][t] = a[k]; sb[k][t] = b[k]; }
        __syncthreads();
        if (tr == 0 && c4 < C4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double ta = a[k], tb = b[k];
                for (int r = 1; r < rows; ++r) { ta += sa[k][r * tcn + tc]; tb += sb[k][r * tcn + tc]; }
                double* o = part + (((long)n * splits + s) * C + c4 * 4 + k) * 2;
                o[0] = ta;
                o[1] = tb;
            }
        }
        __syncthreads();
    }
}

// Pipelined form of k_plane_reduce4 for C4 = C / 4 a power of two <= 256 (round 4).  The form above lives too short: a
// workgroup streams 8 elements per thread in two dependent load-wait-add rounds, then pays two barriers and a serial LDS fold
// by C4 of its 256 threads - loads are in flight ~60 % of a wave's life and the pure-read passes ran at 3.6-4.4 TB/s beside
// 5.8-6.5 for the apply passes.  Here a workgroup walks a longer pixel range (splits sized for ~4 workgroups per CU), the
// NEXT group of four pixels is loaded before the current one is evaluated (two register sets), four elements are summed in
// fp32 before they join the double accumulators (functors that need doubles throughout keep them), and the fold over the
// pixel rows of a workgroup runs through wave shuffles, then one small LDS stage over the four waves.
// Thread t: channel quad c4 = t & (C4 - 1), pixel row r = t >> lgC4 of R = 256 / C4; lanes of a wave that share c4 differ
// in the bits >= lgC4 of the lane number.
template <class F4>
__global__ void __launch_bounds__(256, F4::kMinWaves) k_plane_reduce4p(F4 f, double* __restrict__ part, int HW, int C4, int lgC4,
                                                                       int splits) {
    __shared__ double sm[4][8][64];            // [wave][a0..a3, b0..b3][lane]
    const int n = blockIdx.y, s = blockIdx.x, t = threadIdx.x;
    const int c4 = t & (C4 - 1), r = t >> lgC4, R = 256 >> lgC4;
    const int per = (HW + splits - 1) / splits;
    const int p0 = s * per;
    const int p1 = (p0 + per < HW) ? p0 + per : HW;
    float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f), m1 = m0;
    f.consts(n, c4, m0, m1);
    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
    const long nb = (long)n * HW;
    typename F4::Raw cur[4], nxt[4];
    int p = p0 + r;
    bool have = p + 3 * R < p1;
    if (have) {
#pragma unroll
        for (int u = 0; u < 4; ++u) f.load((nb + p + u * R) * C4 + c4, nb + p + u * R, c4, cur[u]);
    }
    while (have) {
        const int pn = p + 4 * R;
        const bool more = pn + 3 * R < p1;
        if (more) {
#pragma unroll
            for (int u = 0; u < 4; ++u) f.load((nb + pn + u * R) * C4 + c4, nb + pn + u * R, c4, nxt[u]);
        }
        float ea[4][4], eb[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) f.eval(cur[u], nb + p + u * R, c4, m0, m1, ea[u], eb[u]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (F4::kDoubleTree) {
                a[k] += ((double)ea[0][k] + (double)ea[1][k]) + ((double)ea[2][k] + (double)ea[3][k]);
                b[k] += ((double)eb[0][k] + (double)eb[1][k]) + ((double)eb[2][k] + (double)eb[3][k]);
            } else {
                a[k] += (double)((ea[0][k] + ea[1][k]) + (ea[2][k] + ea[3][k]));
                b[k] += (double)((eb[0][k] + eb[1][k]) + (eb[2][k] + eb[3][k]));
            }
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
        }
        p = pn;
        have = more;
    }
    for (; p < p1; p += R) {                    // remainder: fewer than four pixel rows left
        typename F4::Raw one;
        float ea[4], eb[4];
        f.load((nb + p) * C4 + c4, nb + p, c4, one);
        f.eval(one, nb + p, c4, m0, m1, ea, eb);
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[k] += (double)ea[k]; b[k] += (double)eb[k]; }
    }
    // fold over the pixel rows: lanes that share c4 inside a wave (xor offsets C4, 2 C4, ... < 64), then the four waves
    const int lane = t & 63, w = t >> 6;
    for (int off = C4; off < 64; off <<= 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[k] += __shfl_xor(a[k], off, 64); b[k] += __shfl_xor(b[k], off, 64); }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { sm[w][k][lane] = a[k]; sm[w][4 + k][lane] = b[k]; }
    __syncthreads();
    // C4 <= 64: every wave holds every channel quad in its lanes 0..C4-1; C4 = 128 / 256: quad c4 lives in lane c4 & 63 of
    // the waves w with (w * 64 + lane) & (C4 - 1) == c4
    if (t < C4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double ta = 0.0, tb = 0.0;
            if (C4 <= 64) {
#pragma unroll
                for (int v = 0; v < 4; ++v) { ta += sm[v][k][t]; tb += sm[v][4 + k][t]; }
            } else {
                for (int v = t >> 6; v < 4; v += C4 >> 6) { ta += sm[v][k][t & 63]; tb += sm[v][4 + k][t & 63]; }
            }
            double* o = part + (((long)n * splits + s) * (4 * C4) + 4 * t + k) * 2;
            o[0] = ta;
            o[1] = tb;
        }
    }
}
// splits of the pipelined form: ~1024 workgroups in all, at least 8 pixel-row groups (32 R pixels) each, at most one lane
// per split in the finalise kernels
static inline int plane_splits_p(int N, int HW, int C4) {
    const int R = 256 / C4;
    int s = ceil_div(1024, N);
    s = imin(s, imax(1, HW / (32 * R)));
    s = imin(s, PLANE_MAX_SPLITS);
    return imax(s, 1);
}

// ---------------------------------------------------------------------------------------------
// InstanceNorm
struct FStats4 {
    static constexpr bool kDoubleTree = true;
    static constexpr int kMinWaves = 4;
    const float4* x;
    struct Raw { float4 v; };
    __device__ void consts(int, int, float4&, float4&) const {}
    __device__ void load(long i4, long, int, Raw& r) const { r.v = x[i4]; }
    __device__ void eval(const Raw& r, long, int, const float4&, const float4&, float* a, float* b) const {
        a[0] = r.v.x; a[1] = r.v.y; a[2] = r.v.z; a[3] = r.v.w;
        b[0] = r.v.x * r.v.x; b[1] = r.v.y * r.v.y; b[2] = r.v.z * r.v.z; b[3] = r.v.w * r.v.w;
    }
    __device__ void operator()(long i4, long, int, int, float* a, float* b) const {
        float4 v = x[i4];
        a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
        b[0] = v.x * v.x; b[1] = v.y * v.y; b[2] = v.z * v.z; b[3] = v.w * v.w;
    }
};
template <int RELU>
struct FInBwd4 {
    static constexpr bool kDoubleTree = false;
    static constexpr int kMinWaves = 4;
    const float4* x;
    const float* mr;
    const float4* gy;
    int C, gcs4, gco4;
    struct Raw { float4 v, g; };
    __device__ void consts(int n, int c4, float4& m0, float4& m1) const {
        const float4* m = (const float4*)(mr + 2 * ((long)n * C + 4 * c4));
        m0 = m[0]; m1 = m[1];
    }
    __device__ void load(long i4, long pix, int c4, Raw& r) const { r.v = x[i4]; r.g = gy[pix * gcs4 + gco4 + c4]; }
    __device__ void eval(const Raw& r, long, int, const float4& m0, const float4& m1, float* a, float* b) const {
        const float xh[4] = {(r.v.x - m0.x) * m0.y, (r.v.y - m0.z) * m0.w, (r.v.z - m1.x) * m1.y, (r.v.w - m1.z) * m1.w};
        const float gg[4] = {r.g.x, r.g.y, r.g.z, r.g.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float q = (RELU && !(xh[k] > 0.f)) ? 0.f : gg[k];
            a[k] = q;
            b[k] = q * xh[k];
        }
    }
    // i4 = (pix * C4 + c4) with pix = n * HW + p the pixel of the batch, c = 4 * c4 (no integer division per element)
    __device__ void operator()(long i4, long pix, int n, int c, float* a, float* b) const {
        const float4* m = (const float4*)(mr + 2 * ((long)n * C + c));
        float4 m0 = m[0], m1 = m[1], v = x[i4], g = gy[pix * gcs4 + gco4 + (c >> 2)];
        float xh[4] = {(v.x - m0.x) * m0.y, (v.y - m0.z) * m0.w, (v.z - m1.x) * m1.y, (v.w - m1.z) * m1.w};
        float gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float q = (RELU && !(xh[k] > 0.f)) ? 0.f : gg[k];
            a[k] = q;
            b[k] = q * xh[k];
        }
    }
};
template <int RELU>
struct FSpadeBwd4 {
    static constexpr bool kDoubleTree = false;
    static constexpr int kMinWaves = 2;           // four tensors read per element, two register sets of four elements: 128 float4 registers in flight
    const float4* x;
    const float* mr;
    const float4* gamma;
    const float4* beta;
    const float4* gy;
    float4* dgamma;
    float4* dbeta;
    int C4, gbs4;     // channels / 4, gamma-beta pixel stride / 4
    struct Raw { float4 v, g, ga, be; };
    __device__ void consts(int, int c4, float4& m0, float4& m1) const {
        const float4* m = (const float4*)(mr + 8 * c4);
        m0 = m[0]; m1 = m[1];
    }
    __device__ void load(long i4, long pix, int c4, Raw& r) const {
        r.v = x[i4]; r.g = gy[i4]; r.ga = gamma[pix * gbs4 + c4];
        r.be = RELU ? beta[pix * gbs4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __device__ void eval(const Raw& r, long pix, int c4, const float4& m0, const float4& m1, float* a, float* b) const {
        const float xh[4] = {(r.v.x - m0.x) * m0.y, (r.v.y - m0.z) * m0.w, (r.v.z - m1.x) * m1.y, (r.v.w - m1.z) * m1.w};
        const float gg[4] = {r.g.x, r.g.y, r.g.z, r.g.w}, gm[4] = {1.f + r.ga.x, 1.f + r.ga.y, 1.f + r.ga.z, 1.f + r.ga.w};
        const float bb[4] = {r.be.x, r.be.y, r.be.z, r.be.w};
        float dg[4], db[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float q = gg[k];
            if (RELU && !(xh[k] * gm[k] + bb[k] > 0.f)) q = 0.f;
            dg[k] = q * xh[k];
            db[k] = q;
            const float dxh = q * gm[k];
            a[k] = dxh;
            b[k] = dxh * xh[k];
        }
        const long j4 = pix * gbs4 + c4;
        dgamma[j4] = make_float4(dg[0], dg[1], dg[2], dg[3]);
        dbeta[j4] = make_float4(db[0], db[1], db[2], db[3]);
    }
    __device__ void operator()(long i4, long pix, int, int c, float* a, float* b) const {
        const float4* m = (const float4*)(mr + 2 * c);
        const long j4 = pix * gbs4 + (c >> 2);
        float4 m0 = m[0], m1 = m[1], v = x[i4], g = gy[i4], ga = gamma[j4], be = beta[j4];
        float xh[4] = {(v.x - m0.x) * m0.y, (v.y - m0.z) * m0.w, (v.z - m1.x) * m1.y, (v.w - m1.z) * m1.w};
        float gg[4] = {g.x, g.y, g.z, g.w}, gm[4] = {1.f + ga.x, 1.f + ga.y, 1.f + ga.z, 1.f + ga.w};
        float bb[4] = {be.x, be.y, be.z, be.w}, dg[4], db[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float q = gg[k];
            if (RELU && !(xh[k] * gm[k] + bb[k] > 0.f)) q = 0.f;
            dg[k] = q * xh[k];
            db[k] = q;
            float dxh = q * gm[k];
            a[k] = dxh;
            b[k] = dxh * xh[k];
        }
        float4 o1, o2;
        o1.x = dg[0]; o1.y = dg[1]; o1.z = dg[2]; o1.w = dg[3];
        o2.x = db[0]; o2.y = db[1]; o2.z = db[2]; o2.w = db[3];
        dgamma[j4] = o1;
        dbeta[j4] = o2;
    }
};
static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

struct FStats {
    const float* x;
    __device__ void operator()(long i, int, int, float& a, float& b) const { float v = x[i]; a = v; b = v * v; }
};

// One WAVE per (n, c): lane s holds the partial of split s (splits <= 64), summed by the fixed butterfly.  A thread per
// plane walking its splits one dependent load after the other made these launches ~13 us each; they sit on the
// dependency chain between the reduction and the apply pass of every normalisation.
static_assert(PLANE_MAX_SPLITS <= 64, "one lane per split");
__global__ void __launch_bounds__(256) k_inorm_finalize(const double* __restrict__ part, float* __restrict__ mr, int NC, int C,
                                                        int splits, double inv_hw, float eps) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), s = threadIdx.x & 63;
    if (i >= NC) return;              // wave-uniform
    int n = i / C, c = i % C;
    double a = 0.0, b = 0.0;
    if (s < splits) {
        const double* o = part + (((long)n * splits + s) * C + c) * 2;
        a = o[0];
        b = o[1];
    }
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    if (s != 0) return;
    double mean = a * inv_hw;
    double var = b * inv_hw - mean * mean;
    if (var < 0.0) var = 0.0;
    mr[2 * i] = (float)mean;
    mr[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

template <int RELU>
__global__ void k_inorm_apply(const float* __restrict__ x, const float* __restrict__ mr, float* __restrict__ y,
                              long total, int HW, int C, int ycs, int yco) {
    long stride = (long)gridDim.x * blockDim.x;
    long plane = (long)HW * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int c = (int)(i % C);
        int n = (int)(i / plane);
        const float* m = mr + 2 * ((long)n * C + c);
        float v = (x[i] - m[0]) * m[1];
        y[(i / C) * ycs + yco + c] = RELU ? fmaxf(v, 0.f) : v;
    }
}

// float4 variants for C % 4 == 0 and un-sliced tensors (the common case): 16 B per lane streams
template <int RELU>
__global__ void k_inorm_apply4(const float4* __restrict__ x, const float* __restrict__ mr, float4* __restrict__ y, long total4,
                               int HW, int C4, int ycs4, int yco4) {
    long stride = (long)gridDim.x * blockDim.x;
    long plane4 = (long)HW * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
        int c = (int)(i % C4) * 4;
        int n = (int)(i / plane4);
        const float4* m = (const float4*)(mr + 2 * ((long)n * C4 * 4 + c));
        float4 m0 = m[0], m1 = m[1];          // (mean,rstd) pairs of channels c..c+3
        float4 v = x[i], r;
        r.x = (v.x - m0.x) * m0.y; r.y = (v.y - m0.z) * m0.w; r.z = (v.z - m1.x) * m1.y; r.w = (v.w - m1.z) * m1.w;
        if (RELU) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
        y[(i / C4) * ycs4 + yco4 + (i % C4)] = r;
    }
}
template <int RELU>
__global__ void k_inorm_bwd_apply4(const float4* __restrict__ x, const float* __restrict__ mr, const float4* __restrict__ gy,
                                   const float* __restrict__ means, float4* __restrict__ gx, long total4, int HW, int C4,
                                   int gcs4, int gco4) {
    long stride = (long)gridDim.x * blockDim.x;
    long plane4 = (long)HW * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
        int c = (int)(i % C4) * 4;
        int n = (int)(i / plane4);
        long k = 2 * ((long)n * C4 * 4 + c);
        const float4* m = (const float4*)(mr + k);
        const float4* e = (const float4*)(means + k);
        float4 m0 = m[0], m1 = m[1], e0 = e[0], e1 = e[1];
        float4 v = x[i], g = gy[(i / C4) * gcs4 + gco4 + (i % C4)], o;
        float xh, gg;
        xh = (v.x - m0.x) * m0.y; gg = (RELU && !(xh > 0.f)) ? 0.f : g.x; o.x = m0.y * (gg - e0.x - xh * e0.y);
        xh = (v.y - m0.z) * m0.w; gg = (RELU && !(xh > 0.f)) ? 0.f : g.y; o.y = m0.w * (gg - e0.z - xh * e0.w);
        xh = (v.z - m1.x) * m1.y; gg = (RELU && !(xh > 0.f)) ? 0.f : g.z; o.z = m1.y * (gg - e1.x - xh * e1.y);
        xh = (v.w - m1.z) * m1.w; gg = (RELU && !(xh > 0.f)) ? 0.f : g.w; o.w = m1.w * (gg - e1.z - xh * e1.w);
        gx[i] = o;
    }
}

// ---------------------------------------------------------------------------------------------
// Division-free walks (round 4).  The flat-index kernels above spend ~220 vector instructions per float4 on 64-bit integer
// divisions (i / plane4, i % C4, (i / C4) * stride with run-time divisors) and re-load the per-(n, c) constants for every
// element.  When C4 = C / 4 is a power of two <= 256 a 256-thread workgroup covers 256 consecutive float4 = R = 256 / C4 whole
// pixels: thread t keeps its channel quad c4 = t & (C4 - 1) for the whole walk (its constants stay in registers), its pixel is
// p = base + (t >> log2 C4) and advances by R per step; grid.y = image for the per-image constants.  Four independent
// 16-byte loads per tensor in flight per thread.
static inline int ilog2_exact(int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; }
static inline bool walk_ok(int C4) { return C4 >= 1 && C4 <= 256 && ilog2_exact(C4) >= 0; }
// the float4 plane reduction in its pipelined form where C4 allows; returns the number of splits it wrote
template <class F4>
static int launch_plane_reduce4(F4 f, double* part, int N, int HW, int C, hipStream_t st) {
    const int C4 = C / 4;
    static const int pipe_env = []{ const char* e = getenv("VQW_PLANE_PIPE"); return e ? atoi(e) : 3; }();      // bit 0: InstanceNorm functors, bit 1: SPADE
    const int bit = F4::kMinWaves == 2 ? 2 : 1;
    if (walk_ok(C4) && (pipe_env & bit)) {
        const int sp = plane_splits_p(N, HW, C4);
        k_plane_reduce4p<<<dim3(sp, N), 256, 0, st>>>(f, part, HW, C4, ilog2_exact(C4), sp);
        return sp;
    }
    const int sp = plane_splits(N, HW);
    k_plane_reduce4<<<dim3(sp, N), 256, 0, st>>>(f, part, HW, C, sp);
    return sp;
}
static inline int walk_blocks(int N, int HW, int C4) {       // workgroups per image: ~2048 in all, each >= one 4 R-pixel step
    const int R = 256 / C4;
    int per = ceil_div(2048, N);
    int cap = ceil_div(HW, 4 * R);
    return imax(1, imin(per, cap));
}
#define WALK4_SETUP                                                        \
    const int t = threadIdx.x, c4 = t & (C4 - 1), r = t >> lgC4, R = 256 >> lgC4

template <int RELU>
__global__ void __launch_bounds__(256) k_inorm_apply4w(const float4* __restrict__ x, const float* __restrict__ mr, float4* __restrict__ y,
                                                       int HW, int C4, int lgC4, int ycs4, int yco4) {
    WALK4_SETUP;
    const int n = blockIdx.y;
    const float4* m = (const float4*)(mr + 8 * ((long)n * C4 + c4));
    const float4 m0 = m[0], m1 = m[1];
    const float4* xn = x + (long)n * HW * C4 + c4;
    float4* yn = y + (long)n * HW * ycs4 + yco4 + c4;
    for (int p0 = blockIdx.x * 4 * R + r; p0 < HW; p0 += gridDim.x * 4 * R) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (p0 + u * R < HW) v[u] = xn[(long)(p0 + u * R) * C4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (p0 + u * R >= HW) break;
            float4 o;
            o.x = (v[u].x - m0.x) * m0.y; o.y = (v[u].y - m0.z) * m0.w; o.z = (v[u].z - m1.x) * m1.y; o.w = (v[u].w - m1.z) * m1.w;
            if (RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            yn[(long)(p0 + u * R) * ycs4] = o;
        }
    }
}

// out = a + InstanceNorm(+ReLU)(x): the residual add behind a block whose last operator is that norm (the decoder's tail,
// unet_decoder.py:169-171 `x + conv_last(x)`): the normalised tensor is never written
template <int RELU>
__global__ void __launch_bounds__(256) k_inorm_add4w(const float4* __restrict__ x, const float* __restrict__ mr, const float4* __restrict__ a,
                                                     float4* __restrict__ y, int HW, int C4, int lgC4) {
    WALK4_SETUP;
    const int n = blockIdx.y;
    const float4* m = (const float4*)(mr + 8 * ((long)n * C4 + c4));
    const float4 m0 = m[0], m1 = m[1];
    const long base = (long)n * HW * C4 + c4;
    for (int p0 = blockIdx.x * 4 * R + r; p0 < HW; p0 += gridDim.x * 4 * R) {
        float4 v[4], w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (p0 + u * R < HW) { v[u] = x[base + (long)(p0 + u * R) * C4]; w[u] = a[base + (long)(p0 + u * R) * C4]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (p0 + u * R >= HW) break;
            float4 o;
            o.x = (v[u].x - m0.x) * m0.y; o.y = (v[u].y - m0.z) * m0.w; o.z = (v[u].z - m1.x) * m1.y; o.w = (v[u].w - m1.z) * m1.w;
            if (RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            o.x = w[u].x + o.x; o.y = w[u].y + o.y; o.z = w[u].z + o.z; o.w = w[u].w + o.w;
            y[base + (long)(p0 + u * R) * C4] = o;
        }
    }
}
extern "C" int vqw_inorm_add_supported(int C) { return (C % 4 == 0 && walk_ok(C / 4)) ? 1 : 0; }
extern "C" int vqw_inorm_add_fwd(const float* x, const float* mean_rstd, const float* a, float* y, int N, int HW, int C, int relu,
                                 void* stream) {
    VQW_PROF_HBM(stream, 3, (double)N * HW * C);
    VQW_CHECK(x && mean_rstd && a && y && N > 0 && HW > 0 && C > 0, "vqw_inorm_add_fwd: bad arguments");
    VQW_CHECK(vqw_inorm_add_supported(C) && al16(x) && al16(a) && al16(y) && al16(mean_rstd),
              "vqw_inorm_add_fwd: shape not served (query vqw_inorm_add_supported) or unaligned tensors");
    const int C4 = C / 4;
    const dim3 g(walk_blocks(N, HW, C4), N);
    hipStream_t st = (hipStream_t)stream;
    if (relu) k_inorm_add4w<1><<<g, 256, 0, st>>>((const float4*)x, mean_rstd, (const float4*)a, (float4*)y, HW, C4, ilog2_exact(C4));
    else k_inorm_add4w<0><<<g, 256, 0, st>>>((const float4*)x, mean_rstd, (const float4*)a, (float4*)y, HW, C4, ilog2_exact(C4));
    VQW_LAUNCH_CHECK("vqw_inorm_add_fwd");
    return VQW_OK;
}

__device__ __forceinline__ float4 inorm_bwd_elem(const float4 v, const float4 g, const float4 m0, const float4 m1, const float4 e0,
                                                 const float4 e1, const bool relu) {
    float4 o;
    float xh, gg;
    xh = (v.x - m0.x) * m0.y; gg = (relu && !(xh > 0.f)) ? 0.f : g.x; o.x = m0.y * (gg - e0.x - xh * e0.y);
    xh = (v.y - m0.z) * m0.w; gg = (relu && !(xh > 0.f)) ? 0.f : g.y; o.y = m0.w * (gg - e0.z - xh * e0.w);
    xh = (v.z - m1.x) * m1.y; gg = (relu && !(xh > 0.f)) ? 0.f : g.z; o.z = m1.y * (gg - e1.x - xh * e1.y);
    xh = (v.w - m1.z) * m1.w; gg = (relu && !(xh > 0.f)) ? 0.f : g.w; o.w = m1.w * (gg - e1.z - xh * e1.w);
    return o;
}

template <int RELU>
__global__ void __launch_bounds__(256) k_inorm_bwd_apply4w(const float4* __restrict__ x, const float* __restrict__ mr,
                                                           const float4* __restrict__ gy, const float* __restrict__ means,
                                                           float4* __restrict__ gx, int HW, int C4, int lgC4, int gcs4, int gco4) {
    WALK4_SETUP;
    const int n = blockIdx.y;
    const long k = 8 * ((long)n * C4 + c4);
    const float4 m0 = ((const float4*)(mr + k))[0], m1 = ((const float4*)(mr + k))[1];
    const float4 e0 = ((const float4*)(means + k))[0], e1 = ((const float4*)(means + k))[1];
    const float4* xn = x + (long)n * HW * C4 + c4;
    const float4* gn = gy + (long)n * HW * gcs4 + gco4 + c4;
    float4* on = gx + (long)n * HW * C4 + c4;
    for (int p0 = blockIdx.x * 4 * R + r; p0 < HW; p0 += gridDim.x * 4 * R) {
        float4 v[4], g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (p0 + u * R < HW) { v[u] = xn[(long)(p0 + u * R) * C4]; g[u] = gn[(long)(p0 + u * R) * gcs4]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (p0 + u * R >= HW) break;
            on[(long)(p0 + u * R) * C4] = inorm_bwd_elem(v[u], g[u], m0, m1, e0, e1, RELU);
        }
    }
}

__global__ void __launch_bounds__(256) k_inorm_bwd_pair_apply4w(const float4* __restrict__ xa, const float* __restrict__ mra,
                                                                const float* __restrict__ ea, const float4* __restrict__ xb,
                                                                const float* __restrict__ mrb, const float* __restrict__ eb,
                                                                const float4* __restrict__ gy, float4* __restrict__ gxa,
                                                                float4* __restrict__ gxb, int HW, int C4, int lgC4) {
    WALK4_SETUP;
    const int n = blockIdx.y;
    const long k = 8 * ((long)n * C4 + c4);
    const float4 a0 = ((const float4*)(mra + k))[0], a1 = ((const float4*)(mra + k))[1];
    const float4 f0 = ((const float4*)(ea + k))[0], f1 = ((const float4*)(ea + k))[1];
    const float4 b0 = ((const float4*)(mrb + k))[0], b1 = ((const float4*)(mrb + k))[1];
    const float4 h0 = ((const float4*)(eb + k))[0], h1 = ((const float4*)(eb + k))[1];
    const long base = (long)n * HW * C4 + c4;
    for (int p0 = blockIdx.x * 2 * R + r; p0 < HW; p0 += gridDim.x * 2 * R) {
        float4 va[2], vb[2], g[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (p0 + u * R < HW) {
                const long i = base + (long)(p0 + u * R) * C4;
                va[u] = xa[i]; vb[u] = xb[i]; g[u] = gy[i];
            }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (p0 + u * R >= HW) break;
            const long i = base + (long)(p0 + u * R) * C4;
            gxa[i] = inorm_bwd_elem(va[u], g[u], a0, a1, f0, f1, true);
            gxb[i] = inorm_bwd_elem(vb[u], g[u], b0, b1, h0, h1, false);
        }
    }
}

// SPADE: constants per channel only; the walk runs over the P = N * HW pixels of the batch (grid.x only)
// RES 2: the residual is InstanceNorm(+ReLU) of `res`, applied here from its per-(image, channel) statistics rmr [N][C][2]
// (StyledResUpBlock's shortcut branch: its normalised tensor is never written; HW = 2^lgHW pixels per image)
template <int RELU, int RES>
__global__ void __launch_bounds__(256) k_spade_fwd4w(const float4* __restrict__ x, const float* __restrict__ mr,
                                                     const float4* __restrict__ gamma, const float4* __restrict__ beta,
                                                     float4* __restrict__ y, long P, int C4, int lgC4, int gbs4,
                                                     const float4* __restrict__ res, const float* __restrict__ rmr = nullptr,
                                                     int lgHW = 0, int rrelu = 0) {
    WALK4_SETUP;
    const float4 m0 = ((const float4*)(mr + 8 * c4))[0], m1 = ((const float4*)(mr + 8 * c4))[1];
    const float rlo = rrelu ? 0.f : -__builtin_inff();
    for (long p0 = (long)blockIdx.x * 4 * R + r; p0 < P; p0 += (long)gridDim.x * 4 * R) {
        float4 v[4], ga[4], be[4], rr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (p0 + u * R < P) {
                const long p = p0 + u * R;
                v[u] = x[p * C4 + c4]; ga[u] = gamma[p * gbs4 + c4]; be[u] = beta[p * gbs4 + c4];
                if (RES) rr[u] = res[p * C4 + c4];
                if (RES == 2) {
                    const float4* q = (const float4*)(rmr + (((p >> lgHW) << lgC4) + c4) * 8);      // (mean, rstd) of 4 channels
                    const float4 q0 = q[0], q1 = q[1];
                    rr[u].x = fmaxf((rr[u].x - q0.x) * q0.y, rlo); rr[u].y = fmaxf((rr[u].y - q0.z) * q0.w, rlo);
                    rr[u].z = fmaxf((rr[u].z - q1.x) * q1.y, rlo); rr[u].w = fmaxf((rr[u].w - q1.z) * q1.w, rlo);
                }
            }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (p0 + u * R >= P) break;
            float4 o;
            o.x = (v[u].x - m0.x) * m0.y * (1.f + ga[u].x) + be[u].x;
            o.y = (v[u].y - m0.z) * m0.w * (1.f + ga[u].y) + be[u].y;
            o.z = (v[u].z - m1.x) * m1.y * (1.f + ga[u].z) + be[u].z;
            o.w = (v[u].w - m1.z) * m1.w * (1.f + ga[u].w) + be[u].w;
            if (RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            if (RES) { o.x += rr[u].x; o.y += rr[u].y; o.z += rr[u].z; o.w += rr[u].w; }
            y[(p0 + u * R) * C4 + c4] = o;
        }
    }
}

template <int RELU, int TRAIN>
__global__ void __launch_bounds__(256) k_spade_bwd_apply4w(const float4* __restrict__ x, const float* __restrict__ mr,
                                                           const float4* __restrict__ gamma, const float4* __restrict__ beta,
                                                           const float4* __restrict__ gy, const double* __restrict__ sums,
                                                           double inv_count, float4* __restrict__ gx, long P, int C4, int lgC4, int gbs4) {
    WALK4_SETUP;
    const float4 m0 = ((const float4*)(mr + 8 * c4))[0], m1 = ((const float4*)(mr + 8 * c4))[1];
    const float mean[4] = {m0.x, m0.z, m1.x, m1.z}, rs[4] = {m0.y, m0.w, m1.y, m1.w};
    float s1[4], s2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        s1[k] = TRAIN ? (float)(sums[2 * (4 * c4 + k)] * inv_count) : 0.f;
        s2[k] = TRAIN ? (float)(sums[2 * (4 * c4 + k) + 1] * inv_count) : 0.f;
    }
    for (long p0 = (long)blockIdx.x * 2 * R + r; p0 < P; p0 += (long)gridDim.x * 2 * R) {
        float4 v[2], ga4[2], g4[2], be4[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (p0 + u * R < P) {
                const long p = p0 + u * R;
                v[u] = x[p * C4 + c4]; ga4[u] = gamma[p * gbs4 + c4]; g4[u] = gy[p * C4 + c4];
                be4[u] = RELU ? beta[p * gbs4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (p0 + u * R >= P) break;
            const float xv[4] = {v[u].x, v[u].y, v[u].z, v[u].w}, gav[4] = {ga4[u].x, ga4[u].y, ga4[u].z, ga4[u].w};
            const float gv[4] = {g4[u].x, g4[u].y, g4[u].z, g4[u].w}, bev[4] = {be4[u].x, be4[u].y, be4[u].z, be4[u].w};
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float rk = rs[k];
                const float xh = (xv[k] - mean[k]) * rk;
                const float ga = 1.f + gav[k];
                float g = gv[k];
                if (RELU) {
                    const float out = xh * ga + bev[k];
                    if (!(out > 0.f)) g = 0.f;
                }
                const float dxh = g * ga;
                o[k] = TRAIN ? rk * (dxh - s1[k] - xh * s2[k]) : rk * dxh;
            }
            gx[(p0 + u * R) * C4 + c4] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}
static inline int walk_blocks_flat(long P, int C4, int per_step) {
    const int R = 256 / C4;
    long g = (P + (long)per_step * R - 1) / ((long)per_step * R);
    return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

// launchers: the walk form where C4 allows, else the flat-index form
static void launch_inorm_apply4(const float* x, const float* mr, float* y, int N, int HW, int C, int ycs, int yco, int relu, hipStream_t st) {
    const int C4 = C / 4;
    if (walk_ok(C4)) {
        const dim3 g(walk_blocks(N, HW, C4), N);
        if (relu) k_inorm_apply4w<1><<<g, 256, 0, st>>>((const float4*)x, mr, (float4*)y, HW, C4, ilog2_exact(C4), ycs / 4, yco / 4);
        else k_inorm_apply4w<0><<<g, 256, 0, st>>>((const float4*)x, mr, (float4*)y, HW, C4, ilog2_exact(C4), ycs / 4, yco / 4);
        return;
    }
    const long t4 = (long)N * HW * C4;
    if (relu) k_inorm_apply4<1><<<stream_grid(t4, 256), 256, 0, st>>>((const float4*)x, mr, (float4*)y, t4, HW, C4, ycs / 4, yco / 4);
    else k_inorm_apply4<0><<<stream_grid(t4, 256), 256, 0, st>>>((const float4*)x, mr, (float4*)y, t4, HW, C4, ycs / 4, yco / 4);
}
static void launch_inorm_bwd_apply4(const float* x, const float* mr, const float* gy, const float* means, float* gx, int N, int HW, int C,
                                    int gcs, int gco, int relu, hipStream_t st) {
    const int C4 = C / 4;
    if (walk_ok(C4)) {
        const dim3 g(walk_blocks(N, HW, C4), N);
        if (relu) k_inorm_bwd_apply4w<1><<<g, 256, 0, st>>>((const float4*)x, mr, (const float4*)gy, means, (float4*)gx, HW, C4, ilog2_exact(C4), gcs / 4, gco / 4);
        else k_inorm_bwd_apply4w<0><<<g, 256, 0, st>>>((const float4*)x, mr, (const float4*)gy, means, (float4*)gx, HW, C4, ilog2_exact(C4), gcs / 4, gco / 4);
        return;
    }
    const long t4 = (long)N * HW * C4;
    if (relu) k_inorm_bwd_apply4<1><<<stream_grid(t4, 256), 256, 0, st>>>((const float4*)x, mr, (const float4*)gy, means, (float4*)gx, t4, HW, C4, gcs / 4, gco / 4);
    else k_inorm_bwd_apply4<0><<<stream_grid(t4, 256), 256, 0, st>>>((const float4*)x, mr, (const float4*)gy, means, (float4*)gx, t4, HW, C4, gcs / 4, gco / 4);
}

// statistics from per-tile float partials part[n][nparts][C][2] (written by the conv that produced x): one wave per
// (n, c), lanes take the tiles round-robin and sum in double, fixed butterfly
__global__ void __launch_bounds__(256) k_inorm_finalize_parts(const float* __restrict__ part, float* __restrict__ mr, int NC, int C,
                                                              int nparts, double inv_hw, float eps, double inv_tile) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), s = threadIdx.x & 63;
    if (i >= NC) return;              // wave-uniform
    const int n = i / C, c = i % C;
    double a = 0.0, b = 0.0;
    for (int t = s; t < nparts; t += 64) {
        const float* o = part + (((long)n * nparts + t) * C + c) * 2;     // (sum, M2 about the tile mean) of one tile
        const double st = (double)o[0];
        a += st;
        b += (double)o[1] + st * st * inv_tile;                           // -> sum of squares, in double
    }
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    if (s != 0) return;
    double mean = a * inv_hw;
    double var = b * inv_hw - mean * mean;
    if (var < 0.0) var = 0.0;
    mr[2 * i] = (float)mean;
    mr[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}
// two norms of one shape (a ResBlock's tail: main branch and 1x1 branch) in ONE launch: blockIdx.y selects the job
__global__ void __launch_bounds__(256) k_inorm_finalize_parts2(const float* __restrict__ part_a, float* __restrict__ mr_a, int nparts_a,
                                                               const float* __restrict__ part_b, float* __restrict__ mr_b, int nparts_b,
                                                               int NC, int C, double inv_hw, float eps) {
    const float* part = blockIdx.y ? part_b : part_a;
    float* mr = blockIdx.y ? mr_b : mr_a;
    const int nparts = blockIdx.y ? nparts_b : nparts_a;
    const double inv_tile = (double)nparts * inv_hw;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), s = threadIdx.x & 63;
    if (i >= NC) return;              // wave-uniform
    const int n = i / C, c = i % C;
    double a = 0.0, b = 0.0;
    for (int t = s; t < nparts; t += 64) {
        const float* o = part + (((long)n * nparts + t) * C + c) * 2;
        const double st = (double)o[0];
        a += st;
        b += (double)o[1] + st * st * inv_tile;
    }
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    if (s != 0) return;
    double mean = a * inv_hw;
    double var = b * inv_hw - mean * mean;
    if (var < 0.0) var = 0.0;
    mr[2 * i] = (float)mean;
    mr[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}
extern "C" int vqw_inorm_stats_parts2(const float* part_a, int nparts_a, float* mean_rstd_a, const float* part_b, int nparts_b,
                                      float* mean_rstd_b, int N, int HW, int C, float eps, void* stream) {
    VQW_CHECK(part_a && part_b && mean_rstd_a && mean_rstd_b && nparts_a > 0 && nparts_b > 0 && N > 0 && HW > 0 && C > 0,
              "vqw_inorm_stats_parts2: bad arguments");
    k_inorm_finalize_parts2<<<dim3(ceil_div((long)N * C, 4), 2), 256, 0, (hipStream_t)stream>>>(part_a, mean_rstd_a, nparts_a, part_b, mean_rstd_b,
                                                                                                  nparts_b, N * C, C, 1.0 / (double)HW, eps);
    VQW_LAUNCH_CHECK("vqw_inorm_stats_parts2");
    return VQW_OK;
}

extern "C" int vqw_inorm_fwd_parts(const float* x, float* y, int y_cstride, int y_coff, float* mean_rstd, const float* part,
                                   int nparts, int N, int HW, int C, float eps, int relu, void* stream) {
    VQW_PROF_HBM(stream, 2, (double)N * HW * C);
    VQW_CHECK(x && y && mean_rstd && part && nparts > 0 && N > 0 && HW > 0 && C > 0, "vqw_inorm_fwd_parts: bad arguments");
    VQW_CHECK(y_coff >= 0 && y_coff + C <= y_cstride, "vqw_inorm_fwd_parts: output channel slice [%d,%d) outside stride %d", y_coff, y_coff + C, y_cstride);
    hipStream_t st = (hipStream_t)stream;
    k_inorm_finalize_parts<<<ceil_div((long)N * C, 4), 256, 0, st>>>(part, mean_rstd, N * C, C, nparts, 1.0 / (double)HW, eps, (double)nparts / (double)HW);
    long total = (long)N * HW * C;
    if ((C & 3) == 0 && (y_cstride & 3) == 0 && (y_coff & 3) == 0 && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)mean_rstd) & 15) == 0)) {
        launch_inorm_apply4(x, mean_rstd, y, N, HW, C, y_cstride, y_coff, relu, st);
    } else if (relu) k_inorm_apply<1><<<stream_grid(total, 256), 256, 0, st>>>(x, mean_rstd, y, total, HW, C, y_cstride, y_coff);
    else k_inorm_apply<0><<<stream_grid(total, 256), 256, 0, st>>>(x, mean_rstd, y, total, HW, C, y_cstride, y_coff);
    VQW_LAUNCH_CHECK("vqw_inorm_fwd_parts");
    return VQW_OK;
}

// statistics only (mean, rstd per (n, c)): for consumers that normalise while they read (vqw_res_tail_norm_fwd)
extern "C" int vqw_inorm_stats(const float* x, float* mean_rstd, void* ws, size_t ws_bytes, int N, int HW, int C, float eps,
                               void* stream) {
    VQW_CHECK(x && mean_rstd && ws && N > 0 && HW > 0 && C > 0, "vqw_inorm_stats: bad arguments");
    VQW_CHECK(ws_bytes >= vqw_plane_ws_bytes(N, C, HW), "vqw_inorm_stats: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int splits = plane_splits(N, HW);
    if ((C & 3) == 0 && al16(x)) {
        FStats4 f{(const float4*)x};
        splits = launch_plane_reduce4(f, (double*)ws, N, HW, C, st);
    } else {
        FStats f{x};
        k_plane_reduce<<<dim3(splits, N), 256, 0, st>>>(f, (double*)ws, HW, C, splits);
    }
    k_inorm_finalize<<<ceil_div((long)N * C, 4), 256, 0, st>>>((const double*)ws, mean_rstd, N * C, C, splits, 1.0 / (double)HW, eps);
    VQW_LAUNCH_CHECK("vqw_inorm_stats");
    return VQW_OK;
}
extern "C" int vqw_inorm_stats_parts(const float* part, int nparts, float* mean_rstd, int N, int HW, int C, float eps, void* stream) {
    VQW_CHECK(part && mean_rstd && nparts > 0 && N > 0 && HW > 0 && C > 0, "vqw_inorm_stats_parts: bad arguments");
    k_inorm_finalize_parts<<<ceil_div((long)N * C, 4), 256, 0, (hipStream_t)stream>>>(part, mean_rstd, N * C, C, nparts, 1.0 / (double)HW, eps, (double)nparts / (double)HW);
    VQW_LAUNCH_CHECK("vqw_inorm_stats_parts");
    return VQW_OK;
}

extern "C" int vqw_inorm_fwd(const float* x, float* y, int y_cstride, int y_coff, float* mean_rstd, void* ws,
                             size_t ws_bytes, int N, int HW, int C, float eps, int relu, void* stream) {
    VQW_PROF_HBM(stream, 3, (double)N * HW * C);
    VQW_CHECK(x && y && mean_rstd && ws && N > 0 && HW > 0 && C > 0, "vqw_inorm_fwd: bad arguments");
    VQW_CHECK(y_coff >= 0 && y_coff + C <= y_cstride, "vqw_inorm_fwd: output channel slice [%d,%d) outside stride %d", y_coff, y_coff + C, y_cstride);
    VQW_CHECK(ws_bytes >= vqw_plane_ws_bytes(N, C, HW), "vqw_inorm_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int splits = plane_splits(N, HW);
    if ((C & 3) == 0 && al16(x)) {
        FStats4 f{(const float4*)x};
        splits = launch_plane_reduce4(f, (double*)ws, N, HW, C, st);
    } else {
        FStats f{x};
        k_plane_reduce<<<dim3(splits, N), 256, 0, st>>>(f, (double*)ws, HW, C, splits);
    }
    k_inorm_finalize<<<ceil_div((long)N * C, 4), 256, 0, st>>>((const double*)ws, mean_rstd, N * C, C, splits,
                                                                 1.0 / (double)HW, eps);
    long total = (long)N * HW * C;
    if ((C & 3) == 0 && (y_cstride & 3) == 0 && (y_coff & 3) == 0 && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)mean_rstd) & 15) == 0)) {
        launch_inorm_apply4(x, mean_rstd, y, N, HW, C, y_cstride, y_coff, relu, st);
    } else if (relu) k_inorm_apply<1><<<stream_grid(total, 256), 256, 0, st>>>(x, mean_rstd, y, total, HW, C, y_cstride, y_coff);
    else k_inorm_apply<0><<<stream_grid(total, 256), 256, 0, st>>>(x, mean_rstd, y, total, HW, C, y_cstride, y_coff);
    VQW_LAUNCH_CHECK("vqw_inorm_fwd");
    return VQW_OK;
}

// backward: ghat = gy * [xhat > 0] (relu) ; dx = rstd * (ghat - mean(ghat) - xhat * mean(ghat*xhat))
template <int RELU>
struct FInBwd {
    const float* x;
    const float* mr;
    const float* gy;
    int C, gcs, gco;
    __device__ void operator()(long i, int n, int c, float& a, float& b) const {
        const float* m = mr + 2 * ((long)n * C + c);
        float xh = (x[i] - m[0]) * m[1];
        float g = gy[(i / C) * gcs + gco + c];
        if (RELU && !(xh > 0.f)) g = 0.f;
        a = g;
        b = g * xh;
    }
};

__global__ void __launch_bounds__(256) k_plane_sum_finalize(const double* __restrict__ part, float* __restrict__ out, int NC, int C,
                                                            int splits, double scale) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), s = threadIdx.x & 63;       // one wave per (n, c), see k_inorm_finalize
    if (i >= NC) return;
    int n = i / C, c = i % C;
    double a = 0.0, b = 0.0;
    if (s < splits) {
        const double* o = part + (((long)n * splits + s) * C + c) * 2;
        a = o[0];
        b = o[1];
    }
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    if (s != 0) return;
    out[2 * i] = (float)(a * scale);
    out[2 * i + 1] = (float)(b * scale);
}

// two jobs of one shape in one launch (the paired backward of a ResBlock's tail norms): blockIdx.y selects the job
__global__ void __launch_bounds__(256) k_plane_sum_finalize2(const double* __restrict__ part_a, float* __restrict__ out_a,
                                                             const double* __restrict__ part_b, float* __restrict__ out_b, int NC, int C,
                                                             int splits, double scale) {
    const double* part = blockIdx.y ? part_b : part_a;
    float* out = blockIdx.y ? out_b : out_a;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), s = threadIdx.x & 63;
    if (i >= NC) return;
    int n = i / C, c = i % C;
    double a = 0.0, b = 0.0;
    if (s < splits) {
        const double* o = part + (((long)n * splits + s) * C + c) * 2;
        a = o[0];
        b = o[1];
    }
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    if (s != 0) return;
    out[2 * i] = (float)(a * scale);
    out[2 * i + 1] = (float)(b * scale);
}

template <int RELU>
__global__ void k_inorm_bwd_apply(const float* __restrict__ x, const float* __restrict__ mr, const float* __restrict__ gy,
                                  const float* __restrict__ means, float* __restrict__ gx, long total, int HW, int C,
                                  int gcs, int gco) {
    long stride = (long)gridDim.x * blockDim.x;
    long plane = (long)HW * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int c = (int)(i % C);
        int n = (int)(i / plane);
        long k = 2 * ((long)n * C + c);
        float r = mr[k + 1];
        float xh = (x[i] - mr[k]) * r;
        float g = gy[(i / C) * gcs + gco + c];
        if (RELU && !(xh > 0.f)) g = 0.f;
        gx[i] = r * (g - means[k] - xh * means[k + 1]);
    }
}

extern "C" int vqw_inorm_bwd(const float* x, const float* mean_rstd, const float* gy, int gy_cstride, int gy_coff,
                             float* gx, void* ws, size_t ws_bytes, int N, int HW, int C, int relu, void* stream) {
    VQW_PROF_HBM(stream, 5, (double)N * HW * C);
    VQW_CHECK(x && mean_rstd && gy && gx && ws && N > 0 && HW > 0 && C > 0, "vqw_inorm_bwd: bad arguments");
    VQW_CHECK(gy_coff >= 0 && gy_coff + C <= gy_cstride, "vqw_inorm_bwd: gradient channel slice outside stride");
    size_t need = vqw_plane_ws_bytes(N, C, HW);
    VQW_CHECK(ws_bytes >= need, "vqw_inorm_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int splits = plane_splits(N, HW);
    double* part = (double*)ws;
    float* means = (float*)((char*)ws + plane_part_bytes(N, C));
    long total = (long)N * HW * C;
    const bool vec = (C & 3) == 0 && (gy_cstride & 3) == 0 && (gy_coff & 3) == 0 && al16(x) && al16(gy) && al16(mean_rstd);
    if (vec && relu) {
        FInBwd4<1> f{(const float4*)x, mean_rstd, (const float4*)gy, C, gy_cstride / 4, gy_coff / 4};
        splits = launch_plane_reduce4(f, part, N, HW, C, st);
    } else if (vec) {
        FInBwd4<0> f{(const float4*)x, mean_rstd, (const float4*)gy, C, gy_cstride / 4, gy_coff / 4};
        splits = launch_plane_reduce4(f, part, N, HW, C, st);
    } else if (relu) {
        FInBwd<1> f{x, mean_rstd, gy, C, gy_cstride, gy_coff};
        k_plane_reduce<<<dim3(splits, N), 256, 0, st>>>(f, part, HW, C, splits);
    } else {
        FInBwd<0> f{x, mean_rstd, gy, C, gy_cstride, gy_coff};
        k_plane_reduce<<<dim3(splits, N), 256, 0, st>>>(f, part, HW, C, splits);
    }
    k_plane_sum_finalize<<<ceil_div((long)N * C, 4), 256, 0, st>>>(part, means, N * C, C, splits, 1.0 / (double)HW);
    if ((C & 3) == 0 && (gy_cstride & 3) == 0 && (gy_coff & 3) == 0 &&
        ((((uintptr_t)x | (uintptr_t)gy | (uintptr_t)gx | (uintptr_t)mean_rstd | (uintptr_t)means) & 15) == 0)) {
        launch_inorm_bwd_apply4(x, mean_rstd, gy, means, gx, N, HW, C, gy_cstride, gy_coff, relu, st);
    } else if (relu) k_inorm_bwd_apply<1><<<stream_grid(total, 256), 256, 0, st>>>(x, mean_rstd, gy, means, gx, total, HW, C, gy_cstride, gy_coff);
    else k_inorm_bwd_apply<0><<<stream_grid(total, 256), 256, 0, st>>>(x, mean_rstd, gy, means, gx, total, HW, C, gy_cstride, gy_coff);
    VQW_LAUNCH_CHECK("vqw_inorm_bwd");
    return VQW_OK;
}

// The same backward with the two sums taken from per-region partials part[N][nparts][C][2] = (sum gm, sum gm * xhat) that the
// consumer convolution's input-gradient launch left in its epilogue (vqw_conv3x3_wino_fwd_inbwd): no reduction pass over x and
// gy.  One wave per (n, c) adds the regions in double, then the apply kernel of vqw_inorm_bwd.
__global__ void __launch_bounds__(256) k_plane_sum_finalize_f(const float* __restrict__ part, float* __restrict__ out, int NC, int C,
                                                              int nparts, double scale) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), s = threadIdx.x & 63;
    if (i >= NC) return;              // wave-uniform
    const int n = i / C, c = i % C;
    double a = 0.0, b = 0.0;
    for (int t = s; t < nparts; t += 64) {
        const float* o = part + (((long)n * nparts + t) * C + c) * 2;
        a += (double)o[0];
        b += (double)o[1];
    }
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    if (s != 0) return;
    out[2 * i] = (float)(a * scale);
    out[2 * i + 1] = (float)(b * scale);
}

extern "C" int vqw_inorm_bwd_parts(const float* x, const float* mean_rstd, const float* gy, const float* part, int nparts, float* means_ws,
                                   float* gx, int N, int HW, int C, int relu, void* stream) {
    VQW_PROF_HBM(stream, 3, (double)N * HW * C);
    VQW_CHECK(x && mean_rstd && gy && part && means_ws && gx && nparts > 0 && N > 0 && HW > 0 && C > 0, "vqw_inorm_bwd_parts: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    k_plane_sum_finalize_f<<<ceil_div((long)N * C, 4), 256, 0, st>>>(part, means_ws, N * C, C, nparts, 1.0 / (double)HW);
    const long total = (long)N * HW * C;
    if ((C & 3) == 0 && ((((uintptr_t)x | (uintptr_t)gy | (uintptr_t)gx | (uintptr_t)mean_rstd | (uintptr_t)means_ws) & 15) == 0)) {
        launch_inorm_bwd_apply4(x, mean_rstd, gy, means_ws, gx, N, HW, C, C, 0, relu, st);
    } else if (relu) k_inorm_bwd_apply<1><<<stream_grid(total, 256), 256, 0, st>>>(x, mean_rstd, gy, means_ws, gx, total, HW, C, C, 0);
    else k_inorm_bwd_apply<0><<<stream_grid(total, 256), 256, 0, st>>>(x, mean_rstd, gy, means_ws, gx, total, HW, C, C, 0);
    VQW_LAUNCH_CHECK("vqw_inorm_bwd_parts");
    return VQW_OK;
}

// ---------------------------------------------------------------------------------------------
// Backward of TWO InstanceNorms that receive the SAME gradient (the two branches in front of a ResBlock tail: a with
// its ReLU, b without): one reduction and one apply kernel read the common gradient once instead of twice each.
// Arithmetic per element is that of vqw_inorm_bwd.
__global__ void __launch_bounds__(256, 4) k_inorm_bwd_pair_reduce4(const float4* __restrict__ xa, const float* __restrict__ mra,
                                                                const float4* __restrict__ xb, const float* __restrict__ mrb,
                                                                const float4* __restrict__ gy, double* __restrict__ parta,
                                                                double* __restrict__ partb, int HW, int C, int splits) {
    __shared__ double sq[4][4][256];        // [quantity: a.sum, a.dot, b.sum, b.dot][channel of the quad][thread]
    const int n = blockIdx.y, s = blockIdx.x;
    const int C4 = C >> 2;
    const int tcn = C4 < 256 ? C4 : 256;
    const int rows = 256 / tcn;
    const int t = threadIdx.x;
    const int tc = t % tcn, tr = t / tcn;
    const int per = (HW + splits - 1) / splits;
    const int p0 = s * per;
    const int p1 = (p0 + per < HW) ? p0 + per : HW;
    const bool active = tr < rows;
    for (int cb = 0; cb < C4; cb += tcn) {
        const int c4 = cb + tc;
        double acc[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[q][k] = 0.0;
        if (active && c4 < C4) {
            const float4* ma = (const float4*)(mra + 2 * ((long)n * C + c4 * 4));
            const float4* mb = (const float4*)(mrb + 2 * ((long)n * C + c4 * 4));
            const float4 a0 = ma[0], a1 = ma[1], b0 = mb[0], b1 = mb[1];
            for (int p = p0 + tr; p < p1; p += rows) {
                const long i4 = ((long)n * HW + p) * C4 + c4;
                const float4 va = xa[i4], vb = xb[i4], g = gy[i4];
                const float xha[4] = {(va.x - a0.x) * a0.y, (va.y - a0.z) * a0.w, (va.z - a1.x) * a1.y, (va.w - a1.z) * a1.w};
                const float xhb[4] = {(vb.x - b0.x) * b0.y, (vb.y - b0.z) * b0.w, (vb.z - b1.x) * b1.y, (vb.w - b1.z) * b1.w};
                const float gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float qa = !(xha[k] > 0.f) ? 0.f : gg[k];        // branch a: ReLU after the norm
                    acc[0][k] += (double)qa;
                    acc[1][k] += (double)(qa * xha[k]);
                    acc[2][k] += (double)gg[k];
                    acc[3][k] += (double)(gg[k] * xhb[k]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) sq[q][k][t] = acc[q][k];
        __syncthreads();
        if (tr == 0 && c4 < C4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double tq[4] = {acc[0][k], acc[1][k], acc[2][k], acc[3][k]};
                for (int r = 1; r < rows; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q) tq[q] += sq[q][k][r * tcn + tc];
                double* oa = parta + (((long)n * splits + s) * C + c4 * 4 + k) * 2;
                double* ob = partb + (((long)n * splits + s) * C + c4 * 4 + k) * 2;
                oa[0] = tq[0]; oa[1] = tq[1];
                ob[0] = tq[2]; ob[1] = tq[3];
            }
        }
        __syncthreads();
    }
}
// k_res_tail_bwd4 (elementwise.hip) and k_inorm_bwd_pair_reduce4 in ONE pass: a thread takes a 2 x 2 pooling window of its channel
// quad, forms the gradient g in front of the tail's ReLU (g_out + the pooled gradient routed to the window's first maximum, masked
// by out > 0), stores it for the apply pass and adds its four pixels to the two norms' backward sums - g is not read back for
// the reduction.  Same partial layout and LDS fold as k_inorm_bwd_pair_reduce4; the splits cut the image's windows.
__global__ void __launch_bounds__(256, 2) k_res_tail_bwd_pair_reduce4(const float4* __restrict__ out, const float4* __restrict__ gp,
                                                                      const float4* __restrict__ go, const float4* __restrict__ xa,
                                                                      const float* __restrict__ mra, const float4* __restrict__ xb,
                                                                      const float* __restrict__ mrb, float4* __restrict__ gw,
                                                                      double* __restrict__ parta, double* __restrict__ partb, int H,
                                                                      int W, int C, int splits) {
    __shared__ double sq[4][4][256];
    const int n = blockIdx.y, s = blockIdx.x;
    const int C4 = C >> 2, Wo = W >> 1, HWo = (H >> 1) * Wo;
    const int tcn = C4 < 256 ? C4 : 256;
    const int rows = 256 / tcn;
    const int t = threadIdx.x;
    const int tc = t % tcn, tr = t / tcn;
    const int per = (HWo + splits - 1) / splits;
    const int w0 = s * per;
    const int w1 = (w0 + per < HWo) ? w0 + per : HWo;
    const bool active = tr < rows;
    for (int cb = 0; cb < C4; cb += tcn) {
        const int c4 = cb + tc;
        double acc[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[q][k] = 0.0;
        if (active && c4 < C4) {
            const float4* ma = (const float4*)(mra + 2 * ((long)n * C + c4 * 4));
            const float4* mb = (const float4*)(mrb + 2 * ((long)n * C + c4 * 4));
            const float4 a0 = ma[0], a1 = ma[1], b0 = mb[0], b1 = mb[1];
            const float am[4] = {a0.x, a0.z, a1.x, a1.z}, ar[4] = {a0.y, a0.w, a1.y, a1.w};
            const float bm[4] = {b0.x, b0.z, b1.x, b1.z}, br[4] = {b0.y, b0.w, b1.y, b1.w};
            for (int wv = w0 + tr; wv < w1; wv += rows) {
                const int ho = wv / Wo, wo = wv - ho * Wo;
                const long b = (((long)n * H + 2 * ho) * W + 2 * wo) * C4 + c4;
                const long idx[4] = {b, b + C4, b + (long)W * C4, b + (long)W * C4 + C4};
                float4 v[4], g[4], va[4], vb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[k] = out[idx[k]]; va[k] = xa[idx[k]]; vb[k] = xb[idx[k]]; }
                const float4 gy = gp ? gp[((long)n * HWo + wv) * C4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] = go ? go[idx[k]] : make_float4(0.f, 0.f, 0.f, 0.f);
                const float* vf = (const float*)v;
                float* gf = (float*)g;
                const float* af = (const float*)va;
                const float* bf = (const float*)vb;
                const float gyf[4] = {gy.x, gy.y, gy.z, gy.w};
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    float m = vf[ch];
                    int amx = 0;
#pragma unroll
                    for (int k = 1; k < 4; ++k)
                        if (vf[4 * k + ch] > m) { m = vf[4 * k + ch]; amx = k; }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float tt = gf[4 * k + ch] + (k == amx ? gyf[ch] : 0.f);
                        const float gg = vf[4 * k + ch] > 0.f ? tt : 0.f;
                        gf[4 * k + ch] = gg;
                        const float xha = (af[4 * k + ch] - am[ch]) * ar[ch], xhb = (bf[4 * k + ch] - bm[ch]) * br[ch];
                        const float qa = !(xha > 0.f) ? 0.f : gg;        // branch a: ReLU after the norm
                        acc[0][ch] += (double)qa;
                        acc[1][ch] += (double)(qa * xha);
                        acc[2][ch] += (double)gg;
                        acc[3][ch] += (double)(gg * xhb);
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) gw[idx[k]] = g[k];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) sq[q][k][t] = acc[q][k];
        __syncthreads();
        if (tr == 0 && c4 < C4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double tq[4] = {acc[0][k], acc[1][k], acc[2][k], acc[3][k]};
                for (int r = 1; r < rows; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q) tq[q] += sq[q][k][r * tcn + tc];
                double* oa = parta + (((long)n * splits + s) * C + c4 * 4 + k) * 2;
                double* ob = partb + (((long)n * splits + s) * C + c4 * 4 + k) * 2;
                oa[0] = tq[0]; oa[1] = tq[1];
                ob[0] = tq[2]; ob[1] = tq[3];
            }
        }
        __syncthreads();
    }
}
__global__ void k_inorm_bwd_pair_apply4(const float4* __restrict__ xa, const float* __restrict__ mra, const float* __restrict__ ea,
                                        const float4* __restrict__ xb, const float* __restrict__ mrb, const float* __restrict__ eb,
                                        const float4* __restrict__ gy, float4* __restrict__ gxa, float4* __restrict__ gxb,
                                        long total4, int HW, int C4) {
    long stride = (long)gridDim.x * blockDim.x;
    long plane4 = (long)HW * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
        const int c = (int)(i % C4) * 4;
        const int n = (int)(i / plane4);
        const long k = 2 * ((long)n * C4 * 4 + c);
        const float4* m = (const float4*)(mra + k);
        const float4* e = (const float4*)(ea + k);
        float4 m0 = m[0], m1 = m[1], e0 = e[0], e1 = e[1];
        const float4 g = gy[i];
        float4 v = xa[i], o;
        float xh, gg;
        xh = (v.x - m0.x) * m0.y; gg = !(xh > 0.f) ? 0.f : g.x; o.x = m0.y * (gg - e0.x - xh * e0.y);
        xh = (v.y - m0.z) * m0.w; gg = !(xh > 0.f) ? 0.f : g.y; o.y = m0.w * (gg - e0.z - xh * e0.w);
        xh = (v.z - m1.x) * m1.y; gg = !(xh > 0.f) ? 0.f : g.z; o.z = m1.y * (gg - e1.x - xh * e1.y);
        xh = (v.w - m1.z) * m1.w; gg = !(xh > 0.f) ? 0.f : g.w; o.w = m1.w * (gg - e1.z - xh * e1.w);
        gxa[i] = o;
        m = (const float4*)(mrb + k);
        e = (const float4*)(eb + k);
        m0 = m[0]; m1 = m[1]; e0 = e[0]; e1 = e[1];
        v = xb[i];
        xh = (v.x - m0.x) * m0.y; o.x = m0.y * (g.x - e0.x - xh * e0.y);
        xh = (v.y - m0.z) * m0.w; o.y = m0.w * (g.y - e0.z - xh * e0.w);
        xh = (v.z - m1.x) * m1.y; o.z = m1.y * (g.z - e1.x - xh * e1.y);
        xh = (v.w - m1.z) * m1.w; o.w = m1.w * (g.w - e1.z - xh * e1.w);
        gxb[i] = o;
    }
}
// a: InstanceNorm + ReLU, b: InstanceNorm; both get gy.  ws: 2 x vqw_plane_ws_bytes(N, C, HW).
extern "C" int vqw_inorm_bwd_pair(const float* xa, const float* mra, const float* xb, const float* mrb, const float* gy, float* gxa,
                                  float* gxb, void* ws, size_t ws_bytes, int N, int HW, int C, void* stream) {
    VQW_PROF_HBM(stream, 8, (double)N * HW * C);
    VQW_CHECK(xa && mra && xb && mrb && gy && gxa && gxb && ws && N > 0 && HW > 0 && C > 0, "vqw_inorm_bwd_pair: bad arguments");
    VQW_CHECK((C & 3) == 0, "vqw_inorm_bwd_pair: C %% 4 == 0");
    VQW_CHECK(((((uintptr_t)xa | (uintptr_t)xb | (uintptr_t)gy | (uintptr_t)gxa | (uintptr_t)gxb | (uintptr_t)mra | (uintptr_t)mrb) & 15) == 0),
              "vqw_inorm_bwd_pair: 16-byte alignment");
    const size_t one = vqw_plane_ws_bytes(N, C, HW);
    VQW_CHECK(ws_bytes >= 2 * one && (one & 15) == 0, "vqw_inorm_bwd_pair: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int splits = plane_splits(N, HW);
    double* parta = (double*)ws;
    float* ea = (float*)((char*)ws + plane_part_bytes(N, C));
    double* partb = (double*)((char*)ws + one);
    float* eb = (float*)((char*)ws + one + plane_part_bytes(N, C));
    k_inorm_bwd_pair_reduce4<<<dim3(splits, N), 256, 0, st>>>((const float4*)xa, mra, (const float4*)xb, mrb, (const float4*)gy, parta,
                                                               partb, HW, C, splits);
    k_plane_sum_finalize2<<<dim3(ceil_div((long)N * C, 4), 2), 256, 0, st>>>(parta, ea, partb, eb, N * C, C, splits, 1.0 / (double)HW);
    const long t4 = (long)N * HW * C / 4;
    if (walk_ok(C / 4)) {
        const int R = 256 / (C / 4);
        const dim3 g(imax(1, imin(ceil_div(2048, N), ceil_div(HW, 2 * R))), N);
        k_inorm_bwd_pair_apply4w<<<g, 256, 0, st>>>((const float4*)xa, mra, ea, (const float4*)xb, mrb, eb, (const float4*)gy, (float4*)gxa,
                                                     (float4*)gxb, HW, C / 4, ilog2_exact(C / 4));
    } else
    k_inorm_bwd_pair_apply4<<<stream_grid(t4, 256), 256, 0, st>>>((const float4*)xa, mra, ea, (const float4*)xb, mrb, eb,
                                                                  (const float4*)gy, (float4*)gxa, (float4*)gxb, t4, HW, C / 4);
    VQW_LAUNCH_CHECK("vqw_inorm_bwd_pair");
    return VQW_OK;
}

// The backward of a ResBlock's tail and of the two norms in front of it as one entry point: g (N, H, W, C: the gradient in front
// of the tail's ReLU, a workspace tensor of the caller) is written by the fused first kernel and read by the apply pass only.
extern "C" int vqw_res_tail_bwd_pair(const float* out, const float* g_pooled, const float* g_out, const float* xa, const float* mra,
                                     const float* xb, const float* mrb, float* g, float* gxa, float* gxb, void* ws, size_t ws_bytes,
                                     int N, int H, int W, int C, void* stream) {
    const int HW = H * W;
    VQW_PROF_HBM(stream, 10.25, (double)N * HW * C);
    VQW_CHECK(out && xa && mra && xb && mrb && g && gxa && gxb && ws && N > 0 && H > 0 && W > 0 && C > 0, "vqw_res_tail_bwd_pair: bad arguments");
    VQW_CHECK((C & 3) == 0 && (H & 1) == 0 && (W & 1) == 0, "vqw_res_tail_bwd_pair: needs even H, W and C %% 4 == 0");
    VQW_CHECK(((((uintptr_t)out | (uintptr_t)g_pooled | (uintptr_t)g_out | (uintptr_t)xa | (uintptr_t)xb | (uintptr_t)g | (uintptr_t)gxa |
                 (uintptr_t)gxb | (uintptr_t)mra | (uintptr_t)mrb) & 15) == 0), "vqw_res_tail_bwd_pair: 16-byte alignment");
    const size_t one = vqw_plane_ws_bytes(N, C, HW);
    VQW_CHECK(ws_bytes >= 2 * one && (one & 15) == 0, "vqw_res_tail_bwd_pair: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int splits = plane_splits(N, HW);
    double* parta = (double*)ws;
    float* ea = (float*)((char*)ws + plane_part_bytes(N, C));
    double* partb = (double*)((char*)ws + one);
    float* eb = (float*)((char*)ws + one + plane_part_bytes(N, C));
    k_res_tail_bwd_pair_reduce4<<<dim3(splits, N), 256, 0, st>>>((const float4*)out, (const float4*)g_pooled, (const float4*)g_out,
                                                                  (const float4*)xa, mra, (const float4*)xb, mrb, (float4*)g, parta, partb, H, W, C,
                                                                  splits);
    k_plane_sum_finalize2<<<dim3(ceil_div((long)N * C, 4), 2), 256, 0, st>>>(parta, ea, partb, eb, N * C, C, splits, 1.0 / (double)HW);
    const long t4 = (long)N * HW * C / 4;
    if (walk_ok(C / 4)) {
        const int R = 256 / (C / 4);
        const dim3 gr(imax(1, imin(ceil_div(2048, N), ceil_div(HW, 2 * R))), N);
        k_inorm_bwd_pair_apply4w<<<gr, 256, 0, st>>>((const float4*)xa, mra, ea, (const float4*)xb, mrb, eb, (const float4*)g, (float4*)gxa,
                                                      (float4*)gxb, HW, C / 4, ilog2_exact(C / 4));
    } else
    k_inorm_bwd_pair_apply4<<<stream_grid(t4, 256), 256, 0, st>>>((const float4*)xa, mra, ea, (const float4*)xb, mrb, eb,
                                                                  (const float4*)g, (float4*)gxa, (float4*)gxb, t4, HW, C / 4);
    VQW_LAUNCH_CHECK("vqw_res_tail_bwd_pair");
    return VQW_OK;
}

// ---------------------------------------------------------------------------------------------
// BatchNorm statistics (per channel over N*H*W) -> double sums[C][2] so ranks can be summed (SyncBN).
// sums[c] = sum over `rows` partial rows; one workgroup per channel: 256 threads take the rows round-robin, then a
// fixed-order tree through LDS -> deterministic (16 row groups per channel walked 128 rows each one load after the other)
__global__ void __launch_bounds__(256) k_channel_sum_finalize(const double* __restrict__ part, double* __restrict__ sums, int C, int rows) {
    __shared__ double sa[256], sb[256];
    const int c = blockIdx.x, t = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int r = t; r < rows; r += 256) {
        const double* o = part + ((long)r * C + c) * 2;
        a += o[0];
        b += o[1];
    }
    sa[t] = a;
    sb[t] = b;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) { sa[t] += sa[t + w]; sb[t] += sb[t + w]; }
        __syncthreads();
    }
    if (t == 0) {
        sums[2 * c] = sa[0];
        sums[2 * c + 1] = sb[0];
    }
}

// the same from the per-tile float partials part[rows][C][2] a convolution's epilogue left (vqw_conv2d_fwd_stats)
__global__ void __launch_bounds__(256) k_channel_sum_finalize_f(const float* __restrict__ part, double* __restrict__ sums, int C, int rows,
                                                                double inv_tile) {
    __shared__ double sa[256], sb[256];
    const int c = blockIdx.x, t = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int r = t; r < rows; r += 256) {
        const float* o = part + ((long)r * C + c) * 2;          // (sum, M2 about the tile mean)
        const double st = (double)o[0];
        a += st;
        b += (double)o[1] + st * st * inv_tile;
    }
    sa[t] = a;
    sb[t] = b;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) { sa[t] += sa[t + w]; sb[t] += sb[t + w]; }
        __syncthreads();
    }
    if (t == 0) {
        sums[2 * c] = sa[0];
        sums[2 * c + 1] = sb[0];
    }
}
extern "C" int vqw_bn_stats_from_parts(const float* part, double* sums, int rows, int C, double tile_count, void* stream) {
    VQW_CHECK(part && sums && rows > 0 && C > 0 && tile_count >= 1.0, "vqw_bn_stats_from_parts: bad arguments");
    k_channel_sum_finalize_f<<<C, 256, 0, (hipStream_t)stream>>>(part, sums, C, rows, 1.0 / tile_count);
    VQW_LAUNCH_CHECK("vqw_bn_stats_from_parts");
    return VQW_OK;
}

extern "C" int vqw_bn_partial_stats(const float* x, double* sums, void* ws, size_t ws_bytes, int N, int HW, int C,
                                    void* stream) {
    VQW_PROF_HBM(stream, 1, (double)N * HW * C);
    VQW_CHECK(x && sums && ws && N > 0 && HW > 0 && C > 0, "vqw_bn_partial_stats: bad arguments");
    VQW_CHECK(ws_bytes >= vqw_plane_ws_bytes(N, C, HW), "vqw_bn_partial_stats: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int splits = plane_splits(N, HW);
    if ((C & 3) == 0 && al16(x)) {
        FStats4 f{(const float4*)x};
        splits = launch_plane_reduce4(f, (double*)ws, N, HW, C, st);
    } else {
        FStats f{x};
        k_plane_reduce<<<dim3(splits, N), 256, 0, st>>>(f, (double*)ws, HW, C, splits);
    }
    k_channel_sum_finalize<<<C, 256, 0, st>>>((const double*)ws, sums, C, N * splits);
    VQW_LAUNCH_CHECK("vqw_bn_partial_stats");
    return VQW_OK;
}

__global__ void k_bn_finalize(const double* __restrict__ sums, double count, float* __restrict__ mr, float* __restrict__ rm,
                              float* __restrict__ rv, float momentum, float eps, int C) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double mean = sums[2 * c] / count;
    double var = sums[2 * c + 1] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    mr[2 * c] = (float)mean;
    mr[2 * c + 1] = (float)(1.0 / sqrt(var + (double)eps));
    if (rm) {
        double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
        rm[c] = (1.f - momentum) * rm[c] + momentum * (float)mean;
        rv[c] = (1.f - momentum) * rv[c] + momentum * (float)unb;
    }
}
// k_channel_sum_finalize_f + k_bn_finalize in one launch (no collective between them: one GPU, or SyncBN off): same sums, same
// arithmetic; `sums` is still written (the caller may want it)
__global__ void __launch_bounds__(256) k_bn_finalize_parts(const float* __restrict__ part, double* __restrict__ sums, int C, int rows,
                                                           double inv_tile, double count, float* __restrict__ mr, float* __restrict__ rm,
                                                           float* __restrict__ rv, float momentum, float eps) {
    __shared__ double sa[256], sb[256];
    const int c = blockIdx.x, t = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int r = t; r < rows; r += 256) {
        const float* o = part + ((long)r * C + c) * 2;
        const double st = (double)o[0];
        a += st;
        b += (double)o[1] + st * st * inv_tile;
    }
    sa[t] = a;
    sb[t] = b;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) { sa[t] += sa[t + w]; sb[t] += sb[t + w]; }
        __syncthreads();
    }
    if (t == 0) {
        sums[2 * c] = sa[0];
        sums[2 * c + 1] = sb[0];
        double mean = sa[0] / count;
        double var = sb[0] / count - mean * mean;
        if (var < 0.0) var = 0.0;
        mr[2 * c] = (float)mean;
        mr[2 * c + 1] = (float)(1.0 / sqrt(var + (double)eps));
        if (rm) {
            double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
            rm[c] = (1.f - momentum) * rm[c] + momentum * (float)mean;
            rv[c] = (1.f - momentum) * rv[c] + momentum * (float)unb;
        }
    }
}
extern "C" int vqw_bn_finalize_parts(const float* part, int rows, double tile_count, double* sums, double count, float* mean_rstd,
                                     float* running_mean, float* running_var, float momentum, float eps, int C, void* stream) {
    VQW_CHECK(part && sums && mean_rstd && rows > 0 && C > 0 && tile_count >= 1.0 && count > 0, "vqw_bn_finalize_parts: bad arguments");
    VQW_CHECK((running_mean == nullptr) == (running_var == nullptr), "vqw_bn_finalize_parts: running stats must both be set or both NULL");
    k_bn_finalize_parts<<<C, 256, 0, (hipStream_t)stream>>>(part, sums, C, rows, 1.0 / tile_count, count, mean_rstd, running_mean, running_var,
                                                            momentum, eps);
    VQW_LAUNCH_CHECK("vqw_bn_finalize_parts");
    return VQW_OK;
}
extern "C" int vqw_bn_finalize(const double* sums, double count, float* mean_rstd, float* running_mean,
                               float* running_var, float momentum, float eps, int C, void* stream) {
    VQW_CHECK(sums && mean_rstd && count > 0 && C > 0, "vqw_bn_finalize: bad arguments");
    VQW_CHECK((running_mean == nullptr) == (running_var == nullptr), "vqw_bn_finalize: running stats must both be set or both NULL");
    k_bn_finalize<<<ceil_div(C, 256), 256, 0, (hipStream_t)stream>>>(sums, count, mean_rstd, running_mean, running_var,
                                                                     momentum, eps, C);
    VQW_LAUNCH_CHECK("vqw_bn_finalize");
    return VQW_OK;
}
__global__ void k_bn_eval_stats(const float* __restrict__ rm, const float* __restrict__ rv, float* __restrict__ mr, float eps, int C) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mr[2 * c] = rm[c];
    mr[2 * c + 1] = 1.f / sqrtf(rv[c] + eps);
}
extern "C" int vqw_bn_eval_stats(const float* running_mean, const float* running_var, float* mean_rstd, float eps,
                                 int C, void* stream) {
    VQW_CHECK(running_mean && running_var && mean_rstd && C > 0, "vqw_bn_eval_stats: bad arguments");
    k_bn_eval_stats<<<ceil_div(C, 256), 256, 0, (hipStream_t)stream>>>(running_mean, running_var, mean_rstd, eps, C);
    VQW_LAUNCH_CHECK("vqw_bn_eval_stats");
    return VQW_OK;
}

template <int RELU>
__global__ void k_spade_fwd(const float* __restrict__ x, const float* __restrict__ mr, const float* __restrict__ gamma,
                            const float* __restrict__ beta, float* __restrict__ y, long total, int C, int gbs) {
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int c = (int)(i % C);
        long j = (i / C) * gbs + c;
        float xh = (x[i] - mr[2 * c]) * mr[2 * c + 1];
        float v = xh * (1.f + gamma[j]) + beta[j];
        y[i] = RELU ? fmaxf(v, 0.f) : v;
    }
}
template <int RELU>
__global__ void __launch_bounds__(256) k_spade_fwd4(const float4* __restrict__ x, const float* __restrict__ mr,
                                                    const float4* __restrict__ gamma, const float4* __restrict__ beta,
                                                    float4* __restrict__ y, long total4, int C4, int gbs4,
                                                    const float4* __restrict__ res = nullptr) {
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
        int c4 = (int)(i % C4);
        long j = (i / C4) * gbs4 + c4;
        const float4* m = (const float4*)(mr + 8 * c4);
        float4 m0 = m[0], m1 = m[1], v = x[i], ga = gamma[j], be = beta[j], o;
        o.x = (v.x - m0.x) * m0.y * (1.f + ga.x) + be.x;
        o.y = (v.y - m0.z) * m0.w * (1.f + ga.y) + be.y;
        o.z = (v.z - m1.x) * m1.y * (1.f + ga.z) + be.z;
        o.w = (v.w - m1.z) * m1.w * (1.f + ga.w) + be.w;
        if (RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        if (res) {      // residual added AFTER the activation (StyledResUpBlock: shortcut + main path, blocks.py:134)
            const float4 r = res[i];
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        y[i] = o;
    }
}
// y = act(spade(x)) + res: the block's final `shortcut + main` (blocks.py:134) inside the last modulation kernel
extern "C" int vqw_spade_fwd_res(const float* x, const float* mean_rstd, const float* gamma, const float* beta, int gb_stride,
                                 const float* res, float* y, long P, int C, int relu, void* stream) {
    VQW_PROF_HBM(stream, 5, (double)P * C);
    VQW_CHECK(x && mean_rstd && gamma && beta && res && y && P > 0 && C > 0 && gb_stride >= C, "vqw_spade_fwd_res: bad arguments");
    VQW_CHECK((C & 3) == 0 && (gb_stride & 3) == 0 && al16(x) && al16(gamma) && al16(beta) && al16(y) && al16(mean_rstd) && al16(res),
              "vqw_spade_fwd_res: needs C %% 4 == 0 and 16-byte aligned tensors");
    const long t4 = P * C / 4;
    hipStream_t st = (hipStream_t)stream;
    if (walk_ok(C / 4)) {
        const int C4 = C / 4, gr = walk_blocks_flat(P, C4, 4);
        if (relu) k_spade_fwd4w<1, 1><<<gr, 256, 0, st>>>((const float4*)x, mean_rstd, (const float4*)gamma, (const float4*)beta, (float4*)y, P, C4, ilog2_exact(C4), gb_stride / 4, (const float4*)res);
        else k_spade_fwd4w<0, 1><<<gr, 256, 0, st>>>((const float4*)x, mean_rstd, (const float4*)gamma, (const float4*)beta, (float4*)y, P, C4, ilog2_exact(C4), gb_stride / 4, (const float4*)res);
    } else
    if (relu) k_spade_fwd4<1><<<stream_grid(t4, 256), 256, 0, st>>>((const float4*)x, mean_rstd, (const float4*)gamma, (const float4*)beta, (float4*)y, t4, C / 4, gb_stride / 4, (const float4*)res);
    else k_spade_fwd4<0><<<stream_grid(t4, 256), 256, 0, st>>>((const float4*)x, mean_rstd, (const float4*)gamma, (const float4*)beta, (float4*)y, t4, C / 4, gb_stride / 4, (const float4*)res);
    VQW_LAUNCH_CHECK("vqw_spade_fwd_res");
    return VQW_OK;
}
// y = act(spade(x)) + InstanceNorm(+ReLU)(res_raw): the block's shortcut branch normalised while it is read (its statistics
// res_mean_rstd [N][C][2] from vqw_inorm_stats_parts / vqw_inorm_stats) - the normalised shortcut tensor is never materialised
extern "C" int vqw_spade_fwd_res_norm_supported(long HW, int C) {
    return (C % 4 == 0 && walk_ok(C / 4) && HW > 0 && (HW & (HW - 1)) == 0) ? 1 : 0;
}
extern "C" int vqw_spade_fwd_res_norm(const float* x, const float* mean_rstd, const float* gamma, const float* beta, int gb_stride,
                                      const float* res_raw, const float* res_mean_rstd, int res_relu, float* y, int N, long HW, int C,
                                      int relu, void* stream) {
    VQW_PROF_HBM(stream, 5, (double)N * HW * C);
    VQW_CHECK(x && mean_rstd && gamma && beta && res_raw && res_mean_rstd && y && N > 0 && HW > 0 && C > 0 && gb_stride >= C,
              "vqw_spade_fwd_res_norm: bad arguments");
    VQW_CHECK(vqw_spade_fwd_res_norm_supported(HW, C) && (gb_stride & 3) == 0 && al16(x) && al16(gamma) && al16(beta) && al16(y) &&
                  al16(mean_rstd) && al16(res_raw) && al16(res_mean_rstd),
              "vqw_spade_fwd_res_norm: shape not served (query vqw_spade_fwd_res_norm_supported) or unaligned tensors");
    const long P = (long)N * HW;
    int lg = 0;
    while ((1L << lg) < HW) ++lg;
    const int C4 = C / 4, gr = walk_blocks_flat(P, C4, 4);
    hipStream_t st = (hipStream_t)stream;
    if (relu) k_spade_fwd4w<1, 2><<<gr, 256, 0, st>>>((const float4*)x, mean_rstd, (const float4*)gamma, (const float4*)beta, (float4*)y, P, C4, ilog2_exact(C4), gb_stride / 4, (const float4*)res_raw, res_mean_rstd, lg, res_relu);
    else k_spade_fwd4w<0, 2><<<gr, 256, 0, st>>>((const float4*)x, mean_rstd, (const float4*)gamma, (const float4*)beta, (float4*)y, P, C4, ilog2_exact(C4), gb_stride / 4, (const float4*)res_raw, res_mean_rstd, lg, res_relu);
    VQW_LAUNCH_CHECK("vqw_spade_fwd_res_norm");
    return VQW_OK;
}
extern "C" int vqw_spade_fwd(const float* x, const float* mean_rstd, const float* gamma, const float* beta, int gb_stride,
                             float* y, long P, int C, int relu, void* stream) {
    VQW_PROF_HBM(stream, 4, (double)P * C);
    VQW_CHECK(x && mean_rstd && gamma && beta && y && P > 0 && C > 0 && gb_stride >= C, "vqw_spade_fwd: bad arguments");
    long total = P * C;
    hipStream_t st = (hipStream_t)stream;
    if ((C & 3) == 0 && (gb_stride & 3) == 0 && al16(x) && al16(gamma) && al16(beta) && al16(y) && al16(mean_rstd)) {
        long t4 = total / 4;
        if (walk_ok(C / 4)) {
            const int C4 = C / 4, gr = walk_blocks_flat(P, C4, 4);
            if (relu) k_spade_fwd4w<1, 0><<<gr, 256, 0, st>>>((const float4*)x, mean_rstd, (const float4*)gamma, (const float4*)beta, (float4*)y, P, C4, ilog2_exact(C4), gb_stride / 4, nullptr);
            else k_spade_fwd4w<0, 0><<<gr, 256, 0, st>>>((const float4*)x, mean_rstd, (const float4*)gamma, (const float4*)beta, (float4*)y, P, C4, ilog2_exact(C4), gb_stride / 4, nullptr);
        } else
        if (relu) k_spade_fwd4<1><<<stream_grid(t4, 256), 256, 0, st>>>((const float4*)x, mean_rstd, (const float4*)gamma, (const float4*)beta, (float4*)y, t4, C / 4, gb_stride / 4);
        else k_spade_fwd4<0><<<stream_grid(t4, 256), 256, 0, st>>>((const float4*)x, mean_rstd, (const float4*)gamma, (const float4*)beta, (float4*)y, t4, C / 4, gb_stride / 4);
    } else if (relu) k_spade_fwd<1><<<stream_grid(total, 256), 256, 0, st>>>(x, mean_rstd, gamma, beta, y, total, C, gb_stride);
    else k_spade_fwd<0><<<stream_grid(total, 256), 256, 0, st>>>(x, mean_rstd, gamma, beta, y, total, C, gb_stride);
    VQW_LAUNCH_CHECK("vqw_spade_fwd");
    return VQW_OK;
}

// backward phase 1: g = gy * [out>0]; dgamma = g*xhat; dbeta = g; dxhat = g*(1+gamma);
// per-channel sums of (dxhat, dxhat*xhat).  dgamma/dbeta are written as a side effect of the reduction.
template <int RELU>
struct FSpadeBwd {
    const float* x;
    const float* mr;
    const float* gamma;
    const float* beta;
    const float* gy;
    float* dgamma;
    float* dbeta;
    int C, gbs;
    __device__ void operator()(long i, int, int c, float& a, float& b) const {
        const long j = (i / C) * gbs + c;
        float xh = (x[i] - mr[2 * c]) * mr[2 * c + 1];
        float ga = 1.f + gamma[j];
        float g = gy[i];
        if (RELU) {
            float out = xh * ga + beta[j];
            if (!(out > 0.f)) g = 0.f;
        }
        dgamma[j] = g * xh;
        dbeta[j] = g;
        float dxh = g * ga;
        a = dxh;
        b = dxh * xh;
    }
};
extern "C" int vqw_spade_bwd_reduce(const float* x, const float* mean_rstd, const float* gamma, const float* beta,
                                    const float* gy, float* dgamma, float* dbeta, int gb_stride, double* sums, void* ws,
                                    size_t ws_bytes, int N, int HW, int C, int relu, void* stream) {
    VQW_PROF_HBM(stream, 6, (double)N * HW * C);
    VQW_CHECK(x && mean_rstd && gamma && beta && gy && dgamma && dbeta && sums && ws && N > 0 && HW > 0 && C > 0 &&
                  gb_stride >= C, "vqw_spade_bwd_reduce: bad arguments");
    VQW_CHECK(ws_bytes >= vqw_plane_ws_bytes(N, C, HW), "vqw_spade_bwd_reduce: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int splits = plane_splits(N, HW);
    const bool vec = (C & 3) == 0 && (gb_stride & 3) == 0 && al16(x) && al16(gamma) && al16(beta) && al16(gy) && al16(dgamma) && al16(dbeta) && al16(mean_rstd);
    if (vec && relu) {
        FSpadeBwd4<1> f{(const float4*)x, mean_rstd, (const float4*)gamma, (const float4*)beta, (const float4*)gy, (float4*)dgamma, (float4*)dbeta, C / 4, gb_stride / 4};
        splits = launch_plane_reduce4(f, (double*)ws, N, HW, C, st);
    } else if (vec) {
        FSpadeBwd4<0> f{(const float4*)x, mean_rstd, (const float4*)gamma, (const float4*)beta, (const float4*)gy, (float4*)dgamma, (float4*)dbeta, C / 4, gb_stride / 4};
        splits = launch_plane_reduce4(f, (double*)ws, N, HW, C, st);
    } else if (relu) {
        FSpadeBwd<1> f{x, mean_rstd, gamma, beta, gy, dgamma, dbeta, C, gb_stride};
        k_plane_reduce<<<dim3(splits, N), 256, 0, st>>>(f, (double*)ws, HW, C, splits);
    } else {
        FSpadeBwd<0> f{x, mean_rstd, gamma, beta, gy, dgamma, dbeta, C, gb_stride};
        k_plane_reduce<<<dim3(splits, N), 256, 0, st>>>(f, (double*)ws, HW, C, splits);
    }
    k_channel_sum_finalize<<<C, 256, 0, st>>>((const double*)ws, sums, C, N * splits);
    VQW_LAUNCH_CHECK("vqw_spade_bwd_reduce");
    return VQW_OK;
}

template <int RELU, int TRAIN>
__global__ void k_spade_bwd_apply(const float* __restrict__ x, const float* __restrict__ mr, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, const float* __restrict__ gy, const double* __restrict__ sums,
                                  double inv_count, float* __restrict__ gx, long total, int C, int gbs) {
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int c = (int)(i % C);
        long j = (i / C) * gbs + c;
        float r = mr[2 * c + 1];
        float xh = (x[i] - mr[2 * c]) * r;
        float ga = 1.f + gamma[j];
        float g = gy[i];
        if (RELU) {
            float out = xh * ga + beta[j];
            if (!(out > 0.f)) g = 0.f;
        }
        float dxh = g * ga;
        if (TRAIN) {
            float m1 = (float)(sums[2 * c] * inv_count);
            float m2 = (float)(sums[2 * c + 1] * inv_count);
            gx[i] = r * (dxh - m1 - xh * m2);
        } else {
            gx[i] = r * dxh;
        }
    }
}
// float4 form: same per-element arithmetic, four channels per lane
template <int RELU, int TRAIN>
__global__ void __launch_bounds__(256) k_spade_bwd_apply4(const float4* __restrict__ x, const float* __restrict__ mr,
                                                          const float4* __restrict__ gamma, const float4* __restrict__ beta,
                                                          const float4* __restrict__ gy, const double* __restrict__ sums,
                                                          double inv_count, float4* __restrict__ gx, long total4, int C4, int gbs4) {
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
        int c4 = (int)(i % C4);
        long j = (i / C4) * gbs4 + c4;
        const float4* m = (const float4*)(mr + 8 * c4);
        float4 m0 = m[0], m1 = m[1], v = x[i], ga4 = gamma[j], g4 = gy[i], be4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (RELU) be4 = beta[j];
        float mean[4] = {m0.x, m0.z, m1.x, m1.z}, rs[4] = {m0.y, m0.w, m1.y, m1.w};
        float xv[4] = {v.x, v.y, v.z, v.w}, gav[4] = {ga4.x, ga4.y, ga4.z, ga4.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w};
        float bev[4] = {be4.x, be4.y, be4.z, be4.w}, o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float r = rs[k];
            float xh = (xv[k] - mean[k]) * r;
            float ga = 1.f + gav[k];
            float g = gv[k];
            if (RELU) {
                float out = xh * ga + bev[k];
                if (!(out > 0.f)) g = 0.f;
            }
            float dxh = g * ga;
            if (TRAIN) {
                float s1 = (float)(sums[2 * (4 * c4 + k)] * inv_count);
                float s2 = (float)(sums[2 * (4 * c4 + k) + 1] * inv_count);
                o[k] = r * (dxh - s1 - xh * s2);
            } else {
                o[k] = r * dxh;
            }
        }
        gx[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}
extern "C" int vqw_spade_bwd_apply(const float* x, const float* mean_rstd, const float* gamma, const float* beta, int gb_stride,
                                   const float* gy, const double* sums, double count, float* gx, long P, int C,
                                   int relu, int training, void* stream) {
    VQW_PROF_HBM(stream, 4, (double)P * C);
    VQW_CHECK(x && mean_rstd && gamma && beta && gy && gx && P > 0 && C > 0 && gb_stride >= C, "vqw_spade_bwd_apply: bad arguments");
    VQW_CHECK(!training || (sums && count > 0), "vqw_spade_bwd_apply: training needs sums and count");
    long total = P * C;
    hipStream_t st = (hipStream_t)stream;
    double ic = training ? 1.0 / count : 0.0;
    if ((C & 3) == 0 && (gb_stride & 3) == 0 && al16(x) && al16(gamma) && al16(beta) && al16(gy) && al16(gx) && al16(mean_rstd)) {
        long t4 = total / 4;
        int g = stream_grid(t4, 256), C4 = C / 4, s4 = gb_stride / 4;
        const float4 *x4 = (const float4*)x, *ga4 = (const float4*)gamma, *be4 = (const float4*)beta, *gy4 = (const float4*)gy;
        if (walk_ok(C4)) {
            const int gw = walk_blocks_flat(P, C4, 2), lg = ilog2_exact(C4);
            if (relu && training) k_spade_bwd_apply4w<1, 1><<<gw, 256, 0, st>>>(x4, mean_rstd, ga4, be4, gy4, sums, ic, (float4*)gx, P, C4, lg, s4);
            else if (relu) k_spade_bwd_apply4w<1, 0><<<gw, 256, 0, st>>>(x4, mean_rstd, ga4, be4, gy4, sums, ic, (float4*)gx, P, C4, lg, s4);
            else if (training) k_spade_bwd_apply4w<0, 1><<<gw, 256, 0, st>>>(x4, mean_rstd, ga4, be4, gy4, sums, ic, (float4*)gx, P, C4, lg, s4);
            else k_spade_bwd_apply4w<0, 0><<<gw, 256, 0, st>>>(x4, mean_rstd, ga4, be4, gy4, sums, ic, (float4*)gx, P, C4, lg, s4);
        } else
        if (relu && training) k_spade_bwd_apply4<1, 1><<<g, 256, 0, st>>>(x4, mean_rstd, ga4, be4, gy4, sums, ic, (float4*)gx, t4, C4, s4);
        else if (relu) k_spade_bwd_apply4<1, 0><<<g, 256, 0, st>>>(x4, mean_rstd, ga4, be4, gy4, sums, ic, (float4*)gx, t4, C4, s4);
        else if (training) k_spade_bwd_apply4<0, 1><<<g, 256, 0, st>>>(x4, mean_rstd, ga4, be4, gy4, sums, ic, (float4*)gx, t4, C4, s4);
        else k_spade_bwd_apply4<0, 0><<<g, 256, 0, st>>>(x4, mean_rstd, ga4, be4, gy4, sums, ic, (float4*)gx, t4, C4, s4);
    } else {
        int g = stream_grid(total, 256);
        if (relu && training) k_spade_bwd_apply<1, 1><<<g, 256, 0, st>>>(x, mean_rstd, gamma, beta, gy, sums, ic, gx, total, C, gb_stride);
        else if (relu) k_spade_bwd_apply<1, 0><<<g, 256, 0, st>>>(x, mean_rstd, gamma, beta, gy, sums, ic, gx, total, C, gb_stride);
        else if (training) k_spade_bwd_apply<0, 1><<<g, 256, 0, st>>>(x, mean_rstd, gamma, beta, gy, sums, ic, gx, total, C, gb_stride);
        else k_spade_bwd_apply<0, 0><<<g, 256, 0, st>>>(x, mean_rstd, gamma, beta, gy, sums, ic, gx, total, C, gb_stride);
    }
    VQW_LAUNCH_CHECK("vqw_spade_bwd_apply");
    return VQW_OK;
}

// ---------------------------------------------------------------------------------------------
// BatchNorm2d with per-channel affine + LeakyReLU (PatchGAN discriminator, networks/discriminator.py:66-78):
//   y = lrelu(((x - mean) * rstd) * gamma + beta, slope)
// Statistics come from vqw_bn_partial_stats / vqw_bn_finalize (running stats, SyncBN-able sums) like StyledDenorm.
__global__ void k_bn_affine_fwd(const float* __restrict__ x, const float* __restrict__ mr, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float* __restrict__ y, long total, int C, float slope) {
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int c = (int)(i % C);
        float v = ((x[i] - mr[2 * c]) * mr[2 * c + 1]) * gamma[c] + beta[c];
        y[i] = v > 0.f ? v : v * slope;
    }
}
extern "C" int vqw_bn_affine_fwd(const float* x, const float* mean_rstd, const float* gamma, const float* beta, float* y, long P,
                                 int C, float slope, void* stream) {
    VQW_CHECK(x && mean_rstd && gamma && beta && y && P > 0 && C > 0, "vqw_bn_affine_fwd: bad arguments");
    long total = P * C;
    k_bn_affine_fwd<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(x, mean_rstd, gamma, beta, y, total, C, slope);
    VQW_LAUNCH_CHECK("vqw_bn_affine_fwd");
    return VQW_OK;
}

// backward phase 1: g' = gy * lrelu'(z); per-channel sums [sum g', sum g' * xhat]  (= dbeta, dgamma)
struct FBnAffineBwd {
    const float* x;
    const float* mr;
    const float* gamma;
    const float* beta;
    const float* gy;
    float slope;
    __device__ void operator()(long i, int, int c, float& a, float& b) const {
        float xh = (x[i] - mr[2 * c]) * mr[2 * c + 1];
        float z = xh * gamma[c] + beta[c];
        float g = z > 0.f ? gy[i] : gy[i] * slope;
        a = g;
        b = g * xh;
    }
};
extern "C" int vqw_bn_affine_bwd_reduce(const float* x, const float* mean_rstd, const float* gamma, const float* beta,
                                        const float* gy, double* sums, void* ws, size_t ws_bytes, int N, int HW, int C,
                                        float slope, void* stream) {
    VQW_CHECK(x && mean_rstd && gamma && beta && gy && sums && ws && N > 0 && HW > 0 && C > 0, "vqw_bn_affine_bwd_reduce: bad arguments");
    VQW_CHECK(ws_bytes >= vqw_plane_ws_bytes(N, C, HW), "vqw_bn_affine_bwd_reduce: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int splits = plane_splits(N, HW);
    FBnAffineBwd f{x, mean_rstd, gamma, beta, gy, slope};
    k_plane_reduce<<<dim3(splits, N), 256, 0, st>>>(f, (double*)ws, HW, C, splits);
    k_channel_sum_finalize<<<C, 256, 0, st>>>((const double*)ws, sums, C, N * splits);
    VQW_LAUNCH_CHECK("vqw_bn_affine_bwd_reduce");
    return VQW_OK;
}

// phase 2: dx = gamma * rstd * (g' - sum_g'/count - xhat * sum_g'xhat/count)  (training) or gamma * rstd * g' (eval);
// dgamma / dbeta (accumulated into when acc != 0) are written by the first C threads
__global__ void k_bn_affine_bwd_apply(const float* __restrict__ x, const float* __restrict__ mr, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, const float* __restrict__ gy,
                                      const double* __restrict__ sums, double inv_count, float* __restrict__ gx,
                                      float* __restrict__ dgamma, float* __restrict__ dbeta, long total, int C, float slope,
                                      int training, int acc) {
    long stride = (long)gridDim.x * blockDim.x;
    long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i0 < C && dgamma) {
        float dg = (float)sums[2 * i0 + 1], db = (float)sums[2 * i0];
        dgamma[i0] = acc ? dgamma[i0] + dg : dg;
        dbeta[i0] = acc ? dbeta[i0] + db : db;
    }
    for (long i = i0; i < total; i += stride) {
        int c = (int)(i % C);
        float r = mr[2 * c + 1];
        float xh = (x[i] - mr[2 * c]) * r;
        float z = xh * gamma[c] + beta[c];
        float g = z > 0.f ? gy[i] : gy[i] * slope;
        float v = g;
        if (training) v = g - (float)(sums[2 * c] * inv_count) - xh * (float)(sums[2 * c + 1] * inv_count);
        gx[i] = gamma[c] * r * v;
    }
}
extern "C" int vqw_bn_affine_bwd_apply(const float* x, const float* mean_rstd, const float* gamma, const float* beta,
                                       const float* gy, const double* sums, double count, float* gx, float* dgamma,
                                       float* dbeta, long P, int C, float slope, int training, int accumulate, void* stream) {
    VQW_CHECK(x && mean_rstd && gamma && beta && gy && sums && gx && P > 0 && C > 0 && count > 0, "vqw_bn_affine_bwd_apply: bad arguments");
    VQW_CHECK((dgamma == nullptr) == (dbeta == nullptr), "vqw_bn_affine_bwd_apply: dgamma and dbeta go together");
    long total = P * C;
    int grid = stream_grid(total, 256);
    if ((long)grid * 256 < C) grid = ceil_div(C, 256);
    k_bn_affine_bwd_apply<<<grid, 256, 0, (hipStream_t)stream>>>(x, mean_rstd, gamma, beta, gy, sums, 1.0 / count, gx, dgamma, dbeta,
                                                                 total, C, slope, training, accumulate);
    VQW_LAUNCH_CHECK("vqw_bn_affine_bwd_apply");
    return VQW_OK;
}
