// Generic direct convolution kernels (any channel count, VALU fp32) + weight packing, bias gradient and the
// gradient gather for the virtual (up-sampled / concatenated) input.  These are the fallback for shapes the
// MFMA implicit-GEMM kernels (conv_mfma.hip) do not take (Cin or Cout not a multiple of 16, e.g. the 1-channel
// stem and the 32->1 head), and the on-device cross-check for them in the tests.
#include "common.h"
#include <string.h>
#include "conv_common.h"
#include "../../include/vqwnet_hip.h"

// ---------------------------------------------------------------------------------------------
// forward: one thread per output element (pixel, co); adjacent threads = adjacent co (coalesced store).
__global__ void __launch_bounds__(256) k_conv_direct_fwd(ConvIn in, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ y, int N, int H, int W, int Cout, int ks, int dil,
                                                         int relu) {
    const int Cin = in.C0 + in.C1;
    const int taps = ks * ks, half = ks >> 1;
    long total = (long)N * H * W * Cout;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int co = (int)(i % Cout);
        long p = i / Cout;
        int x = (int)(p % W);
        long q = p / W;
        int yy = (int)(q % H);
        int n = (int)(q / H);
        float acc = bias ? bias[co] : 0.f;
        for (int t = 0; t < taps; ++t) {
            int hy = yy + (t / ks - half) * dil, wx = x + (t % ks - half) * dil;
            if (hy < 0 || hy >= H || wx < 0 || wx >= W) continue;
            const float* wr = w + ((long)co * taps + t) * Cin;
            const float* s0 = in.up0 ? in.src0 + (((long)n * (H >> 1) + (hy >> 1)) * (W >> 1) + (wx >> 1)) * in.C0
                                     : in.src0 + (((long)n * H + hy) * W + wx) * in.C0;
            for (int c = 0; c < in.C0; ++c) acc = fmaf(s0[c], wr[c], acc);
            if (in.C1 > 0) {
                const float* s1 = in.src1 + (((long)n * H + hy) * W + wx) * in.C1;
                for (int c = 0; c < in.C1; ++c) acc = fmaf(s1[c], wr[in.C0 + c], acc);
            }
        }
        y[i] = relu ? fmaxf(acc, 0.f) : acc;
    }
}

int conv_direct_fwd(const ConvIn& in, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int ks,
                    int dil, int relu, hipStream_t st) {
    long total = (long)N * H * W * Cout;
    k_conv_direct_fwd<<<stream_grid(total, 256), 256, 0, st>>>(in, w, bias, y, N, H, W, Cout, ks, dil, relu);
    VQW_LAUNCH_CHECK("conv_direct_fwd");
    return VQW_OK;
}

// ---------------------------------------------------------------------------------------------
// wt[ci][2-ky][2-kx][co] = w[co][ky][kx][ci]
__global__ void k_pack_dgrad(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin, int taps) {
    long total = (long)Cout * taps * Cin;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int co = (int)(i % Cout);
        long r = i / Cout;
        int t = (int)(r % taps);
        int ci = (int)(r / taps);
        wt[i] = w[((long)co * taps + (taps - 1 - t)) * Cin + ci];
    }
}
extern "C" int vqw_pack_dgrad_weights(const float* w_ohwi, float* wt, int Cout, int Cin, int ksize, void* stream) {
    VQW_CHECK(w_ohwi && wt && Cout > 0 && Cin > 0 && ksize >= 1 && ksize <= 7, "vqw_pack_dgrad_weights: bad arguments");
    long total = (long)Cout * ksize * ksize * Cin;
    k_pack_dgrad<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(w_ohwi, wt, Cout, Cin, ksize * ksize);
    VQW_LAUNCH_CHECK("vqw_pack_dgrad_weights");
    return VQW_OK;
}

// ---------------------------------------------------------------------------------------------
// generic wgrad: block (output element o = (co,t,ci), split s) reduces its pixel range; partials then summed.
__global__ void __launch_bounds__(256) k_conv_direct_wgrad(ConvIn in, const float* __restrict__ dy, float* __restrict__ part,
                                                           int N, int H, int W, int Cout, int ks, int dil, int splits) {
    __shared__ float s_red[4];
    const int Cin = in.C0 + in.C1;
    const int taps = ks * ks, half = ks >> 1;
    const int o = blockIdx.x, s = blockIdx.y;
    const int ci = o % Cin;
    const int t = (o / Cin) % taps;
    const int co = o / (Cin * taps);
    const int dyy = (t / ks - half) * dil, dxx = (t % ks - half) * dil;
    const long P = (long)N * H * W;
    const long per = (P + splits - 1) / splits;
    const long p0 = s * per, p1 = p0 + per < P ? p0 + per : P;
    float acc = 0.f;
    for (long p = p0 + threadIdx.x; p < p1; p += 256) {
        int x = (int)(p % W);
        long q = p / W;
        int yy = (int)(q % H);
        int n = (int)(q / H);
        int hy = yy + dyy, wx = x + dxx;
        if (hy < 0 || hy >= H || wx < 0 || wx >= W) continue;
        float v;
        if (ci < in.C0) {
            v = in.up0 ? in.src0[(((long)n * (H >> 1) + (hy >> 1)) * (W >> 1) + (wx >> 1)) * in.C0 + ci]
                       : in.src0[(((long)n * H + hy) * W + wx) * in.C0 + ci];
        } else {
            v = in.src1[(((long)n * H + hy) * W + wx) * in.C1 + (ci - in.C0)];
        }
        acc = fmaf(dy[p * Cout + co], v, acc);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[(long)s * gridDim.x + o] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// out[i] = sum_s part[s][i].  Workgroup = 64 columns x 16 row groups: every thread sums rows g, g+16, ... of its
// column (coalesced 256-B row segments), then a fixed-order LDS tree over the 16 groups -> deterministic.
__global__ void __launch_bounds__(1024) k_reduce_rows(const float* __restrict__ part, float* __restrict__ out, long n, int rows,
                                                      int acc) {
    __shared__ float sm[16][64];
    const int cx = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + cx;
    float a = 0.f;
    if (i < n) {
        int r = g;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
        for (; r + 112 < rows; r += 128) {          // many slabs (split-K of the small 1x1 layers): eight loads in flight
            a0 += part[(long)r * n + i];
            a1 += part[(long)(r + 16) * n + i];
            a2 += part[(long)(r + 32) * n + i];
            a3 += part[(long)(r + 48) * n + i];
            b0 += part[(long)(r + 64) * n + i];
            b1 += part[(long)(r + 80) * n + i];
            b2 += part[(long)(r + 96) * n + i];
            b3 += part[(long)(r + 112) * n + i];
        }
        a0 += b0; a1 += b1; a2 += b2; a3 += b3;
        for (; r + 48 < rows; r += 64) {
            a0 += part[(long)r * n + i];
            a1 += part[(long)(r + 16) * n + i];
            a2 += part[(long)(r + 32) * n + i];
            a3 += part[(long)(r + 48) * n + i];
        }
        for (; r < rows; r += 16) a0 += part[(long)r * n + i];
        a = (a0 + a1) + (a2 + a3);
    }
    sm[g][cx] = a;
    __syncthreads();
    if (g == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k][cx];
        out[i] = acc ? out[i] + t : t;
    }
}

// few rows (the wide layers have 8..32 slabs): one lane per float4 column walks the rows; same summation order per column
__global__ void __launch_bounds__(256) k_reduce_rows_few(const float4* __restrict__ part, float4* __restrict__ out, long n4, int rows,
                                                         int acc) {
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 a = part[i];
        for (int r = 1; r < rows; ++r) {
            float4 v = part[(long)r * n4 + i];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        if (acc) {
            float4 o = out[i];
            a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
        }
        out[i] = a;
    }
}

// ---------------------------------------------------------------------------------------------
// Deferred slab folds.  Every weight-gradient kernel leaves split-K slabs and ends in reduce_rows (two short launches per
// layer: dW and dbias, ~250 per training step, each a dependent launch behind its MFMA kernel on a weight-gradient lane).
// With deferral on, reduce_rows only RECORDS the job; vqw_fold_flush_host() then folds everything recorded since the last
// flush in ONE launch (k_fold_multi).  Jobs that target the same output (the two views of a step: overwrite, then
// accumulate) are merged into one record whose segments are summed in recording order, so the result is what the separate
// launches would have produced in that order - fixed, bit-reproducible.  The slab buffers must stay alive until the flush
// (the Python side keeps them).  Process-wide state behind a mutex: autograd runs backward nodes on its own device thread
// and the end-of-pass callback that flushes on the thread that called backward().
#include <mutex>
#include <vector>
namespace {
struct FoldRec {            // device-visible record, 64 bytes
    const float* p0; const float* p1; float* out; long n;
    int rows0, rows1, acc, vec;
    long blk0;               // first workgroup of this record in the launch
    long pad;
};
std::mutex g_fold_mu;
std::vector<FoldRec> g_fold_jobs;
bool g_fold_defer = false;
constexpr int FOLD_COLS = 256;       // floats per workgroup

// one workgroup = FOLD_COLS consecutive output elements of one record: lane = 4 columns, 4 row groups walk the slabs
// (rows g, g + 4, ...), fixed-order fold of the groups through LDS; segment 1 (the second view) after segment 0
__global__ void __launch_bounds__(256) k_fold_multi(const FoldRec* __restrict__ recs, int nrec) {
    __shared__ float4 sm[2][4][64];
    int lo = 0, hi = nrec - 1;
    while (lo < hi) {                          // last record whose first workgroup is <= blockIdx.x
        int mid = (lo + hi + 1) >> 1;
        if (recs[mid].blk0 <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const FoldRec r = recs[lo];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long c0 = ((long)blockIdx.x - r.blk0) * FOLD_COLS + lane * 4;
#pragma unroll
    for (int seg = 0; seg < 2; ++seg) {
        const float* p = seg ? r.p1 : r.p0;
        const int rows = seg ? r.rows1 : r.rows0;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (p && c0 < r.n) {
            if (r.vec) {
                int q = g;
                for (; q + 4 < rows; q += 8) {
                    float4 u = *(const float4*)(p + (long)q * r.n + c0), v = *(const float4*)(p + (long)(q + 4) * r.n + c0);
                    a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
                    b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
                }
                if (q < rows) {
                    float4 u = *(const float4*)(p + (long)q * r.n + c0);
                    a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
                }
            } else {
                for (int q = g; q < rows; q += 4) {
                    const float* s = p + (long)q * r.n + c0;
                    a.x += s[0];
                    if (c0 + 1 < r.n) a.y += s[1];
                    if (c0 + 2 < r.n) a.z += s[2];
                    if (c0 + 3 < r.n) a.w += s[3];
                }
            }
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        sm[seg][g][lane] = a;
    }
    __syncthreads();
    if (g == 0 && c0 < r.n) {
        float4 t[2];
#pragma unroll
        for (int seg = 0; seg < 2; ++seg) {
            float4 u0 = sm[seg][0][lane], u1 = sm[seg][1][lane], u2 = sm[seg][2][lane], u3 = sm[seg][3][lane];
            t[seg] = make_float4((u0.x + u1.x) + (u2.x + u3.x), (u0.y + u1.y) + (u2.y + u3.y), (u0.z + u1.z) + (u2.z + u3.z),
                                 (u0.w + u1.w) + (u2.w + u3.w));
        }
        float* o = r.out + c0;
        float4 v = t[0];
        if (r.acc) {
            if (r.vec) { float4 w = *(const float4*)o; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
            else {
                v.x += o[0];
                if (c0 + 1 < r.n) v.y += o[1];
                if (c0 + 2 < r.n) v.z += o[2];
                if (c0 + 3 < r.n) v.w += o[3];
            }
        }
        if (r.p1) { v.x += t[1].x; v.y += t[1].y; v.z += t[1].z; v.w += t[1].w; }
        if (r.vec) *(float4*)o = v;
        else {
            o[0] = v.x;
            if (c0 + 1 < r.n) o[1] = v.y;
            if (c0 + 2 < r.n) o[2] = v.z;
            if (c0 + 3 < r.n) o[3] = v.w;
        }
    }
}

int reduce_rows_now(const float* part, float* out, long n, int rows, hipStream_t st, int acc);
}  // namespace

extern "C" int vqw_fold_defer(int on) {
    std::lock_guard<std::mutex> lk(g_fold_mu);
    int old = g_fold_defer ? 1 : 0;
    g_fold_defer = on != 0;
    return old;
}
extern "C" int vqw_fold_pending(void) {
    std::lock_guard<std::mutex> lk(g_fold_mu);
    return (int)g_fold_jobs.size();
}
extern "C" size_t vqw_fold_table_bytes(void) {
    std::lock_guard<std::mutex> lk(g_fold_mu);
    return g_fold_jobs.size() * sizeof(FoldRec) + 64;
}
extern "C" int vqw_fold_discard(void) {
    std::lock_guard<std::mutex> lk(g_fold_mu);
    int n = (int)g_fold_jobs.size();
    g_fold_jobs.clear();
    return n;
}
extern "C" int vqw_fold_flush_host(void* table_host, void* table_dev, size_t table_bytes, void* stream) {
    std::lock_guard<std::mutex> lk(g_fold_mu);
    if (g_fold_jobs.empty()) return VQW_OK;
    const size_t need = g_fold_jobs.size() * sizeof(FoldRec);
    VQW_CHECK(table_host && table_dev && table_bytes >= need, "vqw_fold_flush_host: table buffers too small (%zu < %zu)", table_bytes, need);
    long blk = 0;
    for (auto& r : g_fold_jobs) {
        r.blk0 = blk;
        blk += (r.n + FOLD_COLS - 1) / FOLD_COLS;
    }
    memcpy(table_host, g_fold_jobs.data(), need);
    const int nrec = (int)g_fold_jobs.size();
    g_fold_jobs.clear();
    hipStream_t st = (hipStream_t)stream;
    if (hipMemcpyAsync(table_dev, table_host, need, hipMemcpyHostToDevice, st) != hipSuccess) {
        vqw_set_error("vqw_fold_flush_host: table upload failed");
        return VQW_ERR_HIP;
    }
    VQW_CHECK(blk < (1L << 31), "vqw_fold_flush_host: too many workgroups");
    k_fold_multi<<<(unsigned)blk, 256, 0, st>>>((const FoldRec*)table_dev, nrec);
    VQW_LAUNCH_CHECK("fold_multi");
    return VQW_OK;
}

int reduce_rows(const float* part, float* out, long n, int rows, hipStream_t st, int acc) {
    {
        std::lock_guard<std::mutex> lk(g_fold_mu);
        if (g_fold_defer) {
            const int vec = ((n & 3) == 0 && ((((uintptr_t)part | (uintptr_t)out) & 15) == 0)) ? 1 : 0;
            for (auto& r : g_fold_jobs) {
                if (r.out == out) {
                    // a second job for the same output (the other view of the step): summed after the first, in one record
                    if (r.n == n && r.p1 == nullptr && acc) {
                        r.p1 = part; r.rows1 = rows; r.vec = r.vec && vec;
                        return VQW_OK;
                    }
                    vqw_set_error("reduce_rows: more than two deferred folds into one output (flush first)");
                    return VQW_ERR_ARG;
                }
            }
            g_fold_jobs.push_back(FoldRec{part, nullptr, out, n, rows, 0, acc, vec, 0, 0});
            return VQW_OK;
        }
    }
    return reduce_rows_now(part, out, n, rows, st, acc);
}

namespace {
int reduce_rows_now(const float* part, float* out, long n, int rows, hipStream_t st, int acc) {
    if (rows <= 48 && (n & 3) == 0 && n >= 4096 && ((((uintptr_t)part | (uintptr_t)out) & 15) == 0)) {
        k_reduce_rows_few<<<stream_grid(n / 4, 256), 256, 0, st>>>((const float4*)part, (float4*)out, n / 4, rows, acc);
        VQW_LAUNCH_CHECK("reduce_rows_few");
        return VQW_OK;
    }
    k_reduce_rows<<<(unsigned)((n + 63) / 64), 1024, 0, st>>>(part, out, n, rows, acc);
    VQW_LAUNCH_CHECK("reduce_rows");
    return VQW_OK;
}
}  // namespace

int conv_direct_wgrad_splits(long nout, long P) {
    int s = ceil_div(4096, nout);
    long cap = P / 1024 > 1 ? P / 1024 : 1;
    if (s > cap) s = (int)cap;
    if (s > 256) s = 256;
    return s < 1 ? 1 : s;
}

int conv_direct_wgrad(const ConvIn& in, const float* dy, float* dw, float* ws, int N, int H, int W, int Cout, int ks, int dil,
                      hipStream_t st, int acc) {
    const int Cin = in.C0 + in.C1;
    long nout = (long)Cout * ks * ks * Cin;
    int splits = conv_direct_wgrad_splits(nout, (long)N * H * W);
    k_conv_direct_wgrad<<<dim3((unsigned)nout, splits), 256, 0, st>>>(in, dy, ws, N, H, W, Cout, ks, dil, splits);
    VQW_LAUNCH_CHECK("conv_direct_wgrad");
    return reduce_rows(ws, dw, nout, splits, st, acc);
}

// ---------------------------------------------------------------------------------------------
// dbias[co] = sum_p dy[p][co]; two-stage, deterministic
#define BG_ROWS 2048
// a workgroup sums a slice of pixels: threads = (channel lane, pixel lane), eight independent partial sums per thread keep
// eight loads in flight (one dependent load after the other made this launch take 91 us whatever the tensor size: 32
// workgroups walking 256 pixels each at P = 8192); lanes and partials are folded in a fixed order
__global__ void __launch_bounds__(256) k_bias_grad_partial(const float* __restrict__ dy, float* __restrict__ part, long P, int C) {
    __shared__ float sa[256];
    const int tcn = C < 256 ? C : 256;
    const int rows = 256 / tcn;
    const int t = threadIdx.x, tc = t % tcn, tr = t / tcn;
    const long per = (P + gridDim.x - 1) / gridDim.x;
    const long p0 = blockIdx.x * per, p1 = p0 + per < P ? p0 + per : P;
    for (int cb = 0; cb < C; cb += tcn) {
        int c = cb + tc;
        float a = 0.f;
        if (tr < rows && c < C) {
            float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            long p = p0 + tr;
            for (; p + 7L * rows < p1; p += 8L * rows) {
#pragma unroll
                for (int u = 0; u < 8; ++u) s[u] += dy[(p + (long)u * rows) * C + c];
            }
            for (; p < p1; p += rows) s[0] += dy[p * C + c];
            a = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
        }
        sa[t] = a;
        __syncthreads();
        if (tr == 0 && c < C) {
            for (int r = 1; r < rows; ++r) a += sa[r * tcn + tc];
            part[(long)blockIdx.x * C + c] = a;
        }
        __syncthreads();
    }
}
size_t bias_grad_ws_floats(int C) { return (size_t)BG_ROWS * C; }
int bias_grad(const float* dy, float* dbias, float* ws, long P, int C, hipStream_t st, int acc) {
    // ~64 pixels per pixel lane of a workgroup, at most BG_ROWS workgroups
    const int lanes = 256 / (C < 256 ? C : 256);
    int rows = (int)imin(BG_ROWS, imax(1, (int)(P / (64L * (lanes > 0 ? lanes : 1)))));
    k_bias_grad_partial<<<rows, 256, 0, st>>>(dy, ws, P, C);
    VQW_LAUNCH_CHECK("bias_grad");
    return reduce_rows(ws, dbias, C, rows, st, acc);
}

// ---------------------------------------------------------------------------------------------
template <int UP, int ACC>
__global__ void k_input_grad_gather(const float* __restrict__ g, int Ctot, int c_off, int C, float* __restrict__ dst, int N,
                                    int H, int W) {
    const int Ho = UP ? H >> 1 : H, Wo = UP ? W >> 1 : W;
    long total = (long)N * Ho * Wo * C;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int c = (int)(i % C);
        long p = i / C;
        int x = (int)(p % Wo);
        long q = p / Wo;
        int y = (int)(q % Ho);
        int n = (int)(q / Ho);
        float v;
        if (UP) {
            const float* b = g + (((long)n * H + 2 * y) * W + 2 * x) * Ctot + c_off + c;
            v = (b[0] + b[Ctot]) + (b[(long)W * Ctot] + b[(long)W * Ctot + Ctot]);
        } else {
            v = g[(((long)n * H + y) * W + x) * Ctot + c_off + c];
        }
        dst[i] = ACC ? dst[i] + v : v;
    }
}
extern "C" int vqw_input_grad_gather(const float* g_full, int Ctot, int c_off, int C, int up, float* dst, int accumulate,
                                     int N, int H, int W, void* stream) {
    VQW_CHECK(g_full && dst && N > 0 && H > 0 && W > 0 && C > 0 && c_off >= 0 && c_off + C <= Ctot,
              "vqw_input_grad_gather: bad arguments");
    VQW_CHECK(!up || ((H % 2 == 0) && (W % 2 == 0)), "vqw_input_grad_gather: up-sampled source needs even H, W");
    hipStream_t st = (hipStream_t)stream;
    long total = (long)N * (up ? H / 2 : H) * (up ? W / 2 : W) * C;
    int gr = stream_grid(total, 256);
    if (up && accumulate) k_input_grad_gather<1, 1><<<gr, 256, 0, st>>>(g_full, Ctot, c_off, C, dst, N, H, W);
    else if (up) k_input_grad_gather<1, 0><<<gr, 256, 0, st>>>(g_full, Ctot, c_off, C, dst, N, H, W);
    else if (accumulate) k_input_grad_gather<0, 1><<<gr, 256, 0, st>>>(g_full, Ctot, c_off, C, dst, N, H, W);
    else k_input_grad_gather<0, 0><<<gr, 256, 0, st>>>(g_full, Ctot, c_off, C, dst, N, H, W);
    VQW_LAUNCH_CHECK("vqw_input_grad_gather");
    return VQW_OK;
}
