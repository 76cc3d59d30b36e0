// Implicit-GEMM convolution on the CDNA4 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32), NHWC.
//
// forward / dgrad  (conv_mfma_fwd):  Y[p][co] = sum_{tap,ci} X[p + shift(tap)][ci] * W[co][tap][ci]
//     GEMM view: M = pixels (linear n*H*W index, so any H, W works), N = Cout, K = taps*Cin.
//     A tile  [BM pixels][KC channels]  gathered per (tap, channel chunk) with zero fill for the padding halo;
//             the loader does the nearest x2 up-sample and the channel concat by index arithmetic.
//     B tile  [BN couts][KC channels]   straight from the OHWI weights.
//     Both staged global -> registers -> LDS (double-buffered, one barrier per chunk; the global loads for chunk
//     i+1 are issued before the MFMAs of chunk i and written to LDS after them).  LDS rows are padded by 4 floats
//     so the ds_read_b128 fragment reads are bank-conflict free.  Each lane reads 4 consecutive k of its row and
//     feeds 4 MFMAs (k-order inside an 8-group is permuted identically for A and B).
//
// wgrad (conv_mfma_wgrad):  dW[co][tap][ci] = sum_p dY[p][co] * X[p + shift(tap)][ci]
//     GEMM view: M = Cout, N = Cin, K = pixels, one (tap, co tile, ci tile, pixel split) per workgroup;
//     tiles are stored exactly as they sit in memory ([pixel][channel]) and read with ds_read_b32 (32 lanes on
//     32 consecutive banks).  Split-K partial slabs [split][Cout][taps][Cin] are summed in a fixed order by
//     reduce_rows (deterministic; no float atomics).
#include "common.h"
#include "conv_common.h"

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// Bijective XCD-aware remap: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a contiguous run
// of tiles (neighbouring pixel tiles share halo rows and all N tiles of a pixel tile share the A operand in L2).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7;
    int xcd = bid & 7, idx = bid >> 3;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ float4 ld4_or_zero(const float* p, bool ok) {
    float4 z;
    z.x = z.y = z.z = z.w = 0.f;
    return ok ? *(const float4*)p : z;
}

// =============================================================================================
// forward / dgrad
// =============================================================================================
template <int BM, int BN, int WM, int WN, int KC>
__global__ void __launch_bounds__(256, 2)
k_conv_mfma_fwd(ConvIn in, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y, int N, int H,
                int W, int Cout, int ks, int dil, int ntn, int relu) {
    constexpr int LDK = KC + 4;           // padded row length (floats)
    constexpr int C4 = KC / 4;            // float4 per row
    constexpr int LA = BM * C4 / 256;     // A float4 loads per thread per chunk
    constexpr int LB = (BN * C4 + 255) / 256;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
    static_assert(BM * C4 % 256 == 0, "A tile must divide over 256 threads");

    __shared__ __attribute__((aligned(16))) float As[2][BM * LDK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDK];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int Cin = in.C0 + in.C1;
    const int taps = ks * ks, half = ks >> 1;
    const long P = (long)N * H * W;
    const int swz = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = swz % ntn, tile_m = swz / ntn;
    const long p_base = (long)tile_m * BM;
    const int co_base = tile_n * BN;
    const int Hs = H >> 1, Ws = W >> 1;

    // per-thread pixel coordinates of the A rows it loads (fixed for the whole K loop)
    int a_n[LA], a_h[LA], a_w[LA];
    bool a_ok[LA];
    const int a_c4 = tid % C4;
#pragma unroll
    for (int j = 0; j < LA; ++j) {
        int m = (tid + j * 256) / C4;
        long p = p_base + m;
        a_ok[j] = p < P;
        long pp = a_ok[j] ? p : 0;
        a_w[j] = (int)(pp % W);
        long q = pp / W;
        a_h[j] = (int)(q % H);
        a_n[j] = (int)(q / H);
    }
    const int cpt = (Cin + KC - 1) / KC;  // chunks per tap
    const int nchunks = taps * cpt;

    float4 ra[LA], rb[LB];
    auto load_chunk = [&](int it) {
        const int t = it / cpt, cc = (it - t * cpt) * KC;
        const int dyy = (t / ks - half) * dil, dxx = (t % ks - half) * dil;
        const int c = cc + a_c4 * 4;
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            int hy = a_h[j] + dyy, wx = a_w[j] + dxx;
            bool ok = a_ok[j] && hy >= 0 && hy < H && wx >= 0 && wx < W && c < Cin;
            const float* src;
            if (c < in.C0) {
                src = in.up0 ? in.src0 + (((long)a_n[j] * Hs + (hy >> 1)) * Ws + (wx >> 1)) * in.C0 + c
                             : in.src0 + (((long)a_n[j] * H + hy) * W + wx) * in.C0 + c;
            } else {
                src = in.src1 + (((long)a_n[j] * H + hy) * W + wx) * in.C1 + (c - in.C0);
            }
            ra[j] = ld4_or_zero(src, ok);
        }
#pragma unroll
        for (int j = 0; j < LB; ++j) {
            int f = tid + j * 256;
            int row = f / C4, c4 = f % C4;
            int co = co_base + row, cb = cc + c4 * 4;
            bool ok = (BN * C4 % 256 == 0 || f < BN * C4) && co < Cout && cb < Cin;
            rb[j] = ld4_or_zero(w + ((long)co * taps + t) * Cin + cb, ok);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            int f = tid + j * 256;
            *(float4*)&As[buf][(f / C4) * LDK + (f % C4) * 4] = ra[j];
        }
#pragma unroll
        for (int j = 0; j < LB; ++j) {
            int f = tid + j * 256;
            if (BN * C4 % 256 == 0 || f < BN * C4) *(float4*)&Bs[buf][(f / C4) * LDK + (f % C4) * 4] = rb[j];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm0 = (wv / WAVES_N) * WM, wn0 = (wv % WAVES_N) * WN;
    const int lrow = lane & 31, lk = (lane >> 5) * 4;

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int it = 0; it < nchunks; ++it) {
        const int cur = it & 1;
        if (it + 1 < nchunks) load_chunk(it + 1);
#pragma unroll
        for (int kg = 0; kg < KC / 8; ++kg) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(const float4*)&As[cur][(wm0 + i * 32 + lrow) * LDK + kg * 8 + lk];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *(const float4*)&Bs[cur][(wn0 + j * 32 + lrow) * LDK + kg * 8 + lk];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = MFMA32(a[i].x, b[j].x, acc[i][j]);
                    acc[i][j] = MFMA32(a[i].y, b[j].y, acc[i][j]);
                    acc[i][j] = MFMA32(a[i].z, b[j].z, acc[i][j]);
                    acc[i][j] = MFMA32(a[i].w, b[j].w, acc[i][j]);
                }
        }
        if (it + 1 < nchunks) store_chunk(cur ^ 1);
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31 (cout), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int co = co_base + wn0 + j * 32 + (lane & 31);
        const bool cok = co < Cout;
        const float bv = (bias && cok) ? bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                long p = p_base + wm0 + i * 32 + row;
                if (cok && p < P) {
                    float v = acc[i][j][r] + bv;
                    y[p * Cout + co] = relu ? fmaxf(v, 0.f) : v;
                }
            }
        }
    }
}

bool conv_mfma_fwd_ok(const ConvIn& in, int Cout, int ks) {
    (void)ks;
    int Cin = in.C0 + in.C1;
    // float4 channel loads: every source a multiple of 4 channels; tiny Cout (1-channel head) is left to the
    // generic kernel (a 32-wide MFMA tile would be >90% padding).
    return (in.C0 % 4 == 0) && (in.C1 % 4 == 0) && Cin >= 8 && Cout >= 8;
}

template <int BM, int BN, int WM, int WN>
static int launch_fwd(const ConvIn& in, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int ks,
                      int dil, int relu, hipStream_t st) {
    long P = (long)N * H * W;
    int ntm = ceil_div(P, BM), ntn = ceil_div(Cout, BN);
    k_conv_mfma_fwd<BM, BN, WM, WN, 16><<<ntm * ntn, 256, 0, st>>>(in, w, bias, y, N, H, W, Cout, ks, dil, ntn, relu);
    VQW_LAUNCH_CHECK("conv_mfma_fwd");
    return VQW_OK;
}

int conv_mfma_fwd(const ConvIn& in, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int ks, int dil,
                  int relu, hipStream_t st) {
    if (Cout > 64) return launch_fwd<128, 128, 64, 64>(in, w, bias, y, N, H, W, Cout, ks, dil, relu, st);
    if (Cout > 32) return launch_fwd<128, 64, 64, 32>(in, w, bias, y, N, H, W, Cout, ks, dil, relu, st);
    return launch_fwd<256, 32, 64, 32>(in, w, bias, y, N, H, W, Cout, ks, dil, relu, st);
}

// =============================================================================================
// wgrad
// =============================================================================================
template <int BM, int BN, int WM, int WN, int KP>
__global__ void __launch_bounds__(64 * (BM / WM) * (BN / WN))
k_conv_mfma_wgrad(ConvIn in, const float* __restrict__ dy, float* __restrict__ part, int N, int H, int W, int Cout, int ks,
                  int dil, int ntm, int ntn, long per_split) {
    constexpr int NW = (BM / WM) * (BN / WN);
    constexpr int NT = 64 * NW;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int LD = KP * (BM / 4) / NT;  // dy float4 loads per thread per chunk
    constexpr int LX = KP * (BN / 4) / NT;
    static_assert(KP * (BM / 4) % NT == 0 && KP * (BN / 4) % NT == 0, "tiles must divide over the workgroup");

    __shared__ __attribute__((aligned(16))) float Ds[2][KP * BM];
    __shared__ __attribute__((aligned(16))) float Xs[2][KP * BN];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int Cin = in.C0 + in.C1;
    const int taps = ks * ks, half = ks >> 1;
    const long P = (long)N * H * W;
    const int Hs = H >> 1, Ws = W >> 1;

    int b = blockIdx.x;
    const int tile_n = b % ntn; b /= ntn;
    const int tile_m = b % ntm; b /= ntm;
    const int t = b % taps;
    const int split = b / taps;
    const int co_base = tile_m * BM, ci_base = tile_n * BN;
    const int dyy = (t / ks - half) * dil, dxx = (t % ks - half) * dil;
    const long p_begin = (long)split * per_split;
    long p_end = p_begin + per_split;
    if (p_end > P) p_end = P;
    const int nchunks = p_begin < p_end ? (int)((p_end - p_begin + KP - 1) / KP) : 0;

    float4 rd[LD], rx[LX];
    auto load_chunk = [&](int it) {
        const long p0 = p_begin + (long)it * KP;
#pragma unroll
        for (int j = 0; j < LD; ++j) {
            int f = tid + j * NT;
            int k = f / (BM / 4), c4 = f % (BM / 4);
            long p = p0 + k;
            int co = co_base + c4 * 4;
            bool ok = p < p_end && co < Cout;
            rd[j] = ld4_or_zero(dy + p * Cout + co, ok);   // Cout % 4 == 0 (conv_mfma_wgrad_ok)
        }
#pragma unroll
        for (int j = 0; j < LX; ++j) {
            int f = tid + j * NT;
            int k = f / (BN / 4), c4 = f % (BN / 4);
            long p = p0 + k;
            bool ok = p < p_end;
            long pp = ok ? p : 0;
            int x = (int)(pp % W);
            long q = pp / W;
            int yy = (int)(q % H);
            int n = (int)(q / H);
            int hy = yy + dyy, wx = x + dxx;
            int c = ci_base + c4 * 4;
            ok = ok && hy >= 0 && hy < H && wx >= 0 && wx < W && c < Cin;
            const float* src;
            if (c < in.C0) {
                src = in.up0 ? in.src0 + (((long)n * Hs + (hy >> 1)) * Ws + (wx >> 1)) * in.C0 + c
                             : in.src0 + (((long)n * H + hy) * W + wx) * in.C0 + c;
            } else {
                src = in.src1 + (((long)n * H + hy) * W + wx) * in.C1 + (c - in.C0);
            }
            rx[j] = ld4_or_zero(src, ok);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < LD; ++j) {
            int f = tid + j * NT;
            *(float4*)&Ds[buf][f * 4] = rd[j];
        }
#pragma unroll
        for (int j = 0; j < LX; ++j) {
            int f = tid + j * NT;
            *(float4*)&Xs[buf][f * 4] = rx[j];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm0 = (wv / WAVES_N) * WM, wn0 = (wv % WAVES_N) * WN;
    const int lcol = lane & 31, lk = lane >> 5;

    if (nchunks > 0) {
        load_chunk(0);
        store_chunk(0);
    }
    __syncthreads();
    for (int it = 0; it < nchunks; ++it) {
        const int cur = it & 1;
        if (it + 1 < nchunks) load_chunk(it + 1);
#pragma unroll 8
        for (int k = 0; k < KP; k += 2) {
            float a[TM], bb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = Ds[cur][(k + lk) * BM + wm0 + i * 32 + lcol];
#pragma unroll
            for (int j = 0; j < TN; ++j) bb[j] = Xs[cur][(k + lk) * BN + wn0 + j * 32 + lcol];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = MFMA32(a[i], bb[j], acc[i][j]);
        }
        if (it + 1 < nchunks) store_chunk(cur ^ 1);
        __syncthreads();
    }

    // partial slab [split][co][tap][ci]; D layout: col = lane&31 -> ci, rows -> co
    float* o = part + (long)split * Cout * taps * Cin;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ci = ci_base + wn0 + j * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                int co = co_base + wm0 + i * 32 + row;
                if (co < Cout && ci < Cin) o[((long)co * taps + t) * Cin + ci] = acc[i][j][r];
            }
        }
    }
}

bool conv_mfma_wgrad_ok(const ConvIn& in, int Cout, int ks) {
    (void)ks;
    int Cin = in.C0 + in.C1;
    return (in.C0 % 4 == 0) && (in.C1 % 4 == 0) && Cin >= 8 && Cout >= 8 && (Cout % 4 == 0);
}

static inline int wg_tile(int c) { return c > 64 ? 128 : (c > 32 ? 64 : 32); }

static inline int wgrad_splits(int Cin, int Cout, int ks, long P) {
    int bm = wg_tile(Cout), bn = wg_tile(Cin);
    long tiles = (long)ceil_div(Cout, bm) * ceil_div(Cin, bn) * ks * ks;
    int s = ceil_div(1024, tiles);                      // ~4 workgroups per CU
    long cap = P / 256 > 1 ? P / 256 : 1;               // at least 256 pixels per split
    if (s > cap) s = (int)cap;
    if (s > 512) s = 512;
    return s < 1 ? 1 : s;
}

size_t conv_mfma_wgrad_ws_floats(int Cin, int Cout, int ks, long P) {
    return (size_t)wgrad_splits(Cin, Cout, ks, P) * Cout * ks * ks * Cin;
}

template <int BM, int BN, int KP>
static int launch_wgrad(const ConvIn& in, const float* dy, float* dw, float* ws, int N, int H, int W, int Cout, int ks, int dil,
                        hipStream_t st) {
    constexpr int WM = BM > 64 ? 64 : BM, WN = BN > 64 ? 64 : BN;
    constexpr int NT = 64 * (BM / WM) * (BN / WN);
    const int Cin = in.C0 + in.C1;
    const long P = (long)N * H * W;
    const int splits = wgrad_splits(Cin, Cout, ks, P);
    long per = (P + splits - 1) / splits;
    per = ((per + KP - 1) / KP) * KP;
    const int ntm = ceil_div(Cout, BM), ntn = ceil_div(Cin, BN);
    const long nout = (long)Cout * ks * ks * Cin;
    float* part = splits > 1 ? ws : dw;
    k_conv_mfma_wgrad<BM, BN, WM, WN, KP><<<ntm * ntn * ks * ks * splits, NT, 0, st>>>(in, dy, part, N, H, W, Cout, ks, dil,
                                                                                         ntm, ntn, per);
    VQW_LAUNCH_CHECK("conv_mfma_wgrad");
    if (splits > 1) return reduce_rows(ws, dw, nout, splits, st);
    return VQW_OK;
}

int conv_mfma_wgrad(const ConvIn& in, const float* dy, float* dw, float* ws, int N, int H, int W, int Cout, int ks, int dil,
                    hipStream_t st) {
    const int bm = wg_tile(Cout), bn = wg_tile(in.C0 + in.C1);
#define WG_CASE(M_, N_, K_) \
    if (bm == M_ && bn == N_) return launch_wgrad<M_, N_, K_>(in, dy, dw, ws, N, H, W, Cout, ks, dil, st)
    WG_CASE(32, 32, 32);
    WG_CASE(32, 64, 32);
    WG_CASE(64, 32, 32);
    WG_CASE(64, 64, 32);
    WG_CASE(32, 128, 32);
    WG_CASE(128, 32, 32);
    WG_CASE(64, 128, 32);
    WG_CASE(128, 64, 32);
    WG_CASE(128, 128, 32);
#undef WG_CASE
    vqw_set_error("conv_mfma_wgrad: no tile configuration");
    return VQW_ERR_ARG;
}
