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
#include "mfma_util.h"

__device__ __forceinline__ float4 ld4_or_zero(const float* p, bool ok) {
    float4 z;
    z.x = z.y = z.z = z.w = 0.f;
    return ok ? *(const float4*)p : z;
}

// Tap geometry of one launch.  Taps are arbitrary (dy, dx) pixel offsets, so the same kernel serves the plain k x k
// convolution, and the two "collapsed" forms of a 3x3 convolution over a nearest x2 up-sampled input:
//   src_mode 0: source pixel = (y + dy, x + dx) on the M grid            (plain; up0 in ConvIn = virtual up-sampling)
//   src_mode 2: source pixel = (2y + dy, 2x + dx) on a 2H x 2W tensor    (collapsed dgrad: M grid is the low-res map)
//   out_mode 0: output pixel = M-grid pixel
//   out_mode 1: output pixel = (2y + py, 2x + px) of a 2H x 2W tensor    (collapsed forward, one launch per parity)
struct ConvGeom {
    int ntaps;
    int src_mode, out_mode, py, px;
    signed char dy[16], dx[16];
};

// =============================================================================================
// forward / dgrad
// =============================================================================================
// Index arithmetic is 32-bit (the host checks numel < 2^31) and hoisted: pixel coordinates once per thread,
// bounds / base offsets once per TAP, only an add per channel chunk.
template <int BM, int BN, int WM, int WN, int KC, bool DEEP>
__global__ void __launch_bounds__(256, 2)
k_conv_mfma_fwd(ConvIn in, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y, int N, int H,
                int W, int Cout, ConvGeom geo, int ntn, int relu, unsigned nb0, unsigned nb1, unsigned nbw, float* __restrict__ stats) {
    constexpr int LDK = KC + 4;           // padded row length (floats)
    constexpr int C4 = KC / 4;            // float4 per row
    constexpr int LA = BM * C4 / 256;     // A float4 loads per thread per chunk
    constexpr int LB = (BN * C4 + 255) / 256;
    constexpr int RSTEP = 256 / C4;       // A rows covered by one pass of the workgroup
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
    static_assert(BM * C4 % 256 == 0, "A tile must divide over 256 threads");

    __shared__ __attribute__((aligned(16))) float As[2][BM * LDK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDK];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int C0 = in.C0, C1 = in.C1, Cin = C0 + C1;
    const int taps = geo.ntaps;
    const unsigned P = (unsigned)N * H * W;
    const int swz = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = swz % ntn, tile_m = swz / ntn;
    const unsigned p_base = (unsigned)tile_m * BM;
    const int co_base = tile_n * BN;
    const int Hs = H >> 1, Ws = W >> 1;
    const int up0 = in.up0;
    // collapsed forward: the four output parities are the y dimension of the grid, each with its own weight block
    const int par_y = geo.out_mode == 1 ? (int)(blockIdx.y >> 1) : 0, par_x = geo.out_mode == 1 ? (int)(blockIdx.y & 1) : 0;
    if (geo.out_mode == 1) w += (size_t)blockIdx.y * Cout * taps * Cin;
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(in.src0, nb0), rs1 = make_rsrc(in.src1, nb1), rsw = make_rsrc(w, nbw);

    // per-thread pixel coordinates of the A rows it loads (fixed for the whole K loop)
    const int a_c = (tid % C4) * 4;       // channel offset inside a chunk
    const int a_row = tid / C4;
    int a_h[LA], a_w[LA];
    unsigned a_pix[LA], a_upb[LA];
    bool a_ok[LA];
#pragma unroll
    for (int j = 0; j < LA; ++j) {
        unsigned p = p_base + a_row + j * RSTEP;
        a_ok[j] = p < P;
        unsigned pp = a_ok[j] ? p : 0u;
        unsigned q = pp / (unsigned)W;
        a_w[j] = (int)(pp - q * W);
        unsigned n = q / (unsigned)H;
        a_h[j] = (int)(q - n * H);
        a_pix[j] = pp;
        // per-image base of the alternative source grids: the virtual up-sampled source (H/2 x W/2) or, for
        // src_mode 2, the 2H x 2W source
        a_upb[j] = geo.src_mode == 2 ? n * (unsigned)(4 * H * W) : n * (unsigned)(Hs * Ws);
    }
    // weight rows this thread loads (byte offsets; an invalid row sits at the out-of-range offset)
    unsigned b_off[LB];
#pragma unroll
    for (int j = 0; j < LB; ++j) {
        int f = tid + j * 256;
        int co = co_base + f / C4;
        bool ok = (BN * C4 % 256 == 0 || f < BN * C4) && co < Cout;
        b_off[j] = ok ? ((unsigned)co * taps * Cin + (f % C4) * 4) * 4u : nbw;
    }

    // loader state: current tap / channel chunk, per-tap byte offsets (out-of-range offset = zero padding)
    int l_t = 0, l_cc = 0;
    unsigned t_off0[LA], t_off1[LA];
    auto setup_tap = [&]() {
        const int dyy = geo.out_mode == 1 ? par_y - 1 + (l_t >> 1) : geo.dy[l_t];
        const int dxx = geo.out_mode == 1 ? par_x - 1 + (l_t & 1) : geo.dx[l_t];
        if (geo.src_mode == 2) {
#pragma unroll
            for (int j = 0; j < LA; ++j) {
                int hy = 2 * a_h[j] + dyy, wx = 2 * a_w[j] + dxx;
                bool ok = a_ok[j] && (unsigned)hy < (unsigned)(2 * H) && (unsigned)wx < (unsigned)(2 * W);
                unsigned e0 = (a_upb[j] + (unsigned)(hy * 2 * W + wx)) * (unsigned)C0;
                t_off0[j] = ok ? (e0 + a_c) * 4u : nb0;
                t_off1[j] = nb1;
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            int hy = a_h[j] + dyy, wx = a_w[j] + dxx;
            bool ok = a_ok[j] && (unsigned)hy < (unsigned)H && (unsigned)wx < (unsigned)W;
            unsigned pix = a_pix[j] + (unsigned)(dyy * W + dxx);
            unsigned e0 = up0 ? (a_upb[j] + (unsigned)((hy >> 1) * Ws + (wx >> 1))) * (unsigned)C0 : pix * (unsigned)C0;
            t_off0[j] = ok ? (e0 + a_c) * 4u : nb0;
            t_off1[j] = ok ? (pix * (unsigned)C1 + a_c) * 4u : nb1;
        }
    };
    // Two register sets: a chunk's global loads are issued two iterations before they are written to LDS, so HBM/L2
    // latency is covered by two chunk-times of MFMA work (the LDS itself stays double-buffered).
    float4 ra0[LA], rb0[LB], ra1[LA], rb1[LB];
    auto load_chunk = [&](float4 (&ra)[LA], float4 (&rb)[LB]) {   // loads chunk (l_t, l_cc), then advances the loader state
        // a chunk never straddles the two sources (C0 % KC == 0 is required when C1 > 0), so the choice is uniform;
        // channels beyond Cin (ragged last chunk) fall off the end of the pixel row: masked by the offset below
        const bool tail = (l_cc + a_c) >= Cin;
        if (l_cc < C0) {
            const unsigned cb = (unsigned)l_cc * 4u;
#pragma unroll
            for (int j = 0; j < LA; ++j) ra[j] = buf_ld4(rs0, (tail || t_off0[j] == nb0) ? nb0 : t_off0[j] + cb);
        } else {
            const unsigned cb = (unsigned)(l_cc - C0) * 4u;
#pragma unroll
            for (int j = 0; j < LA; ++j) ra[j] = buf_ld4(rs1, (tail || t_off1[j] == nb1) ? nb1 : t_off1[j] + cb);
        }
        const unsigned wofs = (unsigned)(l_t * Cin + l_cc) * 4u;
#pragma unroll
        for (int j = 0; j < LB; ++j) {
            int cbw = l_cc + ((tid + j * 256) % C4) * 4;
            rb[j] = buf_ld4(rsw, (cbw >= Cin || b_off[j] == nbw) ? nbw : b_off[j] + wofs);
        }
        l_cc += KC;
        if (l_cc >= Cin) {
            l_cc = 0;
            ++l_t;
            if (l_t < taps) setup_tap();
        }
    };
    auto store_chunk = [&](int buf, const float4 (&ra)[LA], const float4 (&rb)[LB]) {
#pragma unroll
        for (int j = 0; j < LA; ++j) *(float4*)&As[buf][(a_row + j * RSTEP) * LDK + a_c] = ra[j];
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
    const int nchunks = taps * ((Cin + KC - 1) / KC);

    auto compute = [&](int cur) {
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
    };

    setup_tap();
    load_chunk(ra0, rb0);
    store_chunk(0, ra0, rb0);
    if constexpr (DEEP) {
        // prologue: chunk 0 -> LDS buffer 0; chunks 1 and 2 in flight in the two register sets
        if (nchunks > 1) load_chunk(ra0, rb0);
        if (nchunks > 2) load_chunk(ra1, rb1);
        __syncthreads();
        // steady state, unrolled by two so the register sets are named statically.  At the top of iteration `it`:
        //   LDS[it&1] = chunk it, set A = chunk it+1 (issued two iterations ago), set B = chunk it+2.
        for (int it = 0; it < nchunks; it += 2) {
            if (it + 1 < nchunks) store_chunk(1, ra0, rb0);
            if (it + 3 < nchunks) load_chunk(ra0, rb0);
            compute(0);
            __syncthreads();
            if (it + 1 >= nchunks) break;
            if (it + 2 < nchunks) store_chunk(0, ra1, rb1);
            if (it + 4 < nchunks) load_chunk(ra1, rb1);
            compute(1);
            __syncthreads();
        }
    } else {
        // short K loops (few chunks per tile): one chunk of look-ahead, less prologue and fewer registers
        __syncthreads();
        for (int it = 0; it < nchunks; ++it) {
            const int cur = it & 1;
            if (it + 1 < nchunks) load_chunk(ra0, rb0);
            compute(cur);
            if (it + 1 < nchunks) store_chunk(cur ^ 1, ra0, rb0);
            __syncthreads();
        }
    }

    // Optional: statistics of the output for the InstanceNorm that follows (see k_conv_halo): per-tile (sum, M2
    // about the tile mean) per cout, the wave rows merged through LDS in a fixed order (mfma_util.h: stat_merge).  The host only asks for it when H*W is a
    // multiple of BM (a tile never straddles two images).
    if (stats) {           // uniform
        constexpr int WAVES_M = BM / WM;
        __syncthreads();                      // As is free now
        float* R = &As[0][0];                 // [WAVES_M][BN][2]
        const int wave_m = wv / WAVES_N;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int co = co_base + wn0 + j * 32 + (lane & 31);
            const float bv = (bias && co < Cout) ? bias[co] : 0.f;
            float vals[TM * 16];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) vals[i * 16 + r] = acc[i][j][r] + bv;
            float s1, s2;                     // (sum, M2) of this lane's TM*16 rows, then of the wave's WM rows
            lane_stats<TM * 16>(vals, s1, s2);
            stat_merge_eq(s1, s2, __shfl_xor(s1, 32, 64), __shfl_xor(s2, 32, 64), 1.f / (2 * TM * 16));
            if (lane < 32) { R[(wave_m * BN + wn0 + j * 32 + lane) * 2] = s1; R[(wave_m * BN + wn0 + j * 32 + lane) * 2 + 1] = s2; }
        }
        __syncthreads();
        if (tid < BN && co_base + tid < Cout) {
            float s1 = R[tid * 2], s2 = R[tid * 2 + 1];
#pragma unroll
            for (int m = 1; m < WAVES_M; ++m) stat_merge(s1, s2, (float)(m * WM), R[(m * BN + tid) * 2], R[(m * BN + tid) * 2 + 1], (float)WM);
            const int tpi = (H * W) / BM;                                   // tiles per image of the M grid
            const int npar = geo.out_mode == 1 ? 4 : 1;
            const int n = tile_m / tpi, t = tile_m - n * tpi;
            const int part = (geo.out_mode == 1 ? (int)blockIdx.y * tpi : 0) + t;
            float* o = stats + (((size_t)n * (tpi * npar) + part) * Cout + co_base + tid) * 2;
            o[0] = s1;
            o[1] = s2;
        }
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
                unsigned p = p_base + wm0 + i * 32 + row;
                if (cok && p < P) {
                    float v = acc[i][j][r] + bv;
                    size_t op = p;
                    if (geo.out_mode == 1) {      // scatter to parity (py, px) of the 2H x 2W output
                        unsigned q = p / (unsigned)W, xw = p - q * W;
                        unsigned n = q / (unsigned)H, yh = q - n * H;
                        op = ((size_t)n * 2 * H + 2 * yh + par_y) * (2 * W) + 2 * xw + par_x;
                    }
                    // relu: 0 plain, 1 ReLU, 2 accumulate (y += result: a later member of a gradient group, conv.hip)
                    y[op * Cout + co] = relu == 1 ? fmaxf(v, 0.f) : (relu == 2 ? y[op * Cout + co] + v : v);
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
    return (in.C0 % 4 == 0) && (in.C1 % 4 == 0) && Cin >= 8 && Cout >= 8 && (in.C1 == 0 || in.C0 % 32 == 0);
}

// buffer descriptors address at most 4 GiB per tensor (32-bit byte offsets; the top 32 bytes are kept as the
// guaranteed-out-of-range offset)
static inline bool fits_u32(long P, int Cin, int Cout) {
    long c = Cin > Cout ? Cin : Cout;
    return P * c * 4 <= 0xFFFFFFE0L;
}

static ConvGeom plain_geom(int ks, int dil, int tap0 = -1) {
    if (tap0 < 0) tap0 = ks / 2;
    ConvGeom g{};
    g.ntaps = ks * ks;
    for (int t = 0; t < ks * ks; ++t) {
        g.dy[t] = (signed char)((t / ks - tap0) * dil);
        g.dx[t] = (signed char)((t % ks - tap0) * dil);
    }
    return g;
}

template <int BM, int BN, int WM, int WN, int KC, bool DEEP>
static int launch_fwd(const ConvIn& in, const float* w, const float* bias, float* y, int N, int H, int W, int Cout,
                      const ConvGeom& geo, int relu, hipStream_t st, float* stats = nullptr) {
    if (stats && ((long)H * W) % BM != 0) { vqw_set_error("conv_mfma_fwd: statistics need H*W to be a multiple of the pixel tile"); return VQW_ERR_ARG; }
    long P = (long)N * H * W;
    int ntm = ceil_div(P, BM), ntn = ceil_div(Cout, BN);
    const long src_px = geo.src_mode == 2 ? 4 * P : (in.up0 ? P / 4 : P);
    const unsigned nb0 = (unsigned)(src_px * in.C0 * 4), nb1 = (unsigned)(P * in.C1 * 4);
    const unsigned nbw = (unsigned)((long)Cout * geo.ntaps * (in.C0 + in.C1) * 4);
    k_conv_mfma_fwd<BM, BN, WM, WN, KC, DEEP><<<dim3(ntm * ntn, geo.out_mode == 1 ? 4 : 1), 256, 0, st>>>(
        in, w, bias, y, N, H, W, Cout, geo, ntn, relu, nb0, nb1, nbw, stats);
    VQW_LAUNCH_CHECK("conv_mfma_fwd");
    return VQW_OK;
}

// pixel-tile height (BM) of the variant dispatch_fwd picks
static int dispatch_bm(const ConvIn& in, int Cout) {
    const int Cin = in.C0 + in.C1;
    auto cost = [&](int bn, double eff) { return (double)ceil_div(Cout, bn) * bn / eff; };
    const double c128 = cost(128, 1.0), c64 = cost(64, 0.82), c32 = cost(32, 0.70);
    if ((c128 <= c64 && c128 <= c32) || c64 <= c32 || Cin % 32 == 0) return 128;
    return 256;
}
static int dispatch_fwd(const ConvIn& in, const float* w, const float* bias, float* y, int N, int H, int W, int Cout,
                        const ConvGeom& geo, int relu, hipStream_t st, float* stats = nullptr) {
    const int Cin = in.C0 + in.C1;
    const bool k32 = (Cin % 32 == 0);
    // N tile: the one that pads Cout least, weighted by how well each tile runs (wider tiles reuse A fragments more)
    auto cost = [&](int bn, double eff) { return (double)ceil_div(Cout, bn) * bn / eff; };
    const double c128 = cost(128, 1.0), c64 = cost(64, 0.82), c32 = cost(32, 0.70);
    const bool deep = geo.ntaps * ceil_div(Cin, k32 ? 32 : 16) >= 12;      // enough chunks to amortise the deeper prologue
    // small pixel counts (the 16 x 16 level: 8192 pixels per view): 128 x 128 tiles give at most one workgroup per CU,
    // or leave half the CUs idle (256 couts: 128 workgroups); the 64-wide tile doubles the grid and two workgroups
    // share a CU
    const long wg128 = (long)ceil_div((long)N * H * W, 128) * ceil_div(Cout, 128);
    const bool small_grid = wg128 <= 256 && Cout % 64 == 0 && k32;
    if (c128 <= c64 && c128 <= c32 && !small_grid) {
        if (k32) return launch_fwd<128, 128, 64, 64, 32, true>(in, w, bias, y, N, H, W, Cout, geo, relu, st, stats);
        return launch_fwd<128, 128, 64, 64, 16, true>(in, w, bias, y, N, H, W, Cout, geo, relu, st, stats);
    }
    if (c64 <= c32 || small_grid) {
        if (k32 && deep) return launch_fwd<128, 64, 64, 32, 32, true>(in, w, bias, y, N, H, W, Cout, geo, relu, st, stats);
        if (k32) return launch_fwd<128, 64, 64, 32, 32, false>(in, w, bias, y, N, H, W, Cout, geo, relu, st, stats);
        return launch_fwd<128, 64, 64, 32, 16, false>(in, w, bias, y, N, H, W, Cout, geo, relu, st, stats);
    }
    if (k32) return launch_fwd<128, 32, 32, 32, 32, false>(in, w, bias, y, N, H, W, Cout, geo, relu, st, stats);
    return launch_fwd<256, 32, 64, 32, 16, false>(in, w, bias, y, N, H, W, Cout, geo, relu, st, stats);
}

int conv_mfma_fwd(const ConvIn& in, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int ks, int dil,
                  int relu, hipStream_t st, float* stats) {
    const int Cin = in.C0 + in.C1;
    if (!fits_u32((long)N * H * W, Cin, Cout) || dil > 127) {
        if (stats) { vqw_set_error("conv_mfma_fwd: statistics not available on the generic path"); return VQW_ERR_ARG; }
        return conv_direct_fwd(in, w, bias, y, N, H, W, Cout, ks, dil, relu, st);
    }
    return dispatch_fwd(in, w, bias, y, N, H, W, Cout, plain_geom(ks, dil), relu, st, stats);
}
// statistics partials per plane of the implicit-GEMM forward (0: not available for the shape)
int conv_mfma_stat_tiles(const ConvIn& in, int N, int H, int W, int Cout, int dil) {
    if (!fits_u32((long)N * H * W, in.C0 + in.C1, Cout) || dil > 127) return 0;
    const int bm = dispatch_bm(in, Cout);
    return ((long)H * W) % bm == 0 ? (int)((long)H * W / bm) : 0;
}
int conv_up2_stat_tiles(int Cin, int Cout, int N, int h, int w) {
    if (conv_wino_up_fwd_ok(Cin, Cout, N, h, w) && conv_wino_up_stat_tiles(h, w) > 0) return conv_wino_up_stat_tiles(h, w);
    if (conv_halo_up2_ok(Cin, Cout, N, h, w) && conv_halo_up2_stat_tiles(h, w) > 0) return conv_halo_up2_stat_tiles(h, w);
    ConvIn in{nullptr, nullptr, Cin, 0, 0};
    const int bm = dispatch_bm(in, Cout);
    return ((long)h * w) % bm == 0 ? 4 * (int)((long)h * w / bm) : 0;
}

// ---------------------------------------------------------------------------------------------
// 3x3 convolution over a nearest x2 up-sampled input, collapsed onto the low-resolution grid (4/9 of the FLOPs):
//   forward  y[2y+py, 2x+px] = sum_{a,b in {0,1}} Wc[py][px][a][b] * x[y + py-1+a, x + px-1+b]          (4 launches)
//   dgrad    dx[y, x]        = sum_{r,s in {-1..2}} Wd[r][s] * dY[2y + r, 2x + s]                        (1 launch)
// Wc / Wd are sums of the 3x3 taps that land on the same low-res pixel (k_collapse_up_weights).
__global__ void k_collapse_up_weights(const float* __restrict__ w, float* __restrict__ wc, float* __restrict__ wd, int Cout, int Cin) {
    long nfwd = 4L * Cout * 4 * Cin, nbwd = (long)Cin * 16 * Cout;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nfwd + nbwd; i += stride) collapse_up_element(w, wc, wd, i, Cout, Cin);
}

bool conv_up2_ok(int Cin, int Cout, long Plow) {
    return (Cin % 4 == 0) && Cin >= 8 && Cout >= 8 && fits_u32(4 * Plow, Cin, Cout);
}
static inline bool up2_has_wino(int Cin, int Cout) { return Cin % 8 == 0 && Cout % 8 == 0; }
size_t conv_up2_ws_floats(int Cin, int Cout) { return (size_t)32 * Cout * Cin + (up2_has_wino(Cin, Cout) ? conv_wino_up_ws_floats(Cin, Cout) : 0); }

// ws: [4][Cout][4][Cin] forward weights, [Cin][16][Cout] dgrad weights, then the nine-product Winograd weights (conv_wino_up.hip)
int conv_up2_prepare(const float* w, float* ws, int Cin, int Cout, hipStream_t st) {
    long n = 32L * Cout * Cin;
    // (with the nine-product Winograd weights behind them: ONE launch builds the collapsed and the transformed weights)
    if (up2_has_wino(Cin, Cout)) return conv_wino_up_prepare(w, ws + 32L * Cout * Cin, Cin, Cout, st, ws, ws + 16L * Cout * Cin);
    k_collapse_up_weights<<<stream_grid(n, 256), 256, 0, st>>>(w, ws, ws + 16L * Cout * Cin, Cout, Cin);
    VQW_LAUNCH_CHECK("collapse_up_weights");
    return VQW_OK;
}
int conv_up2_fwd(const float* x_low, const float* ws, const float* bias, float* y, int N, int h, int w, int Cin, int Cout, int relu,
                 hipStream_t st, float* stats) {
    // Winograd form with nine products (conv_wino_up.hip) where the shape allows and the statistics partials (if wanted) come in
    // the layout conv_up2_stat_tiles announced
    if (conv_wino_up_fwd_ok(Cin, Cout, N, h, w) && (!stats || conv_wino_up_stat_tiles(h, w) > 0))
        return conv_wino_up_fwd(x_low, ws + 32L * Cout * Cin, bias, y, N, h, w, Cin, Cout, relu, st, stats);
    // LDS-resident halo tiles (conv_halo.hip, 4-tap form): every input element travels L2 -> LDS 1.3 times per parity
    // instead of 4; only when the statistics partials (if wanted) come in the layout the caller sized for
    if (conv_halo_up2_ok(Cin, Cout, N, h, w) && (!stats || conv_halo_up2_stat_tiles(h, w) > 0))
        return conv_halo_up2_fwd(x_low, ws, bias, y, N, h, w, Cin, Cout, relu, st, stats);
    ConvIn in{x_low, nullptr, Cin, 0, 0};
    ConvGeom g{};
    g.ntaps = 4;
    g.out_mode = 1;      // parity = blockIdx.y; tap offsets and the weight block are derived from it in the kernel
    return dispatch_fwd(in, ws, bias, y, N, h, w, Cout, g, relu, st, stats);
}
bool conv_up2_dgrad_is_wino(int Cin, int Cout, int N, int h, int w) { return conv_wino_up_dgrad_ok(Cin, Cout, N, h, w); }
bool conv_up2_fwd_is_wino(int Cin, int Cout, int N, int h, int w) { return conv_wino_up_fwd_ok(Cin, Cout, N, h, w); }
int conv_up2_dgrad(const float* dy, const float* ws, float* dx_low, int N, int h, int w, int Cin, int Cout, hipStream_t st, int accumulate) {
    if (conv_wino_up_dgrad_ok(Cin, Cout, N, h, w)) return conv_wino_up_dgrad(dy, ws + 32L * Cout * Cin, dx_low, N, h, w, Cin, Cout, st, accumulate);
    if (accumulate) {
        vqw_set_error("conv_up2_dgrad: the accumulating form is served by the nine-product kernel only");
        return VQW_ERR_ARG;
    }
    ConvIn in{dy, nullptr, Cout, 0, 0};
    ConvGeom g{};
    g.ntaps = 16;
    g.src_mode = 2;
    for (int rs = 0; rs < 16; ++rs) {
        g.dy[rs] = (signed char)((rs >> 2) - 1);
        g.dx[rs] = (signed char)((rs & 3) - 1);
    }
    return dispatch_fwd(in, ws + 16L * Cout * Cin, nullptr, dx_low, N, h, w, Cin, g, 0, st);
}

// ---------------------------------------------------------------------------------------------
// 4x4, padding 1 convolutions of the PatchGAN discriminator (gan.hip) on the kernels above.
//   stride 2, forward:  M grid = the (h, w) output, source = the (2h, 2w) input, 16 taps at (2y + ky - 1, 2x + kx - 1):
//                       the geometry of the collapsed dgrad (src_mode 2); weights are the OHWI 4x4 kernel as it is.
//   stride 2, dgrad:    a transposed 4x4 stride-2 conv = four 2x2 convs by output parity = the collapsed forward
//                       (out_mode 1) with weights re-ordered by k_pack_k4s2_dgrad.
//   stride 1 (on a common H x W grid; the caller pads / crops the (H-1) x (W-1) side):  16-tap table with origin 1
//                       (forward) or 2 with flipped taps (dgrad).
__global__ void k_pack_k4s2_dgrad(const float* __restrict__ w, float* __restrict__ wc, int Cout, int Cin) {
    // wc[py*2+px][ci][a*2+b][co] = w[co][ky(py,a)][kx(px,b)][ci],  ky(0,0)=3 ky(0,1)=1 ky(1,0)=2 ky(1,1)=0
    long total = 16L * Cin * Cout;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int co = (int)(i % Cout);
        long r = i / Cout;
        int ab = (int)(r % 4); r /= 4;
        int ci = (int)(r % Cin);
        int par = (int)(r / Cin);
        int py = par >> 1, px = par & 1, a = ab >> 1, b = ab & 1;
        int ky = 3 - py - 2 * a, kx = 3 - px - 2 * b;
        wc[i] = w[(((long)co * 4 + ky) * 4 + kx) * Cin + ci];
    }
}
bool conv_k4_mfma_ok(int Cin, int Cout, long P) { return (Cin % 4 == 0) && (Cout % 4 == 0) && Cin >= 8 && Cout >= 8 && fits_u32(4 * P, Cin, Cout); }
int conv_k4s2_fwd(const float* x_high, const float* w, const float* bias, float* y_low, int N, int h, int w_, int Cin, int Cout,
                  hipStream_t st) {
    ConvIn in{x_high, nullptr, Cin, 0, 0};
    ConvGeom g{};
    g.ntaps = 16;
    g.src_mode = 2;
    for (int rs = 0; rs < 16; ++rs) {
        g.dy[rs] = (signed char)((rs >> 2) - 1);
        g.dx[rs] = (signed char)((rs & 3) - 1);
    }
    return dispatch_fwd(in, w, bias, y_low, N, h, w_, Cout, g, 0, st);
}
size_t conv_k4s2_dgrad_ws_floats(int Cin, int Cout) { return (size_t)16 * Cin * Cout; }
int conv_k4s2_dgrad(const float* gy_low, const float* w, float* ws, float* gx_high, int N, int h, int w_, int Cin, int Cout,
                    hipStream_t st) {
    long n = 16L * Cin * Cout;
    k_pack_k4s2_dgrad<<<stream_grid(n, 256), 256, 0, st>>>(w, ws, Cout, Cin);
    VQW_LAUNCH_CHECK("pack_k4s2_dgrad");
    ConvIn in{gy_low, nullptr, Cout, 0, 0};
    ConvGeom g{};
    g.ntaps = 4;
    g.out_mode = 1;
    return dispatch_fwd(in, ws, nullptr, gx_high, N, h, w_, Cin, g, 0, st);
}
// stride 1 on a common grid: x, y (or gy, gx) are both N x H x W maps; tap0 = 1 forward, 2 for the flipped dgrad weights
int conv_k4s1_grid(const float* src, const float* w16, const float* bias, float* dst, int N, int H, int W, int Csrc, int Cdst,
                   int tap0, hipStream_t st) {
    ConvIn in{src, nullptr, Csrc, 0, 0};
    return dispatch_fwd(in, w16, bias, dst, N, H, W, Cdst, plain_geom(4, 1, tap0), 0, st);
}

// 0 = auto, 1 = per-tap kernel only, (testing / A-B timing)
int g_wgrad_variant = 0;

// =============================================================================================
// wgrad, one tap per workgroup (1x1 convs, ragged widths)
// =============================================================================================
template <int BM, int BN, int WM, int WN, int KP>
__global__ void __launch_bounds__(64 * (BM / WM) * (BN / WN))
k_conv_mfma_wgrad(ConvIn in, const float* __restrict__ dy, float* __restrict__ part, int N, int H, int W, int Cout, int ks,
                  int dil, int ntm, int ntn, long per_split, int tap0) {
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
    const int taps = ks * ks, half = tap0;          // tap (ky, kx) reads pixel offset (ky - tap0, kx - tap0) * dil
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

    // Loader state (32-bit, incremental): every load slot (thread, j) walks pixels p, p+KP, p+2KP, ... so its
    // (h, w) coordinates are advanced by KP per chunk instead of being re-derived by division.
    constexpr int DROW = NT / (BM / 4);   // pixel rows covered by one pass over the dy tile
    constexpr int XROW = NT / (BN / 4);
    static_assert(NT % (BM / 4) == 0 && NT % (BN / 4) == 0, "tile rows must divide the workgroup");
    const int d_c = co_base + (tid % (BM / 4)) * 4;
    const bool d_cok = d_c < Cout;
    const int x_c = ci_base + (tid % (BN / 4)) * 4;
    const bool x_cok = x_c < Cin;
    const bool x_from0 = x_c < in.C0;
    const unsigned xC = x_from0 ? (unsigned)in.C0 : (unsigned)in.C1;
    const float* x_src = x_from0 ? in.src0 + x_c : in.src1 + (x_c - in.C0);
    const bool x_up = x_from0 && in.up0;
    const int shift = dyy * W + dxx;
    unsigned d_p = (unsigned)p_begin + tid / (BM / 4);       // pixel of dy load slot 0 (slot j adds j*DROW)
    unsigned x_p[LX];
    int x_h[LX], x_w[LX], x_n[LX];
#pragma unroll
    for (int j = 0; j < LX; ++j) {
        unsigned p = (unsigned)p_begin + tid / (BN / 4) + j * XROW;
        x_p[j] = p;
        unsigned q = p / (unsigned)W;
        x_w[j] = (int)(p - q * W);
        unsigned n = q / (unsigned)H;
        x_h[j] = (int)(q - n * H);
        x_n[j] = (int)n;
    }
    const unsigned pe = (unsigned)p_end;
    float4 rd[LD], rx[LX];
    auto load_chunk = [&]() {
#pragma unroll
        for (int j = 0; j < LD; ++j) {
            unsigned p = d_p + j * DROW;
            rd[j] = ld4_or_zero(dy + (size_t)p * Cout + d_c, p < pe && d_cok);   // Cout % 4 == 0 (conv_mfma_wgrad_ok)
        }
        d_p += KP;
#pragma unroll
        for (int j = 0; j < LX; ++j) {
            int hy = x_h[j] + dyy, wx = x_w[j] + dxx;
            bool ok = x_p[j] < pe && x_cok && (unsigned)hy < (unsigned)H && (unsigned)wx < (unsigned)W;
            unsigned pix = x_up ? ((unsigned)x_n[j] * Hs + (hy >> 1)) * Ws + (wx >> 1) : x_p[j] + (unsigned)shift;
            rx[j] = ld4_or_zero(x_src + (size_t)pix * xC, ok);
            // advance this slot by KP pixels
            x_p[j] += KP;
            x_w[j] += KP;
            while (x_w[j] >= W) { x_w[j] -= W; x_h[j] += 1; }
            while (x_h[j] >= H) { x_h[j] -= H; x_n[j] += 1; }
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
        load_chunk();
        store_chunk(0);
    }
    __syncthreads();
    for (int it = 0; it < nchunks; ++it) {
        const int cur = it & 1;
        if (it + 1 < nchunks) load_chunk();
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

// =============================================================================================
// wgrad, all nine taps per wave ("wg9"): the 3x3 workhorse
// =============================================================================================
// One WAVE = one worker: a 32(co) x 32(ci) weight tile for ALL 9 taps (9 x 16 accumulator registers) over a strided
// set of 32-pixel chunks.  A chunk lies inside one image row (W % 32 == 0), so the X operand of tap (ky,kx) is the
// row segment of row h+(ky-1)*d shifted by (kx-1)*d: three zero-filled halo segments [32+2d px][32 ci] are staged once
// per chunk in the wave's private LDS area next to the dY tile [32 px][32 co]; every A fragment (dY) read feeds 9 MFMAs.
// No workgroup barrier anywhere: a wave only reads LDS it wrote itself (LDS ops of one wave execute in order).
// Global loads for chunk i+1 are issued before the 144 MFMAs of chunk i (register prefetch) when they fit in registers.
// Partial slabs [worker-split][Cout][9][Cin] are reduced in fixed order by reduce_rows.
template <int NXL, bool PREFETCH>
__global__ void __launch_bounds__(256, 2)
k_conv_wgrad9(ConvIn in, const float* __restrict__ dy, float* __restrict__ part, float* __restrict__ bias_part, int N, int H,
              int W, int Cout, int dil, int n_ci_t, int ntiles, int nsplit_blocks, unsigned nb0, unsigned nb1, unsigned nbd) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int C0 = in.C0, C1 = in.C1, Cin = C0 + C1;
    const int SEG = 32 + 2 * dil;                   // pixels per halo row segment
    const int XF4 = 3 * SEG * 8;                    // float4 per X stage (3 segments x SEG px x 32 ci)
    const int wave_floats = 32 * 32 + 3 * SEG * 32;
    float* Ds = smem + wv * wave_floats;            // [32 px][32 co]
    float* Xs = Ds + 32 * 32;                       // [3][SEG px][32 ci]
    const int tile = blockIdx.x % ntiles, sblk = blockIdx.x / ntiles;
    const int co_base = (tile / n_ci_t) * 32, ci_base = (tile % n_ci_t) * 32;
    const int split = sblk * 4 + wv, nsplits = nsplit_blocks * 4;
    const unsigned P = (unsigned)N * H * W;
    const int nchunks = (int)(P >> 5);
    const int Hs = H >> 1, Ws = W >> 1;

    // chunk-invariant part of this lane's load slots (slot i of a lane = float4 number lane + 64*i of the stage;
    // 8 lanes per pixel, so its pixel slot is (lane>>3) + 8*i — nothing to keep in registers)
    const int d_c = co_base + (lane & 7) * 4;
    const bool d_ok = d_c < Cout;
    const int x_c = ci_base + (lane & 7) * 4;
    const bool x_cok = x_c < Cin;
    const bool x_from0 = ci_base < C0;              // a 32-wide ci tile never straddles the sources (wg9_ok)
    const unsigned xC = x_from0 ? (unsigned)C0 : (unsigned)C1;
    const unsigned x_cb = (unsigned)(x_from0 ? x_c : x_c - C0) * 4u;
    const unsigned nbx = x_from0 ? nb0 : nb1;
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(x_from0 ? in.src0 : in.src1, nbx), rsd = make_rsrc(dy, nbd);
    const bool x_up = x_from0 && in.up0;
    const int lpx = lane >> 3;
    const bool do_bias = bias_part != nullptr && ci_base == 0;

    float4 rd[4], rx[PREFETCH ? NXL : 8];
    float4 bsum;
    bsum.x = bsum.y = bsum.z = bsum.w = 0.f;
    int c_w0 = 0, c_h = 0, lpv = lpx;
    unsigned c_n = 0, c_p0 = 0;
    auto chunk_coords = [&](int c) {
        // opaque copy of the lane's pixel slot: keeps the per-slot (row, column) arithmetic inside the chunk loop
        // instead of being hoisted into ~3 registers per slot (the 144 accumulators leave no room for that)
        lpv = lpx;
        asm volatile("" : "+v"(lpv));
        c_p0 = (unsigned)c << 5;
        const unsigned q = c_p0 / (unsigned)W;
        c_w0 = (int)(c_p0 - q * W);
        c_n = q / (unsigned)H;
        c_h = (int)(q - c_n * H);
    };
    auto load_dy = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned p = c_p0 + lpv + 8 * i;
            rd[i] = buf_ld4(rsd, d_ok ? (p * (unsigned)Cout + d_c) * 4u : nbd);
        }
    };
    auto load_x = [&](int i, float4& dst) {       // slot i (compile-time after unrolling)
        const int px = lpv + 8 * i;
        const int r = (px >= SEG) + (px >= 2 * SEG);
        const int sgm = px - r * SEG;
        const int hy = c_h + (r - 1) * dil;
        const int wx = c_w0 - dil + sgm;
        const bool ok = px < 3 * SEG && x_cok && (unsigned)hy < (unsigned)H && (unsigned)wx < (unsigned)W;
        const unsigned pix = x_up ? (c_n * Hs + (unsigned)(hy >> 1)) * Ws + (unsigned)(wx >> 1)
                                  : (c_n * H + (unsigned)hy) * W + (unsigned)wx;
        dst = buf_ld4(rsx, ok ? pix * xC * 4u + x_cb : nbx);
    };
    auto store_dy = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(float4*)&Ds[(lane + 64 * i) * 4] = rd[i];
            bsum.x += rd[i].x; bsum.y += rd[i].y; bsum.z += rd[i].z; bsum.w += rd[i].w;   // fused bias gradient
        }
    };
    auto store_x = [&](int i, const float4& v) {
        const int f = lane + 64 * i;
        if (f < XF4) *(float4*)&Xs[f * 4] = v;
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int lcol = lane & 31, lk = lane >> 5;
    int c = split;
    if (PREFETCH && c < nchunks) {
        chunk_coords(c);
        load_dy();
#pragma unroll
        for (int i = 0; i < NXL; ++i) load_x(i, rx[i]);
    }
    for (; c < nchunks; c += nsplits) {
        if (PREFETCH) {
            store_dy();
#pragma unroll
            for (int i = 0; i < NXL; ++i) store_x(i, rx[i]);
            if (c + nsplits < nchunks) {
                chunk_coords(c + nsplits);
                load_dy();
#pragma unroll
                for (int i = 0; i < NXL; ++i) load_x(i, rx[i]);
            }
        } else {                                   // large dilation: stage in groups of 8 slots, no prefetch
            chunk_coords(c);
            load_dy();
            store_dy();
#pragma unroll
            for (int g = 0; g < NXL; g += 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (g + i < NXL) load_x(g + i, rx[i]);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (g + i < NXL) store_x(g + i, rx[i]);
            }
        }
        // (LDS accesses of one wave complete in issue order; the compiler inserts the lgkmcnt waits for the reads)
#pragma unroll 2
        for (int k = 0; k < 32; k += 2) {
            const float a = Ds[(k + lk) * 32 + lcol];
            const float* xr = Xs + (k + lk) * 32 + lcol;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float b0 = xr[(r * SEG) * 32];
                const float b1 = xr[(r * SEG + dil) * 32];
                const float b2 = xr[(r * SEG + 2 * dil) * 32];
                acc[r * 3 + 0] = MFMA32(a, b0, acc[r * 3 + 0]);
                acc[r * 3 + 1] = MFMA32(a, b1, acc[r * 3 + 1]);
                acc[r * 3 + 2] = MFMA32(a, b2, acc[r * 3 + 2]);
            }
        }
    }

    // The four waves of a workgroup hold partial sums of the SAME weight tile (different pixel splits): fold them
    // through LDS, tap by tap, and write ONE slab per workgroup (4x less slab traffic, fixed summation order).
    if (do_bias) {     // lanes with equal (lane & 7) hold the same 4 channels: butterfly over lane bits 3..5
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
            bsum.x += __shfl_xor(bsum.x, o, 64); bsum.y += __shfl_xor(bsum.y, o, 64);
            bsum.z += __shfl_xor(bsum.z, o, 64); bsum.w += __shfl_xor(bsum.w, o, 64);
        }
    }
    __syncthreads();                      // every wave is done with its staging area
    float* red = smem;                    // [4 waves][32 co][32 ci] reuses the staging space (>= 16 KiB)
    float* o = part + (size_t)sblk * Cout * 9 * Cin;
    if (do_bias) {
        if (lane < 8) *(float4*)&red[4096 + wv * 32 + lane * 4] = bsum;
        __syncthreads();
        if (tid < 32 && co_base + tid < Cout)
            bias_part[(size_t)sblk * Cout + co_base + tid] = (red[4096 + tid] + red[4096 + 32 + tid]) + (red[4096 + 64 + tid] + red[4096 + 96 + tid]);
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            red[wv * 1024 + row * 32 + (lane & 31)] = acc[t][r];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int idx = tid + e * 256;          // element of the 32x32 tile
            int row = idx >> 5, col = idx & 31;
            int co = co_base + row, ci = ci_base + col;
            float v = (red[idx] + red[1024 + idx]) + (red[2048 + idx] + red[3072 + idx]);
            if (co < Cout && ci < Cin) o[((size_t)co * 9 + t) * Cin + ci] = v;
        }
        __syncthreads();
    }
}

// shapes the wg9 kernel takes, and its split count (shared by the workspace query and the launcher)
static const int g_wino_wgrad = []{ const char* e = getenv("VQW_WINOGRAD_WGRAD"); return e ? atoi(e) : 1; }();      // 0: direct-form weight gradients (A/B)
static inline bool wg9_ok(int C0, int C1, int Cout, int ks, int W, int dil, long P) {
    return ks == 3 && (W % 32 == 0) && (C0 % 4 == 0) && (C1 % 4 == 0) && (Cout % 4 == 0) && dil <= 24 && P >= 32 &&
           (C1 == 0 || C0 % 32 == 0);
}
static inline int wg9_split_blocks(int Cin, int Cout, long P) {
    long tiles = (long)ceil_div(Cout, 32) * ceil_div(Cin, 32);
    long nchunks = P / 32;
    long nb = ceil_div(512, tiles);                 // ~2048 wave-workers = 2 per SIMD over the chip
    long cap = nchunks / 8 > 1 ? nchunks / 8 : 1;   // at least 2 chunks per worker
    if (nb > cap) nb = cap;
    return (int)(nb < 1 ? 1 : nb);
}

// =============================================================================================
// wgrad of a 3x3 conv over a nearest x2 up-sampled input, collapsed ("wgup")
// =============================================================================================
// dW[ky][kx] = sum over the four output parities (py,px) of G[py][px][a(py,ky)][b(px,kx)] with
//   G[py][px][a][b] = sum_{y,x} dY[2y+py, 2x+px] (x) X_low[y + py-1+a, x + px-1+b]            (4/9 of the direct MFMA work)
// One wave = one (32co x 32ci tile, row parity py, split): 8 accumulators G[px][a][b] over chunks of 32 consecutive
// full-resolution pixels of a row Y = 2y+py.  k-steps pair pixels of EQUAL column parity (X, X+2), so both k slots of
// an MFMA share the same (px, a, b); the dY tile and two 18-pixel halo segments of X_low sit in the wave's private LDS.
// Slab layout [split][py][Cout][8][Cin]; k_reduce_wgup folds it into dW (and accumulates if asked).
__global__ void __launch_bounds__(256, 2)
k_conv_wgrad_up(const float* __restrict__ xlow, const float* __restrict__ dy, float* __restrict__ part, float* __restrict__ bias_part,
                int N, int h, int w, int Cin, int Cout, int n_ci_t, int ntiles, int nsplit_blocks, unsigned nbx, unsigned nbd) {
    __shared__ __attribute__((aligned(16))) float smem[4 * (32 * 32 + 2 * 18 * 32)];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float* Ds = smem + wv * (32 * 32 + 2 * 18 * 32);     // [32 px][32 co]
    float* Xs = Ds + 32 * 32;                            // [2 rows][18 cols][32 ci]
    int b = blockIdx.x;
    const int tile = b % ntiles; b /= ntiles;
    const int py = b & 1;
    const int sblk = b >> 1;
    const int co_base = (tile / n_ci_t) * 32, ci_base = (tile % n_ci_t) * 32;
    const int split = sblk * 4 + wv, nsplits = nsplit_blocks * 4;
    const int W2 = 2 * w, H2 = 2 * h;
    const int cpr = W2 >> 5;                              // chunks per full-res row
    const int nchunks = N * h * cpr;                      // rows of this parity only
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(xlow, nbx), rsd = make_rsrc(dy, nbd);
    const int d_c = co_base + (lane & 7) * 4;
    const bool d_ok = d_c < Cout;
    const int x_c = ci_base + (lane & 7) * 4;
    const bool x_cok = x_c < Cin;
    const int lpx = lane >> 3;
    const bool do_bias = bias_part != nullptr && ci_base == 0;

    float4 rd[4], rx[5], bsum;
    bsum.x = bsum.y = bsum.z = bsum.w = 0.f;
    auto load_chunk = [&](int c) {
        const int kx = c % cpr;
        const int q = c / cpr;
        const int y = q % h, n = q / h;
        const unsigned pfull = ((unsigned)(n * H2 + 2 * y + py) * W2 + (kx << 5));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned p = pfull + lpx + 8 * i;
            rd[i] = buf_ld4(rsd, d_ok ? (p * (unsigned)Cout + d_c) * 4u : nbd);
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int ps = lpx + 8 * i;                   // pixel slot in [0, 36): row r, column col of the halo stage
            const int r = ps >= 18;
            const int col = ps - 18 * r;
            const int row = y + py - 1 + r, cx = (kx << 4) - 1 + col;
            const bool ok = ps < 36 && x_cok && (unsigned)row < (unsigned)h && (unsigned)cx < (unsigned)w;
            rx[i] = buf_ld4(rsx, ok ? (((unsigned)(n * h + row) * w + cx) * (unsigned)Cin + x_c) * 4u : nbx);
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(float4*)&Ds[(lane + 64 * i) * 4] = rd[i];
            bsum.x += rd[i].x; bsum.y += rd[i].y; bsum.z += rd[i].z; bsum.w += rd[i].w;
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int f = lane + 64 * i;
            if (f < 2 * 18 * 8) *(float4*)&Xs[f * 4] = rx[i];
        }
    };

    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int lcol = lane & 31, lk = lane >> 5;
    int c = split;
    if (c < nchunks) load_chunk(c);
    for (; c < nchunks; c += nsplits) {
        store_chunk();
        if (c + nsplits < nchunks) load_chunk(c + nsplits);
#pragma unroll 2
        for (int kk = 0; kk < 8; ++kk) {
            const int pe = 4 * kk + 2 * lk;               // even-column pixel of this lane half; odd one is pe + 1
            const int xr = 2 * kk + lk;                   // its low-res column (relative); halo column 0 = xl0 - 1
            const float ae = Ds[pe * 32 + lcol], ao = Ds[(pe + 1) * 32 + lcol];
            const float* x0 = Xs + xr * 32 + lcol;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const float v0 = x0[(a * 18 + 0) * 32], v1 = x0[(a * 18 + 1) * 32], v2 = x0[(a * 18 + 2) * 32];
                acc[0 * 4 + a * 2 + 0] = MFMA32(ae, v0, acc[0 * 4 + a * 2 + 0]);   // px=0: columns xr + b
                acc[0 * 4 + a * 2 + 1] = MFMA32(ae, v1, acc[0 * 4 + a * 2 + 1]);
                acc[1 * 4 + a * 2 + 0] = MFMA32(ao, v1, acc[1 * 4 + a * 2 + 0]);   // px=1: columns xr + 1 + b
                acc[1 * 4 + a * 2 + 1] = MFMA32(ao, v2, acc[1 * 4 + a * 2 + 1]);
            }
        }
    }

    if (do_bias) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
            bsum.x += __shfl_xor(bsum.x, o, 64); bsum.y += __shfl_xor(bsum.y, o, 64);
            bsum.z += __shfl_xor(bsum.z, o, 64); bsum.w += __shfl_xor(bsum.w, o, 64);
        }
        if (lane < 8 && d_ok) *(float4*)&bias_part[((size_t)split * 2 + py) * Cout + d_c] = bsum;
    }
    float* o = part + ((size_t)split * 2 + py) * Cout * 8 * Cin;
    const int ci = ci_base + (lane & 31);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int co = co_base + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (co < Cout && ci < Cin) o[((size_t)co * 8 + t) * Cin + ci] = acc[t][r];
        }
    }
}

// dW[co][ky][kx][ci] (+)= sum_split sum_{py,px} part[split][py][co][px*4 + a(py,ky)*2 + b(px,kx)][ci]
__global__ void __launch_bounds__(1024) k_reduce_wgup(const float* __restrict__ part, float* __restrict__ dw, int Cout, int Cin,
                                                      int nsplits, int acc) {
    __shared__ float sm[16][64];
    const int cx = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long n = (long)Cout * 9 * Cin;
    const long i = (long)blockIdx.x * 64 + cx;
    float a = 0.f;
    if (i < n) {
        const int ci = (int)(i % Cin);
        const long r = i / Cin;
        const int t = (int)(r % 9), co = (int)(r / 9);
        const int ky = t / 3, kx = t % 3;
        for (int s = g; s < nsplits; s += 16) {
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const int aa = py == 0 ? (ky == 0 ? 0 : 1) : (ky == 2 ? 1 : 0);
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    const int bb = px == 0 ? (kx == 0 ? 0 : 1) : (kx == 2 ? 1 : 0);
                    a += part[((((long)s * 2 + py) * Cout + co) * 8 + px * 4 + aa * 2 + bb) * Cin + ci];
                }
            }
        }
    }
    sm[g][cx] = a;
    __syncthreads();
    if (g == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k][cx];
        dw[i] = acc ? dw[i] + t : t;
    }
}

static inline int wgup_split_blocks(int Cin, int Cout, long Plow) {
    long tiles = (long)ceil_div(Cout, 32) * ceil_div(Cin, 32) * 2;
    long nchunks = Plow / 8;                              // N*h*(2w/32) chunks per parity, (2w/32) = w/16
    long nb = ceil_div(768, tiles);                       // ~3 waves per SIMD over the chip
    long cap = nchunks / 16 > 1 ? nchunks / 16 : 1;
    if (nb > cap) nb = cap;
    return (int)(nb < 1 ? 1 : nb);
}
bool conv_up2_wgrad_ok(int Cin, int Cout, int N, int h, int w) {
    return (w % 16 == 0) && (Cin % 4 == 0) && (Cout % 4 == 0) && Cin >= 8 && Cout >= 8 && fits_u32(4L * N * h * w, Cin, Cout);
}
// slabs of k_conv_wgrad_up (also what the 4x4 stride-2 weight gradient below sizes its workspace with)
static size_t wgup_ws_floats(int Cin, int Cout, int N, int h, int w) {
    return (size_t)wgup_split_blocks(Cin, Cout, (long)N * h * w) * 4 * 2 * ((size_t)Cout * 8 * Cin + Cout);
}
size_t conv_up2_wgrad_ws_floats(int Cin, int Cout, int N, int h, int w) {
    const size_t a = wgup_ws_floats(Cin, Cout, N, h, w);
    const size_t b = conv_wino_up_wgrad_ok(Cin, Cout, N, h, w) ? conv_wino_up_wgrad_ws_floats(Cin, Cout, N, h, w) : 0;
    return a > b ? a : b;
}
bool conv_up2_wgrad_is_wino(int Cin, int Cout, int N, int h, int w) { return conv_wino_up_wgrad_ok(Cin, Cout, N, h, w); }
int conv_up2_wgrad(const float* xlow, const float* dy, float* dw, float* dbias, float* ws, int N, int h, int w, int Cin, int Cout,
                   int acc, hipStream_t st) {
    if (conv_wino_up_wgrad_ok(Cin, Cout, N, h, w))        // Winograd form, nine products (conv_wino_up.hip)
        return conv_wino_up_wgrad(xlow, dy, dw, dbias, ws, N, h, w, Cin, Cout, acc, st);
    const long Plow = (long)N * h * w;
    const int n_ci_t = ceil_div(Cin, 32), ntiles = ceil_div(Cout, 32) * n_ci_t;
    const int nsb = wgup_split_blocks(Cin, Cout, Plow);
    const unsigned nbx = (unsigned)(Plow * Cin * 4), nbd = (unsigned)(4 * Plow * Cout * 4);
    float* bpart = dbias ? ws + (size_t)nsb * 4 * 2 * Cout * 8 * Cin : nullptr;
    k_conv_wgrad_up<<<ntiles * 2 * nsb, 256, 0, st>>>(xlow, dy, ws, bpart, N, h, w, Cin, Cout, n_ci_t, ntiles, nsb, nbx, nbd);
    VQW_LAUNCH_CHECK("conv_wgrad_up");
    if (dbias) {
        int rc = reduce_rows(bpart, dbias, Cout, nsb * 4 * 2, st, acc);
        if (rc) return rc;
    }
    const long n = (long)Cout * 9 * Cin;
    k_reduce_wgup<<<(unsigned)((n + 63) / 64), 1024, 0, st>>>(ws, dw, Cout, Cin, nsb * 4, acc);
    VQW_LAUNCH_CHECK("reduce_wgup");
    return VQW_OK;
}

// 4x4 stride-2 pad-1 weight gradient (PatchGAN): dW[co][ky][kx][ci] = sum dYlow[y, x][co] * Xhigh[2y + ky - 1, 2x + kx - 1][ci]
// is the G matrix set of the kernel above with the roles swapped (Xhigh in the "dY" slot, dYlow in the "Xlow" slot):
// slab[split][py][ci][px*4 + a*2 + b][co] with ky = 3 - py - 2a, kx = 3 - px - 2b.
__global__ void __launch_bounds__(256) k_reduce_wg_k4s2(const float* __restrict__ part, float* __restrict__ dw, int Cout, int Cin,
                                                         int nsplits, int acc) {
    const long n = (long)Cout * 16 * Cin;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int ci = (int)(i % Cin);
        const long r = i / Cin;
        const int t = (int)(r % 16), co = (int)(r / 16);
        const int ky = t >> 2, kx = t & 3;
        const int py = (ky & 1) ? 0 : 1, a = ky >= 2 ? 0 : 1;
        const int px = (kx & 1) ? 0 : 1, b = kx >= 2 ? 0 : 1;
        float v = 0.f;
        for (int s = 0; s < nsplits; ++s) v += part[((((long)s * 2 + py) * Cin + ci) * 8 + px * 4 + a * 2 + b) * Cout + co];
        dw[i] = acc ? dw[i] + v : v;
    }
}
bool conv_k4s2_wgrad_ok(int Cin, int Cout, int N, int h, int w) { return conv_up2_wgrad_ok(Cout, Cin, N, h, w); }
size_t conv_k4s2_wgrad_ws_floats(int Cin, int Cout, int N, int h, int w) { return wgup_ws_floats(Cout, Cin, N, h, w); }
int conv_k4s2_wgrad(const float* x_high, const float* gy_low, float* dw, float* ws, int N, int h, int w, int Cin, int Cout, int acc,
                    hipStream_t st) {
    const long Plow = (long)N * h * w;
    const int cin_r = Cout, cout_r = Cin;                 // roles inside k_conv_wgrad_up
    const int n_ci_t = ceil_div(cin_r, 32), ntiles = ceil_div(cout_r, 32) * n_ci_t;
    const int nsb = wgup_split_blocks(cin_r, cout_r, Plow);
    const unsigned nbx = (unsigned)(Plow * cin_r * 4), nbd = (unsigned)(4 * Plow * cout_r * 4);
    k_conv_wgrad_up<<<ntiles * 2 * nsb, 256, 0, st>>>(gy_low, x_high, ws, nullptr, N, h, w, cin_r, cout_r, n_ci_t, ntiles, nsb, nbx, nbd);
    VQW_LAUNCH_CHECK("conv_wgrad_up(k4s2)");
    const long n = (long)Cout * 16 * Cin;
    k_reduce_wg_k4s2<<<stream_grid(n, 256), 256, 0, st>>>(ws, dw, Cout, Cin, nsb * 4, acc);
    VQW_LAUNCH_CHECK("reduce_wg_k4s2");
    return VQW_OK;
}
// 4x4 stride-1 weight gradient on a common N x H x W grid (dy zero-padded to the grid by the caller)
int conv_k4s1_wgrad_grid(const float* x, const float* dy_grid, float* dw, float* ws, int N, int H, int W, int Cin, int Cout, int acc,
                         hipStream_t st);

bool conv_mfma_wgrad_ok(const ConvIn& in, int Cout, int ks) {
    (void)ks;
    int Cin = in.C0 + in.C1;
    return (in.C0 % 4 == 0) && (in.C1 % 4 == 0) && Cin >= 8 && Cout >= 8 && (Cout % 4 == 0);
}

static inline int wg_tile(int c) { return c > 64 ? 128 : (c > 32 ? 64 : 32); }

// Split-K factor of the per-tap wgrad kernel: as many pixel ranges as fill the GPU ONCE with resident workgroups.
// Tiles up to 64 x 64 are single-wave workgroups with 16-32 KB of LDS: ten of them fit a CU, and the small 1x1 layers
// that use them are HBM-bound streams which need that many waves to keep enough loads in flight (16->32 1x1 at 256^2:
// 512 one-wave workgroups = 2 waves per CU read at 1 TB/s).  The 128-wide tiles are MFMA-bound: a second, partly
// filled round of workgroups and short K ranges (slab write + reduction per range) cost more than they balance.
static inline int wgrad_splits(int Cin, int Cout, int ks, long P) {
    int bm = wg_tile(Cout), bn = wg_tile(Cin);
    long tiles = (long)ceil_div(Cout, bm) * ceil_div(Cin, bn) * ks * ks;
    const int nw = (bm > 64 ? 2 : 1) * (bn > 64 ? 2 : 1);                 // waves per workgroup
    const long lds = 2L * 32 * (bm + bn) * 4;                             // KP = 32 pixels, double-buffered
    long per_cu = 160 * 1024 / lds;
    if (per_cu * nw > 12) per_cu = 12 / nw;
    if (per_cu < 1) per_cu = 1;
    long s = 256 * per_cu / tiles;                                        // floor: one round
    long cap = P / 256 > 1 ? P / 256 : 1;               // at least 256 pixels per split
    if (s > cap) s = cap;
    if (s > 4096) s = 4096;
    return s < 1 ? 1 : (int)s;
}

size_t conv_mfma_wgrad_ws_floats(int Cin, int Cout, int ks, long P) {
    size_t a = (size_t)wgrad_splits(Cin, Cout, ks, P) * Cout * ks * ks * Cin;
    size_t b = ks == 3 ? (size_t)wg9_split_blocks(Cin, Cout, P) * ((size_t)Cout * 9 * Cin + Cout) : 0;
    return a > b ? a : b;
}

template <int BM, int BN, int KP>
static int launch_wgrad(const ConvIn& in, const float* dy, float* dw, float* ws, int N, int H, int W, int Cout, int ks, int dil,
                        hipStream_t st, int acc, int tap0 = -1) {
    if (tap0 < 0) tap0 = ks >> 1;
    constexpr int WM = BM > 64 ? 64 : BM, WN = BN > 64 ? 64 : BN;
    constexpr int NT = 64 * (BM / WM) * (BN / WN);
    const int Cin = in.C0 + in.C1;
    const long P = (long)N * H * W;
    const int splits = wgrad_splits(Cin, Cout, ks, P);
    long per = (P + splits - 1) / splits;
    per = ((per + KP - 1) / KP) * KP;
    const int ntm = ceil_div(Cout, BM), ntn = ceil_div(Cin, BN);
    const long nout = (long)Cout * ks * ks * Cin;
    float* part = (splits > 1 || acc) ? ws : dw;
    k_conv_mfma_wgrad<BM, BN, WM, WN, KP><<<ntm * ntn * ks * ks * splits, NT, 0, st>>>(in, dy, part, N, H, W, Cout, ks, dil,
                                                                                         ntm, ntn, per, tap0);
    VQW_LAUNCH_CHECK("conv_mfma_wgrad");
    if (splits > 1 || acc) return reduce_rows(ws, dw, nout, splits, st, acc);
    return VQW_OK;
}

// dbias != nullptr asks the kernel to produce the bias gradient too; returns 1 (not an error) in *bias_done when it did.
// true when conv_mfma_wgrad takes the Winograd form for the layer (the profile scope prices it at the FLOPs it executes)
// dilation 2 in Winograd form: the (32 x 32)-block kernel on the four phase images (VQW_WINOGRAD_DIL2=0: the row-chain kernels)
static const int g_wino_dil2 = []{ const char* e = getenv("VQW_WINOGRAD_DIL2"); return e ? atoi(e) : 1; }();
extern int g_wino_mode;
static bool wgrad_is_wino_dil2(const ConvIn& in, int N, int H, int W, int Cout, int ks, int dil) {
    return g_wino_dil2 && g_wino_mode == 0 && g_wgrad_variant != 1 && g_wino_wgrad && ks == 3 && dil == 2 && in.C1 == 0 && !in.up0 &&
           fits_u32((long)N * H * W, in.C0, Cout) && conv_wino32_wgrad_dil2_ok(in.C0, Cout, H, W);
}
bool conv_mfma_wgrad_is_wino(const ConvIn& in, int N, int H, int W, int Cout, int ks, int dil) {
    if (wgrad_is_wino_dil2(in, N, H, W, Cout, ks, dil)) return true;
    if (in.C1 > 0 && (in.C0 % 16 != 0 || in.C1 % 16 != 0)) return false;      // a 16-channel ci block must not straddle the sources
    if (in.up0 && ((H | W) & 1)) return false;
    return g_wgrad_variant != 1 && g_wino_wgrad && ks == 3 && dil == 1 && (long)N * H * W >= 32 &&
           fits_u32((long)N * H * W, in.C0 + in.C1, Cout) && conv_wino_wgrad_ok(in.C0 + in.C1, Cout, N, H, W);
}
int conv_mfma_wgrad(const ConvIn& in, const float* dy, float* dw, float* dbias, int* bias_done, float* ws, int N, int H, int W,
                    int Cout, int ks, int dil, hipStream_t st, int acc) {
    *bias_done = 0;
    if (!fits_u32((long)N * H * W, in.C0 + in.C1, Cout)) return conv_direct_wgrad(in, dy, dw, ws, N, H, W, Cout, ks, dil, st, acc);
    if (conv_mfma_wgrad_is_wino(in, N, H, W, Cout, ks, dil)) {     // Winograd form (conv_wino.hip): widths that are multiples of 16
        const int Cin = in.C0 + in.C1;
        const long nout = (long)Cout * 9 * Cin;
        *bias_done = dbias != nullptr;
        int kt = 1;
        const bool d2 = wgrad_is_wino_dil2(in, N, H, W, Cout, ks, dil);
        const int nsbw = d2 ? conv_wino32_wgrad_blocks(Cin, Cout, 4 * N, H / 2, W / 2, wg9_split_blocks(Cin, Cout, (long)N * H * W), &kt)
                            : conv_wino_wgrad_blocks(in, Cout, N, H, W, wg9_split_blocks(Cin, Cout, (long)N * H * W), &kt);
        float* bp = dbias ? ws + (size_t)nsbw * nout : nullptr;
        int rc = d2 ? conv_wino32_wgrad(in, dy, ws, bp, N, H, W, Cin, Cout, nsbw, kt, st, 2)
                    : conv_wino_wgrad(in, dy, ws, bp, N, H, W, Cout, nsbw, kt, st);
        if (rc) return rc;
        if (dbias) {
            rc = reduce_rows(bp, dbias, Cout, nsbw, st, acc);
            if (rc) return rc;
        }
        return reduce_rows(ws, dw, nout, nsbw, st, acc);
    }
    if (g_wgrad_variant != 1 && wg9_ok(in.C0, in.C1, Cout, ks, W, dil, (long)N * H * W) &&
        conv_dil_wgrad_ok(in, N, H, W, Cout, ks, dil))       // dilated 32-channel layers: rows walked along the residue chains
        return conv_dil_wgrad(in, dy, dw, ws, wg9_split_blocks(in.C0, Cout, (long)N * H * W), N, H, W, Cout, dil, acc, st);
    if (g_wgrad_variant != 1 && wg9_ok(in.C0, in.C1, Cout, ks, W, dil, (long)N * H * W)) {
        *bias_done = dbias != nullptr;
        const int Cin = in.C0 + in.C1;
        const long P = (long)N * H * W;
        const int n_ci_t = ceil_div(Cin, 32), ntiles = ceil_div(Cout, 32) * n_ci_t;
        const int nsb = wg9_split_blocks(Cin, Cout, P);
        const int SEG = 32 + 2 * dil;
        const int nxl = ceil_div(3 * SEG * 8, 64);
        const size_t lds = (size_t)4 * (32 * 32 + 3 * SEG * 32) * sizeof(float);
        const long nout = (long)Cout * 9 * Cin;
        const unsigned nb0 = (unsigned)((in.up0 ? P / 4 : P) * in.C0 * 4), nb1 = (unsigned)(P * in.C1 * 4);
        const unsigned nbd = (unsigned)(P * Cout * 4);
        if (conv_wgrad_tile_ok(in.C0, in.C1, Cout, ks, W, dil)) {           // block-shared tiles (conv_halo.hip)
            int kt = 1;
            const int nsbt = conv_wgrad_tile_blocks(Cin, Cout, N, H, W, nsb, &kt);
            float* bp = dbias ? ws + (size_t)nsbt * nout : nullptr;
            int rc = conv_wgrad_tile(in, dy, ws, bp, N, H, W, Cout, nsbt, kt, st);
            if (rc) return rc;
            if (dbias) {
                rc = reduce_rows(bp, dbias, Cout, nsbt, st, acc);
                if (rc) return rc;
            }
            return reduce_rows(ws, dw, nout, nsbt, st, acc);
        }
        float* bpart = dbias ? ws + (size_t)nsb * nout : nullptr;          // [workgroups][Cout] after the weight slabs
#define WG9_LAUNCH(NXL_, PF_)                                                                                         \
        do {                                                                                                          \
            static bool attr_set = false;   /* > 64 KiB of dynamic LDS needs the opt-in, once per instantiation */     \
            if (!attr_set) {                                                                                          \
                if (hipFuncSetAttribute((const void*)k_conv_wgrad9<NXL_, PF_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        160 * 1024) != hipSuccess) {                                                  \
                    vqw_set_error("conv_wgrad9: cannot raise the dynamic LDS limit");                                 \
                    return VQW_ERR_HIP;                                                                               \
                }                                                                                                     \
                attr_set = true;                                                                                      \
            }                                                                                                         \
            k_conv_wgrad9<NXL_, PF_><<<ntiles * nsb, 256, lds, st>>>(in, dy, ws, bpart, N, H, W, Cout, dil, n_ci_t, ntiles, nsb, \
                                                                      nb0, nb1, nbd);                                       \
        } while (0)
        if (nxl <= 13) WG9_LAUNCH(13, true);
        else if (nxl <= 14) WG9_LAUNCH(14, true);
        else if (nxl <= 17) WG9_LAUNCH(17, false);
        else if (nxl <= 21) WG9_LAUNCH(21, false);
        else if (nxl <= 26) WG9_LAUNCH(26, false);
        else WG9_LAUNCH(30, false);
#undef WG9_LAUNCH
        VQW_LAUNCH_CHECK("conv_wgrad9");
        if (dbias) {
            int rc = reduce_rows(bpart, dbias, Cout, nsb, st, acc);
            if (rc) return rc;
        }
        return reduce_rows(ws, dw, nout, nsb, st, acc);
    }
    const int bm = wg_tile(Cout), bn = wg_tile(in.C0 + in.C1);
#define WG_CASE(M_, N_, K_) \
    if (bm == M_ && bn == N_) return launch_wgrad<M_, N_, K_>(in, dy, dw, ws, N, H, W, Cout, ks, dil, st, acc)
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

size_t conv_k4s1_wgrad_ws_floats(int Cin, int Cout, long P) { return (size_t)wgrad_splits(Cin, Cout, 4, P) * Cout * 16 * Cin; }
int conv_k4s1_wgrad_grid(const float* x, const float* dy_grid, float* dw, float* ws, int N, int H, int W, int Cin, int Cout, int acc,
                         hipStream_t st) {
    ConvIn in{x, nullptr, Cin, 0, 0};
    const int bm = wg_tile(Cout), bn = wg_tile(Cin);
#define WG_CASE(M_, N_, K_) \
    if (bm == M_ && bn == N_) return launch_wgrad<M_, N_, K_>(in, dy_grid, dw, ws, N, H, W, Cout, 4, 1, st, acc, 1)
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
    vqw_set_error("conv_k4s1_wgrad_grid: no tile configuration");
    return VQW_ERR_ARG;
}
