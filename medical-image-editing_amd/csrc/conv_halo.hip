// 3x3 stride-1 convolution (forward and dgrad) with the input tile + halo RESIDENT in LDS, exact fp32 MFMA, NHWC.
//
// Why a second forward kernel: the implicit-GEMM kernel (conv_mfma.hip) gathers its A operand once per tap, so every
// input element travels L2 -> LDS nine times.  With a narrow N tile (Cout <= 64) that traffic, not the matrix cores,
// bounds it: SQ counters show the MFMA pipes 45-50 % busy while ~60 KB per CU are permanently in flight
// (profiles/r01_sq_counters.txt).  Here a workgroup loads the (TH+2) x 34 pixel halo of a TH x 32 output tile ONCE per
// channel chunk and all nine taps read it from LDS at shifted addresses, so the L2 -> CU traffic drops ~7x.
//
//   * a workgroup walks several consecutive tiles: the halo of the next (tile, channel chunk) item is fetched into
//     registers while the matrix cores work on the current one and written to the other LDS buffer afterwards; one
//     barrier per item.  The trip count is uniform per workgroup, every wave reaches the end.
//   * weights: [9 taps][BN couts][KC channels] in LDS; when the layer has a single channel chunk they are loaded once
//     per workgroup and stay (WPERSIST), otherwise they are double-buffered along with the halo.
//   * LDS rows (one pixel's / one cout's KC channels) are padded by 4 floats: the ds_read_b128 fragment reads of 32
//     consecutive pixels (or couts) are bank-conflict free, and a tap shift is just a constant address offset.
//   * raw buffer loads: pixels outside the image get the out-of-range offset and come back as 0 (the zero padding).
//   * the loader also does the virtual nearest x2 up-sampling and the two-source channel concat (chunks never straddle
//     the sources), so UpBlock's conv over [up(x) | skip] runs here as well.
#include "common.h"
#include "conv_common.h"
#include "mfma_util.h"
#include <cstdlib>

static int halo_kt_env() {
    const char* e = getenv("VQW_HALO_KT");
    return e ? atoi(e) : 0;
}
static const int g_halo_kt = halo_kt_env();     // tuning aid: spatial tiles per workgroup (0 = default)

namespace {

constexpr int HALO_W = 34;     // 32 output columns + 1 on each side

struct HaloArgs {
    ConvIn in;
    const float* w;
    const float* bias;
    float* y;
    int N, H, W, Cout;
    int tilesY, tilesX, nsp;   // spatial tiles per image column / row, total
    int ntn, nch;              // cout tiles, channel chunks
    int kt;                    // consecutive spatial tiles per workgroup
    int relu;
    unsigned nb0, nb1, nbw;
};

template <int NW, int TH, int BN, int KC, bool WPERSIST>
__global__ void __launch_bounds__(64 * NW, 1) k_conv_halo(HaloArgs a) {
    constexpr int NT = 64 * NW;
    constexpr int KP = KC + 4;
    constexpr int C4 = KC / 4;
    constexpr int WC = (BN / 32 > 1 && NW > TH) ? 2 : 1;     // waves across N
    constexpr int WR = NW / WC;                               // waves across tile rows
    constexpr int TM = TH / WR, TN = BN / 32 / WC;
    static_assert(WR * TM == TH && WC * TN * 32 == BN, "wave grid must tile the workgroup tile");
    constexpr int HPIX = (TH + 2) * HALO_W;
    constexpr int HF = HPIX * C4;                 // float4 per halo chunk
    constexpr int LH = (HF + NT - 1) / NT;
    constexpr int WF = 9 * BN * C4;               // float4 per weight chunk
    constexpr int LW = (WF + NT - 1) / NT;
    constexpr int HBUF = HPIX * KP;               // floats
    constexpr int WBUF = 9 * BN * KP;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Hs = smem;                              // [2][HBUF]
    float* Ws = smem + 2 * HBUF;                   // [WPERSIST ? 1 : 2][WBUF]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int H = a.H, W = a.W, Cout = a.Cout;
    const int C0 = a.in.C0, C1 = a.in.C1, Cin = C0 + C1;
    const int Hs2 = H >> 1, Ws2 = W >> 1;
    const int up0 = a.in.up0;
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(a.in.src0, a.nb0), rs1 = make_rsrc(a.in.src1, a.nb1),
                                 rsw = make_rsrc(a.w, a.nbw);

    // work split: a workgroup keeps one cout tile and walks `kt` consecutive spatial tiles.  Tiles are ordered down
    // a 32-pixel column strip first (consecutive items share 2 of their TH+2 halo rows in L1/L2), and the XCD remap
    // keeps neighbouring strips and the cout tiles of one strip on one XCD's L2.
    const int ntn = a.ntn, nch = a.nch;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lb % ntn;
    const int sp0 = (lb / ntn) * a.kt;
    const int co_base = tile_n * BN;
    const int my_tiles = min(a.kt, a.nsp - sp0);
    const int nitems = my_tiles * nch;
    const int per_img = a.tilesY * a.tilesX;

    // loader slots (fixed for the whole kernel)
    int h_lds[LH];
    short h_y[LH], h_x[LH];
    unsigned h_c[LH];
#pragma unroll
    for (int j = 0; j < LH; ++j) {
        int f = tid + j * NT;
        bool ok = (HF % NT == 0) || f < HF;
        int hp = ok ? f / C4 : 0, c4 = f % C4;
        h_lds[j] = ok ? hp * KP + c4 * 4 : -1;
        h_y[j] = (short)(hp / HALO_W);
        h_x[j] = (short)(hp % HALO_W);
        h_c[j] = (unsigned)c4 * 4u;
    }
    unsigned w_off[LW];
    int w_lds[LW];
#pragma unroll
    for (int j = 0; j < LW; ++j) {
        int f = tid + j * NT;
        bool ok = (WF % NT == 0) || f < WF;
        int row = ok ? f / C4 : 0, c4 = f % C4;      // row = tap * BN + n
        int tap = row / BN, n = row % BN;
        int co = co_base + n;
        w_lds[j] = ok ? row * KP + c4 * 4 : -1;
        w_off[j] = (ok && co < Cout) ? (((unsigned)co * 9 + tap) * Cin + c4 * 4) * 4u : a.nbw;
    }

    float4 rh[LH], rw[WPERSIST ? 1 : LW];
    auto issue = [&](int item) {       // global -> registers for item (tile, chunk)
        const int t = item / nch, ch = item - t * nch;
        const int sp = sp0 + t;
        const int n = sp / per_img, rem = sp - n * per_img;
        const int tx = rem / a.tilesY, ty = rem - tx * a.tilesY;
        const int y0 = ty * TH - 1, x0 = tx * 32 - 1;
        const int cc = ch * KC;
        if (cc < C0) {
            const unsigned img = up0 ? (unsigned)n * Hs2 * Ws2 : (unsigned)n * H * W;
#pragma unroll
            for (int j = 0; j < LH; ++j) {
                int yy = y0 + h_y[j], xx = x0 + h_x[j];
                bool ok = h_lds[j] >= 0 && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
                unsigned pix = up0 ? img + (unsigned)((yy >> 1) * Ws2 + (xx >> 1)) : img + (unsigned)(yy * W + xx);
                rh[j] = buf_ld4(rs0, ok ? (pix * (unsigned)C0 + cc + h_c[j]) * 4u : a.nb0);
            }
        } else {
            const unsigned img = (unsigned)n * H * W;
#pragma unroll
            for (int j = 0; j < LH; ++j) {
                int yy = y0 + h_y[j], xx = x0 + h_x[j];
                bool ok = h_lds[j] >= 0 && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
                unsigned pix = img + (unsigned)(yy * W + xx);
                rh[j] = buf_ld4(rs1, ok ? (pix * (unsigned)C1 + (cc - C0) + h_c[j]) * 4u : a.nb1);
            }
        }
        if constexpr (!WPERSIST) {
#pragma unroll
            for (int j = 0; j < LW; ++j) rw[j] = buf_ld4(rsw, w_off[j] == a.nbw ? a.nbw : w_off[j] + (unsigned)cc * 4u);
        }
    };
    auto commit = [&](int buf) {       // registers -> LDS buffer `buf`
#pragma unroll
        for (int j = 0; j < LH; ++j)
            if (h_lds[j] >= 0) *(float4*)&Hs[buf * HBUF + h_lds[j]] = rh[j];
        if constexpr (!WPERSIST) {
#pragma unroll
            for (int j = 0; j < LW; ++j)
                if (w_lds[j] >= 0) *(float4*)&Ws[buf * WBUF + w_lds[j]] = rw[j];
        }
    };

    if (nitems <= 0) return;           // uniform per workgroup
    if constexpr (WPERSIST) {
#pragma unroll
        for (int j = 0; j < LW; ++j) {
            float4 v = buf_ld4(rsw, w_off[j]);
            if (w_lds[j] >= 0) *(float4*)&Ws[w_lds[j]] = v;
        }
    }
    issue(0);
    commit(0);
    __syncthreads();

    const int wr = wv / WC, wc = wv % WC;
    const int lrow = lane & 31, lk = (lane >> 5) * 4;
    // fragment bases (floats): A = pixel (row wr*TM + i, column lrow) of the halo at tap (0,0); B = cout wc*TN*32 + j*32 + lrow
    const int a_base = ((wr * TM) * HALO_W + lrow) * KP + lk;
    const int b_base = (wc * TN * 32 + lrow) * KP + lk;

    f32x16 acc[TM][TN];
    int cur = 0;
    for (int item = 0; item < nitems; ++item) {
        const int t = item / nch, ch = item - t * nch;
        if (item + 1 < nitems) issue(item + 1);
        if (ch == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        const float* Hc = Hs + cur * HBUF + a_base;
        const float* Wc = Ws + (WPERSIST ? 0 : cur * WBUF) + b_base;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int kg = 0; kg < KC / 8; ++kg) {
                float4 av[TM], bv[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) av[i] = *(const float4*)&Hc[((i + ky) * HALO_W + kx) * KP + kg * 8];
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[j] = *(const float4*)&Wc[(tap * BN + j * 32) * KP + kg * 8];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = MFMA32(av[i].x, bv[j].x, acc[i][j]);
                        acc[i][j] = MFMA32(av[i].y, bv[j].y, acc[i][j]);
                        acc[i][j] = MFMA32(av[i].z, bv[j].z, acc[i][j]);
                        acc[i][j] = MFMA32(av[i].w, bv[j].w, acc[i][j]);
                    }
            }
        }
        if (ch == nch - 1) {
            // epilogue: C/D layout of the 32x32 MFMA: col = lane&31 (cout), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel column)
            const int sp = sp0 + t;
            const int n = sp / per_img, rem = sp - n * per_img;
            const int tx = rem / a.tilesY, ty = rem - tx * a.tilesY;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int co = co_base + (wc * TN + j) * 32 + (lane & 31);
                const bool cok = co < Cout;
                const float bvv = (a.bias && cok) ? a.bias[co] : 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int yy = ty * TH + wr * TM + i;
                    if (!cok || yy >= H) continue;
                    float* yrow = a.y + ((size_t)((size_t)n * H + yy) * W + tx * 32) * Cout + co;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int col = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        float v = acc[i][j][r] + bvv;
                        yrow[(size_t)col * Cout] = a.relu ? fmaxf(v, 0.f) : v;
                    }
                }
            }
        }
        if (item + 1 < nitems) commit(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
}

template <int NW, int TH, int BN, int KC, bool WPERSIST>
int launch_halo(const ConvIn& in, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int relu,
                hipStream_t st) {
    constexpr int KP = KC + 4;
    constexpr size_t lds = (size_t)(2 * (TH + 2) * HALO_W * KP + (WPERSIST ? 1 : 2) * 9 * BN * KP) * sizeof(float);
    static_assert(lds <= 160 * 1024, "halo tile does not fit the 160 KB LDS");
    static bool attr_set = false;
    auto kern = k_conv_halo<NW, TH, BN, KC, WPERSIST>;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            vqw_set_error("conv_halo: cannot raise the dynamic LDS limit");
            return VQW_ERR_HIP;
        }
        attr_set = true;
    }
    HaloArgs a;
    a.in = in; a.w = w; a.bias = bias; a.y = y;
    a.N = N; a.H = H; a.W = W; a.Cout = Cout;
    a.tilesY = ceil_div(H, TH); a.tilesX = W / 32; a.nsp = N * a.tilesY * a.tilesX;
    a.ntn = ceil_div(Cout, BN); a.nch = (in.C0 + in.C1) / KC;
    a.relu = relu;
    const long P = (long)N * H * W;
    a.nb0 = (unsigned)((in.up0 ? P / 4 : P) * in.C0 * 4);
    a.nb1 = (unsigned)(P * in.C1 * 4);
    a.nbw = (unsigned)((long)Cout * 9 * (in.C0 + in.C1) * 4);
    // The LDS footprint allows one workgroup per CU: one workgroup per CU, each with an even share of the tiles, so the
    // prologue (weights + first halo, not overlapped with MFMA work) is paid once.  Shorter runs per workgroup
    // (VQW_HALO_KT) would let the dispatcher rebalance when other kernels hold CUs; measured 0.5-1 % slower in the step.
    const int even = ceil_div((long)a.nsp * a.ntn, 256);
    int kt = g_halo_kt > 0 ? g_halo_kt : even;
    if (kt > even) kt = even;
    a.kt = kt < 1 ? 1 : kt;
    k_conv_halo<NW, TH, BN, KC, WPERSIST><<<ceil_div(a.nsp, a.kt) * a.ntn, 64 * NW, lds, st>>>(a);
    VQW_LAUNCH_CHECK("conv_halo");
    return VQW_OK;
}

}  // namespace

int g_halo_mode = 0;     // 0 auto, 1 off (A/B timing, tests of the implicit-GEMM kernel)

// 3x3, dilation 1, rows of 32-pixel tiles, channel chunks that divide both sources.  Measured faster than the
// implicit-GEMM kernel on every such layer of the model, wide ones included (64-wide cout tiles, halo re-read per tile).
bool conv_halo_fwd_ok(const ConvIn& in, int N, int H, int W, int Cout, int ks, int dil) {
    const int Cin = in.C0 + in.C1;
    if (g_halo_mode != 0 || ks != 3 || dil != 1 || W % 32 != 0 || H < 2 || Cout < 16) return false;
    if (Cin % 16 != 0 || (in.C1 > 0 && in.C0 % 16 != 0)) return false;
    if (in.up0 && ((H | W) & 1)) return false;
    return (long)N * H * W * (Cin > Cout ? Cin : Cout) * 4 <= 0xFFFFFFE0L;
}

int conv_halo_fwd(const ConvIn& in, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int relu,
                  hipStream_t st) {
    const int Cin = in.C0 + in.C1;
    const bool wide = Cout > 32;        // 64-wide cout tile
    if (Cin == 32 && in.C1 == 0) {      // single chunk: weights stay in LDS
        if (wide) return launch_halo<8, 4, 64, 32, true>(in, w, bias, y, N, H, W, Cout, relu, st);
        return launch_halo<8, 8, 32, 32, true>(in, w, bias, y, N, H, W, Cout, relu, st);
    }
    if (Cin == 16 && in.C1 == 0) {
        if (wide) return launch_halo<8, 8, 64, 16, true>(in, w, bias, y, N, H, W, Cout, relu, st);
        return launch_halo<8, 8, 32, 16, true>(in, w, bias, y, N, H, W, Cout, relu, st);
    }
    if (wide) return launch_halo<8, 8, 64, 16, false>(in, w, bias, y, N, H, W, Cout, relu, st);
    return launch_halo<8, 8, 32, 16, false>(in, w, bias, y, N, H, W, Cout, relu, st);
}
