// 3x3 convolution over a nearest x2 up-sampled input (StyledResUpBlock `conv`, `conv1`: blocks.py:100-112) in Winograd
// F(2x2, 3x3) form: NINE of the sixteen products.
//
// A Winograd tile of the full-resolution output starts at even coordinates, so its 4 x 4 input patch of up(x) has the rows
// (a, b, b, c) of three low-resolution rows (and the same in columns).  B^T d = (d0 - d2, d1 + d2, d2 - d1, d1 - d3) is then
// (a - b, 2 b, 0, b - c): row 2 and column 2 of V = B^T d B vanish - 9 of the 16 element-wise products remain, a quarter of
// the direct form's matrix work (the low-resolution "collapsed" form of conv_mfma.hip / conv_halo.hip needs 16 of 36).
// The same structure on the other two passes:
//   * input gradient: g_low = sum of the 2 x 2 tile of g = A^T M A, i.e. c^T M c with c = column sums of A^T = (1, 2, 0, -1):
//     row / column 2 of M is never needed, and with c folded into the transformed weights the nine products ACCUMULATE INTO
//     ONE accumulator - the input gradient is a single GEMM with K = 9 Cout whose A operand is the transformed dY patch
//     (no inverse transform, 4 accumulator registers per 16 x 16 block);
//   * weight gradient: dU[xi] = sum_tiles dM[xi]^T V[xi] is zero wherever V[xi] is.
// Built like conv_wino64.hip (see there for the accounting): beside fp32 MFMAs every VALU instruction costs 4.4 matrix-pipe
// cycles, LDS / SALU instructions nothing - per-region load offsets, scalar chunk offsets, immediates everywhere else.
#include "common.h"
#include "conv_common.h"
#include "mfma_util.h"
#include <cstdlib>
#include <type_traits>

namespace {

static int env_int_up(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
static const int g_wup_env = env_int_up("VQW_WINOGRAD_UP", 1);
static const int g_wup_max_blocks = []{ int v = env_int_up("VQW_CONV_MAX_BLOCKS", 256); return v < 8 ? 8 : (v > 256 ? 256 : v); }();
// weight-gradient kernels (they run on the side lanes beside the chain): VQW_WGRAD_MAX_BLOCKS leaves CUs to the chain's kernels (experiment)
static const int g_wup_max_blocks_wg = []{ int v = env_int_up("VQW_WGRAD_MAX_BLOCKS", g_wup_max_blocks); return v < 8 ? 8 : (v > 256 ? 256 : v); }();

typedef float f32x2 __attribute__((ext_vector_type(2)));

// the nine positions: (i, j) with i, j in {0, 1, 3}
__device__ __host__ constexpr int up_row(int s) { return s == 2 ? 3 : s; }       // s = 0, 1, 2 -> Winograd index 0, 1, 3

// Transformed weights of the three passes, all in the chunked layout [K / 8][9][N][8] the kernels stream:
//   uf (forward):        K = Cin,  N = Cout: (G w G^T)[i][j] x 2^(i == 1) x 2^(j == 1)    (the factor 2 of V's row / column 1)
//   ud (input gradient): K = Cout, N = Cin:  c_i c_j (G wt G^T)[i][j], wt = w flipped and transposed, c = (1, 2, -1)
__global__ void k_wino_up_weights(const float* __restrict__ w, float* __restrict__ uf, float* __restrict__ ud, int Cout, int Cin,
                                  float* __restrict__ wc, float* __restrict__ wd) {
    if (wc) {      // the collapsed 2x2-tap weights of the same layer ride in this launch (conv_up2_prepare)
        const long nc = 32L * Cout * Cin;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nc; i += (long)gridDim.x * blockDim.x) collapse_up_element(w, wc, wd, i, Cout, Cin);
    }
    const long n = (long)Cout * Cin;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int co = (int)(e / Cin), ci = (int)(e % Cin);
        double g[3][3];
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) g[ky][kx] = (double)w[(((long)co * 3 + ky) * 3 + kx) * Cin + ci];
        for (int flip = 0; flip < 2; ++flip) {
            double t[4][3], u[4][4];
            for (int kx = 0; kx < 3; ++kx) {
                const double g0 = flip ? g[2][2 - kx] : g[0][kx], g1 = flip ? g[1][2 - kx] : g[1][kx], g2 = flip ? g[0][2 - kx] : g[2][kx];
                t[0][kx] = g0;
                t[1][kx] = 0.5 * (g0 + g1 + g2);
                t[2][kx] = 0.5 * (g0 - g1 + g2);
                t[3][kx] = g2;
            }
            for (int i = 0; i < 4; ++i) {
                u[i][0] = t[i][0];
                u[i][1] = 0.5 * (t[i][0] + t[i][1] + t[i][2]);
                u[i][2] = 0.5 * (t[i][0] - t[i][1] + t[i][2]);
                u[i][3] = t[i][2];
            }
            for (int si = 0; si < 3; ++si)
                for (int sj = 0; sj < 3; ++sj) {
                    const int i = up_row(si), j = up_row(sj), xi = si * 3 + sj;
                    if (!flip) {
                        const double f = (i == 1 ? 2.0 : 1.0) * (j == 1 ? 2.0 : 1.0);
                        uf[((((long)(ci >> 3) * 9 + xi) * Cout + co) * 8) + (ci & 7)] = (float)(f * u[i][j]);
                    } else {
                        const double ci_ = (i == 0 ? 1.0 : i == 1 ? 2.0 : -1.0), cj_ = (j == 0 ? 1.0 : j == 1 ? 2.0 : -1.0);
                        ud[((((long)(co >> 3) * 9 + xi) * Cin + ci) * 8) + (co & 7)] = (float)(ci_ * cj_ * u[i][j]);
                    }
                }
        }
    }
}

constexpr int WU_KPH = 10;                 // floats per halo pixel in LDS (8 channels + 2)

// =====================================================================================================================
// Input gradient: g_low[tile][ci] = sum over the nine xi and the couts of ud[xi][ci][co] V[xi][tile][co],  V = B^T dY B
// =====================================================================================================================
struct WUpDgArgs {
    const float* dy;           // (N, H = 2h, W = 2w, Cout) full resolution
    const float* u;            // ud: [Cout / 8][9][Cin][8]
    float* g;                  // (N, h, w, Cin) low resolution
    int N, H, W, Cin, Cout;    // layer channel counts: K = Cout (chunks of 8), N dimension = Cin
    int tilesY, tilesX, nsp, ntn, nch, kt;
    unsigned nbd, nbu, nbg;
};

// Workgroup = 16 x 32 full-resolution pixels (128 tiles = 8 x 16 low-resolution pixels) x NCO = 16 NBW input channels; wave w =
// tile row w.  One accumulator set (NBW x 4 registers) takes all nine xi.
template <int NBW, bool ACC>             // ACC: g += result (a later member of a gradient group: conv_common.h)
__global__ void __launch_bounds__(512, 1) k_conv_wino_up_dgrad(WUpDgArgs a) {
    constexpr int NT = 512, NCO = 16 * NBW;
    constexpr int RW = 32, TR = 16, HR = TR + 2, HWV = RW + 2, HWS = 34;
    constexpr int HBUF = HR * HWS * WU_KPH;            // 6120 floats
    constexpr int UBUF = 9 * NCO * 8;
    constexpr int HF = HR * HWV * 2;                   // 1224 float4
    constexpr int LH = (HF + NT - 1) / NT;             // 3
    constexpr int UF = 9 * NCO * 2;                    // float4 per U chunk: 2304 / 1152
    constexpr int LU = (UF + NT - 1) / NT;             // 5 / 3
    constexpr int NPOS = 18 * NBW;                     // MFMAs per item and wave: 144 / 72
    constexpr int NOPS = 42;                           // transform: 24 column + 18 row operations
    constexpr int OPSTEP2 = NBW == 8 ? 3 : 2;          // operation o sits at position T0 + o * OPSTEP2 / 2
    constexpr int T0 = NPOS - (NOPS * OPSTEP2 + 1) / 2;
    constexpr int CP = T0 - (LH + LU) - 4;             // first LDS commit of the prefetched data
    static_assert(1 + 2 * (LH + LU) <= CP, "prefetch loads are issued before their commits");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    // layout: halo buffers [2][HBUF], then U buffers [2][UBUF]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
    const __amdgpu_buffer_rsrc_t rsd = make_rsrc(a.dy, a.nbd), rsu = make_rsrc(a.u, a.nbu), rsg = make_rsrc(a.g, a.nbg);

    const int ntn = a.ntn, nch = a.nch;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lb % ntn;
    const int sp0 = (lb / ntn) * a.kt;
    const int co_base = tile_n * NCO;                  // first input channel (N dimension) of the workgroup
    const int my_tiles = min(a.kt, a.nsp - sp0);
    const int per_img = a.tilesY * a.tilesX;
    if (my_tiles <= 0) return;

    // ---- loader slots ----
    const int c4 = tid & 1;
    auto halo_pixel = [&](int j, int& hy, int& hx) {
        int f = tid + j * NT;
        if (f >= HF) f -= HF;
        const int hp = f >> 1;
        hy = hp / HWV;
        hx = hp - hy * HWV;
    };
    int h_lds[LH];
#pragma unroll
    for (int j = 0; j < LH; ++j) {
        int hy, hx;
        halo_pixel(j, hy, hx);
        h_lds[j] = (hy * HWS + hx) * WU_KPH + c4 * 4;
    }
    unsigned h_voff[LH];
    auto region_offsets = [&](int n, int tx, int ty) {
        const int y0 = ty * TR - 1, x0 = tx * RW - 1;
#pragma unroll
        for (int j = 0; j < LH; ++j) {
            int hy, hx;
            halo_pixel(j, hy, hx);
            const int yy = y0 + hy, xx = x0 + hx;
            const bool ok = ((unsigned)yy < (unsigned)H) & ((unsigned)xx < (unsigned)W);
            const unsigned pix = ((unsigned)n * H + (unsigned)yy) * W + (unsigned)xx;
            h_voff[j] = sel_u32(ok, pix * (unsigned)Cout * 4u + (unsigned)c4 * 16u, 0xFFFFFFFFu);
        }
    };
    // U float4 f -> row f / 2 = xi * NCO + n, channel quad f % 2; the last slot wraps (duplicates)
    unsigned u_voff[LU];
    int u_lds[LU];
#pragma unroll
    for (int j = 0; j < LU; ++j) {
        int f = tid + j * NT;
        if (f >= UF) f -= UF;
        const int row = f >> 1, xi = row / NCO, n = row - xi * NCO;
        u_voff[j] = (((unsigned)xi * Cin + co_base + n) * 8u + c4 * 4) * 4u;
        u_lds[j] = 2 * HBUF + row * 8 + ((c4 ^ ((n >> 3) & 1)) * 4);
    }
    const unsigned u_cstride = 9u * Cin * 32u;         // per 8-channel chunk

    float4 rh[LH], ru[LU];
    auto ld4 = [&](__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0);
        float4 f;
        unsigned a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
        f.x = __uint_as_float(a0); f.y = __uint_as_float(a1); f.z = __uint_as_float(a2); f.w = __uint_as_float(a3);
        return f;
    };
    auto issue_h = [&](int j, int chunk) { rh[j] = ld4(rsd, h_voff[j], chunk * 32); };
    // (readfirstlane: left alone the compiler multiplies in a VGPR and wraps the load in a waterfall loop)
    auto issue_u = [&](int j, int chunk) { ru[j] = ld4(rsu, u_voff[j], __builtin_amdgcn_readfirstlane(chunk * u_cstride)); };
    auto commit_h = [&](int j, int buf) {
        float* p = smem + buf * HBUF + h_lds[j];
        f32x2 lo, hi;
        lo.x = rh[j].x; lo.y = rh[j].y; hi.x = rh[j].z; hi.y = rh[j].w;
        *(f32x2*)p = lo;
        *(f32x2*)(p + 2) = hi;
    };
    auto commit_u = [&](int j, int buf) { *(float4*)&smem[u_lds[j] + buf * UBUF] = ru[j]; };

    // ---- item cursors ----
    int cn, ctx, cty, ch = 0;
    {
        cn = sp0 / per_img;
        const int rem = sp0 - cn * per_img;
        ctx = rem / a.tilesY;
        cty = rem - ctx * a.tilesY;
    }
    auto advance = [&](int& n, int& tx, int& ty, int& c) {
        const int adv = c + 1 == nch ? 1 : 0;
        c = adv ? 0 : c + 1;
        const int ty1 = ty + adv, wy = ty1 == a.tilesY ? 1 : 0;
        ty = wy ? 0 : ty1;
        const int tx1 = tx + wy, wx = tx1 == a.tilesX ? 1 : 0;
        tx = wx ? 0 : tx1;
        n += wx;
    };
    int n1 = cn, tx1 = ctx, ty1 = cty, ch1 = 0;
    advance(n1, tx1, ty1, ch1);
    int n2 = n1, tx2 = tx1, ty2 = ty1, ch2 = ch1;
    advance(n2, tx2, ty2, ch2);

    // ---- fragments: lane (tile m = lane & 15, channel pair q = lane >> 4); wave = tile row ----
    const int m = lane & 15, q = lane >> 4;
    auto opaque = [](int x) { asm volatile("" : "+v"(x)); return x; };
    int a_row[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) a_row[r] = opaque(((2 * wv + r) * HWS + 2 * m) * WU_KPH + 2 * q);
    const int b_swz = ((q >> 1) ^ (m >> 3)) * 4 + (q & 1) * 2;
    // (indices in units of f32x2: a float index that went through `opaque` carries no alignment and the 8-byte fragment read
    // would be split into ds_read2_b32 with a VALU add each)
    const int b_u0 = opaque((2 * HBUF + m * 8 + b_swz) / 2), b_u1 = opaque((2 * HBUF + UBUF + m * 8 + b_swz) / 2);      // + (xi * NCO + nb * 16) * 4
    float mone, pone;                  // -1 / +1 the compiler cannot see through: x + pone * y rounds like x + y and is never paired into v_pk_add_f32
    { float s = -1.f; asm volatile("v_mov_b32 %0, %1" : "=v"(mone) : "v"(s)); }
    { float s = 1.f; asm volatile("v_mov_b32 %0, %1" : "=v"(pone) : "v"(s)); }

    f32x4 acc[NBW];
    f32x2 bf[2][NBW];
    f32x2 dcol[2][4];                  // two patch columns in flight
    float e[2][3][4];                  // [channel][row 0, 1, 3][column] after the column pass
    float v[2][2][9];                  // [parity][channel][xi]

    auto read_col = [&](int buf, int c) {
        const float* Hc = smem + buf * HBUF + c * WU_KPH;
#pragma unroll
        for (int r = 0; r < 4; ++r) dcol[c & 1][r] = *(const f32x2*)&Hc[a_row[r]];
    };
    // operation o = 0..41: column pass o < 24: column o / 6, channel (o % 6) / 3, row o % 3; row pass: channel, row, column
    auto xform_op = [&](int par, int o) {
        if (o < 24) {
            const int c = o / 6, t = (o % 6) / 3, s = o % 3;
            const float d0 = dcol[c & 1][0][t], d1 = dcol[c & 1][1][t], d2 = dcol[c & 1][2][t], d3 = dcol[c & 1][3][t];
            e[t][s][c] = s == 0 ? __builtin_fmaf(mone, d2, d0) : s == 1 ? __builtin_fmaf(pone, d2, d1) : __builtin_fmaf(mone, d3, d1);
        } else {
            const int k = o - 24, t = k / 9, s = (k % 9) / 3, sj = k % 3;
            const float e0 = e[t][s][0], e1 = e[t][s][1], e2 = e[t][s][2], e3 = e[t][s][3];
            v[par][t][s * 3 + sj] = sj == 0 ? __builtin_fmaf(mone, e2, e0) : sj == 1 ? __builtin_fmaf(pone, e2, e1) : __builtin_fmaf(mone, e3, e1);
        }
    };
    auto op_pos = [](int o) { return T0 + (o * OPSTEP2) / 2; };
    auto xform_slot = [&](int buf, int par, int p) {
#pragma unroll
        for (int o = 0; o < NOPS; ++o) {
            if (o < 24 && o % 6 == 0 && op_pos(o) - 3 == p) read_col(buf, o / 6);      // a column's reads three positions ahead
            if (op_pos(o) == p) xform_op(par, o);
        }
    };

    // ---- prologue ----
    region_offsets(cn, ctx, cty);
#pragma unroll
    for (int j = 0; j < LH; ++j) issue_h(j, 0);
#pragma unroll
    for (int j = 0; j < LU; ++j) issue_u(j, 0);
#pragma unroll
    for (int j = 0; j < LH; ++j) commit_h(j, 0);
#pragma unroll
    for (int j = 0; j < LU; ++j) commit_u(j, 0);
    region_offsets(n1, tx1, ty1);
#pragma unroll
    for (int j = 0; j < LH; ++j) issue_h(j, ch1);
#pragma unroll
    for (int j = 0; j < LH; ++j) commit_h(j, 1);
    __syncthreads();
#pragma unroll
    for (int o = 0; o < NOPS; ++o) {
        if (o < 24 && o % 6 == 0) read_col(0, o / 6);
        xform_op(0, o);
    }
    __syncthreads();
    region_offsets(n2, tx2, ty2);

    // One item of parity PAR: operands v[PAR], U chunk in U buffer PAR, the next item's halo in halo buffer PAR ^ 1; the halo
    // of item i+2 goes to halo buffer PAR, the U chunk of item i+1 to U buffer PAR ^ 1.
    auto body = [&](auto PAR, auto FIRST) {
        constexpr int par = decltype(PAR)::value;
        constexpr bool first = decltype(FIRST)::value;
        auto ldb = [&](int xi, int nb) {
            bf[xi & 1][nb] = ((const f32x2*)smem)[(par ? b_u1 : b_u0) + (xi * NCO + nb * 16) * 4];
        };
        auto slot = [&](int p) {
            if (p >= 1 && p < 1 + 2 * LH && (p & 1)) issue_h((p - 1) >> 1, ch2);
            if (p >= 1 + 2 * LH && p < 1 + 2 * (LH + LU) && (p & 1)) issue_u((p - 1 - 2 * LH) >> 1, ch1);
            if (p >= CP && p < CP + LH) commit_h(p - CP, par);
            if (p >= CP + LH && p < CP + LH + LU) commit_u(p - CP - LH, par ^ 1);
            xform_slot(par ^ 1, par ^ 1, p);
        };
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) { ldb(0, nb); __builtin_amdgcn_sched_barrier(0); }
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int xi = 0; xi < 9; ++xi)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    if (k == 0 && xi + 1 < 9) ldb(xi + 1, nb);
                    acc[nb] = MFMA16(v[par][k][xi], k == 0 ? bf[xi & 1][nb].x : bf[xi & 1][nb].y,
                                     (first && xi == 0 && k == 0) ? zero4 : acc[nb]);
                    slot((xi * 2 + k) * NBW + nb);
                    __builtin_amdgcn_sched_barrier(0);
                }
    };
    // C/D layout (16x16): col = lane & 15 (channel of the N block), row = 4 (lane >> 4) + r (tile = low-resolution column)
    auto epilogue = [&]() {
        const int ylow = cty * (TR / 2) + wv, xlow0 = ctx * (RW / 2) + 4 * q;
        const unsigned base = (((unsigned)cn * (H >> 1) + (unsigned)ylow) * (W >> 1) + (unsigned)xlow0) * (unsigned)Cin + co_base + m;
        const int voff = (int)sel_u32(ylow < (H >> 1), base * 4u, 0xFFFFFFFFu);
        if (ACC) {
            float old[NBW][4];
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) old[nb][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsg, voff, r * Cin * 4 + nb * 64, 0));
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[nb][r] = old[nb][r] + acc[nb][r];
        }
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[nb][r]), rsg, voff, r * Cin * 4 + nb * 64, 0);
    };

    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    auto shift = [&]() {
        cn = n1; ctx = tx1; cty = ty1; ch = ch1;
        n1 = n2; tx1 = tx2; ty1 = ty2; ch1 = ch2;
        advance(n2, tx2, ty2, ch2);
    };
    for (int reg = 0; reg < my_tiles; ++reg) {     // nch is even: a region is nch / 2 (even, odd) item pairs
        body(P0{}, std::true_type{});
        __syncthreads();
        shift();
        body(P1{}, std::false_type{});
        __syncthreads();
        for (int c = 2; c < nch; c += 2) {
            shift();
            if (ch2 == 0) region_offsets(n2, tx2, ty2);
            body(P0{}, std::false_type{});
            __syncthreads();
            shift();
            body(P1{}, std::false_type{});
            __syncthreads();
        }
        epilogue();
        shift();
        if (ch2 == 0) region_offsets(n2, tx2, ty2);
    }
}


// =====================================================================================================================
// Forward: y (full resolution) = A^T [ sum_ci uf[xi][co][ci] V[xi][tile][ci] ] A,  V from the LOW-resolution 3 x 3 patch
// =====================================================================================================================
//   rows of B^T d for d = (a, b, b, c): R0 = a - b, R1 = b (its factor 2 sits in uf), R3 = b - c; the same along columns:
//   12 add / sub per channel and tile for the nine V.  Workgroup = 16 x 32 output pixels (128 tiles, low-resolution halo
//   10 x 18 pixels) x 64 couts; wave w = tile row w with all nine xi (9 x 4 N blocks x 4 = 144 accumulators): 24 VALU per
//   72 MFMAs, no xi split and no exchange.
struct WUpFwArgs {
    const float* x;            // (N, h, w, Cin) low resolution
    const float* u;            // uf: [Cin / 8][9][Cout][8]
    const float* bias;
    float* y;                  // (N, 2h, 2w, Cout)
    int N, h, w, Cin, Cout;
    int tilesY, tilesX, nsp, ntn, nch, kt;
    int relu;
    unsigned nbx, nbu, nby;
    float* stats;              // optional [N][tilesY * tilesX][Cout][2]
    // pair mode (round 4): Cout = 64 = TWO 32-cout layers of the same input (StyledResUpBlock's `conv` and `conv1`, blocks.py:100-112)
    // run as one 64-cout launch; couts 32..63 go to y2 / stats2, each output tensor (and statistics array) has 32 channels
    float* y2;
    float* stats2;
};

constexpr int WF_KPH = 12;                 // floats per low-resolution halo pixel: adjacent tiles are ONE pixel apart (12 m: conflict-free)

__global__ void __launch_bounds__(512, 1) k_conv_wino_up_fwd(WUpFwArgs a) {
    constexpr int NT = 512, NBW = 4, NCO = 64;
    constexpr int LR = 10, LWV = 18, LWS = 18;         // low-resolution halo rows / columns of a region
    constexpr int HBUF = LR * LWS * WF_KPH;            // 2160 floats
    constexpr int UBUF = 9 * NCO * 8;                  // 4608 floats
    constexpr int HF = LR * LWV * 2;                   // 360 float4
    constexpr int UF = 9 * NCO * 2;                    // 1152 float4
    constexpr int LU = (UF + NT - 1) / NT;             // 3
    constexpr int NPOS = 72, NOPS = 24, T0 = NPOS - NOPS, CP = 30;
    static_assert(HF <= NT && 1 + 2 * (1 + LU) <= CP && CP + 1 + LU <= T0 - 3, "slot layout");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    // layout: halo buffers [2][HBUF], U buffers [2][UBUF], statistics [2][8 waves][64][2]
    float* Rs = smem + 2 * HBUF + 2 * UBUF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = a.h, w = a.w, Cin = a.Cin, Cout = a.Cout;
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(a.x, a.nbx), rsu = make_rsrc(a.u, a.nbu), rsy = make_rsrc(a.y, a.nby);
    const bool split = a.y2 != nullptr;                    // uniform: two 32-channel outputs
    const __amdgpu_buffer_rsrc_t rsy2 = make_rsrc(split ? a.y2 : a.y, a.nby);
    const int Cst = split ? 32 : a.Cout;                   // channels per output pixel as stored

    const int ntn = a.ntn, nch = a.nch;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lb % ntn;
    const int sp0 = (lb / ntn) * a.kt;
    const int co_base = tile_n * NCO;
    const int my_tiles = min(a.kt, a.nsp - sp0);
    const int per_img = a.tilesY * a.tilesX;
    if (my_tiles <= 0) return;

    // ---- loader slots: one halo slot (threads past the 360 float4 repeat another thread's), three U slots ----
    const int c4 = tid & 1;
    const int hf = tid < HF ? tid : tid - HF;
    const int h_hy = (hf >> 1) / LWV, h_hx = (hf >> 1) - h_hy * LWV;
    const int h_lds = (h_hy * LWS + h_hx) * WF_KPH + c4 * 4;
    unsigned h_voff;
    auto region_offsets = [&](int n, int tx, int ty) {
        const int yy = ty * 8 - 1 + h_hy, xx = tx * 16 - 1 + h_hx;
        const bool ok = ((unsigned)yy < (unsigned)h) & ((unsigned)xx < (unsigned)w);
        const unsigned pix = ((unsigned)n * h + (unsigned)yy) * w + (unsigned)xx;
        h_voff = sel_u32(ok, pix * (unsigned)Cin * 4u + (unsigned)c4 * 16u, 0xFFFFFFFFu);
    };
    unsigned u_voff[LU];
    int u_lds[LU];
#pragma unroll
    for (int j = 0; j < LU; ++j) {
        int f = tid + j * NT;
        if (f >= UF) f -= UF;
        const int row = f >> 1, xi = row / NCO, n = row - xi * NCO;
        u_voff[j] = (((unsigned)xi * Cout + co_base + n) * 8u + c4 * 4) * 4u;
        u_lds[j] = 2 * HBUF + row * 8 + ((c4 ^ ((n >> 3) & 1)) * 4);
    }
    const unsigned u_cstride = 9u * Cout * 32u;

    float4 rh, ru[LU];
    auto ld4 = [&](__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0);
        float4 f;
        unsigned a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
        f.x = __uint_as_float(a0); f.y = __uint_as_float(a1); f.z = __uint_as_float(a2); f.w = __uint_as_float(a3);
        return f;
    };
    auto issue_h = [&](int chunk) { rh = ld4(rsx, h_voff, chunk * 32); };
    auto issue_u = [&](int j, int chunk) { ru[j] = ld4(rsu, u_voff[j], __builtin_amdgcn_readfirstlane(chunk * u_cstride)); };
    auto commit_h = [&](int buf) { *(float4*)&smem[buf * HBUF + h_lds] = rh; };
    auto commit_u = [&](int j, int buf) { *(float4*)&smem[u_lds[j] + buf * UBUF] = ru[j]; };

    // ---- item cursors ----
    int cn, ctx, cty, ch = 0;
    {
        cn = sp0 / per_img;
        const int rem = sp0 - cn * per_img;
        ctx = rem / a.tilesY;
        cty = rem - ctx * a.tilesY;
    }
    auto advance = [&](int& n, int& tx, int& ty, int& c) {
        const int adv = c + 1 == nch ? 1 : 0;
        c = adv ? 0 : c + 1;
        const int ty1 = ty + adv, wy = ty1 == a.tilesY ? 1 : 0;
        ty = wy ? 0 : ty1;
        const int tx1 = tx + wy, wx = tx1 == a.tilesX ? 1 : 0;
        tx = wx ? 0 : tx1;
        n += wx;
    };
    int n1 = cn, tx1 = ctx, ty1 = cty, ch1 = 0;
    advance(n1, tx1, ty1, ch1);
    int n2 = n1, tx2 = tx1, ty2 = ty1, ch2 = ch1;
    advance(n2, tx2, ty2, ch2);

    // ---- fragments: lane (tile m = lane & 15 = low-resolution column, channel pair q = lane >> 4); wave = tile row ----
    const int m = lane & 15, q = lane >> 4;
    auto opaque = [](int x) { asm volatile("" : "+v"(x)); return x; };
    int a_row[3];                      // f32x2 units
#pragma unroll
    for (int r = 0; r < 3; ++r) a_row[r] = opaque((((wv + r) * LWS + m) * WF_KPH + 2 * q) / 2);
    const int b_swz = ((q >> 1) ^ (m >> 3)) * 4 + (q & 1) * 2;
    const int b_u0 = opaque((2 * HBUF + m * 8 + b_swz) / 2), b_u1 = opaque((2 * HBUF + UBUF + m * 8 + b_swz) / 2);
    float mone, pone;                  // -1 / +1 the compiler cannot see through: x + pone * y rounds like x + y and is never paired into v_pk_add_f32
    { float s = -1.f; asm volatile("v_mov_b32 %0, %1" : "=v"(mone) : "v"(s)); }
    { float s = 1.f; asm volatile("v_mov_b32 %0, %1" : "=v"(pone) : "v"(s)); }

    f32x4 acc[9][NBW];
    f32x2 bf[2][NBW];
    f32x2 dcol[2][3];                  // two patch columns in flight: low-resolution rows a, b, c
    float rr[2][3][3];                 // [channel][row 0, 1, 3][patch column] after the row combinations
    float v[2][2][9];                  // [parity][channel][xi]

    auto read_col = [&](int buf, int c) {
        const f32x2* Hc = (const f32x2*)smem + (buf * HBUF + c * WF_KPH) / 2;
#pragma unroll
        for (int r = 0; r < 3; ++r) dcol[c & 1][r] = Hc[a_row[r]];
    };
    // operation o = 0..23: o < 12: patch column o / 4, channel (o % 4) / 2, which = o % 2 (R0 = a - b | R3 = b - c; R1 = b);
    // o >= 12: channel (o - 12) / 6, row s = ((o - 12) % 6) / 2, which = o % 2 (V[s][0] = R[l] - R[m] | V[s][3] = R[m] - R[r]; V[s][1] = R[m])
    auto xform_op = [&](int par, int o) {
        if (o < 12) {
            const int c = o / 4, t = (o % 4) / 2;
            const float da = dcol[c & 1][0][t], db = dcol[c & 1][1][t], dc = dcol[c & 1][2][t];
            if ((o & 1) == 0) { rr[t][0][c] = __builtin_fmaf(mone, db, da); rr[t][1][c] = db; }
            else rr[t][2][c] = __builtin_fmaf(mone, dc, db);
        } else {
            const int k = o - 12, t = k / 6, s = (k % 6) / 2;
            const float l = rr[t][s][0], mm = rr[t][s][1], r = rr[t][s][2];
            if ((o & 1) == 0) { v[par][t][s * 3 + 0] = __builtin_fmaf(mone, mm, l); v[par][t][s * 3 + 1] = mm; }
            else v[par][t][s * 3 + 2] = __builtin_fmaf(mone, r, mm);
        }
    };
    auto xform_slot = [&](int buf, int par, int p) {
#pragma unroll
        for (int o = 0; o < NOPS; ++o) {
            if (o < 12 && o % 4 == 0 && T0 + o - 3 == p) read_col(buf, o / 4);
            if (T0 + o == p) xform_op(par, o);
        }
    };

    // ---- prologue ----
    region_offsets(cn, ctx, cty);
    issue_h(0);
#pragma unroll
    for (int j = 0; j < LU; ++j) issue_u(j, 0);
    commit_h(0);
#pragma unroll
    for (int j = 0; j < LU; ++j) commit_u(j, 0);
    region_offsets(n1, tx1, ty1);
    issue_h(ch1);
    commit_h(1);
    __syncthreads();
#pragma unroll
    for (int o = 0; o < NOPS; ++o) {
        if (o < 12 && o % 4 == 0) read_col(0, o / 4);
        xform_op(0, o);
    }
    __syncthreads();
    region_offsets(n2, tx2, ty2);

    float bvv[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) bvv[nb] = a.bias ? a.bias[co_base + nb * 16 + m] : 0.f;
    const float lo = a.relu ? 0.f : -__builtin_inff();
    int spar = 0;

    auto body = [&](auto PAR, auto FIRST) {
        constexpr int par = decltype(PAR)::value;
        constexpr bool first = decltype(FIRST)::value;
        auto ldb = [&](int xi, int nb) {
            bf[xi & 1][nb] = ((const f32x2*)smem)[(par ? b_u1 : b_u0) + (xi * NCO + nb * 16) * 4];
        };
        auto slot = [&](int p) {
            if (p == 1) issue_h(ch2);
            if (p >= 3 && p < 3 + 2 * LU && (p & 1)) issue_u((p - 3) >> 1, ch1);
            if (p == CP) commit_h(par);
            if (p > CP && p <= CP + LU) commit_u(p - CP - 1, par ^ 1);
            xform_slot(par ^ 1, par ^ 1, p);
        };
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) { ldb(0, nb); __builtin_amdgcn_sched_barrier(0); }
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int xi = 0; xi < 9; ++xi)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    if (k == 0 && xi + 1 < 9) ldb(xi + 1, nb);
                    acc[xi][nb] = MFMA16(v[par][k][xi], k == 0 ? bf[xi & 1][nb].x : bf[xi & 1][nb].y, (first && k == 0) ? zero4 : acc[xi][nb]);
                    slot((xi * 2 + k) * NBW + nb);
                    __builtin_amdgcn_sched_barrier(0);
                }
    };
    // Y = A^T M A over the nine M of a (tile, cout) entry: T0[j] = M0j + M1j, T1[j] = M1j - M3j; Y[a][0] = Ta[0] + Ta[1],
    // Y[a][1] = Ta[1] - Ta[3].  C/D layout (16x16): col = lane & 15 (cout), row = 4 (lane >> 4) + r (tile column).
    auto epilogue = [&]() {
        const int yrow0 = cty * 16 + 2 * wv, xcol0 = ctx * 32 + 8 * q;
        const int H = 2 * h, W = 2 * w;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            float yv[16];      // [r][a][b]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t0[3], t1[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    t0[j] = acc[j][nb][r] + acc[3 + j][nb][r];
                    t1[j] = acc[3 + j][nb][r] - acc[6 + j][nb][r];
                }
                yv[r * 4 + 0] = fmaxf((t0[0] + t0[1]) + bvv[nb], lo);
                yv[r * 4 + 1] = fmaxf((t0[1] - t0[2]) + bvv[nb], lo);
                yv[r * 4 + 2] = fmaxf((t1[0] + t1[1]) + bvv[nb], lo);
                yv[r * 4 + 3] = fmaxf((t1[1] - t1[2]) + bvv[nb], lo);
            }
            const bool second = split && nb >= 2;          // pair mode: N blocks 2, 3 are the second layer's couts 0..31
            const unsigned co = second ? (unsigned)((nb - 2) * 16 + m) : (unsigned)(co_base + nb * 16 + m);
            const __amdgpu_buffer_rsrc_t rso = second ? rsy2 : rsy;
#pragma unroll
            for (int aa = 0; aa < 2; ++aa) {
                const int yy = yrow0 + aa;
                const unsigned base = (((unsigned)cn * H + (unsigned)yy) * W + (unsigned)xcol0) * (unsigned)Cst + co;
                const int voff = (int)sel_u32(yy < H, base * 4u, 0xFFFFFFFFu);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yv[r * 4 + aa * 2 + b]), rso, voff, (2 * r + b) * Cst * 4, 0);
            }
            if (a.stats) {     // uniform: h % 8 == 0 whenever statistics are requested
                float s1, s2;
                lane_stats<16>(yv, s1, s2);
                stat_merge_eq(s1, s2, __shfl_xor(s1, 16, 64), __shfl_xor(s2, 16, 64), 1.f / 32.f);
                stat_merge_eq(s1, s2, __shfl_xor(s1, 32, 64), __shfl_xor(s2, 32, 64), 1.f / 64.f);
                if (lane < 16) {
                    float* R = Rs + spar * (8 * NCO * 2) + (wv * NCO + nb * 16 + lane) * 2;
                    R[0] = s1;
                    R[1] = s2;
                }
            }
        }
        if (a.stats) {
            __syncthreads();
            if (tid < NCO) {
                const float* R = Rs + spar * (8 * NCO * 2) + tid * 2;
                float s1 = R[0], s2 = R[1];              // the eight tile rows of 64 pixels, merged in order
#pragma unroll
                for (int r = 1; r < 8; ++r) stat_merge(s1, s2, 64.f * r, R[r * NCO * 2], R[r * NCO * 2 + 1], 64.f);
                const int t = (cn * a.tilesX + ctx) * a.tilesY + cty;
                float* o = split ? (tid < 32 ? a.stats : a.stats2) + ((size_t)t * 32 + (tid & 31)) * 2
                                 : a.stats + ((size_t)t * Cout + co_base + tid) * 2;
                o[0] = s1;
                o[1] = s2;
            }
            spar ^= 1;
        }
    };

    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    auto shift = [&]() {
        cn = n1; ctx = tx1; cty = ty1; ch = ch1;
        n1 = n2; tx1 = tx2; ty1 = ty2; ch1 = ch2;
        advance(n2, tx2, ty2, ch2);
    };
    for (int reg = 0; reg < my_tiles; ++reg) {     // nch is even
        body(P0{}, std::true_type{});
        __syncthreads();
        shift();
        body(P1{}, std::false_type{});
        __syncthreads();
        for (int c = 2; c < nch; c += 2) {
            shift();
            if (ch2 == 0) region_offsets(n2, tx2, ty2);
            body(P0{}, std::false_type{});
            __syncthreads();
            shift();
            body(P1{}, std::false_type{});
            __syncthreads();
        }
        epilogue();
        shift();
        if (ch2 == 0) region_offsets(n2, tx2, ty2);
    }
}


// =====================================================================================================================
// Weight gradient: dU[xi][co][ci] = sum over tiles of dM[xi][tile][co] V[xi][tile][ci] for the nine xi,  dW = G^T dU G
// =====================================================================================================================
//   dM = A dY A^T from the full-resolution 2 x 2 tile of dY: rows (y0., y0. + y1., y1.) x columns (p0, p0 + p1, p1) - 5 add per
//   16-cout block (the signs of row / column 3 are folded into the final transform); V from the low-resolution 3 x 3 patch of
//   x as in the forward kernel (12 add / sub per 16-channel block; the factor 2 of row / column 1 folded likewise).
//   Workgroup = (32 co x 32 ci) block of all nine dU over a run of 8 x 32 pixel regions (64 tiles); its 8 waves split K
//   (wave = tile row w >> 1, tile columns 8 (w & 1) ..: two k-steps of 4 tiles): 34 VALU per 36 MFMAs, operands built one
//   k-step ahead of their MFMAs (across the region barrier too).  One slab [Cout][9][Cin] per workgroup + fused bias partial.
struct WUpWgArgs {
    const float* x;            // (N, h, w, Cin) low resolution
    const float* dy;           // (N, 2h, 2w, Cout)
    float* part;               // [nsb][Cout][9][Cin]
    float* bias_part;          // [nsb][Cout] or null
    int N, h, w, Cin, Cout;
    int tilesY, tilesX, nsp;
    int n_ci_b, nblk, kt;
    unsigned nbx, nbd;
};

constexpr int WW_DP = 40, WW_XP = 48;      // floats per dY pixel (32 co + 8: tiles 2 pixels apart) / per low-res X pixel (32 ci + 16: 1 pixel apart)

__global__ void __launch_bounds__(512, 1) k_conv_wino_up_wgrad(WUpWgArgs a) {
    constexpr int NT = 512;
    constexpr int DBUF = 256 * WW_DP;                      // 8 x 32 pixels of dY
    constexpr int XR = 6, XW = 18, XPIX = XR * XW;         // low-resolution halo of the region's 4 x 16 tiles
    constexpr int XBUF = XPIX * WW_XP;
    constexpr int XF = XPIX * 8;                           // 864 float4
    constexpr int LX = (XF + NT - 1) / NT;                 // 2
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // layout: dY tiles [2][DBUF], X halos [2][XBUF]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int trow = wv >> 1, thalf = wv & 1;
    const int h = a.h, w = a.w, Cin = a.Cin, Cout = a.Cout;
    const int H = 2 * h, W = 2 * w;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int blk = lb % a.nblk, sblk = lb / a.nblk;
    const int co_base = (blk / a.n_ci_b) * 32, ci_base = (blk % a.n_ci_b) * 32;
    const int sp0 = sblk * a.kt;
    const int my_tiles = min(a.kt, a.nsp - sp0);
    const int per_img = a.tilesY * a.tilesX;
    const bool do_bias = a.bias_part != nullptr && ci_base == 0;

    // ---- loader slots ----
    // dY float4 f = tid + 512 j -> pixel f / 8 = (tid >> 3) + 64 j (two rows of 32 per slot), co quad tid & 7
    const int d_px = tid >> 3;
    const unsigned d_fix = ((unsigned)((d_px >> 5) * W + (d_px & 31)) * Cout + co_base + (tid & 7) * 4) * 4u;
    const unsigned d_jstride = (unsigned)(2 * W) * Cout * 4u;
    const int d_lds = d_px * WW_DP + (tid & 7) * 4;                       // + j * 64 * WW_DP
    // X float4 f -> halo pixel f / 8, ci quad f % 8
    unsigned x_fix[LX];
    int x_lds[LX];
    unsigned x_bits = 0;
#pragma unroll
    for (int j = 0; j < LX; ++j) {
        int f = tid + j * NT;
        if (f >= XF) f -= XF;
        const int hp = f >> 3, hy = hp / XW, hx = hp - hy * XW;
        x_fix[j] = ((unsigned)(hy * w + hx) * Cin + ci_base + (tid & 7) * 4) * 4u;      // against the low-res pixel (y0 - 1, x0 - 1)
        x_lds[j] = 2 * DBUF + hp * WW_XP + (tid & 7) * 4;
        x_bits |= (unsigned)((hy == 0 ? 1 : 0) | (hy == XR - 1 ? 2 : 0) | (hx == 0 ? 4 : 0) | (hx == XW - 1 ? 8 : 0)) << (4 * j);
    }
    float4 rd[4], rx[LX];
    float4 bsum;
    bsum.x = bsum.y = bsum.z = bsum.w = 0.f;
    auto ld4 = [&](__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0);
        float4 f;
        unsigned a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
        f.x = __uint_as_float(a0); f.y = __uint_as_float(a1); f.z = __uint_as_float(a2); f.w = __uint_as_float(a3);
        return f;
    };
    __amdgpu_buffer_rsrc_t rsd, rsx;
    unsigned x_edges = 0;
    auto region_setup = [&](int n, int tx, int ty) {      // region = full-res rows 8 ty .., columns 32 tx ..; low-res rows 4 ty .., columns 16 tx ..
        const long dpix = ((long)n * H + 8 * ty) * W + 32 * tx;
        const long doff = dpix * Cout * 4;
        rsd = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.dy + doff), 0, (int)(unsigned)((long)a.nbd - doff), 0x00020000);
        const long xpix = ((long)n * h + 4 * ty - 1) * w + 16 * tx - 1;   // may lie before the tensor (first region): masked
        const long xoff = xpix * Cin * 4;
        const long xleft = (long)a.nbx - xoff;
        rsx = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.x + xoff), 0, (int)(unsigned)(xleft > 0xFFFFFFF0L ? 0xFFFFFFF0L : xleft), 0x00020000);
        x_edges = ((ty == 0 ? 1u : 0u) | (4 * ty + 4 >= h ? 2u : 0u) | (tx == 0 ? 4u : 0u) | (16 * tx + 16 == w ? 8u : 0u)) * 0x11u;
    };
    auto issue_d = [&](int j) { rd[j] = ld4(rsd, d_fix, j * d_jstride); };
    auto issue_x = [&](int j) {
        const unsigned vo = (x_bits & x_edges & (0xFu << (4 * j))) ? 0xFFFFFFFFu : x_fix[j];
        rx[j] = ld4(rsx, vo, 0);
    };
    float once_v = 1.f;
    auto commit_d = [&](int j, int buf) {
        *(float4*)&smem[d_lds + buf * DBUF + j * 64 * WW_DP] = rd[j];
        if (do_bias) {
            bsum.x = __builtin_fmaf(once_v, rd[j].x, bsum.x); bsum.y = __builtin_fmaf(once_v, rd[j].y, bsum.y);
            bsum.z = __builtin_fmaf(once_v, rd[j].z, bsum.z); bsum.w = __builtin_fmaf(once_v, rd[j].w, bsum.w);
        }
    };
    auto commit_x = [&](int j, int buf) { *(float4*)&smem[x_lds[j] + buf * XBUF] = rx[j]; };

    // ---- fragment addressing: lane (channel idx = lane & 15, tile k = lane >> 4 of the k-step) ----
    // k-step s of the wave: tile row trow, tile columns 8 thalf + 4 s + k
    const int idx = lane & 15, k = lane >> 4;
    auto opaque = [](int x) { asm volatile("" : "+v"(x)); return x; };
    const int a_0 = opaque(((2 * trow) * 32 + 2 * (8 * thalf + k)) * WW_DP + idx);            // dY row 2 trow: + s * 8 px, + b px, + cb * 16
    const int a_1 = opaque(((2 * trow + 1) * 32 + 2 * (8 * thalf + k)) * WW_DP + idx);
    int x_r[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) x_r[r] = opaque(2 * DBUF + ((trow + r) * XW + 8 * thalf + k) * WW_XP + idx);      // + s * 4 px, + column px, + nbk * 16
    float mone, pone;                  // -1 / +1 the compiler cannot see through: x + pone * y rounds like x + y and is never paired into v_pk_add_f32
    { float s = -1.f; asm volatile("v_mov_b32 %0, %1" : "=v"(mone) : "v"(s)); }
    { float s = 1.f; asm volatile("v_mov_b32 %0, %1" : "=v"(pone) : "v"(s)); }

    f32x4 acc[9][2][2];                // [xi][co block][ci block]
#pragma unroll
    for (int xi = 0; xi < 9; ++xi)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[xi][cb][nb][q] = 0.f;
    float dm[2][2][9];                 // [operand set][co block][xi]
    float vv[2][2][9];                 // [operand set][ci block][xi]
    float ra[2][4], rb[2][9], rc[2][6];        // raw dY / X values of two groups in flight; row combinations of an X group

    // Operand build of k-step s from buffer nbuf: four groups, g = 0, 1 co block g (4 reads, 5 operations), g = 2, 3 ci block
    // g - 2 (9 reads, 12 operations).  34 operations at positions 2..35 of a phase, a group's reads ahead of them.
    auto rd_grp = [&](int nbuf, int s, int g, int i) {
        if (g < 2) {
            const int o = nbuf * DBUF + (s * 8 + (i & 1)) * WW_DP + g * 16;
            ra[g & 1][i] = i < 2 ? smem[a_0 + o] : smem[a_1 + o];
        } else {
            const int o = nbuf * XBUF + (s * 4 + i % 3) * WW_XP + (g - 2) * 16;
            rb[g & 1][i] = smem[x_r[i / 3] + o];
        }
    };
    auto op_grp = [&](int set, int g, int i) {
        if (g < 2) {       // rows (y0., y0. + y1., y1.), columns (p0, p0 + p1, p1)
            float* d = dm[set][g];
            const float* y = ra[g & 1];           // y00, y01, y10, y11
            if (i == 0) { d[3] = y[0] + y[2]; d[0] = y[0]; d[2] = y[1]; }
            if (i == 1) { d[5] = y[1] + y[3]; d[6] = y[2]; d[8] = y[3]; }
            if (i == 2) d[1] = y[0] + y[1];
            if (i == 3) d[4] = d[3] + d[5];
            if (i == 4) d[7] = y[2] + y[3];
        } else {           // rows R0 = a - b, R1 = b, R3 = b - c per patch column; then V[s][0] = R[l] - R[m], V[s][1] = R[m], V[s][3] = R[m] - R[r]
            float* o = vv[set][g - 2];
            const float* x = rb[g & 1];           // [row][column]
            float* c = rc[g & 1];                 // [which: 0 = R0, 1 = R3][column]
            if (i < 3) c[i] = __builtin_fmaf(mone, x[3 + i], x[i]);
            else if (i < 6) c[i] = __builtin_fmaf(mone, x[6 + (i - 3)], x[3 + (i - 3)]);
            else {
                const int s = (i - 6) / 2, wh = (i - 6) & 1;
                const float l = s == 0 ? c[0] : s == 1 ? x[3] : c[3], mm = s == 0 ? c[1] : s == 1 ? x[4] : c[4], r = s == 0 ? c[2] : s == 1 ? x[5] : c[5];
                if (wh == 0) { o[s * 3] = l - mm; o[s * 3 + 1] = mm; }
                else o[s * 3 + 2] = mm - r;
            }
        }
    };
    auto build_slot = [&](int nbuf, int s, int set, int p) {
        // positions: 0-1 reads A0 (2 each) | 2-6 ops A0, reads A1 | 7-11 ops A1, reads B0 (2 each, 9) | 12-23 ops B0, reads B1 | 24-35 ops B1
        if (p < 2) { rd_grp(nbuf, s, 0, 2 * p); rd_grp(nbuf, s, 0, 2 * p + 1); }
        else if (p < 7) { op_grp(set, 0, p - 2); if (p - 2 < 4) rd_grp(nbuf, s, 1, p - 2); }
        else if (p < 12) {
            op_grp(set, 1, p - 7);
            if (2 * (p - 7) < 9) rd_grp(nbuf, s, 2, 2 * (p - 7));
            if (2 * (p - 7) + 1 < 9) rd_grp(nbuf, s, 2, 2 * (p - 7) + 1);
        }
        else if (p < 24) { op_grp(set, 2, p - 12); if (p - 12 < 9) rd_grp(nbuf, s, 3, p - 12); }
        else op_grp(set, 3, p - 24);
    };

    // ---- region cursors ----
    int cn = sp0 / per_img, ctx, cty;
    {
        const int rem = sp0 - cn * per_img;
        ctx = rem / a.tilesY;
        cty = rem - ctx * a.tilesY;
    }
    auto next_region = [&](int& n, int& tx, int& ty) {
        const int ty1 = ty + 1, wy = ty1 == a.tilesY ? 1 : 0;
        ty = wy ? 0 : ty1;
        const int tx1 = tx + wy, wx = tx1 == a.tilesX ? 1 : 0;
        tx = wx ? 0 : tx1;
        n += wx;
    };
    int ln = cn, ltx = ctx, lty = cty, lcount = 0;
    auto load_advance = [&]() {
        if (lcount + 1 < my_tiles) { next_region(ln, ltx, lty); ++lcount; once_v = 1.f; } else once_v = 0.f;
        region_setup(ln, ltx, lty);
    };
    if (my_tiles > 0) {
        region_setup(cn, ctx, cty);
#pragma unroll
        for (int j = 0; j < 4; ++j) issue_d(j);
#pragma unroll
        for (int j = 0; j < LX; ++j) issue_x(j);
#pragma unroll
        for (int j = 0; j < 4; ++j) commit_d(j, 0);
#pragma unroll
        for (int j = 0; j < LX; ++j) commit_x(j, 0);
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 36; ++p) build_slot(0, 0, 0, p);
        load_advance();
#pragma unroll
        for (int j = 0; j < 4; ++j) issue_d(j);
    }

    // One phase = the 36 MFMAs of a k-step (operand set SET) + the build of the next k-step's operands (set SET ^ 1) from
    // buffer NBUF.  LOADS: 0 = first k-step of a region (commits dY of the next region, issues and commits its X),
    // 1 = second (issues dY of the region after the next).
    auto phase = [&](auto SET, auto SNEXT, auto NBUF, auto WBUF, auto LOADS) {
        constexpr int set = decltype(SET)::value, sn = decltype(SNEXT)::value, nbuf = decltype(NBUF)::value;
        constexpr int wbuf = decltype(WBUF)::value, loads = decltype(LOADS)::value;
#pragma unroll
        for (int xi = 0; xi < 9; ++xi)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const int p = xi * 4 + cb * 2 + nb;
                    acc[xi][cb][nb] = MFMA16(dm[set][cb][xi], vv[set][nb][xi], acc[xi][cb][nb]);
                    build_slot(nbuf, sn, set ^ 1, p);
                    if (loads == 0 && p >= 1 && p < 1 + 2 * LX && (p & 1)) issue_x((p - 1) >> 1);
                    if (loads == 0 && p >= 8 && p < 12) commit_d(p - 8, wbuf);
                    if (loads == 0 && p >= 34 && p < 34 + LX) commit_x(p - 34, wbuf);
                    if (loads == 1 && p >= 2 && p < 10 && (p & 1) == 0) issue_d((p - 2) >> 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    auto region = [&](auto BUF) {
        constexpr int b = decltype(BUF)::value;
        using B = std::integral_constant<int, b>;
        using NB = std::integral_constant<int, b ^ 1>;
        phase(I0{}, I1{}, B{}, NB{}, I0{});        // k-step 0, builds k-step 1; the next region's data land in the other buffers
        __syncthreads();
        load_advance();
        phase(I1{}, I0{}, NB{}, B{}, I1{});        // k-step 1, builds k-step 0 of the next region; issues dY of the one after
    };
    for (int g = 0; g < my_tiles; g += 2) {
        region(I0{});
        if (g + 1 < my_tiles) region(I1{});
    }

    // ---- fold: the eight waves' partial sums meet in LDS one xi at a time; thread e keeps entries e and e + 512 of the 32 x 32 block ----
    __syncthreads();
    float* red = smem;                     // [8 waves][32 co][32 ci] = 32 KB
    if (do_bias) {                         // threads with equal (tid & 7) hold the same 4 couts
        *(float4*)&red[tid * 4] = bsum;
        __syncthreads();
        if (tid < 32) {
            const int cq = tid >> 2, comp = tid & 3;
            float sum = 0.f;
            for (int i = 0; i < 64; ++i) sum += red[(i * 8 + cq) * 4 + comp];
            a.bias_part[(size_t)sblk * Cout + co_base + tid] = sum;
        }
        __syncthreads();
    }
    int fold_w = wv * 1024 + (4 * (lane >> 4)) * 32 + idx, fold_r = tid;
    asm volatile("" : "+v"(fold_w), "+v"(fold_r));
    float du[2][9];
#pragma unroll
    for (int xi = 0; xi < 9; ++xi) {
        // C/D layout (16x16): col = lane & 15 (ci), row = 4 (lane >> 4) + q (co within the block)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[fold_w + (cb * 16 + q) * 32 + nb * 16] = acc[xi][cb][nb][q];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float t = 0.f;
#pragma unroll
            for (int wq = 0; wq < 8; ++wq) t += red[wq * 1024 + e * 512 + fold_r];
            // what the operands left out: the sign of dM's row / column 3, the factor 2 of V's row / column 1
            const int si = xi / 3, sj = xi % 3;
            const float f = (si == 2 ? -1.f : 1.f) * (sj == 2 ? -1.f : 1.f) * (si == 1 ? 2.f : 1.f) * (sj == 1 ? 2.f : 1.f);
            du[e][xi] = f * t;
        }
        __syncthreads();
    }
    // dW = G^T dU G with the rows / columns (0, 1, 3) of G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int ent = e * 512 + tid, co = co_base + (ent >> 5), ci = ci_base + (ent & 31);
        float tc[3][3];                    // [ky][column index 0, 1, 3]
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float u0 = du[e][j], u1 = du[e][3 + j], u3 = du[e][6 + j];
            tc[0][j] = u0 + 0.5f * u1;
            tc[1][j] = 0.5f * u1;
            tc[2][j] = u3 + 0.5f * u1;
        }
        float* o = a.part + (size_t)sblk * Cout * 9 * Cin + ((size_t)co * 9) * Cin + ci;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            o[(ky * 3 + 0) * Cin] = tc[ky][0] + 0.5f * tc[ky][1];
            o[(ky * 3 + 1) * Cin] = 0.5f * tc[ky][1];
            o[(ky * 3 + 2) * Cin] = tc[ky][2] + 0.5f * tc[ky][1];
        }
    }
}

}  // namespace

// Shapes served: full-resolution width a multiple of 32; Cout % 16 (an even number of 8-channel chunks); Cin % 64.
bool conv_wino_up_dgrad_ok(int Cin, int Cout, int N, int h, int w) {
    if (!g_wup_env || g_wino_mode != 0 || Cin % 64 != 0 || Cout % 16 != 0 || (2 * w) % 32 != 0 || h < 1 || N < 1) return false;
    const long P = (long)N * 4 * h * w;
    return P * (Cin > Cout ? Cin : Cout) * 4 <= 0xFFFFFFE0L;
}
size_t conv_wino_up_ws_floats(int Cin, int Cout) { return (size_t)18 * Cout * Cin; }
// ws: uf [Cin / 8][9][Cout][8] then ud [Cout / 8][9][Cin][8] (needs Cin % 8 == 0 and Cout % 8 == 0)
int conv_wino_up_prepare(const float* w, float* ws, int Cin, int Cout, hipStream_t st, float* wc, float* wd) {
    const long n = (long)Cout * Cin;
    k_wino_up_weights<<<(int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256), 256, 0, st>>>(w, ws, ws + 9L * Cout * Cin, Cout, Cin, wc, wd);
    VQW_LAUNCH_CHECK("wino_up_weights");
    return VQW_OK;
}

template <int NBW, bool ACC>
static int launch_up_dgrad(WUpDgArgs& a, hipStream_t st) {
    constexpr size_t lds = (size_t)(2 * 18 * 34 * WU_KPH + 2 * 9 * 16 * NBW * 8) * sizeof(float);
    static_assert(lds <= 160 * 1024, "buffers do not fit the 160 KB LDS");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_conv_wino_up_dgrad<NBW, ACC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            vqw_set_error("conv_wino_up_dgrad: cannot raise the dynamic LDS limit");
            return VQW_ERR_HIP;
        }
        attr_set = true;
    }
    a.ntn = a.Cin / (16 * NBW);
    int groups = g_wup_max_blocks / a.ntn;
    if (groups < 1) groups = 1;
    const int even = ceil_div(a.nsp, groups);
    a.kt = even < 1 ? 1 : even;
    k_conv_wino_up_dgrad<NBW, ACC><<<ceil_div(a.nsp, a.kt) * a.ntn, 512, lds, st>>>(a);
    VQW_LAUNCH_CHECK("conv_wino_up_dgrad");
    return VQW_OK;
}

// dy (N, 2h, 2w, Cout) -> g_low (N, h, w, Cin); ws as written by conv_wino_up_prepare
int conv_wino_up_dgrad(const float* dy, const float* ws, float* g_low, int N, int h, int w, int Cin, int Cout, hipStream_t st,
                       int accumulate) {
    WUpDgArgs a;
    a.dy = dy; a.u = ws + 9L * Cout * Cin; a.g = g_low;
    a.N = N; a.H = 2 * h; a.W = 2 * w; a.Cin = Cin; a.Cout = Cout;
    a.tilesY = ceil_div(a.H, 16); a.tilesX = a.W / 32; a.nsp = N * a.tilesY * a.tilesX;
    a.nch = Cout / 8;
    const long P = (long)N * a.H * a.W;
    a.nbd = (unsigned)(P * Cout * 4);
    a.nbu = (unsigned)(9L * Cout * Cin * 4);
    a.nbg = (unsigned)(P / 4 * Cin * 4);
    if (accumulate) return Cin % 128 == 0 ? launch_up_dgrad<8, true>(a, st) : launch_up_dgrad<4, true>(a, st);
    if (Cin % 128 == 0) return launch_up_dgrad<8, false>(a, st);
    return launch_up_dgrad<4, false>(a, st);
}

// Forward: Cin % 16 (an even number of chunks), Cout % 64, full-resolution width a multiple of 32
bool conv_wino_up_fwd_ok(int Cin, int Cout, int N, int h, int w) {
    if (!g_wup_env || g_wino_mode != 0 || Cin % 16 != 0 || Cout % 64 != 0 || (2 * w) % 32 != 0 || h < 1 || N < 1) return false;
    const long P = (long)N * 4 * h * w;
    return P * (Cin > Cout ? Cin : Cout) * 4 <= 0xFFFFFFE0L;
}
int conv_wino_up_stat_tiles(int h, int w) { return (h % 8 == 0 && w % 16 == 0) ? (h / 8) * (w / 16) : 0; }
int conv_wino_up_fwd(const float* x_low, const float* ws, const float* bias, float* y, int N, int h, int w, int Cin, int Cout, int relu,
                     hipStream_t st, float* stats, float* y2, float* stats2) {
    constexpr size_t lds = (size_t)(2 * 10 * 18 * WF_KPH + 2 * 9 * 64 * 8 + 2 * 8 * 64 * 2) * sizeof(float);
    if (y2 && Cout != 64) { vqw_set_error("conv_wino_up_fwd: pair mode needs two 32-cout layers (Cout = 64)"); return VQW_ERR_ARG; }
    WUpFwArgs a;
    a.x = x_low; a.u = ws; a.bias = bias; a.y = y; a.y2 = y2; a.stats2 = stats2;
    a.N = N; a.h = h; a.w = w; a.Cin = Cin; a.Cout = Cout;
    a.tilesY = ceil_div(h, 8); a.tilesX = w / 16; a.nsp = N * a.tilesY * a.tilesX;
    a.ntn = Cout / 64; a.nch = Cin / 8;
    a.relu = relu;
    a.stats = stats;
    const long Pl = (long)N * h * w;
    a.nbx = (unsigned)(Pl * Cin * 4);
    a.nbu = (unsigned)(9L * Cout * Cin * 4);
    a.nby = (unsigned)(4 * Pl * (y2 ? 32 : Cout) * 4);
    int groups = g_wup_max_blocks / a.ntn;
    if (groups < 1) groups = 1;
    const int even = ceil_div(a.nsp, groups);
    a.kt = even < 1 ? 1 : even;
    k_conv_wino_up_fwd<<<ceil_div(a.nsp, a.kt) * a.ntn, 512, lds, st>>>(a);
    VQW_LAUNCH_CHECK("conv_wino_up_fwd");
    return VQW_OK;
}

// Weight gradient: Cin % 32, Cout % 32, low-resolution maps of whole 4 x 16 regions
bool conv_wino_up_wgrad_ok(int Cin, int Cout, int N, int h, int w) {
    if (!g_wup_env || g_wino_mode != 0 || Cin % 32 != 0 || Cout % 32 != 0 || w % 16 != 0 || h % 4 != 0 || N < 1) return false;
    const long P = (long)N * 4 * h * w;
    return P * (Cin > Cout ? Cin : Cout) * 4 <= 0xFFFFFFE0L;
}
static int wup_wgrad_blocks(int Cin, int Cout, int N, int h, int w, int* kt_out) {
    const int nblk = (Cout / 32) * (Cin / 32);
    const int nsp = N * (h / 4) * (w / 16);
    int nsb = g_wup_max_blocks_wg / nblk;
    if (nsb > nsp) nsb = nsp;
    if (nsb < 1) nsb = 1;
    const int kt = ceil_div(nsp, nsb);
    if (kt_out) *kt_out = kt;
    return ceil_div(nsp, kt);
}
size_t conv_wino_up_wgrad_ws_floats(int Cin, int Cout, int N, int h, int w) {
    return (size_t)wup_wgrad_blocks(Cin, Cout, N, h, w, nullptr) * ((size_t)Cout * 9 * Cin + Cout);
}
int conv_wino_up_wgrad(const float* x_low, const float* dy, float* dw, float* dbias, float* ws, int N, int h, int w, int Cin, int Cout,
                       int acc, hipStream_t st) {
    constexpr size_t lds = (size_t)(2 * 256 * WW_DP + 2 * 6 * 18 * WW_XP) * sizeof(float);
    static_assert(lds <= 160 * 1024 && lds >= 36 * 1024, "wgrad tiles / fold area");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_conv_wino_up_wgrad, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            vqw_set_error("conv_wino_up_wgrad: cannot raise the dynamic LDS limit");
            return VQW_ERR_HIP;
        }
        attr_set = true;
    }
    int kt = 1;
    const int nsb = wup_wgrad_blocks(Cin, Cout, N, h, w, &kt);
    const long nout = (long)Cout * 9 * Cin;
    WUpWgArgs a;
    a.x = x_low; a.dy = dy; a.part = ws; a.bias_part = dbias ? ws + (size_t)nsb * nout : nullptr;
    a.N = N; a.h = h; a.w = w; a.Cin = Cin; a.Cout = Cout;
    a.tilesY = h / 4; a.tilesX = w / 16; a.nsp = N * a.tilesY * a.tilesX;
    a.n_ci_b = Cin / 32; a.nblk = (Cout / 32) * a.n_ci_b; a.kt = kt;
    const long Pl = (long)N * h * w;
    a.nbx = (unsigned)(Pl * Cin * 4);
    a.nbd = (unsigned)(4 * Pl * Cout * 4);
    k_conv_wino_up_wgrad<<<a.nblk * nsb, 512, lds, st>>>(a);
    VQW_LAUNCH_CHECK("conv_wino_up_wgrad");
    if (dbias) {
        int rc = reduce_rows(a.bias_part, dbias, Cout, nsb, st, acc);
        if (rc) return rc;
    }
    return reduce_rows(ws, dw, nout, nsb, st, acc);
}
