// 3x3 stride-1 convolution (forward and dgrad) in Winograd F(2x2, 3x3) form on the fp32 matrix cores, NHWC.
//
//   Y = A^T [ (G g G^T) . (B^T d B) ] A      per 4x4 input patch d -> 2x2 outputs, 16 products instead of 36
//
// With channels: M[xi][tile][co] = sum_ci V[xi][tile][ci] U[xi][co][ci] for the 16 positions xi of the transformed patch,
// i.e. 16 GEMMs that share nothing but their shapes: 4/9 of the direct form's matrix work.  U = G g G^T is computed once
// per weight (k_wino_weights, in double, rounded once) and cached by the caller; V = B^T d B is never stored:
//
//   * a workgroup (8 waves) owns a 16 x 32 pixel output region x 32 couts and walks the input channels in chunks of 8.
//     Wave w takes tile row w (output rows 2w, 2w+1 = 16 Winograd tiles) on v_mfma_f32_16x16x4_f32: M = 16 tiles,
//     N = 16 couts (two N blocks), K = 4 channel pairs -> lane (tile m, pair q) holds channels 2q, 2q+1.
//   * the raw (16+2) x 34 pixel halo of the chunk sits in LDS (10 floats per pixel: ds_read_b64 of 16 tiles x 4 pairs is
//     conflict-free).  A lane reads the 4 x 4 patch of its tile at its two channels (16 ds_read_b64), transforms it in
//     registers (32 add / sub per channel) and then owns the A operands of all 16 xi for this chunk: 64 MFMAs per wave
//     follow from 16 LDS reads + 64 VALU instructions on the A side.  The B operand U[xi][co][ci] is streamed through LDS
//     per chunk ([16][32 couts][12]: conflict-free ds_read_b64).
//   * accumulators: 16 xi x 2 N blocks x 4 = 128 registers; the inverse transform A^T M A is lane-local (a lane holds
//     all 16 xi of its (tile, cout) entries), followed by bias / ReLU / the statistics partials and the stores.
//   * pipeline: an item = (region, chunk).  The halo buffers form a ring of three (items i, i+1, i+2), the U chunks a pair:
//     during item i the halo of item i+2 and the U chunk of item i+1 are fetched into registers behind the first MFMAs
//     and written to LDS behind later ones, and the patch of item i+1 (published by the previous barrier) is read and
//     transformed two VALU instructions per MFMA in the second half - so an item starts with its A operands in registers
//     and needs one barrier.  Measured steps (profiles/r02_*): plain ds_read_b64 instead of the merged ds_read2_b64 took
//     the large layers from 214 to 232 effective TFLOP/s, the ring + hidden transform from 190 to 214.
//
// Rounding: the products differ from the direct form's (sums of four inputs times sums of weights), the result agrees
// with it to a few fp32 ulps of the accumulated magnitude - the same class of difference as a changed summation order.
#include "common.h"
#include "conv_common.h"
#include "mfma_util.h"
#include <cstdlib>
#include <type_traits>

int g_wino_mode = 0;     // 0 auto, 1 off

namespace {

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
static const int g_wino_env = env_int("VQW_WINOGRAD", 1);
static const int g_max_blocks = []{ int v = env_int("VQW_CONV_MAX_BLOCKS", 256); return v < 8 ? 8 : (v > 256 ? 256 : v); }();

// U[xi = i*4 + j][co][ci] = sum_{ky,kx} G[i][ky] g[co][ky][kx][ci] G[j][kx],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
// chunked = 1: [ci / 8][xi][co][ci % 8] - what the kernels stream: the couts of a (chunk, xi) lie in a row, 32 bytes each
// transposed = 1: w is the LAYER's OHWI weight [Cin][3][3][Cout] (its couts = this transform's "input" channels) and U is that
// of the layer's input-gradient convolution, g[ky][kx] = w[ci][2-ky][2-kx][co] - what vqw_pack_dgrad_weights + this kernel give,
// without the packed copy
__global__ void k_wino_weights(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin, int chunked, int transposed) {
    const long n = (long)Cout * Cin;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int co = (int)(e / Cin), ci = (int)(e % Cin);
        double g[3][3], t[4][3];
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx)
                g[ky][kx] = transposed ? (double)w[(((long)ci * 3 + (2 - ky)) * 3 + (2 - kx)) * Cout + co]
                                       : (double)w[(((long)co * 3 + ky) * 3 + kx) * Cin + ci];
        for (int kx = 0; kx < 3; ++kx) {
            t[0][kx] = g[0][kx];
            t[1][kx] = 0.5 * (g[0][kx] + g[1][kx] + g[2][kx]);
            t[2][kx] = 0.5 * (g[0][kx] - g[1][kx] + g[2][kx]);
            t[3][kx] = g[2][kx];
        }
        for (int i = 0; i < 4; ++i) {
            const double r0 = t[i][0], r1 = 0.5 * (t[i][0] + t[i][1] + t[i][2]), r2 = 0.5 * (t[i][0] - t[i][1] + t[i][2]), r3 = t[i][2];
            const double rr[4] = {r0, r1, r2, r3};
            for (int j = 0; j < 4; ++j) {
                const long xi = i * 4 + j;
                const long at = chunked ? ((((long)(ci >> 3) * 16 + xi) * Cout + co) * 8 + (ci & 7)) : ((xi * Cout + co) * Cin + ci);
                u[at] = (float)rr[j];
            }
        }
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct WinoArgs {
    const float* x;
    const float* u;
    const float* bias;
    float* y;
    int N, H, W, Cin, Cout;
    int tilesY, tilesX, nsp;   // regions per image column / row, total
    int ntn, nch;              // cout tiles, channel chunks
    int kt;                    // consecutive regions per workgroup
    int relu;
    unsigned nbx, nbu, nby;
    float* stats;              // optional [N][tilesY*tilesX][Cout][2]: per-region (sum, M2 about the region mean)
};

constexpr int WN_KC = 8;                  // channels per chunk
constexpr int WN_KPH = 10, WN_KPU = 12;   // floats per halo pixel / per U row in LDS
// Region geometry, RW = region width: 16 rows x 32 columns (halo 18 x 34), or - for maps whose width is a multiple of 16
// only (the 16 x 16 level) - 32 rows x 16 columns (halo 34 x 18, LDS row stride 24 pixels: the two tile rows a wave then
// covers land on complementary banks).  Either way 128 tiles per region and 612 halo pixels per chunk.
template <int RW> struct WinoGeo {
    static constexpr int TR = RW == 32 ? 16 : 32;          // output rows per region
    static constexpr int HR = TR + 2, HWV = RW + 2;        // halo rows / valid halo columns
    static constexpr int HWS = RW == 32 ? 34 : 24;         // LDS row stride in pixels
    static constexpr int HBUF = HR * HWS * WN_KPH;         // floats per halo buffer
};
#ifndef WN_CP
#define WN_CP 22                         // MFMA position of the first LDS commit of the prefetched item
#endif

template <int RW>
__global__ void __launch_bounds__(512, 1) k_conv_wino(WinoArgs a) {
    using G = WinoGeo<RW>;
    constexpr int NT = 512;
    constexpr int WN_TR = G::TR, WN_HW = G::HWS;
    constexpr int HPIX = G::HR * G::HWV;              // 612 halo pixels
    constexpr int HF = HPIX * 2;                      // float4 per halo chunk
    constexpr int LH = (HF + NT - 1) / NT;            // 3
    constexpr int WF = 16 * 32 * 2;                   // float4 per U chunk
    constexpr int LW = WF / NT;                       // 2
    constexpr int HBUF = G::HBUF;                     // floats
    constexpr int UBUF = 16 * 32 * WN_KPU;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Hs = smem;                     // [3][HBUF]: ring of the items i, i+1, i+2
    float* Us = smem + 3 * HBUF;          // [2][UBUF]
    float* Rs = Us + 2 * UBUF;            // [2][8 waves][32 couts][2] statistics of the waves' tile rows
    float* Ds = Rs + 2 * 8 * 32 * 2;      // [512][4] parking space of the loader slots past the end of the halo

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(a.x, a.nbx), rsu = make_rsrc(a.u, a.nbu), rsy = make_rsrc(a.y, a.nby);

    const int ntn = a.ntn, nch = a.nch;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lb % ntn;
    const int sp0 = (lb / ntn) * a.kt;
    const int co_base = tile_n * 32;
    const int my_tiles = min(a.kt, a.nsp - sp0);
    const int nitems = my_tiles * nch;
    const int per_img = a.tilesY * a.tilesX;
    if (nitems <= 0) return;           // uniform per workgroup

    // loader slots (fixed for the whole kernel): halo float4 f -> pixel f / 2, channel quad f % 2
    int h_lds[LH], h_st[LH];
    short h_y[LH], h_x[LH];
    bool h_ok[LH];
    const unsigned h_c = (unsigned)(tid & 1) * 4u;
#pragma unroll
    for (int j = 0; j < LH; ++j) {
        const int f = tid + j * NT;
        const bool ok = f < HF;
        const int hp = ok ? f >> 1 : 0;
        h_ok[j] = ok;
        h_lds[j] = ok ? ((hp / G::HWV) * WN_HW + hp % G::HWV) * WN_KPH + (int)h_c : (int)(Ds - Hs) + tid * 4;      // offset from Hs in buffer 0
        h_st[j] = ok ? HBUF : 0;                                                 // ... + buf * h_st in buffer buf
        h_y[j] = (short)(hp / G::HWV);
        h_x[j] = (short)(hp % G::HWV);
    }
    unsigned u_off[LW];
    int u_lds[LW];
#pragma unroll
    for (int j = 0; j < LW; ++j) {
        const int f = tid + j * NT;
        const int row = f >> 1, c4 = f & 1;          // row = xi * 32 + n
        const int xi = row >> 5, n = row & 31;
        const int co = co_base + n;
        u_lds[j] = row * WN_KPU + c4 * 4;
        u_off[j] = co < Cout ? (((unsigned)xi * Cout + co) * 8u + c4 * 4) * 4u : 0xFFFFFFFFu;       // chunked layout [Cin / 8][xi][Cout][8]
    }

    float4 rh[LH], ru[LW];
    unsigned i_img = 0, i_cc4 = 0;
    int i_y0 = 0, i_x0 = 0;
    auto issue_setup = [&](int n, int tx, int ty, int ch) {
        i_img = (unsigned)n * H * W;
        i_y0 = ty * WN_TR - 1;
        i_x0 = tx * RW - 1;
        i_cc4 = (unsigned)(ch * WN_KC) * 4u;
    };
    auto issue_h = [&](int j) {
        const int yy = i_y0 + h_y[j], xx = i_x0 + h_x[j];
        const bool ok = h_ok[j] & ((unsigned)yy < (unsigned)H) & ((unsigned)xx < (unsigned)W);
        const unsigned pix = i_img + (unsigned)(yy * W + xx);
        rh[j] = buf_ld4(rsx, sel_u32(ok, pix * (unsigned)Cin * 4u + i_cc4 + h_c * 4u, a.nbx));
    };
    auto issue_u = [&](int j) { ru[j] = buf_ld4(rsu, sel_u32(u_off[j] == 0xFFFFFFFFu, a.nbu, u_off[j] + (i_cc4 >> 5) * (16u * (unsigned)Cout * 32u))); };
    auto commit_h = [&](int j, int buf) {       // a halo pixel is 40 bytes: two 8-byte-aligned halves
        float* p = &Hs[(h_st[j] ? buf : 0) + h_lds[j]];
        f32x2 lo, hi;
        lo.x = rh[j].x; lo.y = rh[j].y; hi.x = rh[j].z; hi.y = rh[j].w;
        *(f32x2*)p = lo;
        *(f32x2*)(p + 2) = hi;
    };
    auto commit_u = [&](int j, int buf) { *(float4*)&Us[buf * UBUF + u_lds[j]] = ru[j]; };

    // item cursors: (image, strip, region row, chunk) of items i, i+1, i+2
    int cn, ctx, cty, ch = 0;
    {
        cn = sp0 / per_img;
        const int rem = sp0 - cn * per_img;
        ctx = rem / a.tilesY;
        cty = rem - ctx * a.tilesY;
    }
    auto advance = [&](int& n, int& tx, int& ty, int& c) {     // the item after (n, tx, ty, c)
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

    // prologue: halo of items 0 and 1 into halo buffers 0 and 1, U chunk of item 0 into U buffer 0.  (Items past the
    // workgroup's share read another region or nothing - out-of-range offsets return 0 - and are never consumed.)
    issue_setup(cn, ctx, cty, 0);
#pragma unroll
    for (int j = 0; j < LH; ++j) issue_h(j);
#pragma unroll
    for (int j = 0; j < LW; ++j) issue_u(j);
#pragma unroll
    for (int j = 0; j < LH; ++j) commit_h(j, 0);
#pragma unroll
    for (int j = 0; j < LW; ++j) commit_u(j, 0);
    issue_setup(n1, tx1, ty1, ch1);
#pragma unroll
    for (int j = 0; j < LH; ++j) issue_h(j);
#pragma unroll
    for (int j = 0; j < LH; ++j) commit_h(j, HBUF);
    __syncthreads();

    // fragment bases: lane (tile m = lane & 15, channel pair q = lane >> 4)
    const int m = lane & 15, q = lane >> 4;
    // tile of the lane: RW = 32: tile row wv, column m; RW = 16: tile row 2 wv + (m >> 3), column m & 7
    const int a_base = RW == 32 ? ((2 * wv) * WN_HW + 2 * m) * WN_KPH + 2 * q
                                : ((2 * (2 * wv + (m >> 3))) * WN_HW + 2 * (m & 7)) * WN_KPH + 2 * q;      // patch (r, c) adds (r * HWS + c) * KPH
    const int b_base = m * WN_KPU + 2 * q;                               // (xi, nb) adds (xi * 32 + nb * 16) * KPU

    float bvv[2];
    unsigned co_off[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int co = co_base + nb * 16 + m;
        bvv[nb] = (a.bias && co < Cout) ? a.bias[co] : 0.f;
        co_off[nb] = co < Cout ? (unsigned)co : 0xFFFFFFFFu;
    }
    const float lo = a.relu ? 0.f : -__builtin_inff();

    f32x4 acc[16][2];
    int spar = 0;
    int fold_t = -1;
    auto fold_stats = [&]() {          // after the barrier that follows a finished region
        if (fold_t < 0) return;        // uniform
        if (tid < 32 && co_base + tid < Cout) {
            const float* R = Rs + spar * (8 * 32 * 2) + tid * 2;
            float s1 = R[0], s2 = R[1];          // tile rows of 64 pixels, merged in row order
#pragma unroll
            for (int r = 1; r < 8; ++r) stat_merge(s1, s2, (float)(64 * r), R[r * 64], R[r * 64 + 1], 64.f);
            float* o = a.stats + ((size_t)fold_t * Cout + co_base + tid) * 2;
            o[0] = s1;
            o[1] = s2;
        }
        fold_t = -1;
        spar ^= 1;
    };

    // V = B^T d B of the lane's tile at its two channels.  The 64 add / sub of an item's transform are single VALU
    // instructions placed two per MFMA in the second half of the PREVIOUS item, column by column as the patch columns arrive
    // from LDS, then row by row.  They are written as fma with +1 / -1 held in registers the compiler cannot see through:
    // x + 1 * y and x - 1 * y round exactly like the add / sub they stand for, the vector combiner cannot pair them into
    // v_pk_add_f32 over the two channels of a ds_read_b64 (which holds the vector issue port 2-3 times as long beside MFMAs),
    // and - unlike the inline-asm v_add / v_sub this kernel used before - they are ordinary instructions to the compiler's
    // hazard recogniser (a VALU write inside inline asm followed closely by the MFMA that reads it gave a stale operand in
    // the weight-gradient kernel below).
    float pone, mone;
    { float c = 1.f; asm volatile("v_mov_b32 %0, %1" : "=v"(pone) : "v"(c)); }
    { float c = -1.f; asm volatile("v_mov_b32 %0, %1" : "=v"(mone) : "v"(c)); }
    auto fadd = [&](float x, float y) { return __builtin_fmaf(pone, y, x); };
    auto fsub = [&](float x, float y) { return __builtin_fmaf(mone, y, x); };
    f32x2 dcol[2][4];                  // two patch columns in flight
    float e[2][4][4];                  // [channel][row][column] after the column pass
    float v[2][2][16];                 // [parity of the item][channel of the pair][xi]
    // LDS fragment reads must stay plain ds_read_b64 (32-lane groups over 64 banks: conflict-free with these strides, 2 LDS
    // cycles).  Two reads off one base register get merged into ds_read2_b64, which is served in 16-lane groups over 32
    // banks: 8 cycles and 2-way conflicts on these layouts - the LDS, not the matrix cores, then paces the item.  Each read
    // of a pair / quad therefore goes through a base register of its own that the compiler cannot relate to the others.
    auto opaque = [](int x) { asm volatile("" : "+v"(x)); return x; };
    const int a_row0 = opaque(a_base), a_row1 = opaque(a_base + WN_HW * WN_KPH), a_row2 = opaque(a_base + 2 * WN_HW * WN_KPH),
              a_row3 = opaque(a_base + 3 * WN_HW * WN_KPH);
    const int b_nb0 = opaque(b_base), b_nb1 = opaque(b_base + 16 * WN_KPU);
    auto read_col = [&](int hbuf, int c) {
        const float* Hc = Hs + hbuf + c * WN_KPH;
        dcol[c & 1][0] = *(const f32x2*)&Hc[a_row0];
        dcol[c & 1][1] = *(const f32x2*)&Hc[a_row1];
        dcol[c & 1][2] = *(const f32x2*)&Hc[a_row2];
        dcol[c & 1][3] = *(const f32x2*)&Hc[a_row3];
    };
    auto col_op = [&](int c, int k) {          // k = 0..7: channel k / 4, output row k % 4 of column c
        const int t = k >> 2, rr = k & 3;
        const float d0 = dcol[c & 1][0][t], d1 = dcol[c & 1][1][t], d2 = dcol[c & 1][2][t], d3 = dcol[c & 1][3][t];
        e[t][rr][c] = rr == 0 ? fsub(d0, d2) : rr == 1 ? fadd(d1, d2) : rr == 2 ? fsub(d2, d1) : fsub(d1, d3);
    };
    auto row_op = [&](int par, int r, int k) { // k = 0..7: channel k / 4, output column k % 4 of row r
        const int t = k >> 2, cc = k & 3;
        const float e0 = e[t][r][0], e1 = e[t][r][1], e2 = e[t][r][2], e3 = e[t][r][3];
        v[par][t][r * 4 + cc] = cc == 0 ? fsub(e0, e2) : cc == 1 ? fadd(e1, e2) : cc == 2 ? fsub(e2, e1) : fsub(e1, e3);
    };
    {   // item 0: nothing to hide behind yet
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            read_col(0, c);
#pragma unroll
            for (int k = 0; k < 8; ++k) col_op(c, k);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) row_op(0, r, k);
    }

    int hA = 0, hB = HBUF, hC = 2 * HBUF;      // halo buffers (float offsets) of items i, i+1, i+2
    int ucur = 0;

    // One item; PAR = parity of the item (selects which half of v holds its operands; the other half receives the next
    // item's).  Everything but the MFMAs is cut into slots behind single MFMAs (sched_barrier keeps them there).
    auto body = [&](auto PAR, auto FIRST) {
        constexpr int par = decltype(PAR)::value;
        constexpr bool first = decltype(FIRST)::value;      // first chunk of a region: the accumulators start from 0
        if (!first && ch == 0) {
#pragma unroll
            for (int xi = 0; xi < 16; ++xi)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[xi][nb][r] = 0.f;
        }
        const float* Uc = Us + ucur * UBUF;
        f32x2 bf[3][2];                // B fragments run two xi ahead of the MFMAs that consume them
        auto ldb = [&](int xi, int s) {
            bf[s][0] = *(const f32x2*)&Uc[b_nb0 + xi * 32 * WN_KPU];
            bf[s][1] = *(const f32x2*)&Uc[b_nb1 + xi * 32 * WN_KPU];
        };
        auto slot = [&](int p) {       // p = 0..63: MFMA position (compile-time after unrolling)
            // loads (address arithmetic included) behind the first MFMAs: halo of item i+2, U chunk of item i+1
#ifndef WN_EXP_NO_LOADS
            if (p == 0) issue_setup(n2, tx2, ty2, ch2);
            if (p >= 1 && p < 1 + LH) issue_h(p - 1);
            if (p == 1 + LH) i_cc4 = (unsigned)(ch1 * WN_KC) * 4u;
            if (p >= 2 + LH && p < 2 + LH + LW) issue_u(p - 2 - LH);
            if (p >= WN_CP && p < WN_CP + LH) commit_h(p - WN_CP, hC);
            if (p >= WN_CP + 4 && p < WN_CP + 4 + LW) commit_u(p - WN_CP - 4, ucur ^ 1);
#endif
#ifdef WN_EXP_NO_XFORM
            if (true) return;
#endif
            if (p == 29) read_col(hB, 0);
            if (p >= 32 && p < 48) {   // column pass: column c = (p - 32) / 4, two operations per position
                const int c = (p - 32) >> 2, k = ((p - 32) & 3) * 2;
                if (k == 0 && c < 3) read_col(hB, c + 1);
                col_op(c, k);
                col_op(c, k + 1);
            }
            if (p >= 48) {             // row pass: row r = (p - 48) / 4
                const int r = (p - 48) >> 2, k = ((p - 48) & 3) * 2;
                row_op(par ^ 1, r, k);
                row_op(par ^ 1, r, k + 1);
            }
        };
        ldb(0, 0);
        ldb(1, 1);
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            const int s = xi % 3;
#ifndef WN_EXP_NO_BREAD
            if (xi + 2 < 16) ldb(xi + 2, (xi + 2) % 3);
#else
            if (xi == 0) ldb(2, 2);
#endif
            acc[xi][0] = MFMA16(v[par][0][xi], bf[s][0].x, first ? zero4 : acc[xi][0]);
            slot(4 * xi); __builtin_amdgcn_sched_barrier(0);
            acc[xi][1] = MFMA16(v[par][0][xi], bf[s][1].x, first ? zero4 : acc[xi][1]);
            slot(4 * xi + 1); __builtin_amdgcn_sched_barrier(0);
            acc[xi][0] = MFMA16(v[par][1][xi], bf[s][0].y, acc[xi][0]);
            slot(4 * xi + 2); __builtin_amdgcn_sched_barrier(0);
            acc[xi][1] = MFMA16(v[par][1][xi], bf[s][1].y, acc[xi][1]);
            slot(4 * xi + 3); __builtin_amdgcn_sched_barrier(0);
        }

#ifdef WN_EXP_NO_EPI
        if (ch == nch - 1 && acc[3][1][2] == 123.456f) {
#else
        if (ch == nch - 1) {
#endif
            // Y = A^T M A per (tile, cout) entry: lane-local over the 16 xi; then bias / ReLU, statistics, stores
            // C/D rows of a lane = tiles 4 q + r: RW = 32: tile row wv, columns 4 q + r; RW = 16: tile row 2 wv + (q >> 1),
            // columns 4 (q & 1) + r
            const int yrow0 = cty * WN_TR + (RW == 32 ? 2 * wv : 2 * (2 * wv + (q >> 1)));
            const int xcol0 = ctx * RW + (RW == 32 ? 8 * q : 8 * (q & 1));
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                float yv[16];      // [r][a][b]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float sj[2][4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float m0 = acc[0 * 4 + j][nb][r], m1 = acc[1 * 4 + j][nb][r], m2 = acc[2 * 4 + j][nb][r], m3 = acc[3 * 4 + j][nb][r];
                        sj[0][j] = (m0 + m1) + m2;
                        sj[1][j] = (m1 - m2) - m3;
                    }
#pragma unroll
                    for (int aa = 0; aa < 2; ++aa) {
                        yv[r * 4 + aa * 2 + 0] = fmaxf(((sj[aa][0] + sj[aa][1]) + sj[aa][2]) + bvv[nb], lo);
                        yv[r * 4 + aa * 2 + 1] = fmaxf(((sj[aa][1] - sj[aa][2]) - sj[aa][3]) + bvv[nb], lo);
                    }
                }
                // C/D layout (16x16): col = lane & 15 (cout), row = 4 (lane >> 4) + r (tile): pixels (2 wv + a, 2 tile + b)
#pragma unroll
                for (int aa = 0; aa < 2; ++aa) {
                    const int yy = yrow0 + aa;
                    const bool ok = (co_off[nb] != 0xFFFFFFFFu) & (yy < H);
                    const unsigned base = (((unsigned)cn * H + (unsigned)yy) * W + (unsigned)xcol0) * (unsigned)Cout + co_off[nb];
                    const int voff = (int)sel_u32(ok, base * 4u, a.nby);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
#ifndef WN_EXP_NO_STORE          // timing-only A/B builds (tools/wino_ab.sh): results are wrong by construction
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yv[r * 4 + aa * 2 + b]), rsy, voff, (2 * r + b) * Cout * 4, 0);
#else
                            if (yv[r * 4 + aa * 2 + b] == 123.456f) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yv[r * 4 + aa * 2 + b]), rsy, voff, (2 * r + b) * Cout * 4, 0);
#endif
                }
                if (a.stats) {     // uniform: H % 16 == 0 whenever statistics are requested
                    float t1, t2;
                    lane_stats<16>(yv, t1, t2);
                    stat_merge_eq(t1, t2, __shfl_xor(t1, 16, 64), __shfl_xor(t2, 16, 64), 1.f / 32.f);
                    stat_merge_eq(t1, t2, __shfl_xor(t1, 32, 64), __shfl_xor(t2, 32, 64), 1.f / 64.f);
                    if (lane < 16) {
                        float* R = Rs + spar * (8 * 32 * 2);
                        R[(wv * 32 + nb * 16 + lane) * 2] = t1;
                        R[(wv * 32 + nb * 16 + lane) * 2 + 1] = t2;
                    }
                }
            }
            if (a.stats) fold_t = (cn * a.tilesX + ctx) * a.tilesY + cty;
        }
        cn = n1; ctx = tx1; cty = ty1; ch = ch1;
        n1 = n2; tx1 = tx2; ty1 = ty2; ch1 = ch2;
        advance(n2, tx2, ty2, ch2);
#ifndef WN_EXP_NO_BARRIER
        __syncthreads();               // publishes the halo of item i+2 and the U chunk of item i+1
#endif
        const int t = hA; hA = hB; hB = hC; hC = t;
        ucur ^= 1;
        fold_stats();
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    for (int item = 0; item < nitems; item += 2) {      // every branch is uniform per workgroup
        body(P0{}, std::false_type{});
        if (item + 1 < nitems) body(P1{}, std::false_type{});
    }
}

}  // namespace

// 3x3, dilation 1, one source at full resolution, whole 32-pixel column strips, channels in chunks of 8
bool conv_wino_ok(int Cin, int Cout, int N, int H, int W) {
    if (g_wino_mode != 0 || !g_wino_env) return false;
    if (Cin % 8 != 0 || Cin < 16 || Cout < 32 || Cout % 4 != 0 || W % 16 != 0 || H < 2 || N < 1) return false;
    // (a batch beyond the 32-bit descriptor range runs as image groups: one image must fit)
    return (long)H * W * (Cin > Cout ? Cin : Cout) * 4 <= 0xFFFFFFE0L && 16L * Cout * Cin * 4 <= 0xFFFFFFE0L;
}
size_t conv_wino_ws_floats(int Cin, int Cout) { return (size_t)16 * Cout * Cin; }
int conv_wino_prepare(const float* w, float* u, int Cin, int Cout, hipStream_t st, int transposed) {
    const long n = (long)Cout * Cin;
    k_wino_weights<<<(int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256), 256, 0, st>>>(w, u, Cout, Cin, 1, transposed);      // every kernel streams the chunked layout
    VQW_LAUNCH_CHECK("wino_weights");
    return VQW_OK;
}
int conv_wino_stat_tiles(int Cin, int Cout, int H, int W) {
    if (conv_wino64_ok(Cin, Cout, W)) return conv_wino64_stat_tiles(Cin, Cout, H, W);
    const int rw = W % 32 == 0 ? 32 : 16, tr = rw == 32 ? 16 : 32;
    return H % tr == 0 ? (H / tr) * (W / rw) : 0;
}

int conv_wino_fwd(const float* x, const float* u, const float* bias, float* y, int N, int H, int W, int Cin, int Cout, int relu,
                  hipStream_t st, float* stats, const float* mask, int accumulate, const float* in_mr, int in_relu) {
    if ((mask || accumulate || in_mr) && !conv_wino64_ok(Cin, Cout, W)) {
        vqw_set_error("conv_wino_fwd: the masked / accumulating forms are served by the 64-cout kernel only");
        return VQW_ERR_ARG;
    }
    {   // images per launch such that no tensor exceeds the 32-bit descriptor range (as vqw_conv2d_fwd does)
        const long per_image = (long)H * W * (Cin > Cout ? Cin : Cout) * 4;
        const long g = 0xFFFFFFE0L / per_image;
        if (g < N) {
            const int parts = stats ? conv_wino_stat_tiles(Cin, Cout, H, W) : 0;
            for (int n0 = 0; n0 < N; n0 += (int)g) {
                const int nn = N - n0 < g ? N - n0 : (int)g;
                const size_t px = (size_t)n0 * H * W;
                const int rc = conv_wino_fwd(x + px * Cin, u, bias, y + px * Cout, nn, H, W, Cin, Cout, relu, st,
                                             stats ? stats + (size_t)n0 * parts * Cout * 2 : nullptr, mask ? mask + px * Cout : nullptr, accumulate,
                                             in_mr ? in_mr + (size_t)n0 * Cout * 2 : nullptr, in_relu);
                if (rc) return rc;
            }
            return VQW_OK;
        }
    }
    if (conv_wino64_ok(Cin, Cout, W)) return conv_wino64_fwd(x, u, bias, y, N, H, W, Cin, Cout, relu, st, stats, mask, accumulate, in_mr, in_relu);
    const bool wide = W % 32 == 0;
    const size_t lds = (size_t)(3 * (wide ? WinoGeo<32>::HBUF : WinoGeo<16>::HBUF) + 2 * 16 * 32 * WN_KPU + 2 * 8 * 32 * 2 + 512 * 4) * sizeof(float);
    static_assert((size_t)(3 * WinoGeo<16>::HBUF + 2 * 16 * 32 * WN_KPU + 2 * 8 * 32 * 2 + 512 * 4) * sizeof(float) <= 160 * 1024, "Winograd buffers do not fit the 160 KB LDS");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_conv_wino<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_conv_wino<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            vqw_set_error("conv_wino: cannot raise the dynamic LDS limit");
            return VQW_ERR_HIP;
        }
        attr_set = true;
    }
    WinoArgs a;
    a.x = x; a.u = u; a.bias = bias; a.y = y;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.tilesY = ceil_div(H, wide ? 16 : 32); a.tilesX = W / (wide ? 32 : 16); a.nsp = N * a.tilesY * a.tilesX;
    a.ntn = ceil_div(Cout, 32); a.nch = Cin / WN_KC;
    a.relu = relu;
    a.stats = stats;
    const long P = (long)N * H * W;
    a.nbx = (unsigned)(P * Cin * 4);
    a.nbu = (unsigned)(16L * Cout * Cin * 4);
    a.nby = (unsigned)(P * Cout * 4);
    int groups = g_max_blocks / a.ntn;
    if (groups < 1) groups = 1;
    const int even = ceil_div(a.nsp, groups);
    a.kt = even < 1 ? 1 : even;
    if (wide) k_conv_wino<32><<<ceil_div(a.nsp, a.kt) * a.ntn, 512, lds, st>>>(a);
    else k_conv_wino<16><<<ceil_div(a.nsp, a.kt) * a.ntn, 512, lds, st>>>(a);
    VQW_LAUNCH_CHECK("conv_wino");
    return VQW_OK;
}

// =====================================================================================================================
// Weight gradient of the same layers in Winograd form
// =====================================================================================================================
//   dU[xi][co][ci] = sum over tiles of dM[xi][tile][co] V[xi][tile][ci],   dM = A dY A^T (2x2 -> 4x4),  V = B^T d B
//   dW = G^T dU G (4x4 -> 3x3)
// 16 GEMMs with K = tiles: 4/9 of the direct form's matrix work.  A workgroup owns a (32 co x 16 ci) block of dU for a run
// of 8 x 32 pixel regions; its 8 waves split K: wave (tile row tr, half h) takes the tiles 8h..8h+7 of tile row tr in two
// k-steps of 4 tiles on v_mfma_f32_16x16x4_f32 (M = co: two 16-blocks, N = ci, K = tile within the k-step).  Per k-step a
// lane reads the 2x2 dY patch of its tile at its two couts and the 4x4 X patch at its ci (24 ds_read_b32: lanes =
// channels, conflict-free with 40 / 24 floats per pixel), transforms them in registers (2 x 12 + 32 add / sub) and feeds
// 32 MFMAs; the two waves of a SIMD take turns (one transforms while the other holds the matrix pipe).  The dY tile and
// the X halo are double-buffered in LDS like the forward kernel's.  At the end the 8 waves' partial sums are folded through
// LDS, every thread applies G^T . G to one (co, ci) entry and the workgroup writes one slab [Cout][9][Cin] (+ the fused
// bias partial); reduce_rows sums the slabs in a fixed order (deterministic, no float atomics).
namespace {

struct WinoWgArgs {
    const float* x;            // source 0: C0 channels, at half resolution when up0 (nearest x2 up-sampled on the fly)
    const float* x1;           // source 1: C1 channels at full resolution (channel concat [src0 | src1]), or null
    int C0, C1, up0;
    const float* dy;
    float* part;
    float* bias_part;
    int N, H, W, Cin, Cout;
    int tilesY, tilesX, nsp;
    int n_ci_b, nblk, kt;
    unsigned nbx, nbx1, nbd;
};

constexpr int WW_DP = 40, WW_XP = 24;                 // floats per dY pixel (32 co + 8) / per X pixel (16 ci + 8) in LDS
constexpr int WW_D = 256 * WW_DP;                     // floats per dY tile (256 pixels: 8 x 32 or 16 x 16)
constexpr int WW_X = 10 * 34 * WW_XP;                 // floats per X halo buffer (10 x 34 >= 18 x 18 pixels)

// RW = region width: 8 rows x 32 columns, or 16 x 16 for maps whose width is a multiple of 16 only (the 16 x 16 level)
template <int RW>
__global__ void __launch_bounds__(512, 1) k_conv_wino_wgrad(WinoWgArgs a) {
    constexpr int NT = 512;
    constexpr int WW_TR = 256 / RW;                           // output rows per region
    constexpr int XW = RW + 2, XPIX = (WW_TR + 2) * XW;       // halo columns / pixels
    constexpr int LD = 256 * 8 / NT;                          // 4 dY float4 per thread
    constexpr int XF = XPIX * 4;                              // X float4 per halo
    constexpr int LX = (XF + NT - 1) / NT;                    // 3
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ds = smem;                     // [2][WW_D]
    float* Xs = smem + 2 * WW_D;          // [2][WW_X]
    float* Pk = Xs + 2 * WW_X;            // [512][4] parking space of the loader slots past the end of the halo

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
    // the (co, ci) blocks of one spatial split read the same dY / X regions: keep them on one XCD's L2
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int blk = lb % a.nblk, sblk = lb / a.nblk;
    const int co_base = (blk / a.n_ci_b) * 32, ci_base = (blk % a.n_ci_b) * 16;
    const int sp0 = sblk * a.kt;
    const int my_tiles = min(a.kt, a.nsp - sp0);
    const int per_img = a.tilesY * a.tilesX;
    // the 16-channel ci block of the workgroup lies in one of the two sources (C0 % 16 == 0 whenever there are two)
    const bool x_from0 = ci_base < a.C0;
    const unsigned xC = x_from0 ? (unsigned)a.C0 : (unsigned)a.C1;
    const unsigned nbx = x_from0 ? a.nbx : a.nbx1;
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(x_from0 ? a.x : a.x1, nbx), rsd = make_rsrc(a.dy, a.nbd);
    const int x_sh = (x_from0 && a.up0) ? 1 : 0;
    const int x_Hs = H >> x_sh, x_Ws = W >> x_sh;

    // loader: dY float4 f -> pixel f / 8, cout quad f % 8; X float4 f -> halo pixel f / 4, ci quad f % 4
    const int d_c = co_base + (tid & 7) * 4;
    const bool d_ok = d_c < Cout;
    const int x_c = ci_base + (tid & 3) * 4;
    const bool x_cok = x_c < Cin;
    const unsigned x_cb = (unsigned)(x_from0 ? x_c : x_c - a.C0) * 4u;
    const bool do_bias = a.bias_part != nullptr && ci_base == 0;
    float4 rd[LD], rx[LX];
    float4 bsum;
    bsum.x = bsum.y = bsum.z = bsum.w = 0.f;
    int i_n = 0, i_y0 = 0, i_x0 = 0;
    auto issue_setup = [&](int n, int tx, int ty) { i_n = n; i_y0 = ty * WW_TR; i_x0 = tx * RW; };
    auto issue_d = [&](int j) {
        const int pix = (tid >> 3) + 64 * j;            // 0..255
        const int yy = i_y0 + pix / RW, xx = i_x0 + pix % RW;
        const bool ok = d_ok & (yy < H);
        const unsigned p = ((unsigned)i_n * H + (unsigned)yy) * W + (unsigned)xx;
        rd[j] = buf_ld4(rsd, sel_u32(ok, (p * (unsigned)Cout + d_c) * 4u, a.nbd));
    };
    auto issue_x = [&](int j) {
        const int hp = (tid >> 2) + 128 * j;            // 0..XPIX-1 valid
        const int hy = hp / XW, hx = hp - hy * XW;
        const int yy = i_y0 - 1 + hy, xx = i_x0 - 1 + hx;
        const bool ok = (hp < XPIX) & x_cok & ((unsigned)yy < (unsigned)H) & ((unsigned)xx < (unsigned)W);
        const unsigned pix = ((unsigned)i_n * x_Hs + (unsigned)(yy >> x_sh)) * x_Ws + (unsigned)(xx >> x_sh);
        rx[j] = buf_ld4(rsx, sel_u32(ok, pix * xC * 4u + x_cb, nbx));
    };
    // `once`: 1 when the committed tile is one of this workgroup's (the prefetch behind the last tile runs off its share)
    auto commit_d = [&](int j, int buf, float once) {
        const int pix = (tid >> 3) + 64 * j;
        *(float4*)&Ds[buf * WW_D + pix * WW_DP + (tid & 7) * 4] = rd[j];
        bsum.x += once * rd[j].x; bsum.y += once * rd[j].y; bsum.z += once * rd[j].z; bsum.w += once * rd[j].w;   // fused bias gradient
    };
    auto commit_x = [&](int j, int buf) {
        const int hp = (tid >> 2) + 128 * j;
        float* dst = hp < XPIX ? &Xs[buf * WW_X + hp * WW_XP + (tid & 3) * 4] : &Pk[tid * 4];
        *(float4*)dst = rx[j];
    };

    f32x4 acc[16][2];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[xi][mb][r] = 0.f;

    int cn = sp0 / per_img, ctx, cty;
    {
        const int rem = sp0 - cn * per_img;
        ctx = rem / a.tilesY;
        cty = rem - ctx * a.tilesY;
    }
    if (my_tiles > 0) {
        issue_setup(cn, ctx, cty);
#pragma unroll
        for (int j = 0; j < LD; ++j) issue_d(j);
#pragma unroll
        for (int j = 0; j < LX; ++j) issue_x(j);
#pragma unroll
        for (int j = 0; j < LD; ++j) commit_d(j, 0, 1.f);
#pragma unroll
        for (int j = 0; j < LX; ++j) commit_x(j, 0);
    }
    __syncthreads();

    // fragment addressing: lane (channel idx = lane & 15, tile q = lane >> 4 of the k-step); wave (tile row tr, half hh)
    const int idx = lane & 15, q = lane >> 4;
    const int tr = RW == 32 ? wv >> 1 : wv, col0 = RW == 32 ? (wv & 1) * 8 : 0;      // tile row, first tile column of the wave
    // (plain C here, unlike the forward kernel: the transform's last results feed the k-step's first MFMAs directly, and
    // the compiler's hazard recogniser does not see a VALU write inside inline asm - v_sub_f32 in asm followed by the MFMA
    // that reads its result gave a stale B operand)
    auto fadd = [](float x, float y) { return x + y; };
    auto fsub = [](float x, float y) { return x - y; };
    int cur = 0;
    float once = 0.f;
    for (int t = 0; t < my_tiles; ++t) {
        {   // next region (past the last one the loads read another region or nothing; their LDS copy is never consumed)
            const int ty1 = cty + 1, wy = ty1 == a.tilesY ? 1 : 0;
            cty = wy ? 0 : ty1;
            const int tx1 = ctx + wy, wx = tx1 == a.tilesX ? 1 : 0;
            ctx = wx ? 0 : tx1;
            cn += wx;
            issue_setup(cn, ctx, cty);
            once = t + 1 < my_tiles ? 1.f : 0.f;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int tile = col0 + ks * 4 + q;                   // tile column of the region
            // dM = A dY A^T for the lane's two couts, A = [1 0; 1 1; 1 -1; 0 -1]
            float dm[2][16];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const float* Dp = Ds + cur * WW_D + ((2 * tr) * RW + 2 * tile) * WW_DP + mb * 16 + idx;
                const float y00 = Dp[0], y01 = Dp[WW_DP], y10 = Dp[RW * WW_DP], y11 = Dp[(RW + 1) * WW_DP];
                // rows: r0 = y0., r1 = y0. + y1., r2 = y0. - y1., r3 = -y1.   (the sign of r3 is folded into the columns)
                const float r10 = fadd(y00, y10), r11 = fadd(y01, y11), r20 = fsub(y00, y10), r21 = fsub(y01, y11);
                const float rr[4][2] = {{y00, y01}, {r10, r11}, {r20, r21}, {y10, y11}};     // row 3 holds +y1.
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float p0 = rr[i][0], p1 = rr[i][1];
                    if (i < 3) {
                        dm[mb][i * 4 + 0] = p0;
                        dm[mb][i * 4 + 1] = fadd(p0, p1);
                        dm[mb][i * 4 + 2] = fsub(p0, p1);
                        dm[mb][i * 4 + 3] = -p1;
                    } else {               // row 3 = -y1.: negate every entry
                        dm[mb][12] = -p0;
                        dm[mb][13] = -fadd(p0, p1);
                        dm[mb][14] = fsub(p1, p0);
                        dm[mb][15] = p1;
                    }
                }
            }
            // V = B^T d B for the lane's ci
            float vv[16];
            {
                const float* Xp = Xs + cur * WW_X + ((2 * tr) * XW + 2 * tile) * WW_XP + idx;
                float e[4][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float d0 = Xp[c * WW_XP], d1 = Xp[(XW + c) * WW_XP], d2 = Xp[(2 * XW + c) * WW_XP], d3 = Xp[(3 * XW + c) * WW_XP];
                    e[0][c] = fsub(d0, d2); e[1][c] = fadd(d1, d2); e[2][c] = fsub(d2, d1); e[3][c] = fsub(d1, d3);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vv[r * 4 + 0] = fsub(e[r][0], e[r][2]);
                    vv[r * 4 + 1] = fadd(e[r][1], e[r][2]);
                    vv[r * 4 + 2] = fsub(e[r][2], e[r][1]);
                    vv[r * 4 + 3] = fsub(e[r][1], e[r][3]);
                }
            }
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) {
                acc[xi][0] = MFMA16(dm[0][xi], vv[xi], acc[xi][0]);
                if (ks == 0) { if (xi < LD) issue_d(xi); else if (xi < LD + LX) issue_x(xi - LD); }
                else { if (xi >= 4 && xi < 4 + LD) commit_d(xi - 4, cur ^ 1, once); else if (xi >= 4 + LD && xi < 4 + LD + LX) commit_x(xi - 4 - LD, cur ^ 1); }
                __builtin_amdgcn_sched_barrier(0);
                acc[xi][1] = MFMA16(dm[1][xi], vv[xi], acc[xi][1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();               // the region's slots committed buffer cur^1
        cur ^= 1;
    }

    // fold the 8 waves' partial sums through LDS, one xi at a time: thread e = (co, ci) of the 32 x 16 block keeps dU[xi]
    float* red = smem;                    // [8 waves][32 co][16 ci]
    float du[16];
    if (do_bias) {                        // threads with equal (tid & 7) hold the same 4 couts
        *(float4*)&red[tid * 4] = bsum;
        __syncthreads();
        if (tid < 32) {
            const int g = tid >> 2, comp = tid & 3;
            float s = 0.f;
            for (int i = 0; i < 64; ++i) s += red[(i * 8 + g) * 4 + comp];
            if (co_base + tid < Cout) a.bias_part[(size_t)sblk * Cout + co_base + tid] = s;
        }
        __syncthreads();
    }
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
        // C/D layout (16x16): col = lane & 15 (ci), row = 4 (lane >> 4) + r (co within the 16-block)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wv * 512 + (mb * 16 + 4 * q + r) * 16 + idx] = acc[xi][mb][r];
        __syncthreads();
        du[xi] = ((red[tid] + red[512 + tid]) + (red[1024 + tid] + red[1536 + tid])) +
                 ((red[2048 + tid] + red[2560 + tid]) + (red[3072 + tid] + red[3584 + tid]));
        __syncthreads();
    }
    // dW = G^T dU G, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
    {
        const int co = co_base + (tid >> 4), ci = ci_base + (tid & 15);
        float tcol[3][4];       // G^T applied down the rows: [ky][j]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float u0 = du[j], u1 = du[4 + j], u2 = du[8 + j], u3 = du[12 + j];
            tcol[0][j] = u0 + 0.5f * (u1 + u2);
            tcol[1][j] = 0.5f * (u1 - u2);
            tcol[2][j] = u3 + 0.5f * (u1 + u2);
        }
        if (co < Cout && ci < Cin) {
            float* o = a.part + (size_t)sblk * Cout * 9 * Cin + ((size_t)co * 9) * Cin + ci;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float t0 = tcol[ky][0], t1 = tcol[ky][1], t2 = tcol[ky][2], t3 = tcol[ky][3];
                o[(ky * 3 + 0) * Cin] = t0 + 0.5f * (t1 + t2);
                o[(ky * 3 + 1) * Cin] = 0.5f * (t1 - t2);
                o[(ky * 3 + 2) * Cin] = t3 + 0.5f * (t1 + t2);
            }
        }
    }
}

}  // namespace

// (two sources: both channel counts multiples of 16, so that a 16-channel ci block never straddles them)
bool conv_wino_wgrad_ok(int Cin, int Cout, int N, int H, int W) {
    if (g_wino_mode != 0 || !g_wino_env) return false;
    if (Cin % 16 != 0 || Cout % 32 != 0 || W % 16 != 0 || H < 2 || N < 1) return false;
    return (long)N * H * W * (Cin > Cout ? Cin : Cout) * 4 <= 0xFFFFFFE0L;
}
// slabs for a given upper bound
int conv_wino_wgrad_blocks(const ConvIn& in, int Cout, int N, int H, int W, int max_slabs, int* kt_out) {
    const int Cin = in.C0 + in.C1;
    if (conv_wino64_wgrad_ok(in.C0, in.C1, in.up0, Cout, H, W)) return conv_wino64_wgrad_blocks(Cin, Cout, N, H, W, max_slabs, kt_out);
    if (conv_wino32_wgrad_ok(in.C0, in.C1, in.up0, Cout, H, W)) return conv_wino32_wgrad_blocks(Cin, Cout, N, H, W, max_slabs, kt_out);
    const int nblk = (Cout / 32) * (Cin / 16);
    const int rw = W % 32 == 0 ? 32 : 16;
    const int nsp = N * ceil_div(H, 256 / rw) * (W / rw);
    int nsb = g_max_blocks / nblk;
    if (nsb > max_slabs) nsb = max_slabs;
    if (nsb > nsp) nsb = nsp;
    if (nsb < 1) nsb = 1;
    const int kt = ceil_div(nsp, nsb);
    if (kt_out) *kt_out = kt;
    return ceil_div(nsp, kt);
}
int conv_wino_wgrad(const ConvIn& in, const float* dy, float* ws, float* bpart, int N, int H, int W, int Cout, int nsb, int kt,
                    hipStream_t st) {
    const int Cin = in.C0 + in.C1;
    if (conv_wino64_wgrad_ok(in.C0, in.C1, in.up0, Cout, H, W)) return conv_wino64_wgrad(in, dy, ws, bpart, N, H, W, Cin, Cout, nsb, kt, st);
    if (conv_wino32_wgrad_ok(in.C0, in.C1, in.up0, Cout, H, W)) return conv_wino32_wgrad(in, dy, ws, bpart, N, H, W, Cin, Cout, nsb, kt, st);
    constexpr size_t lds = (size_t)(2 * (WW_D + WW_X) + 512 * 4) * sizeof(float);
    static_assert(lds <= 160 * 1024 && lds >= 8 * 512 * sizeof(float), "Winograd wgrad tiles do not fit the LDS");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_conv_wino_wgrad<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_conv_wino_wgrad<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            vqw_set_error("conv_wino_wgrad: cannot raise the dynamic LDS limit");
            return VQW_ERR_HIP;
        }
        attr_set = true;
    }
    const long P = (long)N * H * W;
    WinoWgArgs a;
    a.x = in.src0; a.x1 = in.src1; a.C0 = in.C0; a.C1 = in.C1; a.up0 = in.up0;
    a.dy = dy; a.part = ws; a.bias_part = bpart;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    const int rw = W % 32 == 0 ? 32 : 16;
    a.tilesY = ceil_div(H, 256 / rw); a.tilesX = W / rw; a.nsp = N * a.tilesY * a.tilesX;
    a.n_ci_b = Cin / 16; a.nblk = (Cout / 32) * a.n_ci_b; a.kt = kt;
    a.nbx = (unsigned)((in.up0 ? P / 4 : P) * in.C0 * 4);
    a.nbx1 = (unsigned)(P * in.C1 * 4);
    a.nbd = (unsigned)(P * Cout * 4);
    if (rw == 32) k_conv_wino_wgrad<32><<<a.nblk * nsb, 512, lds, st>>>(a);
    else k_conv_wino_wgrad<16><<<a.nblk * nsb, 512, lds, st>>>(a);
    VQW_LAUNCH_CHECK("conv_wino_wgrad");
    return VQW_OK;
}
