// Winograd F(2x2, 3x3) forward / dgrad for layers with Cout % 64 == 0, Cin % 16 == 0: the 64-cout workgroup tile.
//
// Why a second kernel.  Beside fp32 MFMAs every VALU instruction is ADDITIVE (profiles/r03_mfma_valu_microbench.txt:
// v_mfma_f32_16x16x4_f32 runs on the SIMD's fp32 lanes; one v_add / v_mov / integer add costs 4.4-5 matrix-pipe cycles at
// two waves per SIMD, 9-10 at one, a v_pk_* 10-15, VCC / SGPR-operand forms 6-7; LDS reads and writes, SALU and s_nop cost
// nothing).  k_conv_wino (conv_wino.hip) issues ~115 VALU per 64 MFMAs (64 for the input transform, ~45 for the prefetch
// addresses): 2048 / (2048 + 115 * 4.4 + barrier) = the 0.64 matrix-pipe occupancy its counters show.  This kernel is built
// to issue as few VALU instructions as the algorithm allows:
//
//   * workgroup = 64 Winograd tiles (8 x 32 or 16 x 16 output pixels) x 64 couts; wave (mb, h) = tiles of M block mb (16
//     tiles), HALF of the 16 transform positions (xi rows 2h, 2h+1) and all 64 couts: 8 xi x 4 N blocks x 4 = 128
//     accumulator registers, 64 MFMAs per 8-channel chunk like before - but the input transform of a wave is 32 add / sub
//     (two of the four rows of B^T d B) instead of 64, and it is shared by 64 couts instead of 32: a quarter of the
//     transform VALU per MFMA of the 32-cout kernel.
//   * no address arithmetic in the loop: a halo slot's byte offset is computed once per REGION (with the padding / edge
//     select folded in: an invalid lane holds an out-of-range offset, the buffer load returns 0), the channel chunk is
//     the load's scalar offset (soffset: not part of the range check), the U slots differ by scalar offsets only; LDS
//     addresses are per-lane bases plus immediates because everything that alternates (halo buffer, U buffer, operand
//     set) follows the item's parity, which is a template parameter of the item body.
//   * the accumulators of a region's first chunk start from the MFMA's inline 0 (a body variant), not from 128 v_mov.
//   * region epilogue: each wave applies A^T . A to its half of the xi (linear: y = y_h0 + y_h1), the halves meet through
//     LDS (a wave finalises two of the four N blocks and sends the other two to its partner), then bias / ReLU /
//     statistics partials / stores as in the 32-cout kernel.
//
// Halo ring: two buffers suffice - the patch of item i+1 is read (and transformed) during item i, so the buffer of
// item i is dead from the barrier that ends item i-1 and receives the halo of item i+2 during item i.
#include "common.h"
#include "conv_common.h"
#include "mfma_util.h"
#include <cstdlib>
#include <type_traits>

extern int g_wino_mode;

namespace {

static int env_int64(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
static const int g_w64_env = env_int64("VQW_WINOGRAD64", 1);
static const int g_w64_max_blocks = []{ int v = env_int64("VQW_CONV_MAX_BLOCKS", 256); return v < 8 ? 8 : (v > 256 ? 256 : v); }();
// weight-gradient kernels (they run on the side lanes beside the chain): VQW_WGRAD_MAX_BLOCKS leaves CUs to the chain's kernels (experiment)
static const int g_w64_max_blocks_wg = []{ int v = env_int64("VQW_WGRAD_MAX_BLOCKS", g_w64_max_blocks); return v < 8 ? 8 : (v > 256 ? 256 : v); }();

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct W64Args {
    const float* x;
    const float* u;            // [Cin / 8][16 xi][Cout][8] (k_wino_weights, chunked)
    const float* bias;
    float* y;
    int N, H, W, Cin, Cout;
    int tilesY, tilesX, nsp;   // regions per image column / row, total
    int ntn, nch;              // 64-cout tiles, 8-channel chunks (even)
    int kt;                    // consecutive regions per workgroup
    int relu;
    unsigned nbx, nbu, nby;
    float* stats;              // optional [N][tilesY*tilesX][Cout][2]: per-region (sum, M2 about the region mean)
    const float* mask;         // EPI 1, shaped like y: outputs are zeroed where mask <= 0 (the ReLU whose output the layer's
                               // forward read: an input-gradient launch then delivers the gradient in front of that ReLU)
                               // EPI 2: y += result (a later member of a gradient group adds to the shared buffer)
                               // EPI 3: mask = x, the RAW input of the InstanceNorm (+ReLU) whose output the layer's forward
                               // read; mr = that norm's (mean, rstd) per (image, channel).  The launch is the layer's input
                               // gradient g; beside storing it, it leaves the norm's backward sums per region in `stats`:
                               // (sum gm, sum gm * xhat), gm = g where the norm's ReLU passed (all of g without a ReLU)
    const float* mr;
    int in_relu;
    // EPI 4: the couts split into two tensors: [0, split) -> y (channel stride split; pool0: summed over each 2 x 2 output tile
    // and stored at half resolution - the adjoint of a nearest x2 up-sampling that fed those channels), [split, Cout) -> y2
    float* y2;
    int split, pool0;
    int c1;                    // channels of y2 (<= Cout - split: the couts beyond split + c1 are padding and are not stored)
    unsigned nby2;
    // Pixel (n, y, x) of input AND output sits at pixel index (n >> n_sh) * img_px + ((n >> 1) & n_m) * rowb_px + (n & n_m) + y * rpx
    // + x * ppx.  Dense tensors: n_sh = n_m = 0, img_px = H W, rpx = W, ppx = 1.  A 3x3 layer of dilation 2 is four plain layers
    // on the phase images (rows / columns of one parity) of its tensors: N = 4 x images, H and W halved, n_sh = 2, n_m = 1,
    // img_px = the full image, rowb_px = the full width, rpx = 2 x full width, ppx = 2 (conv_wino64_fwd_dil2).
    unsigned img_px, rowb_px, rpx, ppx;
    int n_sh, n_m;
};

constexpr int W6_KPH = 10;                 // floats per halo pixel in LDS (8 channels + 2: conflict-free ds_read_b64 patches)
// Workgroup shapes.  MBW = M blocks (of 16 tiles) per wave, NBW = 4 / MBW N blocks (of 16 couts) per wave:
//   MBW = 1: 64 tiles x 64 couts (Cout % 64 == 0): 32 transform VALU per 64 MFMAs;
//   MBW = 2: 128 tiles x 32 couts (Cout % 32 == 0, widths that are multiples of 32): 64 per 64 MFMAs - what a 32-cout
//            tile costs in any layout - but none of the 32-cout kernel's address arithmetic, accumulator zeroing and
//            post-barrier bubbles.
// Region geometry, RW = region width: 8 MBW rows x 32 columns (halo rows x 34), or - for maps whose width is a multiple of
// 16 only (the 16 x 16 level, MBW = 1) - 16 x 16 (halo 18 x 18, LDS row stride 24 pixels: the two tile rows of an M block
// then land on complementary banks).
// NW = waves per workgroup: 8 (one workgroup per CU) or 4 - HALF the region rows, two workgroups per CU (same two waves per
// SIMD, same registers per wave).  The two workgroups of a CU are in independent phases: while one sits in its region
// epilogue, at a barrier or behind a prefetch commit, the other one's MFMAs run.  With eight waves in lock step those
// costs ADD to the MFMA time (timing-only A/B builds, 32->32 @256: MFMAs alone 0.138 ms, + epilogue 0.053, + loads 0.032,
// + transform 0.018, + barriers 0.006 = the 0.239 measured).
template <int RW, int MBW, int NW = 8> struct W64Geo {
    static_assert(MBW == 1 || (MBW == 2 && RW == 32), "two M blocks per wave: 32-wide regions only");
    static_assert(NW == 8 || (NW == 4 && RW == 32), "four-wave workgroups: 32-wide regions only");
    static constexpr int NMB = NW / 2;                     // waves per xi half = M-block rows of the workgroup
    static constexpr int NBW = 4 / MBW;
    static constexpr int NCO = 16 * NBW;                   // couts per workgroup
    static constexpr int TR = RW == 32 ? 2 * MBW * NMB : 16;
    static constexpr int HR = TR + 2, HWV = RW + 2;
    static constexpr int HWS = RW == 32 ? 34 : 24;
    static constexpr int HBUF = HR * HWS * W6_KPH;
    static constexpr int UBUF = 16 * NCO * 8;              // floats per U chunk: [xi][couts][8 channels], float4 halves swizzled by cout bit 3
    static constexpr int EXF = 2048 * NW;                  // exchange area of the region epilogue: 8 KB per wave
    static_assert(EXF >= UBUF, "U buffer 1 lives in the exchange area");
    static constexpr size_t LDS_FLOATS = 2 * HBUF + UBUF + EXF + 2 * NMB * NCO * 2;
};
#ifdef W6_EXP_NO_BARRIER
#define W6_ITEM_BARRIER() do {} while (0)
#else
#define W6_ITEM_BARRIER() __syncthreads()
#endif
#ifndef W6_LS
#define W6_LS 4                            // MFMA positions between two prefetch loads (a burst of 48 wave-loads stalls their issue;
                                           // 2 -> 4: 32->64 @256 dgrad 0.447 -> 0.425 ms, 64->128 @128 0.318 -> 0.306, nothing slower)
#endif
constexpr int W6_CP = 26;                  // MFMA position of the first LDS commit of the prefetched data
#ifndef W6_CP1
#define W6_CP1 26                          // ... of the 64-cout shape (MBW = 1), which has registers to hold the loads longer
#endif
constexpr int W6_PB = 33, W6_CB = 58;      // second phase of U slots (four-wave workgroups): first issue, first commit

template <int RW, int MBW, int EPI, int NW = 8>
__global__ void __launch_bounds__(64 * NW, 8 / NW) k_conv_wino64(W64Args a) {
    constexpr bool MASK = EPI >= 1 && EPI <= 3;   // these forms read 16 values per (M block, N block) at the output positions
    using G = W64Geo<RW, MBW, NW>;
    constexpr int NT = 64 * NW, NMB = G::NMB, NBW = G::NBW, NCO = G::NCO;
    constexpr int HWS = G::HWS, HWV = G::HWV, HBUF = G::HBUF, UBUF = G::UBUF;
    constexpr int HPIX = G::HR * HWV;          // 340 / 324 / 612 halo pixels
    constexpr int HF = HPIX * 2;               // float4 per halo chunk: 680 / 648 / 1224
    constexpr int LH = (HF + NT - 1) / NT;     // 2 / 2 / 3 halo slots per thread
    constexpr int LU = 512 * NBW / NT;         // U slots per thread: 512 NBW float4 per chunk
    static_assert(HF > (LH - 1) * NT && HF >= NT, "every halo slot but the last is full");
    // All slots in the first half of an item where they fit; otherwise the second half of the U slots forms a second phase
    // (issued from position W6_PB on, committed from W6_CB on) that re-uses the registers of the first.
    constexpr int CP = MBW == 1 ? W6_CP1 : W6_CP;
    constexpr int CPMAX = MBW == 1 ? 64 : 32;
    constexpr bool ONE_PHASE = 1 + (LH + LU) * W6_LS <= CP && CP + LH + LU <= CPMAX;
    constexpr int LUA = ONE_PHASE ? LU : LU / 2;
    static_assert(1 + (LH + LUA) * W6_LS <= CP && CP + LH + LUA <= CPMAX, "prefetch slots fit the first half of an item");
    static_assert(ONE_PHASE || (LU == 2 * LUA && W6_PB + LUA * W6_LS <= W6_CB && W6_CB + LUA <= 64), "second U phase fits the second half");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Hs = smem;                      // [2][HBUF]
    float* Us = smem + 2 * HBUF;           // [2][UBUF]
    float* Ex = Us + UBUF;                 // exchange area of the region epilogue (64 KB): U buffer 1 (dead by then) + spare
    float* Rs = Ex + G::EXF;               // [2][NMB waves of a half][NCO couts][2] statistics of the waves' pixels

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: scalar branches on h
    const int mb = wv & (NMB - 1), h = wv / NMB;
    const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(a.x, a.nbx), rsu = make_rsrc(a.u, a.nbu), rsy = make_rsrc(a.y, a.nby);
    const __amdgpu_buffer_rsrc_t rsm = make_rsrc((EPI == 1 || EPI == 3) ? a.mask : a.y, a.nby);  // (instantiations of their own: the plain kernels keep their registers)
    const __amdgpu_buffer_rsrc_t rsr = make_rsrc(EPI == 3 ? a.mr : a.y, EPI == 3 ? (unsigned)((size_t)a.N * a.Cout * 8) : a.nby);
    const __amdgpu_buffer_rsrc_t rsy2 = make_rsrc(EPI == 4 ? a.y2 : a.y, EPI == 4 ? a.nby2 : a.nby);

    const int ntn = a.ntn, nch = a.nch;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lb % ntn;
    const int sp0 = (lb / ntn) * a.kt;
    const int co_base = tile_n * NCO;
    const int my_tiles = min(a.kt, a.nsp - sp0);
    const int per_img = a.tilesY * a.tilesX;
    if (my_tiles <= 0) return;             // uniform per workgroup

    // ---- loader slots (fixed per thread) ----
    // halo float4 f -> halo pixel f / 2, channel quad f % 2; the last slot of the threads past the end repeats another
    // thread's slot (same address, same data: a benign duplicate instead of a masked store)
    const int c4 = tid & 1;
    auto halo_pixel = [&](int j, int& hy, int& hx) {       // (recomputed where needed: a division by a constant, no register held)
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
        h_lds[j] = (hy * HWS + hx) * W6_KPH + c4 * 4;
    }
    unsigned h_voff[LH];                   // byte offsets of the slots in the region whose halo is fetched next
    auto basepix = [&](int n) {                // (scalar: n is uniform)
        return (unsigned)(n >> a.n_sh) * a.img_px + (unsigned)((n >> 1) & a.n_m) * a.rowb_px + (unsigned)(n & a.n_m);
    };
    auto region_offsets = [&](int n, int tx, int ty) {
        const int y0 = ty * G::TR - 1, x0 = tx * RW - 1;
        const unsigned nb = basepix(n);
#pragma unroll
        for (int j = 0; j < LH; ++j) {
            int hy, hx;
            halo_pixel(j, hy, hx);
            const int yy = y0 + hy, xx = x0 + hx;
            const bool ok = ((unsigned)yy < (unsigned)H) & ((unsigned)xx < (unsigned)W);
            const unsigned pix = nb + (unsigned)yy * a.rpx + (unsigned)xx * a.ppx;
            h_voff[j] = sel_u32(ok, pix * (unsigned)Cin * 4u + (unsigned)c4 * 16u, 0xFFFFFFFFu);
        }
    };
    // U float4 f = tid + NT j -> row = xi * NCO + n = (tid >> 1) + NT / 2 j, channel quad tid & 1
    // (a wave's slot = 32 couts x 32 bytes = 1 KB in a row of the chunked layout [Cin / 8][16 xi][Cout][8])
    const int u_n = (tid >> 1) & (NCO - 1);
    const unsigned u_voff = (((unsigned)(tid >> 1) / NCO * Cout + co_base + u_n) * 8u + c4 * 4) * 4u;
    const unsigned u_jstride = (unsigned)(NT / 2 / NCO) * Cout * 32u;      // NT / 2 rows = NT / 2 / NCO xi further per slot
    const unsigned u_cstride = 16u * Cout * 32u;                           // per 8-channel chunk
    const int u_lds = (tid >> 1) * 8 + ((c4 ^ ((u_n >> 3) & 1)) * 4);      // + j * NT * 4 floats

    float4 rh[LH], ru[LUA];
    auto issue_h = [&](int j, int chunk) {
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)h_voff[j], chunk * 32, 0);
        unsigned a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
        rh[j].x = __uint_as_float(a0); rh[j].y = __uint_as_float(a1); rh[j].z = __uint_as_float(a2); rh[j].w = __uint_as_float(a3);
    };
    auto issue_u = [&](int j, int chunk) {
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsu, (int)u_voff, (int)(chunk * u_cstride + j * u_jstride), 0);
        unsigned a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
        float4& r = ru[j % LUA];
        r.x = __uint_as_float(a0); r.y = __uint_as_float(a1); r.z = __uint_as_float(a2); r.w = __uint_as_float(a3);
    };
    auto commit_h = [&](int j, float* Hb) {    // a halo pixel is 40 bytes: two 8-byte-aligned halves
        float* p = Hb + h_lds[j];
        f32x2 lo, hi;
        lo.x = rh[j].x; lo.y = rh[j].y; hi.x = rh[j].z; hi.y = rh[j].w;
        *(f32x2*)p = lo;
        *(f32x2*)(p + 2) = hi;
    };
    auto commit_u = [&](int j, float* Ub) { *(float4*)&Ub[u_lds + j * (NT * 4)] = ru[j % LUA]; };

    // ---- item cursors: (image, strip, region row, chunk) of items i, i+1, i+2 ----
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

    // ---- fragment bases: lane (tile m = lane & 15, channel pair q = lane >> 4) ----
    const int m = lane & 15, q = lane >> 4;
    // rows of the 4 x 4 patch this wave needs, as (A, B, C): h = 0: (0, 1, 2) -> e0 = A - C, e1 = B + C;
    // h = 1: (3, 2, 1) -> A - C = -(e3), B - C = e2.  One formula pair with sg = +1 / -1: eP = A - C, eQ = B + sg * C
    // (sg * C is exact, so the fma rounds once like the add / sub it stands for); xi row 3 arrives negated, which the
    // epilogue's h = 1 branch folds into its signs.
    const int rA = h ? 3 : 0, rB = h ? 2 : 1, rC = h ? 1 : 2;
    // M block (mb, mbw) of the wave: RW = 32: tile row MBW mb + mbw, columns m; RW = 16: tile rows 2 mb + (m >> 3), columns m & 7
    const int prow = RW == 32 ? 2 * MBW * mb : 2 * (2 * mb + (m >> 3)), pcol = RW == 32 ? 2 * m : 2 * (m & 7);
    constexpr int MBW_STEP = 2 * HWS * W6_KPH;             // the wave's second M block: one tile row = two halo rows further
    auto opaque = [](int x) { asm volatile("" : "+v"(x)); return x; };     // distinct base registers: no ds_read2 merging
    const int a_A = opaque(((prow + rA) * HWS + pcol) * W6_KPH + 2 * q);
    const int a_B = opaque(((prow + rB) * HWS + pcol) * W6_KPH + 2 * q);
    const int a_C = opaque(((prow + rC) * HWS + pcol) * W6_KPH + 2 * q);
    // local xi row 0 = eP: xi row (h ? 3 : 0); local row 1 = eQ: xi row (h ? 2 : 1)
    const int b_swz = ((q >> 1) ^ (m >> 3)) * 4 + (q & 1) * 2;
    const int b_0 = opaque(((h ? 3 : 0) * 4 * NCO + m) * 8 + b_swz);       // + (cc * NCO + nb * 16) * 8
    const int b_1 = opaque(((h ? 2 : 1) * 4 * NCO + m) * 8 + b_swz);
    // sg, mone live in VGPRs the compiler cannot see through: an SGPR operand costs 1.5 x, and a literal -1 would turn the
    // fma back into a subtraction that the vector combiner pairs into v_pk_add_f32 over the two channels of a ds_read_b64
    // (10-15 pipe cycles instead of 2 x 4.4)
    float sg, mone;
    { float s = h ? -1.f : 1.f; asm volatile("v_mov_b32 %0, %1" : "=v"(sg) : "v"(s)); }
    { float s = -1.f; asm volatile("v_mov_b32 %0, %1" : "=v"(mone) : "v"(s)); }

    f32x4 acc[8][MBW][NBW];
    f32x2 bf[2][NBW];                  // B fragments: xi l in bf[l & 1]; those of an item's last xi cross the barrier
    f32x2 dcol[2][3];                  // two patch columns in flight: rows A, B, C
    float e[2][2][4];                  // [channel of the pair][local row][column] after the column pass (one M block at a time)
    float v[2][MBW][2][8];             // [parity of the item][M block][channel of the pair][local xi = local row * 4 + column]

    // The input transform of an item = MBW x (16 column operations, 16 row operations), one M block after the other:
    // operation o of M block w.  Column pass: column o >> 2, (channel, local row) = o & 3; row pass: o - 16.
    auto read_col = [&](const float* Hb, int w, int c) {
        const float* Hc = Hb + w * MBW_STEP + c * W6_KPH;
        dcol[c & 1][0] = *(const f32x2*)&Hc[a_A];
        dcol[c & 1][1] = *(const f32x2*)&Hc[a_B];
        dcol[c & 1][2] = *(const f32x2*)&Hc[a_C];
    };
    auto xform_op = [&](int par, int w, int o) {
        if (o < 16) {
            const int c = o >> 2, k = o & 3, t = k >> 1;
            const float dA = dcol[c & 1][0][t], dB = dcol[c & 1][1][t], dC = dcol[c & 1][2][t];
            if ((k & 1) == 0) e[t][0][c] = __builtin_fmaf(mone, dC, dA);
            else e[t][1][c] = __builtin_fmaf(sg, dC, dB);
        } else {
            const int k = o - 16, lr = k >> 3, t = (k >> 2) & 1, cc = k & 3;
            const float e0 = e[t][lr][0], e1 = e[t][lr][1], e2 = e[t][lr][2], e3 = e[t][lr][3];
            v[par][w][t][lr * 4 + cc] = cc == 0 ? e0 - e2 : cc == 1 ? e1 + e2 : cc == 2 ? e2 - e1 : e1 - e3;
        }
    };
    // what the transform does beside MFMA position p of an item (reads three positions ahead of a column's first operation);
    // MBW = 1: positions 32..63; MBW = 2: operation g = 32 w + o at position g + 4, the last four doubled up at 60..63
    auto xform_slot = [&](const float* Hn, int par, int p) {
        constexpr int NOPS = 32 * MBW, P0 = MBW == 1 ? 32 : 4;
        auto op = [&](int g) { xform_op(par, g >> 5, g & 31); };
        auto rd = [&](int g) {         // the read that must be in flight before operation g: column (g & 31) >> 2 of M block g >> 5
            if ((g & 31) < 16 && ((g & 31) & 3) == 0) read_col(Hn, g >> 5, (g & 31) >> 2);
        };
        // reads: for the column whose first operation sits at position p + 3
        if (p + 3 - P0 >= 0 && p + 3 - P0 < NOPS) rd(p + 3 - P0);
        if (p - P0 >= 0 && p - P0 < NOPS && p < 64) {
            const int g = p - P0;
            if (P0 + NOPS <= 64 || g < NOPS - 2 * (P0 + NOPS - 64)) op(g);
            else { const int d = g - (NOPS - 2 * (P0 + NOPS - 64)); op(NOPS - 2 * (P0 + NOPS - 64) + 2 * d); op(NOPS - 2 * (P0 + NOPS - 64) + 2 * d + 1); }
        }
    };

    // ---- prologue: halo of items 0 and 1, U chunk of item 0, operands of item 0 ----
    region_offsets(cn, ctx, cty);
#pragma unroll
    for (int j = 0; j < LH; ++j) issue_h(j, 0);
#pragma unroll
    for (int j = 0; j < LUA; ++j) issue_u(j, 0);
#pragma unroll
    for (int j = 0; j < LH; ++j) commit_h(j, Hs);
#pragma unroll
    for (int j = 0; j < LUA; ++j) commit_u(j, Us);
#pragma unroll
    for (int j = LUA; j < LU; ++j) issue_u(j, 0);
#pragma unroll
    for (int j = LUA; j < LU; ++j) commit_u(j, Us);
    region_offsets(n1, tx1, ty1);
#pragma unroll
    for (int j = 0; j < LH; ++j) issue_h(j, ch1);
#pragma unroll
    for (int j = 0; j < LH; ++j) commit_h(j, Hs + HBUF);
    __syncthreads();
#pragma unroll
    for (int w = 0; w < MBW; ++w)
#pragma unroll
        for (int o = 0; o < 32; ++o) {
            if (o < 16 && (o & 3) == 0) read_col(Hs, w, o >> 2);
            xform_op(0, w, o);
        }
    __syncthreads();                   // halo buffer 0 is overwritten during item 0
    region_offsets(n2, tx2, ty2);

    float bvv[NBW / 2];                // bias of the N blocks this wave finalises: nb = h NBW / 2 + i
#pragma unroll
    for (int i = 0; i < NBW / 2; ++i) bvv[i] = a.bias ? a.bias[co_base + (h * (NBW / 2) + i) * 16 + m] : 0.f;
    const float lo = a.relu ? 0.f : -__builtin_inff();
    const bool plain = a.bias == nullptr && !a.relu;
    int spar = 0;

    // One item; PAR = its parity: operands v[PAR], halo of the NEXT item in buffer PAR ^ 1, U chunk in buffer PAR; the halo of
    // item i+2 goes to halo buffer PAR, the U chunk of item i+1 to U buffer PAR ^ 1.  FIRST: first chunk of a region.
    auto body = [&](auto PAR, auto FIRST) {
        constexpr int par = decltype(PAR)::value;
        constexpr bool first = decltype(FIRST)::value;
        const float* Uc = Us + par * UBUF;
        const float* Hn = Hs + (par ^ 1) * HBUF;
        float* Hw = Hs + par * HBUF;
        float* Uw = Us + (par ^ 1) * UBUF;
        auto ldb = [&](int l, int nb) {
            bf[l & 1][nb] = *(const f32x2*)&Uc[(l < 4 ? b_0 : b_1) + ((l & 3) * NCO + nb * 16) * 8];
        };
        auto slot = [&](int p) {       // p = 0..63: MFMA position (compile-time after unrolling)
#if !defined(W6_EXP_NO_LOADS) && !defined(W6_EXP_NO_HLOADS)      // timing-only A/B builds (tools/wino_ab.sh): wrong results
            if (p >= 1 && p < 1 + LH * W6_LS && (p - 1) % W6_LS == 0) issue_h((p - 1) / W6_LS, ch2);
            if (p >= CP && p < CP + LH) commit_h(p - CP, Hw);
#endif
#if !defined(W6_EXP_NO_LOADS) && !defined(W6_EXP_NO_ULOADS)
            if (p >= 1 + LH * W6_LS && p < 1 + (LH + LUA) * W6_LS && (p - 1) % W6_LS == 0) issue_u((p - 1) / W6_LS - LH, ch1);
            if (p >= CP + LH && p < CP + LH + LUA) commit_u(p - CP - LH, Uw);
            if (!ONE_PHASE) {
                if (p >= W6_PB && p < W6_PB + LUA * W6_LS && (p - W6_PB) % W6_LS == 0) issue_u(LUA + (p - W6_PB) / W6_LS, ch1);
                if (p >= W6_CB && p < W6_CB + LUA) commit_u(LUA + p - W6_CB, Uw);
            }
#endif
#ifndef W6_EXP_NO_XFORM
            xform_slot(Hn, par ^ 1, p);
#endif
        };
        // The item's first B fragments are requested right behind the barrier; the eight MFMAs of the PREVIOUS item's last xi
        // (operands and fragments already in registers) run while they arrive - without them every wave of the workgroup
        // would sit through the LDS latency with an idle matrix pipe once per item.
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) { ldb(0, nb); __builtin_amdgcn_sched_barrier(0); }
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        // MFMA j = 0..7 of an xi: M block j / (2 NBW), k-step (j / NBW) & 1, N block j % NBW (two MFMAs on one accumulator are
        // NBW >= 2 positions apart)
        if (first) {
#pragma unroll
            for (int p = 0; p < 8; ++p) slot(p);
#pragma unroll
            for (int w = 0; w < MBW; ++w)
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) acc[7][w][nb] = zero4;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int w = j / (2 * NBW), k = (j / NBW) & 1, nb = j % NBW;
                acc[7][w][nb] = MFMA16(v[par ^ 1][w][k][7], k == 0 ? bf[1][nb].x : bf[1][nb].y, acc[7][w][nb]);
                slot(j);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int l = 0; l < 7; ++l) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int w = j / (2 * NBW), k = (j / NBW) & 1, nb = j % NBW;
#ifndef W6_EXP_NO_BREAD
                if (j < NBW) ldb(l + 1, nb);             // (l = 6: the fragments of xi 7 stay in bf[1] for the next body)
#endif
                acc[l][w][nb] = MFMA16(v[par][w][k][l], k == 0 ? bf[l & 1][nb].x : bf[l & 1][nb].y, (first && k == 0) ? zero4 : acc[l][w][nb]);
                slot(8 + l * 8 + j);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto flush = [&]() {               // the last xi of a region's last item (odd parity)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int w = j / (2 * NBW), k = (j / NBW) & 1, nb = j % NBW;
            acc[7][w][nb] = MFMA16(v[1][w][k][7], k == 0 ? bf[1][nb].x : bf[1][nb].y, acc[7][w][nb]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Region epilogue of a wave (HH = its xi half): partial y = A^T M_h A over the wave's two xi rows for every (tile, cout)
    // entry it holds; N blocks HH NBW / 2 .. are finalised here, the others go to the partner wave (same mb, other h).
    // C/D layout (16x16): col = lane & 15 (cout), row = 4 (lane >> 4) + r (tile of the M block).
    auto partial = [&](auto HH, int w, int nb, int r, float* yv) {       // yv[a * 2 + b]
        constexpr int hh = decltype(HH)::value;
        float p0[4], p1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float l0 = acc[j][w][nb][r], l1 = acc[4 + j][w][nb][r];
            if (hh == 0) { p0[j] = l0 + l1; p1[j] = l1; }          // local rows = m0, m1:   m0 + m1 | m1
            else { p0[j] = l1; p1[j] = l0 - l1; }                  // local rows = -m3, m2:  m2 | -m2 - m3
        }
        yv[0] = (p0[0] + p0[1]) + p0[2];
        yv[1] = (p0[1] - p0[2]) - p0[3];
        yv[2] = (p1[0] + p1[1]) + p1[2];
        yv[3] = (p1[1] - p1[2]) - p1[3];
    };
    // 1. the partner's N blocks -> exchange area [(mb, destination h)][k = 0..7][lane] float4, k = (w NBW / 2 + i) 4 + r
    auto epi_send = [&](auto HH) {
        constexpr int hh = decltype(HH)::value;
        float* Xo = Ex + (((mb * 2 + (hh ^ 1)) * 8) * 64 + lane) * 4;
#pragma unroll
        for (int w = 0; w < MBW; ++w)
#pragma unroll
            for (int i = 0; i < NBW / 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float yv[4];
                    partial(HH, w, (hh ^ 1) * (NBW / 2) + i, r, yv);
                    float4 o;
                    o.x = yv[0]; o.y = yv[1]; o.z = yv[2]; o.w = yv[3];
                    *(float4*)&Xo[((w * (NBW / 2) + i) * 4 + r) * 256] = o;
                }
    };
    // 2. own N blocks: own partial + the partner's, bias / ReLU, statistics, stores
    auto epi_finish = [&](auto HH) {
        constexpr int hh = decltype(HH)::value;
        const float* Xi = Ex + (((mb * 2 + hh) * 8) * 64 + lane) * 4;
        float st1[NBW / 2], st2[NBW / 2];      // statistics of the wave's pixels per finalised N block
#pragma unroll
        for (int w = 0; w < MBW; ++w) {
            // tiles 4 q + r of M block (mb, w): RW = 32: tile row MBW mb + w, columns 4 q + r; RW = 16: tile row 2 mb + (q >> 1),
            // columns 4 (q & 1) + r
            const int yrow0 = cty * G::TR + (RW == 32 ? 2 * (MBW * mb + w) : 2 * (2 * mb + (q >> 1)));
            const int xcol0 = ctx * RW + (RW == 32 ? 8 * q : 8 * (q & 1));
#pragma unroll
            for (int i = 0; i < NBW / 2; ++i) {
                float yv[16];      // [r][a][b]
                const int co0 = co_base + (hh * (NBW / 2) + i) * 16;       // uniform
                const unsigned co = (unsigned)(co0 + m);
                // EPI 4: which tensor this N block belongs to (uniform), its channel stride and the lane's channel in it
                const bool part0 = EPI == 4 && co0 < a.split;
                const bool pooled = part0 && a.pool0;
                const unsigned cs = EPI != 4 ? (unsigned)Cout : (unsigned)(part0 ? a.split : a.c1);
                const unsigned cc = EPI != 4 ? co : (part0 ? co : co - (unsigned)a.split);
                int voffs[2];
#pragma unroll
                for (int aa = 0; aa < 2; ++aa) {
                    const int yy = yrow0 + aa;
                    const unsigned base = (basepix(cn) + (unsigned)yy * a.rpx + (unsigned)xcol0 * a.ppx) * cs + cc;
                    voffs[aa] = (int)sel_u32(yy < H, base * 4u, 0xFFFFFFFFu);
                }
                float mk[16];
                if (MASK) {        // the 16 mask values are requested before the partial sums are formed
#pragma unroll
                    for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int b = 0; b < 2; ++b)
                                mk[r * 4 + aa * 2 + b] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsm, voffs[aa], (2 * r + b) * (int)a.ppx * Cout * 4, 0));
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float own[4];
                    partial(HH, w, hh * (NBW / 2) + i, r, own);
                    const float4 o = *(const float4*)&Xi[((w * (NBW / 2) + i) * 4 + r) * 256];
                    if (plain) {       // uniform: an input-gradient launch (no bias, no ReLU) skips 3 of 4 VALU per output
                        yv[r * 4 + 0] = own[0] + o.x; yv[r * 4 + 1] = own[1] + o.y;
                        yv[r * 4 + 2] = own[2] + o.z; yv[r * 4 + 3] = own[3] + o.w;
                    } else {
                        yv[r * 4 + 0] = fmaxf((own[0] + o.x) + bvv[i], lo);
                        yv[r * 4 + 1] = fmaxf((own[1] + o.y) + bvv[i], lo);
                        yv[r * 4 + 2] = fmaxf((own[2] + o.z) + bvv[i], lo);
                        yv[r * 4 + 3] = fmaxf((own[3] + o.w) + bvv[i], lo);
                    }
                }
                if (EPI == 1) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) yv[k] = mk[k] > 0.f ? yv[k] : 0.f;
                } else if (EPI == 2) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) yv[k] = mk[k] + yv[k];
                } else if (EPI == 3) {
                    // the norm's backward sums over this lane's 16 pixels of (image cn, channel co)
                    const unsigned mo = ((unsigned)(cn >> a.n_sh) * (unsigned)Cout + co) * 8u;
                    const float mean = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsr, (int)mo, 0, 0));
                    const float rstd = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsr, (int)mo, 4, 0));
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const float xh = (mk[k] - mean) * rstd;
                        const float gm = (a.in_relu && !(xh > 0.f)) ? 0.f : yv[k];
                        s1 += gm;
                        s2 = __builtin_fmaf(gm, xh, s2);
                    }
                    s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
                    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                    if (w == 0) { st1[i] = s1; st2[i] = s2; }
                    else { st1[i] += s1; st2[i] += s2; }
                }
                if (EPI == 4 && !part0 && co0 - a.split >= a.c1) {      // (uniform) padding couts of a layer widened to the 64-cout tile
                } else if (EPI == 4 && pooled) {       // one value per tile at half resolution: tile r of the lane = low-res column xcol0 / 2 + r
                    const unsigned base = (((unsigned)cn * (unsigned)(H >> 1) + (unsigned)(yrow0 >> 1)) * (unsigned)(W >> 1) + (unsigned)(xcol0 >> 1)) * cs + cc;
                    const int voff = (int)sel_u32(yrow0 < H, base * 4u, 0xFFFFFFFFu);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float t = (yv[r * 4 + 0] + yv[r * 4 + 1]) + (yv[r * 4 + 2] + yv[r * 4 + 3]);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(t), rsy, voff, r * (int)cs * 4, 0);
                    }
                } else {
#pragma unroll
                    for (int aa = 0; aa < 2; ++aa) {
                        const int voff = voffs[aa];
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int b = 0; b < 2; ++b)
                                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yv[r * 4 + aa * 2 + b]), (EPI == 4 && !part0) ? rsy2 : rsy, voff,
                                                                      (2 * r + b) * (int)(a.ppx * cs) * 4, 0);
                    }
                }
                if (EPI != 3 && a.stats) {     // uniform: H % TR == 0 whenever statistics are requested
                    float t1, t2;
                    lane_stats<16>(yv, t1, t2);
                    stat_merge_eq(t1, t2, __shfl_xor(t1, 16, 64), __shfl_xor(t2, 16, 64), 1.f / 32.f);
                    stat_merge_eq(t1, t2, __shfl_xor(t1, 32, 64), __shfl_xor(t2, 32, 64), 1.f / 64.f);
                    if (w == 0) { st1[i] = t1; st2[i] = t2; }
                    else stat_merge_eq(st1[i], st2[i], t1, t2, 1.f / 128.f);          // the wave's two M blocks: 64 pixels each
                }
            }
        }
        if (a.stats && lane < 16) {
#pragma unroll
            for (int i = 0; i < NBW / 2; ++i) {
                float* R = Rs + spar * (NMB * NCO * 2) + (mb * NCO + (hh * (NBW / 2) + i) * 16 + lane) * 2;
                R[0] = st1[i];
                R[1] = st2[i];
            }
        }
    };
    auto fold_stats = [&]() {          // after the barrier that follows epi_finish
        if (!a.stats) return;          // uniform
        if (tid < NCO) {
            const float* R = Rs + spar * (NMB * NCO * 2) + tid * 2;
            float s1 = R[0], s2 = R[1];              // the NMB waves' 64 MBW pixels each, merged in order
            constexpr float PW = 64.f * MBW;
            if (EPI == 3) {
#pragma unroll
                for (int r = 1; r < NMB; ++r) { s1 += R[r * NCO * 2]; s2 += R[r * NCO * 2 + 1]; }
            } else {
#pragma unroll
                for (int r = 1; r < NMB; ++r) stat_merge(s1, s2, PW * r, R[r * NCO * 2], R[r * NCO * 2 + 1], PW);
            }
            const int t = (cn * a.tilesX + ctx) * a.tilesY + cty;
            float* o = a.stats + ((size_t)t * Cout + co_base + tid) * 2;
            o[0] = s1;
            o[1] = s2;
        }
        spar ^= 1;
    };

    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    auto shift = [&]() {               // item i+1 becomes the current item
        cn = n1; ctx = tx1; cty = ty1; ch = ch1;
        n1 = n2; tx1 = tx2; ty1 = ty2; ch1 = ch2;
        advance(n2, tx2, ty2, ch2);
    };
    // nch is even: a region is nch / 2 (even, odd) item pairs; the first pair starts the accumulators from the MFMA's inline 0.
    // (Explicit region / pair loops rather than one item loop with an `if (first)` diamond: with 128 accumulators live the
    // diamond's PHIs cost register copies and spills.)  Every branch is uniform per workgroup.
    for (int reg = 0; reg < my_tiles; ++reg) {
        body(P0{}, std::true_type{});
        W6_ITEM_BARRIER();             // publishes the halo of item i+2 and the U chunk of item i+1
        shift();
        body(P1{}, std::false_type{});
        W6_ITEM_BARRIER();
        for (int c = 2; c < nch; c += 2) {
            shift();
            if (ch2 == 0) region_offsets(n2, tx2, ty2);   // item i+2 opens a region: its halo offsets
            body(P0{}, std::false_type{});
            W6_ITEM_BARRIER();
            shift();
            body(P1{}, std::false_type{});
            W6_ITEM_BARRIER();
        }
        // region (cn, ctx, cty) is complete
        flush();
#ifdef W6_EXP_NO_EPI
        if (acc[3][0][1][2] == 123.456f)
#endif
        {
        if (h == 0) epi_send(P0{}); else epi_send(P1{});
        __syncthreads();
        if (h == 0) epi_finish(P0{}); else epi_finish(P1{});
        __syncthreads();               // the exchange area is U buffer 1 again from the next item on
        fold_stats();
        }
        shift();
        if (ch2 == 0) region_offsets(n2, tx2, ty2);       // item i+2 opens a region: its halo offsets
    }
}

}  // namespace

// Which workgroup shape serves a layer (0: none - the 32-cout kernel of conv_wino.hip): 64 couts whenever Cout allows, else
// two M blocks x 32 couts on maps whose width is a multiple of 32.  (Cin % 16: an even number of 8-channel chunks.)
static int wino64_shape(int Cin, int Cout, int W) {
    if (!g_w64_env || Cin % 16 != 0) return 0;
    if (Cout % 64 == 0) return 1;
    return (Cout % 32 == 0 && W % 32 == 0) ? 2 : 0;
}
// Waves per workgroup of a shape: the 32-cout shape (MBW = 2) runs as two four-wave workgroups per CU (VQW_WINO64_NW4=0: one
// eight-wave workgroup, the round-3 form)
static const int g_w64_nw4 = env_int64("VQW_WINO64_NW4", 1);
static int wino64_waves(int shape) { return (shape == 2 && g_w64_nw4) ? 4 : 8; }
bool conv_wino64_ok(int Cin, int Cout, int W) { return wino64_shape(Cin, Cout, W) != 0; }
int conv_wino64_stat_tiles(int Cin, int Cout, int H, int W) {
    const int shape = wino64_shape(Cin, Cout, W);
    const int rw = W % 32 == 0 ? 32 : 16, tr = rw == 32 ? shape * wino64_waves(shape) : 16;
    return tr > 0 && H % tr == 0 ? (H / tr) * (W / rw) : 0;
}

template <int RW, int MBW, int EPI, int NW = 8>
static int launch_wino64(W64Args& a, hipStream_t st) {
    using G = W64Geo<RW, MBW, NW>;
    constexpr size_t lds = G::LDS_FLOATS * sizeof(float);
    static_assert(lds * (8 / NW) <= 160 * 1024, "buffers do not fit the 160 KB LDS");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_conv_wino64<RW, MBW, EPI, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            vqw_set_error("conv_wino64: cannot raise the dynamic LDS limit");
            return VQW_ERR_HIP;
        }
        attr_set = true;
    }
    a.tilesY = ceil_div(a.H, G::TR); a.tilesX = a.W / RW; a.nsp = a.N * a.tilesY * a.tilesX;
    a.ntn = a.Cout / G::NCO;
    int groups = g_w64_max_blocks * (8 / NW) / a.ntn;
    if (groups < 1) groups = 1;
    const int even = ceil_div(a.nsp, groups);
    a.kt = even < 1 ? 1 : even;
    k_conv_wino64<RW, MBW, EPI, NW><<<ceil_div(a.nsp, a.kt) * a.ntn, 64 * NW, lds, st>>>(a);
    VQW_LAUNCH_CHECK("conv_wino64");
    return VQW_OK;
}

// A dilation-2 layer = the plain layer on the four phase images of its tensors (see W64Args): even H, W; W / 2 a multiple of 32
bool conv_wino64_dil2_ok(int Cin, int Cout, int H, int W) {
    return H % 2 == 0 && W % 64 == 0 && wino64_shape(Cin, Cout, W / 2) != 0;
}
int conv_wino64_fwd(const float* x, const float* u, const float* bias, float* y, int N, int H, int W, int Cin, int Cout, int relu,
                    hipStream_t st, float* stats, const float* mask, int accumulate, const float* in_mr, int in_relu, int dil) {
    W64Args a;
    const long P = (long)N * H * W;
    a.img_px = (unsigned)(H * W); a.rowb_px = 0; a.rpx = (unsigned)W; a.ppx = 1; a.n_sh = 0; a.n_m = 0;
    if (dil == 2) {
        a.rowb_px = (unsigned)W; a.rpx = 2u * (unsigned)W; a.ppx = 2; a.n_sh = 2; a.n_m = 1;
        N *= 4; H /= 2; W /= 2;
    }
    a.y2 = nullptr; a.split = 0; a.pool0 = 0; a.c1 = 0; a.nby2 = 0;
    a.x = x; a.u = u; a.bias = bias; a.y = y; a.mask = mask; a.mr = in_mr; a.in_relu = in_relu;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.nch = Cin / 8;
    a.relu = relu;
    a.stats = stats;
    a.nbx = (unsigned)(P * Cin * 4);
    a.nbu = (unsigned)(16L * Cout * Cin * 4);
    a.nby = (unsigned)(P * Cout * 4);
    const int shape = wino64_shape(Cin, Cout, W);
    if (shape == 2 && wino64_waves(shape) == 4) {
        if (in_mr) return launch_wino64<32, 2, 3, 4>(a, st);
        if (mask) return launch_wino64<32, 2, 1, 4>(a, st);
        if (accumulate) return launch_wino64<32, 2, 2, 4>(a, st);
        return launch_wino64<32, 2, 0, 4>(a, st);
    }
    if (in_mr) {               // (mask = the norm's raw input, stats = the backward sums)
        if (shape == 2) return launch_wino64<32, 2, 3>(a, st);
        if (W % 32 == 0) return launch_wino64<32, 1, 3>(a, st);
        return launch_wino64<16, 1, 3>(a, st);
    }
    if (mask) {
        if (shape == 2) return launch_wino64<32, 2, 1>(a, st);
        if (W % 32 == 0) return launch_wino64<32, 1, 1>(a, st);
        return launch_wino64<16, 1, 1>(a, st);
    }
    if (accumulate) {
        if (shape == 2) return launch_wino64<32, 2, 2>(a, st);
        if (W % 32 == 0) return launch_wino64<32, 1, 2>(a, st);
        return launch_wino64<16, 1, 2>(a, st);
    }
    if (shape == 2) return launch_wino64<32, 2, 0>(a, st);
    if (W % 32 == 0) return launch_wino64<32, 1, 0>(a, st);
    return launch_wino64<16, 1, 0>(a, st);
}

// Two output tensors from one launch (EPI 4): couts [0, split) -> y0 (pool0: summed over each 2 x 2 tile, at half resolution),
// [split, Cout) -> y1.  Serves (i) the input gradient of a 3x3 layer over [up2x(a) | b] - the gradients of a and of b come
// out of the epilogue instead of out of two gather passes over the concatenated gradient - and (ii) two layers of one input
// as one launch on concatenated weights.  split % 16 == 0 (an N block never straddles the two tensors).
// c1 = channels of the second tensor (0: all the couts behind `split`); couts in [split + c1, Cout) are padding
bool conv_wino64_split_ok(int Cin, int Cout, int split, int pool0, int N, int H, int W, int c1) {
    if (c1 < 0 || c1 % 16 != 0 || split + c1 > Cout) return false;
    if (!conv_wino64_ok(Cin, Cout, W) || g_wino_mode != 0 || split <= 0 || split >= Cout || split % 16 != 0) return false;
    if (pool0 && (H % 2 != 0)) return false;
    return (long)N * H * W * (Cin > Cout ? Cin : Cout) * 4 <= 0xFFFFFFE0L;
}
int conv_wino64_fwd_split(const float* x, const float* u, const float* bias, float* y0, float* y1, int N, int H, int W, int Cin, int Cout,
                          int split, int pool0, int relu, hipStream_t st, int c1) {
    if (c1 <= 0) c1 = Cout - split;
    W64Args a;
    a.img_px = (unsigned)(H * W); a.rowb_px = 0; a.rpx = (unsigned)W; a.ppx = 1; a.n_sh = 0; a.n_m = 0;
    a.x = x; a.u = u; a.bias = bias; a.y = y0; a.y2 = y1; a.mask = nullptr; a.mr = nullptr; a.in_relu = 0;
    a.split = split; a.pool0 = pool0; a.c1 = c1;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.nch = Cin / 8;
    a.relu = relu;
    a.stats = nullptr;
    const long P = (long)N * H * W;
    a.nbx = (unsigned)(P * Cin * 4);
    a.nbu = (unsigned)(16L * Cout * Cin * 4);
    a.nby = (unsigned)((pool0 ? P / 4 : P) * split * 4);
    a.nby2 = (unsigned)(P * c1 * 4);
    const int shape = wino64_shape(Cin, Cout, W);
    if (shape == 2) return wino64_waves(shape) == 4 ? launch_wino64<32, 2, 4, 4>(a, st) : launch_wino64<32, 2, 4>(a, st);
    if (W % 32 == 0) return launch_wino64<32, 1, 4>(a, st);
    return launch_wino64<16, 1, 4>(a, st);
}

// =====================================================================================================================
// Weight gradient in Winograd form for Cout % 64 == 0, Cin % 32 == 0 (one full-resolution source)
// =====================================================================================================================
//   dU[xi][co][ci] = sum over tiles of dM[xi][tile][co] V[xi][tile][ci],   dM = A dY A^T,  V = B^T d B,   dW = G^T dU G
// Same accounting as above: k_conv_wino_wgrad (conv_wino.hip) spends ~3 VALU instructions per MFMA (both operands are
// transformed by the wave that multiplies them, every prefetch address is computed in the loop).  Here a workgroup owns a
// (64 co x 32 ci) block of dU; wave (r, kh) = xi ROW r (4 of the 16 xi) x all 64 x 32 entries (4 x 4 x 2 x 4 = 128
// accumulators) x half of the region's tiles as its K slice.  Per k-step of 4 tiles a lane builds row r of dM for its co in
// 4 instructions per co block (the row of A dY is one fma, the columns p0, p0 + p1, p0 - p1, p1 two more; signs of row 3 /
// column 3 are folded into the final transform) and row r of V for its ci in 8 per ci block: 32 VALU for 32 MFMAs.
// Operands are built one k-step ahead of the MFMAs that consume them, across the region barrier too (one barrier per
// 128 MFMAs); prefetch addresses are per-thread constants against a buffer descriptor whose base moves with the region
// (SALU), image-edge padding costs three VALU instructions per slot and region (edge bits of the slot & the region's flags).
namespace {

struct W64WgArgs {
    const float* x;
    const float* dy;
    float* part;               // [nsb][Cout][9][Cin]
    float* bias_part;          // [nsb][Cout] or null
    int N, H, W, Cin, Cout;
    int tilesY, tilesX, nsp;
    int n_ci_b, nblk, kt;
    unsigned nbx, nbd;         // bytes of x / dy
    // two sources [up2x?(x) | x1] (UpBlock's first convolution): ci blocks [0, C0) read x (C0 channels per pixel, at half
    // resolution when up0), the others x1 (Cin - C0 channels); a ci block never straddles the two (C0 % 32 == 0).  C0 == Cin:
    // one full-resolution source.
    const float* x1;
    int C0, up0;
    unsigned nbx1;
    // pixel index of (n, y, x) as in W64Args (k_conv_wino_wgrad32 only): dense, or the phase images of a dilation-2 layer
    unsigned img_px, rowb_px, rpx, ppx;
    int n_sh, n_m;
};

#ifndef W6_WLS
#define W6_WLS 2                           // MFMA positions between two prefetch loads of the weight-gradient kernel
#endif
constexpr int WG_DP = 72, WG_XP = 40;     // floats per dY pixel (64 co + 8) / X pixel (32 ci + 8): adjacent tiles 16 banks apart
template <int RW> struct WgGeo {
    static constexpr int TRP = RW == 32 ? 4 : 8;           // pixel rows per region: 32 tiles
    static constexpr int XW = RW + 2, XR = TRP + 2, XPIX = XR * XW;
    static constexpr int DBUF = 128 * WG_DP, XBUF = XPIX * WG_XP;
};

#ifdef W6_WG_VGPR                            // experiment: cap the weight-gradient kernel's registers (room for other streams' waves)
#define W6_WG_ATTR __attribute__((amdgpu_num_vgpr(W6_WG_VGPR)))
#else
#define W6_WG_ATTR
#endif
template <int RW>
__global__ void __launch_bounds__(512, 1) W6_WG_ATTR k_conv_wino_wgrad64(W64WgArgs a) {
    using G = WgGeo<RW>;
    constexpr int NT = 512, DBUF = G::DBUF, XBUF = G::XBUF, XW = G::XW;
    constexpr int XF = G::XPIX * 8;                        // float4 per X halo: 1632 / 1440
    constexpr int LX = (XF + NT - 1) / NT;                 // 4 / 3 slots
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // layout: dY tiles [2][DBUF], then X halos [2][XBUF]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = wv & 3, kh = wv >> 2;
#if defined(W6_PRIO) && W6_PRIO == 1
    if (wv >= 4) __builtin_amdgcn_s_setprio(1);
#elif defined(W6_PRIO) && W6_PRIO == 2
    if (wv < 4) __builtin_amdgcn_s_setprio(1);
#endif
    const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
    // the (co, ci) blocks of one spatial split read the same dY / X regions: keep them on one XCD's L2
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int blk = lb % a.nblk, sblk = lb / a.nblk;
    const int co_base = (blk / a.n_ci_b) * 64, ci_base = (blk % a.n_ci_b) * 32;
    const int sp0 = sblk * a.kt;
    const int my_tiles = min(a.kt, a.nsp - sp0);
    const int per_img = a.tilesY * a.tilesX;
    const bool do_bias = a.bias_part != nullptr && ci_base == 0;

    // ---- loader slots ----
    // dY float4 f = tid + 512 j -> pixel f / 16, co quad f % 16 = tid & 15
    const int d_px = tid >> 4;                                            // + 32 j
    const unsigned d_fix = ((unsigned)((RW == 32 ? 0 : (d_px >> 4)) * W + (RW == 32 ? d_px : (d_px & 15))) * Cout + co_base + (tid & 15) * 4) * 4u;
    const unsigned d_jstride = (unsigned)((RW == 32 ? 1 : 2) * W) * Cout * 4u;      // 32 pixels further: one / two rows
    const int d_lds = d_px * WG_DP + (tid & 15) * 4;                      // + j * 32 * WG_DP
    // X float4 f = tid + 512 j -> halo pixel f / 8, ci quad f % 8 = tid & 7; slots past the end repeat another thread's
    // the source of this workgroup's ci block (uniform): its pointer, channels per pixel, first channel, row length
    const bool x_s1 = ci_base >= a.C0;
    const bool x_up = !x_s1 && a.up0;
    const float* x_ptr = x_s1 ? a.x1 : a.x;
    const int x_cs = x_s1 ? Cin - a.C0 : a.C0, x_cb = x_s1 ? ci_base - a.C0 : ci_base;
    const int x_w = x_up ? W >> 1 : W, x_h = x_up ? H >> 1 : H;
    const long x_nb = x_s1 ? (long)a.nbx1 : (long)a.nbx;
    unsigned x_fix[LX];
    int x_lds[LX];
    unsigned x_bits = 0;                   // 4 bits per slot: the pixel lies on the halo's top / bottom row, left / right column
#pragma unroll
    for (int j = 0; j < LX; ++j) {
        int f = tid + j * NT;
        if (f >= XF) f -= XF;
        const int hp = f >> 3, hy = hp / XW, hx = hp - hy * XW;
        // against the pixel (y0 - 1, x0 - 1); an up-sampled source: against the low-resolution pixel (y0 / 2 - 1, x0 / 2 - 1)
        x_fix[j] = ((unsigned)((x_up ? (hy + 1) >> 1 : hy) * x_w + (x_up ? (hx + 1) >> 1 : hx)) * x_cs + x_cb + (tid & 7) * 4) * 4u;
        x_lds[j] = 2 * DBUF + hp * WG_XP + (tid & 7) * 4;
        x_bits |= (unsigned)((hy == 0 ? 1 : 0) | (hy == G::XR - 1 ? 2 : 0) | (hx == 0 ? 4 : 0) | (hx == XW - 1 ? 8 : 0)) << (4 * j);
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
    // region (n, tx, ty): descriptors whose base is the region's first dY pixel / the pixel (y0 - 1, x0 - 1) of X.  Offsets
    // are range-checked against the rest of the tensor; what lies before its start is masked by the lane masks.
    __amdgpu_buffer_rsrc_t rsd, rsx;
    unsigned x_edges = 0;               // the region's edge flags (top, bottom, left, right), once per slot nibble
    auto region_setup = [&](int n, int tx, int ty) {
        const int y0 = ty * G::TRP, x0 = tx * RW;
        const long dpix = ((long)n * H + y0) * W + x0;
        const long doff = dpix * Cout * 4;
        rsd = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.dy + doff), 0, (int)(unsigned)((long)a.nbd - doff), 0x00020000);
        const long xpix = x_up ? ((long)n * x_h + (y0 >> 1) - 1) * x_w + (x0 >> 1) - 1 : dpix - W - 1;
        const long xoff = xpix * x_cs * 4;                                // may lie before the tensor (first region): masked
        const long xleft = x_nb - xoff;
        rsx = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)x_ptr + xoff), 0, (int)(unsigned)(xleft > 0xFFFFFFF0L ? 0xFFFFFFF0L : xleft), 0x00020000);
        x_edges = ((y0 == 0 ? 1u : 0u) | (y0 + G::TRP == H ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + RW == W ? 8u : 0u)) * 0x1111u;
    };
    auto issue_d = [&](int j) { rd[j] = ld4(rsd, d_fix, j * d_jstride); };
    auto issue_x = [&](int j) {
        const unsigned vo = (x_bits & x_edges & (0xFu << (4 * j))) ? 0xFFFFFFFFu : x_fix[j];      // padding: out of range -> 0
        rx[j] = ld4(rsx, vo, 0);
    };
    float once_v = 1.f;                 // 0 for the prefetch past the workgroup's share (a repeat of its last region)
    auto commit_d = [&](int j, int buf) {
        *(float4*)&smem[d_lds + buf * DBUF + j * 32 * WG_DP] = rd[j];
        if (do_bias) {                  // uniform; fused bias gradient
            bsum.x = __builtin_fmaf(once_v, rd[j].x, bsum.x); bsum.y = __builtin_fmaf(once_v, rd[j].y, bsum.y);
            bsum.z = __builtin_fmaf(once_v, rd[j].z, bsum.z); bsum.w = __builtin_fmaf(once_v, rd[j].w, bsum.w);
        }
    };
    auto commit_x = [&](int j, int buf) { *(float4*)&smem[x_lds[j] + buf * XBUF] = rx[j]; };

    // ---- fragment addressing: lane (channel idx = lane & 15, tile k = lane >> 4 of the k-step) ----
    const int idx = lane & 15, k = lane >> 4;
    // k-step s of K half kh: RW = 32: tile row kh, tile columns 4 s + k;  RW = 16: tile row 2 kh + (s >> 1), columns 4 (s & 1) + k
    // rows of A dY: r = 0: y0.; 1: y0. + y1.; 2: y0. - y1.; 3: y1. (negated: folded)  ->  p = yF + sA * yS
    const int aF = r == 3 ? 1 : 0, aS = r == 0 ? 0 : 1;
    const int trow0 = RW == 32 ? kh : 2 * kh;
    auto opaque = [](int x) { asm volatile("" : "+v"(x)); return x; };
    const int a_F = opaque(((2 * trow0 + aF) * RW + 2 * k) * WG_DP + idx);       // + s-part, + b * WG_DP, + mbk * 16
    const int a_S = opaque(((2 * trow0 + aS) * RW + 2 * k) * WG_DP + idx);
    // rows of B^T d: r = 0: d0 - d2; 1: d1 + d2; 2: d2 - d1; 3: d1 - d3  ->  e = dA + sB * dB
    const int iA = r == 0 ? 0 : r == 2 ? 2 : 1, iB = r == 0 ? 2 : r == 1 ? 2 : r == 2 ? 1 : 3;
    // (the X buffers start 72 KB into the LDS, beyond the 64 KB immediate of a ds_read: their offset lives in the base register)
    const int x_A = opaque(2 * DBUF + ((2 * trow0 + iA) * XW + 2 * k) * WG_XP + idx);       // + s-part, + column * WG_XP, + nbk * 16
    const int x_B = opaque(2 * DBUF + ((2 * trow0 + iB) * XW + 2 * k) * WG_XP + idx);
    float sA, sB;
    { float s = (r == 1) ? 1.f : (r == 2) ? -1.f : 0.f; asm volatile("v_mov_b32 %0, %1" : "=v"(sA) : "v"(s)); }
    { float s = (r == 1) ? 1.f : -1.f; asm volatile("v_mov_b32 %0, %1" : "=v"(sB) : "v"(s)); }
    // pixel offset of k-step s inside the region's dY tile / X halo
    auto s_px_d = [](int s) { return RW == 32 ? 8 * s : 2 * (s >> 1) * 16 + 8 * (s & 1); };
    auto s_px_x = [](int s) { return RW == 32 ? 8 * s : 2 * (s >> 1) * XW + 8 * (s & 1); };

    f32x4 acc[4][4][2];                // [xi column][co block][ci block]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[c][mb][nb][q] = 0.f;
    float dm[2][4][4];                 // [operand set][co block][xi column]
    float vv[2][2][4];                 // [operand set][ci block][xi column]
    float ra[2][4], rb[2][8];          // raw LDS values of two groups in flight

    // Operand build for k-step s from the buffers (Db, Xb), as six groups: g = 0..3 co block g (4 reads, 4 VALU),
    // g = 4, 5 ci block g - 4 (8 reads, 8 VALU)
    auto rd_grp = [&](int nbuf, int s, int g, int i) {       // read i of group g from buffer nbuf
        if (g < 4) {
            const int o = nbuf * DBUF + s_px_d(s) * WG_DP + g * 16 + (i & 1) * WG_DP;
            ra[g & 1][i] = i < 2 ? smem[a_F + o] : smem[a_S + o];
        } else {
            const int o = nbuf * XBUF + s_px_x(s) * WG_XP + (g - 4) * 16 + (i & 3) * WG_XP;
            rb[g & 1][i] = i < 4 ? smem[x_A + o] : smem[x_B + o];
        }
    };
    auto op_grp = [&](int set, int g, int i) {          // VALU operation i of group g
        if (g < 4) {
            float* d = dm[set][g];
            const float* q = ra[g & 1];
            if (i == 0) d[0] = __builtin_fmaf(sA, q[2], q[0]);          // p0
            if (i == 1) d[3] = __builtin_fmaf(sA, q[3], q[1]);          // p1 (column 3 = -p1: folded)
            if (i == 2) d[1] = d[0] + d[3];
            if (i == 3) d[2] = d[0] - d[3];
        } else {
            float* o = vv[set][g - 4];
            float* q = rb[g & 1];
            if (i < 4) q[i] = __builtin_fmaf(sB, q[4 + i], q[i]);       // e[i] in place
            if (i == 4) o[0] = q[0] - q[2];
            if (i == 5) o[1] = q[1] + q[2];
            if (i == 6) o[2] = q[2] - q[1];
            if (i == 7) o[3] = q[1] - q[3];
        }
    };
    // position p = 0..31 of a phase -> what is built beside MFMA p:
    //   p 0-3 reads A0 | 4-7 ops A0, reads A1 | 8-11 ops A1, reads A2 | 12-15 ops A2, reads A3 | 16-19 ops A3, reads B0 (2 each)
    //   | 20-27 ops B0, reads B1 | 28-31 ops B1 (2 each)
    auto build_slot = [&](int nbuf, int s, int set, int p) {
        if (p < 4) rd_grp(nbuf, s, 0, p);
        else if (p < 16) { op_grp(set, (p - 4) >> 2, (p - 4) & 3); rd_grp(nbuf, s, ((p - 4) >> 2) + 1, (p - 4) & 3); }
        else if (p < 20) { op_grp(set, 3, p - 16); rd_grp(nbuf, s, 4, 2 * (p - 16)); rd_grp(nbuf, s, 4, 2 * (p - 16) + 1); }
        else if (p < 28) { op_grp(set, 4, p - 20); rd_grp(nbuf, s, 5, p - 20); }
        else { op_grp(set, 5, 2 * (p - 28)); op_grp(set, 5, 2 * (p - 28) + 1); }
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
    if (my_tiles > 0) {
        // prologue: region 0 into buffer 0, operands of its first k-step; the dY loads of region 1 in flight
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
        for (int p = 0; p < 32; ++p) build_slot(0, 0, 0, p);
    }
    // the region whose data is fetched next (clamped to the workgroup's share: past it the last region is fetched again)
    int ln = cn, ltx = ctx, lty = cty, lcount = 0;
    auto load_advance = [&]() {
        if (lcount + 1 < my_tiles) { next_region(ln, ltx, lty); ++lcount; once_v = 1.f; } else once_v = 0.f;
        region_setup(ln, ltx, lty);
    };
    if (my_tiles > 0) {
        load_advance();
#pragma unroll
        for (int j = 0; j < 4; ++j) issue_d(j);
    }

    // One phase = the 32 MFMAs of k-step s (operand set SET) + the build of the next k-step's operands (set SET ^ 1) from
    // buffer NB + what LOADS says: 0 commit dY, 1 issue X, 2 commit X (all of the next region), 3 issue dY of the one after it
    auto phase = [&](auto SET, auto SNEXT, auto NBUF, auto DBUFW, auto LOADS) {
        constexpr int set = decltype(SET)::value, sn = decltype(SNEXT)::value, nbuf = decltype(NBUF)::value;
        constexpr int wbuf = decltype(DBUFW)::value, loads = decltype(LOADS)::value;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const int p = c * 8 + mb * 2 + nb;
                    acc[c][mb][nb] = MFMA16(dm[set][mb][c], vv[set][nb][c], acc[c][mb][nb]);
#ifndef W6_EXP_NO_BUILD               // timing-only A/B builds (tools/wino_ab.sh): wrong results
                    build_slot(nbuf, sn, set ^ 1, p);
#endif
                    // prefetch: a phase (1 us) between a load and its LDS commit (twice that changed nothing and costs registers)
#ifndef W6_EXP_NO_WLOADS
                    if (loads == 0 && p >= 8 && p < 12) commit_d(p - 8, wbuf);
                    if (loads == 1 && p >= 2 && p < 2 + W6_WLS * LX && (p - 2) % W6_WLS == 0) issue_x((p - 2) / W6_WLS);
                    if (loads == 2 && p >= 8 && p < 8 + LX) commit_x(p - 8, wbuf);
                    if (loads == 3 && p >= 2 && p < 2 + W6_WLS * 4 && (p - 2) % W6_WLS == 0) issue_d((p - 2) / W6_WLS);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    auto region = [&](auto BUF) {      // the four k-steps of a region that sits in buffer BUF
        constexpr int b = decltype(BUF)::value;
        using B = std::integral_constant<int, b>;
        using NB = std::integral_constant<int, b ^ 1>;
        phase(I0{}, I1{}, B{}, NB{}, I0{});        // k-step 0 (set 0), builds k-step 1; commits dY of the next region
        phase(I1{}, I2{}, B{}, NB{}, I1{});        // k-step 1, builds 2; issues X of the next region
        phase(I0{}, I3{}, B{}, NB{}, I2{});        // k-step 2, builds 3; commits X
#ifndef W6_EXP_NO_WBARRIER
        __syncthreads();                           // the next region's buffers are complete; this region's are read once more
#endif
        load_advance();
        phase(I1{}, I0{}, NB{}, B{}, I3{});        // k-step 3, builds k-step 0 of the next region; issues dY of the one after
    };
    for (int g = 0; g < my_tiles; g += 2) {        // uniform per workgroup
        region(I0{});
        if (g + 1 < my_tiles) region(I1{});
    }

    // ---- fold: the two K halves and the four xi rows meet in LDS, one co block (16 couts) per round ----
    __syncthreads();
    float* red = smem;                     // [kh][r][c][16 co][32 ci] = 64 KB
    if (do_bias) {                         // threads with equal (tid & 15) hold the same 4 couts
        *(float4*)&red[tid * 4] = bsum;
        __syncthreads();
        if (tid < 64) {
            const int cq = tid >> 2, comp = tid & 3;
            float sum = 0.f;
            for (int i = 0; i < 32; ++i) sum += red[(i * 16 + cq) * 4 + comp];
            a.bias_part[(size_t)sblk * Cout + co_base + tid] = sum;
        }
        __syncthreads();
    }
    // (one base register made opaque HERE: left to itself the compiler computes the 128 store addresses at kernel entry and
    // carries them, spilled, through the main loop)
    int fold_w = ((kh * 4 + r) * 4 * 16 + 4 * (lane >> 4)) * 32 + idx, fold_r = tid;
    asm volatile("" : "+v"(fold_w), "+v"(fold_r));
#ifdef W6_EXP_NO_FOLD
    if (acc[1][2][1][3] == 123.456f)
#endif
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        // C/D layout (16x16): col = lane & 15 (ci), row = 4 (lane >> 4) + q (co within the block)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    red[fold_w + c * 512 + q * 32 + nb * 16] = acc[c][mb][nb][q];
        __syncthreads();
        {
            const int e = fold_r;          // (co16, ci32) = (tid >> 5, tid & 31)
            float tcol[3][4];              // G^T applied down the rows: [ky][column]
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float u[4];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const float t = red[((0 * 4 + rr) * 4 + c) * 512 + e] + red[((1 * 4 + rr) * 4 + c) * 512 + e];
                    u[rr] = ((rr == 3) != (c == 3)) ? -t : t;           // row 3 / column 3 of dM were built without their sign
                }
                tcol[0][c] = u[0] + 0.5f * (u[1] + u[2]);
                tcol[1][c] = 0.5f * (u[1] - u[2]);
                tcol[2][c] = u[3] + 0.5f * (u[1] + u[2]);
            }
            const int co = co_base + mb * 16 + (tid >> 5), ci = ci_base + (tid & 31);
            float* o = a.part + (size_t)sblk * Cout * 9 * Cin + ((size_t)co * 9) * Cin + ci;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float t0 = tcol[ky][0], t1 = tcol[ky][1], t2 = tcol[ky][2], t3 = tcol[ky][3];
                o[(ky * 3 + 0) * Cin] = t0 + 0.5f * (t1 + t2);
                o[(ky * 3 + 1) * Cin] = 0.5f * (t1 - t2);
                o[(ky * 3 + 2) * Cin] = t3 + 0.5f * (t1 + t2);
            }
        }
        __syncthreads();
    }
}


// =====================================================================================================================
// The same weight-gradient structure for layers whose Cout is a multiple of 32 but not of 64 (round 4): a workgroup owns a
// (32 co x 32 ci) block of dU.  Wave (r, kh) as above; a k-step is 16 MFMAs (4 xi columns x 2 co blocks x 2 ci blocks, 64
// accumulators) and its operand build 2 x 4 + 2 x 8 = 24 VALU = 1.5 per MFMA (the 64 x 32 block: 1.0; k_conv_wino_wgrad of
// conv_wino.hip, which these layers ran on: ~3 with its in-loop address arithmetic).  Two VALU operations and two LDS reads per
// MFMA position instead of one; everything else - region walk, descriptors that move with the region, edge bits, fold - is
// the 64-cout kernel's.  32-pixel-wide regions only (the layers concerned live on the 128- and 256-pixel levels).
constexpr int WH_DP = 40, WH_XP = 40;      // floats per dY pixel (32 co + 8) / X pixel (32 ci + 8): adjacent tiles 16 banks apart
struct WhGeo {
    static constexpr int RW = 32, TRP = 4;
    static constexpr int XW = RW + 2, XR = TRP + 2, XPIX = XR * XW;
    static constexpr int DBUF = 128 * WH_DP, XBUF = XPIX * WH_XP;
};

__global__ void __launch_bounds__(512, 1) k_conv_wino_wgrad32(W64WgArgs a) {
    using G = WhGeo;
    constexpr int RW = 32, NT = 512, DBUF = G::DBUF, XBUF = G::XBUF, XW = G::XW;
    constexpr int XF = G::XPIX * 8;                        // float4 per X halo: 1632
    constexpr int LX = (XF + NT - 1) / NT;                 // 4 slots
    constexpr int LD = 2;                                  // dY: 128 pixels x 8 float4 = 1024 = 2 slots
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // layout: dY tiles [2][DBUF], then X halos [2][XBUF]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = wv & 3, kh = wv >> 2;
    const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int blk = lb % a.nblk, sblk = lb / a.nblk;
    const int co_base = (blk / a.n_ci_b) * 32, ci_base = (blk % a.n_ci_b) * 32;
    const int sp0 = sblk * a.kt;
    const int my_tiles = min(a.kt, a.nsp - sp0);
    const int per_img = a.tilesY * a.tilesX;
    const bool do_bias = a.bias_part != nullptr && ci_base == 0;

    // ---- loader slots ----
    // dY float4 f = tid + 512 j -> pixel f / 8 = (tid >> 3) + 64 j (two pixel rows per slot), co quad tid & 7
    const int d_px = tid >> 3;
    const unsigned d_fix = (((unsigned)(d_px >> 5) * a.rpx + (unsigned)(d_px & 31) * a.ppx) * Cout + co_base + (tid & 7) * 4) * 4u;
    const unsigned d_jstride = 2u * a.rpx * Cout * 4u;
    const int d_lds = d_px * WH_DP + (tid & 7) * 4;                       // + j * 64 * WH_DP
    // the source of this workgroup's ci block (uniform): its pointer, channels per pixel, first channel, row length
    const bool x_s1 = ci_base >= a.C0;
    const bool x_up = !x_s1 && a.up0;
    const float* x_ptr = x_s1 ? a.x1 : a.x;
    const int x_cs = x_s1 ? Cin - a.C0 : a.C0, x_cb = x_s1 ? ci_base - a.C0 : ci_base;
    const int x_w = x_up ? W >> 1 : W, x_h = x_up ? H >> 1 : H;
    const long x_nb = x_s1 ? (long)a.nbx1 : (long)a.nbx;
    unsigned x_fix[LX];
    int x_lds[LX];
    unsigned x_bits = 0;
#pragma unroll
    for (int j = 0; j < LX; ++j) {
        int f = tid + j * NT;
        if (f >= XF) f -= XF;
        const int hp = f >> 3, hy = hp / XW, hx = hp - hy * XW;
        const unsigned xpx = x_up ? (unsigned)(((hy + 1) >> 1) * x_w + ((hx + 1) >> 1)) : (unsigned)hy * a.rpx + (unsigned)hx * a.ppx;
        x_fix[j] = (xpx * x_cs + x_cb + (tid & 7) * 4) * 4u;
        x_lds[j] = 2 * DBUF + hp * WH_XP + (tid & 7) * 4;
        x_bits |= (unsigned)((hy == 0 ? 1 : 0) | (hy == G::XR - 1 ? 2 : 0) | (hx == 0 ? 4 : 0) | (hx == XW - 1 ? 8 : 0)) << (4 * j);
    }
    float4 rd[LD], rx[LX];
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
    auto region_setup = [&](int n, int tx, int ty) {
        const int y0 = ty * G::TRP, x0 = tx * RW;
        const long nbase = (long)(n >> a.n_sh) * a.img_px + (long)((n >> 1) & a.n_m) * a.rowb_px + (n & a.n_m);
        const long dpix = nbase + (long)y0 * a.rpx + (long)x0 * a.ppx;
        const long doff = dpix * Cout * 4;
        rsd = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.dy + doff), 0, (int)(unsigned)((long)a.nbd - doff), 0x00020000);
        const long xpix = x_up ? ((long)n * x_h + (y0 >> 1) - 1) * x_w + (x0 >> 1) - 1 : dpix - (long)a.rpx - (long)a.ppx;
        const long xoff = xpix * x_cs * 4;
        const long xleft = x_nb - xoff;
        rsx = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)x_ptr + xoff), 0, (int)(unsigned)(xleft > 0xFFFFFFF0L ? 0xFFFFFFF0L : xleft), 0x00020000);
        x_edges = ((y0 == 0 ? 1u : 0u) | (y0 + G::TRP == H ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + RW == W ? 8u : 0u)) * 0x1111u;
    };
    auto issue_d = [&](int j) { rd[j] = ld4(rsd, d_fix, j * d_jstride); };
    auto issue_x = [&](int j) {
        const unsigned vo = (x_bits & x_edges & (0xFu << (4 * j))) ? 0xFFFFFFFFu : x_fix[j];
        rx[j] = ld4(rsx, vo, 0);
    };
    float once_v = 1.f;
    auto commit_d = [&](int j, int buf) {
        *(float4*)&smem[d_lds + buf * DBUF + j * 64 * WH_DP] = rd[j];
        if (do_bias) {
            bsum.x = __builtin_fmaf(once_v, rd[j].x, bsum.x); bsum.y = __builtin_fmaf(once_v, rd[j].y, bsum.y);
            bsum.z = __builtin_fmaf(once_v, rd[j].z, bsum.z); bsum.w = __builtin_fmaf(once_v, rd[j].w, bsum.w);
        }
    };
    auto commit_x = [&](int j, int buf) { *(float4*)&smem[x_lds[j] + buf * XBUF] = rx[j]; };

    // ---- fragment addressing: lane (channel idx = lane & 15, tile k = lane >> 4 of the k-step) ----
    const int idx = lane & 15, k = lane >> 4;
    const int aF = r == 3 ? 1 : 0, aS = r == 0 ? 0 : 1;
    const int trow0 = kh;
    auto opaque = [](int x) { asm volatile("" : "+v"(x)); return x; };
    const int a_F = opaque(((2 * trow0 + aF) * RW + 2 * k) * WH_DP + idx);
    const int a_S = opaque(((2 * trow0 + aS) * RW + 2 * k) * WH_DP + idx);
    const int iA = r == 0 ? 0 : r == 2 ? 2 : 1, iB = r == 0 ? 2 : r == 1 ? 2 : r == 2 ? 1 : 3;
    const int x_A = opaque(2 * DBUF + ((2 * trow0 + iA) * XW + 2 * k) * WH_XP + idx);
    const int x_B = opaque(2 * DBUF + ((2 * trow0 + iB) * XW + 2 * k) * WH_XP + idx);
    float sA, sB;
    { float s = (r == 1) ? 1.f : (r == 2) ? -1.f : 0.f; asm volatile("v_mov_b32 %0, %1" : "=v"(sA) : "v"(s)); }
    { float s = (r == 1) ? 1.f : -1.f; asm volatile("v_mov_b32 %0, %1" : "=v"(sB) : "v"(s)); }

    f32x4 acc[4][2][2];                // [xi column][co block][ci block]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[c][mb][nb][q] = 0.f;
    float dm[2][2][4];                 // [operand set][co block][xi column]
    float vv[2][2][4];                 // [operand set][ci block][xi column]
    float ra[2][4], rb[2][8];          // raw LDS values: ra[co group], rb[ci group]

    // groups of a k-step: g = 0, 1 co block g (4 reads, 4 VALU), g = 2, 3 ci block g - 2 (8 reads, 8 VALU)
    auto rd_grp = [&](int nbuf, int s, int g, int i) {
        if (g < 2) {
            const int o = nbuf * DBUF + 8 * s * WH_DP + g * 16 + (i & 1) * WH_DP;
            ra[g][i] = i < 2 ? smem[a_F + o] : smem[a_S + o];
        } else {
            const int o = nbuf * XBUF + 8 * s * WH_XP + (g - 2) * 16 + (i & 3) * WH_XP;
            rb[g - 2][i] = i < 4 ? smem[x_A + o] : smem[x_B + o];
        }
    };
    auto op_grp = [&](int set, int g, int i) {
        if (g < 2) {
            float* d = dm[set][g];
            const float* q = ra[g];
            if (i == 0) d[0] = __builtin_fmaf(sA, q[2], q[0]);
            if (i == 1) d[3] = __builtin_fmaf(sA, q[3], q[1]);
            if (i == 2) d[1] = d[0] + d[3];
            if (i == 3) d[2] = d[0] - d[3];
        } else {
            float* o = vv[set][g - 2];
            float* q = rb[g - 2];
            if (i < 4) q[i] = __builtin_fmaf(sB, q[4 + i], q[i]);
            if (i == 4) o[0] = q[0] - q[2];
            if (i == 5) o[1] = q[1] + q[2];
            if (i == 6) o[2] = q[2] - q[1];
            if (i == 7) o[3] = q[1] - q[3];
        }
    };
    // position p = 0..15 of a phase -> what is built beside MFMA p (reads land two positions before their first use):
    //   p 0-1 reads A0 | 2 reads A1 | 3 reads A1, ops A0 | 4 ops A0, reads B0 | 5-6 ops A1, reads B0 | 7 reads B0 |
    //   8-11 ops B0, reads B1 | 12-15 ops B1            (two reads / two operations per position)
    auto build_slot = [&](int nbuf, int s, int set, int p) {
        auto rd2 = [&](int g, int i0) { rd_grp(nbuf, s, g, i0); rd_grp(nbuf, s, g, i0 + 1); };
        auto op2 = [&](int g, int i0) { op_grp(set, g, i0); op_grp(set, g, i0 + 1); };
        if (p == 0) rd2(0, 0);
        else if (p == 1) rd2(0, 2);
        else if (p == 2) rd2(1, 0);
        else if (p == 3) { op2(0, 0); rd2(1, 2); }
        else if (p == 4) { op2(0, 2); rd2(2, 0); }
        else if (p == 5) { op2(1, 0); rd2(2, 2); }
        else if (p == 6) { op2(1, 2); rd2(2, 4); }
        else if (p == 7) rd2(2, 6);
        else if (p < 12) { op2(2, 2 * (p - 8)); rd2(3, 2 * (p - 8)); }
        else op2(3, 2 * (p - 12));
    };

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
    if (my_tiles > 0) {
        region_setup(cn, ctx, cty);
#pragma unroll
        for (int j = 0; j < LD; ++j) issue_d(j);
#pragma unroll
        for (int j = 0; j < LX; ++j) issue_x(j);
#pragma unroll
        for (int j = 0; j < LD; ++j) commit_d(j, 0);
#pragma unroll
        for (int j = 0; j < LX; ++j) commit_x(j, 0);
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 16; ++p) build_slot(0, 0, 0, p);
    }
    int ln = cn, ltx = ctx, lty = cty, lcount = 0;
    auto load_advance = [&]() {
        if (lcount + 1 < my_tiles) { next_region(ln, ltx, lty); ++lcount; once_v = 1.f; } else once_v = 0.f;
        region_setup(ln, ltx, lty);
    };
    if (my_tiles > 0) {
        load_advance();
#pragma unroll
        for (int j = 0; j < LD; ++j) issue_d(j);
    }

    // One phase = the 16 MFMAs of k-step s (operand set SET) + the build of the next k-step's operands (set SET ^ 1) from
    // buffer NB + what LOADS says: 0 commit dY, 1 issue X, 2 commit X (all of the next region), 3 issue dY of the one after it
    auto phase = [&](auto SET, auto SNEXT, auto NBUF, auto DBUFW, auto LOADS) {
        constexpr int set = decltype(SET)::value, sn = decltype(SNEXT)::value, nbuf = decltype(NBUF)::value;
        constexpr int wbuf = decltype(DBUFW)::value, loads = decltype(LOADS)::value;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const int p = c * 4 + mb * 2 + nb;
                    acc[c][mb][nb] = MFMA16(dm[set][mb][c], vv[set][nb][c], acc[c][mb][nb]);
                    build_slot(nbuf, sn, set ^ 1, p);
                    if (loads == 0 && p >= 8 && p < 8 + LD) commit_d(p - 8, wbuf);
                    if (loads == 1 && p >= 2 && p < 2 + 2 * LX && (p - 2) % 2 == 0) issue_x((p - 2) / 2);
                    if (loads == 2 && p >= 8 && p < 8 + LX) commit_x(p - 8, wbuf);
                    if (loads == 3 && p >= 2 && p < 2 + 2 * LD && (p - 2) % 2 == 0) issue_d((p - 2) / 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    auto region = [&](auto BUF) {
        constexpr int b = decltype(BUF)::value;
        using B = std::integral_constant<int, b>;
        using NB = std::integral_constant<int, b ^ 1>;
        phase(I0{}, I1{}, B{}, NB{}, I0{});
        phase(I1{}, I2{}, B{}, NB{}, I1{});
        phase(I0{}, I3{}, B{}, NB{}, I2{});
        __syncthreads();
        load_advance();
        phase(I1{}, I0{}, NB{}, B{}, I3{});
    };
    for (int g = 0; g < my_tiles; g += 2) {
        region(I0{});
        if (g + 1 < my_tiles) region(I1{});
    }

    // ---- fold: the two K halves and the four xi rows meet in LDS, one co block (16 couts) per round ----
    __syncthreads();
    float* red = smem;                     // [kh][r][c][16 co][32 ci] = 64 KB
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
    int fold_w = ((kh * 4 + r) * 4 * 16 + 4 * (lane >> 4)) * 32 + idx, fold_r = tid;
    asm volatile("" : "+v"(fold_w), "+v"(fold_r));
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    red[fold_w + c * 512 + q * 32 + nb * 16] = acc[c][mb][nb][q];
        __syncthreads();
        {
            const int e = fold_r;          // (co16, ci32) = (tid >> 5, tid & 31)
            float tcol[3][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float u[4];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const float t = red[((0 * 4 + rr) * 4 + c) * 512 + e] + red[((1 * 4 + rr) * 4 + c) * 512 + e];
                    u[rr] = ((rr == 3) != (c == 3)) ? -t : t;
                }
                tcol[0][c] = u[0] + 0.5f * (u[1] + u[2]);
                tcol[1][c] = 0.5f * (u[1] - u[2]);
                tcol[2][c] = u[3] + 0.5f * (u[1] + u[2]);
            }
            const int co = co_base + mb * 16 + (tid >> 5), ci = ci_base + (tid & 31);
            float* o = a.part + (size_t)sblk * Cout * 9 * Cin + ((size_t)co * 9) * Cin + ci;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float t0 = tcol[ky][0], t1 = tcol[ky][1], t2 = tcol[ky][2], t3 = tcol[ky][3];
                o[(ky * 3 + 0) * Cin] = t0 + 0.5f * (t1 + t2);
                o[(ky * 3 + 1) * Cin] = 0.5f * (t1 - t2);
                o[(ky * 3 + 2) * Cin] = t3 + 0.5f * (t1 + t2);
            }
        }
        __syncthreads();
    }
}

}  // namespace

// one full-resolution source or two sources [up2x?(a) | b] with 32-channel-aligned widths (VQW_WGRAD_TWO_SRC=0: those on
// k_conv_wino_wgrad, A/B timing); whole regions (4 x 32 or 8 x 16 pixels)
static const int g_wg2src_env = env_int64("VQW_WGRAD_TWO_SRC", 1);
static bool wgrad_sources_ok(int C0, int C1, int up0, int H, int W) {
    if (C1 == 0) return !up0 && C0 % 32 == 0;          // (a single up-sampled source has the nine-product kernel)
    return g_wg2src_env && C0 % 32 == 0 && C1 % 32 == 0 && (!up0 || (H % 2 == 0 && W % 2 == 0));
}
// VQW_WGRAD64=0 (experiment): the (64 co x 32 ci)-block kernel off; its 32-pixel-wide layers then take the (32 x 32)-block kernel
static const int g_wg64_env = env_int64("VQW_WGRAD64", 1);
bool conv_wino64_wgrad_ok(int C0, int C1, int up0, int Cout, int H, int W) {
    if (!g_wg64_env && W % 32 == 0 && H % 4 == 0) return false;
    if (!g_w64_env || !wgrad_sources_ok(C0, C1, up0, H, W) || Cout % 64 != 0 || W % 16 != 0) return false;
    return H % (W % 32 == 0 ? 4 : 8) == 0;
}
int conv_wino64_wgrad_blocks(int Cin, int Cout, int N, int H, int W, int max_slabs, int* kt_out) {
    const int nblk = (Cout / 64) * (Cin / 32);
    const int rw = W % 32 == 0 ? 32 : 16;
    const int nsp = N * (H / (128 / rw)) * (W / rw);
    int nsb = g_w64_max_blocks_wg / nblk;
    if (nsb > max_slabs) nsb = max_slabs;
    if (nsb > nsp) nsb = nsp;
    if (nsb < 1) nsb = 1;
    const int kt = ceil_div(nsp, nsb);
    if (kt_out) *kt_out = kt;
    return ceil_div(nsp, kt);
}
static void wgrad_dense(W64WgArgs& a) {
    a.img_px = (unsigned)(a.H * a.W); a.rowb_px = 0; a.rpx = (unsigned)a.W; a.ppx = 1; a.n_sh = 0; a.n_m = 0;
}
static void wgrad_sources(W64WgArgs& a, const ConvIn& in, long P) {
    wgrad_dense(a);
    a.x = in.src0; a.x1 = in.C1 ? in.src1 : in.src0; a.C0 = in.C0; a.up0 = in.C1 ? in.up0 : 0;
    a.nbx = (unsigned)((a.up0 ? P / 4 : P) * in.C0 * 4);
    a.nbx1 = (unsigned)(P * in.C1 * 4);
}
int conv_wino64_wgrad(const ConvIn& in, const float* dy, float* ws, float* bpart, int N, int H, int W, int Cin, int Cout, int nsb, int kt,
                      hipStream_t st) {
    const int rw = W % 32 == 0 ? 32 : 16;
    const size_t lds = (size_t)2 * (rw == 32 ? WgGeo<32>::DBUF + WgGeo<32>::XBUF : WgGeo<16>::DBUF + WgGeo<16>::XBUF) * sizeof(float);
    static_assert((size_t)2 * (WgGeo<32>::DBUF + WgGeo<32>::XBUF) * sizeof(float) <= 160 * 1024, "wgrad tiles do not fit the LDS");
    static_assert((size_t)2 * (WgGeo<16>::DBUF + WgGeo<16>::XBUF) * sizeof(float) >= 64 * 1024, "the fold needs 64 KB");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_conv_wino_wgrad64<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_conv_wino_wgrad64<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            vqw_set_error("conv_wino64_wgrad: cannot raise the dynamic LDS limit");
            return VQW_ERR_HIP;
        }
        attr_set = true;
    }
    const long P = (long)N * H * W;
    W64WgArgs a;
    a.dy = dy; a.part = ws; a.bias_part = bpart;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.tilesY = H / (128 / rw); a.tilesX = W / rw; a.nsp = N * a.tilesY * a.tilesX;
    a.n_ci_b = Cin / 32; a.nblk = (Cout / 64) * a.n_ci_b; a.kt = kt;
    wgrad_sources(a, in, P);
    a.nbd = (unsigned)(P * Cout * 4);
    if (rw == 32) k_conv_wino_wgrad64<32><<<a.nblk * nsb, 512, lds, st>>>(a);
    else k_conv_wino_wgrad64<16><<<a.nblk * nsb, 512, lds, st>>>(a);
    VQW_LAUNCH_CHECK("conv_wino64_wgrad");
    return VQW_OK;
}

// (32 co x 32 ci) blocks: Cout a multiple of 32 but not of 64 (those take the 64-cout kernel), one full-resolution source,
// 32-pixel-wide whole regions (4 x 32 pixels).  VQW_WINOGRAD32W=0: these layers on k_conv_wino_wgrad (A/B timing).
static const int g_w32w_env = env_int64("VQW_WINOGRAD32W", 1);
bool conv_wino32_wgrad_ok(int C0, int C1, int up0, int Cout, int H, int W) {
    if (!g_w64_env || !g_w32w_env || !wgrad_sources_ok(C0, C1, up0, H, W) || Cout % 32 != 0 || (Cout % 64 == 0 && g_wg64_env) || W % 32 != 0) return false;
    return H % 4 == 0;
}
int conv_wino32_wgrad_blocks(int Cin, int Cout, int N, int H, int W, int max_slabs, int* kt_out) {
    const int nblk = (Cout / 32) * (Cin / 32);
    const int nsp = N * (H / 4) * (W / 32);
    int nsb = g_w64_max_blocks_wg / nblk;
    if (nsb > max_slabs) nsb = max_slabs;
    if (nsb > nsp) nsb = nsp;
    if (nsb < 1) nsb = 1;
    const int kt = ceil_div(nsp, nsb);
    if (kt_out) *kt_out = kt;
    return ceil_div(nsp, kt);
}
// the weight gradient of a dilation-2 layer = the plain one summed over the four phase images of x and dy (see W64Args)
bool conv_wino32_wgrad_dil2_ok(int C0, int Cout, int H, int W) {
    return H % 2 == 0 && W % 2 == 0 && conv_wino32_wgrad_ok(C0, 0, 0, Cout, H / 2, W / 2);
}
int conv_wino32_wgrad(const ConvIn& in, const float* dy, float* ws, float* bpart, int N, int H, int W, int Cin, int Cout, int nsb, int kt,
                      hipStream_t st, int dil) {
    constexpr size_t lds = (size_t)2 * (WhGeo::DBUF + WhGeo::XBUF) * sizeof(float);
    static_assert(lds <= 160 * 1024 && lds >= 64 * 1024, "the fold needs 64 KB");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_conv_wino_wgrad32, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            vqw_set_error("conv_wino32_wgrad: cannot raise the dynamic LDS limit");
            return VQW_ERR_HIP;
        }
        attr_set = true;
    }
    const long P = (long)N * H * W;
    W64WgArgs a;
    a.dy = dy; a.part = ws; a.bias_part = bpart;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    wgrad_sources(a, in, P);
    if (dil == 2) {        // (nsb, kt: from conv_wino32_wgrad_blocks(Cin, Cout, 4 N, H / 2, W / 2, ..))
        a.img_px = (unsigned)(H * W); a.rowb_px = (unsigned)W; a.rpx = 2u * (unsigned)W; a.ppx = 2; a.n_sh = 2; a.n_m = 1;
        a.N = 4 * N; a.H = H / 2; a.W = W / 2;
    }
    a.tilesY = a.H / 4; a.tilesX = a.W / 32; a.nsp = a.N * a.tilesY * a.tilesX;
    a.n_ci_b = Cin / 32; a.nblk = (Cout / 32) * a.n_ci_b; a.kt = kt;
    a.nbd = (unsigned)(P * Cout * 4);
    k_conv_wino_wgrad32<<<a.nblk * nsb, 512, lds, st>>>(a);
    VQW_LAUNCH_CHECK("conv_wino32_wgrad");
    return VQW_OK;
}
