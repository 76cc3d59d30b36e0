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

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
static const int g_halo16 = env_int("VQW_HALO16", 1);        // 0: 16-cout layers on the 32-wide tile (A/B)
static const int g_halo_kt = env_int("VQW_HALO_KT", 0);     // tuning aid: spatial tiles per workgroup (0 = default)
// Workgroups of the one-per-CU kernels in this file.  256 = every CU of an MI355X.  A smaller value leaves CUs whose
// LDS is not taken for kernels of other streams that need LDS of their own (e.g. RCCL collectives in data-parallel
// runs, which otherwise wait for one of these kernels to end).
static const int g_max_blocks = []{ int v = env_int("VQW_CONV_MAX_BLOCKS", 256); return v < 8 ? 8 : (v > 256 ? 256 : v); }();

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
    unsigned nb0, nb1, nbw, nby;
    float* stats;              // optional [N][tilesY*tilesX][Cout][2]: per-tile (sum, M2 about the tile mean) of the output
    int py, px;                // NTAP == 4 (collapsed 3x3 over a x2 up-sampled input): output parity of this launch
};

// NTAP = 9: the 3x3 layer.  NTAP = 4: one output parity (py, px) of a 3x3 layer over a nearest x2 up-sampled input, collapsed
// onto the low-resolution grid (conv_mfma.hip, k_collapse_up_weights): taps (a, b) in {0,1}^2 read the halo at
// (py + a, px + b), weights [co][4][Cin] of that parity, and output pixel (y, x) of the tile is written to (2y + py, 2x + px)
// of the [2H, 2W] output.  The same halo, slots and pipeline; an item has 4/9 of the MFMAs.
template <int NW, int TH, int BN, int KC, bool WPERSIST, bool TWO_SRC, int NTAP = 9>
__global__ void __launch_bounds__(64 * NW, 1) k_conv_halo(HaloArgs a) {
    constexpr int NT = 64 * NW;
    constexpr bool UP2 = NTAP == 4;
    static_assert(NTAP == 9 || (NTAP == 4 && BN != 16 && !TWO_SRC), "tap table: 3x3, or one parity of the collapsed up-sampled form");
    // BN == 16: the 16-cout layers run on v_mfma_f32_16x16x4_f32 (a 32-wide tile would be half padding); a row of 32
    // pixels is two 16-pixel M blocks, rows are padded to 24 floats (conflict-free for its (pixel, k-quarter) lanes)
    constexpr bool M16 = BN == 16;
    constexpr int KP = M16 ? 24 : KC + 4;
    constexpr int C4 = KC / 4;
    constexpr int WC = (BN / 32 > 1 && NW > TH) ? 2 : 1;     // waves across N
    constexpr int WR = NW / WC;                               // waves across tile rows
    constexpr int TM = TH / WR, TN = M16 ? 1 : BN / 32 / WC;
    static_assert(WR * TM == TH && (M16 || WC * TN * 32 == BN), "wave grid must tile the workgroup tile");
    static_assert(!M16 || KC == 16, "the 16-wide path takes 16-channel chunks");
    constexpr int HPIX = (TH + 2) * HALO_W;
    constexpr int HF = HPIX * C4;                 // float4 per halo chunk
    constexpr int LH = (HF + NT - 1) / NT;
    constexpr int WF = NTAP * BN * C4;            // float4 per weight chunk
    constexpr int LW = (WF + NT - 1) / NT;
    constexpr int HBUF = HPIX * KP;               // floats
    constexpr int WBUF = NTAP * BN * KP;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Hs = smem;                              // [2][HBUF]
    float* Ws = smem + 2 * HBUF;                   // [WPERSIST ? 1 : 2][WBUF]
    // InstanceNorm statistics of the output, fused: every wave leaves the column sums of its tile rows here, after the
    // item's barrier BN threads merge the TH rows in a fixed order and write one (sum, M2 about the tile mean) pair per cout and
    // tile; the norm's finalize sums the tiles of a plane in double.  Two buffers: a fold reads while the next tile's
    // rows may already be written (the fold of tile k and the row sums of tile k+1 are one barrier apart).
    float* Rs = smem + 2 * HBUF + (WPERSIST ? 1 : 2) * WBUF;      // [2][TH][BN][2]

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

    // loader slots (fixed for the whole kernel).  The loader is branch-free on purpose: with loads spread over several
    // basic blocks the compiler's waitcnt insertion falls back to vmcnt(0) in the middle of the MFMA loop.  Slots past
    // the end of a tile are given the out-of-range offset (they load 0) and park their LDS write in the padding
    // floats of row 0, which nothing reads.
    constexpr int DUMMY = KC;          // floats KC..KC+3 of row 0: padding
    int h_lds[LH];
    short h_y[LH], h_x[LH];
    bool h_ok[LH];
    const unsigned h_c = (unsigned)(tid % C4) * 4u;      // NT % C4 == 0: every slot of a thread has the same channel group
    static_assert(NT % C4 == 0, "channel group must be slot-invariant");
#pragma unroll
    for (int j = 0; j < LH; ++j) {
        int f = tid + j * NT;
        bool ok = (HF % NT == 0) || f < HF;
        int hp = ok ? f / C4 : 0;
        h_ok[j] = ok;
        h_lds[j] = ok ? hp * KP + (int)h_c : DUMMY;
        h_y[j] = (short)(hp / HALO_W);
        h_x[j] = (short)(hp % HALO_W);
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
        w_lds[j] = ok ? row * KP + c4 * 4 : DUMMY;
        w_off[j] = (ok && co < Cout) ? (((unsigned)co * NTAP + tap) * Cin + c4 * 4) * 4u : a.nbw;
    }

    // ---- prefetch of the next (tile, chunk) item: global -> registers, in pieces -----------------------------------
    // The address arithmetic, buffer loads and the previous tile's stores of an item are cut into SLOTS of a few vector
    // instructions each, and one slot is placed behind each MFMA of the item's second fragment group onwards
    // (sched_barrier keeps them there).  Left to itself the scheduler emits them as one run of ~200 instructions
    // between two MFMAs; both waves of a SIMD reach that run together (they leave the barrier together), so the matrix
    // pipe idled for the length of it once per item.  Everything is select-based ('&' instead of '&&', both arms of
    // a ?: evaluated): no exec-masked blocks, the item body is one basic block.
    float4 rh[LH], rw[WPERSIST ? 1 : LW];
    // wave-uniform state of the prefetch (set by issue_setup, used by the slots)
    __amdgpu_buffer_rsrc_t i_rs = rs0;
    unsigned i_Csrc = 0, i_nbs = 0, i_cb = 0, i_img = 0, i_cc4 = 0;
    int i_sh = 0, i_Wsrc = 0, i_y0 = 0, i_x0 = 0;
    auto issue_setup = [&](int n, int tx, int ty, int ch) {
        const int cc = ch * KC;
        // a chunk never straddles the two sources (single-source layers are a separate instantiation)
        const bool from0 = !TWO_SRC || cc < C0;
        const bool up = from0 && up0;
        i_rs = from0 ? rs0 : rs1;
        i_Csrc = from0 ? (unsigned)C0 : (unsigned)C1;
        i_nbs = from0 ? a.nb0 : a.nb1;
        i_cb = (unsigned)(from0 ? cc : cc - C0) + h_c;
        i_img = up ? (unsigned)n * Hs2 * Ws2 : (unsigned)n * H * W;
        i_sh = up ? 1 : 0;
        i_Wsrc = up ? Ws2 : W;
        i_y0 = ty * TH - 1;
        i_x0 = tx * 32 - 1;
        i_cc4 = (unsigned)cc * 4u;
    };
    auto issue_h = [&](int j) {
        const int yy = i_y0 + h_y[j], xx = i_x0 + h_x[j];
        const bool ok = h_ok[j] & ((unsigned)yy < (unsigned)H) & ((unsigned)xx < (unsigned)W);     // '&': no short-circuit blocks
        const unsigned pix = i_img + (unsigned)((yy >> i_sh) * i_Wsrc + (xx >> i_sh));
        rh[j] = buf_ld4(i_rs, sel_u32(ok, (pix * i_Csrc + i_cb) * 4u, i_nbs));
    };
    auto issue_w = [&](int j) {
        if constexpr (!WPERSIST) rw[j] = buf_ld4(rsw, sel_u32(w_off[j] == a.nbw, a.nbw, w_off[j] + i_cc4));
    };
    // registers -> LDS buffer `buf`, one float4 per piece
    auto commit_h = [&](int j, int buf) { *(float4*)&Hs[buf * HBUF + h_lds[j]] = rh[j]; };
    auto commit_w = [&](int j, int buf) {
        if constexpr (!WPERSIST) *(float4*)&Ws[buf * WBUF + w_lds[j]] = rw[j];
    };

    if (nitems <= 0) return;           // uniform per workgroup
    if constexpr (WPERSIST) {
#pragma unroll
        for (int j = 0; j < LW; ++j) {
            float4 v = buf_ld4(rsw, w_off[j]);
            *(float4*)&Ws[w_lds[j]] = v;
        }
    }
    // tile coordinates are carried incrementally (tiles of a workgroup are consecutive: down a column strip, then the
    // next strip, then the next image): no integer divisions in the item loop
    int cn, ctx, cty, ch = 0;
    {
        cn = sp0 / per_img;
        const int rem = sp0 - cn * per_img;
        ctx = rem / a.tilesY;
        cty = rem - ctx * a.tilesY;
    }
    issue_setup(cn, ctx, cty, 0);
#pragma unroll
    for (int j = 0; j < LH; ++j) issue_h(j);
#pragma unroll
    for (int j = 0; j < LW; ++j) issue_w(j);
#pragma unroll
    for (int j = 0; j < LH; ++j) commit_h(j, 0);
#pragma unroll
    for (int j = 0; j < LW; ++j) commit_w(j, 0);
    __syncthreads();

    const int wr = wv / WC, wc = wv % WC;
    const int lrow = lane & 31, lk = (lane >> 5) * 4;
    // fragment bases (floats): A = pixel (row wr*TM + i, column lrow) of the halo at tap (0,0); B = cout wc*TN*32 + j*32 + lrow
    const int a_base = M16 ? ((wr * TM) * HALO_W + (lane & 15)) * KP + (lane >> 4) * 4 : ((wr * TM) * HALO_W + lrow) * KP + lk;
    const int b_base = M16 ? (lane & 15) * KP + (lane >> 4) * 4 : (wc * TN * 32 + lrow) * KP + lk;

    // Finished tiles are written one item late, through raw buffer stores (an invalid row / cout gets the out-of-range
    // offset and is dropped by the hardware: no branches) that are older than the item's prefetch loads.  Reason: gfx9
    // counts loads and stores in one vmcnt and the compiler treats the mix as unordered, so stores issued between a
    // prefetch and its use turn every later wait into vmcnt(0); with the stores older than the loads, the wait in front
    // of the LDS commit finds both long finished.
    const __amdgpu_buffer_rsrc_t rsy = make_rsrc(a.y, a.nby);
    float bvv[TN];
    unsigned co_off[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int co = M16 ? co_base + (lane & 15) : co_base + (wc * TN + j) * 32 + (lane & 31);
        bvv[j] = (a.bias && co < Cout) ? a.bias[co] : 0.f;
        co_off[j] = co < Cout ? (unsigned)co : 0xFFFFFFFFu;
    }
    const int col0 = M16 ? 4 * (lane >> 4) : 4 * (lane >> 5);
    f32x16 acc[TM][TN], done[TM][TN];      // done: the finished tile with bias (+ReLU) applied, waiting to be stored
    f32x4 acc4[TM][2], done4[TM][2];       // 16-wide path: two 16-pixel blocks per row, C/D = 4 pixel rows x 16 couts per lane
    const float lo = a.relu ? 0.f : -__builtin_inff();      // ReLU epilogue = max against 0, no ReLU = max against -inf
    bool pending = false;                  // a finished tile waits in `done`
    int dn = 0, dtx = 0, dty = 0;          // its coordinates
    // The stores take their data straight from `done` and one offset register per piece; the pixel-column part of the
    // address is a scalar offset.  A pending store pins its source registers until vmcnt says it is done, so stores fed
    // from temporaries would make the code after them wait a full memory round trip.
    // C/D layout (32x32): col = lane&31 (cout), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel column)
    constexpr int NFP = M16 ? TM : 2 * TM * TN;          // store pieces of 8 stores
    auto flush_piece = [&](int s) {
        const int i = M16 ? s : (s >> 1) % TM, j = M16 ? 0 : (s >> 1) / TM, half = s & 1;
        const int yy = dty * TH + wr * TM + i;
        const bool ok = pending & (co_off[j] != 0xFFFFFFFFu) & (yy < H);
        const unsigned base = UP2 ? (((unsigned)dn * 2 * H + (unsigned)(2 * yy + a.py)) * 2 * W + (unsigned)(2 * (dtx * 32 + col0) + a.px)) * (unsigned)Cout + co_off[j]
                                  : (((unsigned)dn * H + (unsigned)yy) * W + (unsigned)(dtx * 32 + col0)) * (unsigned)Cout + co_off[j];
        const int voff = (int)sel_u32(ok, base * 4u, a.nby);       // nothing finished / out of range: dropped by the hardware
        if constexpr (M16) {          // 16x16 C/D layout: col = lane&15 (cout), pixel = blk*16 + 4*(lane>>4) + r
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(done4[i][blk][r]), rsy, voff, (blk * 16 + r) * Cout * 4, 0);     // (never with UP2)
        } else {
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int r = half * 8 + r8;
                const int col = (r & 3) + 8 * (r >> 2);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(done[i][j][r]), rsy, voff, (UP2 ? 2 : 1) * col * Cout * 4, 0);
            }
        }
    };
    constexpr int NLW = WPERSIST ? 0 : LW;
    constexpr int NSLOT = NFP + LH + NLW;                 // stores, then loads: MFMA positions [0, NSLOT) behind group 0
    // The LDS commit of the prefetched item (buffer cur^1: free since the barrier that ended the previous item) is also
    // spread behind MFMAs, in the second half of the item, when the loads have long landed.  At the end of an item only
    // the barrier is left; before, all eight waves wrote their 3-8 float4 there together (LDS write bandwidth:
    // 22-58 KB per item at 128 B/clk) with the matrix pipe empty.
    constexpr int NMF = M16 ? 8 * 8 * TM : (NTAP * (KC / 8) - 1) * 4 * TM * TN;     // MFMA positions behind group 0
    constexpr int NCOM = LH + NLW;
    constexpr int CS = (NMF / 2 > NSLOT ? NMF / 2 : NSLOT);
    static_assert(CS + NCOM <= NMF, "more slots than MFMAs to hide them behind");
    int cur = 0;
    auto slot = [&](int s) {           // s is a compile-time constant after unrolling
        if (s < NFP) flush_piece(s);
        else if (s < NFP + LH) issue_h(s - NFP);
        else if (s < NSLOT) issue_w(s - NFP - LH);
        else if (s >= CS && s < CS + LH) commit_h(s - CS, cur ^ 1);
        else if (s >= CS + LH && s < CS + NCOM) commit_w(s - CS - LH, cur ^ 1);
    };
    auto slotted = [&](int s0, int n) { return s0 >= 0 && (s0 < NSLOT || (s0 + n > CS && s0 < CS + NCOM)); };
    int fold_t = -1, spar = 0;         // tile whose row sums wait in Rs[spar] (-1: none)
    auto fold_stats = [&]() {          // after the barrier that follows a finished tile
        if (fold_t < 0) return;        // uniform
        if (tid < BN && co_base + tid < Cout) {
            const float* R = Rs + spar * (TH * BN * 2) + tid * 2;
            float s1 = R[0], s2 = R[1];          // rows of 32 pixels, merged in row order
#pragma unroll
            for (int r = 1; r < TH; ++r) stat_merge(s1, s2, (float)(32 * r), R[r * BN * 2], R[r * BN * 2 + 1], 32.f);
            float* o = a.stats + ((size_t)fold_t * Cout + co_base + tid) * 2;
            o[0] = s1;
            o[1] = s2;
        }
        fold_t = -1;
        spar ^= 1;
    };
    for (int item = 0; item < nitems; ++item) {
        if (ch == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc4[i][blk][r] = 0.f;
            }
        }
        // next item = next chunk of this tile, or chunk 0 of the next tile.  Past the last item the coordinates run off
        // the workgroup's share: those loads read another tile or nothing (out-of-range offsets return 0) and their
        // LDS copy is never consumed.
        const int nxch = ch + 1 == nch ? 0 : ch + 1;
        const int adv = ch + 1 == nch ? 1 : 0;
        const int ty1 = cty + adv, wy = ty1 == a.tilesY ? 1 : 0;
        const int nty = wy ? 0 : ty1;
        const int tx1 = ctx + wy, wx = tx1 == a.tilesX ? 1 : 0;
        const int ntx = wx ? 0 : tx1;
        const int nn = cn + wx;
        issue_setup(nn, ntx, nty, nxch);
        const float* Hc = Hs + cur * HBUF + a_base;
        const float* Wc = Ws + (WPERSIST ? 0 : cur * WBUF) + b_base;
        // fragment reads run one group (4 k-steps) ahead of the MFMAs that consume them, so a wave's LDS latency hides
        // behind its own matrix work instead of relying on the partner wave
        if constexpr (M16) {
            // one group per tap: lane (pixel m = lane&15, k-quarter q = lane>>4) reads channels 4q..4q+3 of its pixel
            // (block blk adds 16 pixels) and of cout n = lane&15; MFMA j of the group contracts channels {j, 4+j, 8+j, 12+j}
            float4 av[2][TM][2], bv[2];
            auto ldfrag = [&](int tap, int s) {
                const int ky = tap / 3, kx = tap % 3;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int blk = 0; blk < 2; ++blk)
                        av[s][i][blk] = *(const float4*)&Hc[((i + ky) * HALO_W + kx + blk * 16) * KP];
                bv[s] = *(const float4*)&Wc[(tap * 16) * KP];
            };
            constexpr int GM = 8 * TM;         // MFMAs per group
            ldfrag(0, 0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int s = tap & 1;
                if (tap + 1 < 9) ldfrag(tap + 1, s ^ 1);
                const bool sl = slotted((tap - 1) * GM, GM);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int blk = 0; blk < 2; ++blk) {
                        const int m0 = (tap - 1) * GM + (i * 2 + blk) * 4;
                        acc4[i][blk] = MFMA16(av[s][i][blk].x, bv[s].x, acc4[i][blk]);
                        if (sl) { slot(m0); __builtin_amdgcn_sched_barrier(0); }
                        acc4[i][blk] = MFMA16(av[s][i][blk].y, bv[s].y, acc4[i][blk]);
                        if (sl) { slot(m0 + 1); __builtin_amdgcn_sched_barrier(0); }
                        acc4[i][blk] = MFMA16(av[s][i][blk].z, bv[s].z, acc4[i][blk]);
                        if (sl) { slot(m0 + 2); __builtin_amdgcn_sched_barrier(0); }
                        acc4[i][blk] = MFMA16(av[s][i][blk].w, bv[s].w, acc4[i][blk]);
                        if (sl) { slot(m0 + 3); __builtin_amdgcn_sched_barrier(0); }
                    }
                if (!sl) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 2 * TM + 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 8 * TM, 0);
                }
            }
        } else {
            constexpr int KG = KC / 8, NG = NTAP * KG;
            constexpr int GM = 4 * TM * TN;    // MFMAs per group
            float4 av[2][TM], bv[2][TN];
            // halo offset of a tap: compile-time for the 3x3 table, from the launch's parity for the collapsed form
            const int tap_py = UP2 ? a.py : 0, tap_px = UP2 ? a.px : 0;
            auto ldfrag = [&](int g, int s) {
                const int tap = g / KG, kg = g % KG;
                const int ky = UP2 ? tap_py + tap / 2 : tap / 3, kx = UP2 ? tap_px + tap % 2 : tap % 3;
    #pragma unroll
                for (int i = 0; i < TM; ++i) av[s][i] = *(const float4*)&Hc[((i + ky) * HALO_W + kx) * KP + kg * 8];
    #pragma unroll
                for (int j = 0; j < TN; ++j) bv[s][j] = *(const float4*)&Wc[(tap * BN + j * 32) * KP + kg * 8];
            };
            ldfrag(0, 0);
    #pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int s = g & 1;
                if (g + 1 < NG) ldfrag(g + 1, s ^ 1);
                const bool sl = slotted((g - 1) * GM, GM);
    #pragma unroll
                for (int i = 0; i < TM; ++i)
    #pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int m0 = (g - 1) * GM + (i * TN + j) * 4;
                        acc[i][j] = MFMA32(av[s][i].x, bv[s][j].x, acc[i][j]);
                        if (sl) { slot(m0); __builtin_amdgcn_sched_barrier(0); }
                        acc[i][j] = MFMA32(av[s][i].y, bv[s][j].y, acc[i][j]);
                        if (sl) { slot(m0 + 1); __builtin_amdgcn_sched_barrier(0); }
                        acc[i][j] = MFMA32(av[s][i].z, bv[s][j].z, acc[i][j]);
                        if (sl) { slot(m0 + 2); __builtin_amdgcn_sched_barrier(0); }
                        acc[i][j] = MFMA32(av[s][i].w, bv[s][j].w, acc[i][j]);
                        if (sl) { slot(m0 + 3); __builtin_amdgcn_sched_barrier(0); }
                    }
                if (!sl) {
                    __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);          // next group's LDS reads first ...
                    __builtin_amdgcn_sched_group_barrier(0x008, GM, 0);               // ... then this group's MFMAs
                }
            }
        }
        pending = false;               // the slots above stored it
        if (ch == nch - 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) done[i][j][r] = fmaxf(acc[i][j][r] + bvv[j], lo);
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int r = 0; r < 4; ++r) done4[i][blk][r] = fmaxf(acc4[i][blk][r] + bvv[0], lo);
            }
            pending = true;
            dn = cn; dtx = ctx; dty = cty;
            if (a.stats) {             // uniform
                float* R = Rs + spar * (TH * BN * 2);
#pragma unroll
                for (int i = 0; i < TM; ++i) {       // H % TH == 0 whenever statistics are requested: every row counts
                    if constexpr (M16) {      // lane: cout = lane&15, pixels blk*16 + 4*(lane>>4) + r
                        float vals[8], s1, s2;
#pragma unroll
                        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                            for (int r = 0; r < 4; ++r) vals[blk * 4 + r] = done4[i][blk][r];
                        lane_stats<8>(vals, s1, s2);
                        stat_merge_eq(s1, s2, __shfl_xor(s1, 16, 64), __shfl_xor(s2, 16, 64), 1.f / 16.f);
                        stat_merge_eq(s1, s2, __shfl_xor(s1, 32, 64), __shfl_xor(s2, 32, 64), 1.f / 32.f);
                        if (lane < 16) { R[((wr * TM + i) * BN + lane) * 2] = s1; R[((wr * TM + i) * BN + lane) * 2 + 1] = s2; }
                    } else {
#pragma unroll
                        for (int j = 0; j < TN; ++j) {    // lane: cout = lane&31, 16 of the row's 32 pixels; the other half in lane^32
                            float vals[16], s1, s2;
#pragma unroll
                            for (int r = 0; r < 16; ++r) vals[r] = done[i][j][r];
                            lane_stats<16>(vals, s1, s2);
                            stat_merge_eq(s1, s2, __shfl_xor(s1, 32, 64), __shfl_xor(s2, 32, 64), 1.f / 32.f);
                            const int col = (wc * TN + j) * 32 + (lane & 31);
                            if (lane < 32) { R[((wr * TM + i) * BN + col) * 2] = s1; R[((wr * TM + i) * BN + col) * 2 + 1] = s2; }
                        }
                    }
                }
                fold_t = UP2 ? ((dn * 4 + a.py * 2 + a.px) * a.tilesX + dtx) * a.tilesY + dty : (dn * a.tilesX + dtx) * a.tilesY + dty;
            }
        }
        ch = nxch; cn = nn; ctx = ntx; cty = nty;
        __syncthreads();               // the item's slots committed buffer cur^1
        cur ^= 1;
        fold_stats();
    }
#pragma unroll
    for (int s = 0; s < NFP; ++s) flush_piece(s);
}

template <int NW, int TH, int BN, int KC, bool WPERSIST, bool TWO_SRC = false, int NTAP = 9>
int launch_halo(const ConvIn& in, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int relu,
                hipStream_t st, float* stats, int py = 0, int px = 0) {
    constexpr int KP = BN == 16 ? 24 : KC + 4;
    constexpr size_t lds = (size_t)(2 * (TH + 2) * HALO_W * KP + (WPERSIST ? 1 : 2) * NTAP * BN * KP + 2 * TH * BN * 2) * sizeof(float);
    static_assert(lds <= 160 * 1024, "halo tile does not fit the 160 KB LDS");
    static bool attr_set = false;
    auto kern = k_conv_halo<NW, TH, BN, KC, WPERSIST, TWO_SRC, NTAP>;
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
    a.stats = stats;
    a.py = py; a.px = px;
    const long P = (long)N * H * W;
    a.nb0 = (unsigned)((in.up0 ? P / 4 : P) * in.C0 * 4);
    a.nb1 = (unsigned)(P * in.C1 * 4);
    a.nbw = (unsigned)((long)Cout * NTAP * (in.C0 + in.C1) * 4);
    a.nby = (unsigned)((NTAP == 4 ? 4 : 1) * P * Cout * 4);
    // The LDS footprint allows one workgroup per CU: one workgroup per CU, each with an even share of the tiles, so the
    // prologue (weights + first halo, not overlapped with MFMA work) is paid once.  Shorter runs per workgroup
    // (VQW_HALO_KT) would let the dispatcher rebalance when other kernels hold CUs; measured 0.5-1 % slower in the step.
    int groups = g_max_blocks / a.ntn;  // spatial groups: groups * ntn workgroups <= one per CU, never a second partial round
    if (groups < 1) groups = 1;
    const int even = ceil_div(a.nsp, groups);
    int kt = g_halo_kt > 0 ? g_halo_kt : even;
    if (kt > even) kt = even;
    a.kt = kt < 1 ? 1 : kt;
    k_conv_halo<NW, TH, BN, KC, WPERSIST, TWO_SRC, NTAP><<<ceil_div(a.nsp, a.kt) * a.ntn, 64 * NW, lds, st>>>(a);
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

// rows per tile of the variant conv_halo_fwd picks (the statistics partials are per tile)
static int halo_tile_rows(const ConvIn& in, int Cout) {
    const int Cin = in.C0 + in.C1;
    if (Cout <= 16 && g_halo16) return 16;
    if (Cin == 32 && in.C1 == 0 && Cout > 32) return 4;
    return 8;
}
int conv_halo_stat_tiles(const ConvIn& in, int H, int W, int Cout) {     // equal, full tiles only: the partials carry no count
    const int th = halo_tile_rows(in, Cout);
    return H % th == 0 ? (H / th) * (W / 32) : 0;
}

int conv_halo_fwd(const ConvIn& in, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int relu,
                  hipStream_t st, float* stats) {
    const int Cin = in.C0 + in.C1;
    if (Cout <= 16 && g_halo16) {       // 16-cout layers: 16x16x4 MFMA, 16 x 32 pixel tiles
        if (Cin == 16 && in.C1 == 0) return launch_halo<8, 16, 16, 16, true>(in, w, bias, y, N, H, W, Cout, relu, st, stats);
        if (in.C1 > 0) return launch_halo<8, 16, 16, 16, false, true>(in, w, bias, y, N, H, W, Cout, relu, st, stats);
        return launch_halo<8, 16, 16, 16, false>(in, w, bias, y, N, H, W, Cout, relu, st, stats);
    }
    const bool wide = Cout > 32;        // 64-wide cout tile
    if (Cin == 32 && in.C1 == 0) {      // single chunk: weights stay in LDS
        if (wide) return launch_halo<8, 4, 64, 32, true>(in, w, bias, y, N, H, W, Cout, relu, st, stats);
        return launch_halo<8, 8, 32, 32, true>(in, w, bias, y, N, H, W, Cout, relu, st, stats);
    }
    if (Cin == 16 && in.C1 == 0) {
        if (wide) return launch_halo<8, 8, 64, 16, true>(in, w, bias, y, N, H, W, Cout, relu, st, stats);
        return launch_halo<8, 8, 32, 16, true>(in, w, bias, y, N, H, W, Cout, relu, st, stats);
    }
    if (in.C1 > 0) {
        if (wide) return launch_halo<8, 8, 64, 16, false, true>(in, w, bias, y, N, H, W, Cout, relu, st, stats);
        return launch_halo<8, 8, 32, 16, false, true>(in, w, bias, y, N, H, W, Cout, relu, st, stats);
    }
    if (wide) return launch_halo<8, 8, 64, 16, false>(in, w, bias, y, N, H, W, Cout, relu, st, stats);
    return launch_halo<8, 8, 32, 16, false>(in, w, bias, y, N, H, W, Cout, relu, st, stats);
}

// Collapsed 3x3 over a nearest x2 up-sampled input, forward: four parity launches of the 4-tap form on the low-resolution
// grid [N, h, w, Cin] -> [N, 2h, 2w, Cout]; wc = [4 parities][Cout][4][Cin] (k_collapse_up_weights).
static const int g_halo_up2 = env_int("VQW_HALO_UP2", 1);
bool conv_halo_up2_ok(int Cin, int Cout, int N, int h, int w) {
    if (g_halo_mode != 0 || !g_halo_up2 || w % 32 != 0 || h < 2 || Cin % 16 != 0 || Cin < 32 || Cout < 32) return false;
    return 4L * N * h * w * (Cin > Cout ? Cin : Cout) * 4 <= 0xFFFFFFE0L;
}
int conv_halo_up2_stat_tiles(int h, int w) { return h % 8 == 0 ? 4 * (h / 8) * (w / 32) : 0; }
int conv_halo_up2_fwd(const float* x_low, const float* wc, const float* bias, float* y, int N, int h, int w, int Cin, int Cout,
                      int relu, hipStream_t st, float* stats) {
    ConvIn in{x_low, nullptr, Cin, 0, 0};
    for (int par = 0; par < 4; ++par) {
        const float* wp = wc + (size_t)par * Cout * 4 * Cin;
        const int rc = Cout > 32 ? launch_halo<8, 8, 64, 16, false, false, 4>(in, wp, bias, y, N, h, w, Cout, relu, st, stats, par >> 1, par & 1)
                                 : launch_halo<8, 8, 32, 16, false, false, 4>(in, wp, bias, y, N, h, w, Cout, relu, st, stats, par >> 1, par & 1);
        if (rc) return rc;
    }
    return VQW_OK;
}

// =============================================================================================
// wgrad of a 3x3 stride-1 conv from block-shared tiles (k_conv_wgrad_tile)
// =============================================================================================
// dW[co][ky][kx][ci] = sum_p dY[p][co] * X[p + (ky-1, kx-1)][ci].  GEMM view per (32 co x 32 ci) tile: K = pixels.
// The all-taps-per-wave kernel (k_conv_wgrad9, conv_mfma.hip) stages a private 3-row slab per wave: every x row is
// fetched three times, the staging registers leave no room to read LDS fragments ahead of the MFMAs (256 VGPRs, 12
// spilled) and its MFMA pipes stay ~65 % busy.  Here the 8 waves of a workgroup share ONE 8 x 32 pixel dY tile and its
// 10 x 34 X halo in LDS (1.33x halo overhead instead of 3x, 10 instead of 17 staging float4 per lane); wave r takes
// tile row r as its K slice and all nine taps (9 accumulators), with the fragment reads running one k-step ahead.
// The next tile is fetched into registers during the MFMAs and written to the other LDS buffer afterwards; one barrier
// per tile.  The 8 partial sums are folded through LDS and each workgroup writes one slab [Cout][9][Cin] (+ fused bias
// partial), summed over workgroups in a fixed order by reduce_rows: deterministic, no float atomics.
namespace {

struct WgTileArgs {
    ConvIn in;
    const float* dy;
    float* part;
    float* bias_part;
    int N, H, W, Cout;
    int tilesY, tilesX, nsp;
    int n_ci_t, ntiles, kt;
    unsigned nb0, nb1, nbd;
};

constexpr int WT_TH = 8;
constexpr int WT_D = WT_TH * 32 * 32;                 // floats per dY tile
constexpr int WT_X = (WT_TH + 2) * HALO_W * 32;       // floats per X halo

// MB x NB = 2 x 2: the 32 x 32 (co x ci) tile on v_mfma_f32_32x32x2_f32.  Smaller values: 16-wide sides on
// v_mfma_f32_16x16x4_f32 for layers with <= 16 couts and / or <= 16 input channels (a padded 32-wide side would waste
// half or three quarters of the matrix work).
template <int MB, int NB>
__global__ void __launch_bounds__(512, 1) k_conv_wgrad_tile(WgTileArgs a) {
    constexpr int NT = 512;
    constexpr int LD = WT_TH * 32 * 8 / NT;                       // 4 dY float4 per thread
    constexpr int XF = (WT_TH + 2) * HALO_W * 8;                  // 2720 X float4 per tile
    constexpr int LX = (XF + NT - 1) / NT;                        // 6
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ds = smem;                 // [2][WT_D]
    float* Xs = smem + 2 * WT_D;      // [2][WT_X]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int H = a.H, W = a.W, Cout = a.Cout;
    const int C0 = a.in.C0, C1 = a.in.C1, Cin = C0 + C1;
    const int Hs2 = H >> 1, Ws2 = W >> 1;
    const int tile = blockIdx.x % a.ntiles, sblk = blockIdx.x / a.ntiles;
    const int co_base = (tile / a.n_ci_t) * 32, ci_base = (tile % a.n_ci_t) * 32;
    const int sp0 = sblk * a.kt;
    const int my_tiles = min(a.kt, a.nsp - sp0);
    const int per_img = a.tilesY * a.tilesX;

    const int c4 = tid & 7;
    const int d_c = co_base + c4 * 4;
    const bool d_ok = d_c < Cout;
    const int x_c = ci_base + c4 * 4;
    const bool x_cok = x_c < Cin;
    const bool x_from0 = ci_base < C0;              // a 32-wide ci tile never straddles the sources
    const unsigned xC = x_from0 ? (unsigned)C0 : (unsigned)C1;
    const unsigned x_cb = (unsigned)(x_from0 ? x_c : x_c - C0) * 4u;
    const unsigned nbx = x_from0 ? a.nb0 : a.nb1;
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(x_from0 ? a.in.src0 : a.in.src1, nbx), rsd = make_rsrc(a.dy, a.nbd);
    const bool x_up = x_from0 && a.in.up0;
    const bool do_bias = a.bias_part != nullptr && ci_base == 0;
    const int pslot = tid >> 3;                     // pixel slot of load slot 0 (slot j adds 64 pixels)

    float4 rd[LD], rx[LX];
    float4 bsum;
    bsum.x = bsum.y = bsum.z = bsum.w = 0.f;
    // Prefetch of the next tile (global -> registers) and its LDS commit, in pieces that are placed behind single MFMAs
    // of the tile loop (see k_conv_halo): select-based addresses, no exec-masked blocks, one basic block per tile.
    const int x_sh = x_up ? 1 : 0, x_Hs = x_up ? Hs2 : H, x_Ws = x_up ? Ws2 : W;
    int i_n = 0, i_y0 = 0, i_x0 = 0;
    auto issue_setup = [&](int n, int tx, int ty) { i_n = n; i_y0 = ty * WT_TH; i_x0 = tx * 32; };
    auto issue_d = [&](int j) {
        const int pix = pslot + 64 * j;             // 0..255
        const int yy = i_y0 + (pix >> 5), xx = i_x0 + (pix & 31);
        const bool ok = d_ok & (yy < H);
        const unsigned p = ((unsigned)i_n * H + (unsigned)yy) * W + (unsigned)xx;
        rd[j] = buf_ld4(rsd, sel_u32(ok, (p * (unsigned)Cout + d_c) * 4u, a.nbd));
    };
    auto issue_x = [&](int j) {
        const int hp = pslot + 64 * j;              // 0..339 valid
        const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        const int yy = i_y0 - 1 + hy, xx = i_x0 - 1 + hx;
        const bool ok = (hp < (WT_TH + 2) * HALO_W) & x_cok & ((unsigned)yy < (unsigned)H) & ((unsigned)xx < (unsigned)W);
        const unsigned pix = ((unsigned)i_n * x_Hs + (unsigned)(yy >> x_sh)) * x_Ws + (unsigned)(xx >> x_sh);
        rx[j] = buf_ld4(rsx, sel_u32(ok, pix * xC * 4u + x_cb, nbx));
    };
    // `once`: 1 when the committed tile is one of this workgroup's (the prefetch behind the last tile runs off its share)
    auto commit_d = [&](int j, int buf, float once) {
        *(float4*)&Ds[buf * WT_D + (tid + NT * j) * 4] = rd[j];
        bsum.x += once * rd[j].x; bsum.y += once * rd[j].y; bsum.z += once * rd[j].z; bsum.w += once * rd[j].w;   // fused bias gradient
    };
    float* const xdummy = smem + 2 * (WT_D + WT_X) + tid * 4;      // slots past the halo park their write here
    auto commit_x = [&](int j, int buf) {
        const int f = tid + NT * j;
        float* dst = (XF % NT == 0 || f < XF) ? &Xs[buf * WT_X + f * 4] : xdummy;
        *(float4*)dst = rx[j];
    };

    constexpr bool FULL = MB == 2 && NB == 2;
    f32x16 acc[9];
    f32x4 acc4[9][MB][NB];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc4[t][mb][nb][r] = 0.f;
    }

    const int lcol = lane & 31, lk = lane >> 5;
    // tile coordinates carried incrementally (consecutive tiles: down a column strip, next strip, next image)
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
    int cur = 0;
    constexpr int GM = FULL ? 9 : 9 * MB * NB;       // MFMAs per k-step
    constexpr int NKS = FULL ? 16 : 8;               // k-steps per tile
    constexpr int NMF = (NKS - 1) * GM;              // MFMA positions behind the first k-step
    constexpr int NSLOT = LD + LX;                   // loads: positions [0, NSLOT); commits: [CS, CS + NSLOT) in the second half
    constexpr int CS = NMF / 2 > NSLOT ? NMF / 2 : NSLOT;
    static_assert(CS + NSLOT <= NMF, "more slots than MFMAs to hide them behind");
    float once = 0.f;
    auto slot = [&](int s) {
        if (s < LD) issue_d(s);
        else if (s < NSLOT) issue_x(s - LD);
        else if (s >= CS && s < CS + LD) commit_d(s - CS, cur ^ 1, once);
        else if (s >= CS + LD && s < CS + NSLOT) commit_x(s - CS - LD, cur ^ 1);
    };
    auto slotted = [&](int s0) { return s0 >= 0 && (s0 < NSLOT || (s0 + GM > CS && s0 < CS + NSLOT)); };
    for (int t = 0; t < my_tiles; ++t) {
        {   // next tile (past the last one the loads read another tile or nothing; their LDS copy is never consumed)
            const int ty1 = cty + 1, wy = ty1 == a.tilesY ? 1 : 0;
            cty = wy ? 0 : ty1;
            const int tx1 = ctx + wy, wx = tx1 == a.tilesX ? 1 : 0;
            ctx = wx ? 0 : tx1;
            cn += wx;
            issue_setup(cn, ctx, cty);
            once = t + 1 < my_tiles ? 1.f : 0.f;
        }
        if constexpr (FULL) {
            const float* Dr = Ds + cur * WT_D + (wv * 32 + lk) * 32 + lcol;                 // a(k) = Dr[k * 32]
            const float* Xr = Xs + cur * WT_X + (wv * HALO_W + lk) * 32 + lcol;             // b(ky,kx,k) = Xr[(ky*34 + k + kx) * 32]
            float fa[2], fb[2][9];
            auto ldfrag = [&](int k, int s) {
                fa[s] = Dr[k * 32];
    #pragma unroll
                for (int ky = 0; ky < 3; ++ky)
    #pragma unroll
                    for (int kx = 0; kx < 3; ++kx) fb[s][ky * 3 + kx] = Xr[(ky * HALO_W + k + kx) * 32];
            };
            ldfrag(0, 0);
    #pragma unroll
            for (int k = 0; k < 32; k += 2) {
                const int s = (k >> 1) & 1;
                if (k + 2 < 32) ldfrag(k + 2, s ^ 1);
                const int s0 = ((k >> 1) - 1) * GM;
                const bool sl = slotted(s0);
    #pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    acc[tp] = MFMA32(fa[s], fb[s][tp], acc[tp]);
                    if (sl) { slot(s0 + tp); __builtin_amdgcn_sched_barrier(0); }
                }
                if (!sl) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);
                }
            }
        } else {
            // 16x16x4: lane (channel idx = lane&15, pixel q = lane>>4) of a 4-pixel k-step
            const int idx = lane & 15, q = lane >> 4;
            const float* Dr = Ds + cur * WT_D + (wv * 32 + q) * 32 + idx;                // a(mb, k) = Dr[k * 32 + mb * 16]
            const float* Xr = Xs + cur * WT_X + (wv * HALO_W + q) * 32 + idx;            // b(tap, nb, k) = Xr[(ky*34 + k + kx) * 32 + nb * 16]
            float fa[2][MB], fb[2][9][NB];
            auto ldfrag = [&](int k, int s) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) fa[s][mb] = Dr[k * 32 + mb * 16];
#pragma unroll
                for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) fb[s][tp][nb] = Xr[((tp / 3) * HALO_W + k + tp % 3) * 32 + nb * 16];
            };
            ldfrag(0, 0);
#pragma unroll
            for (int k = 0; k < 32; k += 4) {
                const int s = (k >> 2) & 1;
                if (k + 4 < 32) ldfrag(k + 4, s ^ 1);
                const int s0 = ((k >> 2) - 1) * GM;
                const bool sl = slotted(s0);
#pragma unroll
                for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            acc4[tp][mb][nb] = MFMA16(fa[s][mb], fb[s][tp][nb], acc4[tp][mb][nb]);
                            if (sl) { slot(s0 + (tp * MB + mb) * NB + nb); __builtin_amdgcn_sched_barrier(0); }
                        }
                if (!sl) {
                    __builtin_amdgcn_sched_group_barrier(0x100, MB + 9 * NB, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 9 * MB * NB, 0);
                }
            }
        }
        __syncthreads();               // the tile's slots committed buffer cur^1
        cur ^= 1;
    }

    // fold the 8 waves' partial sums through LDS (staging space is free now) and write ONE slab per workgroup
    float* red = smem;                    // [8 waves][32 co][32 ci]
    float* o = a.part + (size_t)sblk * Cout * 9 * Cin;
    if (do_bias) {                        // threads with equal (tid & 7) hold the same 4 channels
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
    for (int t = 0; t < 9; ++t) {
        if constexpr (FULL) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                red[wv * 1024 + row * 32 + (lane & 31)] = acc[t][r];
            }
        } else {          // 16x16 C/D layout: col = lane&15 (ci), row = 4*(lane>>4) + r (co); sides not computed stay unread
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        red[wv * 1024 + (mb * 16 + 4 * (lane >> 4) + r) * 32 + nb * 16 + (lane & 15)] = acc4[t][mb][nb][r];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int idx = tid + e * 512;          // element of the 32x32 tile
            int row = idx >> 5, col = idx & 31;
            int co = co_base + row, ci = ci_base + col;
            float v = ((red[idx] + red[1024 + idx]) + (red[2048 + idx] + red[3072 + idx])) +
                      ((red[4096 + idx] + red[5120 + idx]) + (red[6144 + idx] + red[7168 + idx]));
            if (co < Cout && ci < Cin) o[((size_t)co * 9 + t) * Cin + ci] = v;
        }
        __syncthreads();
    }
}

}  // namespace

int g_wgrad_tile_mode = 0;    // 0 auto, 1 off (A/B timing)

bool conv_wgrad_tile_ok(int C0, int C1, int Cout, int ks, int W, int dil) {
    return g_wgrad_tile_mode == 0 && ks == 3 && dil == 1 && (W % 32 == 0) && (C0 % 4 == 0) && (C1 % 4 == 0) && (Cout % 4 == 0) &&
           (C1 == 0 || C0 % 32 == 0);
}
// number of slabs for a given upper bound (the workspace is sized by the wg9 split count, which is never smaller)
int conv_wgrad_tile_blocks(int Cin, int Cout, int N, int H, int W, int max_blocks, int* kt_out) {
    const int ntiles = ceil_div(Cout, 32) * ceil_div(Cin, 32);
    const int nsp = N * ceil_div(H, WT_TH) * (W / 32);
    int nsb = g_max_blocks / ntiles;             // one workgroup per CU and never a second, nearly empty round
    if (nsb > max_blocks) nsb = max_blocks;
    if (nsb > nsp) nsb = nsp;
    if (nsb < 1) nsb = 1;
    const int kt = ceil_div(nsp, nsb);
    if (kt_out) *kt_out = kt;
    return ceil_div(nsp, kt);                     // every workgroup gets at least one tile
}
int conv_wgrad_tile(const ConvIn& in, const float* dy, float* ws, float* bpart, int N, int H, int W, int Cout, int nsb, int kt,
                    hipStream_t st) {
    constexpr size_t lds = (size_t)(2 * (WT_D + WT_X) + 512 * 4) * sizeof(float);      // + one parking float4 per thread
    static_assert(lds <= 160 * 1024, "wgrad tiles do not fit the LDS");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_conv_wgrad_tile<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_conv_wgrad_tile<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_conv_wgrad_tile<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_conv_wgrad_tile<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            vqw_set_error("conv_wgrad_tile: cannot raise the dynamic LDS limit");
            return VQW_ERR_HIP;
        }
        attr_set = true;
    }
    const int Cin = in.C0 + in.C1;
    const long P = (long)N * H * W;
    WgTileArgs a;
    a.in = in; a.dy = dy; a.part = ws; a.bias_part = bpart;
    a.N = N; a.H = H; a.W = W; a.Cout = Cout;
    a.tilesY = ceil_div(H, WT_TH); a.tilesX = W / 32; a.nsp = N * a.tilesY * a.tilesX;
    a.n_ci_t = ceil_div(Cin, 32); a.ntiles = ceil_div(Cout, 32) * a.n_ci_t; a.kt = kt;
    a.nb0 = (unsigned)((in.up0 ? P / 4 : P) * in.C0 * 4);
    a.nb1 = (unsigned)(P * in.C1 * 4);
    a.nbd = (unsigned)(P * Cout * 4);
    const bool m16 = g_halo16 && Cout <= 16, n16 = g_halo16 && Cin <= 16;
    if (m16 && n16) k_conv_wgrad_tile<1, 1><<<a.ntiles * nsb, 512, lds, st>>>(a);
    else if (m16) k_conv_wgrad_tile<1, 2><<<a.ntiles * nsb, 512, lds, st>>>(a);
    else if (n16) k_conv_wgrad_tile<2, 1><<<a.ntiles * nsb, 512, lds, st>>>(a);
    else k_conv_wgrad_tile<2, 2><<<a.ntiles * nsb, 512, lds, st>>>(a);
    VQW_LAUNCH_CHECK("conv_wgrad_tile");
    return VQW_OK;
}
