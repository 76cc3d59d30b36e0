// Dilated 3x3 convolution (forward and dgrad) of the atrous pyramid's 32-channel branches: full image rows resident in
// LDS, walked along the dilation's residue chains.  Exact fp32 MFMA, NHWC.
//
// Why a third forward kernel: with dilation d the taps of an output row y sit in rows y-d, y, y+d, so a (TH+2) x 34 halo
// tile (conv_halo.hip) would need (TH+2d) x (32+2d) pixels - at d = 18 more than the LDS holds - and the implicit-GEMM
// kernel (conv_mfma.hip) gathers its A operand once per tap: every input element travels L2 -> LDS nine times and the
// matrix pipes wait for it (60-69 TFLOP/s on the four pyramid branches, profiles/r02_*).  Here a workgroup slides down a
// CHAIN of rows y, y+d, y+2d, ... of one image: three whole rows (y-d, y, y+d) stay in LDS, all nine taps read them at
// shifted addresses, and moving to the next output row of the chain loads ONE new row (y+2d) while the oldest is dropped.
// Every input row is fetched once per chain segment: the same 9x reuse the halo kernel has, for any dilation.
//
//   * sequence: the rows of an image in chain order (residue 0: 0, d, 2d, ...; residue 1: 1, d+1, ...), each chain
//     followed by one all-zero row Z (the row below the last and above the first row of a chain is padding).  The images'
//     sequences are concatenated; a workgroup takes a contiguous run of positions.  A Z position computes nothing.
//   * LDS: 3 row slots of (WMAX + 1) pixels x 36 floats (32 channels + 4 padding: conflict-free ds_read_b128 of 32
//     consecutive pixels); pixel WMAX of each slot is a zero pixel that the lanes whose tap column falls outside the image
//     read instead (the horizontal padding costs no instruction: the three per-lane column offsets are loop invariant).
//     The weights [9 taps][32 couts][36] stay in LDS for the whole kernel.  3 x 37,008 + 41,472 B = 152.5 KB.
//   * a step: wave w computes pixels 32w..32w+31 of the row for all 32 couts (144 MFMA 32x32x2).  The taps of the oldest
//     row come first; a barrier after them frees its slot, the row prefetched into registers at the start of the step
//     is written there behind the later MFMAs, the barrier at the end of the step publishes it.  Stores of the previous
//     row's outputs and the prefetch loads are placed one piece per MFMA (see conv_halo.hip for why).
//   * statistics for the following InstanceNorm: one (sum, M2) pair per output row and cout (tiles = rows).
#include "common.h"
#include "conv_common.h"
#include "mfma_util.h"
#include <cstdlib>

int g_dil_mode = 0;     // 0 auto, 1 off (A/B timing, tests of the implicit-GEMM kernel)

namespace {

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
static const int g_dil_env = env_int("VQW_DIL_ROWS", 1);
static const int g_max_blocks = []{ int v = env_int("VQW_CONV_MAX_BLOCKS", 256); return v < 8 ? 8 : (v > 256 ? 256 : v); }();

struct DilArgs {
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    int N, H, W, Cout, dil;
    int per, total;            // sequence positions per workgroup, N * (H + dil) in all
    int relu;
    unsigned nbx, nbw, nby;
    float* stats;              // optional [N][H][Cout][2]: (sum, M2 about the row mean) of every output row
};

// ACC: the output is ADDED to what y holds (input gradients of several convolutions of one tensor summed in place: the
// atrous pyramid's branches); the old values of a row's outputs are fetched behind the first MFMAs of its step.
template <int NW, bool ACC>
__global__ void __launch_bounds__(64 * NW, 1) k_conv_dilrow(DilArgs a) {
    constexpr int NT = 64 * NW, WMAX = 32 * NW, KP = 36;
    constexpr int SLOT = (WMAX + 1) * KP;             // floats per row slot (pixel WMAX = zero pixel)
    constexpr int WBUF = 9 * 32 * KP;
    constexpr int LH = WMAX * 8 / NT;                 // float4 per thread and row
    constexpr int WF = 9 * 32 * 8, LW = (WF + NT - 1) / NT;
    static_assert(LH * NT == WMAX * 8, "row float4s must divide over the threads");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Hs = smem;                     // [3][SLOT]
    float* Ws = smem + 3 * SLOT;          // [9][32][KP]
    float* Rs = Ws + WBUF;                // [NW][32][2] row statistics of the waves

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int H = a.H, W = a.W, Cout = a.Cout, d = a.dil;
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(a.x, a.nbx), rsw = make_rsrc(a.w, a.nbw), rsy = make_rsrc(a.y, a.nby);

    const int t0 = blockIdx.x * a.per;
    const int nsteps = min(a.per, a.total - t0);
    if (nsteps <= 0) return;              // uniform per workgroup

    if (tid < 3 * KP) Hs[(tid / KP) * SLOT + WMAX * KP + tid % KP] = 0.f;
#pragma unroll
    for (int j = 0; j < LW; ++j) {
        const int f = tid + j * NT;
        const bool ok = (WF % NT == 0) || f < WF;
        const int row = ok ? f / 8 : 0, c4 = f % 8;       // row = tap * 32 + cout
        const int tap = row / 32, n = row % 32;
        const float4 v = buf_ld4(rsw, (ok && n < Cout) ? (((unsigned)n * 9 + tap) * 32 + c4 * 4) * 4u : a.nbw);
        if (ok) *(float4*)&Ws[row * KP + c4 * 4] = v;
    }

    // row loader: thread -> LH float4 of a row, the same for every row (a row is one contiguous W * 128 bytes)
    const unsigned row_bytes = (unsigned)W * 128u;
    unsigned h_off[LH];
    int h_lds[LH];
#pragma unroll
    for (int j = 0; j < LH; ++j) {
        const int f = tid + j * NT;
        h_off[j] = f < W * 8 ? (unsigned)f * 16u : 0xFFFFFFFFu;
        h_lds[j] = (f >> 3) * KP + (f & 7) * 4;
    }
    float4 rh[LH];
    auto issue_h = [&](int j, int g) {       // g: global row n * H + y, or -1 (zero row)
        const bool ok = (g >= 0) & (h_off[j] != 0xFFFFFFFFu);
        rh[j] = buf_ld4(rsx, sel_u32(ok, (unsigned)g * row_bytes + h_off[j], a.nbx));
    };
    auto commit_h = [&](int j, int slot_f) { *(float4*)&Hs[slot_f + h_lds[j]] = rh[j]; };
    // head: a position in the chain-ordered row sequence of the batch = image hn, residue hrho, row hy (-1: the zero row
    // that follows chain hrho).  Wave-uniform scalars.
    int hn, hrho, hy;
    if (t0 == 0) {               // the zero row in front of the first chain
        hn = -1; hrho = d - 1; hy = -1;
    } else {
        const int Ls = H + d;
        hn = (t0 - 1) / Ls;
        int u = (t0 - 1) - hn * Ls;
        hrho = 0;
        for (;;) {
            const int L = (H - hrho + d - 1) / d;      // rows of chain hrho (>= 1: d <= H)
            if (u < L) { hy = hrho + u * d; break; }
            if (u == L) { hy = -1; break; }
            u -= L + 1;
            ++hrho;
        }
    }
    auto head_row = [&]() { return (hy >= 0 && hn >= 0 && hn < a.N) ? hn * H + hy : -1; };
    auto head_next = [&]() {
        if (hy < 0) {
            if (++hrho == d) { hrho = 0; ++hn; }
            hy = hrho;
        } else {
            hy += d;
            if (hy >= H) hy = -1;
        }
    };

    // prologue: rows at positions t0-1, t0, t0+1 into slots 0, 1, 2
    int g0 = -1, g1 = -1;                    // rows at positions p, p+1; head stands at p+2
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int g = head_row();
#pragma unroll
        for (int j = 0; j < LH; ++j) issue_h(j, g);
#pragma unroll
        for (int j = 0; j < LH; ++j) commit_h(j, s * SLOT);
        head_next();
        g0 = g1;
        g1 = g;
    }
    int sA = 0, sB = SLOT, sC = 2 * SLOT;    // slots of positions p-1, p, p+1
    __syncthreads();

    // fragment addressing: lane (pixel 32 wv + (lane & 31), k-half lane >> 5); tap column kx shifts the pixel by (kx-1) d
    const int lrow = lane & 31, lk = (lane >> 5) * 4;
    auto col_of = [&](int kx) {
        const int x = wv * 32 + lrow + (kx - 1) * d;
        return ((unsigned)x < (unsigned)W ? x : WMAX) * KP + lk;
    };
    const int cx0 = col_of(0), cx1 = col_of(1), cx2 = col_of(2);
    const float* Wb = Ws + lrow * KP + lk;

    const int co = lane & 31;
    const float bv0 = (a.bias && co < Cout) ? a.bias[co] : 0.f;
    const bool st_ok = (co < Cout) & (wv * 32 < W);
    const int col0 = 4 * (lane >> 5);
    const float lo = a.relu ? 0.f : -__builtin_inff();
    f32x16 acc, done;
    float yold[ACC ? 16 : 1];
    bool pending = false;
    int dg = 0;                              // global row of the tile waiting in `done`
    // C/D layout (32x32): col = lane & 31 (cout), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (pixel column)
    auto flush_piece = [&](int half) {
        const unsigned base = (((unsigned)dg * W + (unsigned)(wv * 32 + col0)) * (unsigned)Cout + (unsigned)co) * 4u;
        const int voff = (int)sel_u32(pending & st_ok, base, a.nby);
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {
            const int r = half * 8 + r8;
            const int col = (r & 3) + 8 * (r >> 2);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(done[r]), rsy, voff, col * Cout * 4, 0);
        }
    };
    const int nwr = W / 32;                  // waves that hold pixels of the row

    for (int i = 0; i < nsteps; ++i) {
        const int g2 = (i + 2 <= nsteps) ? head_row() : -1;      // position p+2 is a neighbour of p+1 <= last position
        if (g0 >= 0) {           // uniform
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            constexpr int NG = 36;
            float4 av[2], bv[2];
            auto ldfrag = [&](int g, int s) {
                const int tap = g / 4, kg = g % 4, ky = tap / 3, kx = tap % 3;
                av[s] = *(const float4*)&Hs[(ky == 0 ? sA : ky == 1 ? sB : sC) + (kx == 0 ? cx0 : kx == 1 ? cx1 : cx2) + kg * 8];
                bv[s] = *(const float4*)&Wb[tap * 32 * KP + kg * 8];
            };
            auto slot = [&](int m) {         // m: MFMA position behind group 0 (compile-time after unrolling)
                if (m < 2) flush_piece(m);
                else if (m < 2 + LH) issue_h(m - 2, g2);
                else if (ACC && m < 2 + LH + 16) {       // old value of output r of this step's row
                    const int r = m - 2 - LH;
                    const unsigned base = (((unsigned)g0 * W + (unsigned)(wv * 32 + col0)) * (unsigned)Cout + (unsigned)co) * 4u;
                    yold[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsy, (int)sel_u32(st_ok, base, a.nby),
                                                                                 ((r & 3) + 8 * (r >> 2)) * Cout * 4, 0));
                }
                else if (m >= 96 && m < 96 + LH) commit_h(m - 96, sA);
            };
            auto slotted = [&](int m0) { return m0 >= 0 && (m0 < 2 + LH + (ACC ? 16 : 0) || (m0 + 4 > 96 && m0 < 96 + LH)); };
            ldfrag(0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int s = g & 1;
                if (g + 1 < NG) ldfrag(g + 1, s ^ 1);
                const int m0 = (g - 1) * 4;
                const bool slt = slotted(m0);
                acc = MFMA32(av[s].x, bv[s].x, acc);
                if (slt) { slot(m0); __builtin_amdgcn_sched_barrier(0); }
                acc = MFMA32(av[s].y, bv[s].y, acc);
                if (slt) { slot(m0 + 1); __builtin_amdgcn_sched_barrier(0); }
                acc = MFMA32(av[s].z, bv[s].z, acc);
                if (slt) { slot(m0 + 2); __builtin_amdgcn_sched_barrier(0); }
                acc = MFMA32(av[s].w, bv[s].w, acc);
                if (slt) { slot(m0 + 3); __builtin_amdgcn_sched_barrier(0); }
                if (!slt) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
                if (g == 11) __syncthreads();       // every wave has consumed its last fragment of the oldest row (groups 0..11)
            }
            pending = false;
#pragma unroll
            for (int r = 0; r < 16; ++r) done[r] = fmaxf(acc[r] + bv0, lo) + (ACC ? yold[r] : 0.f);
            pending = true;
            dg = g0;
            if (a.stats) {       // uniform
                float vals[16], s1, s2;
#pragma unroll
                for (int r = 0; r < 16; ++r) vals[r] = done[r];
                lane_stats<16>(vals, s1, s2);
                stat_merge_eq(s1, s2, __shfl_xor(s1, 32, 64), __shfl_xor(s2, 32, 64), 1.f / 32.f);
                if (lane < 32) { Rs[(wv * 32 + lane) * 2] = s1; Rs[(wv * 32 + lane) * 2 + 1] = s2; }
            }
        } else {
            // zero row: nothing to compute; the slot of position p-1 was last read before the previous step's barrier
#pragma unroll
            for (int j = 0; j < LH; ++j) issue_h(j, g2);
#pragma unroll
            for (int j = 0; j < LH; ++j) commit_h(j, sA);
        }
        __syncthreads();
        if (a.stats && g0 >= 0 && tid < 32 && tid < Cout) {
            float s1 = Rs[tid * 2], s2 = Rs[tid * 2 + 1];
            for (int r = 1; r < nwr; ++r) stat_merge(s1, s2, (float)(32 * r), Rs[(r * 32 + tid) * 2], Rs[(r * 32 + tid) * 2 + 1], 32.f);
            float* o = a.stats + ((size_t)g0 * Cout + tid) * 2;
            o[0] = s1;
            o[1] = s2;
        }
        const int t = sA; sA = sB; sB = sC; sC = t;
        g0 = g1; g1 = g2;
        head_next();
    }
    flush_piece(0);
    flush_piece(1);
}


// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same layers on the same walk: dW[co][ky][kx][ci] = sum_p dY[p][co] X[p + ((ky-1) d, (kx-1) d)][ci].
// GEMM view: M = co, N = ci, K = pixels.  Wave w takes pixels 32w..32w+31 of the output row as its K slice and all nine
// taps (9 accumulators = 144 registers).  The A operand (dY) needs no LDS: lane (co, k-half) reads its 16 values of the
// row straight from global memory in MFMA layout (a wave's load = 2 pixels x 32 channels = 256 contiguous bytes), one row
// ahead.  The X rows [pixel][32 ci] sit in a ring of FOUR slots (a row is 32 KB without padding: lanes = channels read
// 64 consecutive floats), so the row prefetched for the next step goes into the free slot and a step needs one barrier.
// A slot has d zero pixels on either side of the row (written once): tap columns outside the image read them, every
// fragment address is base + immediate and the step loop has no address arithmetic.  Each workgroup folds its 8
// waves through LDS and writes one slab [Cout][9][32]; reduce_rows sums the slabs in a fixed order (deterministic).
struct DilWgArgs {
    const float* x;
    const float* dy;
    float* part;
    int N, H, W, Cout, dil;
    int per, total;
    unsigned nbx, nbd;
};

template <int NW>
__global__ void __launch_bounds__(64 * NW, 1) k_conv_dilrow_wgrad(DilWgArgs a) {
    constexpr int NT = 64 * NW, WMAX = 32 * NW;
    constexpr int LH = WMAX * 8 / NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                     // [4][SLOT]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int H = a.H, W = a.W, Cout = a.Cout, d = a.dil;
    const int SLOT = (W + 2 * d) * 32;    // floats per row slot: d zero pixels, the row, d zero pixels
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(a.x, a.nbx), rsd = make_rsrc(a.dy, a.nbd);

    const int t0 = blockIdx.x * a.per;
    const int nsteps = min(a.per, a.total - t0);
    if (nsteps <= 0) return;              // uniform per workgroup
    for (int e = tid; e < 4 * 2 * d * 32; e += NT) {          // the zero pixels of the four slots
        const int sl = e / (2 * d * 32), r = e - sl * (2 * d * 32);
        Xs[sl * SLOT + (r < d * 32 ? r : r + W * 32)] = 0.f;
    }

    const unsigned row_bytes = (unsigned)W * 128u;
    unsigned h_off[LH];
#pragma unroll
    for (int j = 0; j < LH; ++j) {
        const int f = tid + j * NT;
        h_off[j] = f < W * 8 ? (unsigned)f * 16u : 0xFFFFFFFFu;
    }
    float4 rh[LH];
    auto issue_h = [&](int j, int g) {
        const bool ok = (g >= 0) & (h_off[j] != 0xFFFFFFFFu);
        rh[j] = buf_ld4(rsx, sel_u32(ok, (unsigned)g * row_bytes + h_off[j], a.nbx));
    };
    // (a slot past the end of a row narrower than 256 pixels loaded zeros: it parks them in the slot's first zero pixel)
    int h_lds[LH];
#pragma unroll
    for (int j = 0; j < LH; ++j) h_lds[j] = h_off[j] != 0xFFFFFFFFu ? d * 32 + (tid + j * NT) * 4 : (tid & 7) * 4;
    auto commit_h = [&](int j, int slot_f) { *(float4*)&Xs[slot_f + h_lds[j]] = rh[j]; };

    // head: a position of the sequence = image hn, row hy of chain hy % d (the H rows of an image in chain order, images
    // concatenated; no zero rows here: the taps of a chain's first / last row that fall outside the image are masked)
    int hn, hy;
    {
        const int tt = t0 > 0 ? t0 - 1 : 0;
        hn = tt / H;
        int u = tt - hn * H, rho = 0;
        for (;;) {
            const int L = (H - rho + d - 1) / d;      // rows of chain rho (>= 1: d <= H)
            if (u < L) break;
            u -= L;
            ++rho;
        }
        hy = rho + u * d;
    }
    auto head_row = [&]() { return hn < a.N ? hn * H + hy : -1; };
    auto head_next = [&]() {
        hy += d;
        if (hy >= H) {
            hy = hy % d + 1;             // first row of the next chain
            if (hy == d) { hy = 0; ++hn; }
        }
    };

    // dY fragments: lane (co = lane & 31, k-half = lane >> 5), k-step s -> pixel 32 wv + 2 s + k-half of the row
    const int lcol = lane & 31, lk = lane >> 5;
    const unsigned d_row = (unsigned)W * (unsigned)Cout * 4u;
    const unsigned d_lane = lcol < Cout ? ((unsigned)(wv * 32 + lk) * (unsigned)Cout + (unsigned)lcol) * 4u : 0xFFFFFFFFu;
    const bool d_ok = (lcol < Cout) & (wv * 32 < W);
    float da[16];       // the row's 16 k-steps; value s is reloaded for the next row as soon as k-step s has consumed it
    auto issue_d = [&](int s, int g) {
        const unsigned off = sel_u32((g >= 0) & d_ok, (unsigned)g * d_row + d_lane, a.nbd);
        da[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsd, (int)off, s * 2 * Cout * 4, 0));
    };

    // prologue: X rows at positions t0-1, t0, t0+1 into slots 0, 1, 2; dY row of position t0
    int g0 = -1, g1 = -1;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int g = (s == 0 && t0 == 0) ? -1 : head_row();
#pragma unroll
        for (int j = 0; j < LH; ++j) issue_h(j, g);
#pragma unroll
        for (int j = 0; j < LH; ++j) commit_h(j, s * SLOT);
        if (!(s == 0 && t0 == 0)) head_next();
        g0 = g1;
        g1 = g;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) issue_d(s, g0);
    int sA = 0, sB = SLOT, sC = 2 * SLOT, sD = 3 * SLOT;     // slots of positions p-1, p, p+1 and the free one
    __syncthreads();

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // X fragment addresses (floats, relative to a slot): pixel d + 32 wv + k + k-half + (kx-1) d, channel = lane & 31.
    // A wave past the row's end (W < 256) stays inside the slot by reading pixel 0 onwards; its dY is zero.
    const int pw = wv * 32 < W ? wv * 32 + lk : 0;
    const int c0 = pw * 32 + lcol, c1 = (pw + d) * 32 + lcol, c2 = (pw + 2 * d) * 32 + lcol;

    for (int i = 0; i < nsteps; ++i) {
        const int g2 = (i + 2 <= nsteps) ? head_row() : -1;
        const int gd = (i + 1 < nsteps) ? g1 : -1;           // dY row of the next position
        // rows y - d / y + d outside the image: the slot holds a row of the neighbouring chain, the taps get a zero dY
        const int y = g0 % H;
        const float m0 = y >= d ? 1.f : 0.f, m2 = y + d < H ? 1.f : 0.f;
        float fb[2][9];
        const float* xA0 = Xs + sA + c0; const float* xA1 = Xs + sA + c1; const float* xA2 = Xs + sA + c2;
        const float* xB0 = Xs + sB + c0; const float* xB1 = Xs + sB + c1; const float* xB2 = Xs + sB + c2;
        const float* xC0 = Xs + sC + c0; const float* xC1 = Xs + sC + c1; const float* xC2 = Xs + sC + c2;
        auto ldfrag = [&](int ks, int s) {
            const int k = 2 * ks * 32;
            fb[s][0] = xA0[k]; fb[s][1] = xA1[k]; fb[s][2] = xA2[k];
            fb[s][3] = xB0[k]; fb[s][4] = xB1[k]; fb[s][5] = xB2[k];
            fb[s][6] = xC0[k]; fb[s][7] = xC1[k]; fb[s][8] = xC2[k];
        };
        auto slot = [&](int m) {         // m: MFMA position behind k-step 0
            if (m % 9 == 0) issue_d(m / 9, gd);           // first MFMA of k-step s+1: k-step s has consumed da[s]
            else if (m >= 1 && m < 1 + LH) issue_h(m - 1, g2);
            else if (m >= 100 && m < 100 + LH) commit_h(m - 100, sD);
        };
        ldfrag(0, 0);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int s = ks & 1;
            if (ks + 1 < 16) ldfrag(ks + 1, s ^ 1);
            const int m0p = (ks - 1) * 9;
            const float f1 = da[ks], f0 = f1 * m0, f2 = f1 * m2;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                acc[tp] = MFMA32(tp < 3 ? f0 : tp < 6 ? f1 : f2, fb[s][tp], acc[tp]);
                if (ks > 0) { slot(m0p + tp); __builtin_amdgcn_sched_barrier(0); }
            }
            if (ks == 0) {
                __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);
            }
        }
        issue_d(15, gd);
        __syncthreads();
        const int t = sA; sA = sB; sB = sC; sC = sD; sD = t;
        g0 = g1; g1 = g2;
        head_next();
    }

    // fold the 8 waves' partial sums through LDS (the row slots are free now) and write ONE slab per workgroup
    float* red = smem;                    // [NW][32 co][32 ci]
    float* o = a.part + (size_t)blockIdx.x * Cout * 9 * 32;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            red[wv * 1024 + row * 32 + (lane & 31)] = acc[t][r];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 1024 / NT; ++e) {
            const int idx = tid + e * NT;         // element of the 32 x 32 tile
            const int co = idx >> 5, ci = idx & 31;
            float v = red[idx];
#pragma unroll
            for (int w = 1; w < NW; ++w) v += red[w * 1024 + idx];
            if (co < Cout) o[((size_t)co * 9 + t) * 32 + ci] = v;
        }
        __syncthreads();
    }
}

}  // namespace

// 3x3, dilation 2..H, one 32-channel source, at most 32 couts, whole rows of at most 256 pixels
bool conv_dil_fwd_ok(const ConvIn& in, int N, int H, int W, int Cout, int ks, int dil) {
    if (g_dil_mode != 0 || !g_dil_env || ks != 3 || dil < 2 || dil > H) return false;
    if (in.C0 != 32 || in.C1 != 0 || in.up0 || Cout < 1 || Cout > 32) return false;
    if (W % 32 != 0 || W > 256) return false;
    return (long)N * H * W * 32 * 4 <= 0xFFFFFFE0L;
}
int conv_dil_stat_tiles(int H) { return H; }

int conv_dil_fwd(const ConvIn& in, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int dil, int relu,
                 hipStream_t st, float* stats, int accumulate) {
    constexpr int NW = 8, KP = 36;
    constexpr size_t lds = (size_t)(3 * (32 * NW + 1) * KP + 9 * 32 * KP + NW * 32 * 2) * sizeof(float);
    static_assert(lds <= 160 * 1024, "row slots do not fit the 160 KB LDS");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_conv_dilrow<NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_conv_dilrow<NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            vqw_set_error("conv_dil: cannot raise the dynamic LDS limit");
            return VQW_ERR_HIP;
        }
        attr_set = true;
    }
    DilArgs a;
    a.x = in.src0; a.w = w; a.bias = bias; a.y = y;
    a.N = N; a.H = H; a.W = W; a.Cout = Cout; a.dil = dil;
    a.total = N * (H + dil);
    const int blocks = a.total < g_max_blocks ? a.total : g_max_blocks;
    a.per = ceil_div(a.total, blocks);
    a.relu = relu;
    a.stats = stats;
    const long P = (long)N * H * W;
    a.nbx = (unsigned)(P * 32 * 4);
    a.nbw = (unsigned)((long)Cout * 9 * 32 * 4);
    a.nby = (unsigned)(P * Cout * 4);
    if (accumulate) k_conv_dilrow<NW, true><<<ceil_div(a.total, a.per), 64 * NW, lds, st>>>(a);
    else k_conv_dilrow<NW, false><<<ceil_div(a.total, a.per), 64 * NW, lds, st>>>(a);
    VQW_LAUNCH_CHECK("conv_dil");
    return VQW_OK;
}

// weight gradient of the same layers: slabs in `ws` (at most max_slabs of Cout * 9 * 32 floats), summed into dw
bool conv_dil_wgrad_ok(const ConvIn& in, int N, int H, int W, int Cout, int ks, int dil) {
    return conv_dil_fwd_ok(in, N, H, W, Cout, ks, dil) && W + 2 * dil <= 320;      // four padded row slots in 160 KB
}
int conv_dil_wgrad(const ConvIn& in, const float* dy, float* dw, float* ws, int max_slabs, int N, int H, int W, int Cout, int dil,
                   int acc, hipStream_t st) {
    constexpr int NW = 8;
    size_t lds = (size_t)4 * (W + 2 * dil) * 32 * sizeof(float);
    if (lds < (size_t)NW * 1024 * sizeof(float)) lds = (size_t)NW * 1024 * sizeof(float);     // the fold buffer
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_conv_dilrow_wgrad<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            vqw_set_error("conv_dil_wgrad: cannot raise the dynamic LDS limit");
            return VQW_ERR_HIP;
        }
        attr_set = true;
    }
    DilWgArgs a;
    a.x = in.src0; a.dy = dy; a.part = ws;
    a.N = N; a.H = H; a.W = W; a.Cout = Cout; a.dil = dil;
    a.total = N * H;
    int blocks = a.total < g_max_blocks ? a.total : g_max_blocks;
    if (blocks > max_slabs) blocks = max_slabs;
    if (blocks < 1) blocks = 1;
    a.per = ceil_div(a.total, blocks);
    const int nsb = ceil_div(a.total, a.per);
    const long P = (long)N * H * W;
    a.nbx = (unsigned)(P * 32 * 4);
    a.nbd = (unsigned)(P * Cout * 4);
    k_conv_dilrow_wgrad<NW><<<nsb, 64 * NW, lds, st>>>(a);
    VQW_LAUNCH_CHECK("conv_dil_wgrad");
    return reduce_rows(ws, dw, (long)Cout * 9 * 32, nsb, st, acc);
}
