// Internal declarations shared by the convolution translation units.
#pragma once
#include "common.h"

// Virtual conv input: channel-concat of src0 (C0 channels; up0=1 -> nearest x2 up-sampled from H/2 x W/2)
// and src1 (C1 channels at full resolution; C1 == 0 -> absent).
struct ConvIn {
    const float* src0;
    const float* src1;
    int C0, C1, up0;
};

// One element of the collapsed weights of a 3x3 layer over a nearest x2 up-sampled input (conv_mfma.hip: conv_up2_prepare):
// i < 4 * Cout * 4 * Cin: wc[par][co][a*2+b][ci] (forward, per output parity); else wd[ci][r*4+s][co] (input gradient)
#ifdef __HIPCC__
static __device__ inline void collapse_up_element(const float* __restrict__ w, float* __restrict__ wc, float* __restrict__ wd, long i, int Cout, int Cin) {
    const long nfwd = 4L * Cout * 4 * Cin;
    if (i < nfwd) {          // wc[par][co][a*2+b][ci]
            int ci = (int)(i % Cin);
            long r = i / Cin;
            int ab = (int)(r % 4);
            r /= 4;
            int co = (int)(r % Cout);
            int par = (int)(r / Cout);
            int py = par >> 1, px = par & 1, a = ab >> 1, b = ab & 1;
            // rows of the 3x3 kernel that fall on low-res row offset a for parity py: (0,0)->{0} (0,1)->{1,2} (1,0)->{0,1} (1,1)->{2}
            int ky0 = (py == 0) ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2), ky1 = (py == 0) ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2);
            int kx0 = (px == 0) ? (b == 0 ? 0 : 1) : (b == 0 ? 0 : 2), kx1 = (px == 0) ? (b == 0 ? 0 : 2) : (b == 0 ? 1 : 2);
            float acc = 0.f;
            for (int ky = ky0; ky <= ky1; ++ky)
                for (int kx = kx0; kx <= kx1; ++kx) acc += w[(((long)co * 3 + ky) * 3 + kx) * Cin + ci];
            wc[i] = acc;
        } else {                 // wd[ci][r*4+s][co], offsets r-1, s-1 in {-1,0,1,2}: {-1}->{2} {0}->{1,2} {1}->{0,1} {2}->{0}
            long k = i - nfwd;
            int co = (int)(k % Cout);
            long r = k / Cout;
            int rs = (int)(r % 16);
            int ci = (int)(r / 16);
            int rr = rs >> 2, ss = rs & 3;
            int ky0 = rr == 0 ? 2 : (rr == 1 ? 1 : 0), ky1 = rr == 0 ? 2 : (rr == 1 ? 2 : (rr == 2 ? 1 : 0));
            int kx0 = ss == 0 ? 2 : (ss == 1 ? 1 : 0), kx1 = ss == 0 ? 2 : (ss == 1 ? 2 : (ss == 2 ? 1 : 0));
            float acc = 0.f;
            for (int ky = ky0; ky <= ky1; ++ky)
                for (int kx = kx0; kx <= kx1; ++kx) acc += w[(((long)co * 3 + ky) * 3 + kx) * Cin + ci];
            wd[k] = acc;
        }
}
#endif

// generic VALU kernels (conv_generic.hip)
int conv_direct_fwd(const ConvIn& in, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int ks,
                    int dil, int relu, hipStream_t st);
int conv_direct_wgrad(const ConvIn& in, const float* dy, float* dw, float* ws, int N, int H, int W, int Cout, int ks, int dil,
                      hipStream_t st, int acc = 0);
int conv_direct_wgrad_splits(long nout, long P);
size_t bias_grad_ws_floats(int C);
int bias_grad(const float* dy, float* dbias, float* ws, long P, int C, hipStream_t st, int acc = 0);
int reduce_rows(const float* part, float* out, long n, int rows, hipStream_t st, int acc = 0);

// MFMA implicit-GEMM kernels (conv_mfma.hip)
bool conv_mfma_fwd_ok(const ConvIn& in, int Cout, int ks);
int conv_mfma_fwd(const ConvIn& in, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int ks, int dil,
                  int relu, hipStream_t st, float* stats = nullptr);
int conv_mfma_stat_tiles(const ConvIn& in, int N, int H, int W, int Cout, int dil);     // partials per plane, 0 = not available
int conv_up2_stat_tiles(int Cin, int Cout, int N, int h, int w);
bool conv_mfma_wgrad_ok(const ConvIn& in, int Cout, int ks);
// LDS-resident halo tiles for 3x3 layers with a narrow cout tile (conv_halo.hip)
extern int g_halo_mode, g_wgrad_tile_mode;
bool conv_wgrad_tile_ok(int C0, int C1, int Cout, int ks, int W, int dil);
int conv_wgrad_tile_blocks(int Cin, int Cout, int N, int H, int W, int max_blocks, int* kt_out);
int conv_wgrad_tile(const ConvIn& in, const float* dy, float* ws, float* bpart, int N, int H, int W, int Cout, int nsb, int kt,
                    hipStream_t st);
bool conv_halo_fwd_ok(const ConvIn& in, int N, int H, int W, int Cout, int ks, int dil);
int conv_halo_fwd(const ConvIn& in, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int relu,
                  hipStream_t st, float* stats = nullptr);
int conv_halo_stat_tiles(const ConvIn& in, int H, int W, int Cout);     // partials per plane written when `stats` is set
bool conv_halo_up2_ok(int Cin, int Cout, int N, int h, int w);         // collapsed up-sampled forward on the halo kernel
int conv_halo_up2_stat_tiles(int h, int w);
int conv_halo_up2_fwd(const float* x_low, const float* wc, const float* bias, float* y, int N, int h, int w, int Cin, int Cout,
                      int relu, hipStream_t st, float* stats = nullptr);
// dilated 3x3 layers on LDS-resident rows walked along the dilation's residue chains (conv_dil.hip)
extern int g_dil_mode;
bool conv_dil_fwd_ok(const ConvIn& in, int N, int H, int W, int Cout, int ks, int dil);
int conv_dil_fwd(const ConvIn& in, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int dil, int relu,
                 hipStream_t st, float* stats = nullptr, int accumulate = 0);
int conv_dil_stat_tiles(int H);
bool conv_dil_wgrad_ok(const ConvIn& in, int N, int H, int W, int Cout, int ks, int dil);
int conv_dil_wgrad(const ConvIn& in, const float* dy, float* dw, float* ws, int max_slabs, int N, int H, int W, int Cout, int dil,
                   int acc, hipStream_t st);     // partials per plane written when `stats` is set
// Winograd F(2x2, 3x3) forward / dgrad of plain 3x3 layers (conv_wino.hip)
extern int g_wino_mode;
bool conv_wino_ok(int Cin, int Cout, int N, int H, int W);
size_t conv_wino_ws_floats(int Cin, int Cout);
int conv_wino_prepare(const float* w, float* u, int Cin, int Cout, hipStream_t st, int transposed = 0);
int conv_wino_stat_tiles(int Cin, int Cout, int H, int W);
// ... on the low-VALU kernel of conv_wino64.hip: 64-cout tile (Cout % 64 == 0) or two M blocks x 32 couts (Cout % 32 == 0, W % 32 == 0)
bool conv_wino64_ok(int Cin, int Cout, int W);
int conv_wino64_stat_tiles(int Cin, int Cout, int H, int W);
bool conv_wino64_split_ok(int Cin, int Cout, int split, int pool0, int N, int H, int W, int c1 = 0);
int conv_wino64_fwd_split(const float* x, const float* u, const float* bias, float* y0, float* y1, int N, int H, int W, int Cin, int Cout,
                          int split, int pool0, int relu, hipStream_t st, int c1 = 0);
int conv_wino64_fwd(const float* x, const float* u, const float* bias, float* y, int N, int H, int W, int Cin, int Cout, int relu,
                    hipStream_t st, float* stats = nullptr, const float* mask = nullptr, int accumulate = 0,
                    const float* in_mr = nullptr, int in_relu = 0, int dil = 1);
bool conv_wino64_dil2_ok(int Cin, int Cout, int H, int W);
// in_mr (with mask = the raw input of the InstanceNorm in front of the layer, stats = [N][tiles][Cout][2]): the launch is the
// layer's input gradient and leaves that norm's backward sums per region.  accumulate (low-VALU kernel only): y += result.  mask (optional, shaped like y, low-VALU kernel only: conv_wino64_ok): outputs are zeroed where mask <= 0
int conv_wino_fwd(const float* x, const float* u, const float* bias, float* y, int N, int H, int W, int Cin, int Cout, int relu,
                  hipStream_t st, float* stats = nullptr, const float* mask = nullptr, int accumulate = 0,
                  const float* in_mr = nullptr, int in_relu = 0);
bool conv_wino_wgrad_ok(int Cin, int Cout, int N, int H, int W);
int conv_wino_wgrad_blocks(const ConvIn& in, int Cout, int N, int H, int W, int max_slabs, int* kt_out);
bool conv_wino64_wgrad_ok(int C0, int C1, int up0, int Cout, int H, int W);
bool conv_wino32_wgrad_ok(int C0, int C1, int up0, int Cout, int H, int W);       // (32 co x 32 ci) blocks: Cout % 64 == 32
int conv_wino32_wgrad_blocks(int Cin, int Cout, int N, int H, int W, int max_slabs, int* kt_out);
int conv_wino32_wgrad(const ConvIn& in, const float* dy, float* ws, float* bpart, int N, int H, int W, int Cin, int Cout, int nsb, int kt,
                      hipStream_t st, int dil = 1);
bool conv_wino32_wgrad_dil2_ok(int C0, int Cout, int H, int W);
int conv_wino64_wgrad_blocks(int Cin, int Cout, int N, int H, int W, int max_slabs, int* kt_out);
int conv_wino64_wgrad(const ConvIn& in, const float* dy, float* ws, float* bpart, int N, int H, int W, int Cin, int Cout, int nsb, int kt,
                      hipStream_t st);
int conv_wino_wgrad(const ConvIn& in, const float* dy, float* ws, float* bpart, int N, int H, int W, int Cout, int nsb, int kt,
                    hipStream_t st);
// 3x3 over an up-sampled input in Winograd form, nine of the sixteen products (conv_wino_up.hip)
bool conv_wino_up_dgrad_ok(int Cin, int Cout, int N, int h, int w);
size_t conv_wino_up_ws_floats(int Cin, int Cout);
int conv_wino_up_prepare(const float* w, float* ws, int Cin, int Cout, hipStream_t st, float* wc = nullptr, float* wd = nullptr);
int conv_wino_up_dgrad(const float* dy, const float* ws, float* g_low, int N, int h, int w, int Cin, int Cout, hipStream_t st,
                       int accumulate = 0);
bool conv_up2_dgrad_is_wino(int Cin, int Cout, int N, int h, int w);
bool conv_up2_fwd_is_wino(int Cin, int Cout, int N, int h, int w);
bool conv_up2_wgrad_is_wino(int Cin, int Cout, int N, int h, int w);
bool conv_wino_up_wgrad_ok(int Cin, int Cout, int N, int h, int w);
size_t conv_wino_up_wgrad_ws_floats(int Cin, int Cout, int N, int h, int w);
int conv_wino_up_wgrad(const float* x_low, const float* dy, float* dw, float* dbias, float* ws, int N, int h, int w, int Cin, int Cout,
                       int acc, hipStream_t st);
bool conv_wino_up_fwd_ok(int Cin, int Cout, int N, int h, int w);
int conv_wino_up_stat_tiles(int h, int w);
int conv_wino_up_fwd(const float* x_low, const float* ws, const float* bias, float* y, int N, int h, int w, int Cin, int Cout, int relu,
                     hipStream_t st, float* stats = nullptr, float* y2 = nullptr, float* stats2 = nullptr);
// collapsed 3x3-over-upsampled forward / dgrad (conv_mfma.hip)
bool conv_up2_ok(int Cin, int Cout, long Plow);
size_t conv_up2_ws_floats(int Cin, int Cout);
int conv_up2_prepare(const float* w, float* ws, int Cin, int Cout, hipStream_t st);
int conv_up2_fwd(const float* x_low, const float* ws, const float* bias, float* y, int N, int h, int w, int Cin, int Cout, int relu,
                 hipStream_t st, float* stats = nullptr);
int conv_up2_dgrad(const float* dy, const float* ws, float* dx_low, int N, int h, int w, int Cin, int Cout, hipStream_t st,
                   int accumulate = 0);      // accumulate: dx_low += result (nine-product kernel only: conv_up2_dgrad_is_wino)
bool conv_up2_wgrad_ok(int Cin, int Cout, int N, int h, int w);
size_t conv_up2_wgrad_ws_floats(int Cin, int Cout, int N, int h, int w);
int conv_up2_wgrad(const float* xlow, const float* dy, float* dw, float* dbias, float* ws, int N, int h, int w, int Cin, int Cout,
                   int acc, hipStream_t st);
size_t conv_mfma_wgrad_ws_floats(int Cin, int Cout, int ks, long P);
bool conv_mfma_wgrad_is_wino(const ConvIn& in, int N, int H, int W, int Cout, int ks, int dil);
int conv_mfma_wgrad(const ConvIn& in, const float* dy, float* dw, float* dbias, int* bias_done, float* ws, int N, int H, int W,
                    int Cout, int ks, int dil, hipStream_t st, int acc);

// 4x4 padding-1 convolutions of the PatchGAN discriminator on the MFMA kernels (conv_mfma.hip; callers in gan.hip)
bool conv_k4_mfma_ok(int Cin, int Cout, long P);
int conv_k4s2_fwd(const float* x_high, const float* w, const float* bias, float* y_low, int N, int h, int w_, int Cin, int Cout,
                  hipStream_t st);
size_t conv_k4s2_dgrad_ws_floats(int Cin, int Cout);
int conv_k4s2_dgrad(const float* gy_low, const float* w, float* ws, float* gx_high, int N, int h, int w_, int Cin, int Cout,
                    hipStream_t st);
bool conv_k4s2_wgrad_ok(int Cin, int Cout, int N, int h, int w);
size_t conv_k4s2_wgrad_ws_floats(int Cin, int Cout, int N, int h, int w);
int conv_k4s2_wgrad(const float* x_high, const float* gy_low, float* dw, float* ws, int N, int h, int w, int Cin, int Cout, int acc,
                    hipStream_t st);
int conv_k4s1_grid(const float* src, const float* w16, const float* bias, float* dst, int N, int H, int W, int Csrc, int Cdst,
                   int tap0, hipStream_t st);
size_t conv_k4s1_wgrad_ws_floats(int Cin, int Cout, long P);
int conv_k4s1_wgrad_grid(const float* x, const float* dy_grid, float* dw, float* ws, int N, int H, int W, int Cin, int Cout, int acc,
                         hipStream_t st);

// thin-channel streams (conv_thin.hip): 1-channel stem, 1-channel 1x1 head
// streaming 1x1 channel mixing on the vector lanes (conv_thin.hip): y[p][n] = sum_k x[p][k] B[k][n], B = [K][Nout]
bool conv_pw_stream_ok(int K, int Nout, int N, int HW);
int conv_pw_stream_stat_tiles(int HW);
int conv_pw_stream(const float* x, const float* Bm, const float* bias, float* y, float* part, int N, int HW, int K, int Nout, int mode,
                   hipStream_t st);
bool conv_stem_ok(const ConvIn& in, int Cout, int ks);
bool conv_stem_wgrad_ok(const ConvIn& in, int Cout, int ks);
int conv_stem_fwd(const ConvIn& in, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int ks, int dil,
                  int relu, hipStream_t st);
size_t conv_stem_wgrad_ws_floats(int Cout);
int conv_stem_wgrad(const ConvIn& in, const float* dy, float* dw, float* dbias, float* ws, int N, int H, int W, int Cout, int ks,
                    int dil, int acc, hipStream_t st);
// streaming 1x1 weight gradient of thin layers on large maps (conv_thin.hip)
bool conv_pw_wgrad_ok(const ConvIn& in, int Cout, int ks, long P);
size_t conv_pw_wgrad_ws_floats(int Cin, int Cout);
int conv_pw_wgrad(const ConvIn& in, const float* dy, float* dw, float* ws, long P, int Cout, int acc, hipStream_t st);
bool conv_head_ok(const ConvIn& in, int Cout, int ks);
int conv_head_fwd(const ConvIn& in, const float* w, const float* bias, float* y, long P, int relu, hipStream_t st);
size_t conv_head_wgrad_ws_floats(int Cin);
int conv_head_wgrad(const ConvIn& in, const float* dy, float* dw, float* dbias, float* ws, long P, int acc, hipStream_t st);
