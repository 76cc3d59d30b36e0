/*
 * vqwnet_hip.h — C ABI of libvqwnet_hip.so: the MI355X (gfx950) kernels behind the
 * VQ-W-Net training hot path of Kaz-K/medical-image-editing.
 *
 * The reference has no FFI of its own: every operator below replaces a stock
 * ATen call made from the reference's torch.nn modules (citations are to
 * /root/reference/src).  The drop-in boundary is therefore this library, bound
 * with ctypes from the Python host code in medical-image-editing_amd/hipops,
 * which re-implements the reference's nn.Module classes on top of it.
 *
 * Conventions
 *   - All tensor arguments are DEVICE pointers to dense fp32 in NHWC order
 *     (PyTorch `channels_last`), ids are int64/int32 as stated.
 *   - Conv weights are OHWI: w[co][ky][kx][ci] (PyTorch OIHW storage in
 *     channels_last memory format), i.e. the same bytes a
 *     `weight.contiguous(memory_format=torch.channels_last)` holds.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls
 *     only enqueue work; no call synchronises, allocates or frees device memory
 *     (safe for hipGraph capture).  Scratch comes in through `ws` arguments whose
 *     size the matching *_ws_bytes() query returns.
 *   - Return value: 0 = ok, <0 = error (argument / launch); the message is
 *     available from vqw_last_error() (thread-local).  The Python binding turns a
 *     non-zero status into RuntimeError, mirroring the reference's assert /
 *     torch error behaviour.
 */
#ifndef VQWNET_HIP_H
#define VQWNET_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* vqw_last_error(void);
int vqw_abi_version(void);
/* 0 = auto (MFMA kernels when shapes allow), 1 = force the generic VALU kernels, 2 = MFMA kernels but
 * never the LDS-resident halo-tile forward (A/B timing, tests), 3 = auto but never a Winograd-form kernel (every plain
 * 3x3 layer in direct form: the reference for "the training forward's codebook indices do not depend on the Winograd
 * kernels").  Returns the previous mode. */
int vqw_set_conv_backend(int mode);
/* Measurement aid (bench.py roofline): HIP events recorded on the launch stream around every convolution kernel
 * family - and around the HBM-bound normalisation / element-wise entry points - between begin and end.  end() synchronises
 * on those events and fills out[6][4] = {launches, total ms, total FLOPs, total algorithmic bytes} for {MFMA fwd/dgrad, MFMA
 * wgrad, generic fwd, generic wgrad, Winograd-form fwd/dgrad/wgrad, HBM-bound norm / element-wise (bytes = tensor passes as
 * launched)}; FLOPs are the ones the kernels execute (collapsed up-sampled and Winograd-form layers: 4/9 of the direct
 * form's).  (ABI 6: the sixth family.  ABI 7: vqw_conv3x3_wino_fwd_masked, vqw_conv3x3_wino_fwd_acc, vqw_conv3x3_up2_dgrad_acc,
 * vqw_conv3x3_wino_fwd_inbwd, vqw_inorm_bwd_parts.)
 * Not meant for graph capture; off by default.                                                              */
int vqw_profile_begin(void);
int vqw_profile_end(double* out);
/* Which families record events (bit f = family f; default all).  bench.py keeps the ~390 norm / element-wise launches of a
 * step out of its TIMED region (two event records each would cost the step 2 %) and times them in the serialised pass.
 * Returns the previous mask. */
int vqw_profile_families(unsigned mask);

/* ---- convolution: replaces F.conv2d fwd/bwd behind nn.Conv2d in
 *      networks/blocks.py:5-6,25,45,48,75,79,80,102,108,112,117; networks/aspp.py:19-24;
 *      networks/unet_decoder.py:105.
 * The input is the *virtual* channel-concat [src0 (C0 ch, optionally nearest x2
 * up-sampled from H/2 x W/2), src1 (C1 ch, may be NULL/0)] so that
 * nn.Upsample + torch.cat (blocks.py:12,16-17,106,123) are never materialised.
 * ksize in {1,3}; stride 1; padding = dil*(ksize/2) ("same").                      */
int vqw_conv2d_fwd(const float* src0, int C0, int up0, const float* src1, int C1,
                   const float* w_ohwi, const float* bias, float* y,
                   int N, int H, int W, int Cout, int ksize, int dil, int relu, void* stream);
/* dgrad weights: wt[ci][2-ky][2-kx][co] = w[co][ky][kx][ci]; then
 * dX = vqw_conv2d_fwd(dY, Cout, 0, NULL, 0, wt, NULL, dX, N,H,W, Cin, ksize, dil, 0).
 * relu=1 fuses nn.ReLU into the epilogue (blocks.py:75-77 mlp_shared).             */
int vqw_pack_dgrad_weights(const float* w_ohwi, float* wt, int Cout, int Cin, int ksize, void* stream);
/* conv -> InstanceNorm (blocks.py:45-49): the convolution's epilogue also leaves the norm's statistics as per-tile
 * partials part[N][parts][Cout][2] = (sum, M2 = sum (x - tile mean)^2) of one equal-sized tile each (fp32, merged pairwise
 * inside the tile; the norm combines tiles in double: no fp32 value ever holds a sum of squares), so
 * the norm skips its reduction pass (vqw_inorm_fwd_parts).  ..._stats_parts() returns `parts` for a shape, 0 when
 * the shape is not served (the halo-tile and the implicit-GEMM kernel are; then use vqw_conv2d_fwd + vqw_inorm_fwd).
 * No ReLU epilogue.                                                                                              */
int vqw_conv2d_fwd_stats_parts(int C0, int C1, int up0, int N, int H, int W, int Cout, int ksize, int dil);
int vqw_conv2d_fwd_stats(const float* src0, int C0, int up0, const float* src1, int C1,
                         const float* w_ohwi, const float* bias, float* y, float* part,
                         int N, int H, int W, int Cout, int ksize, int dil, void* stream);
/* y += conv(src0) ('same' conv, no bias): the input gradient of one of several convolutions of the same tensor summed in
 * place (aspp.py:44-47: five branches of one input; autograd would add their gradients pairwise, three passes each).
 * Served for the shapes of the row-chain kernel (dilated 3x3, 32 channels, rows <= 256 pixels) and (ABI 8) for 1x1 layers on
 * the implicit-GEMM kernel, whose epilogue then adds to y: query ..._supported. */
int vqw_conv2d_fwd_acc_supported(int C0, int N, int H, int W, int Cout, int ksize, int dil);
int vqw_conv2d_fwd_acc(const float* src0, int C0, const float* w_ohwi, float* y, int N, int H, int W, int Cout, int ksize,
                       int dil, void* stream);
size_t vqw_conv2d_wgrad_ws_bytes(int C0, int C1, int N, int H, int W, int Cout, int ksize);
/* dW[co][ky][kx][ci] (OHWI) and, if dbias != NULL, dbias[co] = sum_p dY.
 * accumulate=1 adds into dw / dbias (a layer used by both views of a step: one gradient buffer, no extra pass). */
int vqw_conv2d_wgrad(const float* src0, int C0, int up0, const float* src1, int C1,
                     const float* dy, float* dw_ohwi, float* dbias, void* ws, size_t ws_bytes,
                     int N, int H, int W, int Cout, int ksize, int dil, int accumulate, void* stream);
/* 3x3 conv over a nearest x2 up-sampled single source (StyledResUpBlock conv / conv1, blocks.py:106,117,123-126),
 * collapsed onto the low-resolution grid: four 2x2 convs forward, one 4x4 stride-2 gather for the input gradient —
 * 4/9 of the FLOPs of the direct form and no full-resolution gradient intermediate.  x_low is [N,h,w,Cin], y and dy
 * are [N,2h,2w,Cout].  prepare() writes the tap-summed weights for both directions into ws (valid while w is unchanged). */
int vqw_conv3x3_up2_supported(int Cin, int Cout, int N, int h, int w);
size_t vqw_conv3x3_up2_ws_bytes(int Cin, int Cout);
int vqw_conv3x3_up2_prepare(const float* w_ohwi, void* ws, size_t ws_bytes, int Cin, int Cout, void* stream);
int vqw_conv3x3_up2_fwd(const float* x_low, const void* ws, const float* bias, float* y, int N, int h, int w, int Cin,
                        int Cout, int relu, void* stream);
/* forward that also leaves the following norm's statistics partials (see vqw_conv2d_fwd_stats); parts = 0: not served */
int vqw_conv3x3_up2_fwd_stats_parts(int Cin, int Cout, int N, int h, int w);
int vqw_conv3x3_up2_fwd_stats(const float* x_low, const void* ws, const float* bias, float* y, float* part, int N, int h, int w,
                              int Cin, int Cout, void* stream);
/* (ABI 8) TWO 32-cout layers of the same up-sampled input - StyledResUpBlock's shortcut `conv` and `conv1`, blocks.py:100-112 - as ONE
 * 64-cout launch of the nine-product kernel.  ws = vqw_conv3x3_up2_prepare() of the concatenated weights [w_a | w_b] (Cout = 64),
 * bias_cat = [b_a | b_b] or NULL; y_a / y_b (N, 2h, 2w, 32) and their InstanceNorm / BatchNorm statistics partials part_a / part_b
 * ([N][parts][32][2], parts = the value ..._supported returns; 0 = not served) come out as separate tensors. */
int vqw_conv3x3_up2_fwd_pair_supported(int Cin, int Cout_each, int N, int h, int w);
int vqw_conv3x3_up2_fwd_pair(const float* x_low, const void* ws, const float* bias_cat, float* y_a, float* y_b, float* part_a,
                             float* part_b, int N, int h, int w, int Cin, int Cout_each, void* stream);
int vqw_conv3x3_up2_dgrad(const float* dy, const void* ws, float* dx_low, int N, int h, int w, int Cin, int Cout,
                          void* stream);
/* ABI 7.  dx_low += the same input gradient: the second of the two up-sampled convolutions that read one tensor (a
 * StyledResUpBlock's `conv` and `conv1`, blocks.py:100-112) adds to the first one's result in its epilogue. */
int vqw_conv3x3_up2_dgrad_acc_supported(int Cin, int Cout, int N, int h, int w);
int vqw_conv3x3_up2_dgrad_acc(const float* dy, const void* ws, float* dx_low, int N, int h, int w, int Cin, int Cout,
                              void* stream);
/* weight (and bias) gradient of the same layer on the low-resolution grid (needs w % 16 == 0) */
int vqw_conv3x3_up2_wgrad_supported(int Cin, int Cout, int N, int h, int w);
size_t vqw_conv3x3_up2_wgrad_ws_bytes(int Cin, int Cout, int N, int h, int w);
int vqw_conv3x3_up2_wgrad(const float* x_low, const float* dy, float* dw_ohwi, float* dbias, void* ws, size_t ws_bytes,
                          int N, int h, int w, int Cin, int Cout, int accumulate, void* stream);
/* Plain 3x3 stride-1 convolution (every F.conv2d(x, w, padding=1) of blocks.py / unet_*.py on >= 16 input and >= 32
 * output channels) in Winograd F(2x2, 3x3) form: 4/9 of the direct form's matrix work, same fp32 arithmetic type; the
 * result differs from the direct form by a few ulps of the accumulated magnitude (a different product / summation order).
 * prepare() writes U = G w G^T [16][Cout][Cin] into ws (valid while w is unchanged).  The input gradient is the same
 * call on dy with the U of the packed dgrad weights (vqw_pack_dgrad_weights), roles of Cin / Cout swapped.
 * ..._fwd_stats: also leaves the following norm's statistics partials (see vqw_conv2d_fwd_stats); parts = 0: not served. */
int vqw_conv3x3_wino_supported(int Cin, int Cout, int N, int H, int W);
size_t vqw_conv3x3_wino_ws_bytes(int Cin, int Cout);
int vqw_conv3x3_wino_prepare(const float* w_ohwi, void* ws, size_t ws_bytes, int Cin, int Cout, void* stream);
/* The same for a layer's INPUT-GRADIENT convolution, straight from the layer's own weight (ABI 8): w_ohwi = the layer's
 * [Cin][3][3][Cout] tensor (its couts are this convolution's Cin input channels); equals vqw_pack_dgrad_weights followed by
 * vqw_conv3x3_wino_prepare(.., Cin, Cout), bit for bit, without the packed copy. */
int vqw_conv3x3_wino_prepare_dgrad(const float* w_ohwi, void* ws, size_t ws_bytes, int Cin, int Cout, void* stream);
int vqw_conv3x3_wino_fwd(const float* x, const void* ws, const float* bias, float* y, int N, int H, int W, int Cin, int Cout,
                         int relu, void* stream);
/* ABI 7.  The same convolution with its outputs zeroed where mask <= 0 (mask shaped like y, no bias): the input gradient of
 * a layer whose forward read the output of a fused ReLU (StyledDenorm's mlp_shared -> mlp_gamma | mlp_beta, blocks.py:63-66,
 * 85-87) delivered in front of that ReLU - mask = the ReLU's output, i.e. the layer's own saved input - so that the
 * separate mask pass (vqw_relu_bwd) is not needed.  Served where the 64-cout kernel is (..._masked_supported). */
int vqw_conv3x3_wino_masked_supported(int Cin, int Cout, int N, int H, int W);
int vqw_conv3x3_wino_fwd_masked(const float* x, const void* ws, const float* mask, float* y, int N, int H, int W, int Cin,
                                int Cout, void* stream);
/* y += conv(x) (no bias): a later member of a gradient group - several convolutions of one input tensor (ResBlock's 3x3 and
 * 1x1 branches, blocks.py:14-36; the two mlp_shared convolutions of a StyledResUpBlock's StyledDenorms on one style input,
 * blocks.py:100-134) - adds its input gradient to the shared buffer in its epilogue instead of leaving the sum to a separate
 * add pass.  Served where ..._masked_supported says so. */
int vqw_conv3x3_wino_fwd_acc(const float* x, const void* ws, float* y, int N, int H, int W, int Cin, int Cout, void* stream);
/* A 3x3 layer of dilation 2 (the pyramid's first branch, aspp.py:27-30) in Winograd form (ABI 8): the plain kernel on the four
 * phase images of x and y; ws = vqw_conv3x3_wino_prepare of the layer (forward) or vqw_conv3x3_wino_prepare_dgrad (input
 * gradient: x = dY, Cin / Cout swapped).  part (optional, forward with relu == 0): the following norm's statistics partials
 * [N][parts][Cout][2], parts = ..._dil2_stats_parts; accumulate: y += result (no bias, no ReLU).  The weight gradient of such a
 * layer takes the same route inside vqw_conv2d_wgrad.  Served where ..._dil2_supported says so (H even, W a multiple of 64). */
int vqw_conv3x3_wino_dil2_supported(int Cin, int Cout, int N, int H, int W);
int vqw_conv3x3_wino_dil2_stats_parts(int Cin, int Cout, int N, int H, int W);
int vqw_conv3x3_wino_dil2_fwd(const float* x, const void* ws, const float* bias, float* y, float* part, int accumulate,
                              int N, int H, int W, int Cin, int Cout, int relu, void* stream);
/* One launch, two output tensors (ABI 8): couts [0, split) -> y0 [N,H,W,split] - or, with pool0, summed over each 2 x 2 output
 * tile into y0 [N,H/2,W/2,split] - and couts [split, Cout) -> y1 [N,H,W,Cout-split].  Two uses: (i) the input gradient of a 3x3
 * layer over [nearest-up2x(a) | b] (UpBlock, blocks.py:9-18 with unet_encoder.py's torch.cat): x = dY, ws = the transformed
 * input-gradient weights, Cout = Ca + Cb, split = Ca, pool0 = 1: the gradients of a and b leave the epilogue, replacing two
 * vqw_input_grad_gather passes over the concatenated gradient; (ii) two layers of one input on concatenated weights (the two
 * mlp_shared convolutions of a StyledResUpBlock's StyledDenorms, blocks.py:72-75, 100-134): pool0 = 0, bias / relu as in
 * vqw_conv3x3_wino_fwd.  split % 16 == 0; served where ..._split_supported says so. */
int vqw_conv3x3_wino_split_supported(int Cin, int Cout, int split, int pool0, int N, int H, int W);
int vqw_conv3x3_wino_fwd_split(const float* x, const void* ws, const float* bias, float* y0, float* y1, int N, int H, int W,
                               int Cin, int Cout, int split, int pool0, int relu, void* stream);
/* The same for a layer widened to the kernel's cout tile (ABI 8): ws = the transformed weights of Cout couts of which only
 * split + c1 are real (zero weights behind them), y1 has c1 channels, the padding couts are not stored.  The input gradient of the
 * 48-channel two-source layer at the encoder's full-resolution level (unet_encoder.py's last UpBlock) runs as a 64-cout launch. */
int vqw_conv3x3_wino_split_padded_supported(int Cin, int Cout, int split, int c1, int pool0, int N, int H, int W);
int vqw_conv3x3_wino_fwd_split_padded(const float* x, const void* ws, const float* bias, float* y0, float* y1, int N, int H, int W,
                                      int Cin, int Cout, int split, int c1, int pool0, int relu, void* stream);
/* The input gradient of a layer whose forward read the output of an InstanceNorm (+ReLU) (DoubleConv: conv -> norm -> ReLU ->
 * conv, blocks.py:39-61): y = the gradient as usual, and part[N][parts][Cout][2] = that norm's backward sums per region,
 * (sum gm, sum gm * xhat) with xhat = (norm_x - mean) * rstd and gm = the gradient where the norm's ReLU passed - what
 * vqw_inorm_bwd otherwise reduces in a pass of its own over norm_x and the gradient (vqw_inorm_bwd_parts takes them from here).
 * norm_x: the norm's raw input, shaped like y; norm_mean_rstd: its (mean, rstd) per (image, channel).  parts = 0: not served. */
int vqw_conv3x3_wino_fwd_inbwd_parts(int Cin, int Cout, int N, int H, int W);
int vqw_conv3x3_wino_fwd_inbwd(const float* x, const void* ws, const float* norm_x, const float* norm_mean_rstd, int norm_relu,
                               float* y, float* part, int N, int H, int W, int Cin, int Cout, void* stream);
int vqw_conv3x3_wino_fwd_stats_parts(int Cin, int Cout, int N, int H, int W);
int vqw_conv3x3_wino_fwd_stats(const float* x, const void* ws, const float* bias, float* y, float* part, int N, int H, int W,
                               int Cin, int Cout, void* stream);
/* Gradient of the virtual input: g_full is [N,H,W,Ctot]; takes channels
 * [c_off, c_off+C).  up=1: dst[N,H/2,W/2,C] = 2x2 block sums; up=0: plain slice copy.
 * accumulate=1 adds into dst.                                                      */
int vqw_input_grad_gather(const float* g_full, int Ctot, int c_off, int C, int up,
                          float* dst, int accumulate, int N, int H, int W, void* stream);

/* ---- InstanceNorm2d(affine=False) [+ReLU]: blocks.py:26,46-47,49-50,118-119; aspp.py:25-28 */
size_t vqw_plane_ws_bytes(int N, int C, int HW);
/* y / gy may be a channel slice [c_off, c_off+C) of a wider NHWC tensor with `cstride` channels
 * (cstride == C, c_off == 0 for a plain tensor): the ASPP concat (aspp.py:47) is written in place. */
int vqw_inorm_fwd(const float* x, float* y, int y_cstride, int y_coff, float* mean_rstd /*[N][C][2]*/,
                  void* ws, size_t ws_bytes, int N, int HW, int C, float eps, int relu, void* stream);
/* the same with the statistics taken from the producing convolution's partials (vqw_conv2d_fwd_stats) */
int vqw_inorm_fwd_parts(const float* x, float* y, int y_cstride, int y_coff, float* mean_rstd, const float* part,
                        int nparts, int N, int HW, int C, float eps, int relu, void* stream);
/* statistics only: mean_rstd[N][C][2] by reduction over x, or from a convolution's partials */
int vqw_inorm_stats(const float* x, float* mean_rstd, void* ws, size_t ws_bytes, int N, int HW, int C, float eps, void* stream);
int vqw_inorm_stats_parts(const float* part, int nparts, float* mean_rstd, int N, int HW, int C, float eps, void* stream);
/* Two norms of one shape in one launch (ABI 8; a ResBlock's main and 1x1 branches, blocks.py:21-36). */
int vqw_inorm_stats_parts2(const float* part_a, int nparts_a, float* mean_rstd_a, const float* part_b, int nparts_b,
                           float* mean_rstd_b, int N, int HW, int C, float eps, void* stream);
/* y = a + InstanceNorm(+ReLU)(x) from x's statistics mean_rstd [N][C][2] (ABI 8): the residual add behind a block that ends in
 * that norm (the decoder's tail `x + conv_last(x)`, unet_decoder.py:169-171); the normalised tensor is never written.  The
 * backward is vqw_inorm_bwd on the gradient of y (and the gradient of `a` is that gradient).  C / 4 a power of two <= 256. */
int vqw_inorm_add_supported(int C);
int vqw_inorm_add_fwd(const float* x, const float* mean_rstd, const float* a, float* y, int N, int HW, int C, int relu, void* stream);
int vqw_inorm_bwd(const float* x, const float* mean_rstd, const float* gy, int gy_cstride, int gy_coff,
                  float* gx, void* ws, size_t ws_bytes, int N, int HW, int C, int relu, void* stream);
/* ABI 7.  The same with the two sums taken from part[N][nparts][C][2] = (sum gm, sum gm * xhat) per region, left by the
 * consumer convolution's input-gradient launch (vqw_conv3x3_wino_fwd_inbwd); means_ws: N * C * 2 floats of scratch. */
int vqw_inorm_bwd_parts(const float* x, const float* mean_rstd, const float* gy, const float* part, int nparts, float* means_ws,
                        float* gx, int N, int HW, int C, int relu, void* stream);
/* backward of two InstanceNorms fed with the SAME gradient (the two branches in front of a ResBlock tail; a: norm + ReLU,
 * b: norm): the common gradient is read once per pass.  ws: 2 x vqw_plane_ws_bytes(N, C, HW).  C % 4 == 0.          */
int vqw_inorm_bwd_pair(const float* xa, const float* mra, const float* xb, const float* mrb, const float* gy,
                       float* gxa, float* gxb, void* ws, size_t ws_bytes, int N, int HW, int C, void* stream);
/* vqw_res_tail_bwd followed by vqw_inorm_bwd_pair as one entry point (ABI 8; a ResBlock's tail, blocks.py:29-36): the first
 * kernel forms g = [out > 0] (g_out + the pooled gradient routed to each window's first maximum), stores it in `g` (a workspace
 * tensor shaped like out) and adds it to both norms' backward sums in the same pass; the apply pass reads g once.  g_pooled or
 * g_out may be NULL.  ws as for vqw_inorm_bwd_pair. */
int vqw_res_tail_bwd_pair(const float* out, const float* g_pooled, const float* g_out, const float* xa, const float* mra,
                          const float* xb, const float* mrb, float* g, float* gxa, float* gxb, void* ws, size_t ws_bytes,
                          int N, int H, int W, int C, void* stream);

/* ---- StyledDenorm = BatchNorm2d(affine=False)(x)*(1+gamma)+beta [+ReLU]: blocks.py:82-90,126-132.
 * training=1: batch statistics, running stats updated in place (momentum, unbiased var);
 * training=0: running stats.  stats_io: training fwd receives per-channel
 * [sum, sumsq] partial totals via (sum_out) for cross-rank reduction: see vqw_bn_* below.  */
int vqw_bn_partial_stats(const float* x, double* sums /*[C][2]*/, void* ws, size_t ws_bytes,
                         int N, int HW, int C, void* stream);
/* sums[C][2] from the per-tile partials part[rows = N * parts][C][2] of the producing convolution (vqw_conv2d_fwd_stats):
 * each partial is (sum, M2 = sum (x - tile mean)^2) of one tile of `tile_count` = N*HW / rows pixels */
int vqw_bn_stats_from_parts(const float* part, double* sums /*[C][2]*/, int rows, int C, double tile_count, void* stream);
int vqw_bn_finalize(const double* sums /*[C][2]*/, double count, float* mean_rstd /*[C][2]*/,
                    float* running_mean, float* running_var, float momentum, float eps,
                    int C, void* stream);
/* vqw_bn_stats_from_parts + vqw_bn_finalize in one launch (ABI 8): for a BatchNorm without a collective between its sums and its
 * statistics (one rank, or SyncBN off).  `sums` [C][2] doubles is written as vqw_bn_stats_from_parts writes it. */
int vqw_bn_finalize_parts(const float* part, int rows, double tile_count, double* sums, double count, float* mean_rstd,
                          float* running_mean, float* running_var, float momentum, float eps, int C, void* stream);
int vqw_bn_eval_stats(const float* running_mean, const float* running_var, float* mean_rstd,
                      float eps, int C, void* stream);
/* gamma / beta (and dgamma / dbeta) are addressed as ptr[pixel * gb_stride + c]: gb_stride = C for two dense
 * maps, 2C when mlp_gamma and mlp_beta were evaluated as one conv with concatenated output channels
 * (gamma = gb, beta = gb + C). */
int vqw_spade_fwd(const float* x, const float* mean_rstd /*[C][2]*/, const float* gamma,
                  const float* beta, int gb_stride, float* y, long P, int C, int relu, void* stream);
/* the same with the block's residual added after the activation: y = act(...) + res (blocks.py:134, `shortcut + main`) */
int vqw_spade_fwd_res(const float* x, const float* mean_rstd, const float* gamma, const float* beta, int gb_stride,
                      const float* res, float* y, long P, int C, int relu, void* stream);
/* The same with the residual given RAW together with its InstanceNorm statistics (ABI 8): y = act(spade(x)) +
 * InstanceNorm(+ReLU)(res_raw), res_mean_rstd [N][C][2] - the shortcut branch of a StyledResUpBlock (blocks.py:113-116, 134)
 * normalised while it is read, its normalised tensor never written.  HW a power of two, C / 4 a power of two <= 256. */
int vqw_spade_fwd_res_norm_supported(long HW, int C);
int vqw_spade_fwd_res_norm(const float* x, const float* mean_rstd, const float* gamma, const float* beta, int gb_stride,
                           const float* res_raw, const float* res_mean_rstd, int res_relu, float* y, int N, long HW, int C,
                           int relu, void* stream);
/* backward, phase 1: dgamma, dbeta and per-channel sums [sum dxhat, sum dxhat*xhat] */
int vqw_spade_bwd_reduce(const float* x, const float* mean_rstd, const float* gamma, const float* beta,
                         const float* gy, float* dgamma, float* dbeta, int gb_stride,
                         double* sums /*[C][2]*/, void* ws, size_t ws_bytes, int N, int HW, int C,
                         int relu, void* stream);
/* phase 2: dx (training: sums/count terms; training=0: dx = dxhat*rstd) */
int vqw_spade_bwd_apply(const float* x, const float* mean_rstd, const float* gamma, const float* beta,
                        int gb_stride, const float* gy, const double* sums, double count, float* gx,
                        long P, int C, int relu, int training, void* stream);

/* ---- element-wise / pooling: blocks.py:29-30,34-36 (add, ReLU, MaxPool2d(2)), 134; unet_decoder.py:107,159-163 */
int vqw_add(const float* a, const float* b, float* y, long n, int relu, void* stream);
int vqw_relu_bwd(const float* y, const float* gy, float* gx, long n, void* stream);
int vqw_maxpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
int vqw_maxpool2_bwd(const float* x, const float* gy, const float* g_skip /*nullable*/, float* gx,
                     int N, int H, int W, int C, void* stream);
/* ResBlock tail (blocks.py:29-36: out = ReLU(a + b); pooled = MaxPool2d(2)(out)), forward and backward in one pass each:
 * gx = [out > 0] * (g_out + g_pooled routed to each 2x2 window's arg-max) = d/da = d/db.  Either gradient may be
 * NULL (that output unused).  H, W even, C % 4 == 0, 16-byte aligned tensors.                                  */
int vqw_res_tail_fwd(const float* a, const float* b, float* out, float* pooled, int N, int H, int W, int C, void* stream);
/* the tail reading the RAW outputs of the block's two conv branches and their InstanceNorm statistics (mean, rstd per
 * (n, c)): a = ReLU(IN(x2)), b = IN(xid), out = ReLU(a + b) — the two normalisation apply passes disappear.       */
int vqw_res_tail_norm_fwd(const float* x2, const float* mr2, const float* xid, const float* mrid, float* out, float* pooled,
                          int N, int H, int W, int C, void* stream);
int vqw_res_tail_bwd(const float* out, const float* g_pooled /*nullable*/, const float* g_out /*nullable*/, float* gx,
                     int N, int H, int W, int C, void* stream);
int vqw_tanh_fwd(const float* x, float* y, long n, void* stream);
int vqw_tanh_bwd(const float* y, const float* gy, float* gx, long n, void* stream);
int vqw_affine(const float* x, float* y, float scale, float shift, long n, void* stream); /* utils norm/denorm */
int vqw_mse_fwd(const float* a, const float* b, float* loss, void* ws, size_t ws_bytes, long n, void* stream);
int vqw_mse_bwd(const float* a, const float* b, const float* gloss, float* ga, long n, void* stream);
size_t vqw_reduce_ws_bytes(long n);
int vqw_weighted_sum(const float* const* terms_dev /*device array of ptrs*/, const float* weights_dev,
                     int n_terms, float* out, void* stream);
/* the same with HOST arrays of (device) term pointers and weights, passed to the kernel by value: 1..16 terms, no
 * host-to-device copy */
int vqw_weighted_sum_host(const float* const* terms /*host array of device ptrs*/, const float* weights /*host*/,
                          int n_terms, float* out, void* stream);

/* ---- vector quantisation: networks/vq/vq_module.py:45-62,159-211; grad_approximation.py:7-29 */
size_t vqw_vq_ws_bytes(long Npix, int D, int K);
/* which search / statistics route (D, K) takes: 0 = codebook in LDS + matrix-core statistics, 1 = codebook in LDS +
 * sorted statistics, 2 = fused MFMA score GEMM / arg-max + sorted statistics, 3 = generic scalar + sorted statistics */
int vqw_vq_plan(int D, int K);
/* x [Npix][D] (NHWC rows), embed [K][D].  Outputs: ids int64 [Npix], q [Npix][D],
 * commit = mean((x-q)^2); when stats != NULL also counts[K] and embed_sum[D][K]
 * (layout of the reference's embed_avg) as double in `stats` = [K + D*K].
 * ids are written as code + id_base (unet_encoder.py:116 adds 1).                   */
int vqw_vq_fwd(const float* x, const float* embed, int64_t* ids, int id_base, float* q, float* commit,
               double* stats, void* ws, size_t ws_bytes, long Npix, int D, int K, void* stream);
/* EMA + Laplace-smoothed normalisation (vq_module.py:195-200), in place on the buffers.
 * sum_scale multiplies embed_sum before the EMA (1/world_size in the reference's quirk mode). */
int vqw_vq_ema_update(const double* stats, float* embed, float* cluster_size, float* embed_avg,
                      float momentum, float eps, float sum_scale, int D, int K, void* stream);
/* One Lloyd iteration of the k-means codebook initialisation (unet_encoder.py:66-91): centres[k] <- mean of its members
 * from the statistics vqw_vq_fwd leaves (codes without members keep their centre).  shift[0] = sum_k |delta_k|_2,
 * shift[1] = number of empty codes.  ws: 16 K bytes. */
int vqw_kmeans_update(const double* stats, float* centres, double* shift, void* ws, size_t ws_bytes, int D, int K,
                      void* stream);
int vqw_vq_lookup(const int64_t* ids, const float* embed, const uint8_t* mask /*nullable*/,
                  const float* scale_dev /*nullable, 1 float*/, float* out, long Npix, int D, int K,
                  void* stream);
/* gx = g_q (straight-through) + g_commit * 2 (x - q) / numel */
int vqw_vq_bwd(const float* x, const float* q, const float* g_q, const float* g_commit,
               float* gx, long numel, void* stream);
/* mask count -> scale = numel / count (run_recon.py:191-192) */
int vqw_mask_scale(const int64_t* label_map, uint8_t* mask, int64_t* ids0, float* scale_dev,
                   long n, void* stream);

/* ---- losses: functions/embed_loss.py:22-88, functions/onehot.py:11-20 */
size_t vqw_cross_ws_bytes(int B, int K, long HW);
/* labels int32 [B][HW] in [0,K] (0 = out of frame); embed [B][HW][D]; codebook [K][D] (= vq.embed).
 * loss = mean over present (b,k) of sum_p |e-c_k|^2 / (cnt+1e-6); coef[B][K] saved for backward. */
int vqw_cross_loss_fwd(const float* embed, const int32_t* labels, const float* codebook_kd,
                       float* loss, float* coef, void* ws, size_t ws_bytes,
                       int B, long HW, int D, int K, void* stream);
int vqw_cross_loss_bwd(const float* embed, const int32_t* labels, const float* codebook_kd,
                       const float* coef, const float* gloss, float* gembed,
                       int B, long HW, int D, int K, void* stream);
/* general (soft / one-hot float r[B][K][HW] NCHW as the reference passes it) */
int vqw_cross_loss_dense_fwd(const float* embed, const float* r_nchw, const float* codebook_kd,
                             float* loss, float* coef, void* ws, size_t ws_bytes,
                             int B, long HW, int D, int K, void* stream);
int vqw_cross_loss_dense_bwd(const float* embed, const float* r_nchw, const float* codebook_kd,
                             const float* coef, const float* gloss, float* gembed,
                             int B, long HW, int D, int K, void* stream);
/* l_dist / l_reg of embed_loss.py:68-88 (no gradient: the codebook is a buffer); ws: 16 K bytes */
int vqw_codebook_losses(const float* codebook_kd, float margin, float* l_dist, float* l_reg,
                        void* ws, size_t ws_bytes, int D, int K, void* stream);
int vqw_onehot(const int32_t* labels, float* out_nchw, int B, long HW, int n_classes, void* stream);
int vqw_flip_labels(const int64_t* ids, int32_t* out, int border, int B, int H, int W, void* stream);

/* ---- optional paths
 * PixelShuffle(2) in NHWC (blocks.py:100-104): (H, W, C) are the high-resolution output dims; inverse=1 is the backward. */
int vqw_pixel_shuffle2(const float* src, float* dst, int N, int H, int W, int C, int inverse, void* stream);
/* DropBlock (dropblock.py:47-94): keep = 1 - dilate(seed), scale = numel/sum(keep); apply is also its own backward. */
int vqw_dropblock_mask(const float* seed, float* keep, float* scale_dev, int N, int H, int W, int block_size, void* stream);
int vqw_dropblock_apply(const float* x, const float* keep, const float* scale_dev, float* y, long P, int C, void* stream);
/* SoftDice + Focal (functions/seg_loss.py:15-62) on NCHW logits / one-hot targets, C <= 64: loss_out = {dice, focal}. */
size_t vqw_seg_ws_bytes(int C);
int vqw_seg_losses_fwd(const float* logits_nchw, const float* target_nchw, float* loss_out, double* sums, void* ws,
                       size_t ws_bytes, int B, long HW, int C, int ignore_index, float smooth, float gamma, float eps,
                       void* stream);
int vqw_seg_losses_bwd(const float* logits_nchw, const float* target_nchw, const double* sums, const float* g_dice,
                       const float* g_focal, float* glogits, int B, long HW, int C, int ignore_index, float smooth,
                       float gamma, float eps, void* stream);

/* ---- two-view augmentation + id-map warps (networks/random_transform.py:10-112; used at
 * single_window_trainer.py:75-76, 91-96).  The reference builds these from kornia 0.5.1, which is not available
 * offline: the arithmetic is the one oracle/augment_ref.py states (parity unpinned).  Matrices are per-sample 3x3,
 * row-major, mapping a DESTINATION pixel (x, y, 1) to the SOURCE pixel; pixel centres sit on integer coordinates.
 * warp_image: bilinear, zero padding, (B, C, H, W) planes.  warp_labels: nearest (round half to even), 0 = out of frame.
 * photometric params per sample {brightness add, contrast multiplier, posterize bits (8 = off), noise std}: clamp(x+b),
 * clamp(x*c), posterize, + std * noise (noise may be NULL).  gauss_blur: separable, reflect border, apply[b] on/off. */
int vqw_warp_image(const float* src, const float* minv /*[B][9]*/, float* dst, int B, int C, int H, int W, void* stream);
int vqw_warp_labels(const void* ids, int ids_are_int64, const float* minv /*[B][9]*/, int32_t* out, int B, int H, int W,
                    void* stream);
int vqw_photometric(const float* x, const float* params /*[B][4]*/, const float* noise, float* y, int B, long per_sample,
                    void* stream);
int vqw_gauss_blur(const float* x, const float* taps /*[K]*/, const unsigned char* apply /*[B] or NULL*/, float* tmp, float* y,
                   int B, int C, int H, int W, int K, void* stream);

/* ---- second training step: PatchGAN discriminator + GAN losses (networks/discriminator.py:18-87,
 * functions/gan_loss.py:6-10, trainers/single_window_trainer.py:434-488).  NHWC activations, OHWI weights
 * [Cout][k][k][Cin]; output size floor((H + 2 pad - k) / stride) + 1; stride 1 or 2; H, W are INPUT dims.
 * sconv_fwd applies LeakyReLU(slope) in the epilogue (slope = 1: none). */
size_t vqw_sconv_fwd_ws_bytes(int N, int H, int W, int Cin, int Cout, int ks, int stride, int pad);
int vqw_sconv_fwd(const float* x, const float* w_ohwi, const float* bias, float* y, void* ws, size_t ws_bytes, int N, int H,
                  int W, int Cin, int Cout, int ks, int stride, int pad, float slope, void* stream);
size_t vqw_sconv_dgrad_ws_bytes(int N, int H, int W, int Cin, int Cout, int ks, int stride, int pad);
int vqw_sconv_dgrad(const float* gy, const float* w_ohwi, float* gx, void* ws, size_t ws_bytes, int N, int H, int W, int Cin,
                    int Cout, int ks, int stride, int pad, void* stream);
size_t vqw_sconv_wgrad_ws_bytes(int Cin, int Cout, int ks, int N, int H, int W, int stride, int pad);
int vqw_sconv_wgrad(const float* x, const float* gy, float* dw_ohwi, float* dbias, void* ws, size_t ws_bytes, int N, int H,
                    int W, int Cin, int Cout, int ks, int stride, int pad, int accumulate, void* stream);
int vqw_leaky_relu_bwd(const float* y, const float* gy, float* gx, float slope, long n, void* stream);
/* BatchNorm2d(affine) + LeakyReLU: y = lrelu(((x - mean) * rstd) * gamma + beta); statistics through
 * vqw_bn_partial_stats / vqw_bn_finalize.  bwd_reduce: sums[C][2] = {sum g', sum g' * xhat} (dbeta, dgamma);
 * bwd_apply: dx, and dgamma / dbeta (may be NULL) written or accumulated. */
int vqw_bn_affine_fwd(const float* x, const float* mean_rstd, const float* gamma, const float* beta, float* y, long P, int C,
                      float slope, void* stream);
int vqw_bn_affine_bwd_reduce(const float* x, const float* mean_rstd, const float* gamma, const float* beta, const float* gy,
                             double* sums, void* ws, size_t ws_bytes, int N, int HW, int C, float slope, void* stream);
int vqw_bn_affine_bwd_apply(const float* x, const float* mean_rstd, const float* gamma, const float* beta, const float* gy,
                            const double* sums, double count, float* gx, float* dgamma, float* dbeta, long P, int C,
                            float slope, int training, int accumulate, void* stream);
/* mode 0: mean(relu(1 - x)), 1: mean(relu(1 + x)) (hinge_d_loss halves), 2: -mean(x) (generator loss) */
int vqw_hinge_fwd(const float* x, long n, int mode, float* loss, void* stream);
int vqw_hinge_bwd(const float* x, long n, int mode, const float* gloss, float* gx, void* stream);

/* ---- multi-window reconstruction loss (trainers/multi_window_trainer.py:93-109, base.py:290-314):
 * mean((w(a) - w(b))^2) with w(x) = clamp(alpha * x + beta, lo, hi); gradient w.r.t. a (zero where clamped). */
int vqw_window_mse_fwd(const float* a, const float* b, float* loss, void* ws, size_t ws_bytes, long n, float alpha,
                       float beta, float lo, float hi, void* stream);
int vqw_window_mse_bwd(const float* a, const float* b, const float* gloss, float* ga, long n, float alpha, float beta,
                       float lo, float hi, void* stream);

/* ---- deferred split-K folds of the weight gradients (ABI 8).  Every conv weight-gradient entry point ends in one or two
 * short fold launches (dW and dbias slabs -> the gradient).  With vqw_fold_defer(1) those folds are only recorded - the
 * caller must then keep the `ws` buffers of the weight-gradient calls alive - and vqw_fold_flush_host() folds everything
 * recorded so far in ONE launch on `stream` (which must be ordered after the recorded calls' streams).  Two recorded folds
 * into the same output (overwrite, then accumulate: the two views of a training step) are summed in recording order; a
 * third one is an error (flush first).  table_host: pinned host buffer, table_dev: device buffer, both of at least
 * vqw_fold_table_bytes() bytes, both untouched by the caller until the launch has run.  Process-wide state (backward nodes
 * and end-of-pass callbacks run on different host threads).  Replaces nothing upstream: plumbing behind F.conv2d's weight
 * gradient (see the convolution section). */
int vqw_fold_defer(int on);              /* returns the previous setting */
int vqw_fold_pending(void);              /* number of recorded, unflushed fold records */
size_t vqw_fold_table_bytes(void);
int vqw_fold_discard(void);              /* drop the recorded folds (after an aborted backward pass); returns how many */
int vqw_fold_flush_host(void* table_host, void* table_dev, size_t table_bytes, void* stream);

/* ---- optimiser: torch.optim.Adam as built in trainers/base.py:165-175 */
int vqw_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2,
                  void* stream);
/* the same update for many tensors in one launch: chunks_dev points to n_chunks device records
 * {float* p; const float* g; float* m; float* v; int64_t n} (40 bytes each), one workgroup per record */
int vqw_adam_multi(const void* chunks_dev, int n_chunks, float lr, float beta1, float beta2, float eps,
                   float weight_decay, float bias_corr1, float bias_corr2, void* stream);

#ifdef __cplusplus
}
#endif
#endif
