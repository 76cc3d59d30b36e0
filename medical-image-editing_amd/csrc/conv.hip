// C-ABI entry points for convolution: argument validation and kernel selection.
#include "common.h"
#include <stdlib.h>
#include "conv_common.h"
#include "prof.h"
#include "../../include/vqwnet_hip.h"

static int g_conv_backend = 0;  // 0 auto, 1 generic only

// ---------------------------------------------------------------------------------------------
// Optional per-launch timing of the convolution kernels (bench.py roofline): HIP events recorded on the SAME
// stream right around each conv kernel family.  Off by default; nothing is recorded, allocated or synchronised
// unless vqw_profile_begin() was called.  Families: 0 = MFMA fwd/dgrad, 1 = MFMA wgrad (incl. slab reduce),
// 2 = generic fwd, 3 = generic wgrad, 4 = Winograd-form fwd/dgrad/wgrad (FLOPs = the 4/9 the matrix cores execute),
// 5 = HBM-bound normalisation / element-wise kernels (prof.h).
#define PROF_MAX 32768
static bool g_prof_on = false;
static int g_prof_n = 0;
static hipEvent_t* g_prof_ev = nullptr;   // 2 * PROF_MAX events
static double g_prof_flops[PROF_MAX];
static double g_prof_bytes[PROF_MAX];
static int g_prof_family[PROF_MAX];

static const int g_pw_wgrad = []{ const char* e = getenv("VQW_PW_WGRAD"); return e ? atoi(e) : 1; }();      // 0: 1x1 weight gradients on the split-K GEMM kernel (A/B)

extern "C" int vqw_profile_begin(void) {
    if (!g_prof_ev) {
        g_prof_ev = (hipEvent_t*)malloc(sizeof(hipEvent_t) * 2 * PROF_MAX);
        for (int i = 0; i < 2 * PROF_MAX; ++i)
            if (hipEventCreate(&g_prof_ev[i]) != hipSuccess) { vqw_set_error("vqw_profile_begin: hipEventCreate failed"); return VQW_ERR_HIP; }
    }
    g_prof_n = 0;
    g_prof_on = true;
    return VQW_OK;
}
// out[f] = {launches, total ms, total algorithmic flops, total algorithmic bytes} per family.  Synchronises on the
// recorded events.
extern "C" int vqw_profile_end(double* out /*[PROF_FAMILIES][4]*/) {
    g_prof_on = false;
    if (!out) return VQW_OK;
    for (int i = 0; i < PROF_FAMILIES * 4; ++i) out[i] = 0.0;
    for (int i = 0; i < g_prof_n; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(g_prof_ev[2 * i + 1]) != hipSuccess || hipEventElapsedTime(&ms, g_prof_ev[2 * i], g_prof_ev[2 * i + 1]) != hipSuccess) {
            vqw_set_error("vqw_profile_end: event query failed");
            return VQW_ERR_HIP;
        }
        int f = g_prof_family[i];
        out[4 * f] += 1.0;
        out[4 * f + 1] += (double)ms;
        out[4 * f + 2] += g_prof_flops[i];
        out[4 * f + 3] += g_prof_bytes[i];
    }
    return VQW_OK;
}
static unsigned g_prof_mask = 0xFFFFFFFFu;
extern "C" int vqw_profile_families(unsigned mask) {
    const unsigned old = g_prof_mask;
    g_prof_mask = mask;
    return (int)old;
}
int vqw_prof_open(int family, double flops, double bytes, hipStream_t st) {
    if (!g_prof_on || g_prof_n >= PROF_MAX || !((g_prof_mask >> family) & 1u)) return -1;
    const int idx = g_prof_n++;
    g_prof_family[idx] = family;
    g_prof_flops[idx] = flops;
    g_prof_bytes[idx] = bytes;
    (void)hipEventRecord(g_prof_ev[2 * idx], st);
    return idx;
}
void vqw_prof_close(int idx, hipStream_t st) { (void)hipEventRecord(g_prof_ev[2 * idx + 1], st); }
extern "C" int vqw_set_conv_backend(int mode) {
    int old = g_conv_backend == 1 ? 1 : (g_halo_mode ? 2 : (g_wino_mode ? 3 : 0));
    g_conv_backend = mode == 1 ? 1 : 0;
    g_halo_mode = mode == 2 ? 1 : 0;
    g_wgrad_tile_mode = mode == 2 ? 1 : 0;
    g_dil_mode = mode == 2 ? 1 : 0;
    g_wino_mode = (mode == 2 || mode == 3) ? 1 : 0;
    return old;
}

static int check_conv_args(const char* who, const float* src0, int C0, int up0, const float* src1, int C1, int N, int H, int W,
                           int Cout, int ksize, int dil) {
    VQW_CHECK(src0 && C0 > 0, "%s: src0 must be set with C0 > 0", who);
    VQW_CHECK((C1 == 0) == (src1 == nullptr), "%s: src1/C1 mismatch", who);
    VQW_CHECK(C1 >= 0 && N > 0 && H > 0 && W > 0 && Cout > 0, "%s: bad sizes N=%d H=%d W=%d Cout=%d", who, N, H, W, Cout);
    VQW_CHECK(ksize == 1 || ksize == 3, "%s: ksize must be 1 or 3 (got %d)", who, ksize);
    VQW_CHECK(dil >= 1, "%s: dilation must be >= 1", who);
    VQW_CHECK(!up0 || (H % 2 == 0 && W % 2 == 0), "%s: up-sampled src0 needs even H, W", who);
    VQW_CHECK((long)N * H * W * (long)imax(Cout, C0 + C1) < (1L << 40), "%s: tensor too large", who);
    return VQW_OK;
}

// images per launch such that no tensor of the launch exceeds the 32-bit descriptor range (N when everything fits)
static int conv_batch_group(int N, int H, int W, int Cin, int Cout) {
    const long c = Cin > Cout ? Cin : Cout;
    const long per_image = (long)H * W * c * 4;
    if ((long)N * per_image <= 0xFFFFFFE0L || per_image > 0xFFFFFFE0L) return N;      // fits, or not even one image does
    long g = 0xFFFFFFE0L / per_image;
    return (int)(g < 1 ? 1 : g);
}

extern "C" int vqw_conv2d_fwd(const float* src0, int C0, int up0, const float* src1, int C1, const float* w_ohwi,
                              const float* bias, float* y, int N, int H, int W, int Cout, int ksize, int dil, int relu,
                              void* stream) {
    int rc = check_conv_args("vqw_conv2d_fwd", src0, C0, up0, src1, C1, N, H, W, Cout, ksize, dil);
    if (rc) return rc;
    VQW_CHECK(w_ohwi && y, "vqw_conv2d_fwd: weights and output must be set");
    {   // The MFMA kernels address each tensor through a 32-bit buffer descriptor (4 GiB).  Larger batches (288 GB of
        // HBM invite them) are run as consecutive image groups that fit: images are independent in a convolution.
        const int g = conv_batch_group(N, H, W, C0 + C1, Cout);
        if (g < N) {
            for (int n0 = 0; n0 < N; n0 += g) {
                const int nn = N - n0 < g ? N - n0 : g;
                const size_t px = (size_t)n0 * H * W;
                rc = vqw_conv2d_fwd(src0 + (up0 ? px / 4 : px) * C0, C0, up0, src1 ? src1 + px * C1 : nullptr, C1, w_ohwi, bias,
                                    y + px * Cout, nn, H, W, Cout, ksize, dil, relu, stream);
                if (rc) return rc;
            }
            return VQW_OK;
        }
    }
    ConvIn in{src0, src1, C0, C1, up0};
    hipStream_t st = (hipStream_t)stream;
    const double flops = 2.0 * N * H * W * (double)Cout * ksize * ksize * (C0 + C1);
    const double px = (double)N * H * W;
    const double bytes = 4.0 * (px * C0 / (up0 ? 4 : 1) + px * C1 + px * Cout + (double)Cout * ksize * ksize * (C0 + C1));
    if (g_conv_backend == 0 && conv_stem_ok(in, Cout, ksize)) {
        ProfScope ps(2, flops, st, bytes);
        return conv_stem_fwd(in, w_ohwi, bias, y, N, H, W, Cout, ksize, dil, relu, st);
    }
    if (g_conv_backend == 0 && conv_head_ok(in, Cout, ksize)) {
        ProfScope ps(2, flops, st, bytes);
        return conv_head_fwd(in, w_ohwi, bias, y, (long)N * H * W, relu, st);
    }
    if (g_conv_backend == 0 && ksize == 1 && !relu && C1 == 0 && !up0 && conv_pw_stream_ok(C0, Cout, N, H * W)) {
        ProfScope ps(2, flops, st, bytes);          // streaming 1x1 on the vector lanes (HBM-bound)
        return conv_pw_stream(src0, w_ohwi, bias, y, nullptr, N, H * W, C0, Cout, 0, st);
    }
    if (g_conv_backend == 0 && conv_mfma_fwd_ok(in, Cout, ksize)) {
        ProfScope ps(0, flops, st, bytes);
        if (conv_halo_fwd_ok(in, N, H, W, Cout, ksize, dil)) return conv_halo_fwd(in, w_ohwi, bias, y, N, H, W, Cout, relu, st);
        if (conv_dil_fwd_ok(in, N, H, W, Cout, ksize, dil)) return conv_dil_fwd(in, w_ohwi, bias, y, N, H, W, Cout, dil, relu, st);
        return conv_mfma_fwd(in, w_ohwi, bias, y, N, H, W, Cout, ksize, dil, relu, st);
    }
    ProfScope ps(2, flops, st, bytes);
    return conv_direct_fwd(in, w_ohwi, bias, y, N, H, W, Cout, ksize, dil, relu, st);
}

// y += conv(src0): the input gradient of one of several convolutions of the same tensor, summed in place (autograd would
// write each to a tensor of its own and add them pairwise: three more passes over the tensor per extra consumer).
// Served by the row-chain kernel (dilated 3x3, 32 channels: the atrous pyramid's branches); query ..._supported first.
// 1 x 1 layers on the implicit-GEMM kernel (its epilogue adds to y when asked): not the stem / head shapes, tensors within
// the 32-bit descriptor range
static bool conv_pw_acc_ok(const ConvIn& in, int N, int H, int W, int Cout, int ksize) {
    return ksize == 1 && !conv_stem_ok(in, Cout, ksize) && !conv_head_ok(in, Cout, ksize) && conv_mfma_fwd_ok(in, Cout, ksize) &&
           (long)N * H * W * (long)imax(Cout, in.C0) * 4 <= 0xFFFFFFE0L;
}
extern "C" int vqw_conv2d_fwd_acc_supported(int C0, int N, int H, int W, int Cout, int ksize, int dil) {
    ConvIn in{nullptr, nullptr, C0, 0, 0};
    if (g_conv_backend != 0 || N <= 0 || conv_batch_group(N, H, W, C0, Cout) < N) return 0;
    return (conv_dil_fwd_ok(in, N, H, W, Cout, ksize, dil) || conv_pw_acc_ok(in, N, H, W, Cout, ksize)) ? 1 : 0;
}
extern "C" int vqw_conv2d_fwd_acc(const float* src0, int C0, const float* w_ohwi, float* y, int N, int H, int W, int Cout,
                                  int ksize, int dil, void* stream) {
    VQW_CHECK(src0 && w_ohwi && y, "vqw_conv2d_fwd_acc: source, weights and output must be set");
    VQW_CHECK(vqw_conv2d_fwd_acc_supported(C0, N, H, W, Cout, ksize, dil), "vqw_conv2d_fwd_acc: shape not served (query vqw_conv2d_fwd_acc_supported)");
    ConvIn in{src0, nullptr, C0, 0, 0};
    const double px = (double)N * H * W;
    if (ksize == 1) {
        if (conv_pw_stream_ok(C0, Cout, N, H * W)) {
            ProfScope ps(2, 2.0 * px * Cout * C0, (hipStream_t)stream, 4.0 * (px * C0 + 2.0 * px * Cout + (double)Cout * C0));
            return conv_pw_stream(src0, w_ohwi, nullptr, y, nullptr, N, H * W, C0, Cout, 1, (hipStream_t)stream);
        }
        ProfScope ps(0, 2.0 * px * Cout * C0, (hipStream_t)stream, 4.0 * (px * C0 + 2.0 * px * Cout + (double)Cout * C0));
        return conv_mfma_fwd(in, w_ohwi, nullptr, y, N, H, W, Cout, 1, 1, 2, (hipStream_t)stream);      // relu = 2: y += result
    }
    ProfScope ps(0, 2.0 * px * Cout * 9.0 * C0, (hipStream_t)stream, 4.0 * (px * C0 + 2.0 * px * Cout + 9.0 * Cout * C0));
    return conv_dil_fwd(in, w_ohwi, nullptr, y, N, H, W, Cout, dil, 0, (hipStream_t)stream, nullptr, 1);
}

// Convolution that also leaves the InstanceNorm statistics of its output (per-tile partial sums from the epilogue of the
// halo-tile kernel): the norm that follows (blocks.py:45-49: conv -> InstanceNorm -> ReLU) skips its own reduction pass.
// ..._stats_parts() = partial (sum, M2 about the tile mean) pairs per (image, channel), 0 when the shape is not served.
extern "C" int vqw_conv2d_fwd_stats_parts(int C0, int C1, int up0, int N, int H, int W, int Cout, int ksize, int dil) {
    ConvIn in{nullptr, nullptr, C0, C1, up0};
    if (g_conv_backend != 0 || N <= 0 || conv_stem_ok(in, Cout, ksize) || conv_head_ok(in, Cout, ksize) || !conv_mfma_fwd_ok(in, Cout, ksize)) return 0;
    if (conv_batch_group(N, H, W, C0 + C1, Cout) < N) return 0;
    if (ksize == 1 && C1 == 0 && !up0 && conv_pw_stream_ok(C0, Cout, N, H * W)) return conv_pw_stream_stat_tiles(H * W);
    if (conv_halo_fwd_ok(in, N, H, W, Cout, ksize, dil)) return conv_halo_stat_tiles(in, H, W, Cout);
    if (conv_dil_fwd_ok(in, N, H, W, Cout, ksize, dil)) return conv_dil_stat_tiles(H);
    return conv_mfma_stat_tiles(in, N, H, W, Cout, dil);        // implicit-GEMM kernel: 1x1, dilated, ragged widths
}
extern "C" int vqw_conv2d_fwd_stats(const float* src0, int C0, int up0, const float* src1, int C1, const float* w_ohwi,
                                    const float* bias, float* y, float* part, int N, int H, int W, int Cout, int ksize, int dil,
                                    void* stream) {
    int rc = check_conv_args("vqw_conv2d_fwd_stats", src0, C0, up0, src1, C1, N, H, W, Cout, ksize, dil);
    if (rc) return rc;
    VQW_CHECK(w_ohwi && y && part, "vqw_conv2d_fwd_stats: weights, output and partials must be set");
    VQW_CHECK(vqw_conv2d_fwd_stats_parts(C0, C1, up0, N, H, W, Cout, ksize, dil) > 0,
              "vqw_conv2d_fwd_stats: shape not served (query vqw_conv2d_fwd_stats_parts)");
    ConvIn in{src0, src1, C0, C1, up0};
    hipStream_t st = (hipStream_t)stream;
    const double flops = 2.0 * N * H * W * (double)Cout * ksize * ksize * (C0 + C1);
    const double px = (double)N * H * W;
    const double bytes = 4.0 * (px * C0 / (up0 ? 4 : 1) + px * C1 + px * Cout + (double)Cout * ksize * ksize * (C0 + C1));
    if (ksize == 1 && C1 == 0 && !up0 && conv_pw_stream_ok(C0, Cout, N, H * W)) {
        ProfScope ps2(2, flops, st, bytes);
        return conv_pw_stream(src0, w_ohwi, bias, y, part, N, H * W, C0, Cout, 0, st);
    }
    ProfScope ps(0, flops, st, bytes);
    if (conv_halo_fwd_ok(in, N, H, W, Cout, ksize, dil)) return conv_halo_fwd(in, w_ohwi, bias, y, N, H, W, Cout, 0, st, part);
    if (conv_dil_fwd_ok(in, N, H, W, Cout, ksize, dil)) return conv_dil_fwd(in, w_ohwi, bias, y, N, H, W, Cout, dil, 0, st, part);
    return conv_mfma_fwd(in, w_ohwi, bias, y, N, H, W, Cout, ksize, dil, 0, st, part);
}

extern "C" size_t vqw_conv2d_wgrad_ws_bytes(int C0, int C1, int N, int H, int W, int Cout, int ksize) {
    int Cin = C0 + C1;
    if (N > 0 && H > 0 && W > 0) {          // > 4 GiB tensors run as image groups: size for a group
        const int g = conv_batch_group(N, H, W, Cin, Cout);
        if (g < N) return vqw_conv2d_wgrad_ws_bytes(C0, C1, g, H, W, Cout, ksize);
    }
    long P = (long)N * H * W;
    long nout = (long)Cout * ksize * ksize * Cin;
    size_t direct = (size_t)conv_direct_wgrad_splits(nout, P) * nout;
    size_t mfma = conv_mfma_wgrad_ws_floats(Cin, Cout, ksize, P);
    size_t f = direct > mfma ? direct : mfma;
    size_t thin = conv_stem_wgrad_ws_floats(Cout) + conv_head_wgrad_ws_floats(Cin);
    if (thin > f) f = thin;
    if (ksize == 1 && (long)Cin * Cout <= 4096 && conv_pw_wgrad_ws_floats(Cin, Cout) > f) f = conv_pw_wgrad_ws_floats(Cin, Cout);
    return (f + bias_grad_ws_floats(Cout)) * sizeof(float) + 256;
}

extern "C" int vqw_conv2d_wgrad(const float* src0, int C0, int up0, const float* src1, int C1, const float* dy, float* dw_ohwi,
                                float* dbias, void* ws, size_t ws_bytes, int N, int H, int W, int Cout, int ksize, int dil,
                                int accumulate, void* stream) {
    int rc = check_conv_args("vqw_conv2d_wgrad", src0, C0, up0, src1, C1, N, H, W, Cout, ksize, dil);
    if (rc) return rc;
    VQW_CHECK(dy && dw_ohwi && ws, "vqw_conv2d_wgrad: dy, dw and workspace must be set");
    VQW_CHECK(ws_bytes >= vqw_conv2d_wgrad_ws_bytes(C0, C1, N, H, W, Cout, ksize), "vqw_conv2d_wgrad: workspace too small");
    {   // > 4 GiB tensors: image groups, the later ones accumulating into dW / dbias (see vqw_conv2d_fwd)
        const int g = conv_batch_group(N, H, W, C0 + C1, Cout);
        if (g < N) {
            // (the groups share ONE workspace: their slab folds must run between them, not be deferred to the end of the pass)
            const int was = vqw_fold_defer(0);
            for (int n0 = 0; n0 < N && rc == 0; n0 += g) {
                const int nn = N - n0 < g ? N - n0 : g;
                const size_t px = (size_t)n0 * H * W;
                rc = vqw_conv2d_wgrad(src0 + (up0 ? px / 4 : px) * C0, C0, up0, src1 ? src1 + px * C1 : nullptr, C1, dy + px * Cout,
                                      dw_ohwi, dbias, ws, ws_bytes, nn, H, W, Cout, ksize, dil, (accumulate || n0 > 0) ? 1 : 0, stream);
            }
            vqw_fold_defer(was);
            return rc;
        }
    }
    ConvIn in{src0, src1, C0, C1, up0};
    hipStream_t st = (hipStream_t)stream;
    float* wsf = (float*)ws;
    const double flops = 2.0 * N * H * W * (double)Cout * ksize * ksize * (C0 + C1);
    const double px = (double)N * H * W;
    const double bytes = 4.0 * (px * C0 / (up0 ? 4 : 1) + px * C1 + px * Cout + (double)Cout * ksize * ksize * (C0 + C1));
    if (g_conv_backend == 0 && conv_stem_wgrad_ok(in, Cout, ksize)) {
        ProfScope ps(3, flops, st, bytes);
        return conv_stem_wgrad(in, dy, dw_ohwi, dbias, wsf, N, H, W, Cout, ksize, dil, accumulate, st);
    }
    if (g_conv_backend == 0 && conv_head_ok(in, Cout, ksize)) {
        ProfScope ps(3, flops, st, bytes);
        return conv_head_wgrad(in, dy, dw_ohwi, dbias, wsf, (long)N * H * W, accumulate, st);
    }
    if (g_conv_backend == 0 && g_pw_wgrad && conv_pw_wgrad_ok(in, Cout, ksize, (long)N * H * W)) {
        ProfScope ps(1, flops, st, bytes);
        if (dbias) {
            rc = bias_grad(dy, dbias, wsf, (long)N * H * W, Cout, st, accumulate);
            if (rc) return rc;
        }
        return conv_pw_wgrad(in, dy, dw_ohwi, wsf + bias_grad_ws_floats(Cout), (long)N * H * W, Cout, accumulate, st);
    }
    if (g_conv_backend == 0 && conv_mfma_wgrad_ok(in, Cout, ksize)) {
        const bool wino = conv_mfma_wgrad_is_wino(in, N, H, W, Cout, ksize, dil);
        ProfScope ps(wino ? 4 : 1, wino ? flops * (4.0 / 9.0) : flops, st, bytes);
        int bias_done = 0;
        rc = conv_mfma_wgrad(in, dy, dw_ohwi, dbias, &bias_done, wsf + bias_grad_ws_floats(Cout), N, H, W, Cout, ksize, dil, st,
                             accumulate);
        if (rc) return rc;
        if (dbias && !bias_done) return bias_grad(dy, dbias, wsf, (long)N * H * W, Cout, st, accumulate);
        return VQW_OK;
    }
    if (dbias) {
        rc = bias_grad(dy, dbias, wsf, (long)N * H * W, Cout, st, accumulate);
        if (rc) return rc;
        wsf += bias_grad_ws_floats(Cout);
    }
    ProfScope ps(3, flops, st, bytes);
    return conv_direct_wgrad(in, dy, dw_ohwi, wsf, N, H, W, Cout, ksize, dil, st, accumulate);
}

// ---------------------------------------------------------------------------------------------
// 3x3 conv over a nearest x2 up-sampled input (blocks.py:106,123-126), collapsed onto the low-resolution grid
extern "C" int vqw_conv3x3_up2_supported(int Cin, int Cout, int N, int h, int w) {
    // the collapsed form is chosen in forward and must then also serve the input gradient (roles of Cin / Cout swapped:
    // Cout % 4 == 0 too), so a layer that took it can always be back-propagated
    return g_conv_backend == 0 && conv_up2_ok(Cin, Cout, (long)N * h * w) && conv_up2_ok(Cout, Cin, (long)N * h * w) ? 1 : 0;
}
extern "C" size_t vqw_conv3x3_up2_ws_bytes(int Cin, int Cout) { return conv_up2_ws_floats(Cin, Cout) * sizeof(float) + 256; }
extern "C" int vqw_conv3x3_up2_prepare(const float* w_ohwi, void* ws, size_t ws_bytes, int Cin, int Cout, void* stream) {
    VQW_CHECK(w_ohwi && ws && Cin > 0 && Cout > 0, "vqw_conv3x3_up2_prepare: bad arguments");
    VQW_CHECK(ws_bytes >= vqw_conv3x3_up2_ws_bytes(Cin, Cout), "vqw_conv3x3_up2_prepare: workspace too small");
    return conv_up2_prepare(w_ohwi, (float*)ws, Cin, Cout, (hipStream_t)stream);
}
extern "C" int vqw_conv3x3_up2_fwd(const float* x_low, const void* ws, const float* bias, float* y, int N, int h, int w, int Cin,
                                   int Cout, int relu, void* stream) {
    VQW_CHECK(x_low && ws && y && N > 0 && h > 0 && w > 0, "vqw_conv3x3_up2_fwd: bad arguments");
    VQW_CHECK(conv_up2_ok(Cin, Cout, (long)N * h * w), "vqw_conv3x3_up2_fwd: unsupported shape (query vqw_conv3x3_up2_supported)");
    const bool wino = conv_up2_fwd_is_wino(Cin, Cout, N, h, w);     // nine products per tile | 4 parities x 4 taps on the low-res grid
    const double flops = 2.0 * N * h * w * (wino ? 9.0 : 16.0) * Cout * Cin;
    const double bytes = 4.0 * ((double)N * h * w * Cin + 4.0 * N * h * w * Cout + 16.0 * Cout * Cin);
    ProfScope ps(wino ? 4 : 0, flops, (hipStream_t)stream, bytes);
    return conv_up2_fwd(x_low, (const float*)ws, bias, y, N, h, w, Cin, Cout, relu, (hipStream_t)stream);
}
extern "C" int vqw_conv3x3_up2_fwd_stats_parts(int Cin, int Cout, int N, int h, int w) {
    if (g_conv_backend != 0 || !conv_up2_ok(Cin, Cout, (long)N * h * w)) return 0;
    return conv_up2_stat_tiles(Cin, Cout, N, h, w);
}
extern "C" int vqw_conv3x3_up2_fwd_stats(const float* x_low, const void* ws, const float* bias, float* y, float* part, int N, int h,
                                         int w, int Cin, int Cout, void* stream) {
    VQW_CHECK(x_low && ws && y && part && N > 0 && h > 0 && w > 0, "vqw_conv3x3_up2_fwd_stats: bad arguments");
    VQW_CHECK(vqw_conv3x3_up2_fwd_stats_parts(Cin, Cout, N, h, w) > 0, "vqw_conv3x3_up2_fwd_stats: shape not served");
    const bool wino = conv_up2_fwd_is_wino(Cin, Cout, N, h, w);
    const double flops = 2.0 * N * h * w * (wino ? 9.0 : 16.0) * Cout * Cin;
    const double bytes = 4.0 * ((double)N * h * w * Cin + 4.0 * N * h * w * Cout + 16.0 * Cout * Cin);
    ProfScope ps(wino ? 4 : 0, flops, (hipStream_t)stream, bytes);
    return conv_up2_fwd(x_low, (const float*)ws, bias, y, N, h, w, Cin, Cout, 0, (hipStream_t)stream, part);
}
// Two 32-cout layers of the SAME up-sampled input (StyledResUpBlock's shortcut `conv` and `conv1`, blocks.py:100-112) as ONE
// 64-cout launch of the nine-product kernel: ws = vqw_conv3x3_up2_prepare of the concatenated weights [w_a | w_b] (Cout = 64),
// bias_cat = [b_a | b_b] or NULL; y_a / y_b and their statistics partials come out as two 32-channel tensors.  Returns the
// partials per image from ..._supported (0: not served - run the layers one by one).
extern "C" int vqw_conv3x3_up2_fwd_pair_supported(int Cin, int Cout_each, int N, int h, int w) {
    if (g_conv_backend != 0 || Cout_each != 32 || !conv_up2_ok(Cin, 64, (long)N * h * w) || !conv_wino_up_fwd_ok(Cin, 64, N, h, w)) return 0;
    return conv_wino_up_stat_tiles(h, w);
}
extern "C" int vqw_conv3x3_up2_fwd_pair(const float* x_low, const void* ws, const float* bias_cat, float* y_a, float* y_b, float* part_a,
                                        float* part_b, int N, int h, int w, int Cin, int Cout_each, void* stream) {
    VQW_CHECK(x_low && ws && y_a && y_b && part_a && part_b && N > 0 && h > 0 && w > 0, "vqw_conv3x3_up2_fwd_pair: bad arguments");
    VQW_CHECK(vqw_conv3x3_up2_fwd_pair_supported(Cin, Cout_each, N, h, w) > 0, "vqw_conv3x3_up2_fwd_pair: shape not served");
    const double flops = 2.0 * N * h * w * 9.0 * 64.0 * Cin;
    const double bytes = 4.0 * ((double)N * h * w * Cin + 4.0 * N * h * w * 64.0 + 16.0 * 64.0 * Cin);
    ProfScope ps(4, flops, (hipStream_t)stream, bytes);
    return conv_wino_up_fwd(x_low, (const float*)ws + 32L * 64 * Cin, bias_cat, y_a, N, h, w, Cin, 64, 0, (hipStream_t)stream, part_a, y_b, part_b);
}
extern "C" int vqw_conv3x3_up2_dgrad(const float* dy, const void* ws, float* dx_low, int N, int h, int w, int Cin, int Cout,
                                     void* stream) {
    VQW_CHECK(dy && ws && dx_low && N > 0 && h > 0 && w > 0, "vqw_conv3x3_up2_dgrad: bad arguments");
    VQW_CHECK(conv_up2_ok(Cout, Cin, (long)N * h * w), "vqw_conv3x3_up2_dgrad: unsupported shape");
    const bool wino = conv_up2_dgrad_is_wino(Cin, Cout, N, h, w);          // nine products per tile instead of sixteen taps
    const double flops = 2.0 * N * h * w * (wino ? 9.0 : 16.0) * Cout * Cin;
    const double bytes = 4.0 * ((double)N * h * w * Cin + 4.0 * N * h * w * Cout + 16.0 * Cout * Cin);
    ProfScope ps(wino ? 4 : 0, flops, (hipStream_t)stream, bytes);
    return conv_up2_dgrad(dy, (const float*)ws, dx_low, N, h, w, Cin, Cout, (hipStream_t)stream);
}

extern "C" int vqw_conv3x3_up2_dgrad_acc_supported(int Cin, int Cout, int N, int h, int w) {
    return (conv_up2_ok(Cout, Cin, (long)N * h * w) && conv_up2_dgrad_is_wino(Cin, Cout, N, h, w)) ? 1 : 0;
}
extern "C" int vqw_conv3x3_up2_dgrad_acc(const float* dy, const void* ws, float* dx_low, int N, int h, int w, int Cin, int Cout,
                                         void* stream) {
    VQW_CHECK(dy && ws && dx_low && N > 0 && h > 0 && w > 0, "vqw_conv3x3_up2_dgrad_acc: bad arguments");
    VQW_CHECK(vqw_conv3x3_up2_dgrad_acc_supported(Cin, Cout, N, h, w), "vqw_conv3x3_up2_dgrad_acc: unsupported shape");
    const double flops = 2.0 * N * h * w * 9.0 * Cout * Cin;
    const double bytes = 4.0 * (2.0 * N * h * w * Cin + 4.0 * N * h * w * Cout + 16.0 * Cout * Cin);
    ProfScope ps(4, flops, (hipStream_t)stream, bytes);
    return conv_up2_dgrad(dy, (const float*)ws, dx_low, N, h, w, Cin, Cout, (hipStream_t)stream, 1);
}

// ---------------------------------------------------------------------------------------------
// plain 3x3 convolution in Winograd F(2x2, 3x3) form (conv_wino.hip)
extern "C" int vqw_conv3x3_wino_supported(int Cin, int Cout, int N, int H, int W) {
    return g_conv_backend == 0 && conv_wino_ok(Cin, Cout, N, H, W) ? 1 : 0;
}
extern "C" size_t vqw_conv3x3_wino_ws_bytes(int Cin, int Cout) { return conv_wino_ws_floats(Cin, Cout) * sizeof(float) + 256; }
extern "C" int vqw_conv3x3_wino_prepare(const float* w_ohwi, void* ws, size_t ws_bytes, int Cin, int Cout, void* stream) {
    VQW_CHECK(w_ohwi && ws && Cin > 0 && Cout > 0, "vqw_conv3x3_wino_prepare: bad arguments");
    VQW_CHECK(ws_bytes >= vqw_conv3x3_wino_ws_bytes(Cin, Cout), "vqw_conv3x3_wino_prepare: workspace too small");
    return conv_wino_prepare(w_ohwi, (float*)ws, Cin, Cout, (hipStream_t)stream);
}
extern "C" int vqw_conv3x3_wino_prepare_dgrad(const float* w_ohwi, void* ws, size_t ws_bytes, int Cin, int Cout, void* stream) {
    VQW_CHECK(w_ohwi && ws && Cin > 0 && Cout > 0, "vqw_conv3x3_wino_prepare_dgrad: bad arguments");
    VQW_CHECK(ws_bytes >= vqw_conv3x3_wino_ws_bytes(Cin, Cout), "vqw_conv3x3_wino_prepare_dgrad: workspace too small");
    return conv_wino_prepare(w_ohwi, (float*)ws, Cin, Cout, (hipStream_t)stream, 1);
}
extern "C" int vqw_conv3x3_wino_fwd(const float* x, const void* ws, const float* bias, float* y, int N, int H, int W, int Cin,
                                    int Cout, int relu, void* stream) {
    VQW_CHECK(x && ws && y && N > 0 && H > 0 && W > 0, "vqw_conv3x3_wino_fwd: bad arguments");
    VQW_CHECK(conv_wino_ok(Cin, Cout, N, H, W), "vqw_conv3x3_wino_fwd: unsupported shape (query vqw_conv3x3_wino_supported)");
    const double px = (double)N * H * W;
    ProfScope ps(4, 2.0 * px * 4.0 * Cout * Cin, (hipStream_t)stream, 4.0 * (px * Cin + px * Cout + 16.0 * Cout * Cin));
    return conv_wino_fwd(x, (const float*)ws, bias, y, N, H, W, Cin, Cout, relu, (hipStream_t)stream);
}
extern "C" int vqw_conv3x3_wino_masked_supported(int Cin, int Cout, int N, int H, int W) {
    return (g_conv_backend == 0 && conv_wino_ok(Cin, Cout, N, H, W) && conv_wino64_ok(Cin, Cout, W)) ? 1 : 0;
}
extern "C" int vqw_conv3x3_wino_fwd_masked(const float* x, const void* ws, const float* mask, float* y, int N, int H, int W, int Cin,
                                           int Cout, void* stream) {
    VQW_CHECK(x && ws && mask && y && N > 0 && H > 0 && W > 0, "vqw_conv3x3_wino_fwd_masked: bad arguments");
    VQW_CHECK(vqw_conv3x3_wino_masked_supported(Cin, Cout, N, H, W), "vqw_conv3x3_wino_fwd_masked: unsupported shape (query vqw_conv3x3_wino_masked_supported)");
    const double px = (double)N * H * W;
    ProfScope ps(4, 2.0 * px * 4.0 * Cout * Cin, (hipStream_t)stream, 4.0 * (px * Cin + 2.0 * px * Cout + 16.0 * Cout * Cin));
    return conv_wino_fwd(x, (const float*)ws, nullptr, y, N, H, W, Cin, Cout, 0, (hipStream_t)stream, nullptr, mask);
}
extern "C" int vqw_conv3x3_wino_fwd_acc(const float* x, const void* ws, float* y, int N, int H, int W, int Cin, int Cout, void* stream) {
    VQW_CHECK(x && ws && y && N > 0 && H > 0 && W > 0, "vqw_conv3x3_wino_fwd_acc: bad arguments");
    VQW_CHECK(vqw_conv3x3_wino_masked_supported(Cin, Cout, N, H, W), "vqw_conv3x3_wino_fwd_acc: unsupported shape (query vqw_conv3x3_wino_masked_supported)");
    const double px = (double)N * H * W;
    ProfScope ps(4, 2.0 * px * 4.0 * Cout * Cin, (hipStream_t)stream, 4.0 * (px * Cin + 2.0 * px * Cout + 16.0 * Cout * Cin));
    return conv_wino_fwd(x, (const float*)ws, nullptr, y, N, H, W, Cin, Cout, 0, (hipStream_t)stream, nullptr, nullptr, 1);
}
// A 3x3 layer of DILATION 2 in Winograd form: the plain kernel on the four phase images of the tensors (rows / columns of one
// parity: pixel pitch 2, row pitch 2 W), same transformed weights as the plain layer.  part: optional statistics partials
// [N][parts][Cout][2] (forward, needs relu == 0); accumulate: y += result (a member of a gradient group).
static const int g_wino_dil2_fwd = []{ const char* e = getenv("VQW_WINOGRAD_DIL2"); return e ? atoi(e) : 1; }();
extern "C" int vqw_conv3x3_wino_dil2_supported(int Cin, int Cout, int N, int H, int W) {
    if (!g_wino_dil2_fwd || g_conv_backend != 0 || N < 1 || H < 2 || W < 2) return 0;
    if (!conv_wino_ok(Cin, Cout, N, H / 2, W / 2) || !conv_wino64_dil2_ok(Cin, Cout, H, W)) return 0;
    return (long)N * H * W * (Cin > Cout ? Cin : Cout) * 4 <= 0xFFFFFFE0L ? 1 : 0;
}
extern "C" int vqw_conv3x3_wino_dil2_stats_parts(int Cin, int Cout, int N, int H, int W) {
    if (!vqw_conv3x3_wino_dil2_supported(Cin, Cout, N, H, W)) return 0;
    return 4 * conv_wino64_stat_tiles(Cin, Cout, H / 2, W / 2);
}
extern "C" int vqw_conv3x3_wino_dil2_fwd(const float* x, const void* ws, const float* bias, float* y, float* part, int accumulate,
                                         int N, int H, int W, int Cin, int Cout, int relu, void* stream) {
    VQW_CHECK(x && ws && y && N > 0 && H > 0 && W > 0, "vqw_conv3x3_wino_dil2_fwd: bad arguments");
    VQW_CHECK(vqw_conv3x3_wino_dil2_supported(Cin, Cout, N, H, W), "vqw_conv3x3_wino_dil2_fwd: unsupported shape (query vqw_conv3x3_wino_dil2_supported)");
    VQW_CHECK(!part || (!relu && !accumulate && vqw_conv3x3_wino_dil2_stats_parts(Cin, Cout, N, H, W) > 0),
              "vqw_conv3x3_wino_dil2_fwd: statistics partials need relu == 0, accumulate == 0 and a served height");
    VQW_CHECK(!accumulate || (!bias && !relu), "vqw_conv3x3_wino_dil2_fwd: the accumulating form takes no bias / ReLU");
    const double px = (double)N * H * W;
    ProfScope ps(4, 2.0 * px * 4.0 * Cout * Cin, (hipStream_t)stream, 4.0 * (px * Cin + (accumulate ? 2.0 : 1.0) * px * Cout + 16.0 * Cout * Cin));
    return conv_wino64_fwd(x, (const float*)ws, bias, y, N, H, W, Cin, Cout, relu, (hipStream_t)stream, part, nullptr, accumulate, nullptr, 0, 2);
}
extern "C" int vqw_conv3x3_wino_split_supported(int Cin, int Cout, int split, int pool0, int N, int H, int W) {
    return (g_conv_backend == 0 && conv_wino_ok(Cin, Cout, N, H, W) && conv_wino64_split_ok(Cin, Cout, split, pool0, N, H, W)) ? 1 : 0;
}
extern "C" int vqw_conv3x3_wino_fwd_split(const float* x, const void* ws, const float* bias, float* y0, float* y1, int N, int H, int W,
                                          int Cin, int Cout, int split, int pool0, int relu, void* stream) {
    VQW_CHECK(x && ws && y0 && y1 && N > 0 && H > 0 && W > 0, "vqw_conv3x3_wino_fwd_split: bad arguments");
    VQW_CHECK(vqw_conv3x3_wino_split_supported(Cin, Cout, split, pool0, N, H, W), "vqw_conv3x3_wino_fwd_split: unsupported shape (query vqw_conv3x3_wino_split_supported)");
    const double px = (double)N * H * W;
    ProfScope ps(4, 2.0 * px * 4.0 * Cout * Cin, (hipStream_t)stream,
                 4.0 * (px * Cin + (pool0 ? 0.25 : 1.0) * px * split + px * (Cout - split) + 16.0 * Cout * Cin));
    return conv_wino64_fwd_split(x, (const float*)ws, bias, y0, y1, N, H, W, Cin, Cout, split, pool0, relu, (hipStream_t)stream);
}
// The same on a layer WIDENED to the kernel's cout tile: ws holds Cout transformed couts of which only split + c1 are real (the
// others are zero weights); y1 has c1 channels.  Serves a two-source layer whose channel total (48 at the encoder's full-resolution
// level) is not a multiple of the tile.
extern "C" int vqw_conv3x3_wino_split_padded_supported(int Cin, int Cout, int split, int c1, int pool0, int N, int H, int W) {
    return (g_conv_backend == 0 && conv_wino_ok(Cin, Cout, N, H, W) && conv_wino64_split_ok(Cin, Cout, split, pool0, N, H, W, c1)) ? 1 : 0;
}
extern "C" int vqw_conv3x3_wino_fwd_split_padded(const float* x, const void* ws, const float* bias, float* y0, float* y1, int N, int H,
                                                 int W, int Cin, int Cout, int split, int c1, int pool0, int relu, void* stream) {
    VQW_CHECK(x && ws && y0 && y1 && N > 0 && H > 0 && W > 0, "vqw_conv3x3_wino_fwd_split_padded: bad arguments");
    VQW_CHECK(vqw_conv3x3_wino_split_padded_supported(Cin, Cout, split, c1, pool0, N, H, W),
              "vqw_conv3x3_wino_fwd_split_padded: unsupported shape (query vqw_conv3x3_wino_split_padded_supported)");
    const double px = (double)N * H * W;
    ProfScope ps(4, 2.0 * px * 4.0 * Cout * Cin, (hipStream_t)stream,
                 4.0 * (px * Cin + (pool0 ? 0.25 : 1.0) * px * split + px * c1 + 16.0 * Cout * Cin));
    return conv_wino64_fwd_split(x, (const float*)ws, bias, y0, y1, N, H, W, Cin, Cout, split, pool0, relu, (hipStream_t)stream, c1);
}
extern "C" int vqw_conv3x3_wino_fwd_inbwd_parts(int Cin, int Cout, int N, int H, int W) {
    if (!vqw_conv3x3_wino_masked_supported(Cin, Cout, N, H, W)) return 0;
    return conv_wino64_stat_tiles(Cin, Cout, H, W);
}
extern "C" int vqw_conv3x3_wino_fwd_inbwd(const float* x, const void* ws, const float* norm_x, const float* norm_mean_rstd, int norm_relu,
                                          float* y, float* part, int N, int H, int W, int Cin, int Cout, void* stream) {
    VQW_CHECK(x && ws && norm_x && norm_mean_rstd && y && part && N > 0 && H > 0 && W > 0, "vqw_conv3x3_wino_fwd_inbwd: bad arguments");
    VQW_CHECK(vqw_conv3x3_wino_fwd_inbwd_parts(Cin, Cout, N, H, W) > 0, "vqw_conv3x3_wino_fwd_inbwd: shape not served");
    const double px = (double)N * H * W;
    ProfScope ps(4, 2.0 * px * 4.0 * Cout * Cin, (hipStream_t)stream, 4.0 * (px * Cin + 2.0 * px * Cout + 16.0 * Cout * Cin));
    return conv_wino_fwd(x, (const float*)ws, nullptr, y, N, H, W, Cin, Cout, 0, (hipStream_t)stream, part, norm_x, 0, norm_mean_rstd, norm_relu);
}
extern "C" int vqw_conv3x3_wino_fwd_stats_parts(int Cin, int Cout, int N, int H, int W) {
    if (g_conv_backend != 0 || !conv_wino_ok(Cin, Cout, N, H, W)) return 0;
    return conv_wino_stat_tiles(Cin, Cout, H, W);
}
extern "C" int vqw_conv3x3_wino_fwd_stats(const float* x, const void* ws, const float* bias, float* y, float* part, int N, int H,
                                          int W, int Cin, int Cout, void* stream) {
    VQW_CHECK(x && ws && y && part && N > 0 && H > 0 && W > 0, "vqw_conv3x3_wino_fwd_stats: bad arguments");
    VQW_CHECK(vqw_conv3x3_wino_fwd_stats_parts(Cin, Cout, N, H, W) > 0, "vqw_conv3x3_wino_fwd_stats: shape not served");
    const double px = (double)N * H * W;
    ProfScope ps(4, 2.0 * px * 4.0 * Cout * Cin, (hipStream_t)stream, 4.0 * (px * Cin + px * Cout + 16.0 * Cout * Cin));
    return conv_wino_fwd(x, (const float*)ws, bias, y, N, H, W, Cin, Cout, 0, (hipStream_t)stream, part);
}

extern "C" int vqw_conv3x3_up2_wgrad_supported(int Cin, int Cout, int N, int h, int w) {
    return g_conv_backend == 0 && conv_up2_wgrad_ok(Cin, Cout, N, h, w) ? 1 : 0;
}
extern "C" size_t vqw_conv3x3_up2_wgrad_ws_bytes(int Cin, int Cout, int N, int h, int w) {
    return conv_up2_wgrad_ws_floats(Cin, Cout, N, h, w) * sizeof(float) + 256;
}
extern "C" int vqw_conv3x3_up2_wgrad(const float* x_low, const float* dy, float* dw_ohwi, float* dbias, void* ws, size_t ws_bytes,
                                     int N, int h, int w, int Cin, int Cout, int accumulate, void* stream) {
    VQW_CHECK(x_low && dy && dw_ohwi && ws && N > 0 && h > 0 && w > 0, "vqw_conv3x3_up2_wgrad: bad arguments");
    VQW_CHECK(conv_up2_wgrad_ok(Cin, Cout, N, h, w), "vqw_conv3x3_up2_wgrad: unsupported shape (query ..._wgrad_supported)");
    VQW_CHECK(ws_bytes >= vqw_conv3x3_up2_wgrad_ws_bytes(Cin, Cout, N, h, w), "vqw_conv3x3_up2_wgrad: workspace too small");
    const bool wino = conv_up2_wgrad_is_wino(Cin, Cout, N, h, w);
    const double flops = 2.0 * N * h * w * (wino ? 9.0 : 16.0) * Cout * Cin;
    const double bytes = 4.0 * ((double)N * h * w * Cin + 4.0 * N * h * w * Cout + 9.0 * Cout * Cin);
    ProfScope ps(wino ? 4 : 1, flops, (hipStream_t)stream, bytes);
    return conv_up2_wgrad(x_low, dy, dw_ohwi, dbias, (float*)ws, N, h, w, Cin, Cout, accumulate, (hipStream_t)stream);
}
