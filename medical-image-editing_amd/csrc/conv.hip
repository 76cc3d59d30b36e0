// C-ABI entry points for convolution: argument validation and kernel selection.
#include "common.h"
#include "conv_common.h"
#include "../../include/vqwnet_hip.h"

static int g_conv_backend = 0;  // 0 auto, 1 generic only
extern "C" int vqw_set_conv_backend(int mode) {
    int old = g_conv_backend;
    g_conv_backend = mode;
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

extern "C" int vqw_conv2d_fwd(const float* src0, int C0, int up0, const float* src1, int C1, const float* w_ohwi,
                              const float* bias, float* y, int N, int H, int W, int Cout, int ksize, int dil, int relu,
                              void* stream) {
    int rc = check_conv_args("vqw_conv2d_fwd", src0, C0, up0, src1, C1, N, H, W, Cout, ksize, dil);
    if (rc) return rc;
    VQW_CHECK(w_ohwi && y, "vqw_conv2d_fwd: weights and output must be set");
    ConvIn in{src0, src1, C0, C1, up0};
    hipStream_t st = (hipStream_t)stream;
    if (g_conv_backend == 0 && conv_mfma_fwd_ok(in, Cout, ksize))
        return conv_mfma_fwd(in, w_ohwi, bias, y, N, H, W, Cout, ksize, dil, relu, st);
    return conv_direct_fwd(in, w_ohwi, bias, y, N, H, W, Cout, ksize, dil, relu, st);
}

extern "C" size_t vqw_conv2d_wgrad_ws_bytes(int C0, int C1, int N, int H, int W, int Cout, int ksize) {
    int Cin = C0 + C1;
    long P = (long)N * H * W;
    long nout = (long)Cout * ksize * ksize * Cin;
    size_t direct = (size_t)conv_direct_wgrad_splits(nout, P) * nout;
    size_t mfma = conv_mfma_wgrad_ws_floats(Cin, Cout, ksize, P);
    size_t f = direct > mfma ? direct : mfma;
    return (f + bias_grad_ws_floats(Cout)) * sizeof(float) + 256;
}

extern "C" int vqw_conv2d_wgrad(const float* src0, int C0, int up0, const float* src1, int C1, const float* dy, float* dw_ohwi,
                                float* dbias, void* ws, size_t ws_bytes, int N, int H, int W, int Cout, int ksize, int dil,
                                void* stream) {
    int rc = check_conv_args("vqw_conv2d_wgrad", src0, C0, up0, src1, C1, N, H, W, Cout, ksize, dil);
    if (rc) return rc;
    VQW_CHECK(dy && dw_ohwi && ws, "vqw_conv2d_wgrad: dy, dw and workspace must be set");
    VQW_CHECK(ws_bytes >= vqw_conv2d_wgrad_ws_bytes(C0, C1, N, H, W, Cout, ksize), "vqw_conv2d_wgrad: workspace too small");
    ConvIn in{src0, src1, C0, C1, up0};
    hipStream_t st = (hipStream_t)stream;
    float* wsf = (float*)ws;
    if (dbias) {
        rc = bias_grad(dy, dbias, wsf, (long)N * H * W, Cout, st);
        if (rc) return rc;
        wsf += bias_grad_ws_floats(Cout);
    }
    if (g_conv_backend == 0 && conv_mfma_wgrad_ok(in, Cout, ksize))
        return conv_mfma_wgrad(in, dy, dw_ohwi, wsf, N, H, W, Cout, ksize, dil, st);
    return conv_direct_wgrad(in, dy, dw_ohwi, wsf, N, H, W, Cout, ksize, dil, st);
}
