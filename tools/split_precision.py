#!/usr/bin/env python3
"""Split-precision EXPERIMENT (never the product path, never the headline): what would the first training step's gradients look
like if the decoder's 3x3 convolutions with >= 128 input channels multiplied bf16 pieces of their fp32 operands on the bf16
matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulation) instead of fp32 on the fp32 ones?

The arithmetic is emulated exactly up to summation order: an fp32 operand t is split into bf16 pieces t0 + t1 + t2 (ti =
bf16(t - t0 - .. - t(i-1))); a product of two bf16 numbers is exact in fp32, so conv(xi, wj) in fp32 IS what the bf16 MFMA
would accumulate.  Schemes:
    f32        the product path as shipped (hand-written fp32 kernels)
    f32_torch  control: the emulation's plumbing with unsplit fp32 operands
    bf16x3_6   3-way split, the six products whose pieces' ranks sum to <= 2 (x0w0, x0w1, x1w0, x0w2, x1w1, x2w0): "fp32-class"
    bf16x3_3   three products (x0w0, x0w1, x1w0): 16 mantissa bits
    bf16x2_4   2-way split, all four products
    bf16       one product of the rounded operands
    f4_grads   (round 4, review item 8) NOT a split: the input and weight gradients of the selected layers in Winograd F(4x4, 3x3)
               form in fp32 (transforms, per-position accumulation over channels / tiles and inverse transform in fp32, weights
               transformed in double - tools/experiments/winograd_f4_accuracy.py has the per-kernel errors); forward as f32_torch;
               layers whose maps are not multiples of 4 keep the control's gradients.  f2_grads: the same plumbing with F(2x2, 3x3)
Forward, input gradient and weight gradient of the selected layers all use the scheme (the gradient w.r.t. the output is
split like an activation).  The emulation runs on torch's own fp32 convolutions - it is a measuring device in tools/, not a
code path of the package.  Output: per scheme the fp64 gradient gate of tests/helpers.py::check_grads_vs_fp64 (error of every
parameter's gradient against the reference's fp64 gradient, in units of the reference's own fp32 spread), the loss and the ids.
    python tools/split_precision.py [fixture.npz ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.nn.functional as F

SCHEMES = {
    "bf16x3_6": (3, [(2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)]),     # smallest terms first
    "bf16x3_3": (3, [(1, 0), (0, 1), (0, 0)]),
    "bf16x2_4": (2, [(1, 1), (1, 0), (0, 1), (0, 0)]),
    "bf16": (1, [(0, 0)]),
    "f32_torch": (0, [(0, 0)]),        # control: the same plumbing with unsplit fp32 operands (torch's fp32 convolution)
}
MIN_CIN = 128

# Winograd emulation (Lavin & Gray; F(4x4, 3x3) with the points 0, +-1, +-2, inf)
_W = {
    4: (torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                      [0, 4, 0, -5, 0, 1]], dtype=torch.float64),
        torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                      [0, 0, 1]], dtype=torch.float64),
        torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)),
    2: (torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64),
        torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64),
        torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)),
}


def wino_conv(x, w, m):
    """conv2d(x, w, padding=1) through F(m x m, 3x3): fp32 everywhere except the weight transform (double, rounded once)."""
    BT, G, AT = (t.to(x.device) for t in _W[m])
    N, C, H, Wd = x.shape
    K, a = w.shape[0], m + 2
    U = torch.einsum("ij,kcjl,ml->imkc", G, w.double(), G).float()
    tiles = F.pad(x, (1, 1, 1, 1)).unfold(2, a, m).unfold(3, a, m)            # (N, C, th, tw, a, a)
    V = torch.einsum("ij,nctwjl,ml->nctwim", BT.float(), tiles, BT.float())
    M = torch.einsum("imkc,nctwim->nktwim", U, V)
    Y = torch.einsum("pi,nktwim,qm->nktwpq", AT.float(), M, AT.float())
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, K, H, Wd)


def wino_wgrad(x, gy, m):
    """d(conv2d(x, w, padding=1)) / dw in Winograd form: dU = sum over tiles of (A dY A^T) (B^T d B), dW = G^T dU G, fp32."""
    BT, G, AT = (t.to(x.device).float() for t in _W[m])
    a = m + 2
    tiles = F.pad(x, (1, 1, 1, 1)).unfold(2, a, m).unfold(3, a, m)            # (N, C, th, tw, a, a)
    V = torch.einsum("ij,nctwjl,ml->nctwim", BT, tiles, BT)
    gyt = gy.unfold(2, m, m).unfold(3, m, m)                                    # (N, K, th, tw, m, m)
    dM = torch.einsum("pi,nktwpq,qm->nktwim", AT, gyt, AT)
    dU = torch.einsum("nktwim,nctwim->imkc", dM, V)
    return torch.einsum("ir,imkc,ms->kcrs", G, dU, G)


def split(t, ways):
    parts, r = [], t.float()
    if ways == 0:
        return [r]
    for _ in range(ways):
        p = r.to(torch.bfloat16).float()
        parts.append(p)
        r = r - p
    return parts


class SplitConv(torch.autograd.Function):
    wino_layers = 0

    @staticmethod
    def forward(ctx, x, w, scheme, dilation):
        ways, pairs = SCHEMES["f32_torch" if scheme in ("f4_grads", "f2_grads") else scheme]
        xs, ws = split(x, ways), split(w, ways)
        pad = dilation * (w.shape[2] // 2)
        y = None
        for i, j in pairs:
            t = F.conv2d(xs[i], ws[j], None, 1, pad, dilation)
            y = t if y is None else y + t
        ctx.save_for_backward(x, w)
        ctx.scheme, ctx.dilation, ctx.pad = scheme, dilation, pad
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        if ctx.scheme in ("f4_grads", "f2_grads"):
            m = 4 if ctx.scheme == "f4_grads" else 2
            if ctx.dilation == 1 and x.shape[2] % m == 0 and x.shape[3] % m == 0:
                SplitConv.wino_layers += 1
                gx = wino_conv(gy.float(), w.flip(2, 3).transpose(0, 1).contiguous(), m)
                return gx, wino_wgrad(x.float(), gy.float(), m), None, None
            gx = torch.nn.grad.conv2d_input(x.shape, w, gy, 1, ctx.pad, ctx.dilation)
            return gx, torch.nn.grad.conv2d_weight(x, w.shape, gy, 1, ctx.pad, ctx.dilation), None, None
        ways, pairs = SCHEMES[ctx.scheme]
        gs, xs, ws = split(gy, ways), split(x, ways), split(w, ways)
        gx = gw = None
        for i, j in pairs:
            t = torch.nn.grad.conv2d_input(x.shape, ws[j], gs[i], 1, ctx.pad, ctx.dilation)
            gx = t if gx is None else gx + t
            t = torch.nn.grad.conv2d_weight(xs[i], w.shape, gs[j], 1, ctx.pad, ctx.dilation)
            gw = t if gw is None else gw + t
        return gx, gw, None, None


class Patch:
    """Routes ops.conv2d / ops.conv2d_cat calls made inside the decoder's forward with >= MIN_CIN input channels and a 3x3
    kernel through SplitConv."""

    def __init__(self, ops, decoder, scheme):
        self.ops, self.dec, self.scheme = ops, decoder, scheme
        self.inside = 0
        self.hits = 0

    def __enter__(self):
        ops = self.ops
        self.c2, self.cc = ops.conv2d, ops.conv2d_cat
        self.h1 = self.dec.register_forward_pre_hook(lambda m, a: self._enter())
        self.h2 = self.dec.register_forward_hook(lambda m, a, o: self._exit())
        me = self

        def conv2d(x, weight, bias=None, dilation=1, up2x=False, skip=None, relu=False, want_stats=False, grad_group=None, **kw):
            cin = weight.shape[1]
            # (members of a gradient group - the ASPP branches, which sum their input gradients in place - stay on the product path;
            # so do the layer pairs of round 4, ops.conv2d_pair / conv2d_up_pair, which do not come through here)
            if not (me.inside and cin >= MIN_CIN and weight.shape[2] == 3 and grad_group is None):
                return me.c2(x, weight, bias, dilation, up2x=up2x, skip=skip, relu=relu, want_stats=want_stats, grad_group=grad_group, **kw)
            me.hits += 1
            xin = F.interpolate(x, scale_factor=2, mode="nearest") if up2x else x
            if skip is not None:
                xin = torch.cat([xin, skip], 1)
            y = SplitConv.apply(xin, weight, me.scheme, int(dilation))
            if bias is not None:
                y = y + bias.view(1, -1, 1, 1)
            if relu:
                y = torch.relu(y)
            y = y.contiguous(memory_format=torch.channels_last)
            return (y, None) if want_stats else y

        def conv2d_cat(x, wa, ba, wb, bb, **kw):
            if not (me.inside and wa.shape[1] >= MIN_CIN and wa.shape[2] == 3):
                return me.cc(x, wa, ba, wb, bb, **kw)
            me.hits += 2
            ya = SplitConv.apply(x, wa, me.scheme, 1) + ba.view(1, -1, 1, 1)
            yb = SplitConv.apply(x, wb, me.scheme, 1) + bb.view(1, -1, 1, 1)
            return torch.cat([ya, yb], 1).contiguous(memory_format=torch.channels_last)

        ops.conv2d, ops.conv2d_cat = conv2d, conv2d_cat
        return self

    def _enter(self):
        self.inside += 1

    def _exit(self):
        self.inside -= 1

    def __exit__(self, *exc):
        self.ops.conv2d, self.ops.conv2d_cat = self.c2, self.cc
        self.h1.remove()
        self.h2.remove()


def run(name, scheme):
    import test_gpu_parity as T
    from conftest import load_golden
    from helpers import check_grads_vs_fp64
    from hipops import ops
    g = load_golden(name)
    tr, cfg = T._hip_trainer(g)
    tr.encoder.train(); tr.decoder.train()
    img, noise = g.t("step0/image", T.DEV), g.t("step0/noise", T.DEV)
    if scheme == "f32":
        out = tr.training_step({"image": img}, noise=noise)
        hits = 0
    else:
        with Patch(ops, tr.decoder, scheme) as p:
            out = tr.training_step({"image": img}, noise=noise)
        hits = p.hits
    torch.cuda.synchronize()
    grads = {"enc." + k: p.grad for k, p in tr.encoder.named_parameters()}
    grads.update({"dec." + k: p.grad for k, p in tr.decoder.named_parameters()})
    try:
        med, mx = check_grads_vs_fp64(g, grads, 2.0, scheme)
        verdict = "passes"
    except AssertionError as e:
        med, mx = check_grads_vs_fp64(g, grads, 1e9, scheme)
        verdict = "FAILS (" + str(e).splitlines()[0][:90] + ")"
    dec = [k for k in grads if k.startswith("dec.")]
    ids_same = bool((out["ids_1"].cpu().numpy() == np.asarray(g["step0/ids_1"])).all())
    return dict(scheme=scheme, layers=hits, median=med, max=mx, verdict=verdict, total=float(out["total"]),
                ref_total=float(np.asarray(g["step0/total64"])) if "step0/total64" in g.files else float("nan"), ids_same=ids_same, n_dec=len(dec))


def main():
    names = sys.argv[1:] or ["step_rcfg64_warm.npz", "step_cfg4_32.npz"]
    for name in names:
        print("fixture %s: first training step, decoder 3x3 convolutions with >= %d input channels in the named scheme" % (name, MIN_CIN))
        print("  %-10s %7s   %-28s %-24s %s" % ("scheme", "layers", "gate ratio (median, max)", "total loss - f32 run", "gate at factor 2"))
        base = None
        schemes = os.environ.get("VQW_SPLIT_SCHEMES", "f32,f32_torch,bf16x3_6,bf16x2_4,bf16x3_3,bf16").split(",")
        for scheme in schemes:
            SplitConv.wino_layers = 0
            r = run(name, scheme)
            if scheme in ("f4_grads", "f2_grads"):
                r["scheme"] = "%s[%d]" % (scheme, SplitConv.wino_layers)
            base = r["total"] if base is None else base
            r["ref_total"] = base
            print("  %-10s %7d   %8.3f  %10.3f        %-+24.3e %s%s" % (r["scheme"], r["layers"], r["median"], r["max"], r["total"] - r["ref_total"], r["verdict"],
                                                                    "" if r["ids_same"] else "  [ids differ from the fixture]"))


if __name__ == "__main__":
    main()
