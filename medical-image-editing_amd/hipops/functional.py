"""Functional dispatcher operators with autograd formulas (torch.library.custom_op + register_autograd) over the kernel
operators of hipops/library.py - the operator-level surface SURVEY 8(b) lists for the boundary:

    torch.ops.vqw.conv2d(x, weight, bias, dilation, up2x, skip, relu) -> y
    torch.ops.vqw.instance_norm(x, relu, eps)                         -> (y, mean_rstd)
    torch.ops.vqw.vq_forward(x, embed, id_base)                       -> (q, commit, ids)       [search + gather + commit loss]
    torch.ops.vqw.embed_cross_loss(embed, labels, codebook_kd)        -> (loss, coef)
    torch.ops.vqw.res_tail(a, b)                                      -> (pooled, out)          [ReLU(a+b), MaxPool2d(2)]

Every forward / backward body runs the hand-written HIP kernels through torch.ops.vqw.<kernel> (out-variant, schema with
mutation annotations); nothing here falls back to ATen arithmetic.  The in-place codebook update is the kernel operator
torch.ops.vqw.vq_ema_update(stats, embed!, cluster_size!, embed_avg!, ...), fused Adam is torch.ops.vqw.adam_step(p!, g,
m!, v!, ...).  The nn.Module classes use the richer torch.autograd.Function operators of hipops.ops (side-stream weight
gradients, statistics from conv epilogues, cached weight layouts) on the same kernels; these functional operators are the
composable form: they trace under FakeTensor / torch.compile and pass torch.library.opcheck.
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops
from . import library

library.register()
CL = torch.channels_last


def _fake_nhwc(N, C, H, W, like):
    return torch.empty((N, C, H, W), dtype=torch.float32, device=like.device).contiguous(memory_format=CL)


# ---------------------------------------------------------------------------------------------------------------- conv2d
@torch.library.custom_op("vqw::conv2d", mutates_args=(), device_types="cuda")
def conv2d(x: Tensor, weight: Tensor, bias: Optional[Tensor], dilation: int, up2x: bool, skip: Optional[Tensor], relu: bool) -> Tensor:
    ops._in_custom_op = True       # a training forward below the autograd key, not a forward-only use (ops.WINOGRAD_EVAL)
    try:
        with torch.no_grad():
            return ops.conv2d(x, weight, bias, dilation=dilation, up2x=up2x, skip=skip, relu=relu)
    finally:
        ops._in_custom_op = False


@conv2d.register_fake
def _(x, weight, bias, dilation, up2x, skip, relu):
    s = 2 if up2x else 1
    return _fake_nhwc(x.shape[0], weight.shape[0], x.shape[2] * s, x.shape[3] * s, x)


@torch.library.custom_op("vqw::conv2d_backward", mutates_args=(), device_types="cuda")
def conv2d_backward(gy: Tensor, x: Tensor, weight: Tensor, skip: Optional[Tensor], y_relu: Optional[Tensor], dilation: int,
                    up2x: bool, has_bias: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """-> (gx, gskip, gweight, gbias); gskip / gbias are empty tensors when there is no skip / bias."""
    x0, x1, w = ops.nhwc(x), (ops.nhwc(skip) if skip is not None else None), ops.nhwc(weight)
    up_ws = None
    Cout, Cin = w.shape[0], w.shape[1]
    if up2x and x1 is None and w.shape[2] == 3 and dilation == 1 and ops._L().vqw_conv3x3_up2_supported(Cin, Cout, x0.shape[0], x0.shape[2], x0.shape[3]):
        L = ops._L()
        up_ws = ops._ws(L.vqw_conv3x3_up2_ws_bytes(Cin, Cout), x0)
        ops._lib.check(L.vqw_conv3x3_up2_prepare(ops._p(w), ops._p(up_ws), up_ws.numel(), Cin, Cout, ops._st()), "vqw_conv3x3_up2_prepare")
    g0, g1, gw, gb, _ = ops.conv2d_backward_impl(gy, x0, x1, w, y_relu, dilation, up2x, has_bias, up_ws, True, x1 is not None, True, has_bias)
    return g0, (g1 if g1 is not None else torch.empty(0, device=gy.device)), gw, (gb if gb is not None else torch.empty(0, device=gy.device))


@conv2d_backward.register_fake
def _(gy, x, weight, skip, y_relu, dilation, up2x, has_bias):
    return (torch.empty_like(x), torch.empty_like(skip) if skip is not None else torch.empty(0, device=gy.device),
            torch.empty_like(weight), torch.empty(weight.shape[0], device=gy.device) if has_bias else torch.empty(0, device=gy.device))


def _conv2d_setup(ctx, inputs, output):
    x, weight, bias, dilation, up2x, skip, relu = inputs
    ctx.save_for_backward(x, weight, skip, output if relu else None)
    ctx.cfg = (dilation, up2x, bias is not None)


def _conv2d_bwd(ctx, gy):
    x, weight, skip, y_relu = ctx.saved_tensors
    dilation, up2x, has_bias = ctx.cfg
    gx, gs, gw, gb = torch.ops.vqw.conv2d_backward(gy, x, weight, skip, y_relu, dilation, up2x, has_bias)
    return gx, gw, (gb if has_bias else None), None, None, (gs if skip is not None else None), None


torch.library.register_autograd("vqw::conv2d", _conv2d_bwd, setup_context=_conv2d_setup)


# --------------------------------------------------------------------------------------------------------- instance norm
@torch.library.custom_op("vqw::instance_norm", mutates_args=(), device_types="cuda")
def instance_norm(x: Tensor, relu: bool, eps: float) -> Tuple[Tensor, Tensor]:
    x = ops.nhwc(x)
    N, C, H, W = x.shape
    L = ops._L()
    y = torch.empty_like(x, memory_format=CL)
    mr = torch.empty(N * C * 2, dtype=torch.float32, device=x.device)
    ws = ops._ws(L.vqw_plane_ws_bytes(N, C, H * W), x)
    ops._lib.check(L.vqw_inorm_fwd(ops._p(x), ops._p(y), C, 0, ops._p(mr), ops._p(ws), ws.numel(), N, H * W, C, eps, int(relu), ops._st()), "vqw_inorm_fwd")
    return y, mr


@instance_norm.register_fake
def _(x, relu, eps):
    return _fake_nhwc(*x.shape, x), torch.empty(x.shape[0] * x.shape[1] * 2, dtype=torch.float32, device=x.device)


@torch.library.custom_op("vqw::instance_norm_backward", mutates_args=(), device_types="cuda")
def instance_norm_backward(gy: Tensor, x: Tensor, mean_rstd: Tensor, relu: bool) -> Tensor:
    x, gy = ops.nhwc(x), ops.nhwc(gy)
    N, C, H, W = x.shape
    L = ops._L()
    gx = torch.empty_like(x, memory_format=CL)
    ws = ops._ws(L.vqw_plane_ws_bytes(N, C, H * W), x)
    ops._lib.check(L.vqw_inorm_bwd(ops._p(x), ops._p(mean_rstd), ops._p(gy), C, 0, ops._p(gx), ops._p(ws), ws.numel(), N, H * W, C, int(relu), ops._st()),
                   "vqw_inorm_bwd")
    return gx


@instance_norm_backward.register_fake
def _(gy, x, mean_rstd, relu):
    return _fake_nhwc(*x.shape, x)


def _in_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], output[1])
    ctx.relu = inputs[1]
    ctx.mark_non_differentiable(output[1])


def _in_bwd(ctx, gy, _gmr):
    x, mr = ctx.saved_tensors
    return torch.ops.vqw.instance_norm_backward(gy, x, mr, ctx.relu), None, None


torch.library.register_autograd("vqw::instance_norm", _in_bwd, setup_context=_in_setup)


# ------------------------------------------------------------------------------------------------------------ VQ forward
@torch.library.custom_op("vqw::vq_forward", mutates_args=(), device_types="cuda")
def vq_forward(x: Tensor, embed: Tensor, id_base: int) -> Tuple[Tensor, Tensor, Tensor]:
    """Nearest-codebook search + gather + commitment loss (no codebook update) -> (q, commit, ids (N,H,W) int64)."""
    x = ops.nhwc(x)
    N, D, H, W = x.shape
    K = embed.shape[0]
    L = ops._L()
    ids = torch.empty((N, H, W), dtype=torch.int64, device=x.device)
    q = torch.empty_like(x, memory_format=CL)
    commit = torch.empty((), dtype=torch.float32, device=x.device)
    ws = ops._ws(L.vqw_vq_ws_bytes(N * H * W, D, K), x)
    ops._lib.check(L.vqw_vq_fwd(ops._p(x), ops._p(embed.contiguous()), ops._p(ids), id_base, ops._p(q), ops._p(commit), None, ops._p(ws), ws.numel(),
                                N * H * W, D, K, ops._st()), "vqw_vq_fwd")
    return q, commit, ids


@vq_forward.register_fake
def _(x, embed, id_base):
    N, D, H, W = x.shape
    return _fake_nhwc(N, D, H, W, x), torch.empty((), dtype=torch.float32, device=x.device), torch.empty((N, H, W), dtype=torch.int64, device=x.device)


@torch.library.custom_op("vqw::vq_backward", mutates_args=(), device_types="cuda")
def vq_backward(x: Tensor, q: Tensor, g_q: Optional[Tensor], g_commit: Optional[Tensor]) -> Tensor:
    """Straight-through estimator + commitment gradient: gx = g_q + g_commit * 2 (x - q) / numel."""
    x = ops.nhwc(x)
    gx = torch.empty_like(x, memory_format=CL)
    gq = ops.nhwc(g_q) if g_q is not None else None
    gc = g_commit.contiguous() if g_commit is not None else None
    ops._lib.check(ops._L().vqw_vq_bwd(ops._p(x), ops._p(ops.nhwc(q)), ops._p(gq), ops._p(gc), ops._p(gx), x.numel(), ops._st()), "vqw_vq_bwd")
    return gx


@vq_backward.register_fake
def _(x, q, g_q, g_commit):
    return _fake_nhwc(*x.shape, x)


def _vq_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], output[0])
    ctx.mark_non_differentiable(output[2])


def _vq_bwd(ctx, gq, gcommit, _gids):
    x, q = ctx.saved_tensors
    return torch.ops.vqw.vq_backward(x, q, gq, gcommit), None, None


torch.library.register_autograd("vqw::vq_forward", _vq_bwd, setup_context=_vq_setup)


# ------------------------------------------------------------------------------------------------------ embed cross loss
@torch.library.custom_op("vqw::embed_cross_loss", mutates_args=(), device_types="cuda")
def embed_cross_loss(embed: Tensor, labels: Tensor, codebook_kd: Tensor) -> Tuple[Tensor, Tensor]:
    """EmbeddingLoss._calc_cross_loss on integer labels (0 = out of frame) -> (loss, coef (B*K) saved for backward)."""
    e = ops.nhwc(embed)
    B, D, H, W = e.shape
    K = codebook_kd.shape[0]
    L = ops._L()
    loss = torch.empty((), dtype=torch.float32, device=e.device)
    coef = torch.empty(B * K, dtype=torch.float32, device=e.device)
    ws = ops._ws(L.vqw_cross_ws_bytes(B, K, H * W), e)
    ops._lib.check(L.vqw_cross_loss_fwd(ops._p(e), ops._p(labels.contiguous()), ops._p(codebook_kd.contiguous()), ops._p(loss), ops._p(coef), ops._p(ws),
                                        ws.numel(), B, H * W, D, K, ops._st()), "vqw_cross_loss_fwd")
    return loss, coef


@embed_cross_loss.register_fake
def _(embed, labels, codebook_kd):
    return torch.empty((), dtype=torch.float32, device=embed.device), torch.empty(embed.shape[0] * codebook_kd.shape[0], dtype=torch.float32, device=embed.device)


@torch.library.custom_op("vqw::embed_cross_loss_backward", mutates_args=(), device_types="cuda")
def embed_cross_loss_backward(g: Tensor, embed: Tensor, labels: Tensor, codebook_kd: Tensor, coef: Tensor) -> Tensor:
    e = ops.nhwc(embed)
    B, D, H, W = e.shape
    K = codebook_kd.shape[0]
    ge = torch.empty_like(e, memory_format=CL)
    ops._lib.check(ops._L().vqw_cross_loss_bwd(ops._p(e), ops._p(labels.contiguous()), ops._p(codebook_kd.contiguous()), ops._p(coef), ops._p(g.contiguous()),
                                               ops._p(ge), B, H * W, D, K, ops._st()), "vqw_cross_loss_bwd")
    return ge


@embed_cross_loss_backward.register_fake
def _(g, embed, labels, codebook_kd, coef):
    return _fake_nhwc(*embed.shape, embed)


def _cl_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1], inputs[2], output[1])
    ctx.mark_non_differentiable(output[1])


def _cl_bwd(ctx, g, _gcoef):
    e, lab, cb, coef = ctx.saved_tensors
    return torch.ops.vqw.embed_cross_loss_backward(g, e, lab, cb, coef), None, None


torch.library.register_autograd("vqw::embed_cross_loss", _cl_bwd, setup_context=_cl_setup)


# ------------------------------------------------------------------------------------------- ResBlock tail (pool_add_relu)
@torch.library.custom_op("vqw::res_tail", mutates_args=(), device_types="cuda")
def res_tail(a: Tensor, b: Tensor) -> Tuple[Tensor, Tensor]:
    """out = ReLU(a + b), pooled = MaxPool2d(2)(out) (blocks.py:29-36) -> (pooled, out)."""
    a, b = ops.nhwc(a), ops.nhwc(b)
    N, C, H, W = a.shape
    out = torch.empty_like(a, memory_format=CL)
    pooled = ops.empty_nhwc(N, C, H // 2, W // 2, a)
    ops._lib.check(ops._L().vqw_res_tail_fwd(ops._p(a), ops._p(b), ops._p(out), ops._p(pooled), N, H, W, C, ops._st()), "vqw_res_tail_fwd")
    return pooled, out


@res_tail.register_fake
def _(a, b):
    N, C, H, W = a.shape
    return _fake_nhwc(N, C, H // 2, W // 2, a), _fake_nhwc(N, C, H, W, a)


@torch.library.custom_op("vqw::res_tail_backward", mutates_args=(), device_types="cuda")
def res_tail_backward(out: Tensor, g_pooled: Optional[Tensor], g_out: Optional[Tensor]) -> Tensor:
    """[out > 0] * (g_out + g_pooled routed to the window arg-max): the gradient of both a and b."""
    out = ops.nhwc(out)
    N, C, H, W = out.shape
    g = torch.empty_like(out, memory_format=CL)
    gp = ops.nhwc(g_pooled) if g_pooled is not None else None
    go = ops.nhwc(g_out) if g_out is not None else None
    ops._lib.check(ops._L().vqw_res_tail_bwd(ops._p(out), ops._p(gp), ops._p(go), ops._p(g), N, H, W, C, ops._st()), "vqw_res_tail_bwd")
    return g


@res_tail_backward.register_fake
def _(out, g_pooled, g_out):
    return _fake_nhwc(*out.shape, out)


def _rt_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])


def _rt_bwd(ctx, g_pooled, g_out):
    (out,) = ctx.saved_tensors
    g = torch.ops.vqw.res_tail_backward(out, g_pooled, g_out)
    return g, g


torch.library.register_autograd("vqw::res_tail", _rt_bwd, setup_context=_rt_setup)
