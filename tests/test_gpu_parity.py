"""GPU parity tests: the HIP path (through the C ABI / hipops) against the reference's golden vectors and the
CPU oracle.  Run with `pytest -m gpu` on an MI355X.

Tolerances (relative L2 unless stated): fp32 everywhere.
  single kernels vs torch CPU ............ 2e-5
  block forward / backward vs golden ..... 1e-4 / 1e-3
  VQ ids ................................. bit-exact wherever the top-1/top-2 score gap > 1e-4*(1+|gap|)
  training step (step 0) ................. losses 5e-4, recon 4e-4 (eval forward 3e-5), gradients per fixture (GRAD_TOL) + fp64 gate
  training steps 1, 2 .................... within the reference's own multi-step spread (check_later_step; lr = 1e-6 fixture: losses 5e-4,
                                           ids bit-exact where clear, VQ / BatchNorm state 1e-4, parameters elementwise)
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import rel_err, assert_close, build_models, check_init, step_cfg, check_grads_vs_fp64, grad_gate, assert_ids_equal_where_clear
from test_oracle_golden import check_step, check_later_step, GRAD_TOL, apply_warm_state

# Reconstructions against the reference fixtures (round 4: the measured errors are printed by the tests; the bounds are ~2.5x
# the largest of them, down from 5e-3).  Measured: eval / mask-guided forward 5e-6 ... 1.2e-5 on all five fixtures; training
# step 0: 5e-6 (small), 1.1e-5 (64^2 warm), 3.2e-5 (config 4), 1.5e-4 (32^2 cold start: cluster_size = 0).
EVAL_RECON_TOL = 3e-5
STEP0_RECON_TOL = 4e-4

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from hipops import ops
    return ops


# --------------------------------------------------------------------------------------------------
# single kernels
# --------------------------------------------------------------------------------------------------
CONV_CASES = [
    # N, H, W, C0, C1, up, Cout, ks, dil, bias, relu
    (2, 16, 16, 16, 0, False, 32, 3, 1, True, False),
    (2, 16, 16, 32, 0, False, 16, 3, 1, True, True),
    (1, 24, 24, 16, 0, False, 16, 3, 6, False, False),     # dilated (ASPP)
    (1, 20, 20, 32, 0, False, 32, 3, 18, False, False),    # dilation close to the image size
    (2, 16, 16, 32, 16, True, 16, 3, 1, True, False),      # up-sampled + concat (UpBlock)
    (2, 8, 8, 64, 0, True, 32, 3, 1, True, False),         # up-sampled only (StyledResUpBlock)
    (2, 16, 16, 16, 0, False, 32, 1, 1, False, False),     # 1x1 projection
    (3, 12, 10, 48, 0, False, 80, 3, 1, True, False),      # ragged tiles, H != W
    (2, 16, 16, 128, 0, False, 160, 3, 1, True, False),    # >128 couts: two N tiles
    (2, 16, 16, 1, 0, False, 16, 3, 1, True, False),       # 1-channel stem  (generic kernel)
    (2, 16, 16, 32, 0, False, 1, 1, 1, True, False),       # 32->1 head      (generic kernel)
    # stems whose 64 x Cout wave tiles leave through LDS as 1 KB stores (round 4): 1 -> 16 / 32, 3x3 and 1x1, ReLU; the 32 -> 1 head's
    # input gradient is the 1 -> 32 1x1 form; a pixel count that is not a multiple of 64 keeps the per-thread stores
    (2, 24, 16, 1, 0, False, 16, 1, 1, False, False),
    (1, 32, 64, 1, 0, False, 32, 3, 1, True, True),
    (2, 8, 64, 32, 0, False, 1, 1, 1, True, False),
    (1, 10, 6, 1, 0, False, 32, 3, 1, True, False),
    (2, 10, 10, 3, 0, False, 5, 3, 1, True, False),        # odd channels    (generic kernel)
    (1, 2, 2, 32, 0, False, 64, 3, 1, True, False),        # tiny map (deepest level of a 32x32 input)
    (2, 4, 4, 64, 32, True, 32, 3, 1, True, False),
    # widths that are multiples of 32 -> the all-taps-per-wave wgrad kernel
    (2, 32, 32, 32, 0, False, 32, 3, 1, True, False),
    (1, 64, 64, 32, 0, False, 32, 3, 6, False, False),
    (1, 64, 64, 16, 0, False, 16, 3, 18, False, False),
    (2, 32, 32, 32, 16, True, 16, 3, 1, True, False),
    (1, 32, 64, 64, 0, True, 32, 3, 1, True, False),
    (2, 32, 32, 96, 0, False, 80, 3, 2, True, False),
    (2, 32, 32, 128, 0, True, 64, 3, 1, True, False),     # collapsed up-conv incl. its low-res wgrad (w_low = 16)
    (1, 64, 32, 32, 0, True, 48, 3, 1, False, True),
    # LDS-resident halo-tile kernel (3x3, W % 32 == 0, narrow cout tile): every template configuration
    (2, 32, 32, 32, 0, False, 64, 3, 1, True, True),       # one channel chunk, weights stay in LDS, 64-wide cout tile
    (1, 20, 64, 64, 0, False, 32, 3, 1, True, False),      # channel chunks streamed, H not a multiple of the tile
    (2, 16, 32, 64, 0, False, 64, 3, 1, True, False),
    (1, 16, 32, 160, 0, False, 32, 3, 1, True, False),     # ASPP projection: 10 chunks; its dgrad has 3 cout tiles
    (2, 16, 32, 16, 0, False, 32, 3, 1, False, False),
    (3, 12, 32, 48, 0, False, 80, 3, 1, True, False),
    (1, 16, 64, 16, 0, False, 48, 3, 1, True, True),
    # 64-cout Winograd kernel (conv_wino64.hip: Cout % 64 == 0, Cin % 16 == 0; forward under the winograd_forward fixture,
    # input gradient of the mirrored channel counts always): both region shapes, ragged H, several regions per workgroup,
    # two cout tiles, the shortest legal channel walk (two chunks)
    (2, 48, 64, 64, 0, False, 128, 3, 1, True, True),
    (3, 16, 16, 64, 0, False, 64, 3, 1, True, False),
    (1, 20, 16, 128, 0, False, 64, 3, 1, True, False),
    (1, 256, 320, 16, 0, False, 64, 3, 1, True, False),
    (2, 22, 32, 64, 0, False, 16, 3, 1, False, False),     # its input gradient: 16 -> 64
    # (32 co x 32 ci)-block Winograd weight gradient (round 4: Cout % 64 == 32, Cin % 32 == 0, W % 32 == 0, H % 4 == 0): three co
    # blocks, several regions per workgroup and images, an odd number of regions, no bias
    (3, 64, 64, 32, 0, False, 96, 3, 1, True, False),
    (2, 12, 96, 64, 0, False, 32, 3, 1, False, False),
    # streaming 1x1 on the vector lanes (round 4: both channel counts in {16, 32, 64}, >= 128 x 128 pixels per image): forward
    # and - with the roles of the channel counts swapped - input gradient; with and without bias
    (2, 128, 128, 16, 0, False, 32, 1, 1, False, False),
    (1, 128, 256, 32, 0, False, 32, 1, 1, True, False),
    (1, 128, 128, 32, 0, False, 64, 1, 1, False, False),
    (1, 256, 128, 64, 0, False, 16, 1, 1, True, False),
    # up-sampled 3x3 in Winograd form with nine products (conv_wino_up.hip): ragged H, the shortest channel walk (two chunks),
    # several regions per strip and image, both N-tile widths (64 / 128 input channels)
    (1, 40, 32, 64, 0, True, 16, 3, 1, True, False),
    (3, 16, 96, 128, 0, True, 32, 3, 1, True, True),
    (2, 64, 64, 192, 0, True, 96, 3, 1, False, False),
    # 16-cout layers on the 16x16x4 MFMA path of the halo kernel (16 x 32 pixel tiles, ragged H, two sources)
    (2, 32, 32, 16, 0, False, 16, 3, 1, True, True),
    (1, 40, 64, 48, 0, False, 16, 3, 1, True, False),
    (2, 48, 32, 32, 16, True, 16, 3, 1, False, False),
    # more dilated / narrow shapes (ragged sizes, dilation larger than a tile, wide couts on a 16-pixel map)
    (2, 50, 38, 32, 0, False, 32, 3, 6, True, False),
    (1, 96, 96, 32, 0, False, 32, 3, 18, False, True),
    (1, 64, 64, 48, 0, False, 24, 3, 12, True, False),
    (2, 16, 16, 64, 0, False, 128, 3, 1, True, False),
    (3, 24, 40, 16, 0, False, 96, 3, 2, True, False),
    # dilated 32-channel layers on LDS-resident rows walked along the residue chains (conv_dil.hip): full 256-pixel rows,
    # ragged H, fewer than 32 couts, dilation == H, several positions per workgroup, narrow rows (idle waves)
    (2, 37, 256, 32, 0, False, 32, 3, 18, True, True),
    (3, 40, 64, 32, 0, False, 24, 3, 2, True, False),
    (2, 24, 128, 32, 0, False, 32, 3, 12, False, False),
    (1, 8, 32, 32, 0, False, 32, 3, 8, True, False),
    (4, 130, 64, 32, 0, False, 32, 3, 7, False, False),
    (2, 200, 32, 32, 0, False, 32, 3, 5, True, True),
    # dilation 2 in Winograd form on the four phase images of the tensors (round 4: pitch parameters of the 32-cout kernels): even
    # sizes with W / 2 a multiple of 32; several regions per phase image, 2 ... 8 channel chunks, bias / ReLU epilogues
    (2, 16, 64, 32, 0, False, 32, 3, 2, True, False),
    (1, 40, 128, 64, 0, False, 32, 3, 2, False, True),
    (3, 24, 64, 32, 0, False, 96, 3, 2, True, False),
    # Winograd F(2x2, 3x3) form of plain 3x3 layers (conv_wino.hip): ragged H (not a multiple of the 16-row region, odd),
    # several 32-pixel strips, two cout tiles with a partial one, 2 .. 16 channel chunks, bias + ReLU epilogue
    (2, 16, 32, 16, 0, False, 32, 3, 1, True, True),
    (1, 23, 64, 24, 0, False, 48, 3, 1, True, False),
    (3, 40, 32, 64, 0, False, 64, 3, 1, False, False),
    (1, 16, 96, 128, 0, False, 32, 3, 1, True, True),
    (2, 7, 32, 32, 0, False, 96, 3, 1, True, False),
    # ... on maps whose width is a multiple of 16 only (32 x 16 regions: the 16 x 16 level; three strips, ragged H)
    (2, 16, 16, 64, 0, False, 64, 3, 1, True, True),
    (3, 16, 16, 256, 0, False, 256, 3, 1, False, False),
    (1, 40, 48, 32, 0, False, 64, 3, 1, True, False),
    # Winograd-form weight gradient with two sources (up-sampled src0 | skip): UpBlock convs; ci blocks in either source
    (2, 32, 32, 64, 32, True, 32, 3, 1, True, False),
    (1, 16, 64, 32, 16, True, 64, 3, 1, False, True),
    (2, 32, 16, 128, 64, True, 64, 3, 1, True, False),
    (1, 24, 32, 16, 32, False, 32, 3, 1, True, False),
    # ... whose input gradients leave the epilogue of the 64-cout kernel as two tensors (round 4: vqw_conv3x3_wino_fwd_split):
    # both workgroup shapes, ragged heights, with and without the up-sampling
    (2, 20, 32, 128, 64, True, 64, 3, 1, True, False),
    (1, 22, 64, 64, 32, True, 32, 3, 1, False, True),
    (1, 24, 64, 64, 32, False, 32, 3, 1, True, False),
    # collapsed up-sampled forward on the halo kernel's 4-tap form (low-resolution width a multiple of 32): two cout tile
    # widths, ragged low-res height, ReLU epilogue
    (2, 32, 128, 64, 0, True, 32, 3, 1, True, False),
    (1, 48, 64, 128, 0, True, 96, 3, 1, True, True),
    (1, 20, 64, 32, 0, True, 64, 3, 1, False, False),
    # wide 1-channel stem (BASELINE config 4's first block 1 -> 256): lanes over the output channels
    (2, 24, 20, 1, 0, False, 256, 3, 1, True, False),
    (1, 32, 32, 1, 0, False, 256, 1, 1, False, False),
    (2, 16, 16, 1, 0, False, 128, 3, 1, True, True),
    (2, 16, 16, 1, 0, False, 64, 3, 1, True, False),
    (1, 20, 12, 1, 0, False, 192, 3, 2, False, False),
    # thin 1x1 layers on large maps: the streaming weight gradient (pixel pairs through one MFMA per pair)
    (2, 256, 256, 16, 0, False, 32, 1, 1, False, False),
    (2, 256, 256, 32, 0, False, 32, 1, 1, True, False),
    (3, 208, 212, 32, 0, False, 64, 1, 1, False, False),
    (8, 128, 128, 64, 0, False, 64, 1, 1, False, False),
    (4, 192, 192, 16, 0, False, 128, 1, 1, False, False),
]


def _conv_ref(x0, x1, w, b, up, dil, relu):
    xin = x0
    if up:
        xin = xin.repeat_interleave(2, 2).repeat_interleave(2, 3)
    if x1 is not None:
        xin = torch.cat([xin, x1], 1)
    k = w.shape[-1]
    y = F.conv2d(xin, w, b, padding=dil * (k // 2), dilation=dil)
    return torch.relu(y) if relu else y


@pytest.fixture
def winograd_forward():
    """The Winograd form also for the FORWARD of plain 3x3 layers (opt-in in the product: VQW_WINOGRAD_FWD=1; the input
    gradient takes it by default)."""
    ops = _ops()
    old = ops.WINOGRAD_FWD
    ops.WINOGRAD_FWD = True
    yield ops
    ops.WINOGRAD_FWD = old


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[7] == 3 and c[8] == 1 and not c[5] and not c[4] and c[2] % 16 == 0
                                  and c[3] >= 16 and c[3] % 8 == 0 and c[6] >= 32])
def test_conv2d_winograd_forward(case, winograd_forward):
    test_conv2d(case, 0)


@pytest.mark.parametrize("backend", [0, 1, 2])     # auto, generic VALU kernels, implicit-GEMM MFMA without halo tiles
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(case, backend):
    ops = _ops()
    N, H, W, C0, C1, up, Cout, ks, dil, bias, relu = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    hs, wsz = (H // 2, W // 2) if up else (H, W)
    x0 = torch.randn(N, C0, hs, wsz, generator=g, dtype=torch.float64)
    x1 = torch.randn(N, C1, H, W, generator=g, dtype=torch.float64) if C1 else None
    w = torch.randn(Cout, C0 + C1, ks, ks, generator=g, dtype=torch.float64) * 0.2
    b = torch.randn(Cout, generator=g, dtype=torch.float64) if bias else None
    r = torch.randn(N, Cout, H, W, generator=g, dtype=torch.float64)
    leaves = [t.clone().requires_grad_(True) for t in (x0, x1, w, b) if t is not None]
    it = iter(leaves)
    rx0 = next(it); rx1 = next(it) if C1 else None; rw = next(it); rb = next(it) if bias else None
    yref = _conv_ref(rx0, rx1, rw, rb, up, dil, relu)
    (yref * r).sum().backward()

    old = ops.set_conv_backend(backend)
    try:
        dx0 = x0.float().to(DEV).requires_grad_(True)
        dx1 = x1.float().to(DEV).requires_grad_(True) if C1 else None
        dw = w.float().to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        db = b.float().to(DEV).requires_grad_(True) if bias else None
        y = ops.conv2d(dx0, dw, db, dilation=dil, up2x=up, skip=dx1, relu=relu)
        (y * r.float().to(DEV)).sum().backward()
        torch.cuda.synchronize()
    finally:
        ops.set_conv_backend(old)
    tol = 2e-5
    assert_close(y, yref, tol, "y")
    assert_close(dx0.grad, rx0.grad, tol, "dx0")
    if C1:
        assert_close(dx1.grad, rx1.grad, tol, "dx1")
    assert_close(dw.grad, rw.grad, tol, "dw")
    if bias:
        assert_close(db.grad, rb.grad, tol, "db")


def test_mfma_equals_generic_bitwise_shape():
    """The MFMA kernel is a k-ordered fp32 FMA chain like the VALU kernel; they agree to rounding (not bitwise:
    the k order inside an 8-channel group differs)."""
    ops = _ops()
    x = torch.randn(2, 32, 16, 16, device=DEV)
    w = torch.randn(32, 32, 3, 3, device=DEV).contiguous(memory_format=torch.channels_last)
    ya = ops.conv2d(x, w)
    old = ops.set_conv_backend(1)
    yb = ops.conv2d(x, w)
    ops.set_conv_backend(old)
    assert_close(ya, yb, 1e-6, "mfma vs generic")


@pytest.mark.parametrize("shape,relu", [((2, 16, 16, 16), True), ((3, 48, 8, 8), False), ((2, 5, 10, 10), True),
                                        ((1, 512, 4, 4), True), ((2, 160, 12, 12), True)])
def test_instance_norm(shape, relu):
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(*shape, generator=g, dtype=torch.float64) * 2 + 0.3).requires_grad_(True)
    r = torch.randn(*shape, generator=g, dtype=torch.float64)
    y = F.instance_norm(x, eps=1e-5)
    y = torch.relu(y) if relu else y
    (y * r).sum().backward()
    dx = x.detach().float().to(DEV).requires_grad_(True)
    dy = ops.instance_norm(dx, relu=relu)
    (dy * r.float().to(DEV)).sum().backward()
    assert_close(dy, y, 2e-5, "inorm y")
    assert_close(dx.grad, x.grad, 5e-5, "inorm dx")


def test_pool_add_tanh_mse():
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    a = torch.randn(2, 16, 12, 12, generator=g).requires_grad_(True)
    b = torch.randn(2, 16, 12, 12, generator=g).requires_grad_(True)
    tgt = torch.randn(2, 16, 6, 6, generator=g)
    out = torch.relu(a + b)
    pooled = F.max_pool2d(out, 2)
    loss = F.mse_loss(torch.tanh(pooled), tgt) + 0.1 * out.sum()
    loss.backward()
    da = a.detach().to(DEV).requires_grad_(True)
    db = b.detach().to(DEV).requires_grad_(True)
    dout = ops.add(da, db, relu=True)
    dpool = ops.maxpool2(dout)
    dmse = ops.mse_loss(ops.tanh(dpool), tgt.to(DEV))
    dloss = ops.weighted_sum([dmse, (dout * 1.0).sum()], [1.0, 0.1])
    dloss.backward()
    assert_close(dpool, pooled, 1e-6, "pool")
    assert_close(dloss, loss, 1e-5, "loss")
    assert_close(da.grad, a.grad, 1e-5, "da")
    assert_close(db.grad, b.grad, 1e-5, "db")


def test_maxpool_ties_route_to_first():
    """A piecewise-constant (quantised) map puts max-pool windows on exact ties; the gradient must go to the first
    maximum in row-major window order, as ATen does (see oracle STE note)."""
    ops = _ops()
    x = torch.ones(1, 4, 4, 4)
    x[0, :, 2:, :] = 2.0
    xr = x.clone().requires_grad_(True)
    F.max_pool2d(xr, 2).sum().backward()
    dx = x.to(DEV).requires_grad_(True)
    ops.maxpool2(dx).sum().backward()
    assert torch.equal(dx.grad.cpu(), xr.grad)


def test_res_tail_matches_separate_ops_and_aten():
    """ResBlock tail as one autograd node (out = ReLU(a + b); pooled = MaxPool2d(2)(out)): values and both input
    gradients are bit-identical to the separate add / max-pool operators and to ATen on CPU, with exact ties (a
    piecewise-constant map), negative pre-activations and either output unused."""
    ops = _ops()
    torch.manual_seed(7)
    a = torch.randn(2, 8, 12, 16)
    b = torch.randn(2, 8, 12, 16)
    a[0, :, :6] = torch.round(a[0, :, :6])          # ties inside 2x2 windows
    b[0, :, :6] = torch.round(b[0, :, :6])
    rp, ro = torch.randn(2, 8, 6, 8), torch.randn(2, 8, 12, 16)
    for use_p, use_o in ((True, True), (True, False), (False, True)):
        ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        out = F.relu(ar + br)
        pooled = F.max_pool2d(out, 2)
        ((pooled * rp).sum() * use_p + (out * ro).sum() * use_o).backward()
        grads = []
        for fused in (True, False):
            ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
            if fused:
                pd, od = ops.res_tail(ad, bd)
            else:
                od = ops.add(ad, bd, relu=True)
                pd = ops.maxpool2(od)
            assert torch.equal(pd.cpu(), pooled.detach()) and torch.equal(od.cpu(), out.detach())
            loss = 0
            if use_p:
                loss = loss + (pd * rp.to(DEV)).sum()
            if use_o:
                loss = loss + (od * ro.to(DEV)).sum()
            loss.backward()
            grads.append((ad.grad.cpu(), bd.grad.cpu()))
        assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])
        assert torch.equal(grads[0][0], ar.grad) and torch.equal(grads[0][1], br.grad)


STATS_CASES = [
    # N, H, W, C0, C1, up, Cout        every halo-kernel template configuration, ragged rows, partial cout tile
    (2, 32, 32, 32, 0, False, 32),     # one chunk, weights resident
    (2, 32, 32, 32, 0, False, 64),     # 64-wide cout tile, 4-row tiles
    (1, 24, 64, 64, 0, False, 32),     # streamed chunks, three tile rows
    (2, 16, 32, 64, 0, False, 48),     # two cout tiles, the second one partial
    (2, 16, 32, 16, 0, False, 16),     # 16-wide MFMA variant
    (2, 48, 64, 24, 0, False, 96),     # Winograd form: three regions per strip, three cout tiles
    (1, 16, 32, 128, 0, False, 48),    # Winograd form: 16 chunks, partial cout tile
    (2, 32, 16, 64, 0, False, 64),     # Winograd form on a 16-wide map (32 x 16 regions); direct form: no statistics
    (2, 24, 64, 32, 0, False, 128),    # 64-cout Winograd kernel: 8 x 32 regions, two cout tiles
    (3, 16, 16, 64, 0, False, 64),     # ... 16 x 16 regions (one per image)
    (1, 64, 96, 16, 0, False, 64),     # ... two chunks, 24 regions
    (2, 32, 32, 32, 16, True, 16),     # up-sampled + concat source, 16 couts
    (1, 32, 64, 64, 32, True, 32),     # two sources, 32 couts
]
STATS_CASES_GEMM = [
    # N, H, W, Cin, Cout, ks, dil, up     implicit-GEMM kernel (1x1, dilated) and the collapsed up-sampled form
    (2, 32, 32, 16, 32, 1, 1, False),
    (2, 16, 16, 64, 128, 1, 1, False),
    (1, 32, 32, 32, 32, 3, 6, False),
    (2, 16, 16, 32, 160, 3, 2, False),
    (2, 64, 256, 32, 32, 3, 18, False),   # row-chain kernel: one partial per output row
    (3, 48, 64, 32, 24, 3, 2, False),
    (2, 150, 96, 32, 32, 3, 12, False),
    (2, 32, 32, 64, 32, 3, 1, True),      # collapsed: four parity launches, each a quarter of every plane
    (1, 32, 64, 128, 64, 3, 1, True),
    (2, 48, 128, 64, 32, 3, 1, True),     # halo kernel's 4-tap form: 4 x 3 x 2 partials per plane
    (2, 128, 128, 16, 32, 1, 1, False),   # streaming 1x1 kernel: 1024-pixel tiles, shifted sums per thread
    (1, 128, 256, 32, 32, 1, 1, False),
    (2, 128, 192, 32, 64, 1, 1, False),   # 512-pixel tiles
]


def _plane_sums_from_partials(part, N, Cout, HW):
    """(sum, sum of squares) per plane from the conv epilogue's per-tile (sum, M2 about the tile mean) partials, combined in
    double exactly as the norm kernels do: sum x^2 = sum_t [M2_t + (sum_t)^2 / n_t], n_t = HW / tiles."""
    p = part.view(N, -1, Cout, 2).double().cpu()
    nt = HW / p.shape[1]
    return p[..., 0].sum(1), (p[..., 1] + p[..., 0] ** 2 / nt).sum(1)


@pytest.mark.parametrize("case", STATS_CASES_GEMM)
def test_conv_epilogue_statistics_implicit_gemm(case):
    ops = _ops()
    N, H, W, Cin, Cout, ks, dil, up = case
    torch.manual_seed(sum(case))
    x = torch.randn(N, Cin, H // 2 if up else H, W // 2 if up else W, device=DEV)
    w = (torch.randn(Cout, Cin, ks, ks, device=DEV) * 0.1).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device=DEV)
    y, part = ops.conv2d(x, w, b, dilation=dil, up2x=up, want_stats=True)
    assert part is not None, "shape should be served"
    assert torch.equal(y, ops.conv2d(x, w, b, dilation=dil, up2x=up))
    s1, s2 = _plane_sums_from_partials(part, N, Cout, H * W)
    yd = y.double().cpu()
    assert_close(s1, yd.sum((2, 3)), 2e-6, "sum", atol=1e-4)
    assert_close(s2, (yd * yd).sum((2, 3)), 2e-6, "sum of squares")
    assert_close(ops.instance_norm(y, relu=True, part=part), ops.instance_norm(y, relu=True), 2e-6, "instance norm from conv partials")


@pytest.mark.parametrize("case", [c for c in STATS_CASES if not c[4] and not c[5] and c[3] >= 16 and c[6] >= 32])
def test_conv_epilogue_statistics_winograd_forward(case, winograd_forward):
    test_conv_epilogue_statistics(case)


@pytest.mark.parametrize("case", STATS_CASES)
def test_conv_epilogue_statistics(case):
    """conv2d(..., want_stats=True): the per-tile (sum, M2) partials of the halo kernel's epilogue combine to the plane sums
    of the output, and the InstanceNorm fed with them equals the one that reduces itself."""
    ops = _ops()
    N, H, W, C0, C1, up, Cout = case
    torch.manual_seed(sum(case))
    x0 = torch.randn(N, C0, H // 2 if up else H, W // 2 if up else W, device=DEV)
    x1 = torch.randn(N, C1, H, W, device=DEV) if C1 else None
    w = (torch.randn(Cout, C0 + C1, 3, 3, device=DEV) * 0.1).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device=DEV)
    y, part = ops.conv2d(x0, w, b, up2x=up, skip=x1, want_stats=True)
    assert part is not None, "shape should be served by the halo kernel"
    y_ref = ops.conv2d(x0, w, b, up2x=up, skip=x1)
    # the same kernel with and without the statistics epilogue, unless only one of the two forms serves the statistics for
    # the shape (a plain layer whose H is not a multiple of the Winograd region keeps the direct form when they are wanted)
    same_kernel = not (ops.WINOGRAD_FWD or not torch.is_grad_enabled()) or up or C1 or H % (16 if W % 32 == 0 else 32) == 0 or \
        not ops._L().vqw_conv3x3_wino_supported(C0, Cout, N, H, W)
    if same_kernel:
        assert torch.equal(y, y_ref)
    else:
        assert_close(y, y_ref, 2e-6, "direct vs Winograd form")
    s1, s2 = _plane_sums_from_partials(part, N, Cout, H * W)
    yd = y.double().cpu()
    assert_close(s1, yd.sum((2, 3)), 2e-6, "sum", atol=1e-4)
    assert_close(s2, (yd * yd).sum((2, 3)), 2e-6, "sum of squares")
    a = ops.instance_norm(y, relu=True, part=part)
    r = ops.instance_norm(y, relu=True)
    assert_close(a, r, 2e-6, "instance norm from conv partials")


def test_up_block_with_cout_not_multiple_of_four():
    """StyledResUpBlock whose out_channels is not a multiple of 4 (10): the collapsed 3x3-over-upsample form needs
    Cout % 4 == 0 for its input gradient, so the layer must take a route it can also back-propagate."""
    from networks import blocks as Bk
    from oracle import vqwnet_ref as O
    torch.manual_seed(5)
    m = Bk.StyledResUpBlock(16, 8, 10, use_pixel_shuffle=False)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    down, skip, r = torch.randn(2, 16, 8, 8), torch.randn(2, 8, 16, 16), torch.randn(2, 10, 16, 16)
    P = {"m." + k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    for k in O.trainable_keys(P):
        P[k].requires_grad_(True)
    d64, s64 = down.double().requires_grad_(True), skip.double().requires_grad_(True)
    y64 = O.styled_res_up_block(P, "m", d64, s64, True)
    (y64 * r.double()).sum().backward()
    mm = m.to(DEV).train()
    d_, s_ = down.to(DEV).requires_grad_(True), skip.to(DEV).requires_grad_(True)
    y = mm(d_, s_)
    (y * r.to(DEV)).sum().backward()
    assert_close(y, y64, 1e-5, "y")
    assert_close(d_.grad, d64.grad, 1e-5, "g_down")
    assert_close(s_.grad, s64.grad, 1e-5, "g_skip")
    for k, p in mm.named_parameters():
        assert_close(p.grad, P["m." + k].grad, 2e-5, "g " + k, atol=2e-5)     # conv1 / conv2 biases sit in front of a BatchNorm: zero + noise


def test_conv_epilogue_statistics_need_whole_tiles():
    """The partials carry no pixel count: a plane whose height is not a multiple of the tile height is not served (the norm
    then reduces the plane itself) and the result is the same."""
    ops = _ops()
    torch.manual_seed(3)
    x = torch.randn(1, 64, 20, 64, device=DEV)
    w = (torch.randn(32, 64, 3, 3, device=DEV) * 0.1).contiguous(memory_format=torch.channels_last)
    y, part = ops.conv2d(x, w, None, want_stats=True)
    assert part is None
    assert torch.equal(y, ops.conv2d(x, w, None))
    assert torch.equal(ops.instance_norm(y, relu=True, part=part), ops.instance_norm(y, relu=True))


@pytest.mark.parametrize("ks,Cin,Cout,S", [(1, 16, 16, 32), (3, 32, 32, 32), (3, 16, 16, 64), (1, 32, 64, 64), (1, 16, 32, 128)])
def test_conv_epilogue_statistics_far_from_zero(ks, Cin, Cout, S):
    """Planes whose mean is ~60 standard deviations off zero (conv of a nearly constant positive input, as behind ReLUs on a
    quantised map): the InstanceNorm fed from the conv epilogue must be as exact as the one that reduces the plane in
    double - var = E[x^2] - mean^2 from fp32 sums was 3e-4 off here, the (sum, M2) partials are at 1e-5."""
    ops = _ops()
    torch.manual_seed(ks * 100 + Cin)
    x = (torch.randn(2, Cin, S, S) * 0.05 + 3.0 * torch.randn(1, Cin, 1, 1)).to(DEV)
    w = (torch.randn(Cout, Cin, ks, ks) / (Cin * ks * ks) ** 0.5).to(DEV).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device=DEV)
    y, part = ops.conv2d(x, w, b, want_stats=True)
    assert part is not None
    ref = torch.nn.functional.instance_norm(y.double().cpu(), eps=1e-5)
    a = ops.instance_norm(y, part=part)
    r = ops.instance_norm(y)
    e_a = float((a.double().cpu() - ref).norm() / ref.norm())
    e_r = float((r.double().cpu() - ref).norm() / ref.norm())
    assert e_a <= 2.0 * e_r + 1e-6, "epilogue statistics %.2e vs own reduction %.2e" % (e_a, e_r)


def test_res_tail_norm_matches_separate_ops():
    """The ResBlock tail that normalises its raw inputs (res_tail_norm) against InstanceNorm + res_tail: values and both
    input gradients (fp32 rounding of a fused multiply-add is the only difference)."""
    ops = _ops()
    torch.manual_seed(11)
    x2 = torch.randn(2, 16, 16, 32, device=DEV) * 2 + 0.5
    xid = torch.randn(2, 16, 16, 32, device=DEV)
    rp, ro = torch.randn(2, 16, 8, 16, device=DEV), torch.randn(2, 16, 16, 32, device=DEV)
    res = []
    for fused in (True, False):
        a, b = x2.clone().requires_grad_(True), xid.clone().requires_grad_(True)
        if fused:
            pooled, out = ops.res_tail_norm(a, b)
        else:
            pooled, out = ops.res_tail(ops.instance_norm(a, relu=True), ops.instance_norm(b))
        ((pooled * rp).sum() + (out * ro).sum()).backward()
        res.append((pooled.detach(), out.detach(), a.grad, b.grad))
    for u, v, what in zip(res[0], res[1], ("pooled", "out", "d x2", "d xid")):
        assert_close(u, v, 2e-6, what, atol=1e-7)


# --------------------------------------------------------------------------------------------------
# blocks against the reference's golden vectors
# --------------------------------------------------------------------------------------------------
def _run_block(golden, tag, mod, n_in, train=True, fwd_tol=1e-4, bwd_tol=1e-3, file="blocks.npz"):
    g = golden(file)
    sd = {k[2:]: v for k, v in g.group(tag).items() if k.startswith("P.")}
    mod.load_state_dict(sd, strict=True)
    mod.to(DEV).train(train)
    ins = [g.t("%s/in.%d" % (tag, i), DEV).requires_grad_(True) for i in range(n_in)]
    outs = mod(*ins)
    outs = outs if isinstance(outs, tuple) else (outs,)
    loss = None
    for i, o in enumerate(outs):
        t = (o * g.t("%s/R.%d" % (tag, i), DEV)).sum()
        loss = t if loss is None else loss + t
    loss.backward()
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        assert_close(o, g["%s/out.%d" % (tag, i)], fwd_tol, "%s out.%d" % (tag, i))
    for i, x in enumerate(ins):
        key = "%s/gin.%d" % (tag, i)
        if key in g.files:
            assert_close(x.grad, g[key], bwd_tol, key, atol=1e-6)
    for k, p in mod.named_parameters():
        assert_close(p.grad, g["%s/gP.%s" % (tag, k)], bwd_tol, "%s gP.%s" % (tag, k), atol=2e-5)
    for k, v in mod.state_dict().items():
        key = "%s/after.%s" % (tag, k)
        if key in g.files:
            assert_close(v.float(), g[key].astype(np.float32), 1e-5, key)


def test_blocks_golden(golden):
    from networks import blocks as B
    from networks.aspp import ASPP
    _run_block(golden, "double_conv", B.DoubleConv(16, 32), 1)
    _run_block(golden, "double_conv_odd", B.DoubleConv(3, 5), 1)
    _run_block(golden, "res_block", B.ResBlock(16, 32), 1)
    _run_block(golden, "res_block_c1", B.ResBlock(1, 16), 1)
    _run_block(golden, "up_block", B.UpBlock(32 + 16, 16), 2)
    _run_block(golden, "styled_denorm", B.StyledDenorm(16, 32), 2)
    _run_block(golden, "styled_denorm_eval", B.StyledDenorm(16, 16), 2, train=False)
    _run_block(golden, "styled_res_up", B.StyledResUpBlock(32, 16, 16, use_pixel_shuffle=False), 2)
    _run_block(golden, "aspp", ASPP(16, 16, [2, 6, 12, 18]), 1)


def test_styled_denorm_gamma_beta_forms(golden, monkeypatch):
    """mlp_gamma | mlp_beta as one concatenated conv (default) and as two convs both match the reference fixture;
    the concatenated form accumulates over two backward passes and into gradients it did not allocate."""
    from networks import blocks as B
    monkeypatch.setattr(B, "FUSE_GAMMA_BETA", False)
    _run_block(golden, "styled_denorm", B.StyledDenorm(16, 32), 2)
    _run_block(golden, "styled_res_up", B.StyledResUpBlock(32, 16, 16, use_pixel_shuffle=False), 2)
    monkeypatch.setattr(B, "FUSE_GAMMA_BETA", True)
    g = golden("blocks.npz")
    tag = "styled_denorm_eval"      # eval: no running-stat side effects, passes can be repeated
    sd = {k[2:]: v for k, v in g.group(tag).items() if k.startswith("P.")}

    def run(prealloc):
        mod = B.StyledDenorm(16, 16)
        mod.load_state_dict(sd, strict=True)
        mod.to(DEV).eval()
        if prealloc:
            for p in mod.parameters():
                p.grad = torch.zeros_like(p)
        ins = [g.t("%s/in.%d" % (tag, i), DEV) for i in range(2)]
        for _ in range(2):
            (mod(*ins) * g.t(tag + "/R.0", DEV)).sum().backward()
        torch.cuda.synchronize()
        return mod
    for prealloc in (False, True):
        mod = run(prealloc)
        for k, p in mod.named_parameters():
            assert_close(p.grad, 2.0 * g["%s/gP.%s" % (tag, k)], 1e-3, "%s x2 gP.%s prealloc=%s" % (tag, k, prealloc), atol=4e-5)
        if not prealloc:
            ga, gb_ = mod.mlp_gamma.weight.grad, mod.mlp_beta.weight.grad
            assert gb_.data_ptr() == ga.data_ptr() + 4 * ga.numel()      # halves of one buffer, second pass accumulated in place


@pytest.mark.parametrize("C,S", [(128, 32), (64, 64), (32, 64)])
def test_relu_mask_in_the_gamma_beta_input_gradient_epilogue(monkeypatch, C, S):
    """StyledDenorm's mlp_shared ReLU (blocks.py:63-66): the mask of its backward is applied in the epilogue of the gamma | beta
    convolution's input-gradient kernel (vqw_conv3x3_wino_fwd_masked) instead of a separate pass over the gradient.  Same
    values either way, so every gradient must be bit-equal to the run with the separate pass - and the fused launch must
    actually have been taken on these shapes (64-cout tile and the 2 x 32-cout tile)."""
    from networks import blocks as B
    from hipops import ops

    def run(fused):
        monkeypatch.setattr(ops, "FUSE_RELU_MASK", fused)
        torch.manual_seed(3)
        mod = B.StyledDenorm(C, C).to(DEV).train()
        x = torch.randn(2, C, S, S, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        style = torch.randn(2, C, S, S, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        r = torch.randn(2, C, S, S, device=DEV).contiguous(memory_format=torch.channels_last)
        n0 = ops.masked_dgrad_calls
        (mod(x, style) * r).sum().backward()
        torch.cuda.synchronize()
        grads = {"x": x.grad.clone(), "style": style.grad.clone()}
        grads.update({k: p.grad.clone() for k, p in mod.named_parameters()})
        return grads, ops.masked_dgrad_calls - n0
    ref, n_ref = run(False)
    got, n_got = run(True)
    assert n_ref == 0 and n_got == 1, (n_ref, n_got)
    assert not ops._MASKED_GRADS, "a masked gradient was announced and never consumed"
    for k in ref:
        assert torch.equal(ref[k], got[k]), "gradient %s differs between the fused and the separate ReLU mask" % k
    assert float(ref["style"].abs().max()) > 0


@pytest.mark.parametrize("which", ["res_block", "styled_res_up"])
def test_block_inputs_with_two_convolutions_sum_their_gradients_in_the_epilogue(monkeypatch, which):
    """ResBlock's input feeds the 1x1 branch and the first 3x3 convolution (blocks.py:14-36); a StyledResUpBlock's style input
    feeds the mlp_shared convolutions of both StyledDenorms (blocks.py:100-134).  Each pair is an ops.GradGroup: the second
    input gradient is added to the first in the Winograd kernel's epilogue (vqw_conv3x3_wino_fwd_acc; vqw_conv3x3_up2_dgrad_acc
    for the two up-sampled convolutions of a StyledResUpBlock's input) instead of by autograd's add pass.  a + b either way: every gradient bit-equal to the run with the groups off, and the accumulating
    launch actually taken."""
    from networks import blocks as B
    from hipops import ops

    def run(groups):
        monkeypatch.setattr(B, "GRAD_GROUP_BLOCKS", groups)
        torch.manual_seed(5)
        cl = lambda t: t.contiguous(memory_format=torch.channels_last)      # noqa: E731
        if which == "res_block":
            mod = B.ResBlock(64, 64).to(DEV).train()
            ins = [cl(torch.randn(2, 64, 64, 64, device=DEV)).requires_grad_(True)]
        else:
            mod = B.StyledResUpBlock(128, 64, 64).to(DEV).train()
            ins = [cl(torch.randn(2, 128, 32, 32, device=DEV)).requires_grad_(True),
                   cl(torch.randn(2, 64, 64, 64, device=DEV)).requires_grad_(True)]
        n0 = ops.group_acc_calls
        out = mod(*ins)
        outs = out if isinstance(out, tuple) else (out,)
        sum((o * torch.randn_like(o)).sum() for o in outs).backward()
        torch.cuda.synchronize()
        grads = {"in%d" % i: t.grad.clone() for i, t in enumerate(ins)}
        grads.update({k: p.grad.clone() for k, p in mod.named_parameters()})
        return grads, ops.group_acc_calls - n0
    ref, n_ref = run(False)
    got, n_got = run(True)
    # ResBlock: its input; StyledResUpBlock: the style input (two mlp_shared convolutions) and the up-sampled input (conv, conv1)
    assert n_ref == 0 and n_got == (1 if which == "res_block" else 2), (n_ref, n_got)
    for k in ref:
        assert torch.equal(ref[k], got[k]), "gradient %s differs between grouped and autograd-summed input gradients" % k


@pytest.mark.parametrize("C,S,B", [(64, 64, 2), (128, 32, 3), (32, 64, 2)])
def test_instance_norm_backward_sums_from_the_consumer_convolution(monkeypatch, C, S, B):
    """DoubleConv (blocks.py:39-61): the first InstanceNorm(+ReLU) feeds the second convolution only, so that convolution's
    input-gradient launch leaves the norm's backward sums (sum gm, sum gm * xhat) per region in its epilogue
    (vqw_conv3x3_wino_fwd_inbwd) and the norm's backward skips its reduction pass (vqw_inorm_bwd_parts).  Same sums in another
    order: gradients within 2e-5 of the run with the separate reduction (and of each other's scale), the fused route actually
    taken, nothing left in the registry."""
    from networks import blocks as B_
    from hipops import ops

    def run(fused):
        monkeypatch.setattr(ops, "FUSE_IN_BWD", fused)
        torch.manual_seed(11)
        mod = B_.DoubleConv(C, C).to(DEV).train()
        x = torch.randn(B, C, S, S, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        r = torch.randn(B, C, S, S, device=DEV).contiguous(memory_format=torch.channels_last)
        n0 = ops.in_bwd_fused_calls
        (mod(x) * r).sum().backward()
        torch.cuda.synchronize()
        grads = {"x": x.grad.clone()}
        grads.update({k: p.grad.clone() for k, p in mod.named_parameters()})
        return grads, ops.in_bwd_fused_calls - n0
    ref, n_ref = run(False)
    got, n_got = run(True)
    assert n_ref == 0 and n_got == 1, (n_ref, n_got)
    assert not ops._IN_BWD_PARTS
    gmax = max(float(v.abs().max()) for v in ref.values())
    for k in ref:
        if float(ref[k].abs().max()) < 1e-4 * gmax:      # biases in front of a norm: analytically zero, rounding noise only
            continue
        assert_close(got[k], ref[k], 2e-5, "gradient %s, norm-backward sums from the convolution's epilogue" % k)


def test_upsampled_layer_pair_as_one_launch(monkeypatch):
    """StyledResUpBlock's shortcut `conv` and `conv1` (blocks.py:100-112) read the same up-sampled input; at 32 couts each they
    run as ONE 64-cout launch of the nine-product kernel (ops.conv2d_up_pair, vqw_conv3x3_up2_fwd_pair) instead of two launches
    of the collapsed 4-tap form.  Outputs, input / style gradients and every parameter gradient agree with the block run layer
    by layer (VQW_UP_PAIR=0) to rounding; the paired launch is really taken, also without a gradient group."""
    from networks import blocks as B
    from hipops import ops
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)      # noqa: E731

    def run(pair, groups=True):
        monkeypatch.setattr(ops, "UP_PAIR", pair)
        monkeypatch.setattr(B, "GRAD_GROUP_BLOCKS", groups)
        torch.manual_seed(13)
        mod = B.StyledResUpBlock(64, 32, 32).to(DEV).train()
        down = cl(torch.randn(3, 64, 16, 32, device=DEV)).requires_grad_(True)
        skip = cl(torch.randn(3, 32, 32, 64, device=DEV)).requires_grad_(True)
        r = cl(torch.randn(3, 32, 32, 64, device=DEV))
        n0 = ops.up_pair_calls
        out = mod(down, skip)
        (out * r).sum().backward()
        torch.cuda.synchronize()
        grads = {"down": down.grad.clone(), "skip": skip.grad.clone()}
        grads.update({k: p.grad.clone() for k, p in mod.named_parameters()})
        return out.detach(), grads, ops.up_pair_calls - n0
    y0, g0, n0 = run(False)
    y1, g1, n1 = run(True)
    y2, g2, n2 = run(True, groups=False)
    assert n0 == 0 and n1 == 1 and n2 == 1, (n0, n1, n2)
    assert_close(y1, y0, 2e-5, "block output, paired launch vs layer by layer")
    gmax = max(float(v.abs().max()) for v in g0.values())
    for k in g0:
        if float(g0[k].abs().max()) < 1e-4 * gmax:      # a bias in front of a norm: analytically zero
            continue
        assert_close(g1[k], g0[k], 1e-4, "gradient %s, paired launch" % k)
        assert_close(g2[k], g0[k], 1e-4, "gradient %s, paired launch without gradient groups" % k)


def test_input_gradient_winograd_weights_straight_from_the_layer_weight():
    """vqw_conv3x3_wino_prepare_dgrad(w) == vqw_conv3x3_wino_prepare(vqw_pack_dgrad_weights(w)), bit for bit: the transformed
    weights of a layer's input-gradient convolution no longer need the packed copy (one launch less per 3x3 layer and step)."""
    import ctypes
    from hipops import _lib
    L = _lib.load()
    p = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for co, ci in [(32, 16), (64, 64), (24, 40), (128, 96)]:
        w = (torch.randn(co, ci, 3, 3, device=DEV) * 0.3).contiguous(memory_format=torch.channels_last)
        wt = torch.empty(ci * 9 * co, device=DEV)
        _lib.check(L.vqw_pack_dgrad_weights(p(w), p(wt), co, ci, 3, st), "pack")
        nb = L.vqw_conv3x3_wino_ws_bytes(co, ci)
        ua = torch.zeros(nb, dtype=torch.uint8, device=DEV)
        ub = torch.zeros(nb, dtype=torch.uint8, device=DEV)
        _lib.check(L.vqw_conv3x3_wino_prepare(p(wt), p(ua), nb, co, ci, st), "prepare")
        _lib.check(L.vqw_conv3x3_wino_prepare_dgrad(p(w), p(ub), nb, co, ci, st), "prepare_dgrad")
        torch.cuda.synchronize()
        assert torch.equal(ua, ub), (co, ci)


def test_dilation_two_layer_in_winograd_form_on_its_phase_images():
    """A 3x3 layer of dilation 2 (the pyramid's first branch, aspp.py:27-30) is four plain layers on the phase images of its
    tensors: forward with the following norm's statistics partials, input gradient alone and added to a gradient group's buffer,
    weight gradient - against an fp64 convolution (2e-5), and the statistics against the tensor's own (the norm built on them
    against F.instance_norm)."""
    from hipops import ops
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)      # noqa: E731
    for (N, C, K, H, W) in [(2, 32, 32, 16, 64), (1, 64, 32, 24, 128)]:
        g = torch.Generator().manual_seed(N * 100 + C)
        x = torch.randn(N, C, H, W, generator=g, dtype=torch.float64)
        w2 = torch.randn(K, C, 3, 3, generator=g, dtype=torch.float64) * 0.2
        w1 = torch.randn(K, C, 1, 1, generator=g, dtype=torch.float64) * 0.2
        r2 = torch.randn(N, K, H, W, generator=g, dtype=torch.float64)
        r1 = torch.randn(N, K, H, W, generator=g, dtype=torch.float64)
        xd = x.clone().requires_grad_(True); w2d = w2.clone().requires_grad_(True); w1d = w1.clone().requires_grad_(True)
        y2 = F.instance_norm(F.conv2d(xd, w2d, padding=2, dilation=2), eps=1e-5)
        y1 = F.conv2d(xd, w1d)
        ((y2 * r2).sum() + (y1 * r1).sum()).backward()
        dx = cl(x.float().to(DEV)).requires_grad_(True)
        dw2 = cl(w2.float().to(DEV)).requires_grad_(True)
        dw1 = cl(w1.float().to(DEV)).requires_grad_(True)
        grp = ops.GradGroup(2)
        with ops.winograd_forward():
            # the 1x1 member is created last: it runs first in backward and its gradient becomes the group's buffer
            raw, part = ops.conv2d(dx, dw2, None, dilation=2, want_stats=True, grad_group=grp)
            assert part is not None and ops._L().vqw_conv3x3_wino_dil2_supported(C, K, N, H, W) == 1
            z2 = ops.instance_norm(raw, relu=False, eps=1e-5, part=part)
            z1 = ops.conv2d(dx, dw1, None, grad_group=grp)
        ((z2 * r2.float().to(DEV)).sum() + (z1 * r1.float().to(DEV)).sum()).backward()
        torch.cuda.synchronize()
        tag = "dilation 2 on phase images %s" % ((N, C, K, H, W),)
        assert_close(z2, y2, 2e-5, tag + " norm(conv)")
        assert_close(dx.grad, xd.grad, 5e-5, tag + " dx (group of two)")
        assert_close(dw2.grad, w2d.grad, 5e-5, tag + " dw")
        assert_close(dw1.grad, w1d.grad, 2e-5, tag + " dw (1x1 member)")


def test_shortcut_norm_applied_inside_the_modulation_kernel(monkeypatch):
    """StyledResUpBlock's shortcut branch ends in InstanceNorm + ReLU and is added to the main path after norm2 (blocks.py:113-116,
    134).  Its normalisation runs inside norm2's modulation kernel from the statistics the shortcut convolution's epilogue left
    (vqw_spade_fwd_res_norm); the normalised tensor is never written.  Output and every gradient agree with the separate
    normalisation pass (VQW_RES_NORM_FUSED=0) to rounding, in both up-sampling modes and without the output activation."""
    from networks import blocks as B
    from hipops import ops
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)      # noqa: E731

    def run(fused, **kw):
        monkeypatch.setattr(ops, "RES_NORM_FUSED", fused)
        torch.manual_seed(29)
        mod = B.StyledResUpBlock(64, 32, 32, **kw).to(DEV).train()
        down = cl(torch.randn(2, 64, 16, 32, device=DEV)).requires_grad_(True)
        skip = cl(torch.randn(2, 32, 32, 64, device=DEV)).requires_grad_(True)
        r = cl(torch.randn(2, 32, 32, 64, device=DEV))
        out = mod(down, skip)
        (out * r).sum().backward()
        torch.cuda.synchronize()
        grads = {"down": down.grad.clone(), "skip": skip.grad.clone()}
        grads.update({k: p.grad.clone() for k, p in mod.named_parameters()})
        stats = {k: v.clone() for k, v in mod.state_dict().items() if "running" in k}
        return out.detach(), grads, stats
    for kw in (dict(), dict(use_output_act=False), dict(use_pixel_shuffle=True)):
        y0, g0, s0 = run(False, **kw)
        y1, g1, s1 = run(True, **kw)
        assert_close(y1, y0, 1e-6, "block output %s" % (kw,))
        gmax = max(float(v.abs().max()) for v in g0.values())
        for k in g0:
            if float(g0[k].abs().max()) < 1e-4 * gmax:
                continue
            assert_close(g1[k], g0[k], 2e-6, "gradient %s %s" % (k, kw))
        for k in s0:
            assert torch.equal(s0[k], s1[k]), k


def test_norm_applied_inside_the_residual_add():
    """ops.add_norm(a, x, part, relu) = a + instance_norm(x, relu) with the norm applied inside the add's kernel (the decoder's tail
    `x + conv_last(x)`, unet_decoder.py:169-171): output and the gradients of both operands agree with the two-kernel form to
    rounding."""
    from hipops import ops
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)      # noqa: E731
    for (N, C, H, W, relu) in [(2, 32, 24, 32, True), (1, 64, 16, 16, False)]:
        torch.manual_seed(31)
        a0 = cl(torch.randn(N, C, H, W, device=DEV)); x0 = cl(torch.randn(N, C, H, W, device=DEV) * 2 + 0.5)
        w = cl(torch.randn(C, C, 3, 3, device=DEV) * 0.1)
        r = cl(torch.randn(N, C, H, W, device=DEV))
        outs = []
        for fused in (False, True):
            a = a0.clone().requires_grad_(True); x = x0.clone().requires_grad_(True)
            with ops.winograd_forward():
                raw, part = ops.conv2d(x, w, None, want_stats=True)
            if fused:
                assert ops.add_norm_supported(a, raw)
                y = ops.add_norm(a, raw, part, relu=relu, eps=1e-5)
            else:
                y = ops.add(a, ops.instance_norm(raw, relu=relu, eps=1e-5, part=part))
            (y * r).sum().backward()
            torch.cuda.synchronize()
            outs.append((y.detach(), a.grad.clone(), x.grad.clone()))
        # (without the ReLU the compiler contracts `a + (x - mean) * rstd` into one fma: one rounding fewer than the two-kernel form)
        for k, name in enumerate(("y", "grad a", "grad x")):
            assert_close(outs[1][k], outs[0][k], 1e-6, "%s (%s)" % (name, (N, C, H, W, relu)))


def test_res_block_tail_backward_with_the_norm_sums_in_one_pass(monkeypatch):
    """The backward of a ResBlock's tail (ReLU of the branch sum + 2x2 max-pool, blocks.py:29-36) and the backward sums of the two
    InstanceNorms in front of it in ONE pass (vqw_res_tail_bwd_pair): input and parameter gradients agree with the two-kernel
    route (VQW_RES_TAIL_BWD_FUSED=0) to rounding (the sums are taken window by window instead of pixel by pixel), for both
    outputs used, the pooled one only, and a ragged channel count that keeps the separate kernels."""
    from networks import blocks as B
    from hipops import ops
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)      # noqa: E731

    def run(fused, cin, cout, H, W, use):
        monkeypatch.setattr(ops, "RES_TAIL_BWD_FUSED", fused)
        torch.manual_seed(37)
        mod = B.ResBlock(cin, cout).to(DEV).train()
        x = cl(torch.randn(2, cin, H, W, device=DEV)).requires_grad_(True)
        pooled, out = mod(x)
        rp = cl(torch.randn_like(pooled)); ro = cl(torch.randn_like(out))
        loss = (pooled * rp).sum() * (1.0 if "p" in use else 0.0) + (out * ro).sum() * (1.0 if "o" in use else 0.0)
        loss.backward()
        torch.cuda.synchronize()
        g = {"x": x.grad.clone()}
        g.update({k: p.grad.clone() for k, p in mod.named_parameters()})
        return g
    for (cin, cout, H, W, use) in [(16, 32, 24, 32, "po"), (32, 64, 16, 16, "p"), (8, 24, 12, 20, "po")]:
        a, b = run(False, cin, cout, H, W, use), run(True, cin, cout, H, W, use)
        gmax = max(float(v.abs().max()) for v in a.values())
        for k in a:
            if float(a[k].abs().max()) < 1e-4 * gmax:
                continue
            assert_close(b[k], a[k], 2e-6, "gradient %s %s" % (k, (cin, cout, H, W, use)))


def test_style_layer_pair_as_one_launch(monkeypatch):
    """The mlp_shared convolutions (+ReLU) of a StyledResUpBlock's two StyledDenorms read the same style tensor (blocks.py:72-75,
    100-134): inside ops.winograd_forward() they run as ONE launch of the 64-cout Winograd kernel on concatenated weights with a
    two-tensor epilogue (ops.conv2d_pair, vqw_conv3x3_wino_fwd_split).  Outputs and every gradient agree with the block run layer
    by layer (VQW_CONV_PAIR=0) to rounding; outside the scope (the Winograd forward not admitted) the pair is not taken."""
    from networks import blocks as B
    from hipops import ops
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)      # noqa: E731

    def run(pair, scope=True, groups=True, ch=32):
        monkeypatch.setattr(ops, "CONV_PAIR", pair)
        monkeypatch.setattr(B, "GRAD_GROUP_BLOCKS", groups)
        torch.manual_seed(17)
        mod = B.StyledResUpBlock(2 * ch, ch, ch).to(DEV).train()
        down = cl(torch.randn(2, 2 * ch, 20, 32, device=DEV)).requires_grad_(True)
        skip = cl(torch.randn(2, ch, 40, 64, device=DEV)).requires_grad_(True)
        r = cl(torch.randn(2, ch, 40, 64, device=DEV))
        n0 = ops.conv_pair_calls
        import contextlib
        with (ops.winograd_forward() if scope else contextlib.nullcontext()):
            out = mod(down, skip)
        (out * r).sum().backward()
        torch.cuda.synchronize()
        grads = {"down": down.grad.clone(), "skip": skip.grad.clone()}
        grads.update({k: p.grad.clone() for k, p in mod.named_parameters()})
        return out.detach(), grads, ops.conv_pair_calls - n0
    for ch in (32, 64):
        y0, g0, n0 = run(False, ch=ch)
        y1, g1, n1 = run(True, ch=ch)
        y2, g2, n2 = run(True, groups=False, ch=ch)
        assert n0 == 0 and n1 == 1 and n2 == 1, (n0, n1, n2)
        assert_close(y1, y0, 2e-5, "block output, paired style layers vs layer by layer")
        gmax = max(float(v.abs().max()) for v in g0.values())
        for k in g0:
            if float(g0[k].abs().max()) < 1e-4 * gmax:      # a bias in front of a norm: analytically zero
                continue
            assert_close(g1[k], g0[k], 1e-4, "gradient %s, paired style layers" % k)
            assert_close(g2[k], g0[k], 1e-4, "gradient %s, paired style layers without gradient groups" % k)
    _, _, n3 = run(True, scope=False)
    assert n3 == 0, "the pair is a Winograd-form forward: only where that form is admitted"


def test_conv_pair_against_fp64():
    """ops.conv2d_pair itself: both outputs, the input gradient and all four parameter gradients against an fp64 convolution
    (2e-5), on both workgroup shapes' geometries (32-wide regions, ragged height; 16-wide regions)."""
    from hipops import ops
    for (N, C, Ca, H, W, relu) in [(2, 32, 32, 22, 64, True), (1, 64, 48, 16, 32, False), (2, 32, 64, 16, 16, True)]:
        g = torch.Generator().manual_seed(N * 1000 + C + Ca)
        x = torch.randn(N, C, H, W, generator=g, dtype=torch.float64)
        ws = [torch.randn(Ca, C, 3, 3, generator=g, dtype=torch.float64) * 0.2 for _ in range(2)]
        bs = [torch.randn(Ca, generator=g, dtype=torch.float64) for _ in range(2)]
        rs = [torch.randn(N, Ca, H, W, generator=g, dtype=torch.float64) for _ in range(2)]
        leaves = [t.clone().requires_grad_(True) for t in [x] + ws + bs]
        ys = [F.conv2d(leaves[0], leaves[1 + i], leaves[3 + i], padding=1) for i in range(2)]
        if relu:
            ys = [torch.relu(y) for y in ys]
        sum((y * r).sum() for y, r in zip(ys, rs)).backward()
        dx = x.float().to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        dws = [w.float().to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True) for w in ws]
        dbs = [b.float().to(DEV).requires_grad_(True) for b in bs]
        with ops.winograd_forward():
            assert ops.conv2d_pair_supported(dx, dws[0], dws[1])
            ya, yb = ops.conv2d_pair(dx, dws[0], dbs[0], dws[1], dbs[1], relu=relu)
        ((ya * rs[0].float().to(DEV)).sum() + (yb * rs[1].float().to(DEV)).sum()).backward()
        torch.cuda.synchronize()
        tag = "pair %s" % ((N, C, Ca, H, W, relu),)
        assert_close(ya, ys[0], 2e-5, tag + " y_a")
        assert_close(yb, ys[1], 2e-5, tag + " y_b")
        assert_close(dx.grad, leaves[0].grad, 2e-5, tag + " dx")
        for i in range(2):
            assert_close(dws[i].grad, leaves[1 + i].grad, 2e-5, tag + " dw%d" % i)
            assert_close(dbs[i].grad, leaves[3 + i].grad, 2e-5, tag + " db%d" % i)


def test_two_source_input_gradient_leaves_the_epilogue_split(monkeypatch):
    """UpBlock's first convolution reads [nearest-up2x(down) | skip] (blocks.py:9-18): its input-gradient kernel writes the
    gradient of `down` (each 2 x 2 Winograd tile summed to one low-resolution pixel) and of `skip` from its epilogue
    (vqw_conv3x3_wino_fwd_split) instead of materialising the concatenated gradient and gathering it twice.  Both gradients
    agree with the gather route (VQW_SPLIT_DGRAD=0) to rounding; a channel total that is not a multiple of the kernel's cout tile
    (48) runs widened to 64 with zero weights; channel counts that are not multiples of 16 keep the gather route."""
    from hipops import ops
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)      # noqa: E731

    def run(split, N, H, W, C0, C1, up, Cout):
        monkeypatch.setattr(ops, "SPLIT_DGRAD", split)
        torch.manual_seed(23)
        hs, ws_ = (H // 2, W // 2) if up else (H, W)
        x0 = cl(torch.randn(N, C0, hs, ws_, device=DEV)).requires_grad_(True)
        x1 = cl(torch.randn(N, C1, H, W, device=DEV)).requires_grad_(True)
        w = cl(torch.randn(Cout, C0 + C1, 3, 3, device=DEV) * 0.1).requires_grad_(True)
        r = cl(torch.randn(N, Cout, H, W, device=DEV))
        n0 = ops.split_dgrad_calls
        y = ops.conv2d(x0, w, None, up2x=up, skip=x1)
        (y * r).sum().backward()
        torch.cuda.synchronize()
        return x0.grad.clone(), x1.grad.clone(), w.grad.clone(), ops.split_dgrad_calls - n0
    for shape, served in [((2, 20, 32, 64, 32, True, 32), True), ((1, 32, 16, 128, 64, True, 64), True),
                          ((2, 24, 64, 64, 32, False, 32), True), ((1, 22, 64, 256, 128, True, 128), True),
                          ((1, 16, 64, 32, 16, True, 64), True), ((2, 48, 32, 32, 16, True, 16), True),     # 48 channels: widened to 64
                          ((1, 16, 64, 40, 8, True, 32), False)]:
        a0, a1, aw, na = run(False, *shape)
        b0, b1, bw, nb = run(True, *shape)
        assert na == 0 and nb == (1 if served else 0), (shape, na, nb)
        assert_close(b0, a0, 2e-6, "gradient of the up-sampled source %s" % (shape,))
        assert_close(b1, a1, 2e-6, "gradient of the skip source %s" % (shape,))
        assert_close(bw, aw, 1e-6, "weight gradient %s" % (shape,))


def test_deferred_slab_folds_match_immediate_folds(monkeypatch):
    """The weight-gradient kernels' split-K slabs are folded by ONE launch at the end of the backward pass (ops._fold_flush,
    vqw_fold_flush_host) instead of by two short launches behind every weight-gradient kernel.  Same sums in another fixed
    order: every parameter gradient within 2e-6 of the run with immediate folds, over a block mix that covers the Winograd,
    tile, up-sampled, two-source, 1x1, concatenated gamma|beta and dilated weight-gradient kernels, with both views of a step
    (overwrite, then accumulate into the same gradient) and with a second backward pass accumulating on top."""
    from networks import blocks as B
    from networks.aspp import ASPP
    from hipops import ops, _lib

    def run(defer):
        monkeypatch.setattr(ops, "FOLD_DEFER", defer)
        torch.manual_seed(9)
        cl = lambda t: t.contiguous(memory_format=torch.channels_last)      # noqa: E731
        mods = [B.ResBlock(32, 64), B.StyledResUpBlock(64, 32, 32), B.UpBlock(96, 32), ASPP(32, 32, [2, 6]), B.DoubleConv(16, 16)]
        for m in mods:
            m.to(DEV).train()
        xa = [cl(torch.randn(2, 32, 64, 64, device=DEV, requires_grad=True)) for _ in range(2)]      # two "views"
        f0 = ops.fold_flushes
        for rep in range(2):               # the second pass accumulates into the gradients of the first
            loss = 0
            for x in xa:
                pooled, out = mods[0](x)                                   # (2, 64, 32, 32), (2, 64, 64, 64)
                up = mods[1](pooled, x)                                    # (2, 32, 64, 64), style = x
                u2 = mods[2](pooled, up)                                   # [up2x(64) | 32] -> 32
                a = mods[3](u2)                                            # (2, 96, 64, 64)
                d = mods[4](a[:, :16].contiguous(memory_format=torch.channels_last))
                loss = loss + (a * a).mean() + (d * d).mean() + (out * out).mean()
            loss.backward()
            assert _lib.load().vqw_fold_pending() == 0 and not ops._fold_keep, "folds left behind after the backward pass"
        torch.cuda.synchronize()
        grads = {"%d.%s" % (i, k): p.grad.clone() for i, m in enumerate(mods) for k, p in m.named_parameters()}
        return grads, ops.fold_flushes - f0
    ref, n_ref = run(False)
    got, n_got = run(True)
    assert n_ref == 0 and n_got == 2, (n_ref, n_got)
    gmax = max(float(v.abs().max()) for v in ref.values())
    for k in ref:
        assert_close(got[k], ref[k], 2e-6, "gradient %s: one batched fold vs immediate folds" % k, atol=1e-7 * gmax)


@pytest.mark.parametrize("second_first", [False, True])
def test_fusion_notes_are_not_honoured_when_the_activation_has_a_second_consumer(monkeypatch, second_first):
    """The two epilogue fusions pass a note from the consumer's backward to the producer's, keyed by the gradient tensor
    (ops._GradNotes).  The callers promise "one consumer" (norm_input=True / relu_input=True); this test breaks that promise
    on purpose.  With a second consumer autograd's InputBuffer sums the two gradients - IN PLACE into the first one to arrive
    when it owns it, so the sum can keep the annotated gradient's address (second_first=False: the convolution's gradient
    arrives first and is the one added into).  The note must then be ignored: gradients equal the run with both fusions off.
    (Round 3's address-only key took the note in exactly that case and returned a gradient without the second consumer's
    share in the norm's sums / skipped the ReLU mask on the summed gradient.)"""
    from hipops import ops
    C, S = 64, 32
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)      # noqa: E731
    torch.manual_seed(21)
    w0 = cl(torch.randn(C, C, 3, 3, device=DEV) * 0.05)
    w1 = cl(torch.randn(C, C, 3, 3, device=DEV) * 0.05)
    wa, wb = cl(torch.randn(C, C, 3, 3, device=DEV) * 0.05), cl(torch.randn(C, C, 3, 3, device=DEV) * 0.05)
    ba, bb = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    x = cl(torch.randn(2, C, S, S, device=DEV))
    r1, r2 = cl(torch.randn(2, C, S, S, device=DEV)), cl(torch.randn(2, C, S, S, device=DEV))
    r3 = cl(torch.randn(2, 2 * C, S, S, device=DEV))

    def run(fused):
        monkeypatch.setattr(ops, "FUSE_IN_BWD", fused)
        monkeypatch.setattr(ops, "FUSE_RELU_MASK", fused)
        out = {}
        # (1) a norm_input=True activation with a second consumer
        xs = x.clone(memory_format=torch.channels_last).requires_grad_(True)
        ws = [w.clone(memory_format=torch.channels_last).requires_grad_(True) for w in (w0, w1)]
        y, part = ops.conv2d(xs, ws[0], want_stats=True)
        a = ops.instance_norm(y, relu=True, part=part)
        if second_first:
            other = ops.add(a, r2)
            z = ops.conv2d(a, ws[1], norm_input=True)
        else:
            z = ops.conv2d(a, ws[1], norm_input=True)
            other = ops.add(a, r2)
        n0 = ops.in_bwd_fused_calls
        ((z * r1).sum() + (other * r2).sum()).backward()
        torch.cuda.synchronize()
        out["norm"] = (xs.grad.clone(), ws[0].grad.clone(), ws[1].grad.clone(), ops.in_bwd_fused_calls - n0)
        # (2) a relu_input=True activation with a second consumer
        xs = x.clone(memory_format=torch.channels_last).requires_grad_(True)
        w = w0.clone(memory_format=torch.channels_last).requires_grad_(True)
        if second_first:
            act = ops.conv2d(xs, w, relu=True)
            other = ops.add(act, r2)
            gb = ops.conv2d_cat(act, wa, ba, wb, bb, relu_input=True)
        else:
            act = ops.conv2d(xs, w, relu=True)
            gb = ops.conv2d_cat(act, wa, ba, wb, bb, relu_input=True)
            other = ops.add(act, r2)
        ((gb * r3).sum() + (other * r1).sum()).backward()
        torch.cuda.synchronize()
        out["relu"] = (xs.grad.clone(), w.grad.clone())
        return out
    ref = run(False)
    got = run(True)
    assert ref["norm"][3] == 0 and got["norm"][3] == 0, "the norm took its backward sums from a consumer that is not its only one"
    assert not ops._IN_BWD_PARTS and not ops._MASKED_GRADS, "notes must not outlive their backward pass"
    for k in ("norm", "relu"):
        for i, (u, v) in enumerate(zip(ref[k][:3], got[k][:3])):
            assert torch.equal(u, v), "%s case, gradient %d differs from the run with the fusions off" % (k, i)
    # the same wiring WITH the promise kept still takes the fused routes (the check above is not vacuous)
    monkeypatch.setattr(ops, "FUSE_IN_BWD", True)
    xs = x.clone(memory_format=torch.channels_last).requires_grad_(True)
    ws = [w.clone(memory_format=torch.channels_last).requires_grad_(True) for w in (w0, w1)]
    y, part = ops.conv2d(xs, ws[0], want_stats=True)
    n0 = ops.in_bwd_fused_calls
    (ops.conv2d(ops.instance_norm(y, relu=True, part=part), ws[1], norm_input=True) * r1).sum().backward()
    torch.cuda.synchronize()
    assert ops.in_bwd_fused_calls - n0 == 1


def test_grad_notes_key_on_address_version_and_pass():
    """ops._GradNotes in isolation: a note is taken only by the same tensor, unmodified, inside the backward pass that left it."""
    from hipops import ops
    notes = ops._GradNotes()
    seen = {}

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, mode):
            ctx.mode = mode
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            if ctx.mode == "put":
                out = g * 2.0
                notes.put(out, "note")
                seen["put"] = out.data_ptr()
                return out, None
            seen[ctx.mode] = notes.take(g)
            seen[ctx.mode + "_ptr"] = g.data_ptr()
            return g, None
    # one consumer: the producer's backward sees the annotated tensor itself
    x = torch.randn(8, device=DEV, requires_grad=True)
    Probe.apply(Probe.apply(x, "single"), "put").sum().backward()
    assert seen["single"] == "note" and seen["single_ptr"] == seen["put"] and len(notes) == 0
    # two consumers, the annotated gradient arrives first: summed in place (same address, version bumped) or out of place
    x = torch.randn(8, device=DEV, requires_grad=True)
    mid = Probe.apply(x, "double")
    other = mid * 3.0
    (Probe.apply(mid, "put").sum() + other.sum()).backward()
    assert seen["double"] is None, "a summed gradient took the note (address %s, annotated %s)" % (seen["double_ptr"], seen["put"])
    assert len(notes) == 0, "the end-of-pass callback drops what nobody took"
    # a note left outside / in another pass is not honoured
    t = torch.ones(4, device=DEV)
    notes.put(t, "stale")
    x = torch.randn(4, device=DEV, requires_grad=True)

    class Take(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            seen["other_pass"] = notes.take(t)
            return g
    Take.apply(x).sum().backward()
    assert seen["other_pass"] is None
    ops.begin_step()


# --------------------------------------------------------------------------------------------------
# vector quantiser
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["k10", "k64", "k1024"])
def test_vq_golden(golden, tag):
    from networks.vq import VQ
    g = golden("vq.npz")
    K, D = g[tag + "/embed0"].shape
    vq = VQ(emb_dim=D, dict_size=K, momentum=float(g[tag + "/momentum"]), eps=1e-5, knn_backend="torch")
    vq.embed.copy_(g.t(tag + "/embed0"))
    vq.embed_avg.copy_(g.t(tag + "/embed0").t())
    vq.to(DEV).train()
    for call in (1, 2):
        x = g.t("%s/x%d" % (tag, call), DEV).requires_grad_(True)
        q, commit, ids = vq(x)
        ((q * g.t("%s/R%d" % (tag, call), DEV)).sum() + 3.0 * commit).backward()
        ids = ids.transpose(1, 2).cpu().numpy()        # module returns the reference's (B,W,H) order
        gap = g["%s/gap%d" % (tag, call)]
        clear = gap > 1e-4 * (1 + np.abs(gap))
        assert np.array_equal(ids[clear], g["%s/ids%d" % (tag, call)][clear]), "ids differ on tie-free pixels"
        assert np.mean(ids == g["%s/ids%d" % (tag, call)]) > 0.999
        assert_close(q, g["%s/q%d" % (tag, call)], 1e-5 if call == 1 else 5e-5, "q")   # call 2 gathers the EMA-updated codebook
        assert_close(commit, g["%s/commit%d" % (tag, call)], 1e-5 if call == 1 else 5e-5, "commit")
        assert_close(x.grad, g["%s/gx%d" % (tag, call)], 1e-5 if call == 1 else 5e-5, "gx")
        for b in ("embed", "cluster_size", "embed_avg"):
            assert_close(getattr(vq, b), g["%s/%s_after%d" % (tag, b, call)], 5e-5, "%s after call %d" % (b, call))
        assert abs(float(vq.cluster_size.sum())) > 0
    vq.eval()
    before = vq.embed.clone()
    q, commit, ids = vq(g.t(tag + "/x_eval", DEV))
    assert torch.equal(before, vq.embed), "eval mode must not touch the codebook"
    assert np.mean(ids.transpose(1, 2).cpu().numpy() == g[tag + "/ids_eval"]) > 0.999
    assert_close(commit, g[tag + "/commit_eval"], 1e-4, "commit eval")
    look = vq.lookup(torch.from_numpy(g[tag + "/ids_eval"]).transpose(1, 2).contiguous().to(DEV))
    assert_close(look, g[tag + "/lookup_eval"], 1e-4, "lookup")


def test_vq_large_codebook_vs_oracle():
    """BASELINE config 4 shape class (K = 1024, D = 256): MFMA score GEMM + select path against the CPU oracle."""
    from oracle import vqwnet_ref as O
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    K, D, B, S = 1024, 256, 1, 48
    embed = torch.randn(K, D, generator=g)
    x = (torch.randn(B, D, S, S, generator=g) * 1.1).requires_grad_(True)
    V = dict(embed=embed.clone(), cluster_size=torch.full((K,), 3.0), embed_avg=(embed * 3.0).t().contiguous())
    q, commit, ids, gap = O.vq_forward(V, x, True, 0.99)
    r = torch.randn(B, D, S, S, generator=g)
    ((q * r).sum() + commit).backward()
    dx = x.detach().to(DEV).requires_grad_(True)
    e = embed.to(DEV).clone(); cs = torch.full((K,), 3.0, device=DEV); ea = (embed * 3.0).t().contiguous().to(DEV)
    dq, dc, dids = ops.vq_quantize(dx, e, cs, ea, True, 0.99, 1e-5)
    ((dq * r.to(DEV)).sum() + dc).backward()
    gp = gap.numpy()
    clear = gp > 1e-4 * (1 + np.abs(gp))
    assert clear.mean() > 0.98
    assert np.array_equal(dids.cpu().numpy()[clear], ids.numpy()[clear])
    assert_close(dc, commit, 1e-5, "commit")
    assert_close(dx.grad, x.grad, 1e-5, "gx")
    assert abs(float(cs.sum()) - float(V["cluster_size"].sum())) < 1e-3 * float(V["cluster_size"].sum())
    same = torch.from_numpy(dids.cpu().numpy() == ids.numpy()).all()
    if bool(same):
        assert_close(ea, V["embed_avg"], 2e-5, "embed_avg")
        assert_close(e, V["embed"], 2e-5, "embed")


VQ_ROUTES = [        # (K, D, B, S, expected plan): 0 LDS + matrix-core statistics, 1 LDS + sorted, 2 fused MFMA, 3 generic
    (10, 16, 3, 40, 0), (64, 32, 2, 24, 0), (32, 64, 1, 30, 0), (128, 16, 2, 20, 0),
    (180, 16, 2, 24, 1), (100, 32, 1, 20, 1),
    (1024, 256, 1, 34, 2), (1024, 64, 2, 20, 2), (96, 40, 1, 23, 2), (3000, 12, 1, 18, 2), (160, 128, 1, 17, 2), (40, 200, 1, 13, 2),
    (20, 20, 2, 12, 3), (50, 7, 1, 21, 3), (700, 30, 1, 16, 3),
]


@pytest.mark.parametrize("K,D,B,S,plan", VQ_ROUTES)
def test_vq_routes_vs_oracle(K, D, B, S, plan):
    """Every search / statistics route of vq.hip against the CPU oracle: ids bit-exact off ties, q / commit / gradient,
    the three EMA buffers after the update; then the same call again must give the same bits (no float atomics)."""
    from oracle import vqwnet_ref as O
    from hipops import _lib
    ops = _ops()
    assert _lib.load().vqw_vq_plan(D, K) == plan
    g = torch.Generator().manual_seed(K * 1000 + D)
    embed = torch.randn(K, D, generator=g)
    x = (torch.randn(B, D, S, S, generator=g) * 1.1).requires_grad_(True)
    V = dict(embed=embed.clone(), cluster_size=torch.full((K,), 3.0), embed_avg=(embed * 3.0).t().contiguous())
    q, commit, ids, gap = O.vq_forward(V, x, True, 0.9)
    r = torch.randn(B, D, S, S, generator=g)
    ((q * r).sum() + commit).backward()
    outs = []
    for rep in range(2):
        dx = x.detach().to(DEV).requires_grad_(True)
        e = embed.to(DEV).clone(); cs = torch.full((K,), 3.0, device=DEV); ea = (embed * 3.0).t().contiguous().to(DEV)
        dq, dc, dids = ops.vq_quantize(dx, e, cs, ea, True, 0.9, 1e-5)
        ((dq * r.to(DEV)).sum() + dc).backward()
        outs.append((dq.detach().clone(), dc.detach().clone(), dids.clone(), e, cs, ea, dx.grad.clone()))
    dq, dc, dids, e, cs, ea, gx = outs[0]
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b), "VQ is not bit-deterministic"
    gp = gap.numpy()
    clear = gp > 1e-4 * (1 + np.abs(gp))
    assert clear.mean() > 0.97
    assert np.array_equal(dids.cpu().numpy()[clear], ids.numpy()[clear])
    assert_close(dc, commit, 1e-5, "commit")
    assert_close(gx, x.grad, 1e-5, "gx")
    assert abs(float(cs.sum()) - float(V["cluster_size"].sum())) < 1e-4 * float(V["cluster_size"].sum())
    if bool((dids.cpu() == ids).all()):
        assert_close(dq, q.detach(), 1e-6, "q")
        assert_close(cs, V["cluster_size"], 1e-6, "cluster_size")
        assert_close(ea, V["embed_avg"], 2e-5, "embed_avg")
        assert_close(e, V["embed"], 2e-5, "embed")


@pytest.mark.parametrize("K,D", [(1024, 256), (10, 16), (180, 16), (50, 7)])
def test_vq_statistics_skewed_and_exact(K, D):
    """EMA statistics on integer-valued inputs are exact whatever the summation order: one code owning (almost) every
    pixel (the cold-start case: thousands of segment partials of one code), empty codes, a ragged pixel count."""
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    embed = torch.randint(-3, 4, (K, D), generator=g).float() * 8.0
    embed[0] = 0.0
    n = 3 * 61 * 61
    x = torch.randint(-2, 3, (3, D, 61, 61), generator=g).float()         # every pixel is nearest to code 0 ...
    pick = torch.randperm(n, generator=g)[:97]
    xf = x.permute(0, 2, 3, 1).reshape(n, D)
    xf[pick] = embed[torch.randint(1, K, (97,), generator=g)]           # ... except 97 that sit exactly on other codes
    x = xf.reshape(3, 61, 61, D).permute(0, 3, 1, 2).contiguous()
    e = embed.to(DEV).clone(); cs = torch.zeros(K, device=DEV); ea = torch.zeros(D, K, device=DEV)
    q, commit, ids = ops.vq_quantize(x.to(DEV), e, cs, ea, True, 0.0, 1e-5)       # momentum 0: buffers = raw statistics
    d2 = ((xf[:, None, :] - embed[None, :, :]) ** 2).sum(-1) if K * n * D < 5e8 else None
    ids_ref = d2.argmin(1) if d2 is not None else None
    idf = ids.reshape(-1).cpu()
    if ids_ref is not None:
        assert torch.equal(idf, ids_ref)
    counts = torch.bincount(idf, minlength=K).float()
    sums = torch.zeros(K, D).index_add_(0, idf, xf)
    assert torch.equal(cs.cpu(), counts)
    assert torch.equal(ea.cpu(), sums.t())
    assert int(counts[0]) >= n - 97 and int((counts == 0).sum()) >= K - 98 - 1


@pytest.mark.parametrize("K,D", [(1024, 256), (10, 16), (64, 32), (7, 5)])
def test_codebook_losses_vs_float64(K, D):
    """l_dist / l_reg (embed_loss.py:68-88) on the row-parallel kernel against the formula in double, incl. the i == j
    terms and codes closer than the margin."""
    ops = _ops()
    g = torch.Generator().manual_seed(K)
    cb = torch.randn(K, D, generator=g) * (0.05 if K > 100 else 0.5)      # many pairs inside 2 * margin
    cb[1] = cb[0]
    ld, lr = ops.codebook_losses(cb.to(DEV), 0.5)
    c = cb.double()
    dist = torch.cdist(c, c)
    ref_d = (torch.clamp(1.0 - dist, min=0) ** 2).sum() / (2 * K * (K - 1))
    ref_r = c.norm(dim=1).mean()
    assert_close(ld, ref_d, 2e-6, "l_dist")
    assert_close(lr, ref_r, 2e-6, "l_reg")
    ld2, _ = ops.codebook_losses(cb.to(DEV), 0.5)
    assert torch.equal(ld, ld2)


def test_vq_conservation_and_ties():
    """Properties: sum of counts == N pixels; all-equal scores resolve to the lowest index; ids in range."""
    ops = _ops()
    K, D = 10, 16
    embed = torch.randn(K, D, device=DEV)
    embed[3] = embed[7]                      # duplicate code -> exact tie, must pick 3
    x = embed[7][None, :, None, None].expand(2, D, 8, 8).contiguous()
    cs = torch.zeros(K, device=DEV); ea = embed.t().contiguous()
    q, commit, ids = ops.vq_quantize(x, embed.clone(), cs, ea, True, 0.5, 1e-5)
    assert int(ids.min()) == 3 and int(ids.max()) == 3
    assert abs(float(cs.sum()) - 0.5 * 2 * 8 * 8) < 1e-3
    assert float(commit) < 1e-12


# --------------------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------------------
def test_losses_golden(golden):
    from functions import EmbeddingLoss, OneHotEncoder
    g = golden("losses.npz")
    for tag, use_d in (("full", True), ("cross_only", None)):
        cb = g.t(tag + "/cb", DEV)
        K = cb.shape[1]
        oh = OneHotEncoder(K + 1)
        onehot1 = oh(g.t(tag + "/ids1", DEV).int())
        assert np.array_equal(onehot1.cpu().numpy(), g[tag + "/onehot1"])
        L = EmbeddingLoss(K, 0.5, use_d, use_d)
        for route in ("dense", "labels"):
            e1 = g.t(tag + "/e1", DEV).requires_grad_(True)
            e2 = g.t(tag + "/e2", DEV).requires_grad_(True)
            if route == "dense":
                r1 = onehot1[:, 1:].contiguous()
                r2 = oh(g.t(tag + "/ids2", DEV).int())[:, 1:].contiguous()
                lc, ld, lr = L(e1, r1, e2, r2, cb)
            else:
                lc, ld, lr = L.forward_labels(e1, g.t(tag + "/ids1", DEV).int(), e2, g.t(tag + "/ids2", DEV).int(), cb)
            lc.backward()
            assert_close(lc, g[tag + "/l_cross"], 1e-5, "l_cross " + route)
            assert_close(float(ld), g[tag + "/l_dist"], 1e-5, "l_dist")
            assert_close(float(lr), g[tag + "/l_reg"], 1e-5, "l_reg")
            assert_close(e1.grad, g[tag + "/ge1"], 1e-5, "ge1 " + route)
            assert_close(e2.grad, g[tag + "/ge2"], 1e-5, "ge2 " + route)


# --------------------------------------------------------------------------------------------------
# the training step
# --------------------------------------------------------------------------------------------------
def _hip_trainer(g):
    from trainers import FirstStepTrainer, FlipViews
    enc, dec = build_models(g.group("cfg"))
    check_init(g, enc, dec)
    sd = enc.state_dict()
    apply_warm_state(g, sd)
    cfg = step_cfg(g)
    tr = FirstStepTrainer(dict_size=cfg["dict_size"], momentum=cfg["momentum"], margin=cfg["margin"],
                          lr=cfg["optim"]["lr"], betas=cfg["optim"]["betas"], views=FlipViews(border=cfg["border"]),
                          encoder=enc, decoder=dec, device=DEV)
    return tr, cfg


@pytest.mark.parametrize("name", ["step_small.npz", "step_rcfg64_warm.npz", "step_rcfg64_warm_lr1e-6.npz", "step_rcfg32.npz", "step_cfg4_32.npz"])
def test_first_step_golden(golden, name):
    from oracle import vqwnet_ref as O
    g = golden(name)
    tr, cfg = _hip_trainer(g)
    # eval-mode forward + mask-guided reconstruction on the initial state (run_recon.py:179-194)
    tr.encoder.eval(); tr.decoder.eval()
    with torch.no_grad():
        q, _, ids = tr.encoder(g.t("eval/image", DEV))
        rec = tr.decoder(q)
        assert_ids_equal_where_clear(ids, g["eval/ids"], g["eval/gap"], "eval-mode ids (%s)" % name)
        print(name, "HIP eval recon rel %.2e" % rel_err(rec, g["eval/recon"]))
        assert_close(rec, g["eval/recon"], EVAL_RECON_TOL, "eval recon")
        from hipops import ops
        mask, ids0, scale = ops.mask_scale(g.t("recon/label_map", DEV))
        emb = ops.vq_lookup(ids0, tr.encoder.vq.embed, mask=mask, scale=scale)
        assert_close(emb, g["recon/embed"], 1e-6, "masked embed")
        print(name, "HIP mask-guided recon rel %.2e" % rel_err(tr.decoder(emb), g["recon/recon"]))
        assert_close(tr.decoder(emb), g["recon/recon"], EVAL_RECON_TOL, "mask-guided recon")
        from run_recon import reconstruct
        assert_close(reconstruct(tr.encoder, tr.decoder, g.t("recon/label_map", DEV)), g["recon/recon"], EVAL_RECON_TOL, "run_recon")
        e2 = tr.encoder.get_embed_from_ids(ids0)
        assert_close(e2 * mask[:, None].float() * scale, g["recon/embed"], 1e-6, "get_embed_from_ids")
    tr.encoder.train(); tr.decoder.train()
    gt, ml, lb = GRAD_TOL[name]
    lr = cfg["optim"]["lr"]
    for s in range(int(g["cfg/n_steps"])):
        out = tr.training_step({"image": g.t("step%d/image" % s, DEV)}, noise=g.t("step%d/noise" % s, DEV))
        torch.cuda.synchronize()
        sc = tr.scalars(out)
        rec = dict(sc)
        rec.update(ids_1=out["ids_1"], ids_2=out["ids_2"], recon_1=out["recon_1"], recon_2=out["recon_2"])
        rec["grads_enc"] = {k: p.grad for k, p in tr.encoder.named_parameters()}
        rec["grads_dec"] = {k: p.grad for k, p in tr.decoder.named_parameters()}
        PE = dict(tr.encoder.state_dict())
        PD = dict(tr.decoder.state_dict())
        if s > 0 and "step%d/spread.total" % s in g.files:
            # later steps against the reference's own multi-step output, within its own multi-step spread (real tolerances on
            # the lr = 1e-6 fixture, where the reference reproduces itself: test_oracle_golden.py::check_later_step)
            print(name, "HIP, step", s, {k: "%.2e" % v for k, v in check_later_step(g, s, rec, PE, PD, lr, what="HIP").items()})
            continue
        if s == 0:
            print(name, "HIP step 0 recon rel", ["%.2e" % rel_err(rec["recon_" + v], g["step0/recon_" + v]) for v in ("1", "2")])
        check_step(g, s, rec, PE, PD, lr, tight=(s == 0), tol=5e-4, grad_tol=max(gt, 5e-3), max_loose=ml + 2,
                   loose_bound=lb, recon_tol=STEP0_RECON_TOL)
        if s == 0:
            # the principled gate: at most twice as far from the reference's fp64 gradient as the reference's own fp32
            # evaluations are (tests/helpers.py::check_grads_vs_fp64)
            grads = {"enc." + k: v for k, v in rec["grads_enc"].items()}
            grads.update({"dec." + k: v for k, v in rec["grads_dec"].items()})
            print(name, "HIP grad error / reference fp32 spread (median, max):", check_grads_vs_fp64(g, grads, 2.0, "HIP"))


@pytest.mark.parametrize("B,S", [(2, 128), (3, 96), (2, 80)])
def test_step_vs_oracle_128(B, S):
    """R-cfg at 128x128 (halo / tile kernels on every level that is a multiple of 32 wide), 96x96 (mixed: the 48-, 24-,
    12- and 6-pixel levels fall back to the implicit-GEMM kernels, odd batch) and 80x80 (no level is a multiple of 32),
    warm VQ state: HIP step vs the CPU oracle on the same seeded inputs."""
    from oracle import vqwnet_ref as O
    from trainers import FirstStepTrainer, FlipViews
    from networks import UNetEncoder, UNetDecoder
    torch.manual_seed(3)
    K = 10
    enc = UNetEncoder(1, [16, 32, 64, 128, 256], K, 0.999, 'torch', False, 1, True)
    dec = UNetDecoder(16, 1, [32, 64, 128, 256, 512], use_dropblock=False, dropped_skip_layers=[], use_pixel_shuffle=False)
    with torch.no_grad():
        enc.vq.embed.mul_(0.7)
        enc.vq.cluster_size.fill_(B * S * S / K)
        enc.vq.embed_avg.copy_(enc.vq.embed.t() * enc.vq.cluster_size[None, :])
    PE = {k: v.detach().clone().contiguous() for k, v in enc.state_dict().items()}
    PD = {k: v.detach().clone().contiguous() for k, v in dec.state_dict().items()}
    PE0 = {k: v.clone() for k, v in PE.items()}
    PD0 = {k: v.clone() for k, v in PD.items()}
    cfg = dict(dict_size=K, margin=0.5, border=2, momentum=0.999,
               weights=dict(commit=1.0, cross=1.0, dist=1.0, reg=1.0, recon=1.0),
               optim=dict(lr=1e-4, betas=(0.5, 0.999), weight_decay=0.0))
    image, noise = O.synthetic_slices(B, S, 77)
    otr = O.FirstStepTrainer(PE, PD, cfg)
    ref = otr.step(image, noise)
    tr = FirstStepTrainer(dict_size=K, momentum=0.999, margin=0.5, views=FlipViews(border=2), encoder=enc, decoder=dec,
                          device=DEV)
    out = tr.training_step({"image": image.to(DEV)}, noise=noise.to(DEV))
    sc = tr.scalars(out)
    for k in ("total", "cross", "reg", "dist"):
        assert_close(sc[k], float(ref[k]), 5e-4, k)
    assert_close(sc["commit"], float(ref["commit"]), 5e-4, "commit")
    assert_close(sc["recon"], float(ref["recon"]), 5e-4, "recon")
    for v in ("1", "2"):
        assert_ids_equal_where_clear(out["ids_" + v], ref["ids_" + v], ref["gap_" + v], "ids_" + v)
        print("B%d S%d HIP recon_%s rel %.2e" % (B, S, v, rel_err(out["recon_" + v], ref["recon_" + v])))
        assert_close(out["recon_" + v], ref["recon_" + v], 5e-5, "recon_" + v)      # measured 1.6e-5 ... 1.9e-5
    # gradients: at most twice as far from the oracle's fp64 gradient as the oracle's own fp32 evaluations are (default
    # threads, one thread, batch reversed) - the gate of tests/helpers.py, here with the oracle as the reference
    def oracle_grads(dtype, threads, flip):
        n0 = torch.get_num_threads()
        torch.set_num_threads(threads)
        try:
            P1 = {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in PE0.items()}
            P2 = {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in PD0.items()}
            img, noi = (image.flip(0), noise.flip(0)) if flip else (image, noise)
            o = O.FirstStepTrainer(P1, P2, cfg).step(img.to(dtype), noi.to(dtype))
        finally:
            torch.set_num_threads(n0)
        g = {"enc." + k: v for k, v in o["grads_enc"].items()}
        g.update({"dec." + k: v for k, v in o["grads_dec"].items()})
        return g
    truth = {k: (v.double() if v is not None else None) for k, v in oracle_grads(torch.float64, torch.get_num_threads(), False).items()}
    first = {"enc." + k: v for k, v in ref["grads_enc"].items()}
    first.update({"dec." + k: v for k, v in ref["grads_dec"].items()})
    variants = [first, oracle_grads(torch.float32, 1, False), oracle_grads(torch.float32, torch.get_num_threads(), True)]
    if S <= 96:      # cheap here: sample the oracle's spread more densely (planes of 25-36 elements flip easily)
        variants += [oracle_grads(torch.float32, 2, False), oracle_grads(torch.float32, 4, True), oracle_grads(torch.float32, 1, True),
                     oracle_grads(torch.float32, 3, False), oracle_grads(torch.float32, 5, True)]
    test = {"enc." + k: p.grad for k, p in tr.encoder.named_parameters()}
    test.update({"dec." + k: p.grad for k, p in tr.decoder.named_parameters()})
    # Round 4: a parameter's spread is the MEDIAN over the oracle's fp32 evaluations (it was their maximum, which the
    # one-thread run dominates at 5-20x the others).  Against that far tighter scale the HIP gradients sit at a median ratio of
    # 1.1 / 1.6 / 1.6 (128 / 96 / 80) with the largest 2.2 / 2.5 / 3.9 - the per-parameter cap (6) and the median (2) hold with
    # room; what does not fit the old 10 % is the COUNT of parameters between 2x and 4x: 15 / 10 / 33 of 105.  They come in one
    # group: 80x80 has 5x5 planes at the bottom, and ONE ReLU / max-pool decision there that differs from the oracle's moves
    # the gradient of every layer upstream of it (the whole down path, a third of the parameters) together.  Allowed: 20 % at
    # 128 / 96, 35 % at 80 (the count is printed).
    print("B%d S%d HIP grad error / oracle fp32 spread (median, max):" % (B, S),
          grad_gate(truth, variants, test, 2.0, "HIP B%d S%d" % (B, S), max_over_frac=0.20 if S >= 96 else 0.35))


def test_full_size_properties():
    """BASELINE config 2 shapes (256x256, batch 32): properties that do not need the oracle at full size."""
    from trainers import FirstStepTrainer
    from oracle.vqwnet_ref import synthetic_slices
    torch.manual_seed(0)
    B, S, K = 32, 256, 10
    tr = FirstStepTrainer(device=DEV)
    image, noise = synthetic_slices(B, S, 1234)
    image, noise = image.to(DEV), noise.to(DEV)
    cs0 = tr.encoder.vq.cluster_size.clone()
    out = tr.training_step({"image": image}, noise=noise)
    sc = tr.scalars(out)
    assert all(np.isfinite(v) for v in sc.values()), sc
    for v in ("1", "2"):
        ids = out["ids_" + v]
        assert ids.shape == (B, S, S) and int(ids.min()) >= 1 and int(ids.max()) <= K
    # EMA conservation: cluster_size after two updates = m^2*cs0 + (1-m)*(m*N + N) with N = B*S*S pixels per view
    m, N = 0.999, B * S * S
    expect = m * m * float(cs0.sum()) + (1 - m) * (m * N + N)
    assert abs(float(tr.encoder.vq.cluster_size.sum()) - expect) <= 1e-3 * expect
    rec = out["recon_1"]
    assert float(rec.abs().max()) <= 1.0 and rec.shape == (B, 1, S, S)
    # quantised embeddings are exact codebook rows: every pixel equals one of <= K distinct vectors
    e = out["embed_1"].detach().permute(0, 2, 3, 1).reshape(-1, 16)
    assert torch.unique(e[: 65536], dim=0).shape[0] <= K
    for p in list(tr.encoder.parameters()) + list(tr.decoder.parameters()):
        assert p.grad is not None and torch.isfinite(p.grad).all()


def test_config4_full_size_properties():
    """BASELINE config 4 as SURVEY 8d makes it concrete (512x512, dict_size 1024, enc_filters [256,64,128,256,512] so that
    emb_dim = 256, batch 2 per GPU): one whole training step at full size; properties that need no oracle."""
    from trainers import FirstStepTrainer
    from oracle.vqwnet_ref import synthetic_slices
    from hipops import _lib
    torch.manual_seed(0)
    B, S, K, D = 2, 512, 1024, 256
    assert _lib.load().vqw_vq_plan(D, K) == 2          # fused MFMA score GEMM / arg-max route
    tr = FirstStepTrainer(enc_filters=(256, 64, 128, 256, 512), dec_filters=(32, 64, 128, 256, 512), dict_size=K, momentum=0.99,
                          device=DEV)
    with torch.no_grad():      # checkpoint-like VQ state (every code in use)
        tr.encoder.vq.cluster_size.fill_(B * S * S / K)
        tr.encoder.vq.embed_avg.copy_(tr.encoder.vq.embed.t() * tr.encoder.vq.cluster_size[None, :])
    image, noise = synthetic_slices(B, S, 1234)
    cs0 = tr.encoder.vq.cluster_size.clone()
    e0 = tr.encoder.vq.embed.clone()
    with torch.no_grad():      # the features view 1 is quantised from (before the optimiser moves the weights)
        feat = tr.encoder.feature_extraction(image.to(DEV)).clone()
    out = tr.training_step({"image": image.to(DEV)}, noise=noise.to(DEV))
    sc = tr.scalars(out)
    assert all(np.isfinite(v) for v in sc.values()), sc
    for v in ("1", "2"):
        ids = out["ids_" + v]
        assert ids.shape == (B, S, S) and int(ids.min()) >= 1 and int(ids.max()) <= K
    m, N = 0.99, B * S * S
    expect = m * m * float(cs0.sum()) + (1 - m) * (m * N + N)
    assert abs(float(tr.encoder.vq.cluster_size.sum()) - expect) <= 1e-4 * expect
    # view 1 was quantised with the initial codebook: every pixel of embed_1 is exactly the row its id names, and that row
    # is the nearest one (checked in double on a sample of pixels against all 1024 codes)
    e1 = out["embed_1"].detach().permute(0, 2, 3, 1).reshape(-1, D)
    id1 = (out["ids_1"].reshape(-1) - 1)
    pick = torch.randint(0, e1.shape[0], (4096,), device=DEV)
    assert torch.equal(e1[pick], e0[id1[pick]])
    f = feat.permute(0, 2, 3, 1).reshape(-1, D)[pick].double()
    d2 = (f * f).sum(1, keepdim=True) - 2 * f @ e0.double().t() + (e0.double() ** 2).sum(1)[None]
    best = d2.argmin(1)
    mine = d2.gather(1, id1[pick][:, None])[:, 0]
    assert bool(((mine - d2.min(1).values) <= 1e-4 * (1 + d2.min(1).values.abs())).all()), "ids are not the nearest codes"
    assert float((best == id1[pick]).float().mean()) > 0.995
    assert float(out["recon_1"].abs().max()) <= 1.0
    for p in list(tr.encoder.parameters()) + list(tr.decoder.parameters()):
        assert p.grad is not None and torch.isfinite(p.grad).all()


def test_config5_full_size_recon_properties():
    """BASELINE config 5 (run_recon.inner at 256x256, batch 64, eval mode, 10 % masked pixels): full-size properties -
    deterministic, batch-independent (eval-mode normalisation uses running / per-sample statistics only), masked
    pixels enter as zeros and the rescale is numel / count."""
    from networks import UNetEncoder, UNetDecoder
    from run_recon import reconstruct
    from hipops import ops
    torch.manual_seed(0)
    enc = UNetEncoder(1, [16, 32, 64, 128, 256], 10, 0.999, 'torch', False, 1, True).to(DEV).eval()
    dec = UNetDecoder(16, 1, [32, 64, 128, 256, 512], use_dropblock=False, dropped_skip_layers=[], use_pixel_shuffle=False).to(DEV).eval()
    B, S = 64, 256
    g = torch.Generator().manual_seed(1)
    lab = torch.randint(1, 11, (B, S, S), generator=g)
    lab[torch.rand(B, S, S, generator=g) < 0.1] = 0
    lab = lab.to(DEV)
    with torch.no_grad():
        rec = reconstruct(enc, dec, lab)
        assert rec.shape == (B, 1, S, S) and torch.isfinite(rec).all() and float(rec.abs().max()) <= 1.0
        assert torch.equal(rec, reconstruct(enc, dec, lab))
        # the rescale couples samples only through the scalar numel / count: with it fixed, a sample's output does not
        # depend on its batch mates
        mask, ids0, scale = ops.mask_scale(lab)
        assert abs(float(scale) - lab.numel() / float((lab != 0).sum())) < 1e-3
        emb = ops.vq_lookup(ids0, enc.vq.embed, mask=mask, scale=scale)
        assert float(emb.permute(0, 2, 3, 1)[lab == 0].abs().max()) == 0.0
        whole = dec(emb)
        part = dec(emb[5:9].contiguous(memory_format=torch.channels_last))
        assert_close(part, whole[5:9], 1e-5, "batch independence")
        assert_close(whole, rec, 1e-6, "reconstruct == lookup + decoder")


def test_norm_denorm_golden(golden):
    """utils.norm / denorm (utils/__init__.py:81-92) against the reference's own outputs, incl. the in-place contract."""
    import utils as U
    g = golden("utils.npz")
    x = g.t("norm/x", DEV)
    y = U.norm(x)
    assert y.data_ptr() == x.data_ptr()
    assert_close(y, g["norm/y"], 1e-7, "norm")
    assert_close(U.denorm(g.t("norm/x", DEV), 0.0, 1.0), g["denorm/y01"], 1e-6, "denorm [0,1]")
    assert_close(U.denorm(g.t("norm/x", DEV), -1300.0, 200.0), g["denorm/y_hu"], 1e-6, "denorm HU")


# --------------------------------------------------------------------------------------------------
# optional paths
# --------------------------------------------------------------------------------------------------
def test_extras_golden(golden):
    from networks import blocks as B, UNetDecoder, VQWNet
    from networks.dropblock import DropBlock2D
    from functions import SoftDiceLoss, FocalLoss
    from helpers import checksum
    g = golden("extras.npz")
    # pixel-shuffle up block (reuses the block runner on the extras file)
    sd = {k[2:]: v for k, v in g.group("styled_res_up_ps").items() if k.startswith("P.")}
    mod = B.StyledResUpBlock(32, 16, 16, use_pixel_shuffle=True)
    mod.load_state_dict(sd, strict=True)
    mod.to(DEV).train()
    ins = [g.t("styled_res_up_ps/in.%d" % i, DEV).requires_grad_(True) for i in range(2)]
    out = mod(*ins)
    (out * g.t("styled_res_up_ps/R.0", DEV)).sum().backward()
    assert_close(out, g["styled_res_up_ps/out.0"], 1e-4, "ps block out")
    for i in range(2):
        assert_close(ins[i].grad, g["styled_res_up_ps/gin.%d" % i], 1e-3, "ps block gin.%d" % i, atol=1e-6)
    for k, p in mod.named_parameters():
        assert_close(p.grad, g["styled_res_up_ps/gP." + k], 1e-3, "ps block gP." + k, atol=2e-5)
    # decoder with constructor defaults (use_pixel_shuffle=True)
    torch.manual_seed(32)
    dec = UNetDecoder(16, 1, [16, 32, 32, 32, 64], use_dropblock=False, dropped_skip_layers=[]).to(DEV).train()
    x = g.t("dec_ps/x", DEV).requires_grad_(True)
    y = dec(x)
    (y * g.t("dec_ps/R", DEV)).sum().backward()
    assert_close(y, g["dec_ps/y"], 2e-4, "decoder (pixel shuffle) y")
    # conditioning: at 32x32 the bottom levels normalise 2x2 / 4x4 planes, and the reference's own fp32 result is 5.3e-3
    # (relative L2) away from the same computation in fp64 (measured with the reference modules on CPU, one and eight
    # threads agree to 1e-6).  Two fp32 implementations are therefore expected to differ by that order; the forward
    # value above is the tight gate.
    assert_close(x.grad, g["dec_ps/gx"], 1.5e-2, "decoder (pixel shuffle) gx")
    gmax = max(float(g[k]) for k in g.files if k.startswith("dec_ps/gnorm."))
    for k, p in dec.named_parameters():
        ref = float(g["dec_ps/gnorm." + k])
        if ref > 1e-5 * gmax:
            assert abs(float(p.grad.norm()) - ref) <= 5e-3 * ref, "dec_ps grad norm " + k
    # VQWNet monolith
    torch.manual_seed(33)
    net = VQWNet(1, 1, [16, 16, 32, 32, 32], dict_size=6)
    with torch.no_grad():
        net.vq.embed.mul_(0.7)
        net.vq.cluster_size.fill_(2 * 32 * 32 / 6)
        net.vq.embed_avg.copy_(net.vq.embed.t() * net.vq.cluster_size[None, :])
    net.to(DEV).train()
    o = net(g.t("vqwnet/image", DEV))
    ((o["recon"] * g.t("vqwnet/R", DEV)).sum() + o["commit_loss"]).backward()
    assert np.mean(o["ids"].cpu().numpy() == g["vqwnet/ids"]) > 0.999
    assert_close(o["recon"], g["vqwnet/recon"], 2e-4, "vqwnet recon")
    assert_close(o["embed"], g["vqwnet/embed"], 2e-4, "vqwnet embed")
    assert_close(o["commit_loss"], g["vqwnet/commit"], 1e-4, "vqwnet commit")
    for b in ("embed", "cluster_size", "embed_avg"):
        assert_close(getattr(net.vq, b), g["vqwnet/after.vq." + b], 1e-4, "vqwnet vq." + b)
    gmax = max(float(g[k]) for k in g.files if k.startswith("vqwnet/gnorm."))
    for k, p in net.named_parameters():
        ref = float(g["vqwnet/gnorm." + k])
        if ref > 1e-5 * gmax:
            assert abs(float(p.grad.norm()) - ref) <= 1e-2 * ref, "vqwnet grad norm " + k
    gen = net.generate_images_from_ids(torch.from_numpy(g["vqwnet/ids"]).to(DEV) - 1)
    assert_close(gen["recon"], g["vqwnet/gen_recon"], 5e-4, "vqwnet generate_images_from_ids")
    # DropBlock
    db = DropBlock2D(drop_prob=0.3, block_size=4).train()
    xx = g.t("dropblock/x", DEV).requires_grad_(True)
    yy = db.apply_seed_mask(xx, g.t("dropblock/seed", DEV))
    yy.sum().backward()
    assert_close(yy, g["dropblock/y"], 1e-6, "dropblock apply")
    assert_close(xx.grad, g["dropblock/y"] / np.where(g["dropblock/x"] == 0, 1, g["dropblock/x"]), 1e-5, "dropblock grad")
    gl = golden("losses.npz")
    for bs, key in ((4, "keep4"), (5, "keep5")):
        keep, _ = _ops().dropblock_mask(gl.t("dropblock/seed4", DEV), bs)
        assert np.array_equal(keep.cpu().numpy(), gl["dropblock/" + key])
    # segmentation losses
    tgt = gl.t("seg/target", DEV)
    for name, mod in (("dice", SoftDiceLoss()), ("dice_ign", SoftDiceLoss(ignore_index=0)), ("focal", FocalLoss())):
        z = gl.t("seg/logits", DEV).requires_grad_(True)
        l = mod(z, tgt)
        l.backward()
        assert_close(l, gl["seg/" + name], 1e-5, name)
        assert_close(z.grad, gl["seg/g_" + name], 1e-4, "g_" + name)


# --------------------------------------------------------------------------------------------------
# edge cases
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,S", [(1, 16), (3, 32), (1, 64), (2, 96), (1, 160)])
def test_edge_sizes_forward_vs_oracle(B, S):
    """Batch 1 / odd batch and the smallest legal map (16x16: the deepest level is 1x1, InstanceNorm over one pixel
    gives exactly 0): train-mode forward of encoder + decoder against the oracle."""
    from oracle import vqwnet_ref as O
    from networks import UNetEncoder, UNetDecoder
    torch.manual_seed(9)
    K = 6
    enc = UNetEncoder(1, [16, 16, 32, 32, 32], K, 0.9, 'torch', False, 1, True)
    dec = UNetDecoder(16, 1, [16, 32, 32, 32, 64], use_dropblock=False, dropped_skip_layers=[1], use_pixel_shuffle=False)
    with torch.no_grad():
        enc.vq.embed.mul_(0.5)
        enc.vq.cluster_size.fill_(B * S * S / K)
        enc.vq.embed_avg.copy_(enc.vq.embed.t() * enc.vq.cluster_size[None, :])
    PE = {k: v.detach().clone().contiguous() for k, v in enc.state_dict().items()}
    PD = {k: v.detach().clone().contiguous() for k, v in dec.state_dict().items()}
    img, _ = O.synthetic_slices(B, S, 21)
    with torch.no_grad():
        qr, cr, idr, gap = O.encoder_forward(PE, img, True, 0.9)
        rr = O.decoder_forward(PD, qr, True, dropped_skip_layers=(1,))
    enc.to(DEV).train(); dec.to(DEV).train()
    with torch.no_grad():
        q, c, ids = enc(img.to(DEV))
        rec = dec(q)
    clear = gap.numpy() > 1e-3 * (1 + np.abs(gap.numpy()))
    assert np.array_equal(ids.cpu().numpy()[clear], idr.numpy()[clear])
    assert torch.isfinite(rec).all()
    if np.array_equal(ids.cpu().numpy(), idr.numpy()):
        assert_close(c, cr, 1e-4, "commit")
        assert_close(rec, rr, 2e-3, "recon", atol=1e-5)


def test_rejects_bad_arguments():
    """Shape / dtype / layout violations surface as RuntimeError before anything is launched."""
    ops = _ops()
    x = torch.randn(1, 16, 8, 8, device=DEV)
    w = torch.randn(8, 12, 3, 3, device=DEV)
    with pytest.raises(RuntimeError, match="do not match"):
        ops.conv2d(x, w)
    with pytest.raises(RuntimeError, match="fp32"):
        ops.instance_norm(x.double())
    from networks.vq import VQ
    vq = VQ(16, 10, 0.99, 1e-5, "torch").to(DEV)
    with pytest.raises(RuntimeError, match="square"):
        vq(torch.randn(1, 16, 8, 4, device=DEV))
    with pytest.raises(RuntimeError, match="does not match"):
        ops.conv2d(torch.randn(1, 16, 3, 3, device=DEV), torch.randn(8, 32, 3, 3, device=DEV), up2x=True,
                   skip=torch.randn(1, 16, 7, 7, device=DEV))
    with pytest.raises(RuntimeError, match="vqw_maxpool2_fwd"):      # rejected by the C ABI's own argument check
        ops.maxpool2(torch.randn(1, 16, 1, 8, device=DEV))


# --------------------------------------------------------------------------------------------------
# two-view augmentation + id-map warps (oracle/augment_ref.py defines the arithmetic, kornia absent)
# --------------------------------------------------------------------------------------------------
def _rand_mats(B, H, W, seed):
    from oracle import augment_ref as A
    rng = np.random.default_rng(seed)
    mats = []
    for b in range(B):
        kind = b % 4
        if kind == 0:
            mats.append(A.identity_matrix())
        elif kind == 1:
            mats.append(A.hflip_matrix(W))
        elif kind == 2:
            mats.append(A.affine_matrix(0.0, float(rng.integers(-5, 6)), float(rng.integers(-5, 6)), 0.0, 0.0, H, W))
        else:
            mats.append(A.affine_matrix(rng.uniform(-30, 30), rng.uniform(-6, 6), rng.uniform(-6, 6), rng.uniform(-10, 10), 0.0, H, W))
    return np.stack(mats)


@pytest.mark.parametrize("shape", [(8, 32, 32), (5, 40, 24), (4, 256, 256)])
def test_augment_warps_match_oracle(shape):
    from oracle import augment_ref as A
    ops = _ops()
    B, H, W = shape
    rng = np.random.default_rng(7)
    fwd = _rand_mats(B, H, W, 11)
    minv = np.stack([A.dst_to_src(m) for m in fwd])
    img = rng.random((B, 1, H, W), dtype=np.float32)
    ids = rng.integers(1, 11, size=(B, H, W)).astype(np.int64)
    y = ops.warp_image(torch.from_numpy(img).to(DEV), torch.from_numpy(minv).to(DEV))
    assert_close(y, A.warp_image(img, minv), 2e-6, "warp_image", atol=2e-6)
    for dt in (torch.int64, torch.int32):
        w = ops.warp_labels(torch.from_numpy(ids).to(DEV).to(dt), torch.from_numpy(minv).to(DEV))
        assert w.dtype == torch.int32
        assert np.array_equal(w.cpu().numpy(), A.warp_labels(ids, minv)), "warp_labels %s" % dt
    # degenerate matrix (w = 0 row): everything out of frame
    bad = np.zeros((B, 3, 3), dtype=np.float32)
    assert int(ops.warp_labels(torch.from_numpy(ids).to(DEV), torch.from_numpy(bad).to(DEV)).abs().sum()) == 0


def test_augment_photometric_and_blur_match_oracle():
    from oracle import augment_ref as A
    ops = _ops()
    rng = np.random.default_rng(3)
    B, H, W = 6, 48, 40
    x = rng.random((B, 1, H, W), dtype=np.float32)
    noise = rng.standard_normal((B, 1, H, W)).astype(np.float32)
    params = np.array([[0.0, 1.0, 8, 0.0], [0.2, 1.0, 8, 0.0], [-0.3, 1.4, 8, 0.0], [0.0, 0.6, 3, 0.0], [0.1, 1.1, 5, 0.05],
                       [0.0, 1.0, 1, 0.2]], dtype=np.float32)
    y = ops.photometric(torch.from_numpy(x).to(DEV), torch.from_numpy(params).to(DEV), torch.from_numpy(noise).to(DEV))
    assert_close(y, A.photometric(x, params, noise), 1e-6, "photometric", atol=1e-6)
    y0 = ops.photometric(torch.from_numpy(x).to(DEV), torch.from_numpy(params[:, :]).to(DEV), None)
    assert_close(y0, A.photometric(x, params, None), 1e-6, "photometric (no noise)", atol=1e-6)
    for k, sigma in ((3, 1.0), (5, 1.5), (9, 2.0)):
        taps = A.gaussian_taps(k, sigma)
        apply = np.array([1, 0, 1, 1, 0, 1], dtype=np.uint8)
        yb = ops.gauss_blur(torch.from_numpy(x).to(DEV), torch.from_numpy(taps).to(DEV), torch.from_numpy(apply).to(DEV))
        assert_close(yb, A.gauss_blur(x, taps, apply), 2e-6, "gauss_blur k=%d" % k, atol=2e-6)
    with pytest.raises(RuntimeError, match="vqw_gauss_blur"):
        ops.gauss_blur(torch.from_numpy(x).to(DEV), torch.ones(4, device=DEV) / 4)          # even kernel size


def test_random_transform_module_properties():
    """RandomTransform: reverse then forward of the SAME transform is the identity away from the border, flips and
    whole-pixel shifts are exact, the module equals the oracle driven with the matrices it sampled, and the
    trainer runs a step with the reference's two-transform view pair."""
    from oracle import augment_ref as A
    from networks import RandomTransform
    from trainers import FirstStepTrainer, RandomTransformViews
    cfg = dict(modules=["RandomHorizontalFlip", "RandomAffine", "ColorJitter", "RandomGaussianBlur", "RandomPosterize",
                        "RandomGaussianNoise"],
               RandomHorizontalFlip=dict(p=0.5), RandomAffine=dict(degrees=20.0, translate=(0.1, 0.1), shear=8.0, p=0.8),
               ColorJitter=dict(brightness=0.2, contrast=0.2, saturation=0.0, hue=0.0, p=0.8),
               RandomGaussianBlur=dict(kernel=5, sigma=1.2, p=0.5), RandomPosterize=dict(bits=4, p=0.3),
               RandomGaussianNoise=dict(std=0.05, p=0.5))
    B, H, W = 8, 64, 64
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, 1, H, W, generator=g).to(DEV)
    t = RandomTransform(cfg, seed=123)
    aug, clear = t(x)
    assert aug.shape == x.shape and clear.shape == x.shape and len(t._transforms) == 2
    mats = [m.numpy() for m in t._transforms]
    # geometric part against the oracle with the sampled matrices
    ref = x.cpu().numpy()
    for m in mats:
        ref = A.warp_image(ref, np.stack([A.dst_to_src(m[b]) for b in range(B)]))
    assert_close(clear, ref, 5e-6, "clear view", atol=5e-6)
    # id maps: forward / reverse against the oracle, and reverse∘forward = identity on the interior
    ids = torch.randint(1, 11, (B, H, W), generator=g).to(DEV)
    f = t.forward_transform(ids)
    assert np.array_equal(f.cpu().numpy(), A.forward_transform(ids.cpu().numpy(), mats))
    r = t.reverse_transform(f)
    assert np.array_equal(r.cpu().numpy(), A.reverse_transform(f.cpu().numpy(), mats))
    smooth = (torch.arange(H).view(1, H, 1) // 8 * 8 + torch.arange(W).view(1, 1, W) // 8 + 1).expand(B, H, W).contiguous().to(DEV)
    back = t.reverse_transform(t.forward_transform(smooth))
    inside = back != 0
    assert float(inside.float().mean()) > 0.5
    assert float((back[inside] == smooth[inside].int()).float().mean()) > 0.9       # piecewise-constant map survives two nearest warps
    # exact transforms: flip only
    tf = RandomTransform(dict(modules=["RandomHorizontalFlip"], RandomHorizontalFlip=dict(p=1.0)), seed=1)
    a2, c2 = tf(x)
    assert torch.equal(a2, torch.flip(x, dims=[3])) and torch.equal(c2, a2)
    assert torch.equal(tf.forward_transform(ids), torch.flip(ids, dims=[2]).int())
    assert torch.equal(tf.reverse_transform(tf.forward_transform(ids)), ids.int())
    # one training step with the reference's view pair
    tr = FirstStepTrainer(enc_filters=(4, 8, 8, 8, 8), dec_filters=(4, 8, 8, 8, 8), device=DEV,
                          views=RandomTransformViews(RandomTransform(cfg, seed=1), RandomTransform(cfg, seed=2)))
    img = torch.rand(2, 1, 32, 32, generator=g).to(DEV) * 2 - 1
    keep = img.clone()
    out = tr.training_step({"image": img})
    assert torch.equal(img, keep)                                    # the batch is not modified in place
    assert np.isfinite(float(out["total"].detach()))


# --------------------------------------------------------------------------------------------------
# second training step: PatchGAN discriminator, hinge / generator losses, trainer
# --------------------------------------------------------------------------------------------------
def test_discriminator_golden(golden):
    """networks.NLayerDiscriminator against the reference module's outputs, input / parameter gradients and updated
    BatchNorm buffers (train mode) and against its eval-mode pass."""
    from networks import NLayerDiscriminator
    _run_block(golden, "dis_f16", NLayerDiscriminator(1, 1, n_filters=16, n_layers=3), 1, file="gan.npz")
    _run_block(golden, "dis_f8_eval", NLayerDiscriminator(1, 1, n_filters=8, n_layers=2), 1, train=False, file="gan.npz")
    with pytest.raises(NotImplementedError):
        NLayerDiscriminator(normalization='instancenorm')


def test_gan_losses_golden(golden):
    from functions import hinge_d_loss, generator_loss
    g = golden("gan.npz")
    real = g.t("hinge/real", DEV).requires_grad_(True)
    fake = g.t("hinge/fake", DEV).requires_grad_(True)
    l = hinge_d_loss(real, fake)
    (3.0 * l).backward()
    assert_close(l, g["hinge/loss"], 1e-6, "hinge loss")
    assert_close(real.grad, g["hinge/g_real"], 1e-6, "hinge g_real", atol=1e-9)
    assert_close(fake.grad, g["hinge/g_fake"], 1e-6, "hinge g_fake", atol=1e-9)
    x = g.t("gen/x", DEV).requires_grad_(True)
    lg = generator_loss(x)
    (2.0 * lg).backward()
    assert_close(lg, g["gen/loss"], 1e-6, "generator loss", atol=1e-8)
    assert_close(x.grad, g["gen/gx"], 1e-6, "generator gx")


def test_discriminator_update_golden(golden):
    """Two discriminator updates of _train_second_step_nl_dis (hinge on real / fake, Adam) against the reference run."""
    from networks import NLayerDiscriminator
    from functions import hinge_d_loss
    from hipops import Adam, ops
    g = golden("gan.npz")
    dis = NLayerDiscriminator(1, 1, n_filters=8, n_layers=3)
    dis.load_state_dict({k[2:]: v for k, v in g.group("dstep").items() if k.startswith("P.")}, strict=True)
    dis.to(DEV).train()
    opt = Adam(dis.parameters(), lr=1e-3, betas=(0.5, 0.999))
    for s in range(2):
        l_dis = hinge_d_loss(dis(g.t("dstep/real%d" % s, DEV)), dis(g.t("dstep/fake%d" % s, DEV)))
        assert_close(l_dis, g["dstep/loss%d" % s], 2e-4 if s else 1e-5, "l_dis step %d" % s)
        opt.zero_grad()
        ops.weighted_sum([l_dis], [0.8]).backward()
        opt.step()
    torch.cuda.synchronize()
    for k, v in dis.state_dict().items():
        ref = g["dstep/after." + k]
        if "num_batches" in k:
            assert int(v) == int(ref)
        else:
            assert_close(v.float(), ref.astype(np.float32), 2e-3, "after." + k, atol=2e-4)


def test_second_step_trainer_vs_oracle():
    """SecondStepTrainer (frozen encoder -> decoder -> D; generator update, then discriminator update) against the
    oracle's restatement of single_window_trainer.py:434-488 from the same initial state."""
    from oracle import vqwnet_ref as O
    from oracle import gan_ref as G
    from networks import UNetEncoder, UNetDecoder, NLayerDiscriminator
    from trainers import SecondStepTrainer, GanLossWeights
    torch.manual_seed(3)
    ef, df, K = [8, 8, 16, 16, 16], [8, 16, 16, 16, 32], 6
    enc = UNetEncoder(1, ef, K, 0.99, 'torch', False, 1, True)
    dec = UNetDecoder(ef[0], 1, df, use_dropblock=False, dropped_skip_layers=[], use_styled_up_block=True, use_pixel_shuffle=False)
    dis = NLayerDiscriminator(1, 1, n_filters=8, n_layers=3)
    PE = {k: v.clone().contiguous() for k, v in enc.state_dict().items()}
    PD = {k: v.clone().contiguous() for k, v in dec.state_dict().items()}
    PS = {k: v.clone().contiguous() for k, v in dis.state_dict().items()}
    g = torch.Generator().manual_seed(9)
    image = (torch.rand(2, 1, 32, 32, generator=g) * 2 - 1)
    w = GanLossWeights(recon=1.0, gen=0.1, dis=0.8)
    tr = SecondStepTrainer(enc, dec, dis, loss_weight=w, n_inner_loops=1, lr=1e-3, device=DEV)
    out = tr.training_step(image.to(DEV))
    # oracle
    for k in O.trainable_keys(PD):
        PD[k].requires_grad_(True)
    for k in [k for k, v in PS.items() if v.is_floating_point() and "running" not in k]:
        PS[k].requires_grad_(True)
    with torch.no_grad():
        e, _, ids, _ = O.encoder_forward(PE, image, False, 0.99)
    rec = O.decoder_forward(PD, e.detach(), True, 4)
    l_rec = F.mse_loss(rec, image)
    l_gen = G.generator_loss(G.discriminator_forward(PS, rec, True))
    assert torch.equal(out["ids"].cpu(), ids)
    assert_close(out["recon"], l_rec, 2e-5, "l_recon")
    assert_close(out["gen"], l_gen, 2e-4, "l_gen", atol=1e-6)
    assert_close(out["recon_image"], rec, 2e-4, "recon", atol=1e-5)
    gen_total = w.recon * l_rec + w.gen * l_gen
    grads = torch.autograd.grad(gen_total, [PD[k] for k in O.trainable_keys(PD)])
    # discriminator loss of the inner loop: D saw recon once already (running stats updated), decoder weights moved
    # after the generator update only affect the NEXT step, recon is the pre-update tensor (recon.detach())
    l_dis = G.hinge_d_loss(G.discriminator_forward(PS, image, True), G.discriminator_forward(PS, rec.detach(), True))
    assert_close(out["dis_total"], w.dis * l_dis, 2e-4, "l_dis_total")
    assert len(grads) > 0 and np.isfinite(float(out["gen_total"].detach()))


SCONV_CASES = [
    # N, H, W, Cin, Cout, ks, stride, pad, bias, slope
    (2, 32, 32, 16, 32, 4, 2, 1, True, 1.0),      # MFMA: stride-2 gather / transposed conv / swapped-role parity wgrad
    (2, 64, 32, 64, 128, 4, 2, 1, False, 1.0),
    (3, 16, 16, 32, 16, 4, 1, 1, False, 1.0),     # MFMA: stride 1 on the common grid (pad / crop)
    (2, 9, 12, 16, 48, 4, 1, 1, True, 1.0),
    (2, 32, 32, 1, 8, 4, 2, 1, True, 0.2),        # direct kernels: 1-channel ends, LeakyReLU epilogue
    (2, 17, 16, 16, 1, 4, 1, 1, True, 1.0),
    (1, 30, 30, 16, 32, 4, 2, 1, False, 0.2),     # slope in the epilogue keeps the direct kernel
    (2, 11, 13, 3, 5, 3, 2, 1, True, 1.0),        # another kernel size, odd sizes
    (1, 16, 16, 8, 12, 4, 2, 1, True, 1.0),       # low-res width 8: stride-2 wgrad falls back to the direct kernel
]


@pytest.mark.parametrize("case", SCONV_CASES)
def test_sconv2d(case):
    ops = _ops()
    N, H, W, Cin, Cout, ks, stride, pad, bias, slope = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, ks, ks, generator=g, dtype=torch.float64) * 0.2
    b = torch.randn(Cout, generator=g, dtype=torch.float64) if bias else None
    rx, rw = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    rb = b.clone().requires_grad_(True) if bias else None
    yref = F.leaky_relu(F.conv2d(rx, rw, rb, stride=stride, padding=pad), slope) if slope != 1.0 else F.conv2d(rx, rw, rb, stride=stride, padding=pad)
    r = torch.randn(yref.shape, generator=g, dtype=torch.float64)
    (yref * r).sum().backward()
    dx = x.float().to(DEV).requires_grad_(True)
    dw = w.float().to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    db = b.float().to(DEV).requires_grad_(True) if bias else None
    y = ops.sconv2d(dx, dw, db, stride=stride, padding=pad, slope=slope)
    assert tuple(y.shape) == tuple(yref.shape)
    (y * r.float().to(DEV)).sum().backward()
    torch.cuda.synchronize()
    tol = 2e-5
    assert_close(y, yref, tol, "y")
    assert_close(dx.grad, rx.grad, tol, "dx")
    assert_close(dw.grad, rw.grad, tol, "dw")
    if bias:
        assert_close(db.grad, rb.grad, tol, "db")


def test_recon_from_checkpoint_and_nifti(tmp_path):
    """Lightning-style checkpoint -> load_model -> edited NIfTI label map -> reconstruction file;
    equals the in-memory reconstruct() of the same models."""
    from collections import namedtuple
    import run_recon as RR
    from networks import UNetEncoder, UNetDecoder
    from utils.checkpoint import save_lightning_style_ckpt
    torch.manual_seed(5)
    ef, df, K = [8, 8, 16, 16, 16], [8, 16, 16, 16, 32], 6
    enc = UNetEncoder(1, ef, K, 0.99, 'torch', False, 1, False)
    dec = UNetDecoder(ef[0], 1, df, use_dropblock=False, dropped_skip_layers=[], use_styled_up_block=True, use_pixel_shuffle=False)
    ckpt = str(tmp_path / "first_stage.ckpt")
    save_lightning_style_ckpt(ckpt, enc, dec)
    Cfg = namedtuple("Cfg", "in_channels enc_filters dec_filters dict_size momentum knn_backend use_dropblock block_size "
                            "start_value stop_value nr_steps dropped_skip_layers use_pixel_shuffle resume_checkpoint")
    cfg = Cfg(1, ef, df, K, 0.99, 'torch', False, 30, 0.1, 0.5, 20, [], False, ckpt)
    e2, d2 = RR.load_model(cfg, device=DEV)
    g = torch.Generator().manual_seed(2)
    label = torch.randint(0, K + 1, (32, 32), generator=g)
    edited = str(tmp_path / "edited.nii")
    RR.save_as_nifti(label.float(), edited)
    out = str(tmp_path / "recon.nii")
    rec = RR.reconstruct_file(e2, d2, edited, out_nifti=out, device=DEV)
    ref = RR.reconstruct(enc.to(DEV), dec.to(DEV), label.unsqueeze(0).to(DEV))[0, 0].cpu().numpy()
    assert rec.shape == (32, 32) and np.array_equal(rec, ref)
    assert np.allclose(RR.load_from_nifti(out), ref)
    win = RR.reconstruct_file(e2, d2, edited, window=(2000, 0, 2.0), device=DEV)
    assert np.allclose(win, RR.normalize(RR.denormalize(ref, 2000, 0, 2.0), **RR.LUNG_WINDOW))


def test_multi_window_recon_loss():
    """windowed MSE kernel (value and gradient, clamped regions included) against the oracle's
    to_window / F.mse_loss, and the multi-window first step's total against its own parts."""
    from oracle import vqwnet_ref as O
    from trainers import FirstStepTrainer, LossWeights
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    a = (torch.rand(2, 1, 24, 24, generator=g) * 3 - 1.5)
    b = (torch.rand(2, 1, 24, 24, generator=g) * 3 - 1.5)
    dsw = (2000, 0, 2.0)
    for tw in (O.LUNG_WINDOW, O.MEDIASTINAL_WINDOW):
        ra = a.clone().requires_grad_(True)
        ref = F.mse_loss(O.to_window(ra, dict(width=2000, center=0, scale=2.0), tw), O.to_window(b, dict(width=2000, center=0, scale=2.0), tw))
        (1.7 * ref).backward()
        da = a.to(DEV).requires_grad_(True)
        l = ops.window_mse_loss(da, b.to(DEV), dsw, (tw["width"], tw["center"], tw["scale"]))
        (1.7 * l).backward()
        assert_close(l, ref, 2e-5, "window mse")
        assert_close(da.grad, ra.grad, 2e-4, "window mse grad", atol=1e-7)
        assert float((ra.grad == 0).float().mean()) > 0.05          # the case really has clamped pixels
    rw = (1.0, 0.5, 0.25)
    w = LossWeights(commit=1.0, cross=0.5, dist=0.3, reg=0.2, recon=2.0)
    tr = FirstStepTrainer(enc_filters=(4, 8, 8, 8, 8), dec_filters=(4, 8, 8, 8, 8), dict_size=6, device=DEV, loss_weight=w,
                          multi_window=dict(dataset_window=dsw, recon_weights=rw))
    img = (torch.rand(2, 1, 32, 32, generator=g) * 2 - 1).to(DEV)
    out = tr.forward_losses(img)
    c1, c2 = img.cpu(), torch.flip(img, dims=[3]).cpu()
    l_rec = O.multi_window_recon(out["recon_1"].detach().cpu(), c1, out["recon_2"].detach().cpu(), c2,
                                 dict(width=2000, center=0, scale=2.0), rw)
    parts = w.commit * (out["commit_1"] + out["commit_2"]) + w.cross * out["cross"] + w.dist * out["dist"] + w.reg * out["reg"]
    expect = parts.detach().cpu() + w.recon * l_rec
    assert_close(out["total"], expect, 2e-5, "multi-window total")
    tr.training_step(img)


def test_host_throttle_bounds_steps_in_flight_and_allocator_pool():
    """trainers.StepThrottle: never more than two steps enqueued ahead of the GPU, and in steady state a step allocates
    (almost) no new device segments — unthrottled, every step in flight needed a fresh working set because tensors that
    were record_stream'ed to the weight-gradient lanes / the second view cannot be reused before the GPU has passed them."""
    import bench
    from trainers import FirstStepTrainer
    torch.manual_seed(0)
    tr = FirstStepTrainer(device=DEV)
    assert tr.throttle.max_inflight == 2
    img, noise = bench.synthetic_batch(8, 128, 3, torch.device(DEV))
    for _ in range(4):                          # fills the pool for two steps in flight
        tr.training_step({"image": img}, noise=noise)
        assert len(tr.throttle.events) <= 2
    torch.cuda.synchronize()
    s0 = torch.cuda.memory_stats()
    for _ in range(8):
        tr.training_step({"image": img}, noise=noise)
        assert len(tr.throttle.events) <= 2
    torch.cuda.synchronize()
    s1 = torch.cuda.memory_stats()
    grown = s1["reserved_bytes.all.current"] - s0["reserved_bytes.all.current"]
    assert grown <= 0.25 * s0["reserved_bytes.all.current"], "allocator pool grew by %.1f MB over 8 steady-state steps" % (grown / 2**20)


def test_full_size_epilogue_fusions_agree_with_the_separate_passes(monkeypatch):
    """BASELINE config 2's network at 256 x 256 (batch 8): one training step with the backward fusions of round 3 - the ReLU
    mask in the gamma | beta input-gradient epilogue, gradient-group sums of block inputs, InstanceNorm backward sums from the
    consumer convolution - against the same step with every one of them as a separate pass.  The forward is untouched (loss
    and ids bit-equal); mask and group sums are the same additions (a + b), the norm sums are the same terms in another
    order, so every live gradient agrees to 1e-4 of its norm; and the fused step is bit-deterministic run to run."""
    import bench
    from hipops import ops
    from networks import blocks as B_
    from trainers import FirstStepTrainer

    def run(fused):
        monkeypatch.setattr(ops, "FUSE_RELU_MASK", fused)
        monkeypatch.setattr(ops, "FUSE_IN_BWD", fused)
        monkeypatch.setattr(B_, "GRAD_GROUP_BLOCKS", fused)
        monkeypatch.setattr(ops, "GRAD_GROUPS", fused)       # (the pyramid's group and its residual seed, round 4)
        torch.manual_seed(0)
        tr = FirstStepTrainer(device=DEV)
        img, noise = bench.synthetic_batch(8, 256, 77, torch.device(DEV))
        c0 = (ops.masked_dgrad_calls, ops.group_acc_calls, ops.in_bwd_fused_calls)
        out = tr.training_step({"image": img}, noise=noise)
        torch.cuda.synchronize()
        counts = (ops.masked_dgrad_calls - c0[0], ops.group_acc_calls - c0[1], ops.in_bwd_fused_calls - c0[2])
        grads = {"enc." + k: p.grad.detach().clone() for k, p in tr.encoder.named_parameters()}
        grads.update({"dec." + k: p.grad.detach().clone() for k, p in tr.decoder.named_parameters()})
        return float(out["total"].detach()), out["ids_1"].clone(), grads, counts
    l0, ids0, g0, n0 = run(False)
    l1, ids1, g1, n1 = run(True)
    l2, ids2, g2, n2 = run(True)
    assert n0 == (0, 0, 0) and min(n1) >= 8 and n1 == n2, (n0, n1, n2)
    assert l0 == l1 == l2 and torch.equal(ids0, ids1)
    assert all(torch.equal(g1[k], g2[k]) for k in g1), "the fused step is not deterministic"
    gmax = max(float(v.norm()) for v in g0.values())
    for k in g0:
        if float(g0[k].norm()) < 1e-5 * gmax:         # analytically zero (biases in front of a norm): rounding noise only
            continue
        assert_close(g1[k], g0[k], 1e-4, "gradient %s with the epilogue fusions" % k)


def test_training_step_is_bit_deterministic():
    """Two runs of the same seeded training (fresh modules, three steps, two views on two streams, weight gradients on
    the side stream) end in bit-identical parameters, codebook and losses: no result depends on kernel scheduling
    (VQ statistics and the cross-loss partials are reduced in a fixed order; no float atomics on this path)."""
    import bench
    from trainers import FirstStepTrainer

    def run():
        torch.manual_seed(0)
        tr = FirstStepTrainer(enc_filters=(16, 16, 32, 32, 32), dec_filters=(16, 32, 32, 32, 64), device=DEV)
        pool = [bench.synthetic_batch(4, 64, 100 + s, torch.device(DEV)) for s in range(2)]
        for i in range(3):
            img, noise = pool[i % 2]
            out = tr.training_step({"image": img}, noise=noise)
        torch.cuda.synchronize()
        ps = [p.detach().clone() for p in list(tr.encoder.parameters()) + list(tr.decoder.parameters())]
        return ps, float(out["total"].detach()), tr.encoder.vq.embed.clone(), tr.encoder.vq.cluster_size.clone()
    a, b = run(), run()
    assert a[1] == b[1]
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    assert all(torch.equal(x, y) for x, y in zip(a[0], b[0]))


@pytest.mark.parametrize("N,S,Cin,Cout,dil", [(32, 256, 32, 32, 1), (32, 128, 64, 64, 1), (32, 32, 256, 512, 1), (32, 16, 512, 512, 1),
                                              (32, 256, 32, 32, 18), (32, 256, 32, 64, 1)])
def test_conv_forms_agree_at_baseline_sizes(N, S, Cin, Cout, dil):
    """BASELINE config 2's layer shapes at full size (batch 32): the forms the auto backend takes - Winograd-form input and
    weight gradients, LDS-resident halo tiles / rows in the forward - against the implicit-GEMM kernels on the same tensors
    (backend 2), both exact fp32: outputs and all gradients agree to rounding."""
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(N + S + Cin)
    x = torch.randn(N, Cin, S, S, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, device=DEV, generator=g) / (9 * Cin) ** 0.5).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device=DEV, generator=g)
    gy = torch.randn(N, Cout, S, S, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)

    def run(backend):
        old = ops.set_conv_backend(backend)
        try:
            xs, ws, bs = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            y = ops.conv2d(xs, ws, bs, dilation=dil)
            y.backward(gy)
            torch.cuda.synchronize()
            return y.detach(), xs.grad, ws.grad, bs.grad
        finally:
            ops.set_conv_backend(old)
    a, r = run(0), run(2)
    for name, u, v, tol in zip(("y", "dx", "dw", "db"), a, r, (2e-6, 2e-6, 1e-5, 1e-5)):
        assert_close(u, v, tol, "%s: auto backend vs implicit-GEMM kernels" % name)
    with torch.no_grad():        # forward-only use: the Winograd forward (dilation 1) against the same reference
        assert_close(ops.conv2d(x, w, b, dilation=dil), r[0], 2e-6, "forward-only form")


@pytest.mark.parametrize("N,s,Cin,Cout", [(4, 32, 256, 128), (2, 64, 128, 64), (4, 16, 512, 256)])
def test_upsampled_winograd_forms_against_the_collapsed_forms(N, s, Cin, Cout):
    """3x3 over a nearest-x2 up-sampled input (StyledResUpBlock conv / conv1, blocks.py:100-112): the nine-product Winograd
    kernels (conv_wino_up.hip: forward, input gradient incl. its accumulating form, weight gradient) against the 16-tap
    collapsed forms on the same tensors.  Backend 3 ("no Winograd-form kernel anywhere") now switches the nine-product
    kernels off as well, so the 0-vs-3 A/B tests of the step really compare against direct-form arithmetic."""
    ops = _ops()
    L = ops._L()
    g = torch.Generator(device=DEV).manual_seed(N + s + Cin)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)      # noqa: E731
    x = cl(torch.randn(N, Cin, s, s, device=DEV, generator=g))
    wa = cl(torch.randn(Cout, Cin, 3, 3, device=DEV, generator=g) / (9 * Cin) ** 0.5)
    wb = cl(torch.randn(Cout, Cin, 3, 3, device=DEV, generator=g) / (9 * Cin) ** 0.5)
    b = torch.randn(Cout, device=DEV, generator=g)
    gya = cl(torch.randn(N, Cout, 2 * s, 2 * s, device=DEV, generator=g))
    gyb = cl(torch.randn(N, Cout, 2 * s, 2 * s, device=DEV, generator=g))

    def run(backend):
        old = ops.set_conv_backend(backend)
        try:
            has_acc = bool(L.vqw_conv3x3_up2_dgrad_acc_supported(Cin, Cout, N, s, s))
            xs = x.clone(memory_format=torch.channels_last).requires_grad_(True)
            was, wbs = (t.clone(memory_format=torch.channels_last).requires_grad_(True) for t in (wa, wb))
            bs = b.clone().requires_grad_(True)
            grp = ops.GradGroup(2)             # two up-sampled convolutions of one input: the second adds in its epilogue
            n0 = ops.group_acc_calls
            ya = ops.conv2d(xs, was, bs, up2x=True, grad_group=grp)
            yb = ops.conv2d(xs, wbs, None, up2x=True, grad_group=grp)
            torch.autograd.backward([ya, yb], [gya, gyb])
            torch.cuda.synchronize()
            return (ya.detach(), yb.detach(), xs.grad, was.grad, wbs.grad, bs.grad), has_acc, ops.group_acc_calls - n0
        finally:
            ops.set_conv_backend(old)
    (a, acc_a, n_a), (r, acc_r, n_r) = run(0), run(3)
    assert acc_a and not acc_r, "backend 3 must switch the nine-product kernels off (%s / %s)" % (acc_a, acc_r)
    assert n_a == 1 and n_r == 0
    for name, u, v, tol in zip(("ya", "yb", "dx", "dwa", "dwb", "db"), a, r, (2e-6, 2e-6, 3e-6, 1e-5, 1e-5, 1e-5)):
        assert_close(u, v, tol, "%s: nine-product Winograd form vs collapsed 16-tap form" % name)


def test_gradient_group_sums_branch_gradients_in_place():
    """ASPP's five branches read one tensor: with ops.GradGroup their input gradients are summed in place (the row-chain
    kernel's accumulating form for the dilated branches) and autograd sees one gradient; same values as autograd's sum."""
    from networks.aspp import ASPP
    ops = _ops()
    torch.manual_seed(5)
    m = ASPP(32, 32, [2, 6, 12, 18]).to(DEV)
    x = torch.randn(2, 32, 40, 64, device=DEV)
    r = torch.randn(2, 160, 40, 64, device=DEV)

    def run(groups):
        old, ops.GRAD_GROUPS = ops.GRAD_GROUPS, groups
        try:
            xs = x.clone().requires_grad_(True)
            for p_ in m.parameters():
                p_.grad = None
            y = m(xs)
            (y * r).sum().backward()
            torch.cuda.synchronize()
            return y.detach(), xs.grad, [p_.grad.clone() for p_ in m.parameters()]
        finally:
            ops.GRAD_GROUPS = old
    ya, ga, wa = run(True)
    yb, gb, wb = run(False)
    assert torch.equal(ya, yb)
    assert_close(ga, gb, 1e-6, "input gradient: in-place group sum vs autograd's sum")
    for u, v in zip(wa, wb):
        assert torch.equal(u, v)
    # a second pass over the retained graph delivers the same gradient again (the group re-arms itself) ...
    xs = x.clone().requires_grad_(True)
    y = m(xs)
    loss = (y * r).sum()
    loss.backward(retain_graph=True)
    g1 = xs.grad.clone()
    xs.grad = None
    loss.backward()
    torch.cuda.synchronize()
    assert torch.equal(xs.grad, g1) and torch.equal(g1, ga)
    # ... and a pass over only some of the branch outputs raises instead of silently dropping the shared input gradient
    xs = x.clone().requires_grad_(True)
    wts = [(torch.randn(32, 32, 3, 3, device=DEV) * 0.1).contiguous(memory_format=torch.channels_last).requires_grad_(True) for _ in range(2)]
    grp = ops.GradGroup(2)
    y1, y2 = ops.conv2d(xs, wts[0], dilation=2, grad_group=grp), ops.conv2d(xs, wts[1], dilation=6, grad_group=grp)
    with pytest.raises(RuntimeError, match="GradGroup"):
        torch.autograd.grad(y1.sum(), xs, retain_graph=True)
    # the same rejection with leaf weights (their gradients run on the side lanes): the engine skips every end-of-pass callback
    # queued behind the one that raises, so the group's callback joins the lanes itself and the queue flag cannot latch
    with pytest.raises(RuntimeError, match="GradGroup"):
        y1.sum().backward(retain_graph=True)
    assert ops._join_queued_for is None, "the lane join of a rejected pass stayed queued"
    (y1 + y2).sum().backward(retain_graph=True)                   # a following pass queues and runs its own join again
    assert ops._join_queued_for is None
    torch.cuda.synchronize()
    both = torch.autograd.grad((y1 + y2).sum(), xs)[0]          # the group was re-armed: a complete pass still works
    want = torch.autograd.grad((ops.conv2d(xs, wts[0], dilation=2) + ops.conv2d(xs, wts[1], dilation=6)).sum(), xs)[0]
    torch.cuda.synchronize()
    assert_close(both, want, 1e-6, "grouped input gradient after a rejected partial pass")
    # the accumulating kernel itself: y0 + conv(x) (dilated 3x3, 32 channels)
    L = ops._L()
    w = (torch.randn(32, 32, 3, 3, device=DEV) * 0.1).contiguous(memory_format=torch.channels_last)
    for d in (2, 18):
        assert L.vqw_conv2d_fwd_acc_supported(32, 2, 40, 64, 32, 3, d)
        y0 = torch.randn(2, 32, 40, 64, device=DEV).contiguous(memory_format=torch.channels_last)
        want = y0 + ops.conv2d(x, w, dilation=d)
        got = y0.clone(memory_format=torch.channels_last)
        from hipops import _lib
        _lib.check(L.vqw_conv2d_fwd_acc(ops._p(ops.nhwc(x)), 32, ops._p(w), ops._p(got), 2, 40, 64, 32, 3, d, ops._st()), "acc")
        torch.cuda.synchronize()
        assert_close(got, want, 1e-6, "y0 + conv, dilation %d" % d)


def test_conv_tensors_beyond_4gib_run_as_image_groups():
    """A conv whose activation tensors exceed the 32-bit buffer-descriptor range (4 GiB) is run by the library as
    consecutive image groups: forward, input gradient and (accumulated) weight gradient equal the same work done in
    explicit halves that fit."""
    ops = _ops()
    N, C, S = 18, 256, 512                       # 18 x 512 x 512 x 256 fp32 = 4.8 GB per tensor
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(N, C, S, S, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, C, 3, 3, device=DEV, generator=g) * 0.02).contiguous(memory_format=torch.channels_last)
    b = torch.randn(C, device=DEV, generator=g)
    assert x.numel() * 4 > 2 ** 32

    gy = torch.randn(N, C, S, S, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)

    def run(xs, gys):
        xs = xs.detach().requires_grad_(True)
        ws, bs = w.detach().requires_grad_(True), b.detach().requires_grad_(True)
        y = ops.conv2d(xs, ws, bs)
        y.backward(gys)
        torch.cuda.synchronize()
        return y.detach(), xs.grad, ws.grad, bs.grad
    y, gx, gw, gb = run(x, gy)
    h = N // 2
    y1, gx1, gw1, gb1 = run(x[:h], gy[:h])
    y2, gx2, gw2, gb2 = run(x[h:], gy[h:])
    assert torch.equal(y[:h], y1) and torch.equal(y[h:], y2)
    assert torch.equal(gx[:h], gx1) and torch.equal(gx[h:], gx2)
    assert_close(gw, gw1 + gw2, 2e-5, "dw over image groups")
    assert_close(gb, gb1 + gb2, 1e-4, "db over image groups", atol=1e-2)


# --------------------------------------------------------------------------------------------------
# dispatcher operators (the boundary is a set of PyTorch custom ops)
# --------------------------------------------------------------------------------------------------
def test_functional_dispatcher_ops_match_module_ops_and_pass_opcheck():
    """torch.ops.vqw.{conv2d, instance_norm, vq_forward, embed_cross_loss, res_tail}: functional operators with
    register_autograd formulas over the kernel operators - same values and gradients as the autograd.Function operators
    the modules use, and torch.library.opcheck (schema, fake tensors, autograd registration) is clean."""
    ops = _ops()
    from hipops import functional  # noqa: F401  (registers the operators)
    tests = ("test_schema", "test_autograd_registration", "test_faketensor")
    torch.manual_seed(0)
    # conv2d: plain, up-sampled + concat, fused ReLU
    for (Ci, Cs, Co, up, relu, dil) in [(16, 0, 32, False, False, 1), (32, 16, 32, True, True, 1), (16, 0, 16, False, False, 6)]:
        x = torch.randn(2, Ci, 8 if up else 16, 8 if up else 16, device=DEV, requires_grad=True)
        sk = torch.randn(2, Cs, 16, 16, device=DEV, requires_grad=True) if Cs else None
        w = (torch.randn(Co, Ci + Cs, 3, 3, device=DEV) * 0.1).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        b = torch.randn(Co, device=DEV, requires_grad=True)
        r = torch.randn(2, Co, 16, 16, device=DEV)
        y = torch.ops.vqw.conv2d(x, w, b, dil, up, sk, relu)
        gs = torch.autograd.grad((y * r).sum(), [t for t in (x, w, b, sk) if t is not None])
        x2, w2, b2 = x.detach().requires_grad_(True), w.detach().requires_grad_(True), b.detach().requires_grad_(True)
        s2 = sk.detach().requires_grad_(True) if sk is not None else None
        y2 = ops.conv2d(x2, w2, b2, dilation=dil, up2x=up, skip=s2, relu=relu)
        assert torch.equal(y, y2)
        (y2 * r).sum().backward()
        torch.cuda.synchronize()
        for a, ref in zip(gs, [t.grad for t in (x2, w2, b2, s2) if t is not None]):
            assert_close(a, ref, 1e-6, "conv2d functional grad", atol=1e-7)
        torch.library.opcheck(torch.ops.vqw.conv2d.default, (x, w, b, dil, up, sk, relu), test_utils=tests)
    # instance norm (+ReLU)
    x = torch.randn(2, 16, 12, 12, device=DEV, requires_grad=True)
    r = torch.randn(2, 16, 12, 12, device=DEV)
    y, mr = torch.ops.vqw.instance_norm(x, True, 1e-5)
    (gx,) = torch.autograd.grad((y * r).sum(), [x])
    x2 = x.detach().requires_grad_(True)
    y2 = ops.instance_norm(x2, relu=True)
    (y2 * r).sum().backward()
    assert torch.equal(y, y2) and torch.equal(gx, x2.grad)
    torch.library.opcheck(torch.ops.vqw.instance_norm.default, (x, True, 1e-5), test_utils=tests)
    # VQ forward (search + gather + commit) with the straight-through backward
    embed = torch.randn(10, 16, device=DEV)
    x = torch.randn(2, 16, 8, 8, device=DEV, requires_grad=True)
    r = torch.randn(2, 16, 8, 8, device=DEV)
    q, commit, ids = torch.ops.vqw.vq_forward(x, embed, 1)
    (gx,) = torch.autograd.grad((q * r).sum() + 3.0 * commit, [x])
    x2 = x.detach().requires_grad_(True)
    q2, c2, ids2 = ops.vq_quantize(x2, embed.clone(), torch.zeros(10, device=DEV), embed.t().contiguous(), False, 0.9, 1e-5, id_base=1)
    ((q2 * r).sum() + 3.0 * c2).backward()
    assert torch.equal(ids, ids2) and torch.equal(q, q2) and torch.equal(commit, c2) and torch.equal(gx, x2.grad)
    torch.library.opcheck(torch.ops.vqw.vq_forward.default, (x, embed, 1), test_utils=tests)
    # cross loss on integer labels
    lab = torch.randint(0, 11, (2, 8, 8), device=DEV, dtype=torch.int32)
    e = torch.randn(2, 16, 8, 8, device=DEV, requires_grad=True)
    loss, coef = torch.ops.vqw.embed_cross_loss(e, lab, embed)
    (ge,) = torch.autograd.grad(2.0 * loss, [e])
    e2 = e.detach().requires_grad_(True)
    l2 = ops.cross_loss_labels(e2, lab, embed)
    (2.0 * l2).backward()
    assert torch.equal(loss, l2) and torch.equal(ge, e2.grad)
    torch.library.opcheck(torch.ops.vqw.embed_cross_loss.default, (e, lab, embed), test_utils=tests)
    # ResBlock tail
    a = torch.randn(2, 16, 8, 8, device=DEV, requires_grad=True)
    b = torch.randn(2, 16, 8, 8, device=DEV, requires_grad=True)
    rp, ro = torch.randn(2, 16, 4, 4, device=DEV), torch.randn(2, 16, 8, 8, device=DEV)
    pooled, out = torch.ops.vqw.res_tail(a, b)
    ga, gb = torch.autograd.grad((pooled * rp).sum() + (out * ro).sum(), [a, b])
    a2, b2 = a.detach().requires_grad_(True), b.detach().requires_grad_(True)
    p2, o2 = ops.res_tail(a2, b2)
    ((p2 * rp).sum() + (o2 * ro).sum()).backward()
    assert torch.equal(pooled, p2) and torch.equal(out, o2) and torch.equal(ga, a2.grad) and torch.equal(gb, b2.grad)
    torch.library.opcheck(torch.ops.vqw.res_tail.default, (a, b), test_utils=tests)
    # kernel operators carry their mutation annotations: the in-place codebook update
    sch = str(torch.ops.vqw.vq_ema_update.default._schema)
    assert "Tensor(a!)? embed" in sch and "Tensor(b!)? cluster_size" in sch and "Tensor(c!)? embed_avg" in sch


# --------------------------------------------------------------------------------------------------
# config-driven construction and the k-means codebook initialisation
# --------------------------------------------------------------------------------------------------
def test_trainer_built_from_baseline_config_passes_the_golden_step(golden):
    """configs/baseline1_cpu_32x32_b4.json -> trainers.build_first_step_trainer -> the reference's step_rcfg32 vectors:
    the config route constructs the same modules (initial weights by checksum), optimisers and losses as the fixture's
    generator did from explicit arguments."""
    from utils import load_json
    from trainers import build_first_step_trainer, FlipViews
    from helpers import check_init
    g = golden("step_rcfg32.npz")
    cfg = load_json(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "baseline1_cpu_32x32_b4.json"))
    torch.manual_seed(int(g["cfg/seed"]))
    tr = build_first_step_trainer(cfg, device=DEV, views=FlipViews(border=int(g["cfg/border"])))
    check_init(g, tr.encoder, tr.decoder)
    out = tr.training_step({"image": g.t("step0/image", DEV)}, noise=g.t("step0/noise", DEV))
    sc = tr.scalars(out)
    for k in ("total", "commit", "cross", "dist", "reg", "recon"):
        assert_close(sc[k], g["step0/" + k], 5e-4, k)
    for v in ("1", "2"):
        assert_ids_equal_where_clear(out["ids_" + v], g["step0/ids_" + v], g["step0/gap_" + v], "config-built trainer ids_" + v)
    grads = {"enc." + k: p.grad for k, p in tr.encoder.named_parameters()}
    grads.update({"dec." + k: p.grad for k, p in tr.decoder.named_parameters()})
    check_grads_vs_fp64(g, grads, 2.0, "config-built trainer")


@pytest.mark.parametrize("multi", [False, True])
def test_adam_weight_decay_matches_torch(multi, monkeypatch):
    """hipops.Adam with weight_decay != 0 (the reference passes the config's value: trainers/base.py:164-183) against
    torch.optim.Adam on the CPU over three steps: L2 decay added to the gradient before the moments, on the per-tensor
    launches and on the multi-tensor launch (VQW_ADAM_MULTI=1), for conv weights in channels_last layout (incl. a 1x1 and a
    1-channel weight, whose size-1 dimensions have arbitrary strides), a bias and a tensor longer than one launch chunk."""
    from hipops import optim
    monkeypatch.setattr(optim, "MULTI_TENSOR", multi)
    g = torch.Generator().manual_seed(4)
    shapes = [(32, 16, 3, 3), (16, 1, 3, 3), (8, 32, 1, 1), (32,), (70000,)]
    ref = [torch.randn(sh, generator=g).requires_grad_(True) for sh in shapes]
    mine = [(p.detach().clone().contiguous(memory_format=torch.channels_last) if p.dim() == 4 else p.detach().clone()).to(DEV).requires_grad_(True)
            for p in ref]
    kw = dict(lr=3e-3, betas=(0.5, 0.999), eps=1e-8, weight_decay=1e-2)
    o_ref, o_mine = torch.optim.Adam(ref, **kw), optim.Adam(mine, **kw)
    for step in range(3):
        for p, q in zip(ref, mine):
            gr = torch.randn(p.shape, generator=g) * (0.1 + step)
            p.grad = gr.clone()
            q.grad = (gr.contiguous(memory_format=torch.channels_last) if gr.dim() == 4 else gr.clone()).to(DEV)
        o_ref.step(); o_mine.step()
    torch.cuda.synchronize()
    for i, (p, q) in enumerate(zip(ref, mine)):
        assert_close(q, p, 5e-6, "parameter %d after three decayed steps" % i)
        assert_close(o_mine.state[q]["exp_avg"], o_ref.state[p]["exp_avg"], 2e-6, "exp_avg %d" % i)
        # (the C ABI carries beta2 as a float: 1 - beta2 formed from it is 1.3e-5 off the double constant torch multiplies with)
        assert_close(o_mine.state[q]["exp_avg_sq"], o_ref.state[p]["exp_avg_sq"], 3e-5, "exp_avg_sq %d" % i)
    # the decay term is really there: the same gradients without it end measurably elsewhere
    o_plain = torch.optim.Adam([p.detach().clone().requires_grad_(True) for p in ref], lr=kw["lr"], betas=kw["betas"], eps=kw["eps"])
    assert kw["weight_decay"] > 0 and o_plain.defaults["weight_decay"] == 0


@pytest.mark.parametrize("name", ["step_rcfg64_warm.npz", "step_cfg4_32.npz"])
def test_training_ids_do_not_depend_on_the_winograd_kernels(golden, name):
    """The TRAINING forward's codebook indices (and everything else the encoder hands on: quantised map, commitment loss) are
    bit-identical with the Winograd-form kernels in use (default: input / weight gradients everywhere, forward past the
    decoder's last max-pool) and without any of them (conv backend 3): the forward of the encoder and of every layer in
    front of a max-pool stays in direct form (`ops.winograd_forward` placement)."""
    from hipops import ops
    g = golden(name)
    res = []
    for backend in (0, 3):
        old = ops.set_conv_backend(backend)
        try:
            tr, cfg = _hip_trainer(g)
            out = tr.training_step({"image": g.t("step0/image", DEV)}, noise=g.t("step0/noise", DEV))
            torch.cuda.synchronize()
            res.append((out["ids_1"].clone(), out["ids_2"].clone(), tr.scalars(out), tr.encoder.vq.embed.clone()))
        finally:
            ops.set_conv_backend(old)
    (a1, a2, sa, ea), (b1, b2, sb, eb) = res
    assert torch.equal(a1, b1) and torch.equal(a2, b2)
    assert sa["commit"] == sb["commit"] and sa["cross"] == sb["cross"] and sa["dist"] == sb["dist"] and sa["reg"] == sb["reg"]
    assert torch.equal(ea, eb)                                   # the EMA update saw the same assignment and features
    assert abs(sa["recon"] - sb["recon"]) <= 1e-5 * abs(sb["recon"])        # the decoder's pool-free layers differ in form
    for v, ids in (("1", a1), ("2", a2)):
        assert_ids_equal_where_clear(ids, g["step0/ids_" + v], g["step0/gap_" + v], "ids_" + v)


@pytest.mark.parametrize("case", [(4, 512, 16, 11), (8, 1024, 32, 24), (6, 768, 16, 4)])
def test_kmeans_codebook_vs_oracle(case):
    """ops.kmeans_codebook (VQ search + statistics kernels + vqw_kmeans_update) against oracle/kmeans_ref.py on the same
    rows and the same seeded start: the same number of Lloyd iterations, the same assignment (the fixtures' top-1 / top-2
    gaps are >= 3e-3, three orders above fp32 rounding of the scores), centres to 1e-5, the same empty clusters (the second
    case ends with two, which keep their starting rows)."""
    from oracle import kmeans_ref as KR
    ops = _ops()
    K, P, D, seed = case
    x, _, _ = KR.blobs(P, D, K, seed=100 + seed)
    cen, ids, tr = KR.kmeans(x, K, seed=seed)
    assert min(t["min_gap"] for t in tr) > 2e-3
    c, hist, hid = ops.kmeans_codebook(x.to(DEV), K, seed=seed, return_ids=True)
    torch.cuda.synchronize()
    assert len(hist) == len(tr)
    assert np.array_equal(hid.cpu().numpy(), ids)
    assert_close(c, torch.from_numpy(cen), 1e-5, "centres")
    for h, t in zip(hist, tr):
        assert abs(h[0] * P - t["inertia"]) <= 1e-4 * t["inertia"] + 1e-6
        assert abs(h[1] - t["shift"]) <= 1e-4 * t["shift"] + 1e-5
        assert h[2] == t["empty"]


def test_kmeans_codebook_initialisation_properties():
    """UNetEncoder(init_embed=False) runs the one-off k-means on its first forward (unet_encoder.py:66-91).  Own semantics
    (kmeans_pytorch is absent): Lloyd's algorithm - inertia never increases, at the end every centre is the mean of its
    members, well-separated blobs are recovered, the run is reproducible, only vq.embed changes."""
    ops = _ops()
    from networks import UNetEncoder
    g = torch.Generator().manual_seed(0)
    K, D = 12, 16
    true = torch.randn(K, D, generator=g) * 4
    lab = torch.randint(0, K, (6000,), generator=g)
    x = (true[lab] + 0.1 * torch.randn(6000, D, generator=g)).to(DEV)
    c1, hist = ops.kmeans_codebook(x, K, seed=3)
    c2, _ = ops.kmeans_codebook(x, K, seed=3)
    assert torch.equal(c1, c2)
    inertia = [h[0] for h in hist]
    assert all(b <= a * (1 + 1e-6) for a, b in zip(inertia, inertia[1:])), inertia
    assert hist[-1][1] ** 2 < 1e-4 and len(hist) < 100
    d = torch.cdist(x, c1)
    ids = d.argmin(1)
    for k in range(K):
        m = ids == k
        if int(m.sum()):
            assert_close(c1[k], x[m].mean(0), 2e-3, "centre %d = mean of its members" % k)      # stopped at shift^2 < 1e-4, not at the fixed point
    # random-start Lloyd ends in a local optimum (some blobs merged, others split): at least half of the blobs are hit exactly
    found = (torch.cdist(true.to(DEV), c1).min(1).values < 0.2).sum()
    assert int(found) >= K // 2 and inertia[-1] < 0.5 * inertia[0]
    # through the module: first forward initialises, later ones do not
    torch.manual_seed(1)
    enc = UNetEncoder(1, [16, 16, 32, 32, 32], K, 0.99, "torch", False, 1, False).to(DEV).train()
    ea0, cs0, e0 = enc.vq.embed_avg.clone(), enc.vq.cluster_size.clone(), enc.vq.embed.clone()
    img = torch.randn(2, 1, 32, 32, device=DEV)
    with torch.no_grad():
        feat = enc.feature_extraction(img)
    enc.eval()                                       # eval: the forward leaves the EMA buffers alone
    q, commit, ids = enc(img)
    assert enc.init_embed is True and not torch.equal(enc.vq.embed, e0)
    assert torch.equal(enc.vq.embed_avg, ea0) and torch.equal(enc.vq.cluster_size, cs0)
    rows = feat.permute(0, 2, 3, 1).reshape(-1, 16)
    expect, _ = ops.kmeans_codebook(rows, K, seed=0)
    assert torch.equal(enc.vq.embed, expect)
    e1 = enc.vq.embed.clone()
    enc(img)
    assert torch.equal(enc.vq.embed, e1)
    assert float(commit) < float(((rows - e0[torch.cdist(rows, e0).argmin(1)]) ** 2).mean())     # better than the random codebook
