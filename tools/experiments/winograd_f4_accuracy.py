#!/usr/bin/env python3
"""VERDICT r03 item 8 (exploratory): would F(4x4, 3x3) in fp32 pass the per-kernel gate (2e-5 relative L2 against an fp64
convolution) that every convolution form of the package is held to?  CPU emulation of the arithmetic the kernel would run:
fp32 transforms, fp32 products accumulated in fp32 over the input channels (one accumulator per transformed position, as the
MFMA does), fp32 inverse transform; weights transformed in double and rounded to fp32 (as k_wino_weights does for F(2x2)).

    python tools/experiments/winograd_f4_accuracy.py

Prints the relative L2 error of the direct form (torch fp32 conv), F(2x2, 3x3) and F(4x4, 3x3) for the decoder's C >= 128 shapes,
forward (= input gradient: same kernel on dY) and weight gradient.  Also counts the transform additions per tile.
"""
import numpy as np
import torch

torch.manual_seed(0)

# Lavin & Gray, F(4x4, 3x3), interpolation points 0, +-1, +-2, inf
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
               [0, 0, 1]], dtype=np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def wino_fwd(x, w, BT, G, AT, m):
    """x (N, C, H, W) fp32 (H, W multiples of m), w (K, C, 3, 3) fp32 -> y (N, K, H, W) through F(m x m, 3x3) in fp32."""
    N, C, H, W = x.shape
    K = w.shape[0]
    a = m + 2
    U = torch.from_numpy(np.einsum("ij,kcjl,ml->imkc", G, w.double().numpy(), G)).float()          # (a, a, K, C), rounded once
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, a, m).unfold(3, a, m)                       # (N, C, th, tw, a, a)
    BTf = torch.from_numpy(BT).float()
    V = torch.einsum("ij,nctwjl,ml->nctwim", BTf, tiles, BTf)         # fp32 transform (exact small-integer coefficients)
    M = torch.einsum("imkc,nctwim->nktwim", U, V)                     # fp32 accumulation over C per transformed position
    ATf = torch.from_numpy(AT).float()
    Y = torch.einsum("pi,nktwim,qm->nktwpq", ATf, M, ATf)             # (N, K, th, tw, m, m)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, K, H, W)


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


print("shape            direct fp32   F(2x2) fp32   F(4x4) fp32     (relative L2 against the fp64 convolution; gate: 2e-5)")
for C, K, S in ((128, 128, 64), (256, 256, 32), (128, 256, 64), (512, 512, 16)):
    x = torch.randn(2, C, S, S)
    w = torch.randn(K, C, 3, 3) / (9 * C) ** 0.5
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    d = torch.nn.functional.conv2d(x, w, padding=1)
    f2 = wino_fwd(x, w, BT2, G2, AT2, 2)
    f4 = wino_fwd(x, w, BT4, G4, AT4, 4)
    print("%3d->%3d @%2d     %.2e      %.2e      %.2e" % (C, K, S, rel(d, ref), rel(f2, ref), rel(f4, ref)))
# a plane 20 sigma off zero (an un-normalised activation: the case that broke the fp32 statistics in round 2)
x = torch.randn(2, 128, 64, 64) + 20.0
w = torch.randn(128, 128, 3, 3) / (9 * 128) ** 0.5
ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
print("128->128 @64, inputs 20 sigma off zero: direct %.2e  F(2x2) %.2e  F(4x4) %.2e" % (
    rel(torch.nn.functional.conv2d(x, w, padding=1), ref), rel(wino_fwd(x, w, BT2, G2, AT2, 2), ref), rel(wino_fwd(x, w, BT4, G4, AT4, 4), ref)))


def adds(BT):
    """additions / multiplications of y = BT d applied as sparse rows (a +-1 coefficient costs an add, another one an fma)."""
    n = 0
    for row in BT:
        nz = [c for c in row if c != 0]
        n += max(len(nz) - 1, 0) + sum(1 for c in nz[:1] if abs(c) != 1)      # first term: a multiply unless +-1; the rest fma / add
    return n


for name, BT, AT, m in (("F(2x2,3x3)", BT2, AT2, 2), ("F(4x4,3x3)", BT4, AT4, 4)):
    a = m + 2
    inp = adds(BT) * a + adds(BT) * a          # rows then columns of the a x a patch
    out = adds(AT) * a + adds(AT) * m
    print("%s: input transform %3d VALU per (tile, channel) = %.2f per output pixel and channel; inverse transform %3d per (tile, cout); "
          "%2d products per tile = %.2f per output pixel" % (name, inp, inp / (m * m), out, a * a, a * a / (m * m)))
