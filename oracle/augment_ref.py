"""TEST INFRASTRUCTURE ONLY — CPU statement of the two-view augmentation and id-map warps.

Reference: src/networks/random_transform.py:10-112 (RandomTransform.forward / forward_transform / reverse_transform),
used at src/trainers/single_window_trainer.py:75-76, 91-96.  The reference composes them from kornia 0.5.1
(K.augmentation.RandomHorizontalFlip / RandomAffine / ColorJitter / RandomGaussianBlur / RandomPosterize /
RandomGaussianNoise, K.geometry.transform.warp_perspective) and kornia is not installed here and cannot be fetched:

    PARITY UNPINNED.  This file DEFINES the arithmetic the HIP kernels implement (pixel-centre coordinates, zero
    padding, round-half-even nearest sampling, the order brightness -> contrast -> posterize -> noise, reflect-border
    separable blur); it follows kornia's documented conventions where they are known, but it has not been checked
    against kornia's output.  What IS checked: the HIP kernels against this file, and the properties the training step
    relies on (reverse then forward of the same transform is the identity away from the frame border; flips and
    whole-pixel shifts are exact; out-of-frame ids are 0).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package.
"""
import math

import numpy as np


# ------------------------------------------------------------------------------------------------
# 3x3 matrices in pixel coordinates, mapping a SOURCE pixel to a DESTINATION pixel (what kornia's
# return_transform=True hands back, random_transform.py:83-85)
# ------------------------------------------------------------------------------------------------
def identity_matrix():
    return np.eye(3, dtype=np.float64)


def hflip_matrix(W):
    m = np.eye(3, dtype=np.float64)
    m[0, 0] = -1.0
    m[0, 2] = W - 1.0
    return m


def affine_matrix(angle_deg, tx, ty, shear_x_deg, shear_y_deg, H, W):
    """Rotate by angle (counter-clockwise on the screen, y pointing down), shear, all about the image centre
    ((W-1)/2, (H-1)/2), then translate by (tx, ty) pixels."""
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    a = math.radians(angle_deg)
    rot = np.array([[math.cos(a), math.sin(a), 0.0], [-math.sin(a), math.cos(a), 0.0], [0.0, 0.0, 1.0]])
    sh = np.array([[1.0, -math.tan(math.radians(shear_x_deg)), 0.0], [-math.tan(math.radians(shear_y_deg)), 1.0, 0.0],
                   [0.0, 0.0, 1.0]])
    to_c = np.array([[1.0, 0.0, -cx], [0.0, 1.0, -cy], [0.0, 0.0, 1.0]])
    back = np.array([[1.0, 0.0, cx + tx], [0.0, 1.0, cy + ty], [0.0, 0.0, 1.0]])
    return back @ rot @ sh @ to_c


def dst_to_src(m_fwd):
    """The matrix the kernels take: destination pixel -> source pixel, float32 (computed in float64)."""
    return np.linalg.inv(np.asarray(m_fwd, dtype=np.float64)).astype(np.float32)


def _source_coords(minv, H, W):
    m = minv.astype(np.float64)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    u = m[0, 0] * xs + m[0, 1] * ys + m[0, 2]
    v = m[1, 0] * xs + m[1, 1] * ys + m[1, 2]
    w = m[2, 0] * xs + m[2, 1] * ys + m[2, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        return u / w, v / w


def warp_image(src, minv):
    """src (B, C, H, W) float32, minv (B, 3, 3) float32 -> bilinear resample, zero padding."""
    B, C, H, W = src.shape
    out = np.zeros_like(src)
    for b in range(B):
        sx, sy = _source_coords(minv[b], H, W)
        ok = np.isfinite(sx) & np.isfinite(sy) & (np.abs(sx) < 1e9) & (np.abs(sy) < 1e9)
        sx = np.where(ok, sx, -10.0)
        sy = np.where(ok, sy, -10.0)
        x0f, y0f = np.floor(sx), np.floor(sy)
        fx, fy = (sx - x0f).astype(np.float32), (sy - y0f).astype(np.float32)
        x0, y0 = x0f.astype(np.int64), y0f.astype(np.int64)
        one = np.float32(1.0)

        def at(p, yy, xx):
            inside = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            return np.where(inside, p[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], np.float32(0.0))
        for c in range(C):
            p = src[b, c]
            top = at(p, y0, x0) * (one - fx) + at(p, y0, x0 + 1) * fx
            bot = at(p, y0 + 1, x0) * (one - fx) + at(p, y0 + 1, x0 + 1) * fx
            out[b, c] = np.where(ok, top * (one - fy) + bot * fy, np.float32(0.0))
    return out


def warp_labels(ids, minv):
    """ids (B, H, W) integer, minv (B, 3, 3) -> nearest (round half to even) resample as int32, 0 = out of frame."""
    B, H, W = ids.shape
    out = np.zeros((B, H, W), dtype=np.int32)
    for b in range(B):
        sx, sy = _source_coords(minv[b], H, W)
        ok = np.isfinite(sx) & np.isfinite(sy) & (np.abs(sx) < 1e9) & (np.abs(sy) < 1e9)
        xi = np.rint(np.where(ok, sx, -10.0)).astype(np.int64)
        yi = np.rint(np.where(ok, sy, -10.0)).astype(np.int64)
        inside = ok & (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
        out[b] = np.where(inside, ids[b][np.clip(yi, 0, H - 1), np.clip(xi, 0, W - 1)], 0).astype(np.int32)
    return out


# ------------------------------------------------------------------------------------------------
# photometric part (ColorJitter brightness / contrast on a grey image, RandomPosterize, RandomGaussianNoise)
# ------------------------------------------------------------------------------------------------
def photometric(x, params, noise=None):
    """x (B, ...) float32 in [0, 1]; params (B, 4) = {brightness add, contrast multiplier, posterize bits (8 = off),
    noise std}; noise like x (standard normal) or None."""
    B = x.shape[0]
    y = x.astype(np.float32).copy()
    for b in range(B):
        add, mul, bits, std = (np.float32(params[b, 0]), np.float32(params[b, 1]), int(params[b, 2]), np.float32(params[b, 3]))
        v = np.clip(y[b] + add, np.float32(0.0), np.float32(1.0))
        v = np.clip(v * mul, np.float32(0.0), np.float32(1.0))
        if bits < 8:
            u = (v * np.float32(255.0)).astype(np.int32)            # truncation
            u &= (0xFF << (8 - bits)) & 0xFF
            v = u.astype(np.float32) / np.float32(255.0)
        if noise is not None:
            v = v + std * noise[b]
        y[b] = v
    return y


def gaussian_taps(ksize, sigma):
    """Normalised 1-D Gaussian, float32 (the shape kornia's get_gaussian_kernel1d documents)."""
    r = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
    g = np.exp(-(r * r) / (2.0 * sigma * sigma))
    return (g / g.sum()).astype(np.float32)


def _reflect(i, n):
    i = np.abs(i)
    return np.where(i >= n, 2 * (n - 1) - i, i)


def gauss_blur(x, taps, apply=None):
    """x (B, C, H, W); separable, horizontal pass then vertical pass, reflect border; apply (B,) bool or None."""
    B, C, H, W = x.shape
    K = len(taps)
    half = K // 2
    out = x.astype(np.float32).copy()
    for b in range(B):
        if apply is not None and not apply[b]:
            continue
        for c in range(C):
            p = x[b, c].astype(np.float32)
            tmp = np.zeros_like(p)
            cols = np.arange(W)
            for t in range(K):
                tmp = tmp + taps[t] * p[:, _reflect(cols + t - half, W)]
            res = np.zeros_like(p)
            rows = np.arange(H)
            for t in range(K):
                res = res + taps[t] * tmp[_reflect(rows + t - half, H), :]
            out[b, c] = res
    return out


def rgb_to_grayscale(x3):
    """random_transform.py:91-92 on the expanded 3-channel copy of a grey image."""
    return (np.float32(0.299) * x3[:, 0:1] + np.float32(0.587) * x3[:, 1:2] + np.float32(0.114) * x3[:, 2:3]).astype(np.float32)


# ------------------------------------------------------------------------------------------------
# the module's three entry points, for given (already sampled) per-sample parameters
# ------------------------------------------------------------------------------------------------
def forward(x, transforms, photo_ops=()):
    """RandomTransform.forward (random_transform.py:76-94): x (B,1,H,W) in [0,1]; `transforms` = list of (B,3,3)
    forward matrices, applied in order with bilinear warps; the clear copy is taken after the geometric part; then the
    photometric modules in their configured order: ("photometric", params, noise) or ("blur", taps, apply)."""
    for m in transforms:
        minv = np.stack([dst_to_src(m[b]) for b in range(x.shape[0])])
        x = warp_image(x, minv)
    clear = x.copy()
    for op in photo_ops:
        if op[0] == "blur":
            x = gauss_blur(x, op[1], op[2])
        else:
            x = photometric(x, op[1], op[2])
    return x, clear


def forward_transform(ids, transforms):
    """random_transform.py:96-105: warp an id map INTO the view's frame."""
    for m in transforms:
        minv = np.stack([dst_to_src(m[b]) for b in range(ids.shape[0])])
        ids = warp_labels(ids, minv)
    return ids


def reverse_transform(ids, transforms):
    """random_transform.py:107-112: warp an id map of the view back to the un-augmented frame (inverse matrices,
    reverse order).  The destination->source matrix of an inverse warp is the forward matrix itself."""
    for m in reversed(transforms):
        ids = warp_labels(ids, np.asarray(m, dtype=np.float64).astype(np.float32))
    return ids
