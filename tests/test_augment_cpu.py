"""CPU checks of the augmentation oracle (oracle/augment_ref.py).  kornia is not available offline, so the oracle cannot
be pinned to the reference's output ("parity unpinned"); these are the known-answer and property checks that pin its
own statement: exact flips / whole-pixel shifts / quarter turns, inverse round trips, posterize levels, blur invariants."""
import numpy as np

from oracle import augment_ref as A


def _ids(B, H, W, seed=0):
    return np.random.default_rng(seed).integers(1, 11, size=(B, H, W)).astype(np.int64)


def test_exact_geometric_transforms():
    B, H, W = 3, 16, 16
    rng = np.random.default_rng(1)
    img = rng.random((B, 1, H, W), dtype=np.float32)
    ids = _ids(B, H, W)
    flip = np.stack([A.hflip_matrix(W)] * B)
    minv = np.stack([A.dst_to_src(m) for m in flip])
    assert np.array_equal(A.warp_image(img, minv), img[..., ::-1])
    assert np.array_equal(A.warp_labels(ids, minv), ids[..., ::-1].astype(np.int32))
    # whole-pixel shift by (+3, -2): content moves right/up, vacated pixels are 0
    sh = np.stack([A.affine_matrix(0.0, 3.0, -2.0, 0.0, 0.0, H, W)] * B)
    minv = np.stack([A.dst_to_src(m) for m in sh])
    out = A.warp_labels(ids, minv)
    assert np.array_equal(out[:, :H - 2, 3:], ids[:, 2:, :W - 3].astype(np.int32))
    assert not out[:, H - 2:, :].any() and not out[:, :, :3].any()
    # quarter turn about the centre of an even square image is exact: equals np.rot90
    rot = np.stack([A.affine_matrix(90.0, 0.0, 0.0, 0.0, 0.0, H, W)] * B)
    minv = np.stack([A.dst_to_src(m) for m in rot])
    assert np.array_equal(A.warp_labels(ids, minv), np.rot90(ids, k=1, axes=(1, 2)).astype(np.int32))
    assert np.allclose(A.warp_image(img, minv), np.rot90(img, k=1, axes=(2, 3)), atol=1e-5)


def test_reverse_then_forward_is_identity_inside_the_frame():
    B, H, W = 4, 48, 48
    rng = np.random.default_rng(2)
    mats = [np.stack([A.hflip_matrix(W) if b % 2 else A.identity_matrix() for b in range(B)]),
            np.stack([A.affine_matrix(rng.uniform(-25, 25), rng.uniform(-4, 4), rng.uniform(-4, 4), rng.uniform(-8, 8), 0.0, H, W)
                      for b in range(B)])]
    # a piecewise-constant map (like VQ ids of a smooth image) survives the two nearest-neighbour resamplings
    ids = (np.arange(H)[None, :, None] // 8 * 8 + np.arange(W)[None, None, :] // 8 + 1).repeat(B, 0).astype(np.int64)
    back = A.reverse_transform(A.forward_transform(ids, mats), mats)
    inside = back != 0
    assert inside.mean() > 0.5
    assert (back[inside] == ids[inside]).mean() > 0.9
    # exact transforms round-trip exactly
    exact = [mats[0], np.stack([A.affine_matrix(0.0, 2.0, 1.0, 0.0, 0.0, H, W)] * B)]
    rid = _ids(B, H, W, 5)
    back = A.reverse_transform(A.forward_transform(rid, exact), exact)
    inside = back != 0
    assert np.array_equal(back[inside], rid[inside].astype(np.int32)) and inside.mean() > 0.9


def test_photometric_known_answers():
    x = np.linspace(0.0, 1.0, 64, dtype=np.float32).reshape(1, 1, 8, 8)
    ident = np.array([[0.0, 1.0, 8, 0.0]], dtype=np.float32)
    assert np.array_equal(A.photometric(x, ident), x)
    one_bit = A.photometric(x, np.array([[0.0, 1.0, 1, 0.0]], dtype=np.float32))
    assert set(np.unique(one_bit)) == {np.float32(0.0), np.float32(128.0 / 255.0)}
    bright = A.photometric(x, np.array([[0.5, 1.0, 8, 0.0]], dtype=np.float32))
    assert bright.max() == 1.0 and np.allclose(bright[x < 0.5], x[x < 0.5] + 0.5)
    noise = np.ones_like(x)
    assert np.allclose(A.photometric(x, np.array([[0.0, 1.0, 8, 0.25]], dtype=np.float32), noise), x + 0.25)


def test_blur_invariants():
    taps = A.gaussian_taps(5, 1.3)
    assert abs(float(taps.sum()) - 1.0) < 1e-6 and np.allclose(taps, taps[::-1])
    const = np.full((2, 1, 12, 10), 0.37, dtype=np.float32)
    assert np.allclose(A.gauss_blur(const, taps), const, atol=1e-6)            # reflect border keeps a constant image
    x = np.random.default_rng(0).random((2, 1, 12, 10), dtype=np.float32)
    out = A.gauss_blur(x, taps, apply=np.array([1, 0], dtype=np.uint8))
    assert np.array_equal(out[1], x[1]) and not np.array_equal(out[0], x[0])
    assert abs(float(out[0].mean()) - float(x[0].mean())) < 0.02
