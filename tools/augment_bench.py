#!/usr/bin/env python3
"""HBM roofline of the augmentation kernels at the bench shape: B=32, 1x256x256, two views.

    python tools/augment_bench.py [--batch 32] [--size 256] [--iters 50]
Prints achieved GB/s per kernel against the 8 TB/s HBM peak (algorithmic bytes = tensors read + written once).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
import torch  # noqa: E402
from hipops import ops  # noqa: E402
from networks import RandomTransform  # noqa: E402

PEAK = 8.0e12


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    B, S = a.batch, a.size
    dev = "cuda"
    x = torch.rand(B, 1, S, S, device=dev)
    ids = torch.randint(1, 11, (B, S, S), device=dev)
    m = torch.eye(3, device=dev).repeat(B, 1, 1)
    m[:, 0, 0] = -1.0
    m[:, 0, 2] = S - 1.0
    par = torch.tensor([[0.1, 1.1, 5.0, 0.05]], device=dev).repeat(B, 1)
    noise = torch.randn_like(x)
    taps = torch.tensor([0.06, 0.24, 0.4, 0.24, 0.06], device=dev)
    n = x.numel() * 4
    rows = [("warp_image (bilinear)", lambda: ops.warp_image(x, m), 2 * n),
            ("warp_labels (int64 -> int32)", lambda: ops.warp_labels(ids, m), ids.numel() * 12),
            ("photometric (+noise)", lambda: ops.photometric(x, par, noise), 3 * n),
            ("gauss_blur k=5 (2 passes)", lambda: ops.gauss_blur(x, taps), 4 * n)]
    for name, fn, nbytes in rows:
        t = timeit(fn, a.iters)
        print("%-30s %8.1f us  %7.1f GB/s  %5.1f %% of HBM peak" % (name, t * 1e6, nbytes / t / 1e9, 100 * nbytes / t / PEAK))
    cfg = dict(modules=["RandomHorizontalFlip", "RandomAffine", "ColorJitter", "RandomGaussianBlur", "RandomPosterize",
                        "RandomGaussianNoise"],
               RandomHorizontalFlip=dict(p=0.5), RandomAffine=dict(degrees=20.0, translate=(0.1, 0.1), shear=8.0, p=0.8),
               ColorJitter=dict(brightness=0.2, contrast=0.2, p=0.8), RandomGaussianBlur=dict(kernel=5, sigma=1.2, p=0.5),
               RandomPosterize=dict(bits=4, p=0.3), RandomGaussianNoise=dict(std=0.05, p=0.5))
    t1, t2 = RandomTransform(cfg, 1), RandomTransform(cfg, 2)

    def both():
        t1(x); t2(x)
        t2.forward_transform(t1.reverse_transform(ids)); t1.forward_transform(t2.reverse_transform(ids))
    t = timeit(both, 10)
    print("two views + both cross id maps, host sampling included: %.2f ms per step" % (t * 1e3))


if __name__ == "__main__":
    main()
