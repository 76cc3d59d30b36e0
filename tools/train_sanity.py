#!/usr/bin/env python3
"""Does the first step actually train?  Runs N first-step iterations on a small fixed set of synthetic slices and
prints the loss trajectory (reconstruction MSE must fall, nothing may become NaN, codebook usage must stay > 1 code).

    python tools/train_sanity.py [--steps 300] [--size 64] [--batch 8]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from trainers import FirstStepTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--lr", type=float, default=2e-4)
    a = ap.parse_args()
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    tr = FirstStepTrainer(device=dev, lr=a.lr)
    pool = [bench.synthetic_batch(a.batch, a.size, 100 + s, dev) for s in range(4)]
    hist = []
    for i in range(a.steps):
        img, noise = pool[i % len(pool)]
        out = tr.training_step({"image": img}, noise=noise)
        if i % 25 == 0 or i == a.steps - 1:
            sc = tr.scalars(out)
            used = int(torch.unique(out["ids_1"]).numel())
            hist.append((i, sc["total"], sc["recon"], sc["commit"], sc["cross"], used))
            print("step %4d  total %.4f  recon %.4f  commit %.4f  cross %.4f  codes used %d" % hist[-1], flush=True)
    first, last = hist[0], hist[-1]
    ok = all(v == v for h in hist for v in h[1:5]) and last[2] < 0.6 * first[2] and last[5] > 1
    print("recon %.4f -> %.4f (%.0f %%), %s" % (first[2], last[2], 100 * last[2] / first[2], "OK" if ok else "NOT LEARNING"))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
