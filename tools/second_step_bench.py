#!/usr/bin/env python3
"""Throughput of the second training step (frozen encoder -> decoder -> PatchGAN; generator then discriminator update)
at the bench shape.   python tools/second_step_bench.py [--batch 32] [--size 256] [--steps 5]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
import torch  # noqa: E402
from networks import UNetEncoder, UNetDecoder, NLayerDiscriminator  # noqa: E402
from trainers import SecondStepTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    torch.manual_seed(0)
    enc = UNetEncoder(1, [16, 32, 64, 128, 256], 10, 0.999, 'torch', False, 1, True)
    dec = UNetDecoder(16, 1, [32, 64, 128, 256, 512], use_dropblock=False, dropped_skip_layers=[], use_styled_up_block=True,
                      use_pixel_shuffle=False)
    tr = SecondStepTrainer(enc, dec, NLayerDiscriminator(), device="cuda")
    img = torch.rand(a.batch, 1, a.size, a.size, device="cuda") * 2 - 1
    for _ in range(2):
        tr.training_step(img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = tr.training_step(img)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print("second step: %.1f ms/step, %.1f images/s (B=%d, %dx%d); gen_total %.4f dis_total %.4f"
          % (dt * 1e3, a.batch / dt, a.batch, a.size, a.size, float(out["gen_total"].detach()), float(out["dis_total"].detach())))


if __name__ == "__main__":
    main()
