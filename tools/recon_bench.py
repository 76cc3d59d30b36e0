#!/usr/bin/env python3
"""BASELINE config 5: run_recon mask-guided reconstruction (ids -> lookup -> mask*rescale -> decoder, eval mode),
256x256, batch 64, 1 GPU.  Prints images/sec."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
import torch
from networks import UNetEncoder, UNetDecoder
from run_recon import reconstruct
torch.manual_seed(0)
enc = UNetEncoder(1, [16, 32, 64, 128, 256], 10, 0.999, 'torch', False, 1, True).cuda().eval()
dec = UNetDecoder(16, 1, [32, 64, 128, 256, 512], use_dropblock=False, dropped_skip_layers=[], use_pixel_shuffle=False).cuda().eval()
B, S = 64, 256
g = torch.Generator().manual_seed(1)
lab = torch.randint(1, 11, (B, S, S), generator=g)
lab[torch.rand(B, S, S, generator=g) < 0.1] = 0          # 10 % masked pixels
lab = lab.cuda()
for _ in range(2):
    reconstruct(enc, dec, lab)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 5
for _ in range(n):
    rec = reconstruct(enc, dec, lab)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("config 5 (run_recon, batch %d at %dx%d): %.1f ms/batch = %.0f images/s; decoder fwd 74.7 GFLOP/img -> %.1f TFLOP/s"
      % (B, S, S, dt * 1e3, B / dt, 74.7e9 * B / dt / 1e12))
