#!/usr/bin/env python3
"""Wall time of the phases of one training step (forward+losses / backward / optimiser), B=32 at 256x256."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
import torch
from trainers import FirstStepTrainer
from bench import synthetic_batch
torch.manual_seed(0)
tr = FirstStepTrainer(device="cuda")
img, noise = synthetic_batch(32, 256, 1234, "cuda")
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(4):
    t0 = sync(); out = tr.forward_losses(img, noise)
    t1 = sync(); tr.enc_optim.zero_grad(); tr.dec_optim.zero_grad(); out["total"].backward()
    t2 = sync(); tr.enc_optim.step(); tr.dec_optim.step()
    t3 = sync()
    # host-side enqueue time of a whole step (no syncs inside)
    h0 = time.perf_counter(); tr.training_step({"image": img}, noise=noise); h1 = time.perf_counter(); t4 = sync()
    print("iter %d: forward %.1f ms  backward %.1f ms  optimiser %.1f ms | async step: host enqueue %.1f ms, total %.1f ms"
          % (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (h1 - h0) * 1e3, (t4 - h0) * 1e3), flush=True)
print("peak memory %.1f GB" % (torch.cuda.max_memory_allocated() / 2**30))
