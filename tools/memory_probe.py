#!/usr/bin/env python3
"""Device memory of the training step at the bench configuration: peak of live tensors and the caching allocator's pool
once the host throttle (two steps in flight) has settled.

    python tools/memory_probe.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from trainers import FirstStepTrainer  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
tr = FirstStepTrainer(device=dev)
img, noise = bench.synthetic_batch(32, 256, 1234, dev)
for i in range(12):
    tr.training_step({"image": img}, noise=noise)
torch.cuda.synchronize()
print("after 12 steps (B=32, 256x256): peak allocated %.1f GB, reserved pool %.1f GB" %
      (torch.cuda.max_memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30))
