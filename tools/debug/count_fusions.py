#!/usr/bin/env python3
"""Per training step: how many launches took each epilogue fusion (counters of hipops.ops)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
import torch
import bench
from hipops import ops
from trainers import build_first_step_trainer
from utils import load_json
cfg = load_json(os.path.join(ROOT, "configs", "baseline2_256x256_b32_1gpu.json"))
tr = build_first_step_trainer(cfg, device=torch.device("cuda", 0))
img, noise = bench.synthetic_batch(32, 256, 1, torch.device("cuda", 0))
for _ in range(2):
    tr.training_step({"image": img}, noise=noise)
torch.cuda.synchronize()
c0 = (ops.masked_dgrad_calls, ops.group_acc_calls, ops.in_bwd_fused_calls)
tr.training_step({"image": img}, noise=noise)
torch.cuda.synchronize()
print("per step: ReLU masks in an input-gradient epilogue %d, gradient-group sums in an epilogue %d, InstanceNorm backward sums from a convolution's epilogue %d"
      % (ops.masked_dgrad_calls - c0[0], ops.group_acc_calls - c0[1], ops.in_bwd_fused_calls - c0[2]))
