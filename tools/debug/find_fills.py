#!/usr/bin/env python3
"""Which Python call sites issue aten::zero_ / aten::fill_ / aten::zeros inside a training step (torch profiler with stacks)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
import torch
import bench
from trainers import build_first_step_trainer
from utils import load_json
cfg = load_json(os.path.join(ROOT, "configs", "baseline2_256x256_b32_1gpu.json"))
tr = build_first_step_trainer(cfg, device=torch.device("cuda", 0))
img, noise = bench.synthetic_batch(32, 256, 1, torch.device("cuda", 0))
for _ in range(2):
    tr.training_step({"image": img}, noise=noise)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    tr.training_step({"image": img}, noise=noise)
    torch.cuda.synchronize()
NAMES = tuple(sys.argv[1:]) or ("aten::zero_", "aten::fill_", "aten::zeros", "aten::zeros_like", "aten::ones_like", "aten::full")
agg = collections.Counter()
for e in prof.events():
    if e.name in NAMES:
        st = [s for s in (e.stack or []) if "site-packages/torch" not in s and "dist-packages/torch" not in s][:2]
        agg[(e.name, str(e.input_shapes)[:60], " <- ".join(s[-70:] for s in st))] += 1
for k, v in agg.most_common(25):
    print(v, k)
