#!/usr/bin/env python3
"""Which backward nodes produce the gradient contributions that autograd has to ADD to an existing one (aten::add_ inside the
engine's input buffers): shape of the sum, the node that delivered the second contribution, how often per step."""
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
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    tr.training_step({"image": img}, noise=noise)
    torch.cuda.synchronize()
agg = collections.Counter()
for e in prof.events():
    if e.name == "aten::add_" and e.input_shapes and len(e.input_shapes[0]) == 4:
        p = e.cpu_parent
        names = []
        while p is not None and len(names) < 3:
            names.append(p.name.replace("autograd::engine::evaluate_function: ", ""))
            p = p.cpu_parent
        agg[(str(e.input_shapes[0]), " <- ".join(names))] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(v, k)
