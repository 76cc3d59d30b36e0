#!/usr/bin/env python3
"""Host enqueue time vs GPU time of one training step, plain and with the RCCL data-parallel path forced on in a one-rank group
(VQW_DP_FORCE=1): where the data-parallel machinery's single-GPU cost comes from.   python tools/dp_probe/host_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
import torch, torch.distributed as dist
import bench
from trainers import build_first_step_trainer
from utils import load_json
forced = os.environ.get("VQW_DP_FORCE", "0") == "1"
if forced:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29671")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
cfg = load_json(os.path.join(ROOT, "configs", "baseline2_256x256_b32_1gpu.json"))
torch.manual_seed(0)
tr = build_first_step_trainer(cfg, device=torch.device("cuda", 0), data_parallel=forced)
pool = [bench.synthetic_batch(32, 256, 1234 + s, torch.device("cuda", 0)) for s in range(2)]
for i in range(4):
    tr.training_step({"image": pool[i % 2][0]}, noise=pool[i % 2][1])
torch.cuda.synchronize()
host, total = [], []
for i in range(6):
    t0 = time.perf_counter()
    tr.training_step({"image": pool[i % 2][0]}, noise=pool[i % 2][1])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
print("forced" if forced else "plain", "host enqueue ms/step", [round(h, 1) for h in host], "| step alone (sync each) ms", [round(t, 1) for t in total])
if forced:
    dist.destroy_process_group()
