#!/usr/bin/env python3
"""Host time spent INSIDE dist.all_reduce for the four gradient buckets of a step in a one-rank RCCL group, with the GPU a full
backward pass behind the host: does the call return at once (asynchronous) or only when the GPU gets there?
    VQW_DP_FORCE=1 VQW_DP_HOST_TIMING=1 python tools/dp_probe/allreduce_host_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
os.environ.setdefault("VQW_DP_FORCE", "1"); os.environ.setdefault("VQW_DP_HOST_TIMING", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29673")
import torch, torch.distributed as dist
import bench
from trainers import build_first_step_trainer
from utils import load_json
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
cfg = load_json(os.path.join(ROOT, "configs", "baseline2_256x256_b32_1gpu.json"))
torch.manual_seed(0)
tr = build_first_step_trainer(cfg, device=torch.device("cuda", 0), data_parallel=True)
pool = [bench.synthetic_batch(32, 256, 1234 + s, torch.device("cuda", 0)) for s in range(2)]
for i in range(4):
    tr.training_step({"image": pool[i % 2][0]}, noise=pool[i % 2][1])
torch.cuda.synchronize()
red = tr.reducer
n = 8
red.__class__.host_ms_in_all_reduce = 0.0; red.host_ms_in_all_reduce = 0.0
t0 = time.perf_counter()
for i in range(n):
    tr.training_step({"image": pool[i % 2][0]}, noise=pool[i % 2][1])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("overlap schedule %s: %.1f ms/step; host returns from a step's enqueue after %.1f ms; host time inside the 4 dist.all_reduce calls %.2f ms/step"
      % (red.overlap, (t2 - t0) * 1e3 / n, (t1 - t0) * 1e3 / n, red.host_ms_in_all_reduce / n))
# the same four calls with an idle GPU
flat = [torch.zeros(4 << 20, device="cuda") for _ in range(4)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for f in flat:
    dist.all_reduce(f, async_op=True).wait()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("four 16 MiB all-reduces, GPU idle: host %.3f ms, until done %.3f ms" % ((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
dist.destroy_process_group()
