#!/usr/bin/env python3
"""Where does the host spend a data-parallel training step (one-rank RCCL group, every collective forced)?  Times the phases of
FirstStepTrainer.training_step on the host while the GPU runs behind.   VQW_DP_FORCE=1 python tools/dp_probe/host_phases.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
forced = os.environ.get("VQW_DP_FORCE", "0") == "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29674")
import torch, torch.distributed as dist
import bench
from hipops import ops
from trainers import build_first_step_trainer
from utils import load_json
if forced:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
cfg = load_json(os.path.join(ROOT, "configs", "baseline2_256x256_b32_1gpu.json"))
torch.manual_seed(0)
tr = build_first_step_trainer(cfg, device=torch.device("cuda", 0), data_parallel=forced)
pool = [bench.synthetic_batch(32, 256, 1234 + s, torch.device("cuda", 0)) for s in range(2)]
names = ["begin", "forward", "zero+prepare", "backward", "join", "finish", "optim", "end"]
acc = dict.fromkeys(names, 0.0)


def step(image, noise, rec):
    t = [time.perf_counter()]
    def mark(): t.append(time.perf_counter())
    tr.throttle.begin()
    if tr.reducer is not None:
        ops.reset_pending(tr._params)
    mark()
    out = tr.forward_losses(image, noise); mark()
    tr.enc_optim.zero_grad(); tr.dec_optim.zero_grad()
    if tr.reducer is not None:
        tr.reducer.prepare()
    mark()
    out["total"].backward(); mark()
    if tr._s2 is not None:
        torch.cuda.current_stream().wait_stream(tr._s2)
    ops.join_streams(); mark()
    if tr.reducer is not None:
        tr.reducer.finish()
    mark()
    tr.enc_optim.step(); tr.dec_optim.step(); mark()
    tr.throttle.end(); mark()
    if rec:
        for n, a, b in zip(names, t[:-1], t[1:]):
            acc[n] += (b - a) * 1e3


for i in range(4):
    step(pool[i % 2][0], pool[i % 2][1], False)
torch.cuda.synchronize()
n = 8
t0 = time.perf_counter()
for i in range(n):
    step(pool[i % 2][0], pool[i % 2][1], True)
torch.cuda.synchronize()
print("%s: %.1f ms/step; host ms per phase: %s" % ("forced one-rank RCCL" if forced else "plain", (time.perf_counter() - t0) * 1e3 / n,
                                                   ", ".join("%s %.1f" % (k, v / n) for k, v in acc.items())))
if forced:
    dist.destroy_process_group()
