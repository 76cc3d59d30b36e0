#!/usr/bin/env python3
"""How far ahead of the GPU is the host?  Times the host side of training_step (enqueue only, no sync) against the
GPU step time.  If the two are close, the step is launch-bound in places and a HIP graph would pay.

    python tools/host_enqueue_probe.py [--steps 10]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from trainers import FirstStepTrainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
tr = FirstStepTrainer(device=dev)
pool = [bench.synthetic_batch(32, 256, 1234 + s, dev) for s in range(4)]
chain = torch.cuda.Stream(device=dev, priority=-1)
for i in range(3):
    with torch.cuda.stream(chain):
        tr.training_step({"image": pool[i % 4][0]}, noise=pool[i % 4][1])
    torch.cuda.synchronize()
host, fwd = [], []
t0 = time.perf_counter()
for i in range(args.steps):
    img, noise = pool[i % 4]
    a = time.perf_counter()
    with torch.cuda.stream(chain):
        tr.training_step({"image": img}, noise=noise)
    host.append(time.perf_counter() - a)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("GPU step %.1f ms; host enqueue per step: mean %.1f ms, first %.1f, last %.1f (steps queue behind each other: "
      "the first step's host time is the un-throttled one)" % (tot / args.steps * 1e3, sum(host) / len(host) * 1e3, host[0] * 1e3, host[-1] * 1e3))
# host-only phases of one step, un-throttled (GPU idle at start)
torch.cuda.synchronize()
img, noise = pool[0]
with torch.cuda.stream(chain):
    a = time.perf_counter()
    out = tr.forward_losses(img, noise)
    b = time.perf_counter()
    tr.enc_optim.zero_grad(); tr.dec_optim.zero_grad()
    out["total"].backward()
    c = time.perf_counter()
    tr.enc_optim.step(); tr.dec_optim.step()
    d = time.perf_counter()
torch.cuda.synchronize()
e = time.perf_counter()
print("host: forward %.1f ms, backward %.1f ms, optimiser %.1f ms; GPU done %.1f ms after the host" %
      ((b - a) * 1e3, (c - b) * 1e3, (d - c) * 1e3, (e - d) * 1e3))
