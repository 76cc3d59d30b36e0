#!/usr/bin/env python3
"""BASELINE config 4 on one GPU's share of it: 512x512 CT-style slices, dict_size 1024, emb_dim 256
(configs/baseline4_ct512_k1024_dp8.json: enc_filters [256,64,128,256,512], batch 2 per GPU), the whole first training step
through trainers.build_first_step_trainer.  Prints one JSON line: ms/step, images/s, codes in use, peak memory.

    python tools/config4_bench.py [--steps 5] [--warmup 2]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
import torch
import bench
from trainers import build_first_step_trainer
from utils import load_json

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
a = ap.parse_args()
cfg = load_json(os.path.join(ROOT, "configs", "baseline4_ct512_k1024_dp8.json"))
B, S, K = int(cfg.dataset.batch_size), int(cfg.dataset.image_size), int(cfg.model.vqmodel.dict_size)
torch.manual_seed(0)
tr = build_first_step_trainer(cfg, device="cuda", data_parallel=False)
with torch.no_grad():          # checkpoint-like VQ state: every code in use (a cold random codebook collapses onto a few codes)
    tr.encoder.vq.cluster_size.fill_(B * S * S / K)
    tr.encoder.vq.embed_avg.copy_(tr.encoder.vq.embed.t() * tr.encoder.vq.cluster_size[None, :])
pool = [bench.synthetic_batch(B, S, 1234 + s, torch.device("cuda")) for s in range(2)]
for i in range(a.warmup):
    out = tr.training_step({"image": pool[i % 2][0]}, noise=pool[i % 2][1])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps):
    out = tr.training_step({"image": pool[i % 2][0]}, noise=pool[i % 2][1])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
sc = tr.scalars(out)
print(json.dumps(dict(workload="BASELINE config 4, one GPU's share: first training step, 512x512, batch %d, dict_size %d x emb_dim %d" % (B, K, cfg.model.vqmodel.enc_filters[0]),
                      ms_per_step=round(dt * 1e3, 2), images_per_sec=round(B / dt, 3), steps=a.steps,
                      codes_used_view1=int(torch.unique(out["ids_1"]).numel()), loss_total=sc["total"], recon=sc["recon"], commit=sc["commit"],
                      peak_memory_gb=round(torch.cuda.max_memory_allocated() / 1e9, 1))))
