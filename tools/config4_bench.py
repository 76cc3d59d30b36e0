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
pool = [bench.synthetic_batch(B, S, 1234 + s, torch.device("cuda")) for s in range(2)]
D = int(cfg.model.vqmodel.enc_filters[0])
with torch.no_grad():
    # A representative assignment: the codebook is initialised the way the reference initialises it (UNetEncoder.initialize_embed,
    # unet_encoder.py:66-91: k-means over the features of a batch; a cold random 256-d codebook collapses onto ONE code and the
    # deterministic sort / segment-sum statistics would only be timed in their one-code worst case), then given a
    # checkpoint-like EMA state (cluster_size = pixels / K, embed_avg consistent with it).
    from hipops import ops
    feat = tr.encoder.feature_extraction(pool[0][0])
    rows = feat.permute(0, 2, 3, 1).reshape(-1, D)
    centres, hist = ops.kmeans_codebook(rows, K, seed=0, max_iter=8)
    tr.encoder.vq.embed.copy_(centres)
    tr.encoder.vq.cluster_size.fill_(B * S * S / K)
    tr.encoder.vq.embed_avg.copy_(tr.encoder.vq.embed.t() * tr.encoder.vq.cluster_size[None, :])
    tr.encoder.init_embed = True
for i in range(a.warmup):
    out = tr.training_step({"image": pool[i % 2][0]}, noise=pool[i % 2][1])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps):
    out = tr.training_step({"image": pool[i % 2][0]}, noise=pool[i % 2][1])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
sc = tr.scalars(out)
# the VQ call alone (search fused with the arg-max + deterministic statistics + EMA update) on the step's features: its share of
# the fp32-MFMA roofline (2 N D K FLOP per call; 157.3 TFLOP/s)
with torch.no_grad():
    feat = tr.encoder.feature_extraction(pool[0][0]).clone()
tr.encoder.train()
vq_state = {k: v.clone() for k, v in tr.encoder.vq.state_dict().items()}
for _ in range(3):
    tr.encoder.vq(feat, id_base=1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
nvq = 10
for _ in range(nvq):
    q, commit, ids = tr.encoder.vq(feat, id_base=1)
e1.record()
torch.cuda.synchronize()
tr.encoder.vq.load_state_dict(vq_state)
vq_ms = e0.elapsed_time(e1) / nvq
Npix = B * S * S
vq_flop = 2.0 * Npix * D * K
print(json.dumps(dict(workload="BASELINE config 4, one GPU's share: first training step, 512x512, batch %d, dict_size %d x emb_dim %d" % (B, K, cfg.model.vqmodel.enc_filters[0]),
                      ms_per_step=round(dt * 1e3, 2), images_per_sec=round(B / dt, 3), steps=a.steps,
                      codes_used_view1=int(torch.unique(out["ids_1"]).numel()), loss_total=sc["total"], recon=sc["recon"], commit=sc["commit"],
                      codes_used_view2=int(torch.unique(out["ids_2"]).numel()), kmeans_iterations=len(hist),
                      roofline=dict(bound="mfma", kernel="VQ training call (k_vq_mfma search + arg-max, counting-sort statistics, EMA)",
                                    ms_per_call=round(vq_ms, 3), achieved=round(vq_flop / (vq_ms * 1e-3) / 1e12, 1), peak=157.3,
                                    unit="TFLOP/s", frac=round(vq_flop / (vq_ms * 1e-3) / 157.3e12, 3), flop_per_call=vq_flop,
                                    codes_used=int(torch.unique(ids).numel())),
                      eight_gpu_note="BASELINE config 4 is this per-rank workload x 8 ranks (global batch 16): weak scaling, the "
                                     "per-rank line is what one of the 8 GPUs runs between its collectives",
                      peak_memory_gb=round(torch.cuda.max_memory_allocated() / 1e9, 1))))
