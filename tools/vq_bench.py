#!/usr/bin/env python3
"""Timing of the vector-quantiser entry point (vqw_vq_fwd + vqw_vq_ema_update) at the BASELINE shapes.

    python tools/vq_bench.py [--case cfg4|cfg2|all] [--iters 20] [--eval]

cfg4: K = 1024, D = 256, 2 x 512 x 512 pixels (BASELINE config 4, one view of the per-GPU batch of 2).
cfg2: K = 10, D = 16, 32 x 256 x 256 pixels (BASELINE config 2, one view).
Prints per case: ms per call (HIP events on the launch stream), algorithmic GFLOP / MB (2 N D K flop,
2 N D 4 + N 8 bytes) and the achieved TFLOP/s and TB/s.  Run under `rocprofv3 --kernel-trace --stats` for the
per-kernel split; `--pmc` passes for HBM traffic (tools/pmc_traffic.py).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
import torch  # noqa: E402
from hipops import ops  # noqa: E402

CASES = {"cfg4": (1024, 256, 2, 512), "cfg2": (10, 16, 32, 256), "k64": (64, 32, 8, 256), "k1024d64": (1024, 64, 4, 256)}


def synthetic(B, D, S, K, seed=0):
    """Spatially smooth features (SURVEY 8d's slice generator per channel) and a codebook drawn from them, so the
    assignments spread over the codes as in a trained model."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    low = torch.randn(B, D, S // 8, S // 8, generator=g, device="cuda")
    x = torch.nn.functional.interpolate(low, size=(S, S), mode="bilinear", align_corners=False)
    x = (x * 0.6 + 0.05 * torch.randn(B, D, S, S, generator=g, device="cuda")).contiguous(memory_format=torch.channels_last)
    xf = x.permute(0, 2, 3, 1).reshape(-1, D)
    pick = torch.randint(0, xf.shape[0], (K,), generator=g, device="cuda")
    embed = (xf[pick] + 0.01 * torch.randn(K, D, generator=g, device="cuda")).contiguous()
    return x, embed


def run(case, iters, training):
    K, D, B, S = CASES[case]
    x, embed = synthetic(B, D, S, K)
    cs = torch.full((K,), float(B * S * S) / K, device="cuda")
    ea = (embed * cs[:, None]).t().contiguous()
    N = B * S * S
    for _ in range(3):
        q, c, ids = ops.vq_quantize(x, embed.clone(), cs.clone(), ea.clone(), training, 0.99, 1e-5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    bufs = [(embed.clone(), cs.clone(), ea.clone()) for _ in range(iters)]
    torch.cuda.synchronize()
    e0.record()
    for i in range(iters):
        q, c, ids = ops.vq_quantize(x, bufs[i][0], bufs[i][1], bufs[i][2], training, 0.99, 1e-5)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    used = int(torch.unique(ids).numel())
    flop = 2.0 * N * D * K
    byts = 2.0 * N * D * 4 + N * 8
    return dict(case=case, K=K, D=D, pixels=N, training=training, ms=round(ms, 4), codes_used=used,
                gflop=round(flop / 1e9, 2), mbytes=round(byts / 1e6, 1),
                tflops=round(flop / ms / 1e9, 2), tbytes_s=round(byts / ms / 1e9, 3),
                frac_fp32_mfma=round(flop / ms / 1e9 / 157.3, 3), frac_hbm=round(byts / ms / 1e9 / 8.0, 3))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="all")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--eval", action="store_true")
    a = ap.parse_args()
    for case in (CASES if a.case == "all" else [a.case]):
        print(json.dumps(run(case, a.iters, not a.eval)), flush=True)
