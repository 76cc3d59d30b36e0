#!/usr/bin/env python3
"""Timing of the InstanceNorm forward / backward operators on the model's largest shapes (B=32).

    python tools/norm_bench.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
import torch  # noqa: E402
from hipops import ops  # noqa: E402

dev = "cuda"
SHAPES = ((32, 256), (16, 256), (64, 128), (128, 64), (256, 32))
if len(sys.argv) > 2:          # one shape: norm_bench.py C S   (under rocprofv3 --stats: per-kernel durations of that shape,
    SHAPES = ((int(sys.argv[1]), int(sys.argv[2])),)      # with a plain copy and a plain add of the same tensors beside them)
    C, S = SHAPES[0]
    a = torch.randn(32, C, S, S, device=dev).contiguous(memory_format=torch.channels_last)
    b = torch.randn_like(a)
    c = torch.empty_like(a)
    for _ in range(20):
        c.copy_(a)
        torch.add(a, b, out=c)
        ops.add(a, b)
for C, S in SHAPES:
    x = torch.randn(32, C, S, S, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    g = torch.randn_like(x)
    for _ in range(3):
        y = ops.instance_norm(x, relu=True)
        y.backward(g)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    n = 20
    e[0].record()
    ys = [ops.instance_norm(x, relu=True) for _ in range(n)]
    e[1].record()
    for y in ys:
        y.backward(g)
    e[2].record()
    torch.cuda.synchronize()
    mb = x.numel() * 4 / 1e6
    f, b = e[0].elapsed_time(e[1]) / n, e[1].elapsed_time(e[2]) / n
    print("C=%3d @%3d (%6.1f MB): fwd %.3f ms = %.2f TB/s over 3 passes, bwd %.3f ms = %.2f TB/s over 5 passes" %
          (C, S, mb, f, 3 * mb / f / 1e3, b, 5 * mb / b / 1e3))
