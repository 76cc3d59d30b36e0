"""Forward error of the direct and the Winograd form of a plain 3x3 layer against an fp64 convolution.
    python tools/debug/wino_precision.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))
import torch
import torch.nn.functional as F
from hipops import ops

torch.manual_seed(0)
for (Cin, Cout, S, mean) in [(32, 32, 64, 0.0), (64, 64, 64, 0.0), (256, 256, 32, 0.0), (64, 64, 64, 1.0), (64, 64, 64, 3.0)]:
    x = torch.randn(2, Cin, S, S) + mean
    if mean:
        x = torch.relu(x)
    w = torch.randn(Cout, Cin, 3, 3) / (9 * Cin) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    xd = x.cuda()
    wd = w.cuda().contiguous(memory_format=torch.channels_last)
    res = {}
    for name, flag in (("direct", False), ("winograd", True)):
        ops.WINOGRAD_FWD = flag
        y = ops.conv2d(xd, wd).double().cpu()
        e = y - ref
        res[name] = (float(e.norm() / ref.norm()), float(e.abs().max() / ref.abs().max()))
    cpu = F.conv2d(x, w, padding=1).double() - ref
    print("Cin %3d Cout %3d %dx%d input mean %.0f: direct rel %.2e max %.2e | winograd rel %.2e max %.2e | torch cpu fp32 rel %.2e"
          % (Cin, Cout, S, S, mean, *res["direct"], *res["winograd"], float(cpu.norm() / ref.norm())))
