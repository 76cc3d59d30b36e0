"""3x3 conv over a nearest x2 up-sampled input (the collapsed low-resolution form) against fp64 F.conv2d, small grids."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "medical-image-editing_amd")):
    sys.path.insert(0, p)
import torch, torch.nn.functional as F
from hipops import ops

def rel(a, b):
    return float((a.double().cpu() - b.double()).norm() / (b.double().norm() + 1e-300))

torch.manual_seed(0)
for B in (2, 3):
    for (Ci, Co) in [(32, 32), (64, 32), (32, 16), (16, 16), (128, 64)]:
        for h in (2, 4, 8, 16):
            x = torch.randn(B, Ci, h, h); w = torch.randn(Co, Ci, 3, 3) / (Ci * 9) ** 0.5; b = torch.randn(Co); r = torch.randn(B, Co, 2 * h, 2 * h)
            xx, ww = x.double().requires_grad_(True), w.double().requires_grad_(True)
            y = F.conv2d(F.interpolate(xx, scale_factor=2, mode="nearest"), ww, b.double(), padding=1)
            (y * r.double()).sum().backward()
            xc = x.clone().cuda().requires_grad_(True)
            wc = torch.nn.Parameter(w.cuda().contiguous(memory_format=torch.channels_last)); bc = torch.nn.Parameter(b.cuda())
            yc = ops.conv2d(xc, wc, bc, up2x=True)
            (yc * r.cuda()).sum().backward()
            torch.cuda.synchronize()
            e = (rel(yc.detach(), y.detach()), rel(xc.grad, xx.grad), rel(wc.grad, ww.grad))
            print("B%d %3d->%3d low %2dx%-2d  y %.1e  gx %.1e  gw %.1e %s" % (B, Ci, Co, h, h, *e, "   <-----" if max(e) > 1e-5 else ""))
