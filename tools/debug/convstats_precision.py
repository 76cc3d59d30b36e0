"""InstanceNorm fed by the conv epilogue's per-tile statistics vs reducing the plane itself, against fp64, on a
piecewise-constant (quantised-like) input and on a random one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "medical-image-editing_amd")):
    sys.path.insert(0, p)
import torch, torch.nn.functional as F
from hipops import ops

def rel(a, b):
    return float((a.double().cpu() - b.double()).norm() / (b.double().norm() + 1e-300))

torch.manual_seed(0)
for (Ci, Co, S, B, ks) in [(16, 16, 32, 2, 3), (16, 16, 32, 2, 1), (32, 32, 32, 2, 3), (16, 16, 64, 2, 3), (64, 64, 32, 2, 3), (16, 32, 256, 2, 3)]:
    for kind in ("random", "piecewise", "offset"):
        if kind == "piecewise":
            codes = torch.randn(6, Ci)
            ids = (torch.arange(S)[None, :, None] // 6 + torch.arange(S)[None, None, :] // 5 + torch.arange(B)[:, None, None]) % 6
            x = codes[ids].permute(0, 3, 1, 2).contiguous()
        elif kind == "offset":
            x = torch.randn(B, Ci, S, S) * 0.05 + 3.0 * torch.randn(1, Ci, 1, 1)
        else:
            x = torch.randn(B, Ci, S, S)
        w = torch.randn(Co, Ci, ks, ks) / (Ci * ks * ks) ** 0.5; b = torch.randn(Co); r = torch.randn(B, Co, S, S)
        xx, ww, bb = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
        y64 = torch.relu(F.instance_norm(F.conv2d(xx, ww, bb, padding=ks // 2), eps=1e-5))
        (y64 * r.double()).sum().backward()
        x32, w32, b32 = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y32 = torch.relu(F.instance_norm(F.conv2d(x32, w32, b32, padding=ks // 2), eps=1e-5))
        (y32 * r).sum().backward()
        aten = (rel(y32.detach(), y64.detach()), rel(x32.grad, xx.grad), rel(w32.grad, ww.grad))
        out = []
        for stats in (True, False):
            xc = x.clone().cuda().requires_grad_(True)
            wc = torch.nn.Parameter(w.cuda().contiguous(memory_format=torch.channels_last)); bc = torch.nn.Parameter(b.cuda())
            if stats:
                c, part = ops.conv2d(xc, wc, bc, want_stats=True)
                y = ops.instance_norm(c, relu=True, part=part)
            else:
                y = ops.instance_norm(ops.conv2d(xc, wc, bc), relu=True)
            (y * r.cuda()).sum().backward()
            torch.cuda.synchronize()
            out.append((rel(y.detach(), y64.detach()), rel(xc.grad, xx.grad), rel(wc.grad, ww.grad), part is not None if stats else None))
        print("%3d->%3d k%d @%3d %-9s epilogue stats(%s): y %.1e gx %.1e gw %.1e | own reduction: y %.1e gx %.1e gw %.1e | ATen fp32: y %.1e gx %.1e gw %.1e" % (
            Ci, Co, ks, S, kind, out[0][3], out[0][0], out[0][1], out[0][2], out[1][0], out[1][1], out[1][2], *aten))
