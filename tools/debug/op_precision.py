"""Precision of single operators against fp64 CPU autograd (and ATen fp32 CPU beside it): where do 1e-3 gradient errors
come from?   python tools/debug/op_precision.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "medical-image-editing_amd")):
    sys.path.insert(0, p)
import torch, torch.nn.functional as F
from hipops import ops

def rel(a, b):
    return float((a.double().cpu() - b.double()).norm() / b.double().norm())

torch.manual_seed(0)
for (B, C, S, off) in [(2, 32, 32, 0.0), (2, 32, 32, 3.0), (2, 32, 256, 0.0), (2, 16, 64, 1.0)]:
    x = torch.randn(B, C, S, S) * 1.5 + off * torch.randn(1, C, 1, 1)
    r = torch.randn(B, C, S, S)
    outs = {}
    for tag, dt, dev in (("f64", torch.float64, "cpu"), ("aten32", torch.float32, "cpu")):
        xx = x.detach().clone().to(dt).requires_grad_(True)
        y = torch.relu(F.instance_norm(xx, eps=1e-5))
        (y * r.to(dt)).sum().backward()
        outs[tag] = (y.detach(), xx.grad)
    xx = x.detach().clone().cuda().requires_grad_(True)
    y = ops.instance_norm(xx, relu=True)
    (y * r.cuda()).sum().backward()
    print("instance_norm+relu B%d C%d S%d off %.0f: fwd hip %.2e aten32 %.2e | gx hip %.2e aten32 %.2e" % (
        B, C, S, off, rel(y.detach(), outs["f64"][0]), rel(outs["aten32"][0], outs["f64"][0]), rel(xx.grad, outs["f64"][1]), rel(outs["aten32"][1], outs["f64"][1])))

for (B, Ci, Co, S, ks, dil) in [(2, 32, 32, 32, 3, 1), (2, 160, 32, 32, 3, 1), (2, 32, 32, 32, 3, 6), (2, 32, 1, 32, 1, 1), (2, 16, 16, 64, 3, 1), (2, 256, 32, 32, 3, 1)]:
    x = torch.randn(B, Ci, S, S); w = torch.randn(Co, Ci, ks, ks) / (Ci * ks * ks) ** 0.5; b = torch.randn(Co); r = torch.randn(B, Co, S, S)
    res = {}
    for tag, dt in (("f64", torch.float64), ("aten32", torch.float32)):
        xx, ww, bb = x.detach().clone().to(dt).requires_grad_(True), w.detach().clone().to(dt).requires_grad_(True), b.detach().clone().to(dt).requires_grad_(True)
        y = F.conv2d(xx, ww, bb, padding=dil * (ks // 2), dilation=dil)
        (y * r.to(dt)).sum().backward()
        res[tag] = (y.detach(), xx.grad, ww.grad, bb.grad)
    xx = x.detach().clone().cuda().requires_grad_(True)
    ww = torch.nn.Parameter(w.cuda().contiguous(memory_format=torch.channels_last)); bb = torch.nn.Parameter(b.cuda())
    y = ops.conv2d(xx, ww, bb, dilation=dil)
    (y * r.cuda()).sum().backward()
    torch.cuda.synchronize()
    print("conv %dx%d d%d %d->%d @%d: y hip %.2e aten %.2e | gx %.2e %.2e | gw %.2e %.2e | gb %.2e %.2e" % (
        ks, ks, dil, Ci, Co, S, rel(y.detach(), res["f64"][0]), rel(res["aten32"][0], res["f64"][0]), rel(xx.grad, res["f64"][1]), rel(res["aten32"][1], res["f64"][1]),
        rel(ww.grad, res["f64"][2]), rel(res["aten32"][2], res["f64"][2]), rel(bb.grad, res["f64"][3]), rel(res["aten32"][3], res["f64"][3])))

# DoubleConv block (conv -> IN -> ReLU twice) and the decoder tail
from networks import blocks as Bk
torch.manual_seed(1)
for (Ci, Co, S) in [(32, 32, 32), (160, 32, 32), (16, 32, 64)]:
    m = Bk.DoubleConv(Ci, Co)
    x = torch.randn(2, Ci, S, S); r = torch.randn(2, Co, S, S)
    ref = {}
    for tag, dt in (("f64", torch.float64), ("aten32", torch.float32)):
        P = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in m.state_dict().items()}
        xx = x.detach().clone().to(dt).requires_grad_(True)
        h = torch.relu(F.instance_norm(F.conv2d(xx, P["double_conv.0.weight"], P["double_conv.0.bias"], padding=1), eps=1e-5))
        y = torch.relu(F.instance_norm(F.conv2d(h, P["double_conv.3.weight"], P["double_conv.3.bias"], padding=1), eps=1e-5))
        (y * r.to(dt)).sum().backward()
        ref[tag] = (y.detach(), xx.grad, P["double_conv.0.weight"].grad, P["double_conv.3.weight"].grad)
    mm = m.cuda().train()
    xx = x.detach().clone().cuda().requires_grad_(True)
    y = mm(xx)
    (y * r.cuda()).sum().backward()
    torch.cuda.synchronize()
    print("DoubleConv %d->%d @%d: y hip %.2e aten %.2e | gx %.2e %.2e | gw0 %.2e %.2e | gw3 %.2e %.2e" % (
        Ci, Co, S, rel(y.detach(), ref["f64"][0]), rel(ref["aten32"][0], ref["f64"][0]), rel(xx.grad, ref["f64"][1]), rel(ref["aten32"][1], ref["f64"][1]),
        rel(mm.double_conv[0].weight.grad, ref["f64"][2]), rel(ref["aten32"][2], ref["f64"][2]), rel(mm.double_conv[3].weight.grad, ref["f64"][3]), rel(ref["aten32"][3], ref["f64"][3])))
