"""StyledResUpBlock (blocks.py:93-134) on the HIP path against the oracle in fp64, small grids; then its pieces."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "medical-image-editing_amd")):
    sys.path.insert(0, p)
import torch, torch.nn.functional as F
from networks import blocks as Bk
from oracle import vqwnet_ref as O

def rel(a, b):
    return float((a.double().cpu() - b.double()).norm() / (b.double().norm() + 1e-300))

torch.manual_seed(0)
for (Cin, Cs, Co, h, B) in [(32, 32, 32, 4, 2), (64, 32, 32, 2, 2), (32, 32, 32, 8, 2), (32, 16, 16, 16, 2), (32, 32, 32, 4, 8)]:
    m = Bk.StyledResUpBlock(Cin, Cs, Co, use_pixel_shuffle=False)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    down, skip, r = torch.randn(B, Cin, h, h), torch.randn(B, Cs, 2 * h, 2 * h), torch.randn(B, Co, 2 * h, 2 * h)
    ref = {}
    for tag, dt in (("f64", torch.float64), ("o32", torch.float32)):
        P = {"m." + k: (v.detach().clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        for k in O.trainable_keys(P):
            P[k].requires_grad_(True)
        d_, s_ = down.detach().clone().to(dt).requires_grad_(True), skip.detach().clone().to(dt).requires_grad_(True)
        y = O.styled_res_up_block(P, "m", d_, s_, True)
        (y * r.to(dt)).sum().backward()
        ref[tag] = (y.detach(), d_.grad, s_.grad, {k[2:]: P[k].grad for k in O.trainable_keys(P)})
    mm = m.cuda().train()
    d_, s_ = down.clone().cuda().requires_grad_(True), skip.clone().cuda().requires_grad_(True)
    y = mm(d_, s_)
    (y * r.cuda()).sum().backward()
    torch.cuda.synchronize()
    print("== %d+%d->%d low %dx%d B%d: y hip %.1e o32 %.1e | g_down %.1e %.1e | g_skip %.1e %.1e" % (Cin, Cs, Co, h, h, B, rel(y.detach(), ref["f64"][0]), rel(ref["o32"][0], ref["f64"][0]),
          rel(d_.grad, ref["f64"][1]), rel(ref["o32"][1], ref["f64"][1]), rel(s_.grad, ref["f64"][2]), rel(ref["o32"][2], ref["f64"][2])))
    for k, p in mm.named_parameters():
        e, eo = rel(p.grad, ref["f64"][3][k]), rel(ref["o32"][3][k], ref["f64"][3][k])
        if e > 5 * eo + 1e-6:
            print("     %-28s hip %.1e o32 %.1e" % (k, e, eo))
