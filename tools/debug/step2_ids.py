#!/usr/bin/env python3
"""Which pixels of step 2 of the lr = 1e-6 warm fixture get another code than the reference, and how close HIP's own top-2 is there."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-image-editing_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from helpers import build_models, step_cfg
from test_oracle_golden import apply_warm_state
from trainers import FirstStepTrainer, FlipViews


class G:
    def __init__(s, p): s.d = np.load(p); s.files = s.d.files
    def __getitem__(s, k): return s.d[k]
    def t(s, k, dev=None): return torch.from_numpy(s.d[k]).to(dev or "cpu")
    def group(s, p): return {k[len(p) + 1:]: s.d[k] for k in s.files if k.startswith(p + "/")}


g = G(os.path.join(ROOT, "tests/golden/step_rcfg64_warm_lr1e-6.npz"))
enc, dec = build_models(g.group("cfg"))
sd = enc.state_dict(); apply_warm_state(g, sd)
cfg = step_cfg(g)
tr = FirstStepTrainer(dict_size=cfg["dict_size"], momentum=cfg["momentum"], margin=cfg["margin"], lr=cfg["optim"]["lr"],
                      betas=cfg["optim"]["betas"], views=FlipViews(border=cfg["border"]), encoder=enc, decoder=dec, device="cuda")
for s in range(3):
    img, noi = g.t("step%d/image" % s, "cuda"), g.t("step%d/noise" % s, "cuda")
    with torch.no_grad():      # HIP's own gaps before the step moves anything
        x2 = torch.flip(img, dims=[3]) + noi
        f = tr.encoder.feature_extraction(x2)
        B, D, H, W = f.shape
        flat = f.permute(0, 2, 3, 1).reshape(-1, D)
        e = tr.encoder.vq.embed
        sc = 2 * flat @ e.t() - (e * e).sum(1)[None] - (flat * flat).sum(1)[:, None]
        top = sc.topk(2, dim=1)
        gap_hip = (top.values[:, 0] - top.values[:, 1]).reshape(B, H, W).cpu().numpy()
        arg_hip = top.indices[:, 0].reshape(B, H, W).cpu().numpy()
    f_tr = tr.encoder.feature_extraction(x2).detach()          # grad mode: the training forward's kernels
    from hipops import ops
    q, _, ids_k = ops.vq_quantize(f_tr, e, tr.encoder.vq.cluster_size, tr.encoder.vq.embed_avg, False, 0.999, 1e-5, id_base=1)
    ids_k = ids_k.cpu().numpy()
    flat_tr = f_tr.permute(0, 2, 3, 1).reshape(-1, D)
    sc_tr = 2 * flat_tr @ e.t() - (e * e).sum(1)[None] - (flat_tr * flat_tr).sum(1)[:, None]
    arg_tr = sc_tr.argmax(1).reshape(B, H, W).cpu().numpy() + 1
    print("step", s, "max |f_nograd - f_train| %.3e; VQ kernel vs torch argmax on train features: %d differ; torch(train) vs torch(nograd): %d differ"
          % (float((f - f_tr).abs().max()), int((ids_k != arg_tr).sum()), int((arg_tr != arg_hip + 1).sum())))
    for b in np.argwhere(ids_k != arg_tr)[:5]:
        b = tuple(b); print("   kernel/torch mismatch at", b, ids_k[b], arg_tr[b], "scores", sc_tr.reshape(B, H, W, -1)[b].cpu().numpy().round(4))
    out = tr.training_step({"image": img}, noise=noi)
    torch.cuda.synchronize()
    for v in ("1", "2"):
        ids = out["ids_" + v].cpu().numpy(); ref = g["step%d/ids_%s" % (s, v)]; gap = g["step%d/gap_%s" % (s, v)]
        bad = np.argwhere(ids != ref)
        print("step", s, "view", v, "differ:", len(bad))
        for b in bad[:5]:
            b = tuple(b)
            print("   at", b, "hip id", ids[b], "ref id", ref[b], "ref gap %.4e" % gap[b], ("hip torch-side gap %.4e argmax+1 %d" % (gap_hip[b], arg_hip[b] + 1)) if v == "2" else "")
    print("   embed checksum", float(tr.encoder.vq.embed.double().abs().sum()), "ref", float(g["step%d/after_sum.enc.vq.embed" % s][1]))
