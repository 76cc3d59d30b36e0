#!/usr/bin/env python3
"""Generate golden vectors from the upstream reference's own modules.

Runs ONLY in the build container (needs /root/reference; see _refshim.py).  Output:
tests/golden/*.npz — tensors only (inputs, weights or seeds+checksums, expected
outputs / gradients / buffers).  No reference source text or pickled modules.

    python tests/golden/make_golden.py

The training-step vectors drive the reference modules with the first-step
sequence of src/trainers/single_window_trainer.py:68-147 restated here around
them (Lightning / kornia are absent offline): views = identity and horizontal
flip (+ noise on the noised copy), r_ids by index flip with an optional zero
border, MSE reconstruction, torch.optim.Adam x2 as base.py:165-175 builds them.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _refshim  # noqa: E402

R = _refshim.load_reference()
torch.set_num_threads(8)


def npy(t):
    return t.detach().cpu().numpy()


def checksum(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), float(t.numel())])


def save(name, d):
    path = os.path.join(os.environ.get("GOLDEN_OUT", HERE), name)
    np.savez_compressed(path, **d)
    print("wrote %-28s %8.1f KB  (%d arrays)" % (name, os.path.getsize(path) / 1024, len(d)))


def module_case(tag, mod, inputs, out_dict, train=True, call=None):
    """Run `mod(*inputs)`, backprop sum(out*Rnd), record everything under tag/."""
    mod.train(train)
    ins = [x.clone().requires_grad_(x.is_floating_point()) for x in inputs]
    for k, v in mod.state_dict().items():
        out_dict["%s/P.%s" % (tag, k)] = npy(v).copy()
    out = call(mod, *ins) if call else mod(*ins)
    outs = out if isinstance(out, (tuple, list)) else (out,)
    rnds = [torch.randn_like(o) for o in outs]
    loss = sum((o * r).sum() for o, r in zip(outs, rnds))
    loss.backward()
    for i, x in enumerate(inputs):
        out_dict["%s/in.%d" % (tag, i)] = npy(x)
        if ins[i].grad is not None:
            out_dict["%s/gin.%d" % (tag, i)] = npy(ins[i].grad)
    for i, (o, r) in enumerate(zip(outs, rnds)):
        out_dict["%s/out.%d" % (tag, i)] = npy(o)
        out_dict["%s/R.%d" % (tag, i)] = npy(r)
    for k, p in mod.named_parameters():
        out_dict["%s/gP.%s" % (tag, k)] = npy(p.grad)
    for k, v in mod.state_dict().items():
        if "running_" in k or "num_batches" in k:
            out_dict["%s/after.%s" % (tag, k)] = npy(v).copy()


def gen_blocks():
    d = {}
    B = R.blocks
    torch.manual_seed(11)
    module_case("double_conv", B.DoubleConv(16, 32), [torch.randn(2, 16, 16, 16)], d)
    module_case("res_block", B.ResBlock(16, 32), [torch.randn(2, 16, 16, 16)], d)
    module_case("res_block_c1", B.ResBlock(1, 16), [torch.randn(2, 1, 16, 16)], d)
    module_case("up_block", B.UpBlock(32 + 16, 16),
                [torch.randn(2, 32, 8, 8), torch.randn(2, 16, 16, 16)], d)
    module_case("styled_denorm", B.StyledDenorm(16, 32),
                [torch.randn(3, 16, 8, 8) * 2 + 0.5, torch.randn(3, 32, 8, 8)], d)
    sd = B.StyledDenorm(16, 16)
    sd.param_free_norm.running_mean.normal_()
    sd.param_free_norm.running_var.uniform_(0.5, 2.0)
    module_case("styled_denorm_eval", sd,
                [torch.randn(2, 16, 8, 8), torch.randn(2, 16, 8, 8)], d, train=False)
    module_case("styled_res_up", B.StyledResUpBlock(32, 16, 16, use_pixel_shuffle=False),
                [torch.randn(2, 32, 8, 8), torch.randn(2, 16, 16, 16)], d)
    module_case("aspp", R.aspp.ASPP(16, 16, [2, 6, 12, 18]), [torch.randn(2, 16, 24, 24)], d)
    # odd channel counts -> exercises the generic (non-MFMA) kernels
    module_case("double_conv_odd", B.DoubleConv(3, 5), [torch.randn(2, 3, 10, 10)], d)
    save("blocks.npz", d)


def gen_vq():
    d = {}
    for tag, (Bn, D, K, HW, mom) in {"k10": (2, 16, 10, 16, 0.999), "k64": (2, 32, 64, 8, 0.9),
                                     "k1024": (1, 64, 1024, 16, 0.99)}.items():
        torch.manual_seed(5)
        vq = R.vq_module.VQModule(emb_dim=D, dict_size=K, momentum=mom, eps=1e-5, knn_backend="torch")
        d[tag + "/embed0"] = npy(vq.embed).copy()
        d[tag + "/momentum"] = np.array(mom)
        vq.train()
        for call in (1, 2):
            x = (torch.randn(Bn, D, HW, HW) * 1.3).requires_grad_(True)
            embed_before = vq.embed.clone()
            q, commit, ids = vq(x)
            rnd = torch.randn_like(q)
            ((q * rnd).sum() + 3.0 * commit).backward()
            flat = x.detach().permute(0, 2, 3, 1).reshape(-1, D)
            sc, _ = R.vq_module._torch_knn(embed_before, flat, 2, "l2")   # (N,2) best, second
            d["%s/gap%d" % (tag, call)] = npy((sc[:, 0] - sc[:, 1]).reshape(Bn, HW, HW))
            d["%s/x%d" % (tag, call)] = npy(x)
            d["%s/R%d" % (tag, call)] = npy(rnd)
            d["%s/q%d" % (tag, call)] = npy(q)
            d["%s/commit%d" % (tag, call)] = npy(commit)
            # reference returns ids in its transposed (B,W,H) order; store per-pixel (B,H,W)
            d["%s/ids%d" % (tag, call)] = npy(ids.transpose(1, 2))
            d["%s/gx%d" % (tag, call)] = npy(x.grad)
            for b in ("embed", "cluster_size", "embed_avg"):
                d["%s/%s_after%d" % (tag, b, call)] = npy(getattr(vq, b)).copy()
        vq.eval()
        x = torch.randn(Bn, D, HW, HW)
        q, commit, ids = vq(x)
        d[tag + "/x_eval"] = npy(x)
        d[tag + "/q_eval"] = npy(q)
        d[tag + "/commit_eval"] = npy(commit)
        d[tag + "/ids_eval"] = npy(ids.transpose(1, 2))
        look = vq.lookup(ids)
        d[tag + "/lookup_eval"] = npy(look)
    save("vq.npz", d)


def gen_losses():
    d = {}
    torch.manual_seed(21)
    Bn, D, K, HW = 3, 16, 10, 12
    for tag, use_d, use_r in (("full", True, True), ("cross_only", None, None)):
        e1 = torch.randn(Bn, D, HW, HW, requires_grad=True)
        e2 = torch.randn(Bn, D, HW, HW, requires_grad=True)
        cb = torch.randn(D, K)
        ids1 = torch.randint(0, K + 1, (Bn, HW, HW))
        ids2 = torch.randint(0, K + 1, (Bn, HW, HW))
        ids2[0][ids2[0] == 3] = 0       # class 3 absent in sample 0
        ids1[1] = 0                      # a fully out-of-frame sample
        oh = R.OneHotEncoder(K + 1)
        r1 = oh(ids1.int())[:, 1:]
        r2 = oh(ids2.int())[:, 1:]
        L = R.EmbeddingLoss(K, 0.5, use_d, use_r)
        lc, ld, lr = L(e1, r1, e2, r2, cb)
        lc.backward()
        d[tag + "/e1"], d[tag + "/e2"], d[tag + "/cb"] = npy(e1), npy(e2), npy(cb)
        d[tag + "/ids1"], d[tag + "/ids2"] = npy(ids1), npy(ids2)
        d[tag + "/onehot1"] = npy(oh(ids1.int()))
        d[tag + "/l_cross"] = npy(lc)
        d[tag + "/l_dist"] = np.array(float(ld))
        d[tag + "/l_reg"] = np.array(float(lr))
        d[tag + "/ge1"], d[tag + "/ge2"] = npy(e1.grad), npy(e2.grad)
    # segmentation losses
    logits = torch.randn(2, 5, 9, 9, requires_grad=True)
    tgt = F.one_hot(torch.randint(0, 5, (2, 9, 9)), 5).permute(0, 3, 1, 2).float()
    for name, mod in (("dice", R.SoftDiceLoss()), ("dice_ign", R.SoftDiceLoss(ignore_index=0)),
                      ("focal", R.FocalLoss())):
        logits.grad = None
        l = mod(logits, tgt)
        l.backward()
        d["seg/" + name] = npy(l)
        d["seg/g_" + name] = npy(logits.grad)
    d["seg/logits"], d["seg/target"] = npy(logits), npy(tgt)
    # dropblock deterministic part
    db = R.dropblock.DropBlock2D(drop_prob=0.3, block_size=4)
    m = (torch.rand(2, 12, 12) < 0.05).float()
    d["dropblock/seed4"], d["dropblock/keep4"] = npy(m), npy(db._compute_block_mask(m))
    db = R.dropblock.DropBlock2D(drop_prob=0.3, block_size=5)
    d["dropblock/keep5"] = npy(db._compute_block_mask(m))
    save("losses.npz", d)


def ref_first_step(enc, dec, eopt, dopt, image, noise, cfg):
    """single_window_trainer.py:68-147 around the reference modules."""
    K, w = cfg["dict_size"], cfg["weights"]
    oh = R.OneHotEncoder(K + 1)
    L = R.EmbeddingLoss(K, cfg["margin"], True, True)
    n1, c1 = image, image
    c2 = torch.flip(image, dims=[3])
    n2 = c2 + noise
    e1, lc1, ids1 = enc(n1, rank=0)
    e2, lc2, ids2 = enc(n2, rank=0)

    def rid(ids):
        r = torch.flip(ids, dims=[2]).clone()
        b = cfg["border"]
        if b > 0:
            r[:, :b, :] = 0; r[:, -b:, :] = 0; r[:, :, :b] = 0; r[:, :, -b:] = 0
        return r.int()
    r1 = oh(rid(ids1))[:, 1:]
    r2 = oh(rid(ids2))[:, 1:]
    codebook = enc.vq.get_codebook()
    l_cross, l_dist, l_reg = L(e1, r1, e2, r2, codebook)
    rec1, rec2 = dec(e1), dec(e2)
    l_recon = F.mse_loss(rec1, c1) + F.mse_loss(rec2, c2)
    l_commit = lc1 + lc2
    total = (w["commit"] * l_commit + w["cross"] * l_cross + w["dist"] * l_dist
             + w["reg"] * l_reg + w["recon"] * l_recon)
    if eopt is not None:
        eopt.zero_grad(); dopt.zero_grad()
    total.backward()
    grads = {"enc." + k: p.grad.clone() for k, p in enc.named_parameters()}
    grads.update({"dec." + k: p.grad.clone() for k, p in dec.named_parameters()})
    if eopt is not None:
        eopt.step(); dopt.step()
    return dict(total=total, commit=l_commit, cross=l_cross, dist=l_dist, reg=l_reg, recon=l_recon,
                ids_1=ids1, ids_2=ids2, recon_1=rec1, recon_2=rec2, embed_1=e1, embed_2=e2), grads


def ref_vq_gaps(enc, x):
    """Top-1 / top-2 score gap of every pixel of `x` under the encoder's CURRENT codebook (vq_module.py:45-62), per
    pixel in (B, H, W) order: an implementation's ids must be bit-equal wherever this gap is clear of fp32 rounding.
    The extra feature pass changes nothing (no random numbers, no running statistics in the encoder)."""
    with torch.no_grad():
        f = enc(x, skip_vq=True)
        f = f[0] if isinstance(f, (tuple, list)) else f
        Bn, D, Hh, Ww = f.shape
        flat = f.permute(0, 2, 3, 1).reshape(-1, D)
        sc, _ = R.vq_module._torch_knn(enc.vq.embed, flat, 2, "l2")
        return (sc[:, 0] - sc[:, 1]).reshape(Bn, Hh, Ww)


def sample_idx(numel, n=64, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, numel, (min(n, numel),), generator=g)


def ref_grads_fp64(enc, dec, image, noise, cfg):
    """The same first step on `.double()` copies of the reference modules (same weights, buffers and inputs): the
    gradients every fp32 implementation - the reference's own included - is an approximation of."""
    import copy
    enc64, dec64 = copy.deepcopy(enc).double(), copy.deepcopy(dec).double()
    out, grads = ref_first_step(enc64, dec64, None, None, image.double(), noise.double(), cfg)
    return out, grads


def traj_record(enc, dec, out, flipped):
    """What a later step is compared on: losses, ids and reconstructions (batch order undone), VQ buffers, BatchNorm statistics."""
    un = (lambda t: t.flip(0)) if flipped else (lambda t: t)
    r = {k: float(out[k]) for k in ("total", "commit", "cross", "dist", "reg", "recon")}
    for k in ("ids_1", "ids_2", "recon_1", "recon_2"):
        r[k] = un(out[k]).detach().clone()
    for b in ("embed", "cluster_size", "embed_avg"):
        r["vq." + b] = getattr(enc.vq, b).detach().clone()
    r["bn"] = torch.cat([v.detach().reshape(-1).float() for k, v in dec.state_dict().items() if "running_" in k])
    return r


def ref_trajectory(enc, dec, cfg, lr, n_steps, batch, size, flip, threads):
    """The same n_steps training steps on copies of the modules, evaluated in a mathematically equivalent way that changes
    only the association order of fp32 reductions (batch order reversed and / or one thread): the reference's own spread."""
    import copy
    from oracle import vqwnet_ref as O
    enc, dec = copy.deepcopy(enc), copy.deepcopy(dec)
    eopt = torch.optim.Adam(filter(lambda p: p.requires_grad, enc.parameters()), lr=lr, betas=(0.5, 0.999), weight_decay=0)
    dopt = torch.optim.Adam(filter(lambda p: p.requires_grad, dec.parameters()), lr=lr, betas=(0.5, 0.999), weight_decay=0)
    torch.set_num_threads(threads)
    recs = []
    for s in range(n_steps):
        image, noise = O.synthetic_slices(batch, size, 1234 + s)
        if flip:
            image, noise = image.flip(0), noise.flip(0)
        out, _ = ref_first_step(enc, dec, eopt, dopt, image, noise, cfg)
        recs.append(traj_record(enc, dec, out, flip))
    torch.set_num_threads(8)
    return recs


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def gen_step(name, enc_filters, dec_filters, K, size, batch, n_steps, seed, momentum=0.999, warm=False, lr=1e-4, spread=False):
    from oracle import vqwnet_ref as O
    d = {}
    torch.manual_seed(seed)
    enc = R.UNetEncoder(1, enc_filters, K, momentum, "torch", False, 4, True)
    dec = R.UNetDecoder(enc_filters[0], 1, dec_filters, use_dropblock=False,
                        dropped_skip_layers=[], use_styled_up_block=True, use_pixel_shuffle=False)
    enc.train(); dec.train()
    cfg = dict(dict_size=K, margin=0.5, border=2, momentum=momentum,
               weights=dict(commit=1.0, cross=1.0, dist=1.0, reg=1.0, recon=1.0))
    d["cfg/enc_filters"], d["cfg/dec_filters"] = np.array(enc_filters), np.array(dec_filters)
    d["cfg/K"], d["cfg/size"], d["cfg/batch"] = np.array(K), np.array(size), np.array(batch)
    d["cfg/seed"], d["cfg/n_steps"], d["cfg/border"] = np.array(seed), np.array(n_steps), np.array(2)
    d["cfg/momentum"], d["cfg/margin"] = np.array(momentum), np.array(0.5)
    d["cfg/lr"], d["cfg/betas"] = np.array(lr), np.array([0.5, 0.999])
    for pre, m in (("enc", enc), ("dec", dec)):
        for k, v in m.state_dict().items():
            d["init_sum/%s.%s" % (pre, k)] = checksum(v.float())
    d["cfg/warm"] = np.array(int(warm))
    if warm:
        # checkpoint-like VQ state (as after many updates): no cluster_size==0 blow-up, so gradients are
        # well conditioned.  Same reference code, different loaded state.
        with torch.no_grad():
            enc.vq.embed.mul_(0.7)
            enc.vq.cluster_size.fill_(batch * size * size / K)
            enc.vq.embed_avg.copy_(enc.vq.embed.t() * enc.vq.cluster_size[None, :])
        for b in ("embed", "cluster_size", "embed_avg"):
            d["warm/vq." + b] = npy(getattr(enc.vq, b)).copy()
    eopt = torch.optim.Adam(filter(lambda p: p.requires_grad, enc.parameters()), lr=lr, betas=(0.5, 0.999), weight_decay=0)
    dopt = torch.optim.Adam(filter(lambda p: p.requires_grad, dec.parameters()), lr=lr, betas=(0.5, 0.999), weight_decay=0)
    # the reference's own multi-step spread (steps > 0 are compared against it, tests/test_oracle_golden.py::check_later_step)
    trajs = [ref_trajectory(enc, dec, cfg, lr, n_steps, batch, size, f, t) for f, t in ((True, 8), (False, 1))] if spread else []
    # eval-mode forward + mask-guided reconstruction (run_recon.py:179-194) on the INITIAL state (so both sides hold bit-identical weights;
    # after an optimiser step two fp32 implementations drift, see tests/test_oracle_golden.py::check_step)
    enc.eval(); dec.eval()
    with torch.no_grad():
        image, _ = O.synthetic_slices(batch, size, 999)
        e, _, ids = enc(image)
        d["eval/image"], d["eval/ids"], d["eval/recon"] = npy(image), npy(ids), npy(dec(e))
        d["eval/gap"] = npy(ref_vq_gaps(enc, image))
        g = torch.Generator().manual_seed(3)
        lab = torch.randint(0, K + 1, (batch, size, size), generator=g)
        lab[:, : size // 4, :] = lab[:, :1, :1]   # a constant region, like an edited label map
        mask = (lab != 0)
        m = torch.clamp(lab, min=1) - 1
        emb = enc.get_embed_from_ids(m)
        emb = emb * mask[:, None]
        emb = emb * (mask.numel() / mask.sum())
        d["recon/label_map"], d["recon/embed"], d["recon/recon"] = npy(lab), npy(emb), npy(dec(emb))
    enc.train(); dec.train()
    for s in range(n_steps):
        image, noise = O.synthetic_slices(batch, size, 1234 + s)
        d["step%d/image" % s], d["step%d/noise" % s] = npy(image), npy(noise)
        if s == 0:
            import copy
            out64, grads64 = ref_grads_fp64(enc, dec, image, noise, cfg)
            # the reference's own fp32 spread: the same step evaluated in mathematically equivalent ways that change
            # only the association order of its fp32 reductions - batch order reversed, one thread instead of eight
            variants = []
            for flip, threads in ((True, 8), (False, 1), (True, 1)):
                torch.set_num_threads(threads)
                img_v, noi_v = (image.flip(0), noise.flip(0)) if flip else (image, noise)
                variants.append(ref_first_step(copy.deepcopy(enc), copy.deepcopy(dec), None, None, img_v, noi_v, cfg)[1])
            torch.set_num_threads(8)
        # the views of ref_first_step, before the step moves the codebook (every step: later steps of a warm fixture are
        # held to bit-equal ids where the reference's own gap is clear, like step 0)
        d["step%d/gap_1" % s] = npy(ref_vq_gaps(enc, image))
        # view 2 is quantised AFTER view 1's call has moved the codebook (enc(n1) then enc(n2) in ref_first_step, as in
        # single_window_trainer.py:84-85): its decision gaps are those under that intermediate codebook - a copy of the encoder
        # takes view 1's call.  (Round 3 took them under the codebook before the step: off by the EMA update, enough to call a
        # near-tie of the real decision "clear" - found in round 4 when one such pixel of step 2 flipped.)
        import copy
        enc_mid = copy.deepcopy(enc)
        with torch.no_grad():
            enc_mid(image, rank=0)
        d["step%d/gap_2" % s] = npy(ref_vq_gaps(enc_mid, torch.flip(image, dims=[3]) + noise))
        d["step%d/gap_2_before_view1" % s] = npy(ref_vq_gaps(enc, torch.flip(image, dims=[3]) + noise))
        out, grads = ref_first_step(enc, dec, eopt, dopt, image, noise, cfg)
        if s == 0:
            # principled gradient gate (tests/helpers.py::check_grads_vs_fp64): 256 sampled entries of the fp64 gradient and
            # of each fp32 evaluation of the reference at the same positions, the fp64 norm, whole-tensor fp32 errors
            for v in ("1", "2"):
                d["step0/ids64_agree_" + v] = np.array(float((out64["ids_" + v] == out["ids_" + v]).double().mean()))
            d["step0/total64"] = np.array(float(out64["total"]))
            for k, g64 in grads64.items():
                idx = sample_idx(g64.numel(), 256, seed=1)
                d["step0/g64." + k] = npy(g64.reshape(-1)[idx])
                d["step0/gnorm64." + k] = np.array(float(g64.norm()))
                for i, gv in enumerate([grads] + variants):
                    d["step0/g32v%d.%s" % (i, k)] = npy(gv[k].reshape(-1)[idx])
                d["step0/gerr32." + k] = np.array([float((gv[k].double() - g64).norm() / (g64.norm() + 1e-300)) for gv in [grads] + variants])
        for k in ("total", "commit", "cross", "dist", "reg", "recon"):
            d["step%d/%s" % (s, k)] = np.array(float(out[k]))
        for k in ("ids_1", "ids_2", "recon_1", "recon_2"):
            d["step%d/%s" % (s, k)] = npy(out[k])
        d["step%d/embed_1_sum" % s] = checksum(out["embed_1"])
        for k, g in grads.items():
            idx = sample_idx(g.numel())
            d["step%d/g.%s" % (s, k)] = npy(g.reshape(-1)[idx])
            d["step%d/gnorm.%s" % (s, k)] = np.array(float(g.norm()))
        for pre, m in (("enc", enc), ("dec", dec)):
            for k, v in m.state_dict().items():
                v = v.float()
                idx = sample_idx(v.numel())
                d["step%d/after.%s.%s" % (s, pre, k)] = npy(v.reshape(-1)[idx])
                d["step%d/after_sum.%s.%s" % (s, pre, k)] = checksum(v)
        if trajs:
            # distance of each equivalent evaluation of the reference from the run as launched, per compared quantity: the
            # reference's own reproducibility after s optimiser steps (row i = variant i: batch reversed; one thread)
            main = traj_record(enc, dec, out, False)
            d["step%d/bn_running" % s] = npy(main["bn"])
            for k in ("total", "commit", "cross", "dist", "reg", "recon"):
                d["step%d/spread.%s" % (s, k)] = np.array([abs(t[s][k] - main[k]) / (abs(main[k]) + 1e-30) for t in trajs])
            for k in ("recon_1", "recon_2", "vq.embed", "vq.cluster_size", "vq.embed_avg", "bn"):
                d["step%d/spread.%s" % (s, k)] = np.array([rel(t[s][k], main[k]) for t in trajs])
            for k in ("ids_1", "ids_2"):
                d["step%d/spread.%s" % (s, k)] = np.array([float((t[s][k] != main[k]).double().mean()) for t in trajs])
    save(name, d)


def gen_extras():
    """Optional paths: pixel-shuffle up-sampling (the decoder's default), VQWNet monolith, DropBlock apply."""
    d = {}
    B = R.blocks
    torch.manual_seed(31)
    module_case("styled_res_up_ps", B.StyledResUpBlock(32, 16, 16, use_pixel_shuffle=True),
                [torch.randn(2, 32, 8, 8), torch.randn(2, 16, 16, 16)], d)
    # decoder with the constructor defaults that matter (use_pixel_shuffle=True); weights from the seed
    torch.manual_seed(32)
    dec = R.UNetDecoder(16, 1, [16, 32, 32, 32, 64], use_dropblock=False, dropped_skip_layers=[])
    dec.train()
    for k, v in dec.state_dict().items():
        d["dec_ps/init_sum." + k] = checksum(v.float())
    x = torch.randn(2, 16, 32, 32, requires_grad=True)
    y = dec(x)
    r = torch.randn_like(y)
    (y * r).sum().backward()
    d["dec_ps/x"], d["dec_ps/R"], d["dec_ps/y"], d["dec_ps/gx"] = npy(x), npy(r), npy(y), npy(x.grad)
    for k, p in dec.named_parameters():
        d["dec_ps/gnorm." + k] = np.array(float(p.grad.norm()))
    # VQWNet monolith (vqwnet.py), warm VQ state for conditioning
    torch.manual_seed(33)
    net = R.VQWNet(1, 1, [16, 16, 32, 32, 32], dict_size=6)
    for k, v in net.state_dict().items():
        d["vqwnet/init_sum." + k] = checksum(v.float())
    with torch.no_grad():
        net.vq.embed.mul_(0.7)
        net.vq.cluster_size.fill_(2 * 32 * 32 / 6)
        net.vq.embed_avg.copy_(net.vq.embed.t() * net.vq.cluster_size[None, :])
    net.train()
    img = torch.randn(2, 1, 32, 32).clamp_(-1, 1)
    out = net(img)
    r = torch.randn_like(out["recon"])
    ((out["recon"] * r).sum() + out["commit_loss"]).backward()
    d["vqwnet/image"], d["vqwnet/R"] = npy(img), npy(r)
    d["vqwnet/recon"], d["vqwnet/embed"], d["vqwnet/ids"] = npy(out["recon"]), npy(out["embed"]), npy(out["ids"])
    d["vqwnet/commit"] = npy(out["commit_loss"])
    for k, p in net.named_parameters():
        d["vqwnet/gnorm." + k] = np.array(float(p.grad.norm()))
    for b in ("embed", "cluster_size", "embed_avg"):
        d["vqwnet/after.vq." + b] = npy(getattr(net.vq, b)).copy()
    gen = net.generate_images_from_ids((out["ids"] - 1))
    d["vqwnet/gen_recon"] = npy(gen["recon"])
    # DropBlock deterministic apply
    db = R.dropblock.DropBlock2D(drop_prob=0.3, block_size=4)
    m = (torch.rand(2, 12, 12) < 0.05).float()
    keep = db._compute_block_mask(m)
    xx = torch.randn(2, 8, 12, 12)
    d["dropblock/seed"], d["dropblock/x"] = npy(m), npy(xx)
    d["dropblock/y"] = npy(xx * keep[:, None] * keep.numel() / keep.sum())
    save("extras.npz", d)


def gen_gan():
    """Second training step pieces: the reference's NLayerDiscriminator (discriminator.py:18-87)
    forward / backward in train and eval mode, hinge_d_loss (gan_loss.py:6-10), and the discriminator half of
    _train_second_step_nl_dis (single_window_trainer.py:474-487) for two Adam steps."""
    d = {}
    torch.manual_seed(41)
    dis = R.NLayerDiscriminator(in_channels=1, out_channels=1, n_filters=16, n_layers=3)
    module_case("dis_f16", dis, [torch.randn(3, 1, 64, 64)], d)
    torch.manual_seed(42)
    dis2 = R.NLayerDiscriminator(in_channels=1, out_channels=1, n_filters=8, n_layers=2)
    with torch.no_grad():                      # non-trivial running statistics for the eval case
        for m in dis2.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
    module_case("dis_f8_eval", dis2, [torch.randn(2, 1, 40, 48)], d, train=False)
    # hinge loss values and gradients
    torch.manual_seed(43)
    lr_, lf_ = torch.randn(4, 1, 6, 6, requires_grad=True), torch.randn(4, 1, 6, 6, requires_grad=True)
    l = R.hinge_d_loss(lr_, lf_)
    (3.0 * l).backward()
    d["hinge/real"], d["hinge/fake"], d["hinge/loss"] = npy(lr_), npy(lf_), npy(l)
    d["hinge/g_real"], d["hinge/g_fake"] = npy(lr_.grad), npy(lf_.grad)
    g = torch.randn(4, 1, 6, 6, requires_grad=True)
    lg = -torch.mean(g)
    (2.0 * lg).backward()
    d["gen/x"], d["gen/loss"], d["gen/gx"] = npy(g), npy(lg), npy(g.grad)
    # discriminator update loop: two steps of (real, fake) -> hinge -> Adam(lr 1e-3, betas (0.5, 0.999))
    torch.manual_seed(44)
    dis3 = R.NLayerDiscriminator(in_channels=1, out_channels=1, n_filters=8, n_layers=3)
    dis3.train()
    for k, v in dis3.state_dict().items():
        d["dstep/P." + k] = npy(v).copy()
    opt = torch.optim.Adam(dis3.parameters(), lr=1e-3, betas=(0.5, 0.999))
    for s in range(2):
        real, fake = torch.randn(2, 1, 32, 32).clamp_(-1, 1), torch.randn(2, 1, 32, 32).clamp_(-1, 1)
        l_dis = R.hinge_d_loss(dis3(real), dis3(fake))
        opt.zero_grad()
        (0.8 * l_dis).backward()
        opt.step()
        d["dstep/real%d" % s], d["dstep/fake%d" % s], d["dstep/loss%d" % s] = npy(real), npy(fake), npy(l_dis)
    for k, v in dis3.state_dict().items():
        d["dstep/after." + k] = npy(v).copy()
    save("gan.npz", d)


def _vq_dist_worker(rank, world, port, tmp):
    """One rank of the reference's VQModule under torch.distributed (gloo): WORLD_SIZE drives utils.is_distributed()."""
    import torch.distributed as dist
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    Rw = _refshim.load_reference()
    out = {}
    for tag, (Bn, D, K, HW, mom) in {"k10": (2, 16, 10, 16, 0.9), "k64": (1, 32, 64, 12, 0.99)}.items():
        torch.manual_seed(5)                                   # identical replicas, as DDP's broadcast leaves them
        vq = Rw.vq_module.VQModule(emb_dim=D, dict_size=K, momentum=mom, eps=1e-5, knn_backend="torch")
        if rank == 0:
            out[tag + "/embed0"], out[tag + "/momentum"] = npy(vq.embed).copy(), np.array(mom)
        vq.train()
        for call in (1, 2):
            g = torch.Generator().manual_seed(100 * call + rank)
            x = torch.randn(Bn, D, HW, HW, generator=g) * 1.3
            q, commit, ids = vq(x)
            out["%s/r%d/x%d" % (tag, rank, call)] = npy(x)
            out["%s/r%d/ids%d" % (tag, rank, call)] = npy(ids.transpose(1, 2))
            out["%s/r%d/commit%d" % (tag, rank, call)] = npy(commit)
            for b in ("embed", "cluster_size", "embed_avg"):
                out["%s/r%d/%s_after%d" % (tag, rank, b, call)] = npy(getattr(vq, b)).copy()
    np.savez(os.path.join(tmp, "r%d.npz" % rank), **out)
    dist.destroy_process_group()


def gen_vq_dist():
    """The reference-held data-parallel fixture: VQModule under WORLD_SIZE=2, vq_module.py:187-193 -
    embed_sum is all-reduced and divided by the world size, the counts stay local, so the replicas' cluster_size and
    codebooks DIVERGE (the upstream quirk `dist_mode="reference"` reproduces)."""
    import tempfile
    import torch.multiprocessing as mp
    tmp = tempfile.mkdtemp()
    mp.spawn(_vq_dist_worker, args=(2, 29531, tmp), nprocs=2, join=True)
    d = {}
    for r in range(2):
        z = np.load(os.path.join(tmp, "r%d.npz" % r))
        d.update({k: z[k] for k in z.files})
    save("vq_dist.npz", d)


def gen_utils():
    """utils/__init__.py: CT window arithmetic (normalize / denormalize, :17-51), norm / denorm (:81-92) and load_json's
    false -> None convention (:99-106), through the reference's own functions."""
    import json
    import tempfile
    U = _refshim.load_reference_utils()
    d = {}
    g = np.random.default_rng(0)
    hu = np.concatenate([g.uniform(-2200, 2200, 500), np.array([-1300.0, -1299.5, -550.0, 199.5, 200.0, 40.0, -160.0, 240.0])]).astype(np.float32)
    d["hu"] = hu.copy()
    windows = {"default": dict(width=1500, center=-550, scale=2.0), "mediastinal": dict(width=400, center=40, scale=2.0),
               "wide": dict(width=4096, center=0, scale=2.0), "odd": dict(width=401, center=-7, scale=1.0)}
    for name, w in windows.items():
        d["window/%s" % name] = np.array([w["width"], w["center"], w["scale"]], dtype=np.float64)
        n = U.normalize(hu.copy(), **w)
        d["normalize/%s" % name] = n.copy()
        d["denormalize/%s" % name] = U.denormalize(n.copy(), **w)
        # multi_window_trainer.py:93-118: denormalize with the dataset window, normalize with the target window
        x_ds = U.normalize(hu.copy(), **windows["wide"])           # the dataset's units
        d["rewindow/%s" % name] = U.normalize(U.denormalize(torch.from_numpy(x_ds), **windows["wide"]).numpy(), **w)
    x = g.uniform(-1, 1, (2, 1, 8, 8)).astype(np.float32)
    d["norm/x"] = x.copy()
    d["norm/y"] = npy(U.norm(torch.from_numpy(x.copy())))
    d["denorm/y01"] = npy(U.denorm(torch.from_numpy(x.copy()), 0.0, 1.0))
    d["denorm/y_hu"] = npy(U.denorm(torch.from_numpy(x.copy()), -1300.0, 200.0))
    cfgj = {"run": {"seed": 3, "flag_false": False, "flag_true": True, "name": "a"}, "list": [1, 2, False]}
    path = os.path.join(tempfile.mkdtemp(), "c.json")
    json.dump(cfgj, open(path, "w"))
    c = U.load_json(path)
    d["load_json/repr"] = np.array(repr((c.run.seed, c.run.flag_false, c.run.flag_true, c.run.name, c.list)))
    d["load_json/source"] = np.array(json.dumps(cfgj))
    save("utils.npz", d)


if __name__ == "__main__":
    what = sys.argv[1:] or ["gan", "extras", "blocks", "vq", "losses", "steps", "cfg4", "vqdist", "utils"]
    if "gan" in what:
        gen_gan()
    if "extras" in what:
        gen_extras()
    if "blocks" in what:
        gen_blocks()
    if "vq" in what:
        gen_vq()
    if "losses" in what:
        gen_losses()
    if "steps" in what:
        gen_step("step_small.npz", [16, 16, 32, 32, 32], [16, 32, 32, 32, 64], 6, 32, 2, 3, seed=7,
                 momentum=0.9)
        gen_step("step_rcfg32.npz", [16, 32, 64, 128, 256], [32, 64, 128, 256, 512], 10, 32, 4, 1, seed=0)
        # three steps from a warm VQ state: the reference itself is stable there, so Adam's bias correction at t >= 2, the
        # BatchNorm running-statistics momentum and the second / third EMA update are pinned by its own multi-step output
        gen_step("step_rcfg64_warm.npz", [16, 32, 64, 128, 256], [32, 64, 128, 256, 512], 10, 64, 2, 3, seed=0, warm=True, spread=True)
        # the same three steps at lr = 1e-6.  Adam's first steps move every element by ~lr along sign(g), whatever |g|: elements
        # whose gradient is rounding noise move by +-lr at random, and at lr = 1e-4 that alone changes the next step's
        # reconstruction by 23 % between two equivalent evaluations of the reference (spread.* above).  At 1e-6 the reference
        # reproduces itself over three steps, so this fixture holds later steps to real tolerances.
        gen_step("step_rcfg64_warm_lr1e-6.npz", [16, 32, 64, 128, 256], [32, 64, 128, 256, 512], 10, 64, 2, 3, seed=0, warm=True,
                 lr=1e-6, spread=True)
    if "cfg4" in what:
        # BASELINE config 4 scaled down spatially (enc_filters[0] = emb_dim = 256, dict_size 1024)
        gen_step("step_cfg4_32.npz", [256, 64, 128, 256, 512], [32, 64, 128, 256, 512], 1024, 32, 2, 1, seed=0,
                 momentum=0.99, warm=True)
    if "vqdist" in what:
        gen_vq_dist()
    if "utils" in what:
        gen_utils()
