"""The CPU oracle (oracle/vqwnet_ref.py) against golden vectors produced by the reference's own modules.

These pin the oracle; the GPU tests then compare the HIP path with the oracle (and with the same vectors).
"""
import os

import numpy as np
import pytest
import torch

from helpers import rel_err, check_grads_vs_fp64, assert_close, build_models, check_init, step_cfg, sample_idx, checksum, assert_ids_equal_where_clear
from oracle import vqwnet_ref as O

TOL = 2e-5     # fp32, CPU vs CPU, different summation orders only


def _P(g, tag):
    return {k[2:]: v for k, v in g.group(tag).items() if k.startswith("P.")}


def _run_case(g, tag, fn, n_in):
    P = _P(g, tag)
    for k in P:
        if P[k].is_floating_point() and k.endswith((".weight", ".bias")):
            P[k].requires_grad_(True)
    ins = [g.t("%s/in.%d" % (tag, i)).requires_grad_(True) for i in range(n_in)]
    outs = fn(P, *ins)
    outs = outs if isinstance(outs, tuple) else (outs,)
    loss = sum((o * g.t("%s/R.%d" % (tag, i))).sum() for i, o in enumerate(outs))
    loss.backward()
    for i, o in enumerate(outs):
        assert_close(o, g["%s/out.%d" % (tag, i)], TOL, "%s out.%d" % (tag, i))
    for i, x in enumerate(ins):
        key = "%s/gin.%d" % (tag, i)
        if key in g.files:
            assert_close(x.grad, g[key], 20 * TOL, key, atol=1e-6)
    for k, p in P.items():
        key = "%s/gP.%s" % (tag, k)
        if key in g.files:
            assert_close(p.grad, g[key], 20 * TOL, key, atol=2e-5)
    return P


def test_blocks(golden):
    g = golden("blocks.npz")
    _run_case(g, "double_conv", lambda P, x: O.double_conv(_pref(P, "m."), "m", x), 1)
    _run_case(g, "double_conv_odd", lambda P, x: O.double_conv(_pref(P, "m."), "m", x), 1)
    _run_case(g, "res_block", lambda P, x: _res(P, x), 1)
    _run_case(g, "res_block_c1", lambda P, x: _res(P, x), 1)
    _run_case(g, "up_block", lambda P, d, s: O.up_block(_pref(P, "m."), "m", d, s), 2)
    P = _run_case(g, "styled_denorm", lambda P, x, s: O.styled_denorm(_pref(P, "m."), "m", x, s, True), 2)
    for k in ("running_mean", "running_var"):
        assert_close(P["param_free_norm." + k], g["styled_denorm/after.param_free_norm." + k], TOL, k)
    _run_case(g, "styled_denorm_eval", lambda P, x, s: O.styled_denorm(_pref(P, "m."), "m", x, s, False), 2)
    _run_case(g, "styled_res_up", lambda P, d, s: O.styled_res_up_block(_pref(P, "m."), "m", d, s, True), 2)
    _run_case(g, "aspp", lambda P, x: O.aspp(_pref(P, "m."), "m", x), 1)


def _pref(P, pre):
    """View of P under a key prefix that shares tensors (so running stats / grads land in P)."""
    class V(dict):
        def __getitem__(s, k):
            return P[k[len(pre):]]

        def get(s, k, d=None):
            return P.get(k[len(pre):], d)

        def __contains__(s, k):
            return k[len(pre):] in P
    return V()


def _res(P, x):
    return O.res_block(_pref(P, "m."), "m", x)


@pytest.mark.parametrize("tag", ["k10", "k64", "k1024"])
def test_vq(golden, tag):
    g = golden("vq.npz")
    V = dict(embed=g.t(tag + "/embed0").clone(), cluster_size=torch.zeros(g[tag + "/embed0"].shape[0]),
             embed_avg=g.t(tag + "/embed0").t().clone())
    mom = float(g[tag + "/momentum"])
    for call in (1, 2):
        x = g.t("%s/x%d" % (tag, call)).requires_grad_(True)
        q, commit, ids, gap = O.vq_forward(V, x, True, mom)
        ((q * g.t("%s/R%d" % (tag, call))).sum() + 3.0 * commit).backward()
        ref_ids = g["%s/ids%d" % (tag, call)]
        ref_gap = g["%s/gap%d" % (tag, call)]
        clear = ref_gap > 1e-4 * (1 + np.abs(ref_gap))
        assert clear.mean() > 0.99
        assert np.array_equal(ids.numpy()[clear], ref_ids[clear]), "ids differ on tie-free pixels"
        assert_close(q, g["%s/q%d" % (tag, call)], TOL, "q")
        assert_close(commit, g["%s/commit%d" % (tag, call)], TOL, "commit")
        assert_close(x.grad, g["%s/gx%d" % (tag, call)], TOL, "gx")
        for b in ("embed", "cluster_size", "embed_avg"):
            assert_close(V[b], g["%s/%s_after%d" % (tag, b, call)], 5e-5, "%s after call %d" % (b, call))
    x = g.t(tag + "/x_eval")
    q, commit, ids, _ = O.vq_forward(V, x, False, mom)
    assert np.mean(ids.numpy() == g[tag + "/ids_eval"]) > 0.999
    assert_close(q, g[tag + "/q_eval"], 1e-3, "q eval")
    assert_close(commit, g[tag + "/commit_eval"], 1e-4, "commit eval")


def test_losses(golden):
    g = golden("losses.npz")
    for tag, use_d in (("full", True), ("cross_only", False)):
        e1 = g.t(tag + "/e1").requires_grad_(True)
        e2 = g.t(tag + "/e2").requires_grad_(True)
        cb = g.t(tag + "/cb")
        K = cb.shape[1]
        r1 = O.one_hot(g.t(tag + "/ids1"), K + 1)[:, 1:]
        r2 = O.one_hot(g.t(tag + "/ids2"), K + 1)[:, 1:]
        assert np.array_equal(O.one_hot(g.t(tag + "/ids1"), K + 1).numpy(), g[tag + "/onehot1"])
        lc, ld, lr = O.embedding_loss(e1, r1, e2, r2, cb, 0.5, use_d, use_d)
        lc.backward()
        assert_close(lc, g[tag + "/l_cross"], TOL, "l_cross")
        assert_close(float(ld), g[tag + "/l_dist"], TOL, "l_dist")
        assert_close(float(lr), g[tag + "/l_reg"], TOL, "l_reg")
        assert_close(e1.grad, g[tag + "/ge1"], TOL, "ge1")
        assert_close(e2.grad, g[tag + "/ge2"], TOL, "ge2")
    # known-answer: K identical codes -> every pair (i==j included) contributes (2m)^2
    K, m = 7, 0.5
    cb = torch.ones(4, K)
    assert abs(float(O.distance_loss(cb, m)) - K * K * (2 * m) ** 2 / (2 * K * (K - 1))) < 1e-6
    # segmentation losses + dropblock
    logits = g.t("seg/logits").requires_grad_(True)
    tgt = g.t("seg/target")
    for name, fn in (("dice", lambda: O.soft_dice_loss(logits, tgt)),
                     ("dice_ign", lambda: O.soft_dice_loss(logits, tgt, ignore_index=0)),
                     ("focal", lambda: O.focal_loss(logits, tgt))):
        logits.grad = None
        l = fn()
        l.backward()
        assert_close(l, g["seg/" + name], TOL, name)
        assert_close(logits.grad, g["seg/g_" + name], 10 * TOL, "g_" + name)
    assert np.array_equal(O.dropblock_mask(g.t("dropblock/seed4"), 4).numpy(), g["dropblock/keep4"])
    assert np.array_equal(O.dropblock_mask(g.t("dropblock/seed4"), 5).numpy(), g["dropblock/keep5"])


# per-fixture gradient tolerances (relative error of sampled entries / norms vs the reference's gradients):
#   step_small       cold VQ start, small filters : well conditioned                          -> 2e-3, 2 outliers
#   step_rcfg64_warm R-cfg filters, checkpoint-like VQ state : well conditioned               -> 5e-2, 6 outliers
#   step_rcfg32      R-cfg, cold start (BASELINE config 1): cross-loss ~1e5 from the EMA blow-up and 2x2
#                    InstanceNorm planes make gradients chaotic even CPU-vs-CPU               -> norms only, 0.5
GRAD_TOL = {"step_small.npz": (2e-3, 2, 0.25), "step_rcfg64_warm.npz": (5e-2, 6, 0.25), "step_rcfg32.npz": (0.5, 60, 1.5)}


def apply_warm_state(g, sd_enc):
    """Load the checkpoint-like VQ buffers a warm fixture was generated with."""
    if "cfg/warm" in g.files and int(g["cfg/warm"]):
        for b in ("embed", "cluster_size", "embed_avg"):
            sd_enc["vq." + b].copy_(g.t("warm/vq." + b))


def check_step(g, s, out, PE, PD, lr, tight, tol=2e-4, grad_tol=2e-3, max_loose=2, loose_bound=0.25, recon_tol=None):
    """Compare one training step with the golden record.

    Step 0 is the tight gate.  Later steps are inherently chaotic in the REFERENCE itself: (1) cluster_size
    starts at 0, so after the first EMA update unused codes jump to ~1e5 (vq_module.py:156,198-200) and most
    pixels sit near score ties; (2) biases of convs feeding an InstanceNorm have analytically zero gradient, and
    Adam turns their rounding noise into +-lr steps.  Two fp32 implementations therefore drift apart after the
    first optimiser step; for s > 0 only coarse agreement is asserted."""
    def val(k):
        return float(out[k].detach()) if torch.is_tensor(out[k]) else float(out[k])
    if not tight:
        assert abs(val("total") - float(g["step%d/total" % s])) <= 0.05 * abs(float(g["step%d/total" % s]))
        assert abs(val("reg") - float(g["step%d/reg" % s])) <= 0.05 * abs(float(g["step%d/reg" % s]))
        for v in ("1", "2"):
            agree = np.mean(out["ids_" + v].cpu().numpy() == g["step%d/ids_%s" % (s, v)])
            assert agree > 0.5, "ids_%s agreement %.3f at step %d" % (v, agree, s)
        return
    for k in ("total", "commit", "cross", "dist", "reg", "recon"):
        assert_close(val(k), g["step%d/%s" % (s, k)], tol, "step %d %s" % (s, k))
    for v in ("1", "2"):
        ids = out["ids_" + v].cpu().numpy()
        ref = g["step%d/ids_%s" % (s, v)]
        if "step%d/gap_%s" % (s, v) in g.files:       # the reference's own gaps (step 0 of every fixture): bit-exact where clear
            assert_ids_equal_where_clear(ids, ref, g["step%d/gap_%s" % (s, v)], "ids_%s step %d" % (v, s))
        else:
            assert np.mean(ids == ref) > 0.999, "ids_%s agreement %.5f" % (v, np.mean(ids == ref))
        assert_close(out["recon_" + v], g["step%d/recon_%s" % (s, v)], 10 * tol if recon_tol is None else recon_tol, "recon_" + v)
    gmax = max(float(g[k]) for k in g.files if k.startswith("step%d/gnorm." % s))
    n_checked, loose = 0, []
    for pre, grads, P in (("enc", out["grads_enc"], PE), ("dec", out["grads_dec"], PD)):
        for k, gr in grads.items():
            ref_n = float(g["step%d/gnorm.%s.%s" % (s, pre, k)])
            if ref_n < 1e-6 * gmax:        # analytically zero (bias in front of an InstanceNorm): noise only
                assert gr is None or float(gr.norm()) < 1e-4 * gmax
                continue
            gr = gr.detach().cpu()
            idx = sample_idx(gr.numel())
            ref_s = g.t("step%d/g.%s.%s" % (s, pre, k)).double()
            err = float((gr.reshape(-1)[idx].double() - ref_s).norm()) / (float(ref_s.norm()) + ref_n / gr.numel() ** 0.5)
            err = max(err, abs(float(gr.norm()) - ref_n) / ref_n)
            if err > grad_tol:
                # scale-invariant weights (1-channel 1x1 conv feeding an InstanceNorm) only get gradient through
                # eps: ill-conditioned in any fp32 implementation.  Tolerate a couple, bounded.
                assert err < loose_bound, "grad %s.%s: error %.3e" % (pre, k, err)
                loose.append("%s.%s:%.1e" % (pre, k, err))
                continue
            # first Adam step moves every element by lr*sign(g): agree unless g ~ 0
            pv = P[k].detach().cpu().float().reshape(-1)[idx]
            ok = (pv - g.t("step%d/after.%s.%s" % (s, pre, k))).abs() < 0.1 * lr
            assert ok.float().mean() > (0.9 if loose_bound < 1 else 0.6), "after-step %s.%s" % (pre, k)
            n_checked += 1
    assert n_checked + len(loose) > 20 and len(loose) <= max_loose, loose
    for key in ("vq.embed", "vq.cluster_size", "vq.embed_avg"):
        v = PE[key].detach().cpu().float()
        c = g["step%d/after_sum.enc.%s" % (s, key)]
        assert abs(checksum(v)[1] - c[1]) <= 1e-4 * c[1], "after-step " + key
    for k, v in PD.items():
        if "running_" in k:
            c = g["step%d/after_sum.dec.%s" % (s, k)]
            assert abs(checksum(v.float())[1] - c[1]) <= 1e-4 * c[1] + 1e-6, "after-step " + k


def check_later_step(g, s, out, PE, PD, lr, loss_floor=5e-4, recon_floor=5e-3, state_floor=1e-4, what=""):
    """Steps > 0 of a fixture that carries the reference's OWN multi-step spread (`step<s>/spread.*`: the same steps run by the
    reference with the batch reversed and on one thread - mathematically identical, only fp32 association orders change).

    Every compared quantity must be within max(F x that spread, floor) of the reference's run as launched.
    * Where the reference reproduces itself (lr = 1e-6 fixture: spread <= 3e-5 on everything, ids identical) F = 2, the floors
      are the step-0 tolerances and this is a real multi-step pin: losses to `loss_floor`, ids bit-equal wherever the
      reference's gap is clear, VQ buffers after the 2nd / 3rd EMA update and BatchNorm running statistics after 2 / 3 momentum
      updates to 1e-4, parameters after Adam's t = 2, 3 updates elementwise.
    * Where it does not (lr = 1e-4: Adam moves elements whose gradient is rounding noise by +-lr at random, and the
      reference's own reconstructions differ by 16 % after one step and 66 % after two) the bound is what the reference itself
      supports: F = 4 (two variants sample a chaotic spread thinly: the oracle - the same ATen kernels as the reference -
      lands 2.3 x the two-variant spread from it at step 2) and floors ten times the step-0 tolerances.
    Returns the measured errors (printed by the callers)."""
    def val(k):
        return float(out[k].detach()) if torch.is_tensor(out[k]) else float(out[k])

    def sp(*keys):
        return max(float(np.max(g["step%d/spread.%s" % (s, k)])) for k in keys)
    rep, bad = {}, []
    rsp = sp("recon_1", "recon_2")
    stable = rsp < 1e-3
    F = 2.0 if stable else 4.0
    if not stable:
        loss_floor, recon_floor, state_floor = 10 * loss_floor, 10 * recon_floor, 10 * state_floor

    def hold(key, err, bound, note=""):
        rep[key] = err
        if not err <= bound:
            bad.append("%s step %d %s: %.3e > bound %.3e %s" % (what, s, key, err, bound, note))
    for k in ("total", "commit", "cross", "dist", "reg", "recon"):
        ref = float(g["step%d/%s" % (s, k)])
        hold(k, abs(val(k) - ref) / (abs(ref) + 1e-30), max(F * sp(k), loss_floor), "(reference spread %.3e)" % sp(k))
    ids_spread = sp("ids_1", "ids_2")
    for v in ("1", "2"):
        ids = out["ids_" + v].cpu().numpy()
        ref = g["step%d/ids_%s" % (s, v)]
        if ids_spread == 0.0:       # the reference reproduces its ids: bit-equal wherever its top-1 / top-2 gap is clear
            rep["ids_" + v] = float(np.mean(ids != ref))
            assert_ids_equal_where_clear(ids, ref, g["step%d/gap_%s" % (s, v)], "%s ids_%s step %d" % (what, v, s), rel=2e-3)
        else:                       # (one flipped Adam sign in the encoder moves a handful of pixels across a code boundary)
            hold("ids_" + v, float(np.mean(ids != ref)), F * ids_spread + 5e-3, "(fraction of differing ids, reference spread %.4f)" % ids_spread)
    # A code that differs on a pixel whose gap is NOT clear (allowed above) is a different decoder input: the quantised map is
    # piecewise constant and the decoder's low-resolution InstanceNorm planes have almost no variance, so ONE flipped pixel of
    # 8 192 moves that view's reconstruction by 16 % (measured, round 4) and with it this step's decoder gradients, parameter
    # update and BatchNorm statistics.  Those are then held to the bounds of the reference's non-reproducible regime; with
    # identical ids - the usual case - they are held tightly.
    flipped = stable and any(rep["ids_" + v] > 0.0 for v in ("1", "2"))
    rep["ids_flipped_on_near_ties"] = float(flipped)
    if flipped:
        recon_floor, state_floor_bn = 0.5, 5e-3
    for v in ("1", "2"):
        hold("recon_" + v, rel_err(out["recon_" + v], g["step%d/recon_%s" % (s, v)]), max(F * rsp, recon_floor))
    for key in ("vq.embed", "vq.cluster_size", "vq.embed_avg"):
        v = PE[key].detach().cpu().float()
        c = g["step%d/after_sum.enc.%s" % (s, key)]
        idx = sample_idx(v.numel())
        hold(key, max(rel_err(v.reshape(-1)[idx], g["step%d/after.enc.%s" % (s, key)]), abs(checksum(v)[1] - c[1]) / c[1]),
             max(F * sp(key), state_floor))
    bn = torch.cat([v.detach().reshape(-1).float().cpu() for k, v in PD.items() if "running_" in k])
    # (chaotic regime: the running statistics follow the decoder's activations, which differ by tens of per cent: floor 5e-3)
    hold("bn", rel_err(bn, g["step%d/bn_running" % s]), max(F * sp("bn"), state_floor if (stable and not flipped) else 5e-3), "(BatchNorm running statistics, spread %.3e)" % sp("bn"))
    if stable and not flipped:
        # the reference's trajectory is reproducible: parameters after Adam's step t = s + 1, elementwise on the sampled entries.
        # An update is ~lr whatever |g|; elements whose gradient is rounding noise move at random, so the statistic is the
        # fraction of sampled entries within 0.1 lr, over parameters with a real gradient
        gmax = max(float(g[k]) for k in g.files if k.startswith("step%d/gnorm." % s))
        fracs = []
        for pre, P in (("enc", PE), ("dec", PD)):
            for k, p in P.items():
                key = "step%d/gnorm.%s.%s" % (s, pre, k)
                if key not in g.files or float(g[key]) < 1e-6 * gmax:
                    continue
                pv = p.detach().cpu().float().reshape(-1)
                d = (pv[sample_idx(pv.numel())] - g.t("step%d/after.%s.%s" % (s, pre, k))).abs()
                fracs.append((float((d < 0.1 * lr).float().mean()), float((d < 0.5 * lr).float().mean()), pre + "." + k))
        rep["params_within_0.1lr_min"] = min(fracs)[0]
        hold("params_within_0.1lr_median (1 - x)", 1.0 - float(np.median([f[0] for f in fracs])), 0.1)
        hold("params_within_0.5lr_min (1 - x)", 1.0 - min(f[1] for f in fracs), 0.25, str(sorted(fracs, key=lambda f: f[1])[:3]))
    assert not bad, "\n".join(bad + [str({k: "%.2e" % v for k, v in rep.items()})])
    return rep


GRAD_TOL["step_cfg4_32.npz"] = (2e-3, 2, 0.25)      # BASELINE config 4 scaled down (K = 1024, D = 256), warm VQ state
GRAD_TOL["step_rcfg64_warm_lr1e-6.npz"] = GRAD_TOL["step_rcfg64_warm.npz"]


@pytest.mark.parametrize("name", ["step_small.npz", "step_rcfg64_warm.npz", "step_rcfg64_warm_lr1e-6.npz", "step_rcfg32.npz", "step_cfg4_32.npz"])
def test_first_step(golden, name):
    """Full first training step(s): losses, ids, recon, sampled grads, params/buffers after Adam."""
    g = golden(name)
    enc, dec = build_models(g.group("cfg"))
    check_init(g, enc, dec)
    PE = {k: v.detach().clone().contiguous() for k, v in enc.state_dict().items()}
    PD = {k: v.detach().clone().contiguous() for k, v in dec.state_dict().items()}
    apply_warm_state(g, PE)
    gt, ml, lb = GRAD_TOL[name]
    # eval-mode forward and mask-guided reconstruction (run_recon.py:179-194) on the initial state
    with torch.no_grad():
        q, _, ids1, gap = O.encoder_forward(PE, g.t("eval/image"), False, float(g["cfg/momentum"]))
        rec = O.decoder_forward(PD, q, False)
        assert_ids_equal_where_clear(ids1, g["eval/ids"], g["eval/gap"], "eval-mode ids")
        assert_close(rec, g["eval/recon"], 1e-3, "eval recon")
        rec2 = O.recon_from_ids(PE, PD, g.t("recon/label_map"))
        assert_close(rec2, g["recon/recon"], 1e-3, "mask-guided recon")
        emb = O.vq_lookup(O.vq_state(PE), torch.clamp(g.t("recon/label_map"), min=1) - 1)
        m = (g.t("recon/label_map") != 0)
        assert_close(emb * m[:, None] * (m.numel() / m.sum()), g["recon/embed"], 1e-6, "masked embed")
    tr = O.FirstStepTrainer(PE, PD, step_cfg(g))
    lr = float(g["cfg/lr"])
    for s in range(int(g["cfg/n_steps"])):
        out = tr.step(g.t("step%d/image" % s), g.t("step%d/noise" % s))
        if s > 0 and "step%d/spread.total" % s in g.files:
            print(name, "oracle, step", s, {k: "%.2e" % v for k, v in check_later_step(g, s, out, PE, PD, lr, loss_floor=2e-4, recon_floor=1e-3, what="oracle").items()})
            continue
        check_step(g, s, out, PE, PD, lr, tight=(s == 0), grad_tol=gt, max_loose=ml, loose_bound=lb)
        if s == 0:      # the oracle is one more fp32 implementation: at most 2x the reference's own distance from fp64
            grads = {"enc." + k: v for k, v in out["grads_enc"].items()}
            grads.update({"dec." + k: v for k, v in out["grads_dec"].items()})
            print(name, "oracle grad error / reference fp32 error (median, max):", check_grads_vs_fp64(g, grads, 2.0, "oracle"))


def test_extras(golden):
    """Optional paths: pixel-shuffle up block / default decoder, VQWNet monolith, DropBlock apply."""
    from networks import UNetDecoder, VQWNet
    g = golden("extras.npz")
    _run_case(g, "styled_res_up_ps", lambda P, d, s: O.styled_res_up_block(_pref(P, "m."), "m", d, s, True, pixel_shuffle=True), 2)
    torch.manual_seed(32)
    dec = UNetDecoder(16, 1, [16, 32, 32, 32, 64], use_dropblock=False, dropped_skip_layers=[])
    for k, v in dec.state_dict().items():
        assert abs(checksum(v.float())[1] - g["dec_ps/init_sum." + k][1]) <= 1e-9 * max(1.0, g["dec_ps/init_sum." + k][1]), k
    PD = {k: v.detach().clone().contiguous() for k, v in dec.state_dict().items()}
    for k in O.trainable_keys(PD):
        PD[k].requires_grad_(True)
    x = g.t("dec_ps/x").requires_grad_(True)
    y = O.decoder_forward(PD, x, True, pixel_shuffle=True)
    (y * g.t("dec_ps/R")).sum().backward()
    assert_close(y, g["dec_ps/y"], 1e-4, "decoder (pixel shuffle) y")
    assert_close(x.grad, g["dec_ps/gx"], 2e-3, "decoder (pixel shuffle) gx")
    torch.manual_seed(33)
    net = VQWNet(1, 1, [16, 16, 32, 32, 32], dict_size=6)
    P = {k: v.detach().clone().contiguous() for k, v in net.state_dict().items()}
    for k, v in P.items():
        assert abs(checksum(v.float())[1] - g["vqwnet/init_sum." + k][1]) <= 1e-9 * max(1.0, g["vqwnet/init_sum." + k][1]), k
    P["vq.embed"].mul_(0.7)
    P["vq.cluster_size"].fill_(2 * 32 * 32 / 6)
    P["vq.embed_avg"].copy_(P["vq.embed"].t() * P["vq.cluster_size"][None, :])
    out = O.vqwnet_forward(P, g.t("vqwnet/image"), True)
    assert np.mean(out["ids"].numpy() == g["vqwnet/ids"]) > 0.999
    assert_close(out["recon"], g["vqwnet/recon"], 1e-4, "vqwnet recon")
    assert_close(out["embed"], g["vqwnet/embed"], 1e-4, "vqwnet embed")
    assert_close(out["commit_loss"], g["vqwnet/commit"], 1e-4, "vqwnet commit")
    keep = O.dropblock_mask(g.t("dropblock/seed"), 4)
    assert_close(O.dropblock_apply(g.t("dropblock/x"), keep), g["dropblock/y"], 1e-6, "dropblock apply")


def test_gan_oracle(golden):
    """oracle/gan_ref.py against the reference's NLayerDiscriminator / hinge_d_loss vectors (tests/golden/gan.npz)."""
    from oracle import gan_ref as G
    g = golden("gan.npz")
    for tag, nl, train in (("dis_f16", 3, True), ("dis_f8_eval", 2, False)):
        P = {k[2:]: v.clone() for k, v in g.group(tag).items() if k.startswith("P.")}
        fl = [k for k, v in P.items() if v.is_floating_point() and "running" not in k]
        for k in fl:
            P[k].requires_grad_(True)
        x = g.t(tag + "/in.0").requires_grad_(True)
        y = G.discriminator_forward(P, x, train, n_layers=nl)
        (y * g.t(tag + "/R.0")).sum().backward()
        assert_close(y, g[tag + "/out.0"], 1e-5, tag + " out")
        assert_close(x.grad, g[tag + "/gin.0"], 1e-4, tag + " gin", atol=1e-7)
        for k in fl:
            assert_close(P[k].grad, g["%s/gP.%s" % (tag, k)], 1e-4, "%s gP.%s" % (tag, k), atol=1e-6)
        for k in P:
            key = "%s/after.%s" % (tag, k)
            if key in g.files:
                assert_close(P[k].detach().float(), g[key].astype(np.float32), 1e-5, key)
    real, fake = g.t("hinge/real").requires_grad_(True), g.t("hinge/fake").requires_grad_(True)
    l = G.hinge_d_loss(real, fake)
    (3.0 * l).backward()
    assert_close(l, g["hinge/loss"], 1e-6, "hinge")
    assert_close(real.grad, g["hinge/g_real"], 1e-6, "hinge g_real", atol=1e-9)
    x = g.t("gen/x")
    assert_close(G.generator_loss(x), g["gen/loss"], 1e-6, "gen loss", atol=1e-8)


def test_utils_fixture(golden, tmp_path):
    """f4 / a23 pinned to the reference's own utils/__init__.py (tests/golden/utils.npz, generated through the shim): CT
    window normalize / denormalize incl. odd widths (`width // 2`), the denormalize -> normalize re-windowing of
    multi_window_trainer.py:93-118 in its fused affine + clamp form, norm / denorm, load_json's false -> None."""
    import json
    from hipops import ops
    from dataio import window_normalize
    import run_recon as RR
    import utils as U
    g = golden("utils.npz")
    hu = g["hu"]
    wide = tuple(g["window/wide"])
    for name in ("default", "mediastinal", "wide", "odd"):
        w, c, sc = (float(v) for v in g["window/" + name])
        w, c = int(w), int(c)
        n_ref = g["normalize/" + name]
        assert_close(O.window_normalize(torch.from_numpy(hu.copy()), w, c, sc), n_ref, 1e-6, "oracle normalize " + name, atol=1e-7)
        assert_close(window_normalize(hu.copy(), w, c, sc), n_ref, 1e-6, "dataio normalize " + name, atol=1e-7)
        assert_close(RR.normalize(hu.copy(), w, c, sc), n_ref, 1e-6, "run_recon normalize " + name, atol=1e-7)
        assert_close(O.window_denormalize(torch.from_numpy(n_ref.copy()), w, c, sc), g["denormalize/" + name], 1e-6, "oracle denormalize")
        assert_close(RR.denormalize(n_ref.copy(), w, c, sc), g["denormalize/" + name], 1e-6, "run_recon denormalize")
        # the windowed-MSE kernel's map: clamp(alpha * x + beta, lo, hi) on values normalised with the wide dataset window
        x = torch.from_numpy(g["normalize/wide"].copy())
        a, b, lo, hi = ops.window_map((int(wide[0]), int(wide[1]), float(wide[2])), (w, c, sc))
        # re-windowing of exactly representable dataset values: denormalize(wide) then normalize(target)
        ref = O.window_normalize(O.window_denormalize(x, int(wide[0]), int(wide[1]), float(wide[2])), w, c, sc)
        assert_close(torch.clamp(a * x + b, lo, hi), ref, 1e-5, "window_map " + name, atol=2e-6)
        if name != "wide":
            inside = np.abs(g["normalize/wide"]) < 0.999          # the fixture re-windows values the wide window did not clip
            assert_close(torch.clamp(a * x + b, lo, hi)[inside], g["rewindow/" + name][inside], 1e-4, "rewindow " + name, atol=2e-5)
    # norm / denorm are exercised on the GPU (tests/test_gpu_parity.py::test_norm_denorm_golden); load_json here
    src = json.loads(str(g["load_json/source"]))
    path = tmp_path / "c.json"
    path.write_text(json.dumps(src))
    c = U.load_json(str(path))
    assert repr((c.run.seed, c.run.flag_false, c.run.flag_true, c.run.name, c.list)) == str(g["load_json/repr"])


VQ_DIST_WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
from oracle import vqwnet_ref as O
g = np.load(os.path.join(sys.argv[1], "tests", "golden", "vq_dist.npz"))
for tag in ("k10", "k64"):
    e0 = torch.from_numpy(g[tag + "/embed0"].copy())
    V = dict(embed=e0.clone(), cluster_size=torch.zeros(e0.shape[0]), embed_avg=e0.t().contiguous())
    for call in (1, 2):
        x = torch.from_numpy(g["%s/r%d/x%d" % (tag, rank, call)])
        q, ids, gap = O.vq_quantize(V, x, True, float(g[tag + "/momentum"]), world_size=world, all_reduce=lambda t: dist.all_reduce(t))
        clear = gap.numpy() > 1e-4 * (1 + np.abs(gap.numpy()))
        assert np.array_equal(ids.numpy()[clear], g["%s/r%d/ids%d" % (tag, rank, call)][clear])
        for b in ("embed", "cluster_size", "embed_avg"):
            ref = torch.from_numpy(g["%s/r%d/%s_after%d" % (tag, rank, b, call)])
            err = float((V[b] - ref).norm() / ref.norm())
            assert err < 2e-6, (tag, rank, call, b, err)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_vq_dist_oracle_vs_reference_fixture(golden, tmp_path):
    """The reference-held data-parallel fixture (VQModule under WORLD_SIZE=2, gloo; vq_module.py:187-193): the oracle's
    `world_size=2` restatement of the quirk (rank-mean embed_sum, local counts) on two gloo ranks against it; and the
    quirk is real - the two replicas' codebooks in the fixture differ."""
    import subprocess, sys
    g = golden("vq_dist.npz")
    assert not np.allclose(g["k10/r0/cluster_size_after1"], g["k10/r1/cluster_size_after1"])
    assert not np.allclose(g["k10/r0/embed_after1"], g["k10/r1/embed_after1"])
    script = tmp_path / "w.py"
    script.write_text(VQ_DIST_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), root], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


# ----------------------------------------------------------------------------------------------
# k-means codebook initialisation (unet_encoder.py:66-91): the CPU oracle against known answers
# ----------------------------------------------------------------------------------------------
def test_kmeans_oracle_recovers_separated_blobs():
    from oracle import kmeans_ref as KR
    K, P, D = 4, 512, 16
    x, lab, true = KR.blobs(P, D, K, seed=111)
    cen, ids, tr = KR.kmeans(x, K, seed=11)
    assert len(tr) < 100 and tr[-1]["shift"] ** 2 < 1e-4
    inertia = [t["inertia"] for t in tr]
    assert all(b <= a * (1 + 1e-9) for a, b in zip(inertia, inertia[1:])), inertia
    # every blob is one cluster: the assignment is the blob labelling up to a permutation, the centres are the blob means
    perm = ((cen[:, None, :] - true.numpy()[None]) ** 2).sum(-1).argmin(1)
    assert sorted(perm.tolist()) == list(range(K))
    assert np.array_equal(perm[ids], lab.numpy())
    for k in range(K):
        assert np.allclose(cen[k], x.numpy()[ids == k].mean(0), atol=1e-5)
    assert np.array_equal(cen[:0], cen[:0]) and cen.dtype == np.float32


def test_kmeans_oracle_empty_cluster_keeps_its_centre():
    from oracle import kmeans_ref as KR
    K, P, D = 8, 1024, 32
    x, _, _ = KR.blobs(P, D, K, seed=124)
    cen, ids, tr = KR.kmeans(x, K, seed=24)
    assert tr[-1]["empty"] > 0 and len(tr) > 1
    before, _, _ = KR.kmeans(x, K, seed=24, max_iter=len(tr) - 1)        # the centres the last iteration started from
    counts = np.bincount(ids, minlength=K)
    for k in np.nonzero(counts == 0)[0]:
        assert np.array_equal(cen[k], before[k])         # kmeans_pytorch 0.3.0 would write NaN here
    assert np.isfinite(cen).all()
    with pytest.raises(RuntimeError):
        KR.kmeans(x[:5], K)
