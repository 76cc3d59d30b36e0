"""Shared helpers for the parity tests."""
import numpy as np
import torch


def rel_err(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def assert_close(a, b, rtol, what="", atol=0.0):
    """|a-b|_2 <= rtol*|b|_2 + atol*sqrt(numel).  atol covers quantities that are analytically zero
    (e.g. the bias gradient of a conv feeding an InstanceNorm), where only rounding noise is left."""
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape or a.numel() == b.numel(), "%s: shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    d = float((a.reshape(-1) - b.reshape(-1)).norm())
    bound = rtol * float(b.norm()) + atol * (b.numel() ** 0.5)
    assert d <= bound, "%s: |diff| %.3e > bound %.3e (rel %.3e)" % (what, d, bound, d / (float(b.norm()) + 1e-30))


def assert_ids_equal_where_clear(ids, ref_ids, gap, what="", rel=1e-4):
    """Codebook indices must be BIT-EXACT wherever the reference's top-1 / top-2 score gap is clear of fp32 rounding
    (gap > 1e-4 * (1 + |gap|), the bar of DESIGN.md section 2); the remaining pixels - score ties to rounding in the
    reference itself - must still agree almost everywhere.  `gap` comes from the reference (fixture key `.../gap*`) or,
    where the reference cannot travel, from the oracle.  `rel`: 1e-4 while both sides hold bit-identical weights (step 0, eval);
    later training steps pass 2e-3 - after s optimiser steps the two sides' features differ by ~1e-4 (measured: reconstructions
    6e-5 ... 9e-5 apart), which moves a score gap by up to ~1e-3.  For view 2 the gap is the one under the codebook AFTER view
    1's EMA update of the same step (make_golden.py), the codebook its decision is really taken with."""
    ids = np.asarray(ids.detach().cpu() if torch.is_tensor(ids) else ids)
    ref_ids = np.asarray(ref_ids.detach().cpu() if torch.is_tensor(ref_ids) else ref_ids)
    gap = np.asarray(gap.detach().cpu() if torch.is_tensor(gap) else gap, dtype=np.float64)
    assert ids.shape == ref_ids.shape == gap.shape, "%s: shapes %s %s %s" % (what, ids.shape, ref_ids.shape, gap.shape)
    clear = gap > rel * (1.0 + np.abs(gap))
    bad = clear & (ids != ref_ids)
    assert not bad.any(), "%s: %d ids differ on tie-free pixels (smallest gap among them %.3e)" % (what, int(bad.sum()), float(gap[bad].min()))
    assert np.mean(ids == ref_ids) > 0.999, "%s: agreement %.5f" % (what, np.mean(ids == ref_ids))
    return float(clear.mean())


def checksum(t):
    t = t.detach().double().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), float(t.numel())])


def sample_idx(numel, n=64, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, numel, (min(n, numel),), generator=g)


def build_models(cfg_group, device="cpu"):
    """Construct encoder/decoder from a golden step fixture's cfg (weights regenerated from the seed)."""
    from networks import UNetEncoder, UNetDecoder
    enc_f = [int(v) for v in cfg_group["enc_filters"]]
    dec_f = [int(v) for v in cfg_group["dec_filters"]]
    K = int(cfg_group["K"])
    torch.manual_seed(int(cfg_group["seed"]))
    enc = UNetEncoder(1, enc_f, K, float(cfg_group["momentum"]), "torch", False, 4, True)
    dec = UNetDecoder(enc_f[0], 1, dec_f, use_dropblock=False, dropped_skip_layers=[],
                      use_styled_up_block=True, use_pixel_shuffle=False)
    return enc.to(device), dec.to(device)


def check_init(g, enc, dec):
    for pre, m in (("enc", enc), ("dec", dec)):
        for k, v in m.state_dict().items():
            c = g["init_sum/%s.%s" % (pre, k)]
            s = checksum(v.float())
            assert s[2] == c[2] and abs(s[0] - c[0]) <= 1e-9 * max(1.0, abs(c[0])) and abs(s[1] - c[1]) <= 1e-9 * max(1.0, c[1]), \
                "initial %s.%s differs from the reference's initialisation" % (pre, k)


def step_cfg(g):
    return dict(dict_size=int(g["cfg/K"]), margin=float(g["cfg/margin"]), border=int(g["cfg/border"]),
                momentum=float(g["cfg/momentum"]),
                weights=dict(commit=1.0, cross=1.0, dist=1.0, reg=1.0, recon=1.0),
                optim=dict(lr=float(g["cfg/lr"]), betas=tuple(float(b) for b in g["cfg/betas"]), weight_decay=0.0))


SPREAD_STAT = "median"


def check_grads_vs_fp64(g, grads, factor=2.0, what=""):
    """Principled end-to-end gradient gate.  The step fixtures hold, per parameter, 256 sampled entries of the gradient the
    REFERENCE modules produce in fp64 (`step0/g64.*`) and of four fp32 evaluations of the same reference step that are
    mathematically identical and differ only in the association order of fp32 reductions (`step0/g32v{0..3}.*`: as
    run_vqwnet would run it, batch order reversed, one thread instead of eight, both).  Their distance from the fp64
    gradient is the reference's own fp32 spread - 10-100x larger than any single run suggests (the one-thread run of
    the config-4 fixture is 4e-3 from fp64, the eight-thread run 8e-6).  The implementation under test must be at most
    `factor` times that spread from the fp64 gradient, per parameter, with the median spread over all parameters as
    the floor (rounding errors are random: a parameter on which the reference happens to be exact binds nobody).

    `grads`: {"enc.<name>" / "dec.<name>": tensor}.  Returns (median, max) of err_test / max(spread, floor)."""
    names = [k[len("step0/g64."):] for k in g.files if k.startswith("step0/g64.")]
    assert names, "fixture has no fp64 gradients"
    n64 = {k: float(g["step0/gnorm64." + k]) for k in names}
    gmax = max(n64.values())
    live = [k for k in names if n64[k] >= 1e-6 * gmax]
    nvar = len([1 for f in g.files if f.startswith("step0/g32v") and f.endswith("." + names[0])])

    def err(sample, k):
        ref = torch.from_numpy(np.asarray(g["step0/g64." + k])).double()
        numel = int(np.prod(grads[k].shape)) if grads.get(k) is not None else 1
        return float((sample.double() - ref).norm()) / (float(ref.norm()) + n64[k] / numel ** 0.5)

    # per variant: the larger of its sampled and its whole-tensor error; the parameter's spread is the MEDIAN over the
    # variants, not their maximum: the one-thread evaluations sit 20-400x further from fp64 than the run as launched (ATen's
    # single-thread normalisation path, not the mathematics), and a gate scaled by the worst of them would let a 1 % indexing
    # error of a kernel through.  SPREAD_STAT = "max" restores the old, looser gate for comparison.
    def variant_errs(k):
        whole = np.asarray(g["step0/gerr32." + k], dtype=np.float64).reshape(-1)
        return [max(err(torch.from_numpy(np.asarray(g["step0/g32v%d.%s" % (i, k)])), k), float(whole[i]) if i < whole.size else 0.0)
                for i in range(nvar)]
    stat = np.max if SPREAD_STAT == "max" else np.median
    spread = {k: float(stat(variant_errs(k))) for k in live}
    floor = float(np.median(list(spread.values())))
    ratios, over = [], []
    for k in names:
        gr = grads.get(k)
        if k not in live:          # analytically zero (bias in front of an InstanceNorm): rounding noise only
            assert gr is None or float(gr.norm()) < 1e-4 * gmax, "%s %s: analytically zero gradient has norm %.3e" % (what, k, float(gr.norm()))
            continue
        assert gr is not None, "%s %s: no gradient" % (what, k)
        gr = gr.detach().cpu()
        idx = sample_idx(gr.numel(), 256, seed=1)
        e = err(gr.reshape(-1)[idx], k)
        e = max(e, abs(float(gr.double().norm()) - n64[k]) / n64[k])
        ref_e = max(spread[k], floor)
        ratios.append(e / ref_e)
        msg = "%s grad %s: %.3e from the fp64 gradient, the reference's own fp32 spread is %.3e (median over parameters %.3e)" % (
            what, k, e, spread[k], floor)
        # four evaluations sample the reference's spread thinly: a fifth independent fp32 evaluation exceeds twice their
        # maximum on a few parameters by chance alone (one ReLU / max-pool decision that flips on a small plane).  Hence:
        # one such flip moves the gradients of every layer upstream of it, so the exceedances come in groups: every parameter
        # within 3 x factor, at most 10 % of them beyond factor, and the median within factor.
        assert e <= 3 * factor * ref_e, msg
        if e > factor * ref_e:
            over.append(msg)
    assert len(over) <= max(2, int(0.10 * len(ratios))), "\n".join(over)
    assert float(np.median(ratios)) <= factor, "median error ratio %.2f" % float(np.median(ratios))
    return float(np.median(ratios)), float(np.max(ratios))


def grad_gate(truth, variants, test, factor=2.0, what="", max_over_frac=0.10):
    """The same gate on whole tensors: `truth` {name: fp64 gradient}, `variants` list of {name: fp32 gradient} from
    mathematically equivalent evaluations of the reference / oracle, `test` {name: gradient under test}.
    Returns (median, max) of err_test / max(spread, median spread)."""
    gmax = max(float(g.norm()) for g in truth.values() if g is not None)
    live = [k for k, g in truth.items() if g is not None and float(g.norm()) >= 1e-6 * gmax]

    def err(g, k):
        return float((g.detach().double().cpu() - truth[k]).norm() / truth[k].norm())
    # a parameter's spread is the MEDIAN over the variants like check_grads_vs_fp64's (round 4; it was their maximum): the
    # one-thread evaluations sit far further from fp64 than the run as launched, and a gate scaled by the worst of them would
    # let an indexing error through.  SPREAD_STAT = "max" restores the looser gate for comparison.
    stat = max if SPREAD_STAT == "max" else (lambda xs: float(np.median(list(xs))))
    spread = {k: stat([err(v[k], k) for v in variants]) for k in live}
    floor = float(np.median(list(spread.values())))
    ratios, over = [], []
    for k in live:
        e = err(test[k], k)
        ref_e = max(spread[k], floor)
        ratios.append(e / ref_e)
        msg = "%s grad %s: %.3e from the fp64 gradient, fp32 spread of the oracle %.3e (median %.3e)" % (what, k, e, spread[k], floor)
        if e > factor * ref_e:
            over.append((e / ref_e, msg))
    print("%s gradient gate: median ratio %.2f, max %.2f, %d of %d parameters beyond %.0fx (allowed %d)" % (
        what, float(np.median(ratios)), float(np.max(ratios)), len(over), len(ratios), factor, max(2, int(max_over_frac * len(ratios)))))
    worst = max(over)[1] if over else ""
    assert float(np.max(ratios)) <= 3 * factor, worst
    assert len(over) <= max(2, int(max_over_frac * len(ratios))), "\n".join(m for _, m in sorted(over, reverse=True)[:8])
    assert float(np.median(ratios)) <= factor, "median error ratio %.2f" % float(np.median(ratios))
    return float(np.median(ratios)), float(np.max(ratios))
