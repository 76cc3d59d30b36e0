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
