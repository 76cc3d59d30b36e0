"""Functional CPU restatement (plain torch fp32 ops) of the VQ-W-Net hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Parity status: PINNED by
tests/golden/*.npz (vectors produced by the reference's own modules).

The restatement is deliberately *functional*: networks are functions of a flat
``{key: tensor}`` state dict that uses the reference's state_dict key names, so
one set of weights can be fed to the reference (generator side), to this oracle
and to the HIP modules.  Gradients come from torch autograd on the CPU.

Reference locations restated here (paths relative to /root/reference/src):
  conv / instance-norm / ReLU stacks  networks/blocks.py:39-61   (DoubleConv)
  residual down block + max-pool      networks/blocks.py:21-36   (ResBlock)
  upsample+concat block               networks/blocks.py:9-18    (UpBlock)
  SPADE-style de-normalisation        networks/blocks.py:64-90   (StyledDenorm)
  styled residual up block            networks/blocks.py:93-134  (StyledResUpBlock)
  atrous pyramid                      networks/aspp.py:10-47
  encoder wiring                      networks/unet_encoder.py:93-123
  decoder wiring                      networks/unet_decoder.py:115-164
  nearest-codebook search             networks/vq/vq_module.py:45-62
  quantise + EMA codebook update      networks/vq/vq_module.py:159-211
  straight-through estimator          networks/vq/grad_approximation.py:7-29
  cross-view / margin / norm losses   functions/embed_loss.py:22-88
  one-hot encoding                    functions/onehot.py:11-20
  dice / focal losses                 functions/seg_loss.py:15-62
  first training step                 trainers/single_window_trainer.py:68-147
  Adam construction                   trainers/base.py:164-183
  mask-guided reconstruction          run_recon.py:169-228
  drop-block mask                     networks/dropblock.py:47-94
"""
import math

import torch
import torch.nn.functional as F

IN_EPS = 1e-5      # nn.InstanceNorm2d default (blocks.py:26,46; aspp.py:25)
BN_EPS = 1e-5      # nn.BatchNorm2d default (blocks.py:73)
BN_MOMENTUM = 0.1  # nn.BatchNorm2d default
VQ_EPS = 1e-5      # unet_encoder.py:57


# ----------------------------------------------------------------------------
# primitive layers
# ----------------------------------------------------------------------------
def conv(P, key, x, dilation=1):
    """Conv2d 'same' (k=1 or 3, stride 1, pad = dilation*(k//2)), bias if stored."""
    w = P[key + ".weight"]
    b = P.get(key + ".bias")
    k = w.shape[-1]
    return F.conv2d(x, w, b, stride=1, padding=dilation * (k // 2), dilation=dilation)


def inorm(x):
    """InstanceNorm2d(affine=False, no running stats): per (n,c) plane (x - mean) / sqrt(biased var + eps).

    Evaluated by ATen's instance-norm kernel (what nn.InstanceNorm2d runs in the reference): its backward keeps the
    plane reductions in the accumulation type.  The composite formula's autograd backward cancels in fp32 and is up to
    ~100x further from the fp64 gradient on the 160->32 layer behind the ASPP (measured with the fp64 fixtures)."""
    if x.shape[2] * x.shape[3] == 1:      # torch refuses one-element planes (the reference cannot run there): (x - x) / sqrt(eps)
        return x - x
    return F.instance_norm(x, eps=IN_EPS)


def double_conv(P, key, x):
    """blocks.py:39-61 with use_output_act=True."""
    x = torch.relu(inorm(conv(P, key + ".double_conv.0", x)))
    x = torch.relu(inorm(conv(P, key + ".double_conv.3", x)))
    return x


def res_block(P, key, x):
    """blocks.py:21-36.  Returns (pooled, unpooled)."""
    identity = inorm(conv(P, key + ".downsample.0", x))
    out = torch.relu(double_conv(P, key + ".double_conv", x) + identity)
    return F.max_pool2d(out, 2), out


def up2(x):
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


def up_block(P, key, down, skip):
    """blocks.py:9-18: channels = [upsampled down, skip]."""
    return double_conv(P, key + ".double_conv", torch.cat([up2(down), skip], dim=1))


def batch_norm(P, key, x, training):
    """BatchNorm2d(affine=False, track_running_stats=True) (blocks.py:73).

    training: batch statistics (biased var) normalise; running stats get the
    momentum-0.1 update with the *unbiased* variance.  eval: running stats.
    Mutates P[key+'.running_mean'|'.running_var'|'.num_batches_tracked'].
    """
    rm, rv = P[key + ".running_mean"], P[key + ".running_var"]
    if training and key + ".num_batches_tracked" in P:
        with torch.no_grad():
            P[key + ".num_batches_tracked"] += 1
    # ATen's batch-norm kernel, as nn.BatchNorm2d runs it in the reference (same reasoning as inorm above)
    return F.batch_norm(x, rm, rv, None, None, training, BN_MOMENTUM, BN_EPS)


def styled_denorm(P, key, x, style, training):
    """blocks.py:82-90."""
    xn = batch_norm(P, key + ".param_free_norm", x, training)
    a = torch.relu(conv(P, key + ".mlp_shared.0", style))
    gamma = conv(P, key + ".mlp_gamma", a)
    beta = conv(P, key + ".mlp_beta", a)
    return xn * (1 + gamma) + beta


def styled_res_up_block(P, key, down, skip, training, pixel_shuffle=False):
    """blocks.py:122-134."""
    if pixel_shuffle:
        x = F.pixel_shuffle(conv(P, key + ".up_sample.0", down), 2)
    else:
        x = up2(down)
    s = torch.relu(inorm(conv(P, key + ".conv.0", x)))
    x = torch.relu(styled_denorm(P, key + ".norm1", conv(P, key + ".conv1", x), skip, training))
    x = torch.relu(styled_denorm(P, key + ".norm2", conv(P, key + ".conv2", x), skip, training))
    return s + x


def aspp(P, key, x, rates=(2, 6, 12, 18)):
    """aspp.py:31-47: 1x1 and dilated 3x3 branches, each conv(no bias)->IN->ReLU, concat."""
    outs = [torch.relu(inorm(conv(P, key + ".stages.c0.conv", x)))]
    for i, r in enumerate(rates):
        outs.append(torch.relu(inorm(conv(P, key + ".stages.c%d.conv" % (i + 1), x, dilation=r))))
    return torch.cat(outs, dim=1)


# ----------------------------------------------------------------------------
# vector quantisation
# ----------------------------------------------------------------------------
def vq_scores(embed, flat):
    """vq_module.py:50-58: 2 k.q - |k|^2 - |q|^2, (K, N)."""
    s = embed @ flat.t()
    s = s * 2
    s = s - (embed * embed).sum(1, keepdim=True)
    s = s - (flat * flat).sum(1)[None, :]
    return s


def vq_quantize(V, x, training, momentum, world_size=1, all_reduce=None):
    """vq_module.py:168-202 on x (B,D,H,W), H == W.

    V holds 'embed' (K,D), 'cluster_size' (K), 'embed_avg' (D,K) and is updated in
    place when training.  Returns (quantized (B,D,H,W), ids (B,H,W) int64 with
    ids[b,h,w] = code of pixel (h,w), 0-based, top1-top2 score gap (B,H,W)).

    The reference flattens in (B,W,H) order and reshapes ids as (b,h,w)
    (vq_module.py:172,178-180); the encoder transposes them back
    (unet_encoder.py:115).  For H == W the net effect is the per-pixel mapping
    restated here.  EMA sums are order independent.

    world_size>1 reproduces the reference's distributed quirk (vq_module.py:187-193):
    embed_sum is rank-averaged, counts stay local.  `all_reduce` is then a callable
    summing a tensor over ranks in place.
    """
    B, D, H, W = x.shape
    assert H == W, "reference VQ is only self-consistent for square maps"
    embed = V["embed"]
    K = embed.shape[0]
    flat = x.detach().permute(0, 2, 3, 1).reshape(-1, D)
    s = vq_scores(embed, flat)
    top2 = s.topk(k=min(2, K), dim=0).values
    gap = (top2[0] - top2[1]) if K > 1 else torch.full_like(top2[0], float("inf"))
    ids = s.argmax(dim=0)
    quant = embed[ids].reshape(B, H, W, D).permute(0, 3, 1, 2).contiguous()
    if training:
        counts = torch.bincount(ids, minlength=K).to(flat.dtype)
        esum = torch.zeros(K, D, dtype=flat.dtype).index_add_(0, ids, flat).t().contiguous()
        if world_size > 1:
            all_reduce(esum)
            esum = esum / world_size
        V["cluster_size"].mul_(momentum).add_(counts, alpha=1 - momentum)
        V["embed_avg"].mul_(momentum).add_(esum, alpha=1 - momentum)
        n = V["cluster_size"].sum()
        cs = n * (V["cluster_size"] + VQ_EPS) / (n + K * VQ_EPS)
        V["embed"].copy_(V["embed_avg"].t() / cs[:, None])
    return quant, ids.reshape(B, H, W), gap.reshape(B, H, W)


def vq_forward(V, x, training, momentum, **kw):
    """vq_module.py:159-166 + grad_approximation.py: returns (q_ste, commit, ids0, gap)."""
    with torch.no_grad():
        quant, ids, gap = vq_quantize(V, x, training, momentum, **kw)
    commit = F.mse_loss(x, quant)
    # forward must be `quant` BIT-EXACTLY (x - x.detach() is exactly 0): the quantised map is piecewise constant,
    # so the decoder's max-pools sit on exact ties and a 1-ulp change re-routes their gradients.
    q_ste = quant + (x - x.detach())   # backward = identity onto x
    return q_ste, commit, ids, gap


def vq_lookup(V, ids0):
    """vq_module.py:204-207 via unet_encoder.py:120-123: ids0 (B,H,W) 0-based -> (B,D,H,W)."""
    return V["embed"][ids0].permute(0, 3, 1, 2).contiguous()


# ----------------------------------------------------------------------------
# networks
# ----------------------------------------------------------------------------
def vq_state(P, prefix="vq."):
    return {k: P[prefix + k] for k in ("embed", "cluster_size", "embed_avg")}


def encoder_features(P, x):
    """unet_encoder.py:93-103 (plain UpBlock variant)."""
    x, s1 = res_block(P, "down_conv1_1", x)
    x, s2 = res_block(P, "down_conv1_2", x)
    x, s3 = res_block(P, "down_conv1_3", x)
    x, s4 = res_block(P, "down_conv1_4", x)
    x = double_conv(P, "double_conv1", x)
    x = up_block(P, "up_conv1_4", x, s4)
    x = up_block(P, "up_conv1_3", x, s3)
    x = up_block(P, "up_conv1_2", x, s2)
    x = up_block(P, "up_conv1_1", x, s1)
    return x


def encoder_forward(P, x, training, momentum, **kw):
    """unet_encoder.py:105-118: (quantized, commit, ids 1-based (B,H,W), gap)."""
    feat = encoder_features(P, x)
    q, commit, ids0, gap = vq_forward(vq_state(P), feat, training, momentum, **kw)
    return q, commit, ids0 + 1, gap


def decoder_forward(P, x, training, n_levels=4, dropped_skip_layers=(), pixel_shuffle=False,
                    skip_fn=None):
    """unet_decoder.py:115-164 (use_last_pixel_shuffle=False branch)."""
    skips = []
    for i in range(n_levels):
        x, sk = res_block(P, "down_conv2_%d" % (i + 1), x)
        skips.append(sk)
    x = double_conv(P, "double_conv2", x)
    skips.reverse()
    for j, sk in enumerate(skips):
        lvl = n_levels - j
        if j in dropped_skip_layers:
            sk = torch.zeros_like(sk)
        elif skip_fn is not None:
            sk = skip_fn(sk)
        x = styled_res_up_block(P, "up_conv2_%d" % lvl, x, sk, training, pixel_shuffle)
    y = x + double_conv(P, "conv_last.1", aspp(P, "conv_last.0", x))
    return torch.tanh(conv(P, "conv1x1", y))


def vqwnet_forward(P, x, training, momentum=0.99):
    """networks/vqwnet.py:96-152 (freeze_first_half=False, no dropblock): two U-Nets in series around the VQ."""
    feat = encoder_features(P, x)
    q, commit, ids0, gap = vq_forward(vq_state(P), feat, training, momentum)
    h = q
    skips = []
    for i in range(4):
        h, sk = res_block(P, "down_conv2_%d" % (i + 1), h)
        skips.append(sk)
    h = double_conv(P, "double_conv2", h)
    for lvl in (4, 3, 2, 1):
        h = up_block(P, "up_conv2_%d" % lvl, h, skips[lvl - 1])
    recon = torch.tanh(conv(P, "conv_last", h))
    return {"recon": recon, "embed": feat, "commit_loss": commit, "ids": ids0 + 1, "gap": gap, "quantized": q}


# ----------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------
def one_hot(t, n_classes):
    """onehot.py:11-20: (B,H,W) int -> (B,n,H,W) float."""
    return F.one_hot(t.long(), n_classes).permute(0, 3, 1, 2).contiguous().float()


def cross_loss(embed, r, codebook, eps=1e-6):
    """embed_loss.py:46-66 in segmented form.

    embed (B,D,H,W), r (B,K,H,W) weights (one-hot in practice), codebook (D,K) detached.
    per[b,k] = sum_p r*|e_p - c_k|^2 / (sum_p r + eps); mean over (b,k) with sum_p r > 0.
    """
    B, D = embed.shape[:2]
    K = r.shape[1]
    e = embed.reshape(B, D, -1)
    rr = r.reshape(B, K, -1)
    c = codebook.detach()
    d2 = ((e[:, :, None, :] - c[None, :, :, None]) ** 2).sum(1)        # (B,K,n)
    num = (d2 * rr).sum(2)
    cnt = rr.sum(2)
    per = num / (cnt + eps)
    return per[cnt != 0].mean()


def distance_loss(codebook, margin):
    """embed_loss.py:68-84 (i == j terms included, as upstream)."""
    D, K = codebook.shape
    diff = codebook[:, :, None] - codebook[:, None, :]
    dist = torch.sqrt((diff * diff).sum(0))
    return (torch.clamp(2 * margin - dist, min=0) ** 2).sum() / (2 * K * (K - 1))


def regularization_loss(codebook):
    """embed_loss.py:86-88."""
    return torch.sqrt((codebook * codebook).sum(0)).mean()


def embedding_loss(e1, r1, e2, r2, codebook, margin, use_dist=True, use_reg=True):
    """embed_loss.py:22-44."""
    l_cross = cross_loss(e1, r2, codebook) + cross_loss(e2, r1, codebook)
    l_dist = distance_loss(codebook, margin) if use_dist else 0.0
    l_reg = regularization_loss(codebook) if use_reg else 0.0
    return l_cross, l_dist, l_reg


def soft_dice_loss(logits, target, ignore_index=None, smooth=1e-6):
    """seg_loss.py:15-43."""
    p = torch.softmax(logits, dim=1)
    C = p.shape[1]
    pf = p.transpose(0, 1).reshape(C, -1)
    tf = target.float().transpose(0, 1).reshape(C, -1)
    inter = (pf * tf).sum(-1)
    den = pf.sum(-1) + tf.sum(-1)
    if ignore_index is not None:
        keep = [i for i in range(C) if i != ignore_index]
        inter, den = inter[keep], den[keep]
    return 1.0 - 2.0 * inter.sum() / den.sum().clamp(min=smooth)


def focal_loss(logits, target, gamma=2, eps=1e-6):
    """seg_loss.py:46-62."""
    p = torch.softmax(logits, dim=1).clamp(min=eps, max=1 - eps)
    lp = torch.log_softmax(logits, dim=1)
    return ((-target * lp) * (1.0 - p) ** gamma).sum(1).mean()


def dropblock_mask(seed_mask, block_size):
    """dropblock.py:79-91: seed_mask (B,H,W) in {0,1} -> keep mask (B,H,W)."""
    m = F.max_pool2d(seed_mask[:, None], kernel_size=block_size, stride=1, padding=block_size // 2)
    if block_size % 2 == 0:
        m = m[:, :, :-1, :-1]
    return 1 - m[:, 0]


def dropblock_apply(x, keep):
    """dropblock.py:68-74."""
    return x * keep[:, None] * keep.numel() / keep.sum()


# ----------------------------------------------------------------------------
# optimiser + the training step
# ----------------------------------------------------------------------------
class Adam:
    """torch.optim.Adam semantics (base.py:165-175): L2 weight decay added to the grad,
    bias-corrected, eps outside the sqrt-of-corrected-second-moment."""

    def __init__(self, params, lr, betas, weight_decay=0.0, eps=1e-8):
        self.params = list(params)
        self.lr, (self.b1, self.b2), self.wd, self.eps = lr, betas, weight_decay, eps
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    @torch.no_grad()
    def step(self, grads):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        for p, g, m, v in zip(self.params, grads, self.m, self.v):
            if g is None:
                continue
            if self.wd != 0:
                g = g + self.wd * p
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(m, denom, value=-self.lr / bc1)


FLOAT_PARAM_SUFFIXES = (".weight", ".bias")


def trainable_keys(P):
    return [k for k in P if k.endswith(FLOAT_PARAM_SUFFIXES)]


def make_views(image, noise):
    """Exact-integer stand-ins for the kornia views: view 1 = identity,
    view 2 = horizontal flip; photometric noise only on the 'noised' copy of view 2."""
    v1n, v1c = image, image
    v2c = torch.flip(image, dims=[3])
    v2n = v2c + noise
    return (v1n, v1c), (v2n, v2c)


def cross_view_ids(ids, border=0):
    """r_ids = T_other.forward(T_self.reverse(ids)) for identity/h-flip views = h-flip of ids;
    an optional zero border plays the role of out-of-frame pixels (class 0)."""
    r = torch.flip(ids, dims=[2]).clone()
    if border > 0:
        r[:, :border, :] = 0
        r[:, -border:, :] = 0
        r[:, :, :border] = 0
        r[:, :, -border:] = 0
    return r


def first_step_losses(PE, PD, image, noise, cfg, training=True):
    """single_window_trainer.py:68-137 restated (MSE recon; no freq/perceptual).

    PE / PD: encoder / decoder state dicts (float params must be leaf tensors with
    requires_grad for a backward).  cfg keys: momentum, dict_size, margin, weights
    (dict commit/cross/dist/reg/recon), border, n_levels.
    """
    K = cfg["dict_size"]
    w = cfg["weights"]
    (n1, c1), (n2, c2) = make_views(image, noise)
    e1, lc1, ids1, gap1 = encoder_forward(PE, n1, training, cfg["momentum"])
    e2, lc2, ids2, gap2 = encoder_forward(PE, n2, training, cfg["momentum"])
    l_commit = lc1 + lc2
    r1 = one_hot(cross_view_ids(ids1, cfg.get("border", 0)), K + 1)[:, 1:]
    r2 = one_hot(cross_view_ids(ids2, cfg.get("border", 0)), K + 1)[:, 1:]
    codebook = PE["vq.embed"].t()
    l_cross, l_dist, l_reg = embedding_loss(e1, r1, e2, r2, codebook, cfg["margin"])
    rec1 = decoder_forward(PD, e1, training, cfg.get("n_levels", 4))
    rec2 = decoder_forward(PD, e2, training, cfg.get("n_levels", 4))
    l_recon = F.mse_loss(rec1, c1) + F.mse_loss(rec2, c2)
    total = (w["commit"] * l_commit + w["cross"] * l_cross + w["dist"] * l_dist
             + w["reg"] * l_reg + w["recon"] * l_recon)
    return {
        "total": total, "commit": l_commit, "cross": l_cross, "dist": l_dist, "reg": l_reg,
        "recon": l_recon, "ids_1": ids1, "ids_2": ids2, "gap_1": gap1, "gap_2": gap2,
        "recon_1": rec1, "recon_2": rec2, "embed_1": e1, "embed_2": e2,
    }


class FirstStepTrainer:
    """Holds encoder/decoder state + two Adam instances and runs training steps."""

    def __init__(self, PE, PD, cfg):
        self.PE, self.PD, self.cfg = PE, PD, cfg
        self.ek, self.dk = trainable_keys(PE), trainable_keys(PD)
        for k in self.ek:
            PE[k].requires_grad_(True)
        for k in self.dk:
            PD[k].requires_grad_(True)
        o = cfg["optim"]
        self.eopt = Adam([PE[k] for k in self.ek], o["lr"], o["betas"], o.get("weight_decay", 0.0))
        self.dopt = Adam([PD[k] for k in self.dk], o["lr"], o["betas"], o.get("weight_decay", 0.0))

    def step(self, image, noise):
        out = first_step_losses(self.PE, self.PD, image, noise, self.cfg, training=True)
        params = [self.PE[k] for k in self.ek] + [self.PD[k] for k in self.dk]
        grads = torch.autograd.grad(out["total"], params, allow_unused=True)
        ge, gd = grads[:len(self.ek)], grads[len(self.ek):]
        self.eopt.step(ge)
        self.dopt.step(gd)
        out["grads_enc"] = dict(zip(self.ek, ge))
        out["grads_dec"] = dict(zip(self.dk, gd))
        return out


def recon_from_ids(PE, PD, label_map, n_levels=4):
    """run_recon.py:179-194: label map (B,H,W) int, 0 = masked -> recon (eval mode)."""
    mask = (label_map != 0)
    ids0 = torch.clamp(label_map, min=1) - 1
    embed = vq_lookup(vq_state(PE), ids0)
    embed = embed * mask[:, None]
    embed = embed * (mask.numel() / mask.sum())
    return decoder_forward(PD, embed, training=False, n_levels=n_levels)


def synthetic_slices(batch, size, seed, device="cpu"):
    """SURVEY §8(d) synthetic inputs: smooth random field + noise, clamped to [-1,1]."""
    g = torch.Generator().manual_seed(seed)
    low = torch.randn(batch, 1, max(size // 8, 1), max(size // 8, 1), generator=g)
    field = F.interpolate(low, size=(size, size), mode="bilinear", align_corners=False)
    img = (field * 0.6 + 0.05 * torch.randn(batch, 1, size, size, generator=g)).clamp_(-1, 1)
    noise = 0.02 * torch.randn(batch, 1, size, size, generator=g)
    return img.to(device), noise.to(device)


# ----------------------------------------------------------------------------------------------
# CT windows and the multi-window reconstruction loss (utils/__init__.py:17-51, trainers/base.py:33-43, 290-314,
# trainers/multi_window_trainer.py:93-118)
# ----------------------------------------------------------------------------------------------
LUNG_WINDOW = dict(width=1500, center=-550, scale=2.0)
MEDIASTINAL_WINDOW = dict(width=400, center=20, scale=2.0)


def window_normalize(image, width=1500, center=-550, scale=2.0):
    """utils/__init__.py:17-28 on a tensor: clip to the window, map to [-scale/2, scale/2]."""
    vmax, vmin = center + width // 2, center - width // 2
    return ((torch.clamp(image, vmin, vmax) - vmin) / (vmax - vmin) - 0.5) * scale


def window_denormalize(image, width, center, scale):
    """utils/__init__.py:43-51"""
    vmax, vmin = center + width // 2, center - width // 2
    return (image / scale + 0.5) * (vmax - vmin) + vmin


def to_window(image, dataset_window, target_window):
    """base.py:290-314 (to_lung / to_mediastinal): dataset units -> Hounsfield units -> target window."""
    return window_normalize(window_denormalize(image, **dataset_window), **target_window)


def multi_window_recon(rec_1, clear_1, rec_2, clear_2, dataset_window, recon_weights):
    """multi_window_trainer.py:93-118: mean over {full, lung, mediastinal} of w_i * (MSE view 1 + MSE view 2)."""
    terms = []
    for i, tw in enumerate((None, LUNG_WINDOW, MEDIASTINAL_WINDOW)):
        f = (lambda t: t) if tw is None else (lambda t, tw=tw: to_window(t, dataset_window, tw))
        terms.append(recon_weights[i] * (F.mse_loss(f(rec_1), f(clear_1)) + F.mse_loss(f(rec_2), f(clear_2))))
    return torch.mean(torch.stack(terms))
