"""First training step of the VQ-W-Net (reference: trainers/single_window_trainer.py:68-147,
construction per trainers/base.py:164-237, 261-278) on the MI355X kernels.

PyTorch-Lightning and kornia are not part of this build: the trainer is a plain class that
owns the same modules, optimisers and loss weighting and exposes `training_step(batch)`.
The two augmented views are produced by a `views` callable; the default is the exact-integer
pair (identity, horizontal flip + additive noise on the noised copy) whose cross-view id
map is an index flip.
"""
from collections import namedtuple

import torch

from hipops import ops, Adam
from networks import UNetEncoder, UNetDecoder
from functions import EmbeddingLoss, OneHotEncoder
from utils import norm, denorm
from .data_parallel import GradientAllReducer

LossWeights = namedtuple("LossWeights", "commit cross dist reg recon freq perceptual", defaults=(1.0,) * 5 + (0.0, 0.0))


class StepThrottle:
    """At most `max_inflight` training steps enqueued ahead of the GPU (VQW_MAX_INFLIGHT, default 2).

    The host enqueues a step five times faster than the GPU runs it.  Unthrottled it gets many steps ahead, and every
    tensor that was handed to another stream (record_stream: conv inputs / gradients used by the weight-gradient lanes,
    the second view) cannot be reused by the caching allocator until the GPU has passed its last use: the allocator
    then hipMallocs a fresh working set for every step in flight (+10 GB of reserved memory per step measured, with
    sporadic stalls of 0.3-1 s in those calls).  Two steps in flight keep the GPU fed and the pool bounded."""

    def __init__(self, device):
        import collections
        import os
        self.cuda = torch.device(device).type == "cuda"
        self.events = collections.deque()
        self.max_inflight = max(1, int(os.environ.get("VQW_MAX_INFLIGHT", "2")))

    def begin(self):
        while len(self.events) >= self.max_inflight:
            self.events.popleft().synchronize()

    def end(self):
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.events.append(ev)


class FlipViews:
    """view 1 = identity, view 2 = horizontal flip; noise (if given) only on the noised copy of view 2.
    `cross_ids(ids, which)` maps a view's id map into the other view's frame (flip), zeroing a border."""

    def __init__(self, border=0):
        self.border = border

    def __call__(self, image, noise=None):
        flipped = torch.flip(image, dims=[3])
        noised2 = flipped if noise is None else ops.add(flipped, noise)
        return (image, image), (noised2, flipped)

    def cross_ids(self, ids, which=1):
        return ops.flip_labels(ids, self.border)


class RandomTransformViews:
    """The reference's pair of RandomTransform modules (trainers/base.py:280-282) behind the `views` protocol:
    images arrive in [-1, 1], are augmented in [0, 1] and handed back in [-1, 1] (single_window_trainer.py:73-83);
    cross_ids(ids, which) = other.forward_transform(own.reverse_transform(ids)) (:91-96)."""

    def __init__(self, transform_1, transform_2):
        self.t = (transform_1, transform_2)

    def __call__(self, image, noise=None):
        x = denorm(image.clone(), vmin=0, vmax=1)           # norm / denorm work in place: keep the batch intact
        n1, c1 = self.t[0](x.clone())
        n2, c2 = self.t[1](x)
        return (norm(n1), norm(c1)), (norm(n2), norm(c2))

    def cross_ids(self, ids, which=1):
        own, other = (self.t[0], self.t[1]) if which == 1 else (self.t[1], self.t[0])
        return other.forward_transform(own.reverse_transform(ids))


LUNG_WINDOW = (1500, -550, 2.0)             # trainers/base.py:33-43
MEDIASTINAL_WINDOW = (400, 20, 2.0)


class FirstStepTrainer:
    def __init__(self, in_channels=1, enc_filters=(16, 32, 64, 128, 256), dec_filters=(32, 64, 128, 256, 512),
                 dict_size=10, momentum=0.999, margin=0.5, loss_weight=None, lr=1e-4, betas=(0.5, 0.999),
                 weight_decay=0.0, use_pixel_shuffle=False, dropped_skip_layers=(), views=None, device="cuda",
                 encoder=None, decoder=None, data_parallel=False, use_onehot=False, concurrent_views=None,
                 multi_window=None, embed_loss=None, enc_optim=None, dec_optim=None, use_recon_loss=True):
        self.device = torch.device(device)
        self.encoder = encoder if encoder is not None else UNetEncoder(
            in_channels, list(enc_filters), dict_size, momentum, 'torch', False, 1, True)
        self.decoder = decoder if decoder is not None else UNetDecoder(
            enc_filters[0], in_channels, list(dec_filters), use_dropblock=False,
            dropped_skip_layers=list(dropped_skip_layers), use_styled_up_block=True, use_pixel_shuffle=use_pixel_shuffle)
        self.encoder.to(self.device).train()
        self.decoder.to(self.device).train()
        self.dict_size = dict_size
        self.embed_loss = embed_loss if embed_loss is not None else EmbeddingLoss(dict_size, margin, True, True)
        self.use_recon_loss = bool(use_recon_loss)       # config.loss.use_recon_loss (single_window_trainer.py:110-115)
        self.one_hot_encoder = OneHotEncoder(n_classes=dict_size + 1)
        self.use_onehot = use_onehot
        self.w = loss_weight if loss_weight is not None else LossWeights()
        self.views = views if views is not None else FlipViews()
        # multi_window = dict(dataset_window=(width, center, scale), recon_weights=(w_full, w_lung, w_mediastinal)): the
        # reconstruction term of trainers/multi_window_trainer.py:93-118 (same step otherwise)
        self.multi_window = multi_window
        # base.py:165-175: one Adam per sub-network over its trainable parameters
        # (enc_optim / dec_optim: dicts with lr, betas, weight_decay per sub-network, as config.enc_optim / dec_optim give)
        eo = {**dict(lr=lr, betas=betas, weight_decay=weight_decay), **(enc_optim or {})}
        do = {**dict(lr=lr, betas=betas, weight_decay=weight_decay), **(dec_optim or {})}
        self.enc_optim = Adam([p for p in self.encoder.parameters() if p.requires_grad], **eo)
        self.dec_optim = Adam([p for p in self.decoder.parameters() if p.requires_grad], **do)
        # the two views are independent chains (coupled only through in-order VQ / BN buffer updates, which the ops
        # order with events): running them on two streams lets HBM-bound kernels of one overlap MFMA-bound kernels of
        # the other.  VQW_CONCURRENT_VIEWS=0/1 overrides the default.
        if concurrent_views is None:
            import os
            concurrent_views = os.environ.get("VQW_CONCURRENT_VIEWS", "1") != "0"
        self.concurrent_views = bool(concurrent_views) and self.device.type == "cuda"
        self._s2 = None
        self.throttle = StepThrottle(self.device)
        self.reducer = None
        self._params = list(self.encoder.parameters()) + list(self.decoder.parameters())
        if data_parallel:
            params = [p for p in self.decoder.parameters() if p.requires_grad][::-1] + \
                     [p for p in self.encoder.parameters() if p.requires_grad][::-1]
            self.reducer = GradientAllReducer(params)      # buckets in backward order: decoder tail first

    def _forward_losses_two_streams(self, image, noise):
        """Same arithmetic as forward_losses; view 1 on the current stream, view 2 on a second stream."""
        w = self.w
        s1 = torch.cuda.current_stream()
        if self._s2 is None:
            import os
            self._s2 = torch.cuda.Stream(device=self.device, priority=int(os.environ.get("VQW_S2_PRIORITY", "-1")))
        s2 = self._s2
        (noised_1, clear_1), (noised_2, clear_2) = self.views(image, noise)
        s2.wait_event(s1.record_event())
        for t in (noised_2, clear_2):
            t.record_stream(s2)
        with torch.cuda.stream(s2):
            feat_2 = self.encoder.feature_extraction(noised_2)
        embed_1, l_commit_1, ids_1 = self.encoder(noised_1)
        r_ids_1 = self.views.cross_ids(ids_1, 1)
        ev1 = s1.record_event()
        with torch.cuda.stream(s2):
            embed_2, l_commit_2, ids_2 = self.encoder.vq(feat_2, id_base=1)       # ordered after view 1's update
            ids_2 = torch.transpose(ids_2, 1, 2)
            r_ids_2 = self.views.cross_ids(ids_2, 2)
            # The embedding loss runs on view 2's stream.  Autograd replays a node on the stream of its forward and in
            # reverse creation order: on stream 1 the loss's (tiny) backward would queue behind ALL of view 1's decoder
            # backward, and view 2's encoder backward, which needs its gradient, would start only then (stream 2 idle
            # for the last fifth of the step).  On stream 2 it follows view 2's decoder backward directly.
            s2.wait_event(ev1)
            for t in (embed_1, r_ids_1):
                t.record_stream(s2)
            codebook = self.encoder.vq.get_codebook()
            l_cross, l_dist, l_reg = self.embed_loss.forward_labels(embed_1, r_ids_1, embed_2, r_ids_2, codebook)
        recon_1 = self.decoder(embed_1)
        rec_1 = self._recon_terms(recon_1, clear_1)
        with torch.cuda.stream(s2):
            recon_2 = self.decoder(embed_2)
            rec_2 = self._recon_terms(recon_2, clear_2)
            ev2 = s2.record_event()
        s1.wait_event(ev2)
        for t in [l_commit_2, recon_2, l_cross, embed_2, r_ids_2, ids_2] + [t for t, _ in rec_2] + \
                [t for t in (l_dist, l_reg) if torch.is_tensor(t)]:
            t.record_stream(s1)
        l_rec_1, l_rec_2 = rec_1[0][0], rec_2[0][0]
        l_total = ops.weighted_sum(
            [l_commit_1, l_commit_2, l_cross, l_dist, l_reg] + [t for t, _ in rec_1 + rec_2],
            [w.commit, w.commit, w.cross, w.dist, w.reg] + [c for _, c in rec_1 + rec_2])
        return dict(total=l_total, commit_1=l_commit_1, commit_2=l_commit_2, cross=l_cross, dist=l_dist, reg=l_reg,
                    recon_l1=l_rec_1, recon_l2=l_rec_2, ids_1=ids_1, ids_2=ids_2, recon_1=recon_1, recon_2=recon_2,
                    embed_1=embed_1, embed_2=embed_2)

    def _recon_terms(self, recon, clear):
        """[(loss term, weight)] of one view's reconstruction loss: plain MSE, or the multi-window mean of
        recon_weights[i] * MSE on the full / lung / mediastinal windows (multi_window_trainer.py:93-118)."""
        if not self.use_recon_loss:          # l_recon = 0.0 upstream: the term is still reported, with weight zero
            return [(ops.mse_loss(recon.detach(), clear), 0.0)]
        if self.multi_window is None:
            return [(ops.mse_loss(recon, clear), self.w.recon)]
        dw, rw = self.multi_window["dataset_window"], self.multi_window["recon_weights"]
        terms = [ops.mse_loss(recon, clear), ops.window_mse_loss(recon, clear, dw, LUNG_WINDOW),
                 ops.window_mse_loss(recon, clear, dw, MEDIASTINAL_WINDOW)]
        return [(t, self.w.recon * float(r) / 3.0) for t, r in zip(terms, rw)]

    def forward_losses(self, image, noise=None):
        """Lines 73-137 of the reference step.  `image` is in [-1, 1] (dataloader convention)."""
        if self.concurrent_views and not self.use_onehot:
            return self._forward_losses_two_streams(image, noise)
        w = self.w
        (noised_1, clear_1), (noised_2, clear_2) = self.views(image, noise)
        embed_1, l_commit_1, ids_1 = self.encoder(noised_1)
        embed_2, l_commit_2, ids_2 = self.encoder(noised_2)
        r_ids_1 = self.views.cross_ids(ids_1, 1)
        r_ids_2 = self.views.cross_ids(ids_2, 2)
        codebook = self.encoder.vq.get_codebook()
        if self.use_onehot:      # the reference's literal route: (B,K+1,H,W) one-hot, class 0 dropped
            oh1 = self.one_hot_encoder(r_ids_1)[:, 1:, ...]
            oh2 = self.one_hot_encoder(r_ids_2)[:, 1:, ...]
            l_cross, l_dist, l_reg = self.embed_loss(embed_1, oh1, embed_2, oh2, codebook)
        else:
            l_cross, l_dist, l_reg = self.embed_loss.forward_labels(embed_1, r_ids_1, embed_2, r_ids_2, codebook)
        recon_1 = self.decoder(embed_1)
        recon_2 = self.decoder(embed_2)
        rec_1, rec_2 = self._recon_terms(recon_1, clear_1), self._recon_terms(recon_2, clear_2)
        l_rec_1, l_rec_2 = rec_1[0][0], rec_2[0][0]
        l_total = ops.weighted_sum(
            [l_commit_1, l_commit_2, l_cross, l_dist, l_reg] + [t for t, _ in rec_1 + rec_2],
            [w.commit, w.commit, w.cross, w.dist, w.reg] + [c for _, c in rec_1 + rec_2])
        return dict(total=l_total, commit_1=l_commit_1, commit_2=l_commit_2, cross=l_cross, dist=l_dist, reg=l_reg,
                    recon_l1=l_rec_1, recon_l2=l_rec_2, ids_1=ids_1, ids_2=ids_2, recon_1=recon_1, recon_2=recon_2,
                    embed_1=embed_1, embed_2=embed_2)

    def training_step(self, batch, noise=None):
        image = batch['image'] if isinstance(batch, dict) else batch
        self.throttle.begin()
        ops.begin_step()
        if self.reducer is not None:
            ops.reset_pending(self._params)
        out = self.forward_losses(image, noise)
        self.enc_optim.zero_grad()
        self.dec_optim.zero_grad()
        if self.reducer is not None:
            self.reducer.prepare()
        out["total"].backward()
        if self._s2 is not None:
            torch.cuda.current_stream().wait_stream(self._s2)
        ops.join_streams()
        if self.reducer is not None:
            self.reducer.finish()
        self.enc_optim.step()
        self.dec_optim.step()
        self.throttle.end()
        return out

    @staticmethod
    def scalars(out):
        """Host copies of the logged scalars (one sync; keep out of timed regions)."""
        f = lambda t: float(t.detach()) if torch.is_tensor(t) else float(t)  # noqa: E731
        return dict(total=f(out["total"]), commit=f(out["commit_1"]) + f(out["commit_2"]), cross=f(out["cross"]),
                    dist=f(out["dist"]), reg=f(out["reg"]), recon=f(out["recon_l1"]) + f(out["recon_l2"]))
