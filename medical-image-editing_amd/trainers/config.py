"""Config-driven construction: the reference's config namedtuple (utils.load_json: JSON `false` -> None) to modules,
optimisers, loss and the first-step trainer - TrainerBase.configure_models / configure_optimizers / configure_losses /
set_transform (trainers/base.py:189-237, 164-183, 261-282) with the reference's key names.

    config.run.{num_gpus, training_mode, seed}
    config.model.vqmodel.{in_channels, enc_filters, dec_filters, dict_size, momentum, knn_backend, enc_use_styled_up_block,
                          dec_use_styled_up_block, use_init_embed, use_dropblock, block_size, start_value, stop_value,
                          nr_steps, dropped_skip_layers, use_pixel_shuffle}
    config.loss.{loss_weight.{commit, cross, dist, reg, recon, freq, perceptual}, embed_loss.{margin, use_distance_loss,
                 use_regularization_loss}, use_recon_loss, use_frequency_loss, use_perceptual_loss}
    config.{enc_optim, dec_optim}.{lr, b1, b2, weight_decay}
    config.augmentation.{modules, ...}            (optional: absent -> the exact-integer flip views of the benchmark)
    config.dataset.{window_width, window_center, window_scale}, config.loss.recon_weights   (multi-window runs, -w)

Frequency (focal-frequency-loss) and perceptual (VGG / LPIPS weights that must be fetched) losses are third-party and
absent offline: a config that switches them on raises instead of silently training something else.
"""
from functions import EmbeddingLoss
from hipops import Adam
from networks import UNetEncoder, UNetDecoder, RandomTransform

from .first_step import FirstStepTrainer, FlipViews, RandomTransformViews, LossWeights


def _get(cfg, name, default=None):
    return getattr(cfg, name) if hasattr(cfg, name) else default


def configure_models(config):
    """-> (UNetEncoder, UNetDecoder) exactly as base.py:189-237 calls the constructors (`init_embed = not use_init_embed`)."""
    g = config.model.vqmodel
    if _get(g, "model_name") == "VQGAN":
        raise NotImplementedError("model_name 'VQGAN' (transformer-stage decoder) is outside the hot path this build covers")
    encoder = UNetEncoder(
        in_channels=g.in_channels,
        filters=list(g.enc_filters),
        dict_size=g.dict_size,
        momentum=g.momentum,
        knn_backend=g.knn_backend,
        use_styled_up_block=bool(g.enc_use_styled_up_block),
        num_gpus=config.run.num_gpus,
        init_embed=not g.use_init_embed,
    )
    decoder = UNetDecoder(
        in_channels=g.enc_filters[0],
        out_channels=g.in_channels,
        filters=list(g.dec_filters),
        use_dropblock=bool(g.use_dropblock),
        block_size=g.block_size,
        start_value=g.start_value,
        stop_value=g.stop_value,
        nr_steps=g.nr_steps,
        dropped_skip_layers=list(g.dropped_skip_layers or []),
        use_styled_up_block=bool(g.dec_use_styled_up_block),
        use_pixel_shuffle=bool(g.use_pixel_shuffle),
    )
    return encoder, decoder


def _adam_kwargs(o):
    return dict(lr=o.lr, betas=(o.b1, o.b2), weight_decay=o.weight_decay or 0.0)


def configure_optimizers(config, encoder, decoder):
    """-> (enc_optim, dec_optim): Adam over each sub-network's trainable parameters (base.py:164-175)."""
    return (Adam([p for p in encoder.parameters() if p.requires_grad], **_adam_kwargs(config.enc_optim)),
            Adam([p for p in decoder.parameters() if p.requires_grad], **_adam_kwargs(config.dec_optim)))


def configure_losses(config):
    """-> EmbeddingLoss (base.py:261-278); None-for-false flags pass through as the reference's do."""
    c = config.loss
    if _get(c, "use_perceptual_loss"):
        raise NotImplementedError("use_perceptual_loss needs pretrained VGG / LPIPS weights that cannot be fetched offline")
    if _get(c, "use_frequency_loss"):
        raise NotImplementedError("use_frequency_loss needs the third-party focal-frequency-loss package (absent offline)")
    return EmbeddingLoss(dict_size=config.model.vqmodel.dict_size, margin=c.embed_loss.margin,
                         use_distance_loss=c.embed_loss.use_distance_loss,
                         use_regularization_loss=c.embed_loss.use_regularization_loss)


def loss_weights(config):
    w = config.loss.loss_weight
    return LossWeights(**{k: float(_get(w, k) or 0.0) for k in LossWeights._fields})


def set_transform(config, seed=0):
    """The pair of RandomTransform modules (base.py:280-282) behind the trainer's `views` protocol; without an
    `augmentation` section: identity / horizontal flip (the benchmark's exact-integer views)."""
    aug = _get(config, "augmentation")
    if aug is None:
        return FlipViews()
    return RandomTransformViews(RandomTransform(aug, seed=seed), RandomTransform(aug, seed=seed + 1))


def build_first_step_trainer(config, device="cuda", data_parallel=None, views=None, multi_window=None):
    """config -> FirstStepTrainer (the `first_step` training mode of run_vqwnet.py).  data_parallel defaults to
    torch.distributed being initialised with more than one rank."""
    import torch.distributed as dist
    mode = _get(config.run, "training_mode", "first_step")
    if mode != "first_step":
        raise NotImplementedError("training_mode %r: use trainers.SecondStepTrainer for the GAN step" % mode)
    encoder, decoder = configure_models(config)
    if data_parallel is None:
        data_parallel = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if multi_window is None and _get(config.loss, "recon_weights") is not None and _get(config.dataset, "window_width") is not None:
        d = config.dataset
        multi_window = dict(dataset_window=(d.window_width, d.window_center, d.window_scale), recon_weights=tuple(config.loss.recon_weights))
    g = config.model.vqmodel
    return FirstStepTrainer(
        in_channels=g.in_channels, enc_filters=tuple(g.enc_filters), dec_filters=tuple(g.dec_filters), dict_size=g.dict_size,
        momentum=g.momentum, margin=config.loss.embed_loss.margin, loss_weight=loss_weights(config),
        views=views if views is not None else set_transform(config, seed=_get(config.run, "seed", 0) or 0), device=device,
        encoder=encoder, decoder=decoder, data_parallel=data_parallel, multi_window=multi_window,
        embed_loss=configure_losses(config), enc_optim=_adam_kwargs(config.enc_optim), dec_optim=_adam_kwargs(config.dec_optim),
        use_recon_loss=bool(_get(config.loss, "use_recon_loss", True)))
