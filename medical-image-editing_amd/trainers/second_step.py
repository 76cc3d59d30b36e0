"""Second training step (generator + PatchGAN discriminator) of the VQ-W-Net: reference
trainers/single_window_trainer.py:434-488 (`_train_second_step_nl_dis`), optimisers per trainers/base.py:165-181.

The encoder is frozen (eval mode, no_grad); the decoder is trained on  w.recon * MSE(recon, image) + w.gen * (-mean(D(recon)));
then the discriminator on  w.dis * hinge_d_loss(D(image), D(recon.detach()))  for n_inner_loops.  Frequency / perceptual
terms (LPIPS / VGG weights, FFT loss) are not part of this build, as in the first step.
"""
from collections import namedtuple

import torch

from hipops import ops, Adam
from networks.discriminator import NLayerDiscriminator
from functions.gan_loss import hinge_d_loss, generator_loss

GanLossWeights = namedtuple("GanLossWeights", "recon gen dis", defaults=(1.0, 1.0, 1.0))


class SecondStepTrainer:
    def __init__(self, encoder, decoder, dis=None, loss_weight=None, n_inner_loops=1, lr=1e-4, betas=(0.5, 0.999),
                 weight_decay=0.0, device="cuda", data_parallel=False):
        self.device = torch.device(device)
        from .first_step import StepThrottle
        self.throttle = StepThrottle(self.device)      # at most two steps enqueued ahead of the GPU
        self.encoder = encoder.to(self.device)
        self.decoder = decoder.to(self.device).train()
        self.dis = (dis if dis is not None else NLayerDiscriminator()).to(self.device).train()
        self.w = loss_weight if loss_weight is not None else GanLossWeights()
        self.n_inner_loops = int(n_inner_loops)
        self.dec_optim = Adam([p for p in self.decoder.parameters() if p.requires_grad], lr=lr, betas=betas,
                              weight_decay=weight_decay)
        self.dis_optim = Adam([p for p in self.dis.parameters() if p.requires_grad], lr=lr, betas=betas,
                              weight_decay=weight_decay)
        # one process per GPU (run_vqwnet.py:112-121): bucketed gradient all-reduce per optimiser, overlapped with
        # the rest of its backward pass; BatchNorm statistics (StyledDenorm and the discriminator's) are synchronised
        # inside their kernels' host code when a process group is up
        self.dec_reducer = self.dis_reducer = None
        if data_parallel:
            from .data_parallel import GradientAllReducer
            self.dec_reducer = GradientAllReducer(list(reversed([p for p in self.decoder.parameters() if p.requires_grad])))
            self.dis_reducer = GradientAllReducer(list(reversed([p for p in self.dis.parameters() if p.requires_grad])))

    def training_step(self, batch):
        image = batch['image'] if isinstance(batch, dict) else batch
        w = self.w
        self.throttle.begin()
        ops.begin_step()
        if self.dec_reducer is not None:
            ops.reset_pending(self.dec_optim.param_groups[0]["params"])
        self.encoder.eval()
        with torch.no_grad():
            embed, _, ids = self.encoder(image)
        recon = self.decoder(embed.detach())
        l_recon = ops.mse_loss(recon, image)
        # The reference lets autograd fill the discriminator's parameter gradients in this pass and discards them
        # (dis_optim.zero_grad() below); they are not computed here.  Same decoder gradients, same update.
        dis_params = [p for p in self.dis.parameters() if p.requires_grad]
        for p in dis_params:
            p.requires_grad_(False)
        try:
            l_gen = generator_loss(self.dis(recon))
            l_gen_total = ops.weighted_sum([l_recon, l_gen], [w.recon, w.gen])
            self.dec_optim.zero_grad()
            if self.dec_reducer is not None:
                self.dec_reducer.prepare()
            l_gen_total.backward()
        finally:
            for p in dis_params:
                p.requires_grad_(True)
        ops.join_streams()
        if self.dec_reducer is not None:
            self.dec_reducer.finish()
        self.dec_optim.step()
        l_dis_total = None
        for _ in range(self.n_inner_loops):
            l_real = self.dis(image.detach())
            l_fake = self.dis(recon.detach())
            l_dis = hinge_d_loss(l_real, l_fake)
            l_dis_total = ops.weighted_sum([l_dis], [w.dis])
            self.dis_optim.zero_grad()
            if self.dis_reducer is not None:
                self.dis_reducer.prepare()
            l_dis_total.backward()
            if self.dis_reducer is not None:
                self.dis_reducer.finish()
            self.dis_optim.step()
        self.throttle.end()
        return dict(gen_total=l_gen_total, recon=l_recon, gen=l_gen, dis_total=l_dis_total, ids=ids, recon_image=recon)
