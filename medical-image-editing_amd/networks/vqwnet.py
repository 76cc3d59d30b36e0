"""Monolithic VQ-W-Net: two U-Nets in series with the quantiser in between (reference: networks/vqwnet.py:13-176).
Not used by the trainers upstream; same blocks and kernels as UNetEncoder, plain UpBlock decoder half."""
import torch
import torch.nn as nn

from hipops import ops
from .vq import VQ
from .blocks import UpBlock, Conv2d
from ._unet import add_half, run_half, level_modules
from .dropblock import LinearScheduler, DropBlock2D
from .initialize import init_weights


class VQWNet(nn.Module):
    """Constructor arguments as vqwnet.py:15-26 (in_channels, out_channels, filters, dict_size, knn_backend,
    use_dropblock, block_size, drop_prob, nr_steps, freeze_first_half)."""

    def __init__(self, in_channels, out_channels, filters=[64, 128, 256, 512, 1024], dict_size=512, knn_backend='torch',
                 use_dropblock=False, block_size=30, drop_prob=0.3, nr_steps=100, freeze_first_half=False):
        super().__init__()
        assert in_channels == out_channels
        self.freeze_first_half = freeze_first_half
        f = list(filters)

        def up_block(k):
            return UpBlock(f[k - 1] + f[k], f[k - 1])
        self._levels = add_half(self, 1, in_channels, f, up_block)
        self.vq = VQ(emb_dim=f[0], dict_size=dict_size, momentum=0.99, eps=1e-5, knn_backend=knn_backend)
        self.dropblock = (LinearScheduler(DropBlock2D(block_size=block_size, drop_prob=0.), start_value=0.,
                                          stop_value=drop_prob, nr_steps=nr_steps) if use_dropblock else (lambda t: t))
        add_half(self, 2, f[0], f, up_block)
        self.conv_last = Conv2d(f[0], out_channels, kernel_size=1)
        self.final_act = nn.Tanh()
        init_weights(self, 'kaiming')
        if freeze_first_half:
            # upstream only sets a plain attribute on the sub-modules (a no-op for autograd, vqwnet.py:84-94); the
            # actual freezing is the no_grad / detach in forward.  Kept as is.
            downs, mid, ups = level_modules(self, 1, self._levels)
            for m in downs + [mid] + ups + [self.vq]:
                m.requires_grad = False

    @property
    def name(self):
        return 'VQWNet'

    def _first_half(self, x):
        return run_half(self, 1, self._levels, x)

    def _second_half(self, x):
        return ops.tanh(self.conv_last(run_half(self, 2, self._levels, x)))

    def forward(self, x):
        if not self.freeze_first_half:
            embed = self._first_half(x)
            x, commit_loss, ids = self.vq(embed, id_base=1)
            ids = torch.transpose(ids, 1, 2)
        else:
            with torch.no_grad():
                x, commit_loss, ids = self.vq(self._first_half(x), id_base=1)
                ids = torch.transpose(ids, 1, 2)
                x = x.detach()
                embed = x
        x = self.dropblock(x)
        return {'recon': self._second_half(x), 'embed': embed, 'commit_loss': commit_loss, 'ids': ids}

    def generate_images_from_ids(self, ids):
        # upstream transposes ids, looks them up as (B,W,H,D) and swaps dims 1<->3 back; the transposes cancel
        with torch.no_grad():
            x = ops.vq_lookup(ids, self.vq.embed)
            out = self._second_half(x)
        return {'recon': out, 'ids': torch.transpose(ids, 1, 2)}
