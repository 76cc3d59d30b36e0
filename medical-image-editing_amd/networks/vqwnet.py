"""Monolithic VQ-W-Net: two U-Nets in series with the quantiser in between (reference: networks/vqwnet.py:13-176).
Not used by the trainers upstream; same blocks and kernels as UNetEncoder, plain UpBlock decoder half."""
import torch
import torch.nn as nn

from hipops import ops
from .vq import VQ
from .blocks import ResBlock, UpBlock, DoubleConv, Conv2d
from .dropblock import LinearScheduler, DropBlock2D
from .initialize import init_weights


class VQWNet(nn.Module):

    def __init__(self,
                 in_channels: int,
                 out_channels: int,
                 filters: list = [64, 128, 256, 512, 1024],
                 dict_size: int = 512,
                 knn_backend: str = 'torch',
                 use_dropblock: bool = False,
                 block_size: int = 30,
                 drop_prob: float = 0.3,
                 nr_steps: int = 100,
                 freeze_first_half: bool = False,
                 ):
        super().__init__()
        assert in_channels == out_channels
        self.freeze_first_half = freeze_first_half
        f = filters
        self.down_conv1_1 = ResBlock(in_channels, f[0])
        self.down_conv1_2 = ResBlock(f[0], f[1])
        self.down_conv1_3 = ResBlock(f[1], f[2])
        self.down_conv1_4 = ResBlock(f[2], f[3])
        self.double_conv1 = DoubleConv(f[3], f[4])
        self.up_conv1_4 = UpBlock(f[3] + f[4], f[3])
        self.up_conv1_3 = UpBlock(f[2] + f[3], f[2])
        self.up_conv1_2 = UpBlock(f[1] + f[2], f[1])
        self.up_conv1_1 = UpBlock(f[1] + f[0], f[0])
        self.vq = VQ(emb_dim=f[0], dict_size=dict_size, momentum=0.99, eps=1e-5, knn_backend=knn_backend)
        if use_dropblock:
            self.dropblock = LinearScheduler(DropBlock2D(block_size=block_size, drop_prob=0.),
                                             start_value=0., stop_value=drop_prob, nr_steps=nr_steps)
        else:
            self.dropblock = lambda x: x
        self.down_conv2_1 = ResBlock(f[0], f[0])
        self.down_conv2_2 = ResBlock(f[0], f[1])
        self.down_conv2_3 = ResBlock(f[1], f[2])
        self.down_conv2_4 = ResBlock(f[2], f[3])
        self.double_conv2 = DoubleConv(f[3], f[4])
        self.up_conv2_4 = UpBlock(f[3] + f[4], f[3])
        self.up_conv2_3 = UpBlock(f[2] + f[3], f[2])
        self.up_conv2_2 = UpBlock(f[1] + f[2], f[1])
        self.up_conv2_1 = UpBlock(f[1] + f[0], f[0])
        self.conv_last = Conv2d(f[0], out_channels, kernel_size=1)
        self.final_act = nn.Tanh()
        init_weights(self, 'kaiming')
        if self.freeze_first_half:
            self._freeze_first_half()

    @property
    def name(self):
        return 'VQWNet'

    def _freeze_first_half(self):
        # upstream only sets a plain attribute on the sub-modules (a no-op for autograd, vqwnet.py:84-94); the
        # actual freezing is the no_grad / detach in forward.  Kept as is.
        for m in (self.down_conv1_1, self.down_conv1_2, self.down_conv1_3, self.down_conv1_4, self.double_conv1,
                  self.up_conv1_4, self.up_conv1_3, self.up_conv1_2, self.up_conv1_1, self.vq):
            m.requires_grad = False

    def _first_half(self, x):
        x, s1 = self.down_conv1_1(x)
        x, s2 = self.down_conv1_2(x)
        x, s3 = self.down_conv1_3(x)
        x, s4 = self.down_conv1_4(x)
        x = self.double_conv1(x)
        x = self.up_conv1_4(x, s4)
        x = self.up_conv1_3(x, s3)
        x = self.up_conv1_2(x, s2)
        return self.up_conv1_1(x, s1)

    def _second_half(self, x):
        x, s1 = self.down_conv2_1(x)
        x, s2 = self.down_conv2_2(x)
        x, s3 = self.down_conv2_3(x)
        x, s4 = self.down_conv2_4(x)
        x = self.double_conv2(x)
        x = self.up_conv2_4(x, s4)
        x = self.up_conv2_3(x, s3)
        x = self.up_conv2_2(x, s2)
        x = self.up_conv2_1(x, s1)
        return ops.tanh(self.conv_last(x))

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
