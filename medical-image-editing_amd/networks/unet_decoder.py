"""SPADE-style U-Net decoder (reference: networks/unet_decoder.py:19-164) on the HIP kernels."""
import contextlib

import torch
import torch.nn as nn

from hipops import ops
from .blocks import DoubleConv, StyledResUpBlock, Conv2d
from ._unet import add_half, level_modules
from .dropblock import LinearScheduler, DropBlock2D
from .initialize import init_weights
from .aspp import ASPP


class UNetDecoder(nn.Module):
    """Constructor arguments as unet_decoder.py:21-34: in_channels, out_channels, filters, use_dropblock, block_size,
    start_value, stop_value, nr_steps, dropped_skip_layers, use_styled_up_block, use_pixel_shuffle,
    use_last_pixel_shuffle."""

    def __init__(self, in_channels, out_channels, filters=[64, 128, 256, 512, 1024], use_dropblock=False, block_size=30,
                 start_value=0.3, stop_value=0.9, nr_steps=100, dropped_skip_layers=[5, 6], use_styled_up_block=True,
                 use_pixel_shuffle=True, use_last_pixel_shuffle=False):
        super().__init__()
        assert use_styled_up_block
        if use_last_pixel_shuffle:
            raise NotImplementedError("use_last_pixel_shuffle heads are not built (off by default upstream)")
        self.use_last_pixel_shuffle = use_last_pixel_shuffle
        self.dropped_skip_layers = dropped_skip_layers
        self.dropblock = (LinearScheduler(DropBlock2D(block_size=block_size, drop_prob=start_value), start_value=start_value,
                                          stop_value=stop_value, nr_steps=nr_steps) if use_dropblock else (lambda t: t))
        f = list(filters)

        def up_block(k):        # level k: f[k] channels coming up, f[k-1] skip channels as the SPADE style input
            return StyledResUpBlock(f[k], f[k - 1], f[k - 1], use_pixel_shuffle=use_pixel_shuffle)
        n = add_half(self, 2, in_channels, f, up_block)
        self.down_convs, _, self.up_convs = level_modules(self, 2, n)      # plain lists, as upstream keeps them
        init_weights(self, 'kaiming')      # upstream initialises here: the head below keeps the default initialisation
        self.conv_last = nn.Sequential(ASPP(f[0], f[0], [2, 6, 12, 18]), DoubleConv(5 * f[0], f[0]))
        self.conv1x1 = Conv2d(f[0], out_channels, kernel_size=1)
        self.final_act = nn.Tanh()

    @property
    def name(self):
        return 'UNetDecoder'

    def forward(self, x):
        d_skips = []
        for d in self.down_convs:
            x, d_skip = d(x)
            d_skips.append(d_skip)
        d_skips.reverse()
        for i in range(len(d_skips)):
            if i in self.dropped_skip_layers:
                d_skips[i] = torch.zeros_like(d_skips[i])
            else:
                d_skips[i] = self.dropblock(d_skips[i])
        # the SPADE modulation maps need only the skips: queue them on the branch stream, deepest level first, so
        # they run beside the bottleneck and the up-path trunk (same arithmetic as evaluating them inside each block)
        # Past the last max-pool nothing downstream decides anything by comparing activations of different pixels, so the
        # layers from here on may take the Winograd forward in training too (ops.winograd_forward; the down path above
        # keeps the direct form: its pools sit on exact ties of the piecewise constant input)
        scope = ops.winograd_forward() if ops.WINOGRAD_FWD_POOLFREE else contextlib.nullcontext()
        with scope:
            maps = [u.style_maps(d_skip) for u, d_skip in zip(self.up_convs, d_skips)]
            x = self.double_conv2(x)
            for u, d_skip, m in zip(self.up_convs, d_skips, maps):
                x = u(x, d_skip, maps=m)
            # x feeds the residual and the pyramid's five convolutions: ONE gradient group - the residual's gradient is the
            # buffer the branches' input-gradient kernels add to (no add pass by autograd: ops.add(..., a_group=))
            aspp, dc = self.conv_last[0], self.conv_last[1]
            grp = ops.GradGroup(len(list(aspp.stages.children())) + 1) if (ops.GRAD_GROUPS and torch.is_grad_enabled() and x.requires_grad) else None
            y = aspp(x, grad_group=grp)
            if dc.ends_in_norm_relu() and ops.add_norm_supported(x, x) and x.shape[1] == dc.double_conv[3].weight.shape[0]:
                # the DoubleConv's last InstanceNorm + ReLU is applied inside the residual add (its tensor is never written)
                raw, part = dc(y, raw_tail=True)
                out = ops.add_norm(x, raw, part, relu=True, eps=dc.double_conv[4].eps, a_group=grp)
            else:
                out = ops.add(x, dc(y), a_group=grp)
            out = self.conv1x1(out)
            return ops.tanh(out)
