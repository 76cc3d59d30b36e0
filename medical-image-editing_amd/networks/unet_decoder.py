"""SPADE-style U-Net decoder (reference: networks/unet_decoder.py:19-164) on the HIP kernels."""
import torch
import torch.nn as nn

from hipops import ops
from .blocks import ResBlock, DoubleConv, StyledResUpBlock, Conv2d
from .dropblock import LinearScheduler, DropBlock2D
from .initialize import init_weights
from .aspp import ASPP


class UNetDecoder(nn.Module):

    def __init__(self,
                 in_channels: int,
                 out_channels: int,
                 filters: list = [64, 128, 256, 512, 1024],
                 use_dropblock: bool = False,
                 block_size: int = 30,
                 start_value: float = 0.3,
                 stop_value: float = 0.9,
                 nr_steps: int = 100,
                 dropped_skip_layers: list = [5, 6],
                 use_styled_up_block: bool = True,
                 use_pixel_shuffle: bool = True,
                 use_last_pixel_shuffle: bool = False,
                 ):
        super().__init__()
        assert use_styled_up_block
        if use_last_pixel_shuffle:
            raise NotImplementedError("use_last_pixel_shuffle heads are not built (off by default upstream)")
        self.use_last_pixel_shuffle = use_last_pixel_shuffle
        self.dropped_skip_layers = dropped_skip_layers

        if use_dropblock:
            self.dropblock = LinearScheduler(
                DropBlock2D(block_size=block_size, drop_prob=start_value),
                start_value=start_value, stop_value=stop_value, nr_steps=nr_steps)
        else:
            self.dropblock = lambda x: x

        n = len(filters) - 1
        self.down_convs = []
        for i in range(n):
            block = ResBlock(in_channels if i == 0 else filters[i - 1], filters[i])
            self.add_module('down_conv2_{}'.format(i + 1), block)
            self.down_convs.append(block)

        self.double_conv2 = DoubleConv(filters[n - 1], filters[n])

        self.up_convs = []
        for i in reversed(range(n)):
            block = StyledResUpBlock(filters[i + 1], filters[i], filters[i], use_pixel_shuffle=use_pixel_shuffle)
            self.add_module('up_conv2_{}'.format(i + 1), block)
            self.up_convs.append(block)

        init_weights(self, 'kaiming')

        self.conv_last = nn.Sequential(
            ASPP(filters[0], filters[0], [2, 6, 12, 18]),
            DoubleConv(5 * filters[0], filters[0]),
        )
        self.conv1x1 = Conv2d(filters[0], out_channels, kernel_size=1)
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
        maps = [u.style_maps(d_skip) for u, d_skip in zip(self.up_convs, d_skips)]
        x = self.double_conv2(x)
        for u, d_skip, m in zip(self.up_convs, d_skips, maps):
            x = u(x, d_skip, maps=m)
        out = ops.add(x, self.conv_last(x))
        out = self.conv1x1(out)
        return ops.tanh(out)
