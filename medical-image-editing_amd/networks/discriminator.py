"""PatchGAN discriminator of the second training step (reference: networks/discriminator.py:18-87) on the HIP kernels.

Same constructor, `self.main` Sequential layout and state_dict keys (`main.0.weight`, `main.3.running_mean`, ...) and the
reference's `weights_init` (:9-15).  Convolutions are 4x4, padding 1, stride 2 (last two: stride 1); BatchNorm2d +
LeakyReLU(0.2) run as one kernel, the first layer's LeakyReLU in the conv epilogue.
"""
import torch
import torch.nn as nn

from hipops import ops


class SConv2d(nn.Conv2d):
    """nn.Conv2d holder (weight kept channels_last = OHWI) evaluated by the strided direct-conv kernels."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=bias)
        self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)

    def forward(self, x, slope=1.0):
        return ops.sconv2d(x, self.weight, self.bias, self.stride[0], self.padding[0], slope)


class FusedLeakyReLU(nn.Identity):
    """Placeholder keeping the reference's Sequential indices: the activation is fused into the previous kernel."""

    def __init__(self, negative_slope=0.2, inplace=True):
        super().__init__()
        self.negative_slope = negative_slope


def weights_init(m):
    classname = m.__class__.__name__
    if classname.find('Conv') != -1:
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif classname.find('BatchNorm') != -1:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0)


class NLayerDiscriminator(nn.Module):
    def __init__(self, in_channels=1, out_channels=1, n_filters=64, n_layers=3, normalization='batchnorm'):
        super().__init__()
        assert normalization in {'instancenorm', 'batchnorm', 'actnorm'}
        if normalization != 'batchnorm':
            raise NotImplementedError("only the default normalization='batchnorm' is built")
        use_bias = False                      # BatchNorm2d has affine parameters (reference :50-53)
        kw, padw = 4, 1
        sequence = [SConv2d(in_channels, n_filters, kw, stride=2, padding=padw), FusedLeakyReLU(0.2)]
        nf_mult = 1
        for n in range(1, n_layers):
            nf_mult_prev, nf_mult = nf_mult, min(2 ** n, 8)
            sequence += [SConv2d(n_filters * nf_mult_prev, n_filters * nf_mult, kw, stride=2, padding=padw, bias=use_bias),
                         nn.BatchNorm2d(n_filters * nf_mult), FusedLeakyReLU(0.2)]
        nf_mult_prev, nf_mult = nf_mult, min(2 ** n_layers, 8)
        sequence += [SConv2d(n_filters * nf_mult_prev, n_filters * nf_mult, kw, stride=1, padding=padw, bias=use_bias),
                     nn.BatchNorm2d(n_filters * nf_mult), FusedLeakyReLU(0.2)]
        sequence += [SConv2d(n_filters * nf_mult, out_channels, kw, stride=1, padding=padw)]
        self.main = nn.Sequential(*sequence)
        self.apply(weights_init)
        for m in self.modules():              # init rewrote the conv weights: keep them channels_last
            if isinstance(m, SConv2d):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)

    def forward(self, input):
        x = input
        layers = list(self.main)
        i = 0
        while i < len(layers):
            m = layers[i]
            if isinstance(m, SConv2d):
                nxt = layers[i + 1] if i + 1 < len(layers) else None
                if isinstance(nxt, FusedLeakyReLU):
                    x = m(x, slope=nxt.negative_slope)
                    i += 2
                    continue
                x = m(x)
            elif isinstance(m, nn.BatchNorm2d):
                nxt = layers[i + 1] if i + 1 < len(layers) else None
                slope = nxt.negative_slope if isinstance(nxt, FusedLeakyReLU) else 1.0
                x = ops.batch_norm_lrelu(x, m.weight, m.bias, m.running_mean, m.running_var, self.training,
                                         momentum=m.momentum, eps=m.eps, slope=slope,
                                         num_batches_tracked=m.num_batches_tracked if self.training else None)
                if slope != 1.0:
                    i += 2
                    continue
            i += 1
        return x
