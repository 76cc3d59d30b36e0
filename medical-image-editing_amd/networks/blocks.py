"""Building blocks of the VQ-W-Net on the MI355X HIP kernels.

Same classes, constructor signatures, attribute names and state_dict keys as the
reference's networks/blocks.py (UpBlock :9-18, ResBlock :21-36, DoubleConv :39-61,
StyledDenorm :64-90, StyledResUpBlock :93-134); the forward passes call
hipops.ops (hand-written HIP through the C ABI) instead of ATen.  nn.Upsample and
torch.cat are folded into the consuming convolution's loader; ReLU is fused into
the InstanceNorm / SPADE / conv-epilogue kernels.
"""
import os

import torch
import torch.nn as nn

from hipops import ops


class Conv2d(nn.Conv2d):
    """nn.Conv2d parameter holder (same init, same state_dict entries) whose forward is the HIP
    implicit-GEMM convolution.  'same' padding, stride 1, kernel 1 or 3 only — all the path uses."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=True):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, bias=bias)
        k, d = self.kernel_size[0], self.dilation[0]
        if self.kernel_size[0] != self.kernel_size[1] or k not in (1, 3) or self.stride != (1, 1) \
                or self.padding != (d * (k // 2),) * 2 or self.dilation[0] != self.dilation[1]:
            raise NotImplementedError("HIP conv supports kernel 1/3, stride 1, padding = dilation*(k//2)")
        # OHWI storage (channels_last); logical shape / state_dict unchanged
        self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)

    def forward(self, x, up2x=False, skip=None, relu=False, want_stats=False, grad_group=None, norm_input=False):
        return ops.conv2d(x, self.weight, self.bias, self.dilation[0], up2x=up2x, skip=skip, relu=relu, want_stats=want_stats,
                          grad_group=grad_group, norm_input=norm_input)


def conv3x3(in_channels, out_channels, stride=1, padding=1, bias=True):
    return Conv2d(in_channels, out_channels, 3, stride, padding, bias=bias)


class InstanceNorm2d(nn.Module):
    """InstanceNorm2d(affine=False, track_running_stats=False), optionally with the following ReLU fused."""

    def __init__(self, num_features, relu=False, eps=1e-5):
        super().__init__()
        self.num_features, self.relu, self.eps = num_features, relu, eps

    def forward(self, x, part=None):
        return ops.instance_norm(x, relu=self.relu, eps=self.eps, part=part)


class FusedReLU(nn.Identity):
    """Placeholder that keeps nn.Sequential indices (and so state_dict keys) equal to the reference's;
    the ReLU itself runs inside the preceding normalisation kernel."""


class UpBlock(nn.Module):
    def __init__(self, in_channels, out_channels, use_output_act=True):
        super().__init__()
        self.up_sample = nn.Upsample(scale_factor=2, mode='nearest')   # folded into the conv loader
        self.double_conv = DoubleConv(in_channels, out_channels, use_output_act=use_output_act)

    def forward(self, down_input, skip_input):
        # channels = [up2x(down) | skip]; neither the up-sampled map nor the concat is materialised
        return self.double_conv(down_input, up2x=True, skip=skip_input)


class ResBlock(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.downsample = nn.Sequential(
            Conv2d(in_channels, out_channels, kernel_size=1, stride=1, bias=False),
            InstanceNorm2d(out_channels),
        )
        self.double_conv = DoubleConv(in_channels, out_channels)
        self.down_sample = nn.MaxPool2d(2)
        self.relu = nn.ReLU()

    def forward(self, x):
        # both branches end in an InstanceNorm: the tail kernel normalises their raw conv outputs while it reads them
        dc, ds = self.double_conv, self.downsample
        # x feeds the 1x1 branch and the first 3x3 convolution: their input gradients are summed in the second one's
        # epilogue (ops.GradGroup) instead of by an add pass
        grp = ops.GradGroup(2) if (GRAD_GROUP_BLOCKS and x.requires_grad) else None
        # (the 1 x 1 branch is created FIRST: autograd then runs it LAST in backward, where its input-gradient kernel adds to the
        # 3 x 3 convolution's gradient in its epilogue - every shape has that route, while a 16-channel 3 x 3 layer has no
        # accumulating kernel and its gradient used to be added by a separate three-pass kernel)
        if RES_TAIL_NORM and dc.ends_in_norm_relu() and ds[1].eps == dc.double_conv[4].eps and not ds[1].relu:
            xid, partid = ds[0](x, want_stats=True, grad_group=grp)
            x2, part2 = dc(x, raw_tail=True, grad_group=grp)
            got = ops.res_tail_norm(x2, xid, eps=ds[1].eps, part2=part2, partid=partid)
            if got is not None:
                return got
            return ops.res_tail(dc.double_conv[4](x2, part=part2), ds[1](xid, part=partid))      # shape not served: separate norms
        xid, partid = ds[0](x, want_stats=True, grad_group=grp)
        return ops.res_tail(dc(x, grad_group=grp), ds[1](xid, part=partid))      # (pooled, out) with a single backward kernel


class DoubleConv(nn.Module):
    def __init__(self, in_channels, out_channels, use_output_act=True):
        super().__init__()
        layers = [
            Conv2d(in_channels, out_channels, kernel_size=3, padding=1),
            InstanceNorm2d(out_channels, relu=True),
            FusedReLU(),
            Conv2d(out_channels, out_channels, kernel_size=3, padding=1),
        ]
        if use_output_act:
            layers += [InstanceNorm2d(out_channels, relu=True), FusedReLU()]
        self.double_conv = nn.Sequential(*layers)

    def ends_in_norm_relu(self):
        m = self.double_conv
        return len(m) == 6 and isinstance(m[4], InstanceNorm2d) and m[4].relu

    def forward(self, x, up2x=False, skip=None, raw_tail=False, grad_group=None):
        """raw_tail=True: stop before the last InstanceNorm(+ReLU) and return (raw conv output, its statistics partials or
        None) for a consumer that normalises while it reads (ResBlock)."""
        # conv -> InstanceNorm pairs: the conv's epilogue leaves the norm's statistics (ops.conv2d want_stats)
        layers = list(self.double_conv)
        if raw_tail:
            layers = layers[:3]
            x = DoubleConv._run(layers, x, up2x, skip, grad_group)
            return self.double_conv[3](x, want_stats=True, norm_input=True)     # x = the first norm's output, read here only
        return DoubleConv._run(layers, x, up2x, skip, grad_group)

    @staticmethod
    def _run(layers, x, up2x, skip, grad_group=None):
        i = 0
        while i < len(layers):
            layer = layers[i]
            # a convolution behind a norm inside this chain is that norm's only consumer
            kw = dict(up2x=up2x, skip=skip, grad_group=grad_group) if i == 0 else \
                (dict(norm_input=True) if isinstance(layer, Conv2d) and i >= 2 and isinstance(layers[i - 2], InstanceNorm2d) else {})
            if isinstance(layer, Conv2d) and i + 1 < len(layers) and isinstance(layers[i + 1], InstanceNorm2d):
                x, part = layer(x, want_stats=True, **kw)
                x = layers[i + 1](x, part=part)
                i += 2
            else:
                x = layer(x, **kw)
                i += 1
        return x


FUSE_GAMMA_BETA = os.environ.get("VQW_FUSE_GAMMA_BETA", "1") != "0"
GRAD_GROUP_BLOCKS = os.environ.get("VQW_GRAD_GROUP_BLOCKS", "1") != "0"     # 0: autograd sums the ResBlock / style-input gradients (A/B)
RES_TAIL_NORM = os.environ.get("VQW_RES_TAIL_NORM", "1") != "0"      # 0: ResBlock branches apply their norms themselves (A/B)


class StyledDenorm(nn.Module):
    """SPADE-style de-normalisation: BatchNorm2d(affine=False)(x) * (1 + gamma(style)) + beta(style)."""

    def __init__(self, in_channels, style_channels) -> None:
        super().__init__()
        # buffer holder only (running_mean / running_var / num_batches_tracked keep their reference keys)
        self.param_free_norm = nn.BatchNorm2d(in_channels, affine=False)
        self.mlp_shared = nn.Sequential(
            conv3x3(style_channels, in_channels),
            FusedReLU(),
        )
        self.mlp_gamma = conv3x3(in_channels, in_channels)
        self.mlp_beta = conv3x3(in_channels, in_channels)

    def style_maps(self, style, grad_group=None, actv=None):
        """(gamma, beta) of reference blocks.py:85-87: a function of the style input only, so a caller may evaluate
        it ahead of / beside the trunk and hand it to forward().  grad_group: the ops.GradGroup of the convolutions that
        read this style tensor (the two StyledDenorms of a StyledResUpBlock).  actv: mlp_shared's output when the caller has
        computed it already (both StyledDenorms' mlp_shared convolutions in one launch, ops.conv2d_pair)."""
        if actv is None:
            actv = self.mlp_shared[0](style, relu=True, grad_group=grad_group)
        if FUSE_GAMMA_BETA:     # one conv with [gamma | beta] output channels
            return ops.conv2d_cat(actv, self.mlp_gamma.weight, self.mlp_gamma.bias, self.mlp_beta.weight, self.mlp_beta.bias, relu_input=True), None
        return self.mlp_gamma(actv), self.mlp_beta(actv)

    def forward(self, x, style, relu=False, maps=None, residual=None, part=None, residual_norm=None):
        bn = self.param_free_norm
        gamma, beta = maps if maps is not None else self.style_maps(style)
        return ops.spade_norm(x, gamma, beta, bn.running_mean, bn.running_var, self.training,
                              momentum=bn.momentum, eps=bn.eps, relu=relu,
                              num_batches_tracked=bn.num_batches_tracked if self.training else None, residual=residual,
                              part=part, residual_norm=residual_norm)


class PixelShuffle(nn.Module):
    """nn.PixelShuffle(2) as an NHWC permutation kernel."""

    def __init__(self, r):
        super().__init__()
        if r != 2:
            raise NotImplementedError("only PixelShuffle(2) is used on this path")
        self.r = r

    def forward(self, x):
        return ops.pixel_shuffle2(x)


class StyledResUpBlock(nn.Module):
    def __init__(self, in_channels, style_channels, out_channels, use_output_act=True, use_pixel_shuffle=False):
        super().__init__()
        self.use_pixel_shuffle = use_pixel_shuffle
        if use_pixel_shuffle:
            self.up_sample = nn.Sequential(
                Conv2d(in_channels, in_channels * 4, kernel_size=3, padding=1),
                PixelShuffle(2),
            )
        else:
            self.up_sample = nn.Upsample(scale_factor=2, mode='nearest')   # folded into the conv loaders

        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=3, padding=1)
        self.norm1 = StyledDenorm(out_channels, style_channels)
        self.act1 = nn.ReLU(inplace=True)

        self.conv2 = Conv2d(out_channels, out_channels, kernel_size=3, padding=1)
        self.norm2 = StyledDenorm(out_channels, style_channels)
        self.use_output_act = use_output_act
        self.act2 = nn.ReLU(inplace=True) if use_output_act else nn.Identity()

        self.conv = nn.Sequential(
            Conv2d(in_channels, out_channels, kernel_size=3, padding=1),
            InstanceNorm2d(out_channels, relu=True),
            FusedReLU(),
        )

    def style_maps(self, skip_input):
        """Modulation maps of both StyledDenorms on the branch stream (they do not depend on down_input)."""
        # both mlp_shared convolutions read skip_input: one gradient group (their input gradients meet in the second one's epilogue)
        grp = ops.GradGroup(2) if (GRAD_GROUP_BLOCKS and skip_input.requires_grad) else None
        with ops.Branch(skip_input) as br:
            c1, c2 = self.norm1.mlp_shared[0], self.norm2.mlp_shared[0]
            if ops.conv2d_pair_supported(skip_input, c1.weight, c2.weight):
                # both mlp_shared convolutions (+ReLU) read skip_input: one launch on the concatenated weights
                a1, a2 = ops.conv2d_pair(skip_input, c1.weight, c1.bias, c2.weight, c2.bias, relu=True, grad_group=grp)
                m1 = self.norm1.style_maps(skip_input, actv=a1)
                m2 = self.norm2.style_maps(skip_input, actv=a2)
            else:
                m1 = self.norm1.style_maps(skip_input, grad_group=grp)
                m2 = self.norm2.style_maps(skip_input, grad_group=grp)
        return br, m1, m2

    def forward(self, down_input, skip_input, maps=None):
        br, m1, m2 = maps if maps is not None else self.style_maps(skip_input)
        if self.use_pixel_shuffle:
            x, up = self.up_sample(down_input), False
        else:
            x, up = down_input, True
        # x feeds the shortcut convolution and conv1: one gradient group (the second input gradient is added in its kernel's epilogue)
        gx = ops.GradGroup(2) if (GRAD_GROUP_BLOCKS and x.requires_grad) else None
        if up and ops.conv2d_up_pair_supported(x, self.conv[0].weight, self.conv1.weight):
            # the 32-channel level: shortcut conv and conv1 read the same up-sampled input - one 64-cout launch of the nine-product
            # kernel instead of two 32-cout launches of the collapsed form (a third of its rate)
            (s, part), (h, part1) = ops.conv2d_up_pair(x, self.conv[0].weight, self.conv[0].bias, self.conv1.weight, self.conv1.bias,
                                                       grad_group=gx)
        else:
            s, part = self.conv[0](x, up2x=up, want_stats=True, grad_group=gx)      # the shortcut conv's epilogue leaves its norm's statistics
            h, part1 = self.conv1(x, up2x=up, want_stats=True, grad_group=gx)      # ... and norm1's batch statistics
        # the shortcut's InstanceNorm(+ReLU) is applied where its output is consumed - inside norm2's modulation kernel, from the
        # statistics its convolution's epilogue left: the normalised shortcut tensor is never written (ops.spade_norm, residual_norm)
        sn, part_s = self.conv[1], part
        br.join(*m1, *m2)
        h = self.norm1(h, skip_input, relu=True, maps=m1, part=part1)
        h, part = self.conv2(h, want_stats=True)       # the epilogue leaves norm2's batch statistics
        return self.norm2(h, skip_input, relu=self.use_output_act, maps=m2, residual=s, part=part,
                          residual_norm=(part_s, sn.relu, sn.eps))   # shortcut + main, in the kernel
