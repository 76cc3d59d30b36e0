"""Atrous spatial pyramid (reference: networks/aspp.py:10-47) on the HIP kernels."""
import torch
import torch.nn as nn

from hipops import ops
from .blocks import Conv2d, InstanceNorm2d, FusedReLU


class _ConvBnReLU(nn.Sequential):
    """conv (no bias) -> InstanceNorm -> ReLU; the member is called 'bn' upstream although it is an InstanceNorm."""

    def __init__(self, in_ch, out_ch, kernel_size, stride, padding, dilation, relu=True):
        super().__init__()
        self.add_module("conv", Conv2d(in_ch, out_ch, kernel_size, stride, padding, dilation, bias=False))
        self.add_module("bn", InstanceNorm2d(out_ch, relu=relu))
        self.with_relu = relu
        if relu:
            self.add_module("relu", FusedReLU())


class ASPP(nn.Module):
    def __init__(self, in_ch, out_ch, rates):
        super().__init__()
        self.stages = nn.Module()
        self.stages.add_module("c0", _ConvBnReLU(in_ch, out_ch, 1, 1, 0, 1))
        for i, rate in enumerate(rates):
            self.stages.add_module("c{}".format(i + 1), _ConvBnReLU(in_ch, out_ch, 3, 1, padding=rate, dilation=rate))

    def forward(self, x, grad_group=None):
        """grad_group: an ops.GradGroup the caller has sized for this module's five convolutions PLUS its own further
        consumers of x (the decoder's `x + conv_last(x)`: the residual's gradient seeds the group's buffer)."""
        # every branch normalises straight into its channel slice of the concatenated output
        # (each conv's epilogue leaves the statistics of its branch's norm)
        # the five branches read one tensor: their input gradients are summed in place (ops.GradGroup), not by autograd
        stages = list(self.stages.children())
        grp = grad_group
        if grp is None and torch.is_grad_enabled() and x.requires_grad:
            grp = ops.GradGroup(len(stages))
        # Autograd runs the branches' backward in reverse order of their creation and only the dilated branches have an
        # accumulating input-gradient kernel: the 1 x 1 branch is created LAST, so it runs first and its gradient becomes the
        # group's buffer (created first it ran last and was added by a separate three-pass add kernel).  Output order unchanged.
        order = list(range(1, len(stages))) + [0]
        res = {}
        for i in order:
            res[i] = stages[i].conv(x, want_stats=True, grad_group=grp)
        raw, parts = zip(*[res[i] for i in range(len(stages))])
        return ops.instance_norm_cat(list(raw), relu=True, eps=1e-5, parts=parts)
