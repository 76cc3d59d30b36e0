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

    def forward(self, x):
        # every branch normalises straight into its channel slice of the concatenated output
        # (each conv's epilogue leaves the statistics of its branch's norm)
        # the five branches read one tensor: their input gradients are summed in place (ops.GradGroup), not by autograd
        stages = list(self.stages.children())
        grp = ops.GradGroup(len(stages)) if (torch.is_grad_enabled() and x.requires_grad) else None
        raw, parts = zip(*[stage.conv(x, want_stats=True, grad_group=grp) for stage in stages])
        return ops.instance_norm_cat(list(raw), relu=True, eps=1e-5, parts=parts)
