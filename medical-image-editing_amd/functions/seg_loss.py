"""Soft Dice and Focal losses (reference: functions/seg_loss.py:15-62) on one fused HIP kernel pair."""
import torch.nn as nn

from hipops import ops


class SoftDiceLoss(nn.Module):
    def __init__(self, ignore_index=None, smooth=1e-6):
        super().__init__()
        self.smooth = smooth
        self.ignore_index = ignore_index

    def forward(self, output, target):
        dice, _ = ops.seg_losses(output, target, ignore_index=-1 if self.ignore_index is None else self.ignore_index,
                                 smooth=self.smooth)
        return dice


class FocalLoss(nn.Module):
    epsilon = 1e-6

    def __init__(self, gamma=2, alpha=None):
        super().__init__()
        self.gamma = gamma
        self.alpha = alpha

    def forward(self, output, target):
        _, focal = ops.seg_losses(output, target, gamma=float(self.gamma), eps=self.epsilon)
        return focal
