"""DropBlock regulariser and its linear schedule (reference: networks/dropblock.py:8-94).

Optional on the path (use_dropblock is off in the only concrete upstream config).  The Bernoulli
seed mask is drawn on the host exactly as upstream does (`torch.rand` on the CPU, dropblock.py:58);
block dilation and the masked rescale are small device ops.
"""
import numpy as np
import torch
from torch import nn

from hipops import ops


class LinearScheduler(nn.Module):
    def __init__(self, dropblock, start_value, stop_value, nr_steps):
        super().__init__()
        self.dropblock = dropblock
        self.i = 0
        self.drop_values = np.linspace(start=start_value, stop=stop_value, num=int(nr_steps))

    def forward(self, x):
        return self.dropblock(x)

    def step(self):
        if self.i < len(self.drop_values):
            self.dropblock.drop_prob = self.drop_values[self.i]
        self.i += 1


class DropBlock2D(nn.Module):
    def __init__(self, drop_prob, block_size):
        super().__init__()
        self.drop_prob = drop_prob
        self.block_size = block_size

    def forward(self, x):
        if not self.training:
            return x
        assert x.dim() == 4, "Expected input with 4 dimensions (bsize, channels, height, width)"
        gamma = self._compute_gamma(x)
        mask = (torch.rand(x.shape[0], *x.shape[2:]) < gamma).float().to(x.device)    # host RNG, as upstream
        return self.apply_seed_mask(x, mask)

    def apply_seed_mask(self, x, mask):
        """Deterministic part (dropblock.py:61-74) given the Bernoulli seed mask (B,H,W) on the device."""
        keep, scale = ops.dropblock_mask(mask, self.block_size)
        self.block_mask = keep
        if self.drop_prob == 0.:
            return x
        return ops.dropblock_apply(x, keep, scale)

    def _compute_block_mask(self, mask):
        """keep = 1 - dilate(mask, block_size) (stride-1 max-pool, pad block_size//2, crop for even sizes)."""
        b = self.block_size
        pad = b // 2
        B, H, W = mask.shape
        padded = torch.zeros(B, H + 2 * pad, W + 2 * pad, dtype=mask.dtype, device=mask.device)
        padded[:, pad:pad + H, pad:pad + W] = mask
        Ho, Wo = H + 2 * pad - b + 1, W + 2 * pad - b + 1
        out = torch.zeros(B, Ho, Wo, dtype=mask.dtype, device=mask.device)
        for dy in range(b):
            for dx in range(b):
                out = torch.maximum(out, padded[:, dy:dy + Ho, dx:dx + Wo])
        if b % 2 == 0:
            out = out[:, :-1, :-1]
        return 1 - out

    def _compute_gamma(self, x):
        return self.drop_prob / (self.block_size ** 2)
