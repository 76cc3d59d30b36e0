"""One-hot encoder (reference: functions/onehot.py:11-20) as one HIP kernel."""
import torch.nn as nn

from hipops import ops


class OneHotEncoder(nn.Module):
    def __init__(self, n_classes):
        super().__init__()
        self.n_classes = n_classes

    def forward(self, t):
        """(B, *spatial) integer labels -> (B, n_classes, *spatial) float32."""
        return ops.onehot(t, self.n_classes)
