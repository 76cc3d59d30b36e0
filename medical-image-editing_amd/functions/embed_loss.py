"""Cross-view embedding loss (reference: functions/embed_loss.py:8-88) on fused HIP kernels.

`forward(embed_1, r_ids_1, embed_2, r_ids_2, codebook)` keeps the reference signature: r_ids_* are the
(B, K, H, W) float one-hot (or soft) maps, codebook is the (D, K) view from `vq.get_codebook()`.
`forward_labels` is the fast path used by the trainer: integer (B, H, W) maps in [0, K] (0 = out of frame), so
the one-hot tensor is never built.  Neither path materialises upstream's (b, D, K, n_loc) broadcast.
"""
import torch.nn as nn

from hipops import ops


class EmbeddingLoss(nn.Module):
    epsilon = 1e-6

    def __init__(self, dict_size: int, margin: float, use_distance_loss: bool, use_regularization_loss: bool):
        super().__init__()
        self.margin = margin
        self.use_distance_loss = use_distance_loss
        self.use_regularization_loss = use_regularization_loss

    def _codebook_terms(self, cb_kd):
        l_dist, l_reg = 0.0, 0.0
        if self.use_distance_loss or self.use_regularization_loss:
            ld, lr = ops.codebook_losses(cb_kd, self.margin)
            l_dist = ld if self.use_distance_loss else 0.0
            l_reg = lr if self.use_regularization_loss else 0.0
        return l_dist, l_reg

    def forward(self, embed_1, r_ids_1, embed_2, r_ids_2, codebook):
        cb_kd = codebook.detach().t().contiguous()            # (K, D); a no-op copy for the vq.embed view
        l_cross = ops.weighted_sum([ops.cross_loss_dense(embed_1, r_ids_2, cb_kd),
                                    ops.cross_loss_dense(embed_2, r_ids_1, cb_kd)], [1.0, 1.0])
        l_dist, l_reg = self._codebook_terms(cb_kd)
        return l_cross, l_dist, l_reg

    def forward_labels(self, embed_1, labels_1, embed_2, labels_2, codebook):
        cb_kd = codebook.detach().t().contiguous()
        l_cross = ops.weighted_sum([ops.cross_loss_labels(embed_1, labels_2, cb_kd),
                                    ops.cross_loss_labels(embed_2, labels_1, cb_kd)], [1.0, 1.0])
        l_dist, l_reg = self._codebook_terms(cb_kd)
        return l_cross, l_dist, l_reg
