"""Vector quantiser with EMA codebook (reference: networks/vq/vq_module.py:140-211) on the HIP kernels.

Buffers, constructor and method names follow the reference.  The nearest-codebook
search, gather, commitment loss, EMA statistics and the straight-through backward
are one fused kernel family (csrc/vq.hip); the reference's K x N score matrix, N x K
one-hot and D x N x K GEMM are never built.  The faiss branch of the reference is
replaced by that kernel (knn_backend is accepted and ignored).
"""
from typing import Optional

import torch
from torch import nn

from hipops import ops


class VQModule(nn.Module):
    # multi-GPU EMA statistics: 'global' = counts and sums over the global batch (equals a single-process run on
    # the concatenated batch); 'reference' = the upstream quirk (rank-mean sums, local counts); 'local' = no exchange
    dist_mode = "global"

    def __init__(self, emb_dim: int, dict_size: int, momentum: float, eps: float, knn_backend: Optional[str]) -> None:
        super().__init__()
        assert knn_backend in {"faiss", "torch", "hip", None}
        self.emb_dim = emb_dim
        self.dict_size = dict_size
        self.momentum = momentum
        self.eps = eps
        self._knn_backend = knn_backend
        embed = torch.randn(self.dict_size, self.emb_dim)
        self.register_buffer('embed', embed)
        self.register_buffer('cluster_size', torch.zeros(self.dict_size))
        self.register_buffer('embed_avg', self.embed.T.clone(memory_format=torch.contiguous_format))

    def forward(self, input: torch.Tensor, id_base: int = 0):
        """-> (quantized (B,D,H,W) with straight-through grad, commit_loss, ids).

        ids follow the reference's layout quirk: a (B, W, H)-ordered view (vq_module.py:172-180), i.e.
        ids[b, i, j] is the code of pixel (h=j, w=i); UNetEncoder transposes it back."""
        assert input.size(1) == self.emb_dim
        if input.size(2) != input.size(3):
            raise RuntimeError("VQ expects square maps (the reference's flatten order is only consistent for H == W)")
        q, commit, ids = ops.vq_quantize(input, self.embed, self.cluster_size, self.embed_avg, self.training,
                                         self.momentum, self.eps, dist_mode=self.dist_mode, id_base=id_base)
        return q, commit, ids.transpose(1, 2)

    def lookup(self, ids: torch.Tensor) -> torch.Tensor:
        """F.embedding(ids, embed): ids (B, A, C) -> (B, A, C, D)."""
        return ops.vq_lookup(ids, self.embed).permute(0, 2, 3, 1)

    def get_codebook(self):
        return self.embed.transpose(0, 1)
