"""U-Net encoder + vector quantiser (reference: networks/unet_encoder.py:16-123) on the HIP kernels."""
import torch
import torch.nn as nn

from hipops import ops
from .vq import VQ
from .blocks import UpBlock, StyledResUpBlock
from ._unet import add_half, run_half
from .initialize import init_weights


class UNetEncoder(nn.Module):
    """in_channels, filters, dict_size, momentum, knn_backend, use_styled_up_block, num_gpus, init_embed: the reference's
    constructor (unet_encoder.py:18-27), positionally compatible."""

    def __init__(self, in_channels, filters=[64, 128, 256, 512, 1024], dict_size=512, momentum=0.99, knn_backend='torch',
                 use_styled_up_block=False, num_gpus=4, init_embed=False):
        super().__init__()
        self.dict_size, self.init_embed, self.num_gpus = dict_size, init_embed, num_gpus
        self.dims = filters[0]
        f = list(filters)

        def up_block(k):
            if not use_styled_up_block:
                return UpBlock(f[k - 1] + f[k], f[k - 1])          # concat [up(x) | skip]
            # the reference feeds level 1 with filters[0] channels (unet_encoder.py:45)
            return StyledResUpBlock(f[k] if k > 1 else f[0], f[k - 1], f[k - 1])
        self._levels = add_half(self, 1, in_channels, f, up_block)
        self.vq = VQ(emb_dim=f[0], dict_size=dict_size, momentum=momentum, eps=1e-5, knn_backend=knn_backend)
        init_weights(self, 'kaiming')

    @property
    def name(self):
        return 'UNetEncoder'

    def initialize_embed(self, embed, rank):
        """One-off k-means initialisation of the codebook from the first batch's feature maps (unet_encoder.py:66-91):
        the ranks' features are all-gathered, rank 0 clusters them and broadcasts `vq.embed`.  Only `vq.embed` changes,
        as upstream (`embed_avg` / `cluster_size` keep their constructor values).  kmeans_pytorch is not available
        offline: hipops.ops.kmeans_codebook restates Lloyd's iteration on the VQ kernels (own semantics, documented
        there); `kmeans_seed` / `kmeans_max_iter` are attributes of this module."""
        import torch.distributed as dist
        feats = embed.detach().contiguous()          # NCHW rows for the collective (the gather outputs match it)
        on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        world = dist.get_world_size() if on else 1
        # every rank knows the global row count: refuse BEFORE any collective, or the others would wait in it for ever
        rows_total = world * feats.shape[0] * feats.shape[2] * feats.shape[3]
        if rows_total < self.dict_size:
            raise RuntimeError("k-means needs at least dict_size = %d feature rows, got %d" % (self.dict_size, rows_total))
        if on:
            parts = [torch.empty_like(feats) for _ in range(world)]
            dist.all_gather(parts, feats)
            feats = torch.cat(parts, dim=0)
        is_root = (not on) or dist.get_rank() == 0
        failure = None
        if is_root:
            try:
                rows = feats.permute(0, 2, 3, 1).reshape(-1, self.dims)
                centres, self.kmeans_history = ops.kmeans_codebook(rows, self.dict_size, seed=getattr(self, "kmeans_seed", 0),
                                                                   max_iter=getattr(self, "kmeans_max_iter", 100))
                with torch.no_grad():
                    self.vq.embed.copy_(centres)
            except Exception as e:       # tell the other ranks before re-raising: they are about to wait for the codebook
                failure = e
        if on:
            status = torch.tensor([0 if failure is None else 1], dtype=torch.int32, device=self.vq.embed.device)
            dist.broadcast(status, 0)
            if int(status.item()):
                raise failure if failure is not None else RuntimeError("k-means codebook initialisation failed on rank 0")
            dist.broadcast(self.vq.embed, 0)
        elif failure is not None:
            raise failure
        self.init_embed = True

    def feature_extraction(self, x):
        if ops.WINOGRAD_FWD_ENCODER:
            with ops.winograd_forward():
                return run_half(self, 1, self._levels, x)
        return run_half(self, 1, self._levels, x)

    def forward(self, x, skip_vq=False, rank=False):
        x = self.feature_extraction(x)
        if skip_vq:
            return x
        if not self.init_embed:
            self.initialize_embed(x, rank)
        # the kernel writes code+1 directly (upstream: ids += 1 after the transpose, :115-116)
        x, commit_loss, ids = self.vq(x, id_base=1)
        ids = torch.transpose(ids, 1, 2)
        return x, commit_loss, ids

    def get_embed_from_ids(self, ids):
        # upstream transposes ids, looks up (B,W,H,D) and transposes dims 1<->3 back; the two transposes cancel
        return ops.vq_lookup(ids, self.vq.embed)
