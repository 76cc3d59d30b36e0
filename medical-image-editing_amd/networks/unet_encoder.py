"""U-Net encoder + vector quantiser (reference: networks/unet_encoder.py:16-123) on the HIP kernels."""
import torch
import torch.nn as nn

from hipops import ops
from .vq import VQ
from .blocks import ResBlock, UpBlock, DoubleConv, StyledResUpBlock
from .initialize import init_weights


class UNetEncoder(nn.Module):

    def __init__(self,
                 in_channels: int,
                 filters: list = [64, 128, 256, 512, 1024],
                 dict_size: int = 512,
                 momentum: float = 0.99,
                 knn_backend: str = 'torch',
                 use_styled_up_block: bool = False,
                 num_gpus: int = 4,
                 init_embed: bool = False,
                 ):
        super().__init__()
        self.dict_size = dict_size
        self.init_embed = init_embed
        self.dims = filters[0]
        self.num_gpus = num_gpus

        self.down_conv1_1 = ResBlock(in_channels, filters[0])
        self.down_conv1_2 = ResBlock(filters[0], filters[1])
        self.down_conv1_3 = ResBlock(filters[1], filters[2])
        self.down_conv1_4 = ResBlock(filters[2], filters[3])

        self.double_conv1 = DoubleConv(filters[3], filters[4])

        if use_styled_up_block:
            self.up_conv1_4 = StyledResUpBlock(filters[4], filters[3], filters[3])
            self.up_conv1_3 = StyledResUpBlock(filters[3], filters[2], filters[2])
            self.up_conv1_2 = StyledResUpBlock(filters[2], filters[1], filters[1])
            self.up_conv1_1 = StyledResUpBlock(filters[0], filters[0], filters[0])
        else:
            self.up_conv1_4 = UpBlock(filters[3] + filters[4], filters[3])
            self.up_conv1_3 = UpBlock(filters[2] + filters[3], filters[2])
            self.up_conv1_2 = UpBlock(filters[1] + filters[2], filters[1])
            self.up_conv1_1 = UpBlock(filters[1] + filters[0], filters[0])

        self.vq = VQ(emb_dim=filters[0], dict_size=self.dict_size, momentum=momentum, eps=1e-5,
                     knn_backend=knn_backend)

        init_weights(self, 'kaiming')

    @property
    def name(self):
        return 'UNetEncoder'

    def initialize_embed(self, embed, rank):
        """Upstream runs a one-off k-means (third-party kmeans_pytorch, unet_encoder.py:66-91) over the
        gathered feature maps.  That dependency is not part of this build: supply the codebook
        (`encoder.vq.embed.copy_(centres)`) and construct with init_embed=True."""
        raise RuntimeError("k-means codebook initialisation is not available; load a codebook into "
                           "encoder.vq.embed and pass init_embed=True (config use_init_embed falsy)")

    def feature_extraction(self, x):
        x, skip1 = self.down_conv1_1(x)
        x, skip2 = self.down_conv1_2(x)
        x, skip3 = self.down_conv1_3(x)
        x, skip4 = self.down_conv1_4(x)
        x = self.double_conv1(x)
        x = self.up_conv1_4(x, skip4)
        x = self.up_conv1_3(x, skip3)
        x = self.up_conv1_2(x, skip2)
        x = self.up_conv1_1(x, skip1)
        return x

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
