"""Drop-in `networks` package: `from networks import UNetEncoder, UNetDecoder` (reference:
trainers/base.py:13-20, run_recon.py:13-14) resolves to the MI355X HIP implementation."""
from .unet_encoder import UNetEncoder  # noqa: F401
from .unet_decoder import UNetDecoder  # noqa: F401
from .blocks import UpBlock, ResBlock, DoubleConv, StyledDenorm, StyledResUpBlock  # noqa: F401
from .aspp import ASPP  # noqa: F401
from .vq import VQ  # noqa: F401
from .vqwnet import VQWNet  # noqa: F401
from .random_transform import RandomTransform  # noqa: F401
from .discriminator import NLayerDiscriminator  # noqa: F401
