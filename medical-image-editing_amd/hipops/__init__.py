"""hipops — host-side binding of the MI355X VQ-W-Net kernels (libvqwnet_hip.so) for PyTorch-ROCm."""
from . import _lib  # noqa: F401
from . import ops   # noqa: F401
from .optim import Adam  # noqa: F401
