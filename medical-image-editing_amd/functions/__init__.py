"""Drop-in `functions` package (reference: functions/__init__.py): losses of the first training step."""
from .embed_loss import EmbeddingLoss  # noqa: F401
from .onehot import OneHotEncoder  # noqa: F401
from .seg_loss import SoftDiceLoss, FocalLoss  # noqa: F401
from .gan_loss import hinge_d_loss, generator_loss  # noqa: F401
