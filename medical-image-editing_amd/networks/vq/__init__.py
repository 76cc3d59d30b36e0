from .vq_module import VQModule as VQ  # noqa: F401
from .vq_module import VQModule  # noqa: F401
