"""Data side of the training loop (reference: src/dataio): pre-sliced `.npy` CT slices with window normalisation."""
from .lung_dataset import NCCLungDataset, MICCAIBraTSDataset, CRCDataset, window_normalize  # noqa: F401
from .data_loader import get_data_loader, ToTensor, SqueezeAxis, NormalizeIntensity  # noqa: F401
