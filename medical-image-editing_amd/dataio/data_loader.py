"""get_data_loader for the lung CT slices (reference: dataio/data_loader.py:15-149, the NCCLungDataset branch; the
BraTS / CRC branches and the CPU-side torchvision augmentations are not part of this build: augmentation runs on the
device in networks.RandomTransform)."""
import numpy as np
import torch
from torch.utils import data

from .lung_dataset import NCCLungDataset


class ToTensor:
    """dataio/transforms.py:20-34: (H, W) numpy slice -> (1, H, W) float tensor."""

    def __call__(self, sample):
        image = sample['image']
        if image.ndim == 2:
            image = image[np.newaxis, ...]
        sample['image'] = torch.from_numpy(np.ascontiguousarray(image)).float()
        return sample


class SqueezeAxis:
    """dataio/transforms.py:37-51"""

    def __call__(self, sample):
        image = sample['image']
        if image.ndim == 4:
            assert image.size(0) == 1
            sample['image'] = image.squeeze(0)
        return sample


class _Compose:
    def __init__(self, ts):
        self.ts = ts

    def __call__(self, sample):
        for t in self.ts:
            sample = t(sample)
        return sample


def get_data_loader(mode, dataset_name, root_dir_path, batch_size, num_workers, modality=None, augmentations=None,
                    drop_last=False, window_width=None, window_center=None, window_scale=None):
    assert mode in {'train', 'val', 'test'}
    if dataset_name != 'NCCLungDataset':
        raise NotImplementedError("only dataset_name='NCCLungDataset' is built")
    if mode == 'train':
        if augmentations:
            raise NotImplementedError("CPU-side augmentations are not built; use networks.RandomTransform on the device")
        transform, shuffle = _Compose([ToTensor(), SqueezeAxis()]), True
    else:
        assert augmentations is None
        transform, shuffle = _Compose([ToTensor()]), mode == 'val'
    dataset = NCCLungDataset(root_dir_path, transform, window_width, window_center, window_scale)
    return data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers, drop_last=drop_last,
                           pin_memory=True)
