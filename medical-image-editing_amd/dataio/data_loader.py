"""get_data_loader (reference: dataio/data_loader.py:15-149): the three dataset branches with their transform chains.  The
CPU-side kornia augmentations (RandomAffineTransform / RandomHorizontalFlipTransform per sample) are not built: augmentation
runs on the device in networks.RandomTransform (a non-empty `augmentations` list raises)."""
import numpy as np
import torch
from torch.utils import data

from .lung_dataset import NCCLungDataset, MICCAIBraTSDataset, CRCDataset


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


class NormalizeIntensity:
    """dataio/transforms.py:53-72: clamp to [vmin, vmax] = [0, 255], map to [-1, 1] (in place on the sample's tensor)."""

    def __init__(self, vmin=0, vmax=255):
        self.vmin, self.vmax = vmin, vmax

    def __call__(self, sample):
        image = torch.clamp(sample['image'], min=self.vmin, max=self.vmax)
        image -= self.vmin
        image /= (self.vmax - self.vmin)
        image *= 2.0
        image -= 1.0
        sample['image'] = image
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
    assert dataset_name in {'MICCAIBraTSDataset', 'NCCLungDataset', 'CRCDataset'}
    intensity = [] if dataset_name == 'NCCLungDataset' else [NormalizeIntensity()]      # CT slices are windowed by the dataset
    if mode == 'train':
        if augmentations:
            raise NotImplementedError("CPU-side augmentations are not built; use networks.RandomTransform on the device")
        transform, shuffle = _Compose([ToTensor()] + intensity + [SqueezeAxis()]), True
    else:
        assert augmentations is None
        transform, shuffle = _Compose([ToTensor()] + intensity), mode == 'val'
    if dataset_name == 'MICCAIBraTSDataset':
        dataset = MICCAIBraTSDataset(root_dir_path, modality, transform)
    elif dataset_name == 'CRCDataset':
        dataset = CRCDataset(root_dir_path, transform)
    else:
        dataset = NCCLungDataset(root_dir_path, transform, window_width, window_center, window_scale)
    return data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers, drop_last=drop_last,
                           pin_memory=True)
