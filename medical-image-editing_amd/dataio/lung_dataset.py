"""Slice dataset of the lung CT training set (reference: dataio/lung_dataset.py:17-80): one directory per patient, one
`*_img_<slice>.npy` file per axial slice in Hounsfield units, mapped to the network's range by a CT window."""
import glob
import os
import random

import numpy as np
from torch.utils import data


def window_normalize(image, width=1500, center=-550, scale=2.0):
    """utils/__init__.py:17-28: clip to [center - width//2, center + width//2], map to [-scale/2, scale/2]."""
    vmax, vmin = center + width // 2, center - width // 2
    out = np.clip(image, vmin, vmax).astype(np.float32)
    out -= vmin
    out /= (vmax - vmin)
    out -= 0.5
    out *= scale
    return out


class NCCLungDataset(data.Dataset):
    def __init__(self, root_dir_path, transform=None, window_width=None, window_center=None, window_scale=None):
        super().__init__()
        self.root_dir_path = str(root_dir_path)
        self.transform = transform
        self.window = (window_width, window_center, window_scale)
        self.files = []
        for patient_id in os.listdir(self.root_dir_path):
            for path in sorted(glob.glob(os.path.join(self.root_dir_path, patient_id, '*_img_*'))):
                stem = os.path.splitext(os.path.basename(path))[0]
                self.files.append({'patient_id': patient_id, 'slice_num': int(stem.split('_')[-1]), 'image_path': path})
        random.shuffle(self.files)          # upstream shuffles the file list once at construction (:37)

    def __len__(self):
        return len(self.files)

    def __getitem__(self, index):
        sample = dict(self.files[index])
        image = np.load(sample['image_path']).astype(np.float32)
        if all(v is not None for v in self.window):
            image = window_normalize(image, *self.window)
        sample['image'] = image
        return self.transform(sample) if self.transform else sample


class MICCAIBraTSDataset(data.Dataset):
    """dataio/miccai_dataset.py:15-68: `<patient>/<x>_<modality>_<slice>.npy` MR slices (intensities 0..255; the loader's
    NormalizeIntensity maps them to [-1, 1]); the file list keeps directory order (no shuffle at construction, as upstream)."""

    def __init__(self, root_dir_path, modality, transform=None):
        super().__init__()
        assert modality in {'t1', 't1ce', 't2', 'flair'}
        self.root_dir_path, self.modality, self.transform = str(root_dir_path), modality, transform
        self.files = []
        for patient_id in os.listdir(self.root_dir_path):
            for path in sorted(glob.glob(os.path.join(self.root_dir_path, patient_id, '*_{}_*'.format(modality)))):
                stem = os.path.splitext(os.path.basename(path))[0]
                self.files.append({'patient_id': patient_id, 'slice_num': int(stem.split('_')[-1]), 'modality': modality,
                                   'image_path': path})

    def __len__(self):
        return len(self.files)

    def __getitem__(self, index):
        sample = dict(self.files[index])
        sample['image'] = np.load(sample['image_path']).astype(np.float32)
        return self.transform(sample) if self.transform else sample


class CRCDataset(data.Dataset):
    """dataio/crc_dataset.py:17-67: `<patient>/<slice>.npy`; the file list is shuffled once at construction (:31)."""

    def __init__(self, root_dir_path, transform=None):
        super().__init__()
        self.root_dir_path, self.transform = str(root_dir_path), transform
        self.files = []
        for patient_id in os.listdir(self.root_dir_path):
            for path in sorted(glob.glob(os.path.join(self.root_dir_path, patient_id, '*.npy'))):
                self.files.append({'patient_id': patient_id, 'slice_num': int(os.path.splitext(os.path.basename(path))[0]),
                                   'image_path': path})
        random.shuffle(self.files)

    def __len__(self):
        return len(self.files)

    def __getitem__(self, index):
        sample = dict(self.files[index])
        sample['image'] = np.load(sample['image_path']).astype(np.float32)
        return self.transform(sample) if self.transform else sample
