"""Datasets with the reference's names and item format (model/dataset.py:82-159):
{'image': float tensor (C,[D,]H,W), 'label': long tensor ([D,]H,W)}."""
import numpy as np
import torch
from torch.utils.data import Dataset

from . import _io


class datasetModelSegwithnpy(Dataset):
    """model/dataset.py:82-115 — .npy volumes (D,H,W) already normalised; label .npy (D,H,W)."""

    def __init__(self, images, labels, targetsize=(16, 64, 128, 128)):
        self.labels, self.images, self.targetsize = labels, images, targetsize

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, index):
        image = np.load(self.images[index])
        d, h, w = image.shape[0], image.shape[1], image.shape[2]
        image = np.reshape(image, (1, d, h, w))
        assert tuple(image.shape) == tuple(self.targetsize), (image.shape, self.targetsize)
        label = np.reshape(np.load(self.labels[index]), (d, h, w))
        return {"image": torch.as_tensor(image).float(), "label": torch.as_tensor(label).long()}


class datasetModelSegwithopencv(Dataset):
    """model/dataset.py:119-159 — grey image files, resized to the target size, z-scored."""

    def __init__(self, images, labels, targetsize=(1, 512, 512)):
        self.labels, self.images, self.targetsize = labels, images, targetsize

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, index):
        c, th, tw = self.targetsize
        image = _io.resize(_io.imread_gray(self.images[index]), (th, tw)).astype(np.float64)
        image = (image - image.mean()) / image.std()
        h, w = image.shape
        image = np.reshape(image, (1, h, w))
        assert tuple(image.shape) == tuple(self.targetsize)
        label = _io.resize(_io.imread_gray(self.labels[index]), (th, tw))
        return {"image": torch.as_tensor(image).float(), "label": torch.as_tensor(np.reshape(label, (h, w))).long()}
