"""Minimal image transforms (torchvision is not a dependency): the reference's CIFAR ResNet pipeline
(core/data/data.py:4-19) = RandomCrop(32, padding=4) + RandomHorizontalFlip + ColorJitter(brightness=63/255)
+ ToTensor + Normalize(MEAN, STD); test = ToTensor + Normalize.  Operate on HWC uint8 numpy / PIL images,
draw from torch's global RNG (seeded per epoch by the trainer, core/trainer.py:584)."""
import numpy as np
import torch

CIFAR_MEAN = [0.5071, 0.4866, 0.4409]   # core/data/data.py:5
CIFAR_STD = [0.2675, 0.2565, 0.2761]    # core/data/data.py:6


def _to_hwc_u8(img):
    a = np.asarray(img)
    if a.ndim == 2:
        a = np.stack([a] * 3, axis=-1)
    return a


class Compose:
    def __init__(self, ts):
        self.transforms = list(ts)

    def __call__(self, img):
        for t in self.transforms:
            img = t(img)
        return img


class RandomCrop:
    def __init__(self, size, padding=0):
        self.size, self.padding = size, padding

    def __call__(self, img):
        a = _to_hwc_u8(img)
        p = self.padding
        if p:
            a = np.pad(a, ((p, p), (p, p), (0, 0)))
        h, w = a.shape[:2]
        i = int(torch.randint(0, h - self.size + 1, (1,)))
        j = int(torch.randint(0, w - self.size + 1, (1,)))
        return a[i:i + self.size, j:j + self.size]


class RandomHorizontalFlip:
    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, img):
        a = _to_hwc_u8(img)
        return a[:, ::-1] if float(torch.rand(1)) < self.p else a


class ColorJitter:
    """brightness only (factor uniform in [1-b, 1+b]), as used by the reference"""

    def __init__(self, brightness=0.0):
        self.b = brightness

    def __call__(self, img):
        a = _to_hwc_u8(img)
        f = float(torch.empty(1).uniform_(max(0.0, 1 - self.b), 1 + self.b))
        return np.clip(a.astype(np.float32) * f, 0, 255).astype(np.uint8)


class ToTensor:
    def __call__(self, img):
        a = _to_hwc_u8(img)
        return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div_(255.0)


class Normalize:
    def __init__(self, mean, std):
        self.mean = torch.tensor(mean).view(-1, 1, 1)
        self.std = torch.tensor(std).view(-1, 1, 1)

    def __call__(self, t):
        return (t - self.mean) / self.std


def cifar_resnet_transform(mode, size=32):
    common = [ToTensor(), Normalize(CIFAR_MEAN, CIFAR_STD)]
    if mode == "train":
        return Compose([RandomCrop(size, padding=4), RandomHorizontalFlip(), ColorJitter(brightness=63 / 255), *common])
    return Compose(common)
