"""Minimal image transforms (torchvision is not a dependency): the reference's CIFAR ResNet pipeline
(core/data/data.py:4-19) = RandomCrop(32, padding=4) + RandomHorizontalFlip + ColorJitter(brightness=63/255)
+ ToTensor + Normalize(MEAN, STD); test = ToTensor + Normalize -- and the YAML-declared pipelines of the ViT
configs (`train_trfms` / `test_trfms`, core/data/dataloader.py:17-37: RandomResizedCrop, Resize, CenterCrop, ...),
built by `create_transforms`.  Operate on HWC uint8 numpy / PIL images, draw from torch's global RNG (seeded per
epoch by the trainer, core/trainer.py:584)."""
import math

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


_PIL_MODES = {"NEAREST": 0, "LANCZOS": 1, "BILINEAR": 2, "BICUBIC": 3, "BOX": 4, "HAMMING": 5}


def _interp(v):
    if isinstance(v, str):
        return _PIL_MODES[v.upper()]
    return 2 if v is None else int(v)


def _pil(img):
    from PIL import Image
    return img if isinstance(img, Image.Image) else Image.fromarray(_to_hwc_u8(img))


class Resize:
    """int size: shorter side -> size keeping the aspect ratio (torchvision semantics); (h, w): exact"""

    def __init__(self, size, interpolation="BILINEAR"):
        self.size, self.interp = size, _interp(interpolation)

    def __call__(self, img):
        im = _pil(img)
        w, h = im.size
        if isinstance(self.size, int):
            if (w <= h and w == self.size) or (h <= w and h == self.size):
                return np.asarray(im)
            ow, oh = (self.size, int(self.size * h / w)) if w < h else (int(self.size * w / h), self.size)
        else:
            oh, ow = self.size
        return np.asarray(im.resize((ow, oh), self.interp))


class CenterCrop:
    def __init__(self, size):
        self.size = (size, size) if isinstance(size, int) else tuple(size)

    def __call__(self, img):
        a = _to_hwc_u8(img)
        th, tw = self.size
        h, w = a.shape[:2]
        if h < th or w < tw:
            a = np.pad(a, ((max(0, (th - h + 1) // 2),) * 2, (max(0, (tw - w + 1) // 2),) * 2, (0, 0)))
            h, w = a.shape[:2]
        i, j = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
        return a[i:i + th, j:j + tw]


class RandomResizedCrop:
    """torchvision.transforms.RandomResizedCrop: area in `scale` x image area, log-uniform aspect in `ratio`, 10 tries then
    a centre crop at the closest valid aspect; resized to `size`"""

    def __init__(self, size, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), interpolation="BILINEAR"):
        self.size = (size, size) if isinstance(size, int) else tuple(size)
        self.scale, self.ratio, self.interp = tuple(scale), tuple(ratio), _interp(interpolation)

    def __call__(self, img):
        im = _pil(img)
        w, h = im.size
        area = h * w
        log_r = (math.log(self.ratio[0]), math.log(self.ratio[1]))
        box = None
        for _ in range(10):
            ta = area * float(torch.empty(1).uniform_(self.scale[0], self.scale[1]))
            ar = math.exp(float(torch.empty(1).uniform_(log_r[0], log_r[1])))
            cw, ch = int(round(math.sqrt(ta * ar))), int(round(math.sqrt(ta / ar)))
            if 0 < cw <= w and 0 < ch <= h:
                i = int(torch.randint(0, h - ch + 1, (1,)))
                j = int(torch.randint(0, w - cw + 1, (1,)))
                box = (j, i, j + cw, i + ch)
                break
        if box is None:
            in_ratio = w / h
            if in_ratio < self.ratio[0]:
                cw, ch = w, int(round(w / self.ratio[0]))
            elif in_ratio > self.ratio[1]:
                ch, cw = h, int(round(h * self.ratio[1]))
            else:
                cw, ch = w, h
            i, j = (h - ch) // 2, (w - cw) // 2
            box = (j, i, j + cw, i + ch)
        # crop THEN resize (torchvision's F.resized_crop): the interpolation clamps at the crop border instead of reading the
        # source pixels around the box, which `resize(box=...)` would do
        return np.asarray(im.crop(box).resize((self.size[1], self.size[0]), self.interp))


def split_deterministic_head(trfms):
    """(leading Resize / CenterCrop transforms, Compose of the rest): the head does not depend on the random state, so a
    preloading dataset applies it once per image instead of once per sample per epoch"""
    if not isinstance(trfms, Compose):
        return [], trfms
    ts = list(trfms.transforms)
    k = 0
    while k < len(ts) and isinstance(ts[k], (Resize, CenterCrop)):
        k += 1
    return ts[:k], Compose(ts[k:])


_BY_NAME = {}


def create_transforms(cfg):
    """YAML list of {Name: {kwargs}} -> Compose (core/data/dataloader.py:17-37)"""
    out = []
    for item in cfg:
        for name, params in item.items():
            if name not in _BY_NAME:
                raise NotImplementedError(f"transform {name} is not available (have: {sorted(_BY_NAME)})")
            out.append(_BY_NAME[name](**(params or {})))
    return Compose(out)


def cifar_resnet_transform(mode, size=32):
    common = [ToTensor(), Normalize(CIFAR_MEAN, CIFAR_STD)]
    if mode == "train":
        return Compose([RandomCrop(size, padding=4), RandomHorizontalFlip(), ColorJitter(brightness=63 / 255), *common])
    return Compose(common)


_BY_NAME.update(RandomCrop=RandomCrop, RandomHorizontalFlip=RandomHorizontalFlip, ColorJitter=ColorJitter, ToTensor=ToTensor, Normalize=Normalize,
                Resize=Resize, CenterCrop=CenterCrop, RandomResizedCrop=RandomResizedCrop)
