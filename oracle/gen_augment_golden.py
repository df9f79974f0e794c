"""tests/golden/augment.npz: expected outputs of the reference's CIFAR input pipelines (core/data/data.py:4-35) on fixed images with
fixed random parameters (TEST INFRASTRUCTURE; build container only: needs /root/reference).

    python -m oracle.gen_augment_golden

torchvision is absent from this image, so the reference's `transforms.Compose` objects cannot be instantiated.  What CAN be taken
from the reference is everything that parameterises them -- the tables of `CIFARTransform` are parsed out of the reference's own
source text (no copy: only the numbers end up in the fixture) -- and what torchvision does with a PIL image for these five
transforms is PIL itself (torchvision/transforms/_functional_pil.py calls exactly these PIL entry points):
    RandomCrop(32, padding=4)          ImageOps.expand(border=4, fill=0) then Image.crop at the drawn (top, left)
    RandomHorizontalFlip()             Image.transpose(FLIP_LEFT_RIGHT)
    ColorJitter(brightness=b)          ImageEnhance.Brightness(img).enhance(f), f drawn in [max(0, 1-b), 1+b]
    RandomResizedCrop(S)               Image.crop(box) then Image.resize((S, S), BILINEAR)     (F.resized_crop)
    ToTensor() / Normalize(m, s)       uint8 HWC / 255 -> CHW float32, (x - m) / s
The random parameters (crop offsets, flip bits, brightness factors, crop boxes) are stored with the images: the GPU kernels take them
as inputs (the loader draws them on the host), so kernel parity is a deterministic function of (image, parameters)."""
import ast
import os
import re

import numpy as np
from PIL import Image, ImageEnhance, ImageOps

from . import detrand

REF_DATA_PY = "/root/reference/core/data/data.py"


def reference_tables():
    """MEAN / STD / padding / brightness / RandomResizedCrop size / ViT normalisation, parsed from the reference's source text"""
    src = open(REF_DATA_PY).read()
    body = src[src.index("class CIFARTransform"):]

    def lit(pattern):
        m = re.search(pattern, body)
        assert m, pattern
        return ast.literal_eval(m.group(1))
    t = dict(mean=lit(r"MEAN = (\[[^\]]+\])"), std=lit(r"STD = (\[[^\]]+\])"),
             crop=lit(r"RandomCrop\((\d+), padding=\d+\)"), padding=lit(r"RandomCrop\(\d+, padding=(\d+)\)"),
             brightness=eval(re.search(r"ColorJitter\(brightness=([0-9 /.]+)\)", body).group(1), {"__builtins__": {}}),
             rrc_size=lit(r"RandomResizedCrop\((\d+)\)"), vit_mean=lit(r"dset_mean = (\([^)]+\))"), vit_std=lit(r"dset_std = (\([^)]+\))"))
    assert t["crop"] == 32 and t["padding"] == 4 and abs(t["brightness"] - 63 / 255) < 1e-12 and t["rrc_size"] == 224
    return t


def to_tensor_normalize(img, mean, std):
    a = np.asarray(img, dtype=np.uint8).astype(np.float32) / np.float32(255.0)
    a = a.transpose(2, 0, 1)
    return ((a - np.asarray(mean, np.float32).reshape(3, 1, 1)) / np.asarray(std, np.float32).reshape(3, 1, 1)).astype(np.float32)


def main():
    t = reference_tables()
    out = dict(mean=np.asarray(t["mean"]), std=np.asarray(t["std"]), padding=np.asarray([t["padding"]]), brightness=np.asarray([t["brightness"]]),
               rrc_size=np.asarray([t["rrc_size"]]), vit_mean=np.asarray(t["vit_mean"]), vit_std=np.asarray(t["vit_std"]))
    # ---- resnet_train_transform: RandomCrop(32, padding 4) + flip + brightness jitter + ToTensor + Normalize
    B, S, pad = 6, t["crop"], t["padding"]
    imgs = (detrand.uniform("augment/cifar/images", (B, S, S, 3), 0.0, 256.0)).astype(np.uint8)
    off = (detrand.uniform("augment/cifar/offsets", (B, 2), 0.0, 2 * pad + 1 - 1e-9)).astype(np.int32)          # top, left in [0, 2 pad]
    off[0] = (0, 0); off[1] = (2 * pad, 2 * pad); off[2] = (pad, pad)                                            # the corners and the identity crop
    flip = (detrand.uniform("augment/cifar/flip", (B,), 0.0, 1.0) < 0.5).astype(np.int32)
    b = t["brightness"]
    fac = detrand.uniform("augment/cifar/brightness", (B,), max(0.0, 1 - b), 1 + b).astype(np.float32)
    fac[2] = 1.0
    want = np.zeros((B, 3, S, S), np.float32)
    for k in range(B):
        im = ImageOps.expand(Image.fromarray(imgs[k]), border=pad, fill=0)
        top, left = int(off[k, 0]), int(off[k, 1])
        im = im.crop((left, top, left + S, top + S))
        if flip[k]:
            im = im.transpose(Image.FLIP_LEFT_RIGHT)
        im = ImageEnhance.Brightness(im).enhance(float(fac[k]))
        want[k] = to_tensor_normalize(im, t["mean"], t["std"])
    out.update(cifar_images=imgs, cifar_offsets=off, cifar_flip=flip, cifar_brightness=fac, cifar_train_expected=want,
               cifar_test_expected=np.stack([to_tensor_normalize(Image.fromarray(imgs[k]), t["mean"], t["std"]) for k in range(2)]))
    # ---- vit_train_transform on CIFAR-sized images (BASELINE configs[3]: CIFAR-100 at 224 x 224): crop box, up-scaling bilinear resize, flip
    S2 = t["rrc_size"]
    yy, xx = np.meshgrid(np.arange(32), np.arange(32), indexing="ij")
    vimgs = np.stack([np.stack([(128 + 100 * np.sin(0.2 * (c + 1) * xx + 0.13 * k * yy) + 20 * np.cos(0.7 * yy * (k + 1))) for c in range(3)], -1)
                      for k in range(4)]).clip(0, 255).astype(np.uint8)                                          # smooth: the 224 x 224 outputs compress
    boxes = np.asarray([[0, 0, 32, 32, 0], [3, 5, 20, 17, 1], [10, 2, 9, 28, 0], [16, 16, 16, 16, 1]], np.int32)  # top, left, height, width, flip
    sizes = [S2, 96, 96, 96]                                         # the reference's 224 once; smaller outputs keep the fixture small
    out.update(vit_images=vimgs, vit_boxes=boxes, vit_sizes=np.asarray(sizes))
    for k in range(4):
        top, left, h, w, fl = (int(v) for v in boxes[k])
        im = Image.fromarray(vimgs[k]).crop((left, top, left + w, top + h)).resize((sizes[k], sizes[k]), Image.BILINEAR)
        if fl:
            im = im.transpose(Image.FLIP_LEFT_RIGHT)
        out[f"vit_train_expected_u8_{k}"] = np.asarray(im)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "augment.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
