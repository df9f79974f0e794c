"""tests/golden/augment_aa.npz: expected outputs of the reference's ImageNet-R input pipeline on fixed images of mixed sizes with
fixed crop boxes (TEST INFRASTRUCTURE; build container only: needs /root/reference).

    python -m oracle.gen_augment_aa_golden

The pipeline is declared in the reference's YAML (config/InfLoRA_opt-vit-imagenetr-b20-20-10.yaml:28-43) and instantiated from
torchvision.transforms by core/data/dataloader.py:17-37.  torchvision is absent from this image; the parameters are read from the
reference's YAML and what torchvision does with a PIL image for these transforms is PIL itself (see gen_augment_golden.py):
    RandomResizedCrop(S, scale, ratio)  Image.crop(box) then Image.resize((S, S), BILINEAR) -- anti-aliased when it shrinks
    Resize(256, BICUBIC)                Image.resize, shorter side -> 256
    CenterCrop(S)                       Image.crop of the centred S x S window
The crop boxes are stored with the images (the GPU kernel takes them as inputs)."""
import os

import numpy as np
import yaml
from PIL import Image

REF_YAML = "/root/reference/config/InfLoRA_opt-vit-imagenetr-b20-20-10.yaml"


def reference_pipeline():
    cfg = yaml.safe_load(open(REF_YAML))
    tr = {k: v for item in cfg["train_trfms"] for k, v in item.items()}
    te = {k: v for item in cfg["test_trfms"] for k, v in item.items()}
    assert list(tr) == ["RandomResizedCrop", "RandomHorizontalFlip", "ToTensor"] and list(te) == ["Resize", "CenterCrop", "ToTensor"]
    return dict(size=tr["RandomResizedCrop"]["size"], scale=tr["RandomResizedCrop"]["scale"], ratio=tr["RandomResizedCrop"]["ratio"],
                flip_p=tr["RandomHorizontalFlip"]["p"], resize=te["Resize"]["size"], resize_interp=te["Resize"]["interpolation"],
                center=te["CenterCrop"]["size"])


def image(k, h, w):
    """smooth colour waves plus a fine checker / stripe region (what an aliased down-scaling would get wrong); few distinct
    values so that the fixture compresses"""
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    base = np.stack([128 + 90 * np.sin(0.05 * (c + 1) * (xx // 4) * 4 + 0.03 * (k + 1) * (yy // 4) * 4) for c in range(3)], -1)
    fine = ((xx + yy * (k % 2 + 1)) % 2) * 120.0 - 60.0
    mask = ((yy // 16 + xx // 16 + k) % 3 == 0)[..., None]
    return (np.round((base + mask * fine[..., None]) / 16) * 16).clip(0, 255).astype(np.uint8)


def main():
    p = reference_pipeline()
    S = p["size"]
    assert S == 224 and p["resize"] == 256 and p["center"] == 224 and p["resize_interp"] == "BICUBIC"
    sizes = [(97, 131), (180, 120), (240, 320), (64, 64), (333, 250), (500, 375)]           # ImageNet-R's typical 500 x 375 among them
    imgs = [image(k, h, w) for k, (h, w) in enumerate(sizes)]
    # top, left, height, width, flip: whole images, strong and mild down-scaling, mixed up/down-scaling, a box narrower than one output pixel row
    boxes = np.asarray([[0, 0, 97, 131, 0], [20, 10, 150, 100, 1], [0, 40, 240, 240, 0], [8, 8, 40, 50, 1], [3, 1, 330, 248, 0],
                        [100, 50, 300, 225, 1], [0, 0, 500, 375, 0], [17, 200, 451, 90, 1]], np.int32)
    which = np.asarray([0, 1, 2, 3, 4, 5, 5, 5], np.int64)
    out_sizes = np.asarray([56, 56, 56, 56, 56, S, 56, 56])         # the reference's 224 once; smaller outputs keep the fixture small
    out = dict(size=np.asarray([S]), scale=np.asarray(p["scale"], np.float64), ratio=np.asarray(p["ratio"], np.float64),
               flip_p=np.asarray([p["flip_p"]]), hw=np.asarray(sizes, np.int32), boxes=boxes, which=which, out_sizes=out_sizes)
    for k, a in enumerate(imgs):
        out[f"image_{k}"] = a
    for j in range(len(boxes)):
        top, left, h, w, fl = (int(v) for v in boxes[j])
        im = Image.fromarray(imgs[which[j]]).crop((left, top, left + w, top + h)).resize((int(out_sizes[j]),) * 2, Image.BILINEAR)
        if fl:
            im = im.transpose(Image.FLIP_LEFT_RIGHT)
        out[f"train_expected_u8_{j}"] = np.asarray(im)
    # the test pipeline on one image (shorter side -> 256 with the bicubic filter, centre 224 x 224 window)
    for k in (1,):
        im = Image.fromarray(imgs[k])
        w, h = im.size
        ow, oh = (p["resize"], int(p["resize"] * h / w)) if w < h else (int(p["resize"] * w / h), p["resize"])
        im = im.resize((ow, oh), Image.BICUBIC)
        i, j = int(round((oh - S) / 2.0)), int(round((ow - S) / 2.0))
        out[f"test_expected_u8_{k}"] = np.asarray(im.crop((j, i, j + S, i + S)))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "augment_aa.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
