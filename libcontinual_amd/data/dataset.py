"""Per-task datasets / loaders (reference core/data/dataset.py:20-99, 232-303, core/data/dataloader.py:76-129).

`SingleDataset` reads a class-folder image tree (data_root/{train,test}/<class>/<image>), the reference's
on-disk format (docs/tutorials/en/data_module_en.md:15-39); `images` are paths relative to data_root/mode
and `labels` ints, both plain lists so the trainer's rehearsal merge (core/trainer.py:305-312) and the
herding buffer can edit them.  `ArrayDataset` is the same interface over in-memory uint8 arrays (synthetic
data, tests).  Batches are dicts {"image": FloatTensor[B,3,H,W], "label": LongTensor[B]}.
"""
import os

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import transforms as T


class SingleDataset(Dataset):
    def __init__(self, data_root, mode, cls_map, trfms, start_idx=0, end_idx=0, init=True):
        self.data_root, self.mode, self.cls_map, self.trfms = data_root, mode, cls_map, trfms
        self.images, self.labels, self.labels_name = [], [], []
        if init:
            for label in range(start_idx, end_idx):
                name = cls_map[label]
                self.labels_name.append(name)
                d = os.path.join(data_root, mode, name)
                for f in sorted(os.listdir(d)):
                    self.images.append(os.path.join(name, f))
                    self.labels.append(label)

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, idx):
        from PIL import Image
        img = Image.open(os.path.join(self.data_root, self.mode, self.images[idx])).convert("RGB")
        return {"image": self.trfms(img), "label": int(self.labels[idx])}


class RaggedStore:
    """uint8 images of DIFFERENT sizes in one flat buffer (ImageNet-R: the reference decodes a JPEG per sample per epoch,
    core/data/dataset.py:248-266; here every image is decoded once): image i = flat[offsets[i]:offsets[i+1]] viewed as
    [hw[i,0], hw[i,1], 3].  `store[i]` gives that view, so the CPU transforms work on it like on an [N,H,W,3] array."""

    def __init__(self, frames):
        self.hw = np.asarray([f.shape[:2] for f in frames], dtype=np.int32).reshape(-1, 2)
        sizes = self.hw[:, 0].astype(np.int64) * self.hw[:, 1] * 3
        self.offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        self.flat = np.empty(int(self.offsets[-1]), np.uint8)
        for i, f in enumerate(frames):
            assert f.dtype == np.uint8 and f.ndim == 3 and f.shape[2] == 3
            self.flat[self.offsets[i]:self.offsets[i + 1]] = f.reshape(-1)

    def __len__(self):
        return len(self.hw)

    def __getitem__(self, i):
        h, w = self.hw[int(i)]
        return self.flat[self.offsets[int(i)]:self.offsets[int(i) + 1]].reshape(h, w, 3)


class DeviceRaggedStore:
    def __init__(self, store, device):
        self.flat, self.offsets, self.hw = (torch.as_tensor(a).to(device) for a in (store.flat, store.offsets[:-1].copy(), store.hw))
        self.hw_host = torch.as_tensor(store.hw)


def store_hw(store):
    """((H, W) or None, ragged) of an image store: the two facts `gpu_plan` needs"""
    return (None, True) if isinstance(store, RaggedStore) else (tuple(store.shape[1:3]), False)


class ArrayDataset(Dataset):
    """`store` is a uint8 array [N,H,W,3] (or a RaggedStore); `images` holds indices into it.  The store object is shared by
    every per-task view (and by copies of a view); `device_store()` uploads it once and keeps it resident for the GPU input
    pipeline."""

    def __init__(self, store, images, labels, trfms, mode="train", resident=None):
        self.store, self.trfms, self.mode, self.data_root = store, trfms, mode, None
        self.images, self.labels = list(images), list(labels)
        self._resident = resident if resident is not None else {}       # device -> uint8 tensor, shared between views

    def __deepcopy__(self, memo):
        """views are copied (the trainer and the plugins edit `images` / `labels` of copies), the image store and its device
        copy are shared"""
        return ArrayDataset(self.store, list(self.images), list(self.labels), self.trfms, self.mode, self._resident)

    def device_store(self, device):
        key = str(device)
        if key not in self._resident:
            self._resident[key] = (DeviceRaggedStore(self.store, device) if isinstance(self.store, RaggedStore)
                                   else torch.as_tensor(np.ascontiguousarray(self.store)).to(device))
        return self._resident[key]

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, idx):
        return {"image": self.trfms(self.store[int(self.images[idx])]), "label": int(self.labels[idx])}


class ContinualDatasets:
    """task t covers labels [start_t, end_t): 0..init for t=0, then inc per task (dataset.py:81-92).
    train: get_loader(t) -> the task's loader; test: the list of loaders of tasks 0..t (dataset.py:94-99)."""

    def __init__(self, mode, task_num, init_cls_num, inc_cls_num, make_dataset, batch_size, num_workers=0, cls_map=None, device=None):
        self.mode, self.task_num, self.cls_map = mode, task_num, cls_map
        self.dataloaders = []
        for i in range(task_num):
            s = 0 if i == 0 else init_cls_num + (i - 1) * inc_cls_num
            e = s + (init_cls_num if i == 0 else inc_cls_num)
            self.dataloaders.append(make_loader(make_dataset(s, e), batch_size, True, num_workers, device))

    def get_loader(self, task_idx):
        assert 0 <= task_idx < self.task_num
        return self.dataloaders[task_idx] if self.mode == "train" else self.dataloaders[: task_idx + 1]


def make_loader(dataset, batch_size, shuffle, num_workers=0, device=None, drop_last=False):
    """the loader of a (per-task or merged) dataset: batches produced on the GPU when the dataset has a resident store, a
    transform pipeline the augment kernels implement and `device` is a HIP device; otherwise torch's DataLoader"""
    if device is not None and torch.device(device).type == "cuda" and hasattr(dataset, "device_store"):
        from .gpu_loader import GpuBatchLoader, gpu_plan
        plan = gpu_plan(dataset.trfms, *store_hw(dataset.store))
        if plan is not None:
            return GpuBatchLoader(dataset, batch_size, shuffle, device, plan, drop_last=drop_last)
    return DataLoader(dataset, shuffle=shuffle, batch_size=batch_size, drop_last=drop_last, num_workers=num_workers, pin_memory=False)


def get_dataloader(config, mode, cls_map=None, device=None):
    """class order: `class_order` key if present else np.random.permutation (seeded by init_seed ->
    the seed-1993 PyCIL order), reference dataloader.py:113-122"""
    data_root = config["data_root"]
    if f"{mode}_trfms" in config:
        trfms = T.create_transforms(config[f"{mode}_trfms"])            # YAML-declared pipeline (dataloader.py:54-55)
    else:
        trfms = T.cifar_resnet_transform(mode, config.get("image_size", 32))
    bs = config.get(f"{mode}_batch_size", config["batch_size"])
    if not config.get("gpu_input_pipeline", True):
        device = None
    if config["dataset"] == "synthetic":
        return synthetic_datasets(config, mode, trfms, bs, device)
    if config.get("preload", False):
        return preloaded_datasets(config, mode, trfms, bs, cls_map, device)
    if cls_map is None:
        cls_list = sorted(os.listdir(os.path.join(data_root, mode)))
        perm = config["class_order"] if "class_order" in config else np.random.permutation(len(cls_list))
        cls_map = {label: cls_list[ori] for label, ori in enumerate(perm)}
    mk = lambda s, e: SingleDataset(data_root, mode, cls_map, trfms, s, e)
    return ContinualDatasets(mode, config["task_num"], config["init_cls_num"], config["inc_cls_num"], mk, bs,
                             config["num_workers"], cls_map)


def preloaded_datasets(config, mode, trfms, bs, cls_map, device):
    """`preload: true`: decode the whole class-folder tree ONCE into a uint8 store so that the per-task datasets are index views
    and the GPU input pipeline can serve every batch from HBM (CIFAR-100 PNG export: 50 000 x 32 x 32 x 3 = 150 MB; ImageNet-R:
    about 13 GB of differently sized images in a RaggedStore).  The DETERMINISTIC head of the transform list (Resize / CenterCrop
    of the test pipelines, config/InfLoRA_opt-vit-imagenetr-b20-20-10.yaml:37-43) is applied here, once, with the same PIL calls
    the per-sample path makes, and removed from the per-batch transform.  Same class order / task split as the on-disk path."""
    from PIL import Image
    data_root = config["data_root"]
    if cls_map is None:
        cls_list = sorted(os.listdir(os.path.join(data_root, mode)))
        perm = config["class_order"] if "class_order" in config else np.random.permutation(len(cls_list))
        cls_map = {label: cls_list[ori] for label, ori in enumerate(perm)}
    n_cls = config["init_cls_num"] + (config["task_num"] - 1) * config["inc_cls_num"]
    head, trfms = T.split_deterministic_head(trfms)
    frames, labels = [], []
    for label in range(n_cls):
        d = os.path.join(data_root, mode, cls_map[label])
        for f in sorted(os.listdir(d)):
            img = Image.open(os.path.join(d, f)).convert("RGB")
            for t in head:
                img = t(img)
            frames.append(np.ascontiguousarray(np.asarray(img)))
            labels.append(label)
    store = np.stack(frames) if len({f.shape for f in frames}) == 1 else RaggedStore(frames)
    labels = np.asarray(labels)
    shared = {}

    def mk(s, e):
        idx = np.flatnonzero((labels >= s) & (labels < e))
        return ArrayDataset(store, idx.tolist(), labels[idx].tolist(), trfms, mode, shared)
    return ContinualDatasets(mode, config["task_num"], config["init_cls_num"], config["inc_cls_num"], mk, bs, config["num_workers"], cls_map, device)


def synthetic_store(n_classes, per_class, size, seed, split=0):
    """class-conditional random images: a class colour times a class-specific oriented grating (so that the
    classes survive the random crop / flip augmentation; shared by the train and test splits), blended with
    per-image noise and a per-image random phase (different per split)"""
    gc = np.random.RandomState(seed)
    colour = gc.rand(n_classes, 3)
    freq = gc.uniform(-4.0, 4.0, size=(n_classes, 2))
    g = np.random.RandomState(seed + 7919 * (split + 1))
    yy, xx = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    imgs = np.empty((n_classes * per_class, size, size, 3), np.uint8)
    labels = np.repeat(np.arange(n_classes), per_class)
    for c in range(n_classes):
        phase = g.uniform(0, 2 * np.pi, size=(per_class, 1, 1))
        wave = 0.5 + 0.5 * np.sin(2 * np.pi * (freq[c, 0] * xx + freq[c, 1] * yy)[None] / size + phase)
        noise = g.rand(per_class, size, size, 3)
        img = 0.6 * wave[..., None] * colour[c] + 0.4 * noise
        imgs[c * per_class:(c + 1) * per_class] = np.clip(img * 255, 0, 255).astype(np.uint8)
    return imgs, labels


def synthetic_datasets(config, mode, trfms, bs, device=None):
    n_cls = config["init_cls_num"] + (config["task_num"] - 1) * config["inc_cls_num"]
    per = config.get("synthetic_per_class", 20) if mode == "train" else config.get("synthetic_test_per_class", 5)
    store, labels = synthetic_store(n_cls, per, config["image_size"], config["seed"], 0 if mode == "train" else 1)

    shared = {}

    def mk(s, e):
        idx = [i for i in range(len(labels)) if s <= labels[i] < e]
        return ArrayDataset(store, idx, [int(labels[i]) for i in idx], trfms, mode, shared)
    return ContinualDatasets(mode, config["task_num"], config["init_cls_num"], config["inc_cls_num"], mk, bs, 0, None, device)
