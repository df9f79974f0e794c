"""Trainer-side buffer updates (reference core/model/buffer/update.py:7-80): `random_update` and the per-class greedy
`herding_update` the trainer calls after every task (core/trainer.py:410-418)."""
import copy

import numpy as np
import torch
from torch.utils.data import DataLoader

__all__ = ["random_update", "herding_update", "construct_examplar"]


def _pool(datasets, buffer):
    """candidates = the task's samples followed by what the buffer already holds"""
    return np.asarray(list(datasets.images) + list(buffer.images)), np.asarray(list(datasets.labels) + list(buffer.labels))


def random_update(datasets, buffer):
    """keep a uniformly random subset of the pool (global numpy RNG, like the reference: update.py:7-16)"""
    images, labels = _pool(datasets, buffer)
    keep = np.random.permutation(labels.shape[0])[: buffer.buffer_size]
    buffer.images, buffer.labels = images[keep].tolist(), labels[keep].tolist()


def herding_update(datasets, buffer, feature_extractor, device):
    """buffer_size // total_classes exemplars per seen class, chosen by `construct_examplar` (update.py:18-45)"""
    quota = buffer.buffer_size // buffer.total_classes
    images, labels = _pool(datasets, buffer)
    kept_images, kept_labels = [], []
    for cls in range(buffer.total_classes):
        mine = np.where(labels == cls)
        ci, cl = construct_examplar(copy.copy(datasets), images[mine], labels[mine], feature_extractor, quota, device)
        kept_images += list(ci)
        kept_labels += list(cl)
    buffer.images, buffer.labels = kept_images, kept_labels


def _features_of(datasets, images, labels, feature_extractor, device):
    datasets.images, datasets.labels = list(images), list(labels)
    out = []
    with torch.no_grad():
        for batch in DataLoader(datasets, shuffle=False, batch_size=256, drop_last=False):
            out.append(feature_extractor(batch["image"].to(device))["features"].float().cpu())
    return torch.cat(out).numpy().astype(np.float64)


def construct_examplar(datasets, images, labels, feature_extractor, per_classes, device):
    """Greedy herding with the reference's two quirks (update.py:47-80): the running term S is the MEAN of the features picked
    so far (not their sum), and a picked row leaves the candidate set.  Candidates are masked instead of deleted; the argmin
    over the survivors in their original order is the same row `np.delete` + `argmin` would pick."""
    if len(images) <= per_classes:
        return list(images), list(labels)
    feats = _features_of(datasets, images, labels, feature_extractor, device)
    target = feats.mean(axis=0)
    alive = np.ones(feats.shape[0], dtype=bool)
    picked = []
    for k in range(1, per_classes + 1):
        S = feats[picked].mean(axis=0) if picked else np.zeros_like(target)
        dist = np.sqrt((((S + feats) / k - target) ** 2).sum(axis=1))
        dist[~alive] = np.inf
        j = int(np.argmin(dist))
        alive[j] = False
        picked.append(j)
    images, labels = np.asarray(images), np.asarray(labels)
    return [images[j] for j in picked], [labels[j] for j in picked]
