"""Trainer-side buffer updates (reference core/model/buffer/update.py:7-80)."""
import copy
from collections import Counter

import numpy as np
import torch
from torch.utils.data import DataLoader

from ... import ops


def random_update(datasets, buffer):
    images = np.array(list(datasets.images) + list(buffer.images))
    labels = np.array(list(datasets.labels) + list(buffer.labels))
    perm = np.random.permutation(len(labels))
    buffer.images = images[perm[: buffer.buffer_size]].tolist()
    buffer.labels = labels[perm[: buffer.buffer_size]].tolist()


def herding_update(datasets, buffer, feature_extractor, device):
    """per class, pick buffer_size // total_classes exemplars (update.py:18-45)"""
    per_classes = buffer.buffer_size // buffer.total_classes
    sel_images, sel_labels = [], []
    images = np.array(list(datasets.images) + list(buffer.images))
    labels = np.array(list(datasets.labels) + list(buffer.labels))
    for cls in range(buffer.total_classes):
        idx = np.where(labels == cls)
        ci, cl = construct_examplar(copy.copy(datasets), images[idx], labels[idx], feature_extractor, per_classes, device)
        sel_images.extend(ci)
        sel_labels.extend(cl)
    buffer.images, buffer.labels = list(sel_images), list(sel_labels)


def construct_examplar(datasets, images, labels, feature_extractor, per_classes, device):
    """update.py:47-80: greedy choice where S is the MEAN of the already selected features (sic) and the
    chosen row is deleted from the candidate set."""
    if len(images) <= per_classes:
        return list(images), list(labels)
    datasets.images, datasets.labels = list(images), list(labels)
    loader = DataLoader(datasets, shuffle=False, batch_size=256, drop_last=False)
    feats = []
    with torch.no_grad():
        for data in loader:
            feats.append(feature_extractor(data["image"].to(device))["features"].float().cpu())
    features = torch.cat(feats).numpy().astype(np.float64)
    images, labels = np.array(images), np.array(labels)
    class_mean = np.mean(features, axis=0)
    sel_images, sel_labels, sel_feats = [], [], []
    for k in range(1, per_classes + 1):
        S = np.zeros_like(features[0]) if not sel_feats else np.mean(np.array(sel_feats), axis=0)
        mu_p = (S + features) / k
        i = int(np.argmin(np.sqrt(np.sum((class_mean - mu_p) ** 2, axis=1))))
        sel_images.append(images[i]); sel_labels.append(labels[i]); sel_feats.append(features[i])
        features = np.delete(features, i, axis=0)
        images = np.delete(images, i)
        labels = np.delete(labels, i)
    return sel_images, sel_labels
