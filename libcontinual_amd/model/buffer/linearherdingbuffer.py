"""iCaRL herding rehearsal memory (reference core/model/buffer/linearherdingbuffer.py:10-165).

Same attributes and methods.  Feature extraction is the HIP backbone in eval mode; the greedy
mean-matching loop (linearherdingbuffer.py:140-161) is one kernel launch per class
(clhip_herding_select) instead of an O(m) Python loop over CPU tensors.  Single-GPU by design.
"""
from typing import List

import numpy as np
import torch
from torch.utils.data import DataLoader

from ... import ops


class LinearHerdingBuffer:
    def __init__(self, buffer_size, batch_size):
        self.buffer_size = buffer_size
        self.strategy = None
        self.batch_size = batch_size
        self.images, self.labels = [], []
        self.total_classes = 0

    def is_empty(self):
        return len(self.labels) == 0

    def clear(self):
        self.images = []
        self.labels = []

    def get_all_data(self):
        return np.array(self.images), np.array(self.labels)

    def add_data(self, data: List[str], targets: List[str]):
        self.images.extend(data)
        self.labels.extend(targets)

    def update(self, model, train_loader, val_transform, task_idx, total_cls_num, cur_cls_indexes, device):
        chosen = self.herding_select(model, train_loader, val_transform, task_idx, total_cls_num, cur_cls_indexes, device)
        ds = train_loader.dataset
        self.add_data([ds.images[i] for i in chosen], [ds.labels[i] for i in chosen])

    def _samples_per_class(self, total_cls_num):
        spc = self.buffer_size // total_cls_num
        if spc == 0:
            print(f"Warning: Buffer size ({self.buffer_size}) is too small for total classes ({total_cls_num}). ",
                  "Samples per class will be set to 1, to avoid empty buffer.")
            spc = 1
        return spc

    def reduce_old_data(self, task_idx, total_cls_num):
        """keep the first buffer_size // total_cls_num exemplars of every stored class (:55-75)"""
        spc = self._samples_per_class(total_cls_num)
        if task_idx > 0:
            X, Y = self.get_all_data()
            self.clear()
            for y in np.unique(Y):
                idx = Y == y
                self.add_data(list(X[idx][:spc]), list(Y[idx][:spc]))

    def herding_select(self, model, train_loader, val_transform, task_idx, total_cls_num, cur_cls_indexes, device):
        """NB: like the reference this MUTATES the task dataset: it is filtered to the current classes
        (class order) and its transform swapped for the test transform (:83-100)."""
        dataset = train_loader.dataset
        labels = np.array(dataset.labels)
        new_images, new_labels = [], []
        for c in cur_cls_indexes:
            ind = np.nonzero(labels == c)[0]
            new_images.extend([dataset.images[i] for i in ind])
            new_labels.extend([dataset.labels[i] for i in ind])
        dataset.images, dataset.labels = new_images, new_labels
        dataset.trfms = val_transform
        loader = DataLoader(dataset, shuffle=False, batch_size=256, drop_last=False)   # eval-mode features: batch size is immaterial
        spc = self._samples_per_class(total_cls_num)
        feats, targets = [], []
        with torch.no_grad():
            model.eval()
            for data in loader:
                f = model.backbone(data["image"].to(device))["features"]
                feats.append(ops.l2_normalize_rows(f))
                targets.append(data["label"].to(device))
        feats = torch.cat(feats)
        targets = torch.cat(targets).cpu().numpy()
        # class samples are contiguous after the filtering above: all classes of the task in ONE launch, one block per class (the reference's
        # per-class loop, :140-161; picks identical to per-class ops.herding_select calls)
        classes, firsts, counts = np.unique(targets, return_index=True, return_counts=True)
        order = np.argsort(firsts)
        firsts, counts = firsts[order], counts[order]
        assert int(firsts[0]) == 0 and all(int(firsts[i]) + int(counts[i]) == int(firsts[i + 1]) for i in range(len(firsts) - 1))
        picks = ops.herding_select_classes(feats, counts.tolist(), spc)
        chosen_of = {int(classes[order[i]]): (picks[i].cpu().numpy() + int(firsts[i])) for i in range(len(counts))}
        result = []
        for c in np.unique(targets):
            result.extend(chosen_of[int(c)].tolist())
        return result
