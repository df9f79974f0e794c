"""Batches produced on the MI355X: the uint8 image store of an `ArrayDataset` is uploaded once and
stays resident in HBM; every batch is ONE gather + augment + normalise kernel (csrc/augment.hip) instead of the reference's
PIL-per-sample transforms in DataLoader workers (core/data/dataset.py:248-266).  `GpuBatchLoader` is a drop-in for the
`torch.utils.data.DataLoader` objects the trainer and the plugins handle (`dataset`, `batch_size`, `len()`, iteration over
{"image", "label"} dicts) -- the tensors it yields already live on the device.

Random parameters are drawn on the host with torch's global generator (seeded per epoch by the trainer, core/trainer.py:584):
one permutation per epoch and, in the same breath, the crop offsets / flips / brightness factors (or RandomResizedCrop boxes)
of the whole epoch -- a single upload; per batch only pointer offsets change.  The draw
ORDER differs from the per-sample CPU pipeline, so the augmentations of a given seed differ; their distribution is the same.
"""
import ctypes as C
import math

import numpy as np
import torch

from .. import _lib
from .._lib import call
from . import transforms as T


def gpu_plan(trfms, store_hw=None):
    """descriptor of a transform pipeline the augment kernels can run, or None (then the CPU DataLoader path is used)"""
    if not isinstance(trfms, T.Compose):
        return None
    ts = list(trfms.transforms)
    kinds = [type(t).__name__ for t in ts]
    mean, std = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)
    if kinds and kinds[-1] == "Normalize":
        mean, std = tuple(float(v) for v in ts[-1].mean.flatten()), tuple(float(v) for v in ts[-1].std.flatten())
        ts, kinds = ts[:-1], kinds[:-1]
    if not kinds or kinds[-1] != "ToTensor":
        return None
    ts, kinds = ts[:-1], kinds[:-1]
    plan = dict(mean=mean, std=std, flip=0.0, brightness=0.0)
    if kinds == []:
        return dict(plan, kind="crop_flip", size=None, pad=0)
    if kinds[0] == "RandomCrop":
        plan.update(kind="crop_flip", size=ts[0].size, pad=ts[0].padding)
    elif kinds[0] == "RandomResizedCrop":
        if ts[0].size[0] != ts[0].size[1] or ts[0].interp != 2:
            return None
        if store_hw is not None and max(store_hw) > ts[0].size[0]:
            return None       # the resize kernel is a plain 2-tap bilinear: fine for up-scaling (32 -> 224), aliased for down-scaling, where PIL
                              # (the reference's pipeline) widens the filter support -> stay on the CPU pipeline
        plan.update(kind="rrc_flip", size=ts[0].size[0], scale=ts[0].scale, ratio=ts[0].ratio)
    else:
        return None
    for t, k in zip(ts[1:], kinds[1:]):
        if k == "RandomHorizontalFlip":
            plan["flip"] = float(t.p)
        elif k == "ColorJitter" and plan["kind"] == "crop_flip":
            plan["brightness"] = float(t.b)
        else:
            return None
    return plan


def _rrc_boxes(B, H, W, scale, ratio):
    """torchvision RandomResizedCrop.get_params for B samples at once: 10 candidate boxes each, first valid one, else the
    centre crop at the closest valid aspect ratio -> int32 [B, 4] = (y0, x0, h, w)"""
    area = H * W
    ta = torch.empty(B, 10).uniform_(scale[0], scale[1]) * area
    ar = torch.exp(torch.empty(B, 10).uniform_(math.log(ratio[0]), math.log(ratio[1])))
    w = torch.round(torch.sqrt(ta * ar)).long()
    h = torch.round(torch.sqrt(ta / ar)).long()
    ok = (w > 0) & (w <= W) & (h > 0) & (h <= H)
    first = torch.where(ok.any(1), ok.float().argmax(1), torch.zeros(B, dtype=torch.long))
    w, h = w.gather(1, first[:, None]).squeeze(1), h.gather(1, first[:, None]).squeeze(1)
    u = torch.rand(B, 2)
    y0 = (u[:, 0] * (H - h + 1).clamp(min=1).float()).long()
    x0 = (u[:, 1] * (W - w + 1).clamp(min=1).float()).long()
    bad = ~ok.any(1)
    if bool(bad.any()):
        in_ratio = W / H
        if in_ratio < ratio[0]:
            fw, fh = W, int(round(W / ratio[0]))
        elif in_ratio > ratio[1]:
            fh, fw = H, int(round(H * ratio[1]))
        else:
            fw, fh = W, H
        w[bad], h[bad], y0[bad], x0[bad] = fw, fh, (H - fh) // 2, (W - fw) // 2
    return torch.stack([y0, x0, h, w], 1).int()


class GpuBatchLoader:
    def __init__(self, dataset, batch_size, shuffle, device, plan=None, drop_last=False, rank=0, world=1, num_workers=0):
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, int(batch_size), bool(shuffle), bool(drop_last)
        self.device, self.rank, self.world, self.num_workers, self.pin_memory = torch.device(device), rank, world, num_workers, False
        self.plan = plan if plan is not None else gpu_plan(dataset.trfms, tuple(dataset.store.shape[1:3]) if hasattr(dataset, "store") else None)
        if self.plan is None:
            raise ValueError("this transform pipeline has no GPU plan")
        if self.device.type != "cuda":
            raise _lib.ClhipError("GpuBatchLoader needs a HIP device (use torch.utils.data.DataLoader on the host)")

    def shard(self, rank, world):
        """per-rank view for data parallelism: this rank's slice of every epoch permutation, batch_size // world per rank
        (what DistributedSampler + a per-rank DataLoader do in core/trainer.py:229-241)"""
        return GpuBatchLoader(self.dataset, max(1, self.batch_size // world), self.shuffle, self.device, self.plan, self.drop_last, rank, world)

    def _count(self):
        """samples per rank: the permutation is padded (wrapped around) to a multiple of the world size, like DistributedSampler,
        so every rank runs the same number of batches -- the per-step gradient all-reduce needs that"""
        n = len(self.dataset.labels)
        return (n + self.world - 1) // self.world if self.world > 1 else n

    def __len__(self):
        n = self._count()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        ds, dev, plan = self.dataset, self.device, self.plan
        store = ds.device_store(dev)
        _, H, W, _ = store.shape
        n = len(ds.labels)
        order = torch.randperm(n) if self.shuffle else torch.arange(n)
        if self.world > 1:
            total_padded = (n + self.world - 1) // self.world * self.world
            if total_padded > n:
                order = torch.cat([order, order[:total_padded - n]])
            order = order[self.rank::self.world]
        rows = torch.as_tensor(np.asarray(ds.images, dtype=np.int64))[order].to(dev)
        labels = torch.as_tensor(np.asarray(ds.labels, dtype=np.int64))[order].to(dev)
        S = plan["size"] or H
        mean, std = (C.c_float * 3)(*plan["mean"]), (C.c_float * 3)(*plan["std"])
        st = torch.cuda.current_stream(dev).cuda_stream
        total = order.numel()
        stop = (total // self.batch_size) * self.batch_size if self.drop_last else total
        # the whole epoch's random parameters in one draw and ONE upload (per batch only pointer offsets change)
        flip = (torch.rand(total) < plan["flip"]).int() if plan["flip"] > 0 else torch.zeros(total, dtype=torch.int32)
        bright = None
        if plan["kind"] == "crop_flip":
            span_y, span_x = H + 2 * plan["pad"] - S + 1, W + 2 * plan["pad"] - S + 1
            dy = torch.randint(0, span_y, (total,), dtype=torch.int32) if span_y > 1 else torch.zeros(total, dtype=torch.int32)
            dx = torch.randint(0, span_x, (total,), dtype=torch.int32) if span_x > 1 else torch.zeros(total, dtype=torch.int32)
            params = torch.stack([dy, dx, flip], 1).contiguous().to(dev)
            if plan["brightness"] > 0:
                b = plan["brightness"]
                bright = torch.empty(total).uniform_(max(0.0, 1 - b), 1 + b).to(dev)
        else:
            params = torch.cat([_rrc_boxes(total, H, W, plan["scale"], plan["ratio"]), flip[:, None]], 1).contiguous().to(dev)
        for s in range(0, stop, self.batch_size):
            idx = rows[s:s + self.batch_size]
            B = idx.numel()
            out = torch.empty(B, 3, S, S, device=dev, dtype=torch.float32)
            par = params[s:s + B]
            if plan["kind"] == "crop_flip":
                call("clhip_augment_crop_flip", store.data_ptr(), idx.data_ptr(), par.data_ptr(), bright[s:s + B].data_ptr() if bright is not None else None,
                     out.data_ptr(), B, H, W, S, plan["pad"], mean, std, st)
            else:
                call("clhip_augment_rrc_flip", store.data_ptr(), idx.data_ptr(), par.data_ptr(), out.data_ptr(), B, H, W, S, mean, std, st)
            yield {"image": out, "label": labels[s:s + self.batch_size]}
