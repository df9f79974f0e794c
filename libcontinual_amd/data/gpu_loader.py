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
from .dataset import DeviceRaggedStore, store_hw


def gpu_plan(trfms, store_hw=None, ragged=False):
    """descriptor of a transform pipeline the augment kernels can run, or None (then the CPU DataLoader path is used).
    store_hw: (H, W) of a uniform store when known; ragged: the store holds images of different sizes"""
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
        return dict(plan, kind="crop_flip", size=None, pad=0) if not ragged else None
    if kinds[0] == "RandomCrop":
        if ragged:
            return None
        plan.update(kind="crop_flip", size=ts[0].size, pad=ts[0].padding)
    elif kinds[0] == "RandomResizedCrop":
        if ts[0].size[0] != ts[0].size[1] or ts[0].interp != 2:
            return None
        # rrc_flip is a plain 2-tap bilinear: right for up-scaling (32 -> 224).  Stores with images larger than the output (or of
        # mixed sizes: ImageNet-R) take rrc_aa, Pillow's anti-aliased resize bit for bit
        aa = ragged or (store_hw is not None and max(store_hw) > ts[0].size[0])
        plan.update(kind="rrc_aa" if aa else "rrc_flip", size=ts[0].size[0], scale=ts[0].scale, ratio=ts[0].ratio)
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


def _rrc_boxes(Hs, Ws, scale, ratio):
    """torchvision RandomResizedCrop.get_params for B samples at once (Hs, Ws: int64 tensors [B], the size of each source image):
    10 candidate boxes each, first valid one, else the centre crop at the closest valid aspect ratio -> int32 [B, 4] = (y0, x0, h, w)"""
    B = Hs.numel()
    Hc, Wc = Hs[:, None], Ws[:, None]
    ta = torch.empty(B, 10).uniform_(scale[0], scale[1]) * (Hc * Wc).float()
    ar = torch.exp(torch.empty(B, 10).uniform_(math.log(ratio[0]), math.log(ratio[1])))
    w = torch.round(torch.sqrt(ta * ar)).long()
    h = torch.round(torch.sqrt(ta / ar)).long()
    ok = (w > 0) & (w <= Wc) & (h > 0) & (h <= Hc)
    first = torch.where(ok.any(1), ok.float().argmax(1), torch.zeros(B, dtype=torch.long))
    w, h = w.gather(1, first[:, None]).squeeze(1), h.gather(1, first[:, None]).squeeze(1)
    u = torch.rand(B, 2)
    bad = ~ok.any(1)
    if bool(bad.any()):
        in_ratio = Ws.double() / Hs.double()
        narrow, wide = in_ratio < ratio[0], in_ratio > ratio[1]
        fw = torch.where(wide, torch.round(Hs.double() * ratio[1]).long(), Ws)
        fh = torch.where(narrow, torch.round(Ws.double() / ratio[0]).long(), Hs)
        w, h = torch.where(bad, fw, w), torch.where(bad, fh, h)
    y0 = (u[:, 0] * (Hs - h + 1).clamp(min=1).float()).long()
    x0 = (u[:, 1] * (Ws - w + 1).clamp(min=1).float()).long()
    y0, x0 = torch.where(bad, (Hs - h) // 2, y0), torch.where(bad, (Ws - w) // 2, x0)
    return torch.stack([y0, x0, h, w], 1).int()


class GpuBatchLoader:
    def __init__(self, dataset, batch_size, shuffle, device, plan=None, drop_last=False, rank=0, world=1, num_workers=0):
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, int(batch_size), bool(shuffle), bool(drop_last)
        self.device, self.rank, self.world, self.num_workers, self.pin_memory = torch.device(device), rank, world, num_workers, False
        self.plan = plan if plan is not None else gpu_plan(dataset.trfms, *store_hw(dataset.store))
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
        ragged = isinstance(store, DeviceRaggedStore)
        H, W = (0, 0) if ragged else store.shape[1:3]
        n = len(ds.labels)
        order = torch.randperm(n) if self.shuffle else torch.arange(n)
        if self.world > 1:
            total_padded = (n + self.world - 1) // self.world * self.world
            if total_padded > n:
                order = torch.cat([order, order[:total_padded - n]])
            order = order[self.rank::self.world]
        rows_host = torch.as_tensor(np.asarray(ds.images, dtype=np.int64))[order]
        rows = rows_host.to(dev)
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
            if ragged:
                Hs, Ws = store.hw_host[rows_host, 0].long(), store.hw_host[rows_host, 1].long()
            else:
                Hs, Ws = torch.full((total,), H, dtype=torch.long), torch.full((total,), W, dtype=torch.long)
            boxes = _rrc_boxes(Hs, Ws, plan["scale"], plan["ratio"])
            params = torch.cat([boxes, flip[:, None]], 1).contiguous().to(dev)
            if plan["kind"] == "rrc_aa":
                max_box = int(boxes[:, 2:4].max()) if total else 1          # sizes the coefficient table of the whole epoch
                ws = torch.empty(max(1, _lib.lib().clhip_augment_rrc_aa_ws_bytes(self.batch_size, S, max_box)), dtype=torch.uint8, device=dev)
        for s in range(0, stop, self.batch_size):
            idx = rows[s:s + self.batch_size]
            B = idx.numel()
            out = torch.empty(B, 3, S, S, device=dev, dtype=torch.float32)
            par = params[s:s + B]
            if plan["kind"] == "crop_flip":
                call("clhip_augment_crop_flip", store.data_ptr(), idx.data_ptr(), par.data_ptr(), bright[s:s + B].data_ptr() if bright is not None else None,
                     out.data_ptr(), B, H, W, S, plan["pad"], mean, std, st)
            elif plan["kind"] == "rrc_flip":
                call("clhip_augment_rrc_flip", store.data_ptr(), idx.data_ptr(), par.data_ptr(), out.data_ptr(), B, H, W, S, mean, std, st)
            elif ragged:
                call("clhip_augment_rrc_aa", store.flat.data_ptr(), store.offsets.data_ptr(), store.hw.data_ptr(), idx.data_ptr(), par.data_ptr(),
                     out.data_ptr(), ws.data_ptr(), B, 0, 0, S, max_box, mean, std, st)
            else:
                call("clhip_augment_rrc_aa", store.data_ptr(), None, None, idx.data_ptr(), par.data_ptr(), out.data_ptr(), ws.data_ptr(), B, H, W, S,
                     max_box, mean, std, st)
            yield {"image": out, "label": labels[s:s + self.batch_size]}
