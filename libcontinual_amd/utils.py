"""Host utilities with the reference's semantics (core/utils/utils.py): `get_instance` reflection (:77-92),
`init_seed` (:56-75), accuracy-table metrics `compute_bwt` / `compute_frgt` (:202-232), parameter counters.
`AverageMeter` keeps the reference's interface but is plain Python floats (no pandas writes in the step loop,
SURVEY.md section 3.3)."""
import os
import random

import numpy as np
import torch


class AverageMeter:
    def __init__(self, name, keys, writer=None):
        self.name, self.keys, self.writer = name, list(keys), writer
        self.reset()

    def reset(self):
        self._last = {k: 0.0 for k in self.keys}
        self._total = {k: 0.0 for k in self.keys}
        self._count = {k: 0 for k in self.keys}
        self._pending = {}

    def update(self, key, value, n=1):
        """`value` may be a device-resident scalar (ops.Deferred or 0-dim tensor): it is queued and only
        read back when an average is requested (one sync per epoch instead of 2-3 per step)."""
        if hasattr(value, "tensor") or torch.is_tensor(value):
            self._pending.setdefault(key, []).append((value, n))
        else:
            self._last[key] = value
            self._total[key] += value * n
        self._count[key] += n

    def _flush(self, key):
        pend = self._pending.pop(key, None)
        if not pend:
            return
        ts, ws = [], []
        for v, n in pend:
            if torch.is_tensor(v):
                ts.append(v.detach().reshape(()).float()); ws.append(float(n))
            else:
                ts.append(v.tensor.detach().reshape(()).float()); ws.append(float(v.scale) * n)
        vals = torch.stack(ts).cpu().double() * torch.tensor(ws, dtype=torch.float64)
        self._total[key] += float(vals.sum())
        self._last[key] = float(vals[-1]) / (pend[-1][1] or 1)

    def avg(self, key):
        self._flush(key)
        return self._total[key] / self._count[key] if self._count[key] else 0.0

    def result(self):
        return {k: self.avg(k) for k in self.keys}

    def last(self, key):
        self._flush(key)
        return self._last[key]

    def total(self, key):
        self._flush(key)
        return self._total[key]


def quiesce_gc():
    """Collect once, then move every surviving object to the permanent generation (`gc.freeze`).  A training process holds
    ~10^6 long-lived Python objects (torch, numpy, the model); every generation-2 collection walks all of them -- measured
    80-90 ms of host stall in the middle of a 30-step bench run, during which the launch queue drains and the GPU idles.
    After the freeze a full collection only visits what was allocated since, i.e. the per-step garbage."""
    import gc
    gc.collect()
    gc.freeze()


def init_seed(seed=0, deterministic=False):
    os.environ["PYTHONHASHSEED"] = str(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    # (cudnn flags of the reference have no counterpart: libclhip kernels are deterministic except for
    #  the fp32 atomic accumulation order of the weight gradients)


def get_instance(module, name, config, **kwargs):
    """getattr(module, config[name]['name'])(**kwargs, **config[name]['kwargs'])"""
    if config[name].get("kwargs") is not None:
        kwargs.update(config[name]["kwargs"])
    return getattr(module, config[name]["name"])(**kwargs)


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def count_all_parameters(model):
    return sum(p.numel() for p in model.parameters())


def compute_bwt(acc_table, curr_acc, task_idx):
    """backward transfer, formula of core/utils/utils.py:202-221"""
    if task_idx > 1:
        bwt = 0.0
        for i in range(2, task_idx):
            for j in range(i - 1):
                bwt += acc_table[i, j] - acc_table[j, j]
        for j in range(task_idx - 1):
            bwt += curr_acc[j] - acc_table[j, j]
        return (bwt * 2) / (task_idx * (task_idx + 1))
    return 0.0


def compute_frgt(acc_table, curr_acc, task_idx):
    """forgetting, formula of core/utils/utils.py:224-232"""
    if task_idx > 1:
        return sum(np.diag(acc_table)[: task_idx - 1] - np.asarray(curr_acc)[: task_idx + 1][:-2]) / task_idx
    return 0.0


# ------------------------------------------------------------------------------------------------ per-task SVDs (InfLoRA / DualGPM)
def device_svd(a, full_matrices=False, compute_uv=True, device=None):
    """numpy.linalg.svd semantics (float64 in, float64 out) with the factorisation done ON THE GPU in fp64 through torch.linalg
    (rocSOLVER) when one is there -- SURVEY.md section 8(f) rank 3 ("768 x 768 SVD (rocSOLVER)").  This is per-task host math of the reference
    (core/model/InfLoRA_opt.py:251-369, InfLoRA.py:108-308), not the training step: measured on the MI355X box a 768 x 768 factorisation takes 138 ms
    on the device in fp64 (reconstruction error 1.7e-13) against 0.5 s in numpy (fp64) and 1.2-10 s in torch's CPU fp32 driver (the reference's
    `torch.linalg.svd(cur_matrix)`, 128 host threads; error 1.7e-6) -- before_task of a 2 400-image ImageNet-R task 15.3 s -> 2 s
    (profiles/r04_inflora_task_boundary.md).  CLHIP_SVD=host keeps numpy.  `a`: numpy array or CPU / device tensor."""
    import numpy as np
    import torch
    arr = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    use_dev = os.environ.get("CLHIP_SVD", "device") != "host" and torch.cuda.is_available()
    if not use_dev:
        return np.linalg.svd(arr.astype(np.float64), full_matrices=full_matrices, compute_uv=compute_uv)
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    t = torch.as_tensor(arr, dtype=torch.float64).to(dev)
    if not compute_uv:
        return torch.linalg.svdvals(t).cpu().numpy()
    U, S, Vh = torch.linalg.svd(t, full_matrices=full_matrices)
    return U.cpu().numpy(), S.cpu().numpy(), Vh.cpu().numpy()
