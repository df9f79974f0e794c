"""Scenario definitions shared by the golden generator (reference side) and the tests (oracle / HIP side).
TEST INFRASTRUCTURE (see oracle/__init__).  All inputs and weights are rebuilt from tags via
`oracle.detrand`, so fixtures only store expected outputs.
"""
import math

import numpy as np
import torch

from . import detrand, nets

CIFAR_MEAN = (0.5071, 0.4866, 0.4409)   # reference core/data/data.py:5
CIFAR_STD = (0.2009, 0.1984, 0.2023)    # reference core/data/data.py:6

_DTYPE = [torch.float32]


class use_dtype:
    """Run a scenario in another floating dtype (fp64 = the 'exact' run that pins semantics: several
    quantities on this path -- Fisher of BN scales, lamda=1000 penalties -- amplify fp32 rounding to
    the 1e-3 level, so fp32-vs-fp32 comparisons cannot be tight)."""

    def __init__(self, dt):
        self.dt = dt

    def __enter__(self):
        self.prev = (_DTYPE[0], torch.get_default_dtype())
        _DTYPE[0] = self.dt
        torch.set_default_dtype(self.dt)

    def __exit__(self, *a):
        _DTYPE[0] = self.prev[0]
        torch.set_default_dtype(self.prev[1])


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(_DTYPE[0])


def det_backbone_state(arch, tag):
    """(P, Bf): deterministic backbone weights with the reference's init scale."""
    P, Bf = {}, {}
    for pn, shp in nets.param_shapes(arch):
        t = f"{tag}/{arch}/{pn}"
        if len(shp) == 4:
            cout, cin, k, _ = shp
            std = math.sqrt(2.0 / (k * k * cout))
            a = math.sqrt(3.0) * std
            P[pn] = _t(detrand.uniform(t, shp, -a, a))
        elif pn.startswith("fc."):
            b = 1.0 / math.sqrt(512)
            P[pn] = _t(detrand.uniform(t, shp, -b, b))
        elif pn.endswith(".weight"):
            P[pn] = _t(detrand.uniform(t, shp, 0.8, 1.2))
        else:
            P[pn] = _t(detrand.uniform(t, shp, -0.1, 0.1))
    for bn_, shp in nets.buffer_shapes(arch):
        if bn_.endswith("running_var"):
            Bf[bn_] = torch.ones(shp, dtype=_DTYPE[0])
        elif bn_.endswith("running_mean"):
            Bf[bn_] = torch.zeros(shp, dtype=_DTYPE[0])
        else:
            Bf[bn_] = torch.zeros((), dtype=torch.long)
    return P, Bf


def det_linear(tag, out_f, in_f):
    b = 1.0 / math.sqrt(in_f)
    w = _t(detrand.uniform(tag + "/w", (out_f, in_f), -b, b))
    bias = _t(detrand.uniform(tag + "/b", (out_f,), -b, b))
    return w, bias


def det_images(tag, n, size=32):
    """U[0,1) pixels normalised with the CIFAR mean/std (SURVEY.md section 8d synthetic inputs)."""
    x = detrand.uniform(tag, (n, 3, size, size), 0.0, 1.0)
    m = np.asarray(CIFAR_MEAN, np.float32).reshape(1, 3, 1, 1)
    s = np.asarray(CIFAR_STD, np.float32).reshape(1, 3, 1, 1)
    return _t((x - m) / s)


def det_batch(tag, n, lo, hi):
    return det_images(tag + "/x", n), torch.from_numpy(detrand.randint(tag + "/y", (n,), lo, hi))


def class_images(tag, cls, n, size=32):
    """Images with a weak class-dependent signal (so features of a class cluster)."""
    x = detrand.uniform(f"{tag}/c{cls}", (n, 3, size, size), 0.0, 1.0)
    pat = detrand.uniform(f"{tag}/pattern{cls}", (1, 3, size, size), 0.0, 1.0)
    x = 0.6 * x + 0.4 * pat
    m = np.asarray(CIFAR_MEAN, np.float32).reshape(1, 3, 1, 1)
    s = np.asarray(CIFAR_STD, np.float32).reshape(1, 3, 1, 1)
    return _t(((x - m) / s).astype(np.float32))


def summarize(named):
    """dict name->tensor  ->  (names, float64 [n, 6]): sum, abs-sum, sq-sum, first, last, numel."""
    names, rows = [], []
    for n, t in named.items():
        a = t.detach().double().reshape(-1)
        names.append(n)
        rows.append([a.sum().item(), a.abs().sum().item(), (a * a).sum().item(), a[0].item(), a[-1].item(), float(a.numel())])
    return names, np.asarray(rows, np.float64)


def assert_summary_close(got, want, rtol, atol_scale=1e-7, what=""):
    """compare two `summarize` row arrays: sums compared with an absolute slack proportional to abs-sum."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    for i in range(got.shape[0]):
        scale = max(want[i, 1], 1e-30)
        for c in (0, 1):
            assert abs(got[i, c] - want[i, c]) <= rtol * scale + atol_scale, (what, i, c, got[i], want[i])
        assert abs(got[i, 2] - want[i, 2]) <= 2 * rtol * max(want[i, 2], 1e-30) + atol_scale, (what, i, got[i], want[i])
        n = want[i, 5]
        tol_el = rtol * scale / max(n, 1.0) * 50 + rtol * abs(want[i, 3]) + atol_scale
        assert abs(got[i, 3] - want[i, 3]) <= tol_el + rtol * abs(want[i, 3]), (what, i, 3, got[i], want[i])
        assert got[i, 5] == want[i, 5]
