"""Functional CPU restatement of the in-scope ResNet backbones (TEST INFRASTRUCTURE, see oracle/__init__).

Three architectures, all sequences of "conv -> BatchNorm -> (+ residual) -> (ReLU)" units followed by
a global average pool:

* ``cifar_resnet32``  -- reference `core/model/backbone/resnet.py:289-316, 324-412, 760-763`
* ``resnet18``        -- CIFAR stem, reference `core/model/backbone/resnet.py:26-64, 110-246, 259-267`
* ``resnet32_V2``     -- LUCIR variant without the last ReLU, reference `core/model/backbone/resnet.py:473-576, 769-773`

plus the pre-activation ``cifar_resnet32_V2`` (= ResNet_BIC(32), `resnet.py:589-680, 765-767`), restated block by block in
`_preact_forward` (its units are BN -> ReLU -> conv with the shortcut added to the raw sums, so it does not fit the list above).

Parameters live in a flat ``dict name -> tensor`` whose names/shapes equal the reference's
``named_parameters()`` / buffers, so reference state_dicts load unchanged.
Arithmetic is torch CPU; backward comes from autograd.
"""
from collections import namedtuple
import math

import torch
import torch.nn.functional as F

Unit = namedtuple("Unit", "conv bn cin cout k stride pad src dst res relu")

BN_MOMENTUM = 0.1   # nn.BatchNorm2d default, reference resnet.py:296
BN_EPS = 1e-5


def _basic_stage(units, prefix, nblocks, cin, cout, stride, src, names, last_no_relu=False):
    """names = (conv_a, bn_a, conv_b, bn_b) attribute names of the block type."""
    ca, ba, cb, bb = names
    for i in range(nblocks):
        s = stride if i == 0 else 1
        inp = cin if i == 0 else cout
        blk = f"{prefix}.{i}"
        a1 = f"{blk}.a"
        units.append(Unit(f"{blk}.{ca}", f"{blk}.{ba}", inp, cout, 3, s, 1, src, a1, None, True))
        res = src
        if i == 0 and (s != 1 or inp != cout):
            res = f"{blk}.ds"
            units.append(Unit(f"{blk}.downsample.0", f"{blk}.downsample.1", inp, cout, 1, s, 0, src, res, None, False))
        out = f"{blk}.out"
        relu = not (last_no_relu and i == nblocks - 1)
        units.append(Unit(f"{blk}.{cb}", f"{blk}.{bb}", cout, cout, 3, 1, 1, a1, out, res, relu))
        src = out
    return src


PREACT = "cifar_resnet32_V2"


def _preact_blocks(depth=32):
    """[(prefix, inplanes, planes, stride, has_shortcut)] in forward order: resnet.py:638-641, 650-665"""
    n = (depth - 2) // 6
    out, inplanes = [], 16
    for name, planes, stride in (("layer1", 16, 1), ("layer2", 32, 2), ("layer3", 64, 2)):
        for b in range(n):
            s = stride if b == 0 else 1
            out.append((f"{name}.{b}", inplanes, planes, s, s != 1 or inplanes != planes))
            inplanes = planes
    return out


def _preact_param_shapes():
    """registration order of ResNet_BIC / BasicBlock2: conv1; per block bn1, conv1, bn2, conv2, downsample.0; final bn"""
    out = [("conv1.weight", (16, 3, 3, 3))]
    for blk, cin, planes, s, ds in _preact_blocks():
        out += [(f"{blk}.bn1.weight", (cin,)), (f"{blk}.bn1.bias", (cin,)), (f"{blk}.conv1.weight", (planes, cin, 3, 3)),
                (f"{blk}.bn2.weight", (planes,)), (f"{blk}.bn2.bias", (planes,)), (f"{blk}.conv2.weight", (planes, planes, 3, 3))]
        if ds:
            out.append((f"{blk}.downsample.0.weight", (planes, cin, 1, 1)))
    return out + [("bn.weight", (64,)), ("bn.bias", (64,))]


def _preact_forward(P, Bf, x, train):
    """ResNet_BIC.forward (resnet.py:667-678) with BasicBlock2.forward (:601-617) inlined"""
    def bn(name, t):
        y = F.batch_norm(t, Bf[name + ".running_mean"], Bf[name + ".running_var"], P[name + ".weight"], P[name + ".bias"], train,
                         BN_MOMENTUM, BN_EPS)
        if train:
            Bf[name + ".num_batches_tracked"] += 1
        return y
    s = F.conv2d(x, P["conv1.weight"], None, 1, 1)
    for blk, cin, planes, stride, ds in _preact_blocks():
        out = F.conv2d(F.relu(bn(blk + ".bn1", s)), P[blk + ".conv1.weight"], None, stride, 1)
        out = F.conv2d(F.relu(bn(blk + ".bn2", out)), P[blk + ".conv2.weight"], None, 1, 1)
        residual = F.conv2d(s, P[blk + ".downsample.0.weight"], None, stride, 0) if ds else s      # the RAW block input
        s = out + residual
    y = F.avg_pool2d(F.relu(bn("bn", s)), 8)
    return y.reshape(y.shape[0], -1)


def arch(name):
    """-> (units, feat_dim, extra_params) ; extra_params = [(name, shape)] that exist but are unused."""
    units = []
    if name == PREACT:
        return units, 256, []          # feat_dim: resnet.py:644 (64 channels x 2 x 2 windows of a 64 x 64 input)
    if name == "cifar_resnet32":
        # stem conv3x3 3->16 + BN + ReLU (resnet.py:337-338, 382-383); 3 stages x 5 blocks (:341-343)
        units.append(Unit("conv_1_3x3", "bn_1", 3, 16, 3, 1, 1, "input", "stem", None, True))
        src = "stem"
        nm = ("conv_a", "bn_a", "conv_b", "bn_b")
        src = _basic_stage(units, "stage_1", 5, 16, 16, 1, src, nm)
        src = _basic_stage(units, "stage_2", 5, 16, 32, 2, src, nm)
        src = _basic_stage(units, "stage_3", 5, 32, 64, 2, src, nm)
        return units, 64, []
    if name == "resnet32_V2":
        units.append(Unit("conv1", "bn1", 3, 16, 3, 1, 1, "input", "stem", None, True))
        src = "stem"
        nm = ("conv1", "bn1", "conv2", "bn2")
        src = _basic_stage(units, "layer1", 5, 16, 16, 1, src, nm)
        src = _basic_stage(units, "layer2", 5, 16, 32, 2, src, nm)
        src = _basic_stage(units, "layer3", 5, 32, 64, 2, src, nm, last_no_relu=True)  # resnet.py:501-502
        return units, 64, []
    if name == "resnet18":
        # CIFAR stem: conv3x3 3->64 s1 + BN + ReLU, no maxpool (resnet.py:133-135)
        units.append(Unit("conv1.0", "conv1.1", 3, 64, 3, 1, 1, "input", "stem", None, True))
        src = "stem"
        nm = ("conv1", "bn1", "conv2", "bn2")
        src = _basic_stage(units, "layer1", 2, 64, 64, 1, src, nm)
        src = _basic_stage(units, "layer2", 2, 64, 128, 2, src, nm)
        src = _basic_stage(units, "layer3", 2, 128, 256, 2, src, nm)
        src = _basic_stage(units, "layer4", 2, 256, 512, 2, src, nm)
        # unused head kept by the reference (resnet.py:183): part of parameters()
        return units, 512, [("fc.weight", (20, 512)), ("fc.bias", (20,))]
    raise KeyError(name)


def param_shapes(name):
    """Ordered [(pname, shape)] matching the reference's named_parameters() order."""
    if name == PREACT:
        return _preact_param_shapes()
    units, _, extra = arch(name)
    # reference registration order: per block conv_a, bn_a, conv_b, bn_b, downsample
    out = []
    by_block = {}
    order = []
    for u in units:
        blk = u.conv.rsplit(".", 1)[0] if "." in u.conv else ""
        if u.conv.endswith("downsample.0"):
            blk = u.conv[: -len(".downsample.0")]
        if u.src == "input":
            blk = "__stem__"
        if blk not in by_block:
            by_block[blk] = []
            order.append(blk)
        by_block[blk].append(u)
    for blk in order:
        us = by_block[blk]
        main = [u for u in us if not u.conv.endswith("downsample.0")]
        ds = [u for u in us if u.conv.endswith("downsample.0")]
        for u in main + ds:
            out.append((u.conv + ".weight", (u.cout, u.cin, u.k, u.k)))
            out.append((u.bn + ".weight", (u.cout,)))
            out.append((u.bn + ".bias", (u.cout,)))
    out.extend(extra)
    return out


def buffer_shapes(name):
    if name == PREACT:
        out = []
        for pn, shp in _preact_param_shapes():
            if len(shp) == 1 and pn.endswith(".weight"):
                b = pn[: -len(".weight")]
                out += [(b + ".running_mean", shp), (b + ".running_var", shp), (b + ".num_batches_tracked", ())]
        return out
    units, _, _ = arch(name)
    out = []
    for u in units:
        out.append((u.bn + ".running_mean", (u.cout,)))
        out.append((u.bn + ".running_var", (u.cout,)))
        out.append((u.bn + ".num_batches_tracked", ()))
    return out


def init_params(name, generator=None):
    """Reference-distribution init (not RNG-stream equal; SURVEY.md section 7 'RNG parity')."""
    g = generator
    P = {}
    for pn, shp in param_shapes(name):
        if len(shp) == 4:
            cout, cin, k, _ = shp
            # cifar_resnet32: N(0, sqrt(2/(k*k*cout))) resnet.py:348-351 ; others kaiming fan_out == same std
            std = math.sqrt(2.0 / (k * k * cout))
            P[pn] = torch.randn(shp, generator=g) * std
        elif pn.startswith("fc."):
            bound = 1.0 / math.sqrt(512)
            P[pn] = (torch.rand(shp, generator=g) * 2 - 1) * bound
        elif pn.endswith(".weight"):
            P[pn] = torch.ones(shp)
        else:
            P[pn] = torch.zeros(shp)
    return P


def init_buffers(name):
    Bf = {}
    for bn_, shp in buffer_shapes(name):
        if bn_.endswith("running_var"):
            Bf[bn_] = torch.ones(shp)
        elif bn_.endswith("running_mean"):
            Bf[bn_] = torch.zeros(shp)
        else:
            Bf[bn_] = torch.zeros((), dtype=torch.long)
    return Bf


def forward(name, P, Bf, x, train, return_acts=False):
    """Features [B, feat_dim] of NCHW fp32 `x`.  In train mode BN uses batch statistics and updates
    the running stats in `Bf` in place (momentum 0.1, unbiased var) exactly like nn.BatchNorm2d."""
    if name == PREACT:
        return _preact_forward(P, Bf, x, train)
    units, feat_dim, _ = arch(name)
    acts = {"input": x}
    for u in units:
        z = F.conv2d(acts[u.src], P[u.conv + ".weight"], None, u.stride, u.pad)
        rm, rv = Bf[u.bn + ".running_mean"], Bf[u.bn + ".running_var"]
        y = F.batch_norm(z, rm, rv, P[u.bn + ".weight"], P[u.bn + ".bias"], train, BN_MOMENTUM, BN_EPS)
        if train:
            Bf[u.bn + ".num_batches_tracked"] += 1
        if u.res is not None:
            y = y + acts[u.res]
        if u.relu:
            y = F.relu(y)
        acts[u.dst] = y
        if return_acts:
            acts[u.dst + "#z"] = z
    last = acts[units[-1].dst]
    feats = last.mean(dim=(2, 3))   # AvgPool2d(8) on 8x8 / AdaptiveAvgPool2d(1): resnet.py:160, 344, 518
    if return_acts:
        return feats, acts
    return feats
