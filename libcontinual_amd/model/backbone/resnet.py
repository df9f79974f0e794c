"""ResNet backbones executed as ONE HIP plan per direction (libclhip `clhip_plan_*`).

Drop-in for the reference factories `cifar_resnet20/32`, `resnet18/34` (CIFAR stem), `resnet32_V2`
(core/model/backbone/resnet.py:755-778): same constructor kwargs, same `named_parameters()` /
`named_buffers()` names and shapes (so reference state_dicts load unchanged, and EWC's Fisher / ref
dicts keep their keys), same return contract `{'features': [B, D], 'fmaps': [...]}` and `feature(x)`.

What differs is below the surface:
  * all parameters live in ONE flat fp32 buffer (`_flat`) and all gradients in another (`_gflat`);
    the nn.Parameters are views.  Conv weights keep the logical [K,C,R,S] shape with channels_last
    strides, i.e. K,R,S,C in memory -- the layout the implicit-GEMM kernels read.  Optimizer, EWC and
    the data-parallel all-reduce then work on one contiguous range (one launch / one bucket).
  * forward = clhip_plan_forward (NHWC bf16 activations, MFMA implicit-GEMM conv, BN statistics in the
    conv epilogue), backward = clhip_plan_backward; autograd sees a single Function whose backward
    accumulates straight into `_gflat`.
There is no torch fallback: CPU tensors raise.
"""
import math
import os
import weakref

import torch
import torch.nn as nn

from ... import _lib
from ..._lib import call, require_gpu

__all__ = ["HipResNet", "_Scratch", "_ScratchMixin", "cifar_resnet20", "cifar_resnet32", "cifar_resnet32_V2", "resnet18", "resnet34", "resnet32_V2",
           "CosineLinear", "SplitCosineLinear"]


def _dtype_code(dtype):
    if dtype is None:
        dtype = os.environ.get("CLHIP_DTYPE", "bf16")
    if dtype in ("bf16", torch.bfloat16, _lib.BF16):
        return _lib.BF16
    if dtype in ("f32", "fp32", "float32", torch.float32):
        return _lib.F32
    raise ValueError(f"unsupported compute dtype {dtype!r}")


# ---------------------------------------------------------------------------------------- topology
class _U:
    """one conv -> BN -> (+res) -> (ReLU) unit.  `flags`: extra CLHIP_UNIT_* bits (pre-activation networks); `bn` None = no BN"""
    __slots__ = ("conv", "bn", "cin", "cout", "k", "stride", "pad", "src", "res", "relu", "flags")

    def __init__(self, conv, bn, cin, cout, k, stride, pad, src, res, relu, flags=0):
        self.conv, self.bn, self.cin, self.cout, self.k = conv, bn, cin, cout, k
        self.stride, self.pad, self.src, self.res, self.relu, self.flags = stride, pad, src, res, relu, flags


_PRE_RES, _RAW_SRC, _NO_BN = 2, 4, 8        # include/clhip.h: CLHIP_UNIT_*


def _preact_topology(depth):
    """ResNet_BIC (resnet.py:619-680): conv1, three stages of BasicBlock2 (:589-617), final bn + ReLU, AvgPool2d(8).  A block is
    bn1-ReLU-conv1-bn2-ReLU-conv2 (+ shortcut on the raw sums), so every conv is grouped with the BatchNorm that FOLLOWS it:
    conv1 of the net with layer1.0.bn1, a block's conv1 with its bn2, its conv2 (+ shortcut) with the next block's bn1 (or the
    final bn); the 1x1 shortcut convs (:652-656) have no BatchNorm and read the raw block input."""
    n = (depth - 2) // 6
    spec = [("layer1", 16, 1), ("layer2", 32, 2), ("layer3", 64, 2)]
    blocks = [(name, b, planes, stride if b == 0 else 1) for name, planes, stride in spec for b in range(n)]
    units = [_U("conv1", "layer1.0.bn1", 3, 16, 3, 1, 1, 0, -1, True)]
    cur, inplanes = 1, 16
    for j, (name, b, planes, stride) in enumerate(blocks):
        blk = f"{name}.{b}"
        nxt = f"{blocks[j + 1][0]}.{blocks[j + 1][1]}.bn1" if j + 1 < len(blocks) else "bn"
        units.append(_U(f"{blk}.conv1", f"{blk}.bn2", inplanes, planes, 3, stride, 1, cur, -1, True))
        a1, skip = len(units), cur
        if stride != 1 or inplanes != planes:
            units.append(_U(f"{blk}.downsample.0", None, inplanes, planes, 1, stride, 0, cur, -1, False, _RAW_SRC | _NO_BN))
            skip = len(units)
        units.append(_U(f"{blk}.conv2", nxt, planes, planes, 3, 1, 1, a1, skip, True, _PRE_RES))
        cur, inplanes = len(units), planes
    return units


def _stage(units, prefix, blocks, cin, cout, stride, src, names, no_last_relu=False):
    """Append `blocks` basic blocks; `src` = activation index feeding the stage; returns the last index.
    Activation index i+1 is the output of units[i] (0 = network input)."""
    ca, ba, cb, bb = names
    for i in range(blocks):
        s = stride if i == 0 else 1
        c_in = cin if i == 0 else cout
        blk = f"{prefix}.{i}"
        units.append(_U(f"{blk}.{ca}", f"{blk}.{ba}", c_in, cout, 3, s, 1, src, -1, True))
        a1 = len(units)
        res = src
        if i == 0 and (s != 1 or c_in != cout):   # 1x1 strided conv + BN on the skip path
            units.append(_U(f"{blk}.downsample.0", f"{blk}.downsample.1", c_in, cout, 1, s, 0, src, -1, False))
            res = len(units)
        relu = not (no_last_relu and i == blocks - 1)
        units.append(_U(f"{blk}.{cb}", f"{blk}.{bb}", cout, cout, 3, 1, 1, a1, res, relu))
        src = len(units)
    return src


def _topology(kind, depth=None, layers=None):
    units = []
    if kind == "cifar":            # CifarResNet: resnet.py:324-395
        n = (depth - 2) // 6
        units.append(_U("conv_1_3x3", "bn_1", 3, 16, 3, 1, 1, 0, -1, True))
        nm = ("conv_a", "bn_a", "conv_b", "bn_b")
        s = _stage(units, "stage_1", n, 16, 16, 1, 1, nm)
        s = _stage(units, "stage_2", n, 16, 32, 2, s, nm)
        s = _stage(units, "stage_3", n, 32, 64, 2, s, nm)
        stages = [("stage_1", n), ("stage_2", n), ("stage_3", n)]
        return units, 64, [], stages
    if kind == "modified":         # modified_ResNet: resnet.py:508-560 (no ReLU after the last block)
        n = layers[0]
        units.append(_U("conv1", "bn1", 3, 16, 3, 1, 1, 0, -1, True))
        nm = ("conv1", "bn1", "conv2", "bn2")
        s = _stage(units, "layer1", layers[0], 16, 16, 1, 1, nm)
        s = _stage(units, "layer2", layers[1], 16, 32, 2, s, nm)
        s = _stage(units, "layer3", layers[2], 32, 64, 2, s, nm, no_last_relu=True)
        return units, 64, [], [("layer1", layers[0]), ("layer2", layers[1]), ("layer3", layers[2])]
    if kind == "imagenet_style":   # ResNet with the CIFAR stem: resnet.py:110-223
        units.append(_U("conv1.0", "conv1.1", 3, 64, 3, 1, 1, 0, -1, True))
        nm = ("conv1", "bn1", "conv2", "bn2")
        s = _stage(units, "layer1", layers[0], 64, 64, 1, 1, nm)
        s = _stage(units, "layer2", layers[1], 64, 128, 2, s, nm)
        s = _stage(units, "layer3", layers[2], 128, 256, 2, s, nm)
        s = _stage(units, "layer4", layers[3], 256, 512, 2, s, nm)
        extra = [("fc.weight", (20, 512)), ("fc.bias", (20,))]   # unused head the reference keeps (resnet.py:183)
        return units, 512, extra, [(f"layer{i + 1}", layers[i]) for i in range(4)]
    if kind == "preact":           # ResNet_BIC: resnet.py:619-680
        return _preact_topology(depth), 256, [], []      # feat_dim = 256 is the reference's constant (:644), right for 64 x 64 inputs
    raise ValueError(kind)


def _attach(root, dotted, kind, tensor):
    """register `tensor` under a dotted name, creating container modules on the way."""
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, nn.Module())
        m = m._modules[p]
    if kind == "param":
        m.register_parameter(parts[-1], tensor)
    else:
        m.register_buffer(parts[-1], tensor)


class _PlanHandle:
    """owns clhip_plan pointers (one per input shape); deliberately not deep-copied or pickled."""

    def __init__(self):
        self.plans = {}

    def __deepcopy__(self, memo):
        return _PlanHandle()

    def __getstate__(self):
        return {}

    def __setstate__(self, s):
        self.plans = {}

    def __del__(self):
        try:
            L = _lib.lib()
            for p, _ in self.plans.values():
                L.clhip_plan_destroy(p)
        except Exception:
            pass


class _Feats(dict):
    """{'features': Tensor, 'fmaps': [...]}; fmaps are materialised (fp32 NCHW copies of the stage
    outputs, no grad) only when read."""

    def __init__(self, feats, owner, state):
        super().__init__(features=feats)
        self._owner, self._state = owner, state

    def __getitem__(self, k):
        if k == "fmaps" and not dict.__contains__(self, "fmaps"):
            dict.__setitem__(self, "fmaps", self._owner._read_fmaps(self._state))
        return dict.__getitem__(self, k)

    def __contains__(self, k):
        return k == "fmaps" or dict.__contains__(self, k)

    def keys(self):
        return ["features", "fmaps"]


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, module, state):
        ctx.module, ctx.state = module, state
        return module._forward_impl(x, state)

    @staticmethod
    def backward(ctx, dfeat):
        ctx.module._backward_impl(dfeat, ctx.state)
        return None, None, None, None


class _NoSave:
    """`with resnet.no_backward_follows():` -- forwards made inside it under torch.no_grad() tell the library that nothing they compute is needed by a backward
    (ops.TeacherPass: a teacher that runs on batch statistics): the stage-level launches then skip the stores of z and of the activations inside a run"""
    depth = 0

    def __enter__(self):
        _NoSave.depth += 1

    def __exit__(self, *a):
        _NoSave.depth -= 1


def no_backward_follows():
    return _NoSave()


_LIVE = weakref.WeakSet()        # every HipResNet of the process (ops.TeacherPass asks the training ones whether they take stage-level launches)


class HipResNet(nn.Module):
    def __init__(self, kind, depth=None, layers=None, dtype=None, init="normal_fan_out"):
        super().__init__()
        _LIVE.add(self)
        self._units, self.out_dim, extra, self._stages = _topology(kind, depth, layers)
        self.feat_dim = self.out_dim
        self._dtype = _dtype_code(dtype)
        # ---- flat parameter layout (forward order; the unused `fc` stays outside the flat range so
        #      that, as in the reference, it never receives a gradient and optimizers skip it)
        off = 0
        self._layout = []     # (name, shape, offset, is_conv)
        self._unit_off = []   # flat offset of each unit's first parameter
        self._pool_win = 8 if kind == "preact" else 0
        self._returns_tensor = kind == "preact"      # ResNet_BIC.forward returns the feature tensor, the others a dict
        for u in self._units:
            self._unit_off.append(off)
            tensors = [(u.conv + ".weight", (u.cout, u.cin, u.k, u.k), True)]
            if u.bn is not None:
                tensors += [(u.bn + ".weight", (u.cout,), False), (u.bn + ".bias", (u.cout,), False)]
            for nm, shp, is_conv in tensors:
                n = math.prod(shp)
                self._layout.append((nm, shp, off, is_conv))
                off += (n + 3) // 4 * 4          # keep every tensor 16-byte aligned
        self._nflat = off
        soff = 0
        self._stat_layout = []
        for u in self._units:
            if u.bn is None:
                continue
            for nm in (u.bn + ".running_mean", u.bn + ".running_var"):
                self._stat_layout.append((nm, (u.cout,), soff))
                soff += u.cout
        self._nstat = soff
        flat = torch.zeros(self._nflat)
        stats = torch.zeros(self._nstat)
        nbt = torch.zeros(len(self._units), dtype=torch.long)
        self._flat, self._gflat, self._stats, self._nbt = flat, None, stats, nbt
        self._params = []
        for nm, shp, o, is_conv in self._layout:
            p = nn.Parameter(self._view(flat, shp, o, is_conv))
            _attach(self, nm, "param", p)
            self._params.append(p)
        for nm, shp, o in self._stat_layout:
            _attach(self, nm, "buffer", stats[o:o + shp[0]])
        for i, u in enumerate(self._units):
            if u.bn is not None:
                _attach(self, u.bn + ".num_batches_tracked", "buffer", nbt[i])
        for nm, shp in extra:
            _attach(self, nm, "param", nn.Parameter(torch.zeros(shp)))
        self.reset_parameters(init)
        self._tag_params()
        self._handle = _PlanHandle()
        self._ws = None
        self._shadow = None
        self._shadow_version = None
        self._generation = 0
        self._last_state = None
        self._grad_segment_hook = None      # set by parallel.GradientReducer while a DP training loop runs
        self._grad_segment_cuts = []

    # ------------------------------------------------------------------ parameter plumbing
    def _tag_params(self):
        ref = weakref.ref(self)
        for p in self._params:
            p._clhip_owner = ref      # lets the fused optimizers find the flat buffer a parameter lives in

    @staticmethod
    def _view(flat, shp, off, is_conv):
        n = math.prod(shp)
        v = flat[off:off + n]
        if is_conv:
            K, Cc, R, S = shp
            return v.view(K, R, S, Cc).permute(0, 3, 1, 2)      # logical [K,C,R,S], memory K,R,S,C
        return v.view(shp)

    def reset_parameters(self, init="normal_fan_out"):
        """conv: N(0, sqrt(2/(k*k*cout))) (resnet.py:348-351 == kaiming_normal fan_out/relu :164-166, :523-525);
        BN weight 1 / bias 0; `fc`: nn.Linear default."""
        with torch.no_grad():
            for (nm, shp, o, is_conv), p in zip(self._layout, self._params):
                if is_conv:
                    K, Cc, R, S = shp
                    p.normal_(0, math.sqrt(2.0 / (R * S * K)))
                elif nm.endswith(".weight"):
                    p.fill_(1.0)
                else:
                    p.zero_()
            for nm, shp, o in self._stat_layout:
                if nm.endswith("running_var"):
                    self._stats[o:o + shp[0]].fill_(1.0)
            if "fc" in self._modules:
                b = 1.0 / math.sqrt(512)
                self.fc.weight.uniform_(-b, b)
                self.fc.bias.uniform_(-b, b)

    def _ensure_flat(self, device):
        """nn.Module._apply (.to/.cuda), deepcopy and `.data = ...` assignments break the views; re-pack
        lazily.  Returns True if a re-pack happened."""
        ok = self._flat.device == device and self._stats.device == device and self._nbt.device == device
        if ok:
            # fast path: the addresses of all parameters and running statistics are what they were after the last (re)pack --
            # any .to() / dtype change / `.data = ...` moves at least one of them
            sig = self.__dict__.get("_pack_sig")
            if sig is not None and sig == self._pointer_signature():
                return False
        if ok:
            base = self._flat.data_ptr()
            for (nm, shp, o, is_conv), p in zip(self._layout, self._params):
                if p.data_ptr() != base + 4 * o or p.device != device or p.dtype != torch.float32:
                    ok = False
                    break
        if ok:
            sb = self._stats.data_ptr()
            for (mod, bname), (nm, shp, o) in zip(self._stat_holders(), self._stat_layout):
                if mod._buffers[bname].data_ptr() != sb + 4 * o:
                    ok = False
                    break
        if ok:
            self.__dict__["_pack_sig"] = self._pointer_signature()
            return False
        with torch.no_grad():
            flat = torch.zeros(self._nflat, device=device, dtype=torch.float32)
            for (nm, shp, o, is_conv), p in zip(self._layout, self._params):
                v = self._view(flat, shp, o, is_conv)
                v.copy_(p.data.to(device=device, dtype=torch.float32))
                p.data = v
                p.grad = None
            stats = torch.zeros(self._nstat, device=device, dtype=torch.float32)
            nbt = torch.zeros(len(self._units), device=device, dtype=torch.long)
            mods = dict(self.named_modules())
            for nm, shp, o in self._stat_layout:
                mname, bname = nm.rsplit(".", 1)
                m = mods[mname]
                stats[o:o + shp[0]].copy_(m._buffers[bname].to(device=device, dtype=torch.float32))
                m._buffers[bname] = stats[o:o + shp[0]]
            for i, u in enumerate(self._units):
                if u.bn is None:
                    continue
                m = mods[u.bn]
                nbt[i] = m._buffers["num_batches_tracked"].to(device)
                m._buffers["num_batches_tracked"] = nbt[i]
            self._flat, self._stats, self._nbt = flat, stats, nbt
            self._gflat = None
            self._shadow_version = None
        return True

    def _pointer_signature(self):
        dp = torch.Tensor.data_ptr
        return (self._flat.data_ptr(), self._stats.data_ptr(), tuple(map(dp, self._params)),
                tuple(dp(m._buffers[n]) for m, n in self._stat_holders()))

    def _stat_holders(self):
        """(module, buffer name) of every running statistic, in `_stat_layout` order -- resolved once: walking
        `named_buffers()` on every forward cost 0.2 ms of host time per call"""
        h = self.__dict__.get("_stat_holder_cache")
        if h is None:
            mods = dict(self.named_modules())
            h = [(mods[nm.rsplit(".", 1)[0]], nm.rsplit(".", 1)[1]) for nm, _, _ in self._stat_layout]
            self.__dict__["_stat_holder_cache"] = h
        return h

    def flat_parameters(self):
        """(flat fp32 parameter buffer, flat gradient buffer) after making sure the views are packed."""
        dev = self._params[0].device
        self._ensure_flat(dev)
        if self._gflat is None or self._gflat.device != dev:
            self._gflat = torch.zeros(self._nflat, device=dev, dtype=torch.float32)
        return self._flat, self._gflat

    def _grad_view(self, i):
        nm, shp, o, is_conv = self._layout[i]
        return self._view(self._gflat, shp, o, is_conv)

    def _grad_views(self):
        """the per-parameter views of the flat gradient buffer, built once per buffer (a step re-attaches the SAME tensor
        objects after `zero_grad(set_to_none=True)`; building ~100 fresh views per backward cost 0.3 ms of host time)"""
        c = self.__dict__.get("_grad_view_cache")
        if c is None or c[0] is not self._gflat:
            c = (self._gflat, [self._grad_view(i) for i in range(len(self._params))])
            self.__dict__["_grad_view_cache"] = c
        return c[1]

    def grads_attached(self):
        """True when every trainable parameter's .grad IS its view of the flat buffer (identity test, no pointer arithmetic)"""
        c = self.__dict__.get("_grad_view_cache")
        if c is None or c[0] is not self._gflat:
            return False
        views = c[1]
        return all(p.grad is views[i] for i, p in enumerate(self._params) if p.requires_grad)

    def attach_grads(self):
        """make every parameter's .grad the matching view of the flat gradient buffer"""
        views = self._grad_views()
        for i, p in enumerate(self._params):
            if p.requires_grad and p.grad is not views[i]:
                g = views[i]
                if p.grad is not None and p.grad.data_ptr() != g.data_ptr():      # foreign grad tensor: keep its content
                    g.copy_(p.grad)
                p.grad = g

    def begin_grad_write(self):
        """Every writer of the flat gradient buffer (this plan's backward, the fused EWC term) calls this first.  torch
        semantics on a flat buffer: a parameter whose `.grad` is None receives a fresh gradient, one that holds a gradient
        accumulates.  So if NO parameter holds a gradient -- the state `optimizer.zero_grad()` leaves, whether it ran before
        the forward or, as in the reference trainer (core/trainer.py:602-604), between forward and backward -- the buffer is
        zeroed and everything written during this backward, by any writer in any order, is a pure accumulation (the first
        writer attaches the views, so the later ones see live gradients and add).  Returns the gradient buffer."""
        _, g = self.flat_parameters()
        if all(p.grad is None for p in self._params):
            if not getattr(self, "_gflat_zeroed", False):           # (the fused SGD left it zeroed: optim._FusedBase.zero_grads_in_step)
                g.zero_()
        self._gflat_zeroed = False                                   # whoever called is about to write into it
        return g

    # ------------------------------------------------------------------------------ execution
    def _plan_for(self, x):
        N, Cin, H, W = x.shape
        key = (N, H, W, self._dtype)
        ent = self._handle.plans.get(key)
        if ent is None:
            descs = (_lib.UnitDesc * len(self._units))()
            offs = {nm: o for nm, shp, o, c in self._layout}
            soffs = {nm: o for nm, shp, o in self._stat_layout}
            for i, u in enumerate(self._units):
                d = descs[i]
                d.cin, d.cout, d.ksize, d.stride, d.pad = u.cin, u.cout, u.k, u.stride, u.pad
                d.src, d.res, d.relu = u.src, u.res, int(u.relu) | u.flags
                d.w_off = offs[u.conv + ".weight"]
                if u.bn is not None:
                    d.gamma_off, d.beta_off = offs[u.bn + ".weight"], offs[u.bn + ".bias"]
                    d.rm_off, d.rv_off = soffs[u.bn + ".running_mean"], soffs[u.bn + ".running_var"]
            L = _lib.lib()
            p = L.clhip_plan_create_ex(descs, len(self._units), N, H, W, Cin, self._dtype, self._pool_win)
            if not p:
                raise _lib.ClhipError("clhip_plan_create failed: " + L.clhip_last_error().decode())
            ent = (p, (int(L.clhip_plan_workspace_bytes(p)), int(L.clhip_plan_shadow_bytes(p)), int(L.clhip_plan_feat_dim(p))))
            self._handle.plans[key] = ent
        return ent

    def _prep_weights(self, plan, shadow_bytes, dev):
        if self._shadow is None or self._shadow.device != dev or self._shadow.numel() < shadow_bytes:
            self._shadow = torch.empty(shadow_bytes, device=dev, dtype=torch.uint8)
            self._shadow_version = None
        ver = (self._flat._version, self._flat.data_ptr())
        if self._shadow_version != ver:
            call("clhip_plan_prep_weights", plan, self._flat.data_ptr(), self._shadow.data_ptr(),
                 torch.cuda.current_stream().cuda_stream)
            self._shadow_version = ver

    def compute_dtype(self, dtype):
        """context manager: run the passes inside it in another compute dtype (its own plan, workspace and weight copies; the
        fp32 master parameters, running statistics and gradient buffer are the same).  EWC uses it for the Fisher pass: squared
        gradients of bf16 activations are 20-40 % off per entry, the fp32 pass matches the reference to 1e-3."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            old, new = self._dtype, _dtype_code(dtype)
            if new != old:
                self._dtype, self._shadow_version = new, None
            try:
                yield self
            finally:
                if new != old:
                    self._dtype, self._shadow_version = old, None
        return scope()

    def mark_params_modified(self):
        """tell the backbone that libclhip kernels rewrote the flat parameters (bypassing torch's version counter)"""
        self._shadow_version = None

    def _forward_impl(self, x, state):
        plan, ws, training = state["plan"], state["ws"], state["training"]
        feat = torch.empty(x.shape[0], state["feat_dim"], device=x.device, dtype=torch.float32)
        # num_batches_tracked += 1 rides on the forward's first launch (it was a torch add kernel per step)
        call("clhip_plan_forward_ex", plan, x.data_ptr(), self._flat.data_ptr(), self._stats.data_ptr(), self._shadow.data_ptr(),
             ws.data_ptr(), feat.data_ptr(), int(state.get("mode", training)), self._nbt.data_ptr() if training else None,
             torch.cuda.current_stream().cuda_stream)
        return feat

    def _backward_impl(self, dfeat, state):
        if state["gen"] != self._generation:
            raise _lib.ClhipError("backward through a backbone forward whose saved activations were overwritten by a "
                                  "later forward of the same module")
        if not state["training"]:
            raise _lib.ClhipError("backward through an eval-mode (running-stat BatchNorm) forward is not supported")
        dfeat = dfeat.contiguous().float()
        g = self.begin_grad_write()
        st = torch.cuda.current_stream().cuda_stream
        hook = self._grad_segment_hook
        if hook is None:
            call("clhip_plan_backward", state["plan"], dfeat.data_ptr(), self._flat.data_ptr(), self._shadow.data_ptr(),
                 state["ws"].data_ptr(), g.data_ptr(), st)
        else:
            # data parallel: run the backward in pieces and announce each finished range of the flat gradient buffer
            # (deepest layers first = the tail of the buffer), so its all-reduce overlaps the rest of the backward
            hi = len(self._units)
            for lo in self._grad_segment_cuts + [0]:
                call("clhip_plan_backward_range", state["plan"], dfeat.data_ptr(), self._flat.data_ptr(), self._shadow.data_ptr(),
                     state["ws"].data_ptr(), g.data_ptr(), hi, lo, st)
                end = self._nflat if hi == len(self._units) else self._unit_off[hi]
                hook(self, self._unit_off[lo], end)
                hi = lo
        self.attach_grads()

    def takes_stage_launches(self, x):
        """True when this backbone's TRAINING passes at x's shape run as stage-level launches (csrc/stage_train.hip: one workgroup per image on every compute unit,
        waiting on in-launch exchanges) -- a second network's forward then gains nothing on a side stream (its kernels find no free compute unit) and pays the fork / join"""
        if x.dim() != 4:
            return False
        ent = self._handle.plans.get((x.shape[0], x.shape[2], x.shape[3], self._dtype))
        if ent is None:
            return False
        L = _lib.lib()
        if int(L.clhip_plan_stage_info(ent[0], 0)) == 0:
            return False
        sw = L.clhip_config_get(b"STAGE_TRAIN")
        return int(sw) != 0 if sw not in (None, b"") else x.shape[0] > 64

    def stage_status(self):
        """0 unless an in-launch wait of a stage-level training launch (csrc/stage_train.hip) ran out -- its grid was not co-resident: two such launches on two
        streams, or several processes on one GPU.  Synchronises the device: call at task / epoch boundaries."""
        L = _lib.lib()
        bad = 0
        for p, _ in self._handle.plans.values():
            bad |= int(L.clhip_plan_stage_status(p))
        return bad

    def grad_cut_for_fraction(self, frac=0.5):
        """unit index k such that the parameters of units >= k (the tail of the flat buffer, whose gradients the backward
        produces first) hold at least `frac` of all parameter elements, k as large as possible"""
        for k in range(len(self._units) - 1, 0, -1):
            if self._nflat - self._unit_off[k] >= frac * self._nflat:
                return k
        return 0

    def forward(self, x):
        require_gpu(x)
        if x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        dev = x.device
        self._ensure_flat(dev)
        plan, (ws_bytes, sh_bytes, feat_dim) = self._plan_for(x)
        self._prep_weights(plan, sh_bytes, dev)
        if self._ws is None or self._ws.device != dev or self._ws.numel() < ws_bytes:
            self._ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        self._generation += 1
        need_grad = torch.is_grad_enabled() and self._params[0].requires_grad
        state = dict(plan=plan, ws=self._ws, training=self.training, gen=self._generation, shape=tuple(x.shape), feat_dim=feat_dim)
        if self.training and not need_grad and _NoSave.depth > 0:
            state["mode"] = 2                                         # (clhip_plan_forward_ex: batch statistics, nothing kept for a backward)
        self._last_state = state
        if need_grad:
            self.flat_parameters()
            feats = _BackboneFn.apply(x, self._params[0], self, state)
        else:
            feats = self._forward_impl(x, state)
        if self._returns_tensor:
            return feats
        return _Feats(feats, self, state)

    def feature(self, x):
        return self.forward(x)["features"]

    def _act_dims(self, H, W):
        dims = {0: (H, W)}
        for i, u in enumerate(self._units):
            hs, ws_ = dims[u.src]
            dims[i + 1] = ((hs + 2 * u.pad - u.k) // u.stride + 1, (ws_ + 2 * u.pad - u.k) // u.stride + 1)
        return dims

    def _read_act(self, state, act, which, channels):
        if state["gen"] != self._generation:
            raise _lib.ClhipError("activation requested after the workspace was reused by a later forward")
        N, _, H, W = state["shape"]
        h, w = self._act_dims(H, W)[act]
        t = torch.empty(N, channels, h, w, device=state["ws"].device, dtype=torch.float32)
        call("clhip_plan_read_act", state["plan"], state["ws"].data_ptr(), act, which, t.data_ptr(),
             torch.cuda.current_stream().cuda_stream)
        return t

    def _read_fmaps(self, state):
        """fp32 NCHW copies of the stage outputs ('fmaps', reference resnet.py:392-395, 220-223)"""
        fmaps = []
        for sname, _ in self._stages:
            last = max(i for i, u in enumerate(self._units) if u.conv.startswith(sname + "."))
            fmaps.append(self._read_act(state, last + 1, 0, self._units[last].cout))
        return fmaps

    def debug_read(self, act, which=0):
        """tests: fp32 NCHW copy of activation `act` of the LAST forward (0 = padded input);
        which: 0 = y, 1 = pre-BN conv output z, 2 = gradient wrt y"""
        c = 8 if act == 0 else self._units[act - 1].cout
        return self._read_act(self._last_state, act, which, c)

    # deepcopy (teachers: lwf.py:49, icarl.py:172, lucir.py:87) must not share workspaces or plans
    def __deepcopy__(self, memo):
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        import copy
        for k, v in self.__dict__.items():
            if k in ("_ws", "_shadow", "_gflat", "_last_state", "_stat_holder_cache", "_grad_view_cache", "_pack_sig"):
                new.__dict__[k] = None
            elif k == "_handle":
                new.__dict__[k] = _PlanHandle()
            else:
                new.__dict__[k] = copy.deepcopy(v, memo)
        new._shadow_version = None
        # re-point the parameter list at the copied Parameter objects
        named = dict(new.named_parameters())
        new._params = [named[nm] for nm, *_ in new._layout]
        new._tag_params()
        _LIVE.add(new)
        return new


# ---------------------------------------------------------------------------------------- factories
def cifar_resnet20(pretrained=False, **kwargs):
    return HipResNet("cifar", depth=20, dtype=kwargs.get("dtype"))


def cifar_resnet32(pretrained=False, **kwargs):
    """reference factory resnet.py:760-763 (EWC / iCaRL backbone)"""
    return HipResNet("cifar", depth=32, dtype=kwargs.get("dtype"))


def cifar_resnet32_V2(pretrained=False, **kwargs):
    """ResNet_BIC(32) (resnet.py:924-927 `cifar_resnet32_V2`): the pre-activation backbone of the BiC configs; returns the
    feature TENSOR (AvgPool2d(8) + flatten: 256 features for the 64 x 64 inputs its `feat_dim` is written for)"""
    return HipResNet("preact", depth=32, dtype=kwargs.get("dtype"))


def resnet32_V2(pretrained=False, **kwargs):
    """LUCIR backbone, reference factory resnet.py:769-773"""
    return HipResNet("modified", layers=[5, 5, 5], dtype=kwargs.get("dtype"))


def _imagenet_style(layers, kwargs):
    args = kwargs.get("args")
    assert args is not None, "you should pass args to resnet"          # resnet.py:132
    ds = args["dataset"]
    if not ("cifar" in ds or "5-datasets" in ds):
        raise NotImplementedError("only the CIFAR stem (conv3x3 s1, no max-pool; resnet.py:133-135) is on the hot path")
    if kwargs.get("pretrained"):
        raise NotImplementedError
    return HipResNet("imagenet_style", layers=layers, dtype=kwargs.get("dtype"))


def resnet18(pretrained=False, progress=True, **kwargs):
    """reference factory resnet.py:259-267 (LwF backbone with args.dataset containing 'cifar')"""
    return _imagenet_style([2, 2, 2, 2], dict(kwargs, pretrained=pretrained))


def resnet34(pretrained=False, progress=True, **kwargs):
    return _imagenet_style([3, 4, 6, 3], dict(kwargs, pretrained=pretrained))


# ------------------------------------------------------------------------------------ cosine heads
class _Scratch:
    """per-forward tensors (autograd-attached) that must not follow the module through deepcopy/pickle"""

    def __deepcopy__(self, memo):
        return _Scratch()

    def __getstate__(self):
        return {}


class _ScratchMixin:
    @property
    def last_scores(self):
        return self._scratch.scores

    @property
    def last_features(self):
        return self._scratch.features


class CosineLinear(_ScratchMixin, nn.Module):
    """sigma * normalize(x) @ normalize(W)^T  (reference resnet.py:418-441) on libclhip kernels."""

    def __init__(self, in_features, out_features, sigma=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self._scratch = _Scratch()
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        if sigma:
            self.sigma = nn.Parameter(torch.empty(1))
        else:
            self.register_parameter("sigma", None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.weight.size(1))
        self.weight.data.uniform_(-stdv, stdv)
        if self.sigma is not None:
            self.sigma.data.fill_(1)

    def forward(self, input):
        from ... import ops
        out = ops.cosine_linear(input, self.weight)
        self._scratch.scores = out                 # pre-sigma scores (what the reference's forward hooks capture)
        if self.sigma is not None:
            out = ops.sigma_scale(out, self.sigma)
        return out


class SplitCosineLinear(_ScratchMixin, nn.Module):
    """fc1 (old classes) and fc2 (new classes) concatenated, then sigma (reference resnet.py:443-463)."""

    def __init__(self, in_features, out_features1, out_features2, sigma=True):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features1 + out_features2
        self._scratch = _Scratch()
        self.fc1 = CosineLinear(in_features, out_features1, False)
        self.fc2 = CosineLinear(in_features, out_features2, False)
        if sigma:
            self.sigma = nn.Parameter(torch.empty(1))
            self.sigma.data.fill_(1)
        else:
            self.register_parameter("sigma", None)

    def forward(self, x):
        from ... import ops
        # one cosine kernel over the stacked weight rows == cat(fc1(x), fc2(x)); torch.cat only moves data
        w = torch.cat((self.fc1.weight, self.fc2.weight), dim=0)
        out = ops.cosine_linear(x, w)
        self._scratch.scores = out
        if self.sigma is not None:
            out = ops.sigma_scale(out, self.sigma)
        return out
