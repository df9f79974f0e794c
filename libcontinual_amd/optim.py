"""Fused optimizers on libclhip (replace torch.optim.SGD / Adam constructed at core/trainer.py:159-166).

`SGD` / `Adam` here are torch.optim.Optimizer subclasses (so every LR scheduler keeps working and YAML
`optimizer.name: SGD` resolves to them through the trainer), with torch.optim semantics: params whose
grad is None are skipped, per-group lr / momentum / weight_decay.  The difference is the launch count:
all parameters of a HipResNet live in one flat buffer, so a group that holds a whole backbone is updated
with ONE kernel over the flat range (clhip_sgd_step / clhip_adam_step); any other parameter (heads) is
one launch per tensor.  `grad_scale` folds the data-parallel 1/world_size (or a clip factor) into the step.  After a reduce-scattered exchange
(parallel.GradientReducer, exchange="reduce_scatter") a backbone carries `_dp_shard`: the step then runs on this rank's slice of the
flat buffer only, with optimizer state for that slice only, and all-gathers the updated slices.
"""
import torch

from . import ops
from ._lib import require_gpu


def _owner_of(p):
    ref = getattr(p, "_clhip_owner", None)
    return ref() if ref is not None else None


def _tag_backbones(params):
    """find HipResNet owners of the parameters (set lazily by walking `_clhip_owner` tags)"""
    owners = {}
    for p in params:
        o = _owner_of(p)
        if o is not None:
            owners.setdefault(id(o), (o, []))[1].append(p)
    return owners


class _FusedBase(torch.optim.Optimizer):
    grad_scale = 1.0
    capture_safe = False        # may trainer.GraphedStep capture step() into a HIP graph?
    # the fused SGD may leave a backbone's flat gradient buffer ZEROED after consuming it (the next backward then needs no fill launch: 44.7 MB per
    # ResNet-18 step).  Off by default -- torch semantics: p.grad still holds the gradient after step() -- and switched on by trainer.train_steps
    # for its loop, where zero_grad() follows every step() and nothing reads the gradients in between.
    zero_grads_in_step = False

    def whole_backbones(self):
        """the backbones this optimizer updates with ONE launch over their flat buffer (and whose `_dp_shard` it therefore honours):
        every parameter of the backbone in one group, gradients attached to the flat gradient buffer"""
        out = []
        for group in self.param_groups:
            out += self._split(group)[0]
        return out

    def _check_unconsumed_shards(self, rest):
        """a reduce-scattered exchange leaves the summed gradient in the rank's shard only; a backbone that reaches the per-tensor
        path would step on local, un-reduced gradients and the replicas would diverge silently"""
        for p in rest:
            o = _owner_of(p)
            if o is not None and getattr(o, "_dp_shard", None) is not None:
                raise RuntimeError("a reduce-scattered gradient shard was not consumed: the backbone's parameters are split across "
                                   "param groups or partly frozen; use the all_reduce exchange (parallel.GradientReducer)")

    def _split(self, group):
        """-> (list of (owner) whose full parameter set is in this group with live flat grads, leftover params)"""
        params = [p for p in group["params"]]
        owners = _tag_backbones(params)
        whole, covered = [], set()
        for oid, (o, ps) in owners.items():
            if len(ps) == len(o._params) and o._gflat is not None and all(p.grad is not None for p in o._params):
                ok = o.grads_attached() or all(p.grad.data_ptr() == o._gflat.data_ptr() + 4 * o._layout[i][2] for i, p in enumerate(o._params))
                if ok:
                    whole.append(o)
                    covered.update(id(p) for p in ps)
        rest = [p for p in params if id(p) not in covered and p.grad is not None]
        return whole, rest

    @staticmethod
    def _dense(p):
        """(tensor whose memory is the parameter's elements in storage order)"""
        if p.is_contiguous():
            return p.data.view(-1)
        if p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last):
            return p.data.permute(0, 2, 3, 1).reshape(-1)      # view: K,R,S,C memory order
        raise RuntimeError("parameter is neither contiguous nor channels_last")

    @staticmethod
    def _dense_like(p, g):
        if g.is_contiguous() and p.is_contiguous():
            return g.view(-1)
        if g.dim() == 4 and g.is_contiguous(memory_format=torch.channels_last) and p.is_contiguous(memory_format=torch.channels_last):
            return g.permute(0, 2, 3, 1).reshape(-1)
        # layouts differ: bring the gradient into the parameter's memory order
        if p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last):
            return g.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(-1)
        return g.contiguous().view(-1)


def _dp_plan(o):
    """[(parameter slice, gradient slice, state-key suffix)] of a backbone's flat buffer for this step, and the shard record to publish
    afterwards (None: the whole buffer, no data-parallel sharding)"""
    d = getattr(o, "_dp_shard", None)
    if d is None:
        return [(o._flat, o._gflat, "")], None
    o._dp_shard = None                                   # one exchange feeds one step
    n = o._flat.numel()
    parts = []
    if d["hi"] > d["lo"]:
        parts.append((o._flat[d["lo"]:d["hi"]], d["grad"], "_shard"))
    if d["prefix"] < n:                                  # the few elements that do not divide: all-reduced, updated on every rank
        parts.append((o._flat[d["prefix"]:], o._gflat[d["prefix"]:], "_tail"))
    return parts, d


class SGD(_FusedBase):
    capture_safe = True         # lr / momentum / weight decay are part of GraphedStep's key; everything else is a device pointer

    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0, weight_decay=0.0, nesterov=False):
        if dampening != 0 or nesterov:
            raise NotImplementedError("dampening / nesterov are not used by the reference configs")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        # Groups with the same (lr, momentum, weight decay) share a launch: the reference's `get_parameters` hands the backbone and the classifier over as two
        # groups with identical hyper-parameters (core/model/finetune.py:59-64) -- two launches per step for one update rule.
        batches = {}                                      # (lr, momentum, wd) -> [items, after, zero_mask, owners whose gradient buffer the launch zeroes]
        for group in self.param_groups:
            lr, mom, wd = group["lr"], group["momentum"], group["weight_decay"]
            whole, rest = self._split(group)
            self._check_unconsumed_shards(rest)
            # every tensor of the group goes into ONE launch (clhip_sgd_step_multi, up to eight per launch): a backbone's flat buffer + the head's
            # weight and bias were three launches per step
            entry = batches.setdefault((lr, mom, wd), [[], [], 0, []])
            items, after = entry[0], entry[1]
            for o in whole:
                require_gpu(o._flat)
                st = self.state[o._params[0]]
                parts, shard = _dp_plan(o)
                if self.zero_grads_in_step and shard is None and len(parts) == 1 and len(items) < 8:
                    # the whole flat gradient buffer is consumed by this launch: it leaves zeroed, and the backbone skips its fill at the next backward
                    entry[2] |= 1 << len(items)
                    entry[3].append(o)                # flagged AFTER the launch that really zeroes it (the per-tensor fallback below does not)
                for flat, gflat, sfx in parts:
                    buf = None
                    if mom != 0:
                        buf = st.get("flat_momentum" + sfx)
                        if buf is None or buf.data_ptr() == 0 or buf.numel() != flat.numel() or buf.device != flat.device:
                            buf = torch.zeros_like(flat)
                            st["flat_momentum" + sfx] = buf
                    items.append((flat, gflat, buf))
                after.append((o, shard))
            for p in rest:
                require_gpu(p)
                st = self.state[p]
                pd = self._dense(p)
                gd = self._dense_like(p, p.grad)
                buf = None
                if mom != 0:
                    buf = st.get("momentum_buffer")
                    if buf is None:
                        buf = torch.zeros(pd.numel(), device=p.device, dtype=torch.float32)
                        st["momentum_buffer"] = buf
                items.append((pd, gd, buf))
                o = _owner_of(p)
                if o is not None:
                    after.append((o, None))
        for (lr, mom, wd), (items, after, zero_mask, zeroed) in batches.items():
            same_dev = len({it[0].device for it in items}) <= 1
            if (len(items) >= 2 or zero_mask) and same_dev:
                for k in range(0, len(items), 8):
                    ops.sgd_step_multi(items[k:k + 8], lr, mom, wd, self.grad_scale, zero_mask if k == 0 else 0)
                for o in zeroed:
                    o._gflat_zeroed = True
            else:
                for pd, gd, buf in items:
                    ops.sgd_step(pd, gd, buf, lr, mom, wd, self.grad_scale)
            for o, shard in after:
                if shard is not None:
                    o._dp_shard = shard
                    shard["reducer"].gather_params(o)
                    o._dp_shard = None
                o.mark_params_modified()
        return None


class Adam(_FusedBase):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            lr, (b1, b2), eps, wd = group["lr"], group["betas"], group["eps"], group["weight_decay"]
            whole, rest = self._split(group)
            self._check_unconsumed_shards(rest)
            items, publish = [], []
            for o in whole:
                parts, shard = _dp_plan(o)
                items += [((o._params[0], sfx), flat, gflat, o) for flat, gflat, sfx in parts]
                if shard is not None:
                    publish.append((o, shard))
            items += [((p, ""), self._dense(p), self._dense_like(p, p.grad), _owner_of(p)) for p in rest]
            for (key, sfx), pd, gd, o in items:
                require_gpu(pd)
                st = self.state[key]
                if "step" + sfx not in st or st["exp_avg" + sfx].numel() != pd.numel():
                    st["step" + sfx] = 0
                    st["exp_avg" + sfx] = torch.zeros(pd.numel(), device=pd.device, dtype=torch.float32)
                    st["exp_avg_sq" + sfx] = torch.zeros(pd.numel(), device=pd.device, dtype=torch.float32)
                st["step" + sfx] += 1
                ops.adam_step(pd, gd, st["exp_avg" + sfx], st["exp_avg_sq" + sfx], lr, b1, b2, eps, wd, self.grad_scale, st["step" + sfx])
                if o is not None:
                    o.mark_params_modified()
            for o, shard in publish:
                o._dp_shard = shard
                shard["reducer"].gather_params(o)
                o._dp_shard = None
                o.mark_params_modified()
        return None
