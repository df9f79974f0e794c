"""Data-parallel training: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm, "gloo" for CPU tests).  The reference only carries a dead DDP scaffold (core/trainer.py:37-40,
206-210, 229-241: `assert not self.distribute`); single-GPU semantics are the oracle.

Design for MI355X: the gradients of a whole backbone are ONE flat fp32 buffer, so the
exchange is a single all-reduce of one bucket (ResNet-32 1.9 MB, ResNet-18 44.7 MB) plus one tiny bucket for
the head; the mean (1/world) is folded into the fused optimizer step (`grad_scale`) instead of a separate
pass.  Terms that are identical on every rank (the EWC penalty gradient) are added before the all-reduce and
therefore come out exact after the 1/world scaling.  BatchNorm uses per-rank batch statistics (DDP-faithful).
Fisher accumulation and herding stay on every rank's full copy of the data (single-GPU semantics).

Second exchange (`GradientReducer(exchange="reduce_scatter")` or CLHIP_DP_EXCHANGE=reduce_scatter): the flat
gradient buffer is reduce-scattered, every rank runs the fused optimizer on ITS 1/world shard of the flat parameter buffer only
(momentum / Adam moments exist for that shard only), and the updated shards are all-gathered in place over the flat parameter buffer.
Same bytes on the links as the ring all-reduce, 1/world of the optimizer traffic and state; it gives up the overlap of the exchange
with the backward, so the all-reduce stays the default and the checked fallback.  The < 4 * world elements that do not divide are
all-reduced and updated on every rank.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(device_is_cuda=True):
    """Initialise from torchrun-style env (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        # this pool's host driver only supports dmabuf IPC: without it RCCL's peer setup fails with hipIpcGetMemHandle errors
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # CLHIP_DIST_BACKEND=gloo: test hook -- several ranks sharing ONE GPU (RCCL refuses duplicate devices; gloo stages
        # device tensors through the host), used by tests/test_dp_two_ranks_gpu.py to run the real N>1 code path on a 1-GPU box
        backend = os.environ.get("CLHIP_DIST_BACKEND") or ("nccl" if device_is_cuda else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def _flat_grad_buckets(module):
    """[(tensor to all-reduce)] : every HipResNet flat gradient buffer once + one packed bucket for the rest"""
    from .model.backbone.resnet import HipResNet
    buckets, covered = [], set()
    for m in module.modules():
        if isinstance(m, HipResNet) and m._gflat is not None and m._params[0].requires_grad and m._params[0].grad is not None:
            buckets.append(m._gflat)
            covered.update(id(p) for p in m._params)
    rest = [p for p in module.parameters() if id(p) not in covered and p.requires_grad and p.grad is not None]
    return buckets, rest


class GradientReducer:
    """call `reduce(module)` between loss.backward() and optimizer.step(); set optimizer.grad_scale = 1/world."""

    def __init__(self, group=None, exchange=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.exchange = exchange or os.environ.get("CLHIP_DP_EXCHANGE", "all_reduce")
        if self.exchange not in ("all_reduce", "reduce_scatter"):
            raise ValueError(f"unknown data-parallel exchange {self.exchange!r}")
        self._early = {}          # id(backbone) -> (lowest element offset already handed to an async all-reduce, [works])
        self._rest = {}           # the flat bucket of the tensors outside the backbones' buffers (_reduce_rest)
        self.optimizer = None     # set by attach(): the sharded exchange needs to know which backbones the optimizer steps as a whole

    # ---- overlap of the exchange with the backward (ResNet-18: 75 % of the parameters sit in layer4, whose gradients are
    #      complete after the first quarter of the backward)
    def overlap(self, module, fraction=0.5):
        """context manager for a training loop: every HipResNet runs its backward in two pieces and the finished tail of its
        flat gradient buffer is all-reduced asynchronously while the rest of the backward runs; `reduce()` then only
        exchanges the remaining head of the buffer and waits.  Outside the context (Fisher passes, herding) nothing fires."""
        red = self

        class _Ctx:
            def __enter__(self_c):
                from .model.backbone.resnet import HipResNet
                on = red.world > 1 and red.exchange == "all_reduce"       # the sharded exchange needs the whole buffer at once
                self_c.mods = [m for m in module.modules() if isinstance(m, HipResNet)] if on else []
                for m in self_c.mods:
                    k = m.grad_cut_for_fraction(fraction)
                    if k > 0:
                        m._grad_segment_cuts, m._grad_segment_hook = [k], red._on_segment
                return red

            def __exit__(self_c, *a):
                for m in self_c.mods:
                    m._grad_segment_cuts, m._grad_segment_hook = [], None
                red._early.clear()
        return _Ctx()

    def _on_segment(self, bb, lo, hi):
        if lo == 0 or not bb._params[0].requires_grad:
            return                                   # the head of the buffer goes with reduce()
        done_lo, works = self._early.get(id(bb), (bb._nflat, []))
        if hi != done_lo:
            return                                   # not the contiguous continuation of what is in flight: leave it to reduce()
        works.append(dist.all_reduce(bb._gflat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._early[id(bb)] = (lo, works)

    def shard_bounds(self, n):
        """(elements per rank, length of the divisible prefix) of a flat buffer of n elements: shards are multiples of 4 elements"""
        per = (n // (4 * self.world)) * 4
        return per, per * self.world

    def _reduce_rest(self, rest):
        """the tensors outside the flat backbone buffers (heads, LoRA matrices, prompts) as ONE bucket: packed by one `torch.cat`, all-reduced, and handed back as the
        gradients themselves (`p.grad = view of the bucket`) instead of one `copy_` per tensor (ViT-B/16 InfLoRA: 24 LoRA matrices + the head were 25 small
        kernels per step behind the collective).  Where a loop keeps its gradients alive and accumulates into them in place (zero_grad(set_to_none=False)), the
        views survive and the bucket goes to the collective as it is, without the `torch.cat` either."""
        if not rest:
            return
        st = self._rest
        key = tuple(id(p) for p in rest)
        flat = st.get("flat")
        ok = flat is not None and st.get("key") == key and flat.device == rest[0].grad.device
        if ok:
            base, off = flat.data_ptr(), 0
            for p in rest:
                g = p.grad
                if g.data_ptr() != base + off * flat.element_size() or g.dtype != flat.dtype or not g.is_contiguous():
                    ok = False
                    break
                off += g.numel()
        if not ok:
            same = all(p.grad.dtype == rest[0].grad.dtype for p in rest)
            flat = torch.cat([p.grad.reshape(-1).to(rest[0].grad.dtype) for p in rest])
            if same:
                off = 0
                for p in rest:
                    n = p.grad.numel()
                    p.grad = flat[off:off + n].view_as(p.grad)          # (no reference to the view is kept: autograd adds in place only into a gradient nobody else holds)
                    off += n
                self._rest = dict(flat=flat, key=key)
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                return
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)      # mixed dtypes: the copying form
            off = 0
            for p in rest:
                n = p.grad.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
            return
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)

    def _reduce_scatter(self, module):
        """sum over ranks of every flat gradient buffer, delivered as this rank's shard: `bb._dp_shard` tells the fused optimizer
        which slice of the flat PARAMETER buffer to update and how to publish it (gather_params)"""
        from .model.backbone.resnet import HipResNet
        buckets, rest = _flat_grad_buckets(module)
        owners = {m._gflat.data_ptr(): m for m in module.modules() if isinstance(m, HipResNet) and m._gflat is not None}
        # only a backbone the attached optimizer updates with one launch over its flat buffer consumes `_dp_shard`; anything else
        # (torch.optim fallback, a backbone split across param groups or partly frozen) would step on its LOCAL gradients: those
        # buckets take the in-place all-reduce instead
        whole = getattr(self.optimizer, "whole_backbones", None)
        sharded_ok = {id(o) for o in whole()} if whole is not None else set()
        works = []
        for b in buckets:
            bb = owners[b.data_ptr()]
            if id(bb) not in sharded_ok:
                works.append(dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                continue
            n = b.numel()
            per, prefix = self.shard_bounds(n)
            shard = getattr(bb, "_dp_gshard", None)
            if shard is None or shard.numel() != per or shard.device != b.device:
                shard = bb._dp_gshard = torch.empty(per, dtype=b.dtype, device=b.device)
            if per > 0:
                works.append(dist.reduce_scatter_tensor(shard, b[:prefix], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            if prefix < n:
                works.append(dist.all_reduce(b[prefix:], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            bb._dp_shard = dict(lo=self.rank * per, hi=(self.rank + 1) * per, prefix=prefix, grad=shard, reducer=self)
        self._reduce_rest(rest)
        for w in works:
            w.wait()

    def gather_params(self, bb):
        """publish this rank's updated shard of the flat parameter buffer to every rank (in place)"""
        d = bb._dp_shard
        flat = bb._flat
        if d["hi"] > d["lo"]:
            dist.all_gather_into_tensor(flat[:d["prefix"]], flat[d["lo"]:d["hi"]], group=self.group)

    def reduce(self, module, full=False):
        """`full`: the caller reads the whole summed gradient afterwards (plugins that clip it): always the all-reduce"""
        if self.world == 1:
            return
        if self.exchange == "reduce_scatter" and not full:
            return self._reduce_scatter(module)
        buckets, rest = _flat_grad_buckets(module)
        works = []
        from .model.backbone.resnet import HipResNet
        owners = {m._gflat.data_ptr(): m for m in module.modules() if isinstance(m, HipResNet) and m._gflat is not None}
        for b in buckets:
            bb = owners.get(b.data_ptr())
            done_lo, early = self._early.pop(id(bb), (b.numel(), [])) if bb is not None else (b.numel(), [])
            works += early
            if done_lo > 0:
                works.append(dist.all_reduce(b[:done_lo], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._reduce_rest(rest)
        for w in works:
            w.wait()

    def reduce_mean(self, module):
        """all-reduce AND divide by the world size in place: for plugins that post-process the averaged gradient inside
        `observe` (L2P clips its norm there, l2p.py:103-104 -- the clip must see the reduced gradient)"""
        if self.world == 1:
            return
        self.reduce(module, full=True)
        s = 1.0 / self.world
        for p in module.parameters():
            if p.requires_grad and p.grad is not None:
                if p.grad.is_cuda:
                    from . import ops
                    ops.scale_(p.grad, s)
                else:
                    p.grad.mul_(s)

    def mean_scalar(self, value, device):
        """average a python float over ranks (epoch loss/acc, core/trainer.py:347-354)"""
        if self.world == 1:
            return value
        t = torch.tensor([value], device=device, dtype=torch.float64)
        dist.all_reduce(t, group=self.group)
        return float(t.item()) / self.world


def attach(model, optimizer, reducer):
    """Wire data parallelism into a (plugin, optimizer) pair.  Plugins that run backward inside `observe` and then touch the
    gradient (class attribute `reduces_own_gradients`) get the reducer and average the gradient themselves; for all others
    the trainer reduces after backward and the 1/world factor is folded into the fused optimizer step.  Returns True when
    the plugin owns the reduction."""
    own = reducer is not None and bool(getattr(model, "reduces_own_gradients", False))
    if reducer is not None:
        reducer.optimizer = optimizer
        # the hardware-queue cap the multi-stream steps were measured under: say so (once) if this process started the HIP runtime without it
        from . import hw_queue_cap_state
        reducer.hw_queue_cap = hw_queue_cap_state()
        if reducer.world > 1 and os.environ.get("CLHIP_BRANCH_STREAM") is None:
            # multi-rank steps already run main + weight-gradient + the collective's stream(s): the plans' shortcut-branch stream (+0.5 % on one GPU)
            # stays off -- a fifth stream in one step is what made the LwF teacher step 2.4x slower (profiles/r03_step_notes.md), and the multi-GPU
            # combination cannot be measured on the 1-GPU boxes this was built on
            from . import _lib
            try:
                _lib.lib().clhip_config(b"BRANCH_STREAM", b"0")
            except Exception:
                pass                                          # CPU-only processes (gloo tests) have no library to steer
        if reducer.world > 1 and os.environ.get("CLHIP_STAGE_TRAIN") is None:
            # the stage-level training launches (stage_train.hip) need every workgroup of a launch resident at once: ranks that SHARE a GPU (tests, oversubscribed
            # nodes) would hold each other's compute units -- such a process keeps the per-unit launches
            try:
                import torch
                local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
                shared = bool(os.environ.get("CLHIP_SHARED_GPU")) or (torch.cuda.is_available() and local_world > torch.cuda.device_count())
                if shared:
                    from . import _lib
                    _lib.lib().clhip_config(b"STAGE_TRAIN", b"0")
            except Exception:
                pass
    if hasattr(model, "grad_reducer") or own:
        model.grad_reducer = reducer if own else None
    if hasattr(optimizer, "grad_scale"):
        optimizer.grad_scale = 1.0 if (own or reducer is None) else 1.0 / reducer.world
    return own


def broadcast_module_state(module, src=0, group=None):
    """make every rank start from rank `src`'s parameters and buffers"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    from .model.backbone.resnet import HipResNet
    covered = set()
    for m in module.modules():
        if isinstance(m, HipResNet):
            flat, _ = m.flat_parameters()
            dist.broadcast(flat, src, group=group)
            dist.broadcast(m._stats, src, group=group)
            m.mark_params_modified()
            covered.update(id(p) for p in m._params)
            covered.update(id(b) for b in m.buffers())
    for t in list(module.parameters()) + list(module.buffers()):
        if id(t) not in covered:
            dist.broadcast(t.data, src, group=group)
