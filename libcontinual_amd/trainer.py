"""Trainer: task loop -> epoch loop -> batch loop, with the reference's hook order and special cases for
the in-scope methods (core/trainer.py:26-720).

Reproduced exactly: `before_task` -> fresh optimizer + scheduler every task
(:294) -> rehearsal merge by dataset concatenation and reshuffle (:305-322) -> per epoch
`model.train()` on the WHOLE plugin (:575), `init_seed(seed + epoch)` (:584), observe / zero_grad / backward /
step (:602-606), scheduler.step per epoch (:404) -> `after_task` -> trainer-side buffer update by `strategy`
(:410-418) -> `testing_times` evaluations, accuracy table, forgetting, BWT (:457-526); the
`(epoch+1) == inc_epoch` validation quirk (:362); `int(acc * B)` correct-count reconstruction (:645).

Different by design (MI355X-first): no per-step host sync -- loss / accuracy stay on the device as
`ops.Deferred` and are read once per epoch; optimizers named SGD / Adam resolve to the fused libclhip
ones; `n_gpu > 1` is real data parallelism (one process per GPU, launched by torchrun; gradient all-reduce of
the flat buffers over RCCL, see parallel.py) instead of the reference's asserted-off scaffold (:37-40).
"""
import copy
import os
import sys
from time import time

import numpy as np
import torch
from torch.utils.data import DataLoader

from . import ops, parallel
from . import scheduler as sched
from .utils import AverageMeter, compute_bwt, compute_frgt, count_all_parameters, count_parameters, get_instance, init_seed, quiesce_gc

_OBSERVE_DOES_BACKWARD = ("L2P",)      # core/trainer.py:593-596 (subset on the hot path)


_UNIT_GRADS = {}


def _backward(loss):
    """loss.backward() (core/trainer.py:604) with the root gradient taken from a cached device scalar: autograd otherwise materialises
    `ones_like(loss)` with a fill kernel on every step"""
    if loss.is_cuda and loss.dim() == 0 and loss.dtype == torch.float32:
        one = _UNIT_GRADS.get(loss.device)
        if one is None:
            one = _UNIT_GRADS[loss.device] = torch.ones((), device=loss.device, dtype=torch.float32)
            ops.UNIT_GRAD_PTRS.add(one.data_ptr())             # the fused loss nodes skip their "scale by the upstream gradient" launch for it
        torch.autograd.backward(loss, grad_tensors=one)
    else:
        loss.backward()


class GraphedStep:
    """One training step (observe -> zero_grad -> backward -> step) captured into a HIP graph and replayed.

    At 32 images per GPU a ResNet-32 step is ~250 kernels of 3-8 us each and the host cannot enqueue them as fast as the GPU runs them
    (eager 1.76 ms per step, replayed 1.34 ms: tools/graph_step.py, profiles/r02_small_batch_notes.md).  The step is already free of host
    synchronisation (losses and accuracies stay on the device), every buffer it touches has a fixed address (plan workspace, parameter
    and gradient arenas, momentum), so a capture on the compute stream replays it exactly.  Rules:
      * the whole loop runs on a stream of its own (see __init__);
      * the first WARM steps of every (input shapes, learning rates, mode) key run eagerly -- lazy initialisation (kernel attributes,
        workspaces, the autograd graph's buffers) must not happen inside a capture -- and are REAL steps; the capture itself
        executes nothing, its first replay is the next real step;
      * a new key (ragged last batch of an epoch, a scheduler step) captures a new graph; the last few keys are kept;
      * only single-process runs of methods that declare `cuda_graph_safe` (a step without data-dependent host control flow);
      * a replay re-runs the captured LAUNCHES only: the host-side bookkeeping of the plans behind them (plan.hip `lazy_live`, `res_pending`,
        `wt_pending`, `bwd_sums_ready`: which activations a consumer applies lazily, whose BatchNorm-backward sums a dgrad epilogue produced) and
        ops.TeacherPass's clhip_config toggles stay as the capture left them.  That is consistent as long as every replay has the structure of the
        captured step, so everything that changes the structure is part of the graph key: input shapes, optimizer hyper-parameters, train / eval
        mode, and the values of the clhip_config switches that steer a plan's launch sequence (`_PLAN_SWITCHES`) -- flipping one between two
        steps captures a new graph instead of replaying a stale one.  Reading a lazily applied activation between two replays
        (clhip_plan_read_act: debugging hooks) materialises it with a launch of its own and does not disturb the next replay: the replayed
        forward rewrites every buffer it reads."""
    WARM, KEEP = 2, 4
    PROBE = 12                      # auto mode: steps of each kind timed per key before the faster one is kept (see _pick)
    _PLAN_SWITCHES = (b"BN_INPUT", b"BN_INPUT_WT", b"BN_RES_INPUT", b"BN_GRAD", b"BN_GRAD_RES", b"BN_FUSE", b"BRANCH_STREAM", b"WGRAD_STREAM", b"CONV6_PAIR",
                      b"STAGE_EVAL", b"EVAL_LAZY", b"STAGE_TRAIN", b"STAGE_TRAIN_BWD", b"STAGE_ENTRY", b"STAGE_POOL", b"STAGE_XCH3", b"GEMM8", b"GEMM_TAIL", b"ATTN_BWD")

    def __init__(self, model, optimizer, method_name, reducer=None):
        self.model, self.optimizer, self.method_name = model, optimizer, method_name
        # data parallel (round 5): the gradient exchange is PART of the captured step -- backward, the RCCL all-reduce(s)
        # of the flat gradient buffers (the early tail fired from inside the backward included) and the fused optimizer replay as one graph.
        # RCCL collectives are capturable; torch's ProcessGroupNCCL records the fork / join between the step's stream and the collective's
        # as captured event dependencies, and Work.wait() is a stream wait, not a host wait.  See _graph_mode for when this is legal.
        self.reducer = reducer
        # the backbones whose flat parameter buffers the captured optimizer launches rewrite: a replay runs those launches on the
        # device only, so the host-side "weights changed" mark (HipResNet.mark_params_modified, normally set by optimizer.step())
        # has to be set here after every replay -- otherwise the next eager forward (validation, after_task) would skip the
        # weight-preparation kernel and read bf16 / re-arranged copies that are one optimizer step stale
        from .optim import _owner_of
        owners = {}
        for group in optimizer.param_groups:
            for p in group["params"]:
                o = _owner_of(p)
                if o is not None:
                    owners[id(o)] = o
        self.owners = list(owners.values())
        self.graphs = {}            # key -> (graph, static inputs, (output, acc, loss))
        self.seen = {}              # key -> eager steps so far
        # every step of a graphed loop -- warm-up, capture, replay -- runs on this stream: autograd binds a parameter's gradient
        # accumulation to the stream of its first backward, and a capture cannot depend on the legacy default stream
        self.stream = torch.cuda.Stream()
        self.choice = {}            # key -> "replay" | "eager": what the probe of this key measured to be faster (auto mode only)
        self._probe = {}            # key -> probe state while it is being measured
        self.disabled = False       # a capture failed under the auto mode: this loop stays eager
        self.fallback_ok = True     # ... which only the auto mode allows (train_steps sets it)

    def _key(self, batch):
        shapes = tuple((k, tuple(v.shape), str(v.dtype)) for k, v in sorted(batch.items()) if torch.is_tensor(v))
        lrs = tuple((g.get("lr"), g.get("momentum"), g.get("weight_decay")) for g in self.optimizer.param_groups)
        from . import _lib
        L = _lib.lib()
        cfg = tuple(L.clhip_config_get(k) for k in self._PLAN_SWITCHES)
        red = self.reducer
        dp = None if red is None else (red.world, red.exchange, getattr(self.optimizer, "grad_scale", None))
        return shapes, lrs, self.model.training, cfg, dp

    def _step(self, batch):
        if self.method_name in _OBSERVE_DOES_BACKWARD:
            self.optimizer.zero_grad()
            out = self.model.observe(batch)
        else:
            out = self.model.observe(batch)
            self.optimizer.zero_grad()
            _backward(out[2])
        if self.reducer is not None:
            self.reducer.reduce(self.model)
        self.optimizer.step()
        return out

    def _pick(self, key):
        """Auto mode only (CLHIP_CUDA_GRAPH unset): replay or eager for THIS step of a captured key?  A replay takes the host out of the step, but HIP's graph launch has a
        per-node cost of its own: the 32-image CifarResNet-32 step replays at 0.74 ms on every box, while the eager enqueue runs 0.67 ms on a box with a fast, idle host and
        0.87-1.03 ms on others (profiles/r05_dp_graph_micro.txt, r05_notes.md).  So the first PROBE steps of a key are replayed and the next PROBE enqueued eagerly, each
        block timed with a pair of events (two stream synchronisations per key, every probe step a real training step: both paths give the same bits), and the faster kind
        is kept -- replay unless eager wins by more than 3 %."""
        c = self.choice.get(key)
        if c is not None:
            return c == "replay"
        st = self._probe.get(key)
        if st is None:
            st = self._probe[key] = dict(kind="replay", n=0, ev=torch.cuda.Event(enable_timing=True), ms={})
            st["ev"].record()
        if st["n"] == self.PROBE:                            # a block is complete: read its time, start the next one or decide
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            end.synchronize()
            st["ms"][st["kind"]] = st["ev"].elapsed_time(end) / self.PROBE
            if st["kind"] == "replay":
                st.update(kind="eager", n=0, ev=torch.cuda.Event(enable_timing=True))
                st["ev"].record()
            else:
                choice = "eager" if st["ms"]["eager"] < 0.97 * st["ms"]["replay"] else "replay"
                if self.reducer is not None and getattr(self.reducer, "world", 1) > 1:
                    # data parallel: every rank times its own probe while its step time depends on the other ranks' -- they must not decide apart (a mixture
                    # still matches collectives one to one, but half the ranks would pay the slower kind for nothing, and a test could not pin it down):
                    # rank 0's measurement decides for all (ranks reach this point in the same step: the probe counts steps, not time)
                    import torch.distributed as dist
                    flag = torch.tensor([1 if choice == "replay" else 0], device=torch.device("cuda", torch.cuda.current_device()), dtype=torch.int32)
                    dist.broadcast(flag, src=0, group=getattr(self.reducer, "group", None))
                    choice = "replay" if int(flag.item()) else "eager"
                self.choice[key] = choice
                self.probe_ms = dict(st["ms"])                # (last probe's figures: diagnostics, tests)
                del self._probe[key]
                if choice == "eager":                         # the graph of a key that stays eager is dead weight (its memory pool included)
                    self.graphs[key] = None
                return choice == "replay"
        st["n"] += 1
        return st["kind"] == "replay"

    def __call__(self, batch):
        if self.disabled:
            return self._step(batch)
        key = self._key(batch)
        if self.choice.get(key) == "eager":
            return self._step(batch)
        ent = self.graphs.get(key)
        if ent is None:
            n = self.seen.get(key, 0)
            if n < self.WARM:
                self.seen[key] = n + 1
                return self._step(batch)
            # static inputs live on the device: a host tensor here would put a pageable host-to-device copy inside the capture
            dev = torch.device("cuda", torch.cuda.current_device())
            static = {k: (v.to(dev, copy=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            try:
                with torch.cuda.graph(g, stream=self.stream, capture_error_mode="thread_local"):
                    out = self._step(static)
            except Exception as e:
                # the step is NOT capturable after all (a wrapper around observe() that reads a loss on the host, a plugin whose class-level
                # `cuda_graph_safe` promise an instance breaks).  Under the default (auto) mode that must not cost the user the run: nothing of
                # the captured step has executed, so drop the capture, stay eager for the rest of this loop's life and say so once.  With
                # CLHIP_CUDA_GRAPH=1 the caller asked for replay explicitly: fail loudly.
                del g
                if not self.fallback_ok:
                    raise
                torch.cuda.synchronize()
                self.disabled = True
                if self.reducer is not None and hasattr(self.reducer, "_early"):
                    self.reducer._rest = {}
                    self.reducer._early.clear()   # (segment hooks that fired inside the dropped capture left offsets and Work objects of an invalidated capture)
                for o in self.owners:             # launches "made" during the dropped capture never ran: the weight copies they were to refresh are stale
                    o.mark_params_modified()
                import warnings
                warnings.warn(f"libcontinual_amd: the training step of {self.method_name or type(self.model).__name__} could not be captured into a HIP graph "
                              f"({type(e).__name__}: {str(e).splitlines()[0][:120]}); continuing eagerly (CLHIP_CUDA_GRAPH=0 silences the attempt)")
                return self._step(batch)
            while len(self.graphs) >= self.KEEP:
                self.graphs.pop(next(iter(self.graphs)))
            ent = self.graphs[key] = (g, static, out)
        g, static, out = ent
        if self.fallback_ok and not self._pick(key):
            return self._step(batch)
        for k, v in batch.items():
            if torch.is_tensor(v):
                static[k].copy_(v, non_blocking=True)
        g.replay()
        for o in self.owners:
            o.mark_params_modified()
        output, acc, loss = out
        # the captured outputs are overwritten by the next replay: hand out copies (two tiny device copies, no synchronisation)
        acc = ops.Deferred(acc.tensor.clone(), acc.scale) if isinstance(acc, ops.Deferred) else acc
        return output, acc, loss.detach().clone()


def _graph_mode(model, reducer, device, optimizer=None):
    """CLHIP_CUDA_GRAPH = 1: replay where legal; 0: never; unset: replay where legal AND the batch is small (see below).  Legal = single process, a plugin that declares
    `cuda_graph_safe`, and an optimizer whose step is capture-safe (class attribute `capture_safe`: the fused SGD -- every argument of
    its launches is a device pointer or a value covered by the graph key; the fused Adam passes its step count by value, torch.optim
    fallbacks synchronise or allocate).  Opt-in because it only pays when the HOST is the limit: with an
    otherwise idle host the 32-image ResNet steps are bound by the GPU's own launch cadence (~250 dependent kernels of 3-8 us:
    eager 1.33 ms, replayed 1.41 ms in bench.py), while a loop that shares its process with a busy loader went 1.76 -> 1.34 ms
    (tools/graph_step.py; profiles/r02_small_batch_notes.md)"""
    env = os.environ.get("CLHIP_CUDA_GRAPH")
    legal = (device is not None and torch.device(device).type == "cuda" and _reducer_capturable(reducer) and getattr(model, "grad_reducer", None) is None
             and getattr(model, "cuda_graph_safe", False) and getattr(optimizer, "capture_safe", False))
    if not legal or env == "0":
        return None
    # Round 4: replay is the DEFAULT for small per-GPU batches (<= GRAPH_AUTO_MAX_BATCH rows; CLHIP_CUDA_GRAPH=0 opts out, =1 replays every
    # size).  With ~100 launches left in a 32-image CifarResNet-32 step the replay is never slower than the best eager run and takes the host
    # out of the picture (eager 0.985 ... 1.106 ms box to box and run to run, replayed 0.846-0.849 ms: profiles/r03_step_notes.md); the large
    # batches keep the eager path -- their steps are GPU-bound and use side streams a capture keeps on one stream.
    return "always" if env == "1" else "auto"


def _reducer_capturable(reducer):
    """A data-parallel step replays from a graph when its exchange is made of capturable launches only: the RCCL backend (gloo stages through
    the host), the in-place all-reduce exchange (the sharded one publishes parameters with a second collective behind the optimizer and keeps
    per-shard state the replay cannot re-point), and CLHIP_DP_GRAPH = 1.  Every rank takes the same decision (same env, same backend), so
    either all ranks replay or none does -- a mixture would still match collectives one to one, replayed or not."""
    if reducer is None:
        return True
    # (opt-in, CLHIP_DP_GRAPH=1: the capture of a reduced step has only ever run on a ONE-rank RCCL group -- tests/test_dp_gpu.py -- no multi-GPU node was available to
    #  the builder; until a >= 2-rank capture has been exercised the default data-parallel step is the eager one)
    if os.environ.get("CLHIP_DP_GRAPH", "0") != "1" or getattr(reducer, "exchange", None) != "all_reduce":
        return False
    import torch.distributed as dist
    try:
        return dist.is_initialized() and dist.get_backend(getattr(reducer, "group", None)) == "nccl"
    except Exception:
        return False


GRAPH_AUTO_MAX_BATCH = 64          # (a plugin may raise it for itself: `cuda_graph_auto_max_batch`, e.g. LUCIR, whose batch-256 step is still host-enqueue-bound)


def _batch_rows(batch):
    for v in batch.values():
        if torch.is_tensor(v) and v.dim() >= 1:
            return int(v.shape[0])
    return 0


def train_steps(model, optimizer, batches, reducer=None, method_name="", meter=None, device=None):
    """The per-batch hot path (core/trainer.py:585-612): observe -> zero_grad -> backward -> [grad all-reduce]
    -> step -> meters.  Shared by Trainer._train and bench.py so that the benchmark times exactly what
    training runs.  Loss / accuracy stay on the device (ops.Deferred): no host sync inside the loop.
    Graph-safe methods under the fused SGD replay a captured HIP graph of the step (GraphedStep) -- by default for per-GPU batches of at most
    GRAPH_AUTO_MAX_BATCH rows, for every size with CLHIP_CUDA_GRAPH=1, never with =0 (see _graph_mode)."""
    on_gpu = device is not None and torch.device(device).type == "cuda"
    import contextlib
    overlap = reducer.overlap(model) if (reducer is not None and hasattr(reducer, "overlap")) else contextlib.nullcontext()
    mode = _graph_mode(model, reducer, device, optimizer)
    gs = None
    if mode is not None:
        gs = getattr(model, "_graphed_step", None)
        if gs is None or gs.optimizer is not optimizer or gs.reducer is not reducer:
            gs = model._graphed_step = GraphedStep(model, optimizer, method_name, reducer)
        gs.fallback_ok = mode == "auto"
    caller = torch.cuda.current_stream() if gs is not None else None
    if gs is not None:
        gs.stream.wait_stream(caller)
    # in this loop zero_grad() follows every step() and nothing reads the gradients in between: the fused SGD may hand the flat gradient buffer
    # back zeroed (no fill launch in front of the next backward); single-process only -- the sharded exchange consumes slices
    zero_prev = getattr(optimizer, "zero_grads_in_step", None)
    if zero_prev is not None and reducer is None and os.environ.get("CLHIP_SGD_ZERO", "1") != "0":
        optimizer.zero_grads_in_step = True
    try:
        _train_loop(model, optimizer, batches, reducer, method_name, meter, on_gpu, overlap, mode, gs, caller)
    finally:
        if zero_prev is not None:
            optimizer.zero_grads_in_step = zero_prev
    if gs is not None:
        caller.wait_stream(gs.stream)


def _train_loop(model, optimizer, batches, reducer, method_name, meter, on_gpu, overlap, mode, gs, caller):
    import contextlib
    with ops.deferred_metrics(on_gpu), overlap, (torch.cuda.stream(gs.stream) if gs is not None else contextlib.nullcontext()):
        for b, batch in enumerate(batches):
            batch["batch_id"] = b
            if gs is not None and (mode == "always" or _batch_rows(batch) <= getattr(model, "cuda_graph_auto_max_batch", GRAPH_AUTO_MAX_BATCH)):
                gs.stream.wait_stream(caller)                  # loaders that produce their batches on the caller's stream
                output, acc, loss = gs({k: v for k, v in batch.items() if k != "batch_id"})
            else:
                if gs is not None:
                    gs.stream.wait_stream(caller)
                if method_name in _OBSERVE_DOES_BACKWARD:
                    optimizer.zero_grad()
                    output, acc, loss = model.observe(batch)
                else:
                    output, acc, loss = model.observe(batch)
                    optimizer.zero_grad()
                    _backward(loss)
                if reducer is not None and getattr(model, "grad_reducer", None) is None:
                    reducer.reduce(model)
                optimizer.step()
            if meter is not None:
                meter.update("acc1", 100 * acc)
                meter.update("loss", loss.detach() if on_gpu else loss.item())


class Trainer:
    def __init__(self, rank, config, model_namespace=None, optim_namespace=None, dataloaders=None, log=print):
        self.rank = rank
        self.config = config
        self.log = log if rank == 0 else (lambda *a, **k: None)
        if model_namespace is None:
            from . import model as model_namespace
        if optim_namespace is None:
            from . import optim as optim_namespace
        self.arch, self.optim_ns = model_namespace, optim_namespace
        self.distribute = config["n_gpu"] > 1
        self.device = self._init_device(config)
        if self.distribute:
            self.rank, self.world = parallel.init_distributed(self.device.type == "cuda")
        else:
            self.world = 1
        self.reducer = parallel.GradientReducer(exchange=config.get("dp_exchange")) if self.distribute else None
        self.init_cls_num, self.inc_cls_num, self.task_num = config["init_cls_num"], config["inc_cls_num"], config["task_num"]
        self.model = self._init_model(config)
        if dataloaders is not None:
            self.train_loader, self.test_loader = dataloaders
        else:
            self.train_loader, self.test_loader = self._init_dataloader(config)
        self.buffer = get_instance(self.arch, "buffer", config)
        self.task_idx = 0
        self.init_epoch, self.inc_epoch, self.optimizer, self.scheduler = self._init_optim(config)
        self.train_meter = AverageMeter("train", ["batch_time", "data_time", "calc_time", "loss", "acc1"])
        self.test_meter = AverageMeter("test", ["batch_time", "data_time", "calc_time", "acc1"])
        self.val_per_epoch = config["val_per_epoch"]
        self.hook_trace = []          # (event, task, epoch) -- used by the parity tests of the call order

    # ------------------------------------------------------------------------------------ set-up
    def _init_device(self, config):
        init_seed(config["seed"], config["deterministic"])
        if config.get("device") == "cpu" or not torch.cuda.is_available():
            if config.get("device") != "cpu":
                raise RuntimeError("no HIP device visible; libcontinual_amd has no CPU path (set device: cpu only for "
                                   "host-logic tests with your own CPU plugin)")
            return torch.device("cpu")
        ids = config.get("device_ids", "auto")
        local = int(os.environ.get("LOCAL_RANK", self.rank))
        if os.environ.get("CLHIP_SHARED_GPU"):          # test hook (tests/test_dp_two_ranks_gpu.py): every rank on cuda:0, exchange through gloo
            local, ids = 0, "auto"
        if isinstance(ids, (list, tuple)):
            idx = ids[local]
        elif isinstance(ids, int) and not self.distribute:
            idx = ids
        else:
            idx = local
        dev = torch.device(f"cuda:{idx}")
        torch.cuda.set_device(dev)
        return dev

    def _init_model(self, config):
        try:
            backbone = get_instance(self.arch, "backbone", config, **{"device": self.device})
        except TypeError:
            backbone = get_instance(self.arch, "backbone", config)
        model = get_instance(self.arch, "classifier", config, **{"device": self.device, "backbone": backbone}).to(self.device)
        if self.distribute:
            parallel.broadcast_module_state(model)
        return model

    def _init_dataloader(self, config):
        from .data import get_dataloader
        train = get_dataloader(config, "train", device=self.device)
        test = get_dataloader(config, "test", cls_map=train.cls_map, device=self.device)
        return train, test

    def _shard_loader(self, loader):
        """per-rank loader: DistributedSampler(shuffle) + batch_size // n_gpu (core/trainer.py:229-241)"""
        if not self.distribute:
            return loader
        if hasattr(loader, "shard"):                                  # GPU input pipeline: rank slice of every permutation
            return loader.shard(self.rank, self.world)
        sampler = torch.utils.data.distributed.DistributedSampler(loader.dataset, num_replicas=self.world, rank=self.rank, shuffle=True)
        return DataLoader(loader.dataset, sampler=sampler, batch_size=max(1, loader.batch_size // self.world),
                          num_workers=loader.num_workers, drop_last=loader.drop_last)

    def _init_optim(self, config):
        init_epoch = config["init_epoch"] if "init_epoch" in config else config["epoch"]
        key = "init_optimizer" if (self.task_idx == 0 and "init_optimizer" in config) else "optimizer"
        ns = self.optim_ns if hasattr(self.optim_ns, config[key]["name"]) else torch.optim
        optimizer = get_instance(ns, key, config, params=self.model.get_parameters(config))
        parallel.attach(self.model, optimizer, self.reducer)
        name = config["lr_scheduler"]["name"]
        kw = config["lr_scheduler"].get("kwargs") or {}
        if name == "CosineSchedule":
            scheduler = sched.CosineSchedule(optimizer, K=kw["K"])
        elif name == "PatienceSchedule":
            scheduler = sched.PatienceSchedule(optimizer, patience=kw["patience"], factor=kw["factor"])
        elif name == "Constant":
            scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda e: 1)
        elif name == "CosineAnnealingWarmUp":
            t_max = len(self.train_loader.get_loader(self.task_idx)) * (init_epoch if self.task_idx == 0 else config["epoch"])
            scheduler = sched.CosineAnnealingWarmUp(optimizer, kw["warmup_length"], t_max)
        else:
            scheduler = get_instance(torch.optim.lr_scheduler, "lr_scheduler", config, optimizer=optimizer)
        return init_epoch, config["epoch"], optimizer, scheduler

    # ---------------------------------------------------------------------------------- main loop
    def train_loop(self):
        t_begin = time()
        method_name = self.config["classifier"]["name"]
        testing_times = self.config["testing_times"]
        T = self.task_num
        batch_last_acc_list, task_last_acc_list = np.zeros(T), np.zeros(T)
        best_batch_last_acc_list, best_task_last_acc_list = np.zeros(T), np.zeros(T)
        acc_table = np.zeros((T, T))
        bwt_list, frgt_list = [], []
        model = self.model
        for task_idx in range(T):
            self.task_idx = task_idx
            self.log(f"================Task {task_idx} Start!================")
            if hasattr(model, "before_task"):
                self.hook_trace.append(("before_task", task_idx, -1))
                model.before_task(task_idx, self.buffer, self.train_loader.get_loader(task_idx), self.test_loader.get_loader(task_idx))
            self.log(f"Trainable Parameters for Task {task_idx} : {count_parameters(model)} / {count_all_parameters(model)}")
            _, _, self.optimizer, self.scheduler = self._init_optim(self.config)
            dataloader = self.train_loader.get_loader(task_idx)
            val_bias_dataloader = None
            if method_name == "bic":
                # hard-wired stage-1 recipe and the plugin's own train / validation split (core/trainer.py:297-303)
                sgd = getattr(self.optim_ns, "SGD", torch.optim.SGD)
                self.optimizer = sgd(model.get_parameters(self.config), lr=0.1, momentum=0.9, weight_decay=2e-4 * self.task_num / (task_idx + 1))
                parallel.attach(self.model, self.optimizer, self.reducer)
                self.scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, milestones=[100, 150, 200], gamma=0.1)
                dataloader, val_bias_dataloader = model.spilt_and_update(dataloader, self.buffer, task_idx, self.config)
            quiesce_gc()                                  # model, optimizer and loaders of this task are built: keep them out of later GC passes
            from .model.buffer import LinearBuffer, LinearHerdingBuffer
            if isinstance(self.buffer, (LinearBuffer, LinearHerdingBuffer)) and self.buffer.buffer_size > 0 and task_idx > 0:
                ds = dataloader.dataset                               # rehearsal union at dataset level (:305-322)
                if isinstance(ds.images, list):
                    ds.images.extend(self.buffer.images)
                    ds.labels.extend(self.buffer.labels)
                elif isinstance(ds.images, np.ndarray):
                    ds.images = np.concatenate((ds.images, self.buffer.images), axis=0)
                    ds.labels = np.concatenate((ds.labels, self.buffer.labels), axis=0)
                else:
                    assert 0
                from .data import make_loader
                dataloader = make_loader(ds, self.config["batch_size"], True, self.config["num_workers"],
                                         self.device if self.config.get("gpu_input_pipeline", True) else None)
            dataloader = self._shard_loader(dataloader)
            self.log(f"================Task {task_idx} Training!================")
            self.log(f"The training samples number : {len(dataloader.dataset)}")
            best_batch_last_acc, best_task_last_acc = 0.0, 0.0
            best_bwt, best_frgt = float("-inf"), float("inf")
            n_epoch = self.init_epoch if task_idx == 0 else self.inc_epoch
            for epoch_idx in range(n_epoch):
                t0 = time()
                sampler = getattr(dataloader, "sampler", None)          # the GPU batch loader has none: its shard is cut in __iter__
                if self.distribute and sampler is not None and hasattr(sampler, "set_epoch"):
                    sampler.set_epoch(epoch_idx)
                meter = self._train(epoch_idx, dataloader)
                acc1, loss = meter.avg("acc1"), meter.avg("loss")
                if self.distribute:
                    acc1 = self.reducer.mean_scalar(acc1, self.device)
                    loss = self.reducer.mean_scalar(loss, self.device)
                if self.device.type == "cuda":
                    torch.cuda.synchronize()
                dt = time() - t0
                n_img = len(dataloader.dataset)
                self.log(f"Epoch [{epoch_idx}/{n_epoch}] Learning Rate {self.scheduler.get_last_lr()}\t|\tLoss: {loss:.4f} "
                         f"\tAverage Acc: {acc1:.2f} \t{n_img / dt:.0f} img/s")
                self.last_epoch_stats = dict(task=task_idx, epoch=epoch_idx, loss=loss, acc1=acc1, images_per_sec=n_img / dt)
                if (epoch_idx + 1) % self.val_per_epoch == 0 or (epoch_idx + 1) == self.inc_epoch:
                    test_acc = self._validate(task_idx)
                    batch_last_acc, per_task_acc = test_acc["avg_acc"], test_acc["per_task_acc"]
                    best_batch_last_acc = max(batch_last_acc, best_batch_last_acc)
                    task_last_acc = np.mean(per_task_acc)
                    best_task_last_acc = max(task_last_acc, best_task_last_acc)
                    frgt, bwt = compute_frgt(acc_table, per_task_acc, task_idx), compute_bwt(acc_table, per_task_acc, task_idx)
                    best_frgt, best_bwt = min(frgt, best_frgt), max(bwt, best_bwt)
                    self.log(f" * [Batch] Last Average Acc: {batch_last_acc:.2f} (Best: {best_batch_last_acc:.2f})")
                    self.log(f" * Per-Task Acc: {per_task_acc}")
                if self.config["lr_scheduler"]["name"] == "PatienceSchedule":
                    self.scheduler.step(loss)               # the loss averaged over the ranks: every replica cuts the lr / stops at the same epoch
                    if self.scheduler.get_last_lr() < self.config["lr_scheduler"]["kwargs"]["stopping_lr"]:
                        break
                else:
                    self.scheduler.step()
            if self.distribute:
                # parameters are identical on every rank (same reduced gradients); BatchNorm running statistics are not (per-rank
                # batches).  Before anything is DERIVED from the model -- herding picks, class means, Fisher, evaluation -- every
                # rank takes rank 0's buffers (what DDP's broadcast_buffers does on every forward), so the replicas stay one model
                parallel.broadcast_module_state(model)
            # a stage-level training launch whose workgroups were not all resident leaves wrong numbers and a sticky error word, not a hung GPU: say so here
            for mod in getattr(model, "modules", lambda: [])():
                if hasattr(mod, "stage_status") and mod.stage_status():
                    raise RuntimeError("libcontinual_amd: a stage-level training launch (csrc/stage_train.hip) ran out of its bounded in-launch wait during task "
                                       f"{task_idx}: its workgroups were not all resident (several processes on one GPU, or two training passes on two streams). "
                                       "Set CLHIP_STAGE_TRAIN=0 for such runs.")
            if hasattr(model, "after_task"):
                self.hook_trace.append(("after_task", task_idx, -1))
                model.after_task(task_idx, self.buffer, self.train_loader.get_loader(task_idx), self.test_loader.get_loader(task_idx))
            # trainer-side buffer update (:410-418); BiC re-cuts its split buffer itself in `spilt_and_update`
            if method_name != "bic":
                self.buffer.total_classes += self.init_cls_num if task_idx == 0 else self.inc_cls_num
            if method_name != "bic" and self.buffer.buffer_size > 0:
                from .model.buffer import herding_update, random_update
                if self.buffer.strategy == "herding":
                    herding_update(self.train_loader.get_loader(task_idx).dataset, self.buffer, model.backbone, self.device)
                elif self.buffer.strategy == "random":
                    random_update(self.train_loader.get_loader(task_idx).dataset, self.buffer)
            if method_name == "bic" and task_idx > 0:
                # stage 2 (core/trainer.py:420-455): the current task's bias layer on the validation split, the rest frozen
                for epoch_idx in range(self.config["stage2_epoch"]):
                    meter = self.stage2_train(epoch_idx, val_bias_dataloader)
                    self.log(f"Epoch [{epoch_idx}/{self.config['stage2_epoch']}] (stage2)\t|\tLoss: {meter.avg('loss'):.4f} \tAverage Acc: {meter.avg('acc1'):.2f} ")
                    if (epoch_idx + 1) % self.val_per_epoch == 0 or (epoch_idx + 1) == self.inc_epoch:
                        test_acc = self._validate(task_idx)
                        batch_last_acc, per_task_acc = test_acc["avg_acc"], test_acc["per_task_acc"]
                        best_batch_last_acc = max(batch_last_acc, best_batch_last_acc)
                        best_task_last_acc = max(np.mean(per_task_acc), best_task_last_acc)
                        self.log(f" * [Batch] Last Average Acc: {batch_last_acc:.2f} (Best: {best_batch_last_acc:.2f})")
            for test_idx in range(testing_times):
                test_acc = self._validate(task_idx)
                batch_last_acc, per_task_acc = test_acc["avg_acc"], test_acc["per_task_acc"]
                best_batch_last_acc = max(batch_last_acc, best_batch_last_acc)
                task_last_acc = np.mean(per_task_acc)
                best_task_last_acc = max(task_last_acc, best_task_last_acc)
                batch_last_acc_list[task_idx] += batch_last_acc
                task_last_acc_list[task_idx] += task_last_acc
                acc_table[task_idx][: task_idx + 1] += np.array(per_task_acc)
            best_batch_last_acc_list[task_idx] = best_batch_last_acc
            best_task_last_acc_list[task_idx] = best_task_last_acc
            batch_last_acc_list[task_idx] /= testing_times
            task_last_acc_list[task_idx] /= testing_times
            acc_table[task_idx] /= testing_times
            batch_last_acc, task_last_acc = batch_last_acc_list[task_idx], task_last_acc_list[task_idx]
            frgt, bwt = compute_frgt(acc_table, acc_table[task_idx], task_idx), compute_bwt(acc_table, acc_table[task_idx], task_idx)
            if task_idx > 1:
                frgt_list.append(frgt)
                bwt_list.append(bwt)
            self.log(f"================Result of Task {task_idx} Testing!================")
            self.log(f" * [Batch] Last Average Acc: {batch_last_acc:.2f} (Best: {best_batch_last_acc:.2f})")
            self.log(f" * [Task] Last Average Acc: {task_last_acc:.2f} (Best: {best_task_last_acc:.2f})")
            self.log(f" * Forgetting: {frgt:.3f}  Backward Transfer: {bwt:.2f}")
            self.log(f" * Per-Task Acc: {acc_table[task_idx][:task_idx + 1]}")
        result = dict(
            batch_last_acc=float(batch_last_acc), task_last_acc=float(task_last_acc),
            batch_ovr_avg_acc=float(np.mean(batch_last_acc_list)),
            task_ovr_avg_acc=float(np.sum(np.sum(acc_table[: task_idx + 1], axis=1) / np.arange(1, task_idx + 2)) / (task_idx + 1)),
            ovr_bwt=float(np.mean(bwt_list)) if bwt_list else float("-inf"),
            ovr_frgt=float(np.mean(frgt_list)) if frgt_list else float("inf"),
            acc_table=acc_table, time=time() - t_begin)
        self.log(f"================Overall Result of {self.task_num} Tasks!================")
        self.log(f" * [Batch] Last Average Acc: {result['batch_last_acc']:.2f}   Overall Avg Acc: {result['batch_ovr_avg_acc']:.2f}")
        self.log(f" * Time Costs : {result['time']:.2f} sec")
        return result

    def _train(self, epoch_idx, dataloader):
        """the hot loop (core/trainer.py:563-614)"""
        model = self.model
        model.train()                      # whole plugin, teachers included (quirk a10/a11/a12)
        meter = copy.deepcopy(self.train_meter)
        meter.reset()
        init_seed(self.config["seed"] + epoch_idx, self.config["deterministic"])
        self.hook_trace.append(("train_epoch", self.task_idx, epoch_idx))
        train_steps(model, self.optimizer, dataloader, self.reducer, self.config["classifier"]["name"], meter, self.device)
        return meter

    def stage2_train(self, epoch_idx, dataloader):
        """BiC's second stage (core/trainer.py:534-561): everything in eval mode, `model.stage2` steps its own optimizer"""
        model = self.model
        model.eval()
        for layer in model.bias_layers:
            layer.train()
        meter = self.train_meter
        meter.reset()
        self.hook_trace.append(("stage2_epoch", self.task_idx, epoch_idx))
        on_gpu = self.device.type == "cuda"
        with ops.deferred_metrics(on_gpu):
            for batch in dataloader:
                output, acc, loss = model.stage2(batch)
                meter.update("acc1", 100 * acc)
                meter.update("loss", loss.detach() if on_gpu else loss.item())
        return meter

    def _validate(self, task_idx):
        """core/trainer.py:616-720 (testing_per_task branch and merged branch)"""
        dataloaders = self.test_loader.get_loader(task_idx)
        model = self.model
        model.eval()
        self.hook_trace.append(("validate", task_idx, -1))
        per_task_acc, count_all, correct_all = [], 0, 0
        with torch.no_grad():
            if self.config["testing_per_task"]:
                for t, dl in enumerate(dataloaders):
                    correct_task, count_task = 0, 0
                    for batch in dl:
                        if self.config["setting"] == "task-aware":
                            output, acc = model.inference(batch, task_id=t)
                        else:
                            output, acc = model.inference(batch)
                        correct_task += int(acc * batch["label"].shape[0])
                        count_task += batch["label"].shape[0]
                    correct_all += correct_task
                    count_all += count_task
                    per_task_acc.append(round(correct_task * 100 / count_task, 2))
            else:
                datasets = [dl.dataset for dl in dataloaders]
                merged = copy.deepcopy(datasets[0])
                merged.images = np.concatenate([ds.images for ds in datasets], axis=0)
                merged.labels = np.concatenate([ds.labels for ds in datasets], axis=0)
                from .data import make_loader
                loader = make_loader(merged, self.config["batch_size"], True, self.config["num_workers"],
                                     self.device if self.config.get("gpu_input_pipeline", True) else None)
                bounds, s = [], 0
                for t in range(task_idx + 1):
                    n = self.init_cls_num if t == 0 else self.inc_cls_num
                    bounds.append((s, s + n))
                    s += n
                cb, nb = np.zeros(task_idx + 1, dtype=int), np.zeros(task_idx + 1, dtype=int)
                for batch in loader:
                    output, acc = model.inference(batch)
                    preds, labels = output.cpu().numpy(), batch["label"].cpu().numpy()
                    correct_all += int(np.sum(preds == labels))
                    count_all += len(labels)
                    for t, (a, e) in enumerate(bounds):
                        m = (labels >= a) & (labels < e)
                        if np.any(m):
                            cb[t] += np.sum(preds[m] == labels[m])
                            nb[t] += np.sum(m)
                per_task_acc = [round(c * 100 / n, 2) if n > 0 else 0 for c, n in zip(cb, nb)]
        return {"avg_acc": round(correct_all * 100 / count_all, 2), "per_task_acc": per_task_acc}
