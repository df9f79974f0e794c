"""torch-tensor front end of the C ABI: every function here only extracts device pointers / the current
stream and calls libclhip (no arithmetic in torch).  autograd.Function wrappers keep the reference's
`loss.backward()` contract (core/trainer.py:602-604) for the head / loss kernels.
"""
import torch

from . import _lib
from ._lib import call, require_gpu

_NULL = None
UNIT_GRAD_PTRS = set()      # data pointers of the trainer's cached unit root gradients (trainer._backward)


def _st():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _dev(*ts):
    """all tensors handed to a kernel must live on the same HIP device (a host pointer would fault the GPU)"""
    d = None
    for t in ts:
        if t is None:
            continue
        require_gpu(t)
        if d is None:
            d = t.device
        elif t.device != d:
            raise _lib.ClhipError(f"tensors on different devices: {d} vs {t.device}")


def _f32c(t):
    require_gpu(t)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# ------------------------------------------------------------------------------------------ heads
class _LinearFn(torch.autograd.Function):
    """out = x W^T + b via clhip_linear_fwd / clhip_linear_bwd (nn.Linear heads: ewc.py:50, lwf.py:29-40)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w = _f32c(x), _f32c(w)
        bb = _f32c(b) if b is not None else None
        B, D = x.shape
        O = w.shape[0]
        out = torch.empty(B, O, device=x.device, dtype=torch.float32)
        call("clhip_linear_fwd", _ptr(x), _ptr(w), _ptr(bb), _ptr(out), B, D, O, _st())
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w = ctx.saved_tensors
        dout = _f32c(dout)
        B, D = x.shape
        O = w.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w)
        db = torch.empty(O, device=x.device, dtype=torch.float32) if ctx.has_bias else None
        call("clhip_linear_bwd", _ptr(x), _ptr(w), _ptr(dout), _ptr(dx), _ptr(dw), _ptr(db), B, D, O, 0, _st())
        return dx, dw, db


def linear(x, w, b=None):
    return _LinearFn.apply(x, w, b)


class _CosineLinearFn(torch.autograd.Function):
    """scores = normalize(x) @ normalize(w)^T  (CosineLinear, backbone/resnet.py:436-438)."""

    @staticmethod
    def forward(ctx, x, w):
        x, w = _f32c(x), _f32c(w)
        B, D = x.shape
        O = w.shape[0]
        out = torch.empty(B, O, device=x.device, dtype=torch.float32)
        xn = torch.empty(B, device=x.device, dtype=torch.float32)
        wn = torch.empty(O, device=x.device, dtype=torch.float32)
        call("clhip_cosine_linear_fwd", _ptr(x), _ptr(w), _ptr(out), _ptr(xn), _ptr(wn), B, D, O, _st())
        ctx.save_for_backward(x, w, out, xn, wn)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, out, xn, wn = ctx.saved_tensors
        dout = _f32c(dout)
        B, D = x.shape
        O = w.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        call("clhip_cosine_linear_bwd", _ptr(x), _ptr(w), _ptr(out), _ptr(xn), _ptr(wn), _ptr(dout), _ptr(dx), _ptr(dw),
             B, D, O, 0, _st())
        return dx, dw


def cosine_linear(x, w):
    return _CosineLinearFn.apply(x, w)


class _SigmaScaleFn(torch.autograd.Function):
    """logits = sigma * scores with sigma a 1-element parameter (backbone/resnet.py:439-441)."""

    @staticmethod
    def forward(ctx, scores, sigma):
        scores, sigma = _f32c(scores), _f32c(sigma)
        out = torch.empty_like(scores)
        call("clhip_sigma_scale_fwd", _ptr(scores), _ptr(sigma), _ptr(out), scores.numel(), _st())
        ctx.save_for_backward(scores, sigma)
        return out

    @staticmethod
    def backward(ctx, dout):
        scores, sigma = ctx.saved_tensors
        dout = _f32c(dout)
        ds = torch.empty_like(scores)
        dsig = torch.empty(1, device=scores.device, dtype=torch.float32)
        call("clhip_sigma_scale_bwd", _ptr(scores), _ptr(sigma), _ptr(dout), _ptr(ds), _ptr(dsig), 0, scores.numel(), _st())
        return ds, dsig.view_as(sigma)


def sigma_scale(scores, sigma):
    return _SigmaScaleFn.apply(scores, sigma)


# ----------------------------------------------------------------------------------------- losses
class Deferred:
    """A scalar that still lives on the device: value = tensor * scale.  float() synchronises.  The
    reference reads accuracy / loss with .item() two or three times per step;
    the trainer of this package keeps them as Deferred and resolves a whole epoch at once."""
    __slots__ = ("tensor", "scale")

    def __init__(self, tensor, scale=1.0):
        self.tensor, self.scale = tensor, scale

    def __mul__(self, k):
        return Deferred(self.tensor, self.scale * k)

    __rmul__ = __mul__

    def __truediv__(self, k):
        return Deferred(self.tensor, self.scale / k)

    def __float__(self):
        return float(self.tensor.item()) * self.scale

    def item(self):
        return float(self)


_DEFER = [False]


class deferred_metrics:
    """context: plugin.observe() returns accuracy as a Deferred instead of synchronising"""

    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        self.prev = _DEFER[0]
        _DEFER[0] = self.on

    def __exit__(self, *a):
        _DEFER[0] = self.prev


class LossAux:
    """pred / correct-count produced by the fused CE kernel (no host sync until .acc() is read)."""
    __slots__ = ("pred", "correct", "batch")

    def acc(self):
        if _DEFER[0]:
            return Deferred(self.correct, 1.0 / self.batch)
        return self.correct.item() / self.batch


class _ClassifyLossFn(torch.autograd.Function):
    """weight_ce * CE(logits[:, lo:hi], y - lo)  [+ weight_kd * KD(logits[:, :k], teacher[:, :k], T)]
    Forward computes loss AND d(loss)/d(logits) in the same kernels (ce_slice + kd); backward only
    scales by the upstream gradient on device.  ewc.py:87-100, lwf.py:57-65, icarl.py:197-221."""

    @staticmethod
    def forward(ctx, logits, labels, lo, hi, pred_hi, w_ce, teacher, k, T, w_kd, aux, pred_lo=0):
        logits = _f32c(logits)
        B, O = logits.shape
        labels = labels.to(device=logits.device, dtype=torch.int64).contiguous()
        loss = torch.empty(1, device=logits.device, dtype=torch.float32)
        dlog = torch.empty_like(logits)
        pred = torch.empty(B, device=logits.device, dtype=torch.int64)
        correct = torch.empty(1, device=logits.device, dtype=torch.int32)
        call("clhip_ce_window", _ptr(logits), _ptr(labels), B, O, lo, hi, pred_lo, pred_hi, float(w_ce), _ptr(loss), 0, _ptr(dlog), 0,
             _ptr(pred), _ptr(correct), _st())
        if teacher is not None:
            teacher = _f32c(teacher)
            call("clhip_kd_loss", _ptr(logits), O, _ptr(teacher), teacher.shape[1], B, k, float(T), float(w_kd), _ptr(loss), 1,
                 _ptr(dlog), 1, _st())
        if aux is not None:
            aux.pred, aux.correct, aux.batch = pred, correct, B
        ctx.save_for_backward(dlog)
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        (dlog,) = ctx.saved_tensors
        if gout.is_cuda and gout.data_ptr() in UNIT_GRAD_PTRS:      # the trainer's cached unit root gradient: d(loss)/d(logits) as computed
            return (dlog,) + (None,) * 11
        gout = _f32c(gout.reshape(1))
        out = torch.empty_like(dlog)
        call("clhip_scale_dev", _ptr(dlog), _ptr(out), dlog.numel(), 1.0, _ptr(gout), _st())
        return (out,) + (None,) * 11


def classify_loss(logits, labels, lo=0, hi=None, pred_hi=None, w_ce=1.0, teacher=None, k=0, T=2.0, w_kd=0.0, aux=None, pred_lo=0):
    O = logits.shape[1]
    hi = O if hi is None else hi
    pred_hi = O if pred_hi is None else pred_hi
    return _ClassifyLossFn.apply(logits, labels, lo, hi, pred_hi, w_ce, teacher, k, T, w_kd, aux, pred_lo)


def predict(logits, labels, pred_hi=None):
    """argmax + correct count without a loss (inference paths, ewc.py:135-145)."""
    logits = _f32c(logits)
    B, O = logits.shape
    pred_hi = O if pred_hi is None else pred_hi
    labels = labels.to(device=logits.device, dtype=torch.int64).contiguous()
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    pred = torch.empty(B, device=logits.device, dtype=torch.int64)
    correct = torch.empty(1, device=logits.device, dtype=torch.int32)
    call("clhip_ce_slice", _ptr(logits), _ptr(labels), B, O, 0, pred_hi, pred_hi, 0.0, _ptr(loss), 0, None, 0, _ptr(pred),
         _ptr(correct), _st())
    return pred, correct


class _CosEmbedFn(torch.autograd.Function):
    """weight * mean(1 - cos(a, b.detach()))  (nn.CosineEmbeddingLoss with target 1, lucir.py:182-183)."""

    @staticmethod
    def forward(ctx, a, b, weight):
        a, b = _f32c(a), _f32c(b)
        B, D = a.shape
        loss = torch.empty(1, device=a.device, dtype=torch.float32)
        da = torch.empty_like(a)
        call("clhip_cos_embed_loss", _ptr(a), _ptr(b), B, D, float(weight), _ptr(loss), 0, _ptr(da), 0, _st())
        ctx.save_for_backward(da)
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        (da,) = ctx.saved_tensors
        gout = _f32c(gout.reshape(1))
        out = torch.empty_like(da)
        call("clhip_scale_dev", _ptr(da), _ptr(out), da.numel(), 1.0, _ptr(gout), _st())
        return out, None, None


def cos_embed_loss(a, b, weight=1.0):
    return _CosEmbedFn.apply(a, b, weight)


class _MarginRankFn(torch.autograd.Function):
    """LUCIR hard-negative margin ranking on the pre-sigma scores (lucir.py:187-205)."""

    @staticmethod
    def forward(ctx, scores, labels, num_old, K, margin, weight):
        scores = _f32c(scores)
        B, O = scores.shape
        labels = labels.to(device=scores.device, dtype=torch.int64).contiguous()
        loss = torch.empty(1, device=scores.device, dtype=torch.float32)
        ds = torch.empty_like(scores)
        hard = torch.empty(1, device=scores.device, dtype=torch.int32)
        call("clhip_margin_rank_loss", _ptr(scores), _ptr(labels), B, O, int(num_old), int(K), float(margin), float(weight),
             _ptr(loss), 0, _ptr(ds), 0, _ptr(hard), _st())
        ctx.save_for_backward(ds)
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        (ds,) = ctx.saved_tensors
        gout = _f32c(gout.reshape(1))
        out = torch.empty_like(ds)
        call("clhip_scale_dev", _ptr(ds), _ptr(out), ds.numel(), 1.0, _ptr(gout), _st())
        return out, None, None, None, None, None


def margin_rank_loss(scores, labels, num_old, K, margin, weight=1.0):
    return _MarginRankFn.apply(scores, labels, num_old, K, margin, weight)


# ------------------------------------------------------------------------------- flat-buffer family
def ewc_penalty(p, ref, fisher, weight, loss_out, accumulate):
    _dev(p, ref, fisher, loss_out)
    call("clhip_ewc_penalty", _ptr(p), _ptr(ref), _ptr(fisher), p.numel(), float(weight), _ptr(loss_out), int(accumulate), _st())


def ewc_grad(p, ref, fisher, g, weight, dev_scale=None):
    _dev(p, ref, fisher, g, dev_scale)
    call("clhip_ewc_grad", _ptr(p), _ptr(ref), _ptr(fisher), _ptr(g), p.numel(), float(weight), _ptr(dev_scale), _st())


def _ptr_array(ts):
    import ctypes as C
    return (C.c_void_p * len(ts))(*[(t.data_ptr() if t is not None else None) for t in ts])


def ewc_penalty_multi(segs, weight, loss_out, accumulate):
    """segs: [(p, ref, fisher), ...] (<= 4) -- one launch for all of them"""
    import ctypes as C
    for p, r, f in segs:
        _dev(p, r, f)
    _dev(loss_out)
    n = (C.c_int64 * len(segs))(*[p.numel() for p, _, _ in segs])
    call("clhip_ewc_penalty_multi", len(segs), _ptr_array([s[0] for s in segs]), _ptr_array([s[1] for s in segs]), _ptr_array([s[2] for s in segs]), n,
         float(weight), _ptr(loss_out), int(accumulate), _st())


def ewc_grad_multi(segs, weight, dev_scale=None):
    """segs: [(p, ref, fisher, g), ...] (<= 4): g += weight * F * (p - ref), one launch"""
    import ctypes as C
    for p, r, f, g in segs:
        _dev(p, r, f, g)
    n = (C.c_int64 * len(segs))(*[s[0].numel() for s in segs])
    call("clhip_ewc_grad_multi", len(segs), _ptr_array([s[0] for s in segs]), _ptr_array([s[1] for s in segs]), _ptr_array([s[2] for s in segs]),
         _ptr_array([s[3] for s in segs]), n, float(weight), _ptr(dev_scale), _st())


def fisher_accum(fisher, g, scale):
    _dev(fisher, g)
    call("clhip_fisher_accum", _ptr(fisher), _ptr(g), fisher.numel(), float(scale), _st())


def fisher_merge(new_f, old_f, alpha):
    _dev(new_f, old_f)
    call("clhip_fisher_merge", _ptr(new_f), _ptr(old_f), old_f.numel(), float(alpha), _st())


def sgd_step(p, g, mom, lr, momentum=0.0, weight_decay=0.0, grad_scale=1.0, ewc_ref=None, ewc_fisher=None, ewc_weight=0.0):
    _dev(p, g, mom, ewc_ref, ewc_fisher)
    call("clhip_sgd_step", _ptr(p), _ptr(g), _ptr(mom), p.numel(), float(lr), float(momentum), float(weight_decay),
         float(grad_scale), _ptr(ewc_ref), _ptr(ewc_fisher), float(ewc_weight), _st())


def sgd_step_multi(items, lr, momentum=0.0, weight_decay=0.0, grad_scale=1.0, zero_mask=0):
    """items: [(p, g, mom or None), ...] (<= 8 flat fp32 tensors on one device): clhip_sgd_step on each, ONE launch; bit k of `zero_mask`: the
    gradient of item k is zeroed once consumed"""
    import ctypes as C
    for p, g, m in items:
        _dev(p, g, m)
    n = (C.c_int64 * len(items))(*[it[0].numel() for it in items])
    moms = _ptr_array([it[2] for it in items]) if momentum != 0 else None
    call("clhip_sgd_step_multi_zero", len(items), _ptr_array([it[0] for it in items]), _ptr_array([it[1] for it in items]), moms, n, float(lr), float(momentum),
         float(weight_decay), float(grad_scale), int(zero_mask), _st())


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, grad_scale, step):
    _dev(p, g, m, v)
    call("clhip_adam_step", _ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
         float(weight_decay), float(grad_scale), int(step), _st())


def sq_norm(g, out, accumulate=False):
    _dev(g, out)
    call("clhip_sq_norm", _ptr(g), g.numel(), _ptr(out), int(accumulate), _st())


def scale_(g, s):
    _dev(g)
    call("clhip_scale", _ptr(g), g.numel(), float(s), _st())


def l2_normalize_rows(x):
    x = _f32c(x)
    out = torch.empty_like(x)
    call("clhip_l2_normalize_rows", _ptr(x), _ptr(out), x.shape[0], x.shape[1], _st())
    return out


def ncm_classify(feats, means):
    feats, means = _f32c(feats), _f32c(means)
    pred = torch.empty(feats.shape[0], device=feats.device, dtype=torch.int64)
    call("clhip_ncm_classify", _ptr(feats), _ptr(means), feats.shape[0], means.shape[0], feats.shape[1], _ptr(pred), _st())
    return pred


def herding_select(feats_normed, m):
    """indices (int32 tensor [min(m,n)]) chosen by the greedy mean-matching of linearherdingbuffer.py:140-161."""
    f = _f32c(feats_normed)
    n, D = f.shape
    m = min(int(m), n)
    chosen = torch.empty(m, device=f.device, dtype=torch.int32)
    ws = torch.empty(2 * D + n, device=f.device, dtype=torch.float32)
    call("clhip_herding_select", _ptr(f), n, D, m, _ptr(chosen), _ptr(ws), _st())
    return chosen


def herding_select_classes(feats_normed, counts, m):
    """herding_select for every class of a task in ONE launch: `feats_normed` [sum(counts), D] holds the classes' rows back to back,
    `counts` their row counts.  Returns a list of int32 tensors (class-local indices, min(m, count) each) -- the picks of
    len(counts) herding_select calls (linearherdingbuffer.py:140-161 per class)."""
    f = _f32c(feats_normed)
    N, D = f.shape
    counts = [int(c) for c in counts]
    assert sum(counts) == N and len(counts) > 0
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    offsets = torch.tensor(offs, dtype=torch.int32, device=f.device)
    chosen = torch.empty(len(counts), int(m), device=f.device, dtype=torch.int32)
    ws = torch.empty(2 * D * len(counts) + N, device=f.device, dtype=torch.float32)
    call("clhip_herding_select_batched", _ptr(f), _ptr(offsets), len(counts), max(counts), D, int(m), _ptr(chosen), _ptr(ws), _st())
    return [chosen[i, : min(int(m), c)] for i, c in enumerate(counts)]


def clip_grad_norm_(params, max_norm, eps=1e-6):
    """torch.nn.utils.clip_grad_norm_ (l2p.py:104) without a host sync: total = sqrt(sum ||g||^2) over the given
    parameters, every gradient scaled by min(1, max_norm / (total + eps)) on the device.  Returns the norm tensor."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return None
    _dev(*grads)
    sq = torch.zeros(1, device=grads[0].device, dtype=torch.float32)
    flat = []
    for g in grads:
        if g.dtype != torch.float32 or not g.is_contiguous():
            raise _lib.ClhipError("clip_grad_norm_: gradients must be contiguous fp32")
        call("clhip_sq_norm", _ptr(g), g.numel(), _ptr(sq), 1, _st())
        flat.append(g)
    norm = sq.sqrt()
    coef = (max_norm / (norm + eps)).clamp(max=1.0)
    for g in flat:
        call("clhip_scale_dev", _ptr(g), _ptr(g), g.numel(), 1.0, _ptr(coef), _st())
    return norm


# ------------------------------------------------------------------------------ frozen teacher on a second stream
_SIDE_STREAMS = {}


def _env_unset(name):
    return __import__("os").environ.get(name) is None


class TeacherPass:
    """Runs a frozen teacher's forward on a second HIP stream, concurrently with the student's forward, and joins before the
    loss.  The two forwards are independent until then, and same-shape conv layers spend their time in alternating
    load / compute / store phases that add up instead of overlapping (profiles/r01_conv3_ablation_and_pmc.txt): kernels of the
    other stream fill those gaps -- LwF ResNet-18 task >= 1 step 4.11 -> 3.55 ms.  `CLHIP_TEACHER_STREAM=0` runs it inline.

        tp = ops.TeacherPass(x, lambda: teacher(x))      # before the student's forward
        ... student forward ...
        soft = tp.result()                                # main stream now waits for the side stream
    """

    enabled = __import__("os").environ.get("CLHIP_TEACHER_STREAM", "1") != "0"

    def __init__(self, x, fn):
        self._fn, self._out, self._side = fn, None, None
        # beside a student whose training passes are stage-level launches (CifarResNet-32 above batch 64) the teacher runs INLINE, behind the student's forward: those
        # launches hold every compute unit, so a side stream's kernels only ran in the gaps between them, one launch per unit -- iCaRL at batch 256 1.24 -> 1.12 ms,
        # LUCIR 1.34 -> 1.31 inline (where the teacher itself takes the stage-level launches); LwF ResNet-18 keeps the side stream (2.53 vs 2.75 ms inline)
        inline = False
        if self.enabled and x.is_cuda:
            from .model.backbone import resnet as _resnet
            inline = any(bb.training and bb.takes_stage_launches(x) for bb in list(_resnet._LIVE))
        if self.enabled and x.is_cuda and not inline:
            key = x.device.index or 0
            side = _SIDE_STREAMS.get(key)
            if side is None:
                side = _SIDE_STREAMS[key] = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))        # x (and the teacher's weights) are ready
            # two networks on two streams: their plans keep to ONE stream each in the forward while this pass is in flight (the shortcut
            # branch streams of both make five streams in one step: measured 6.36 ms per LwF ResNet-18 step instead of 2.60, profiles/r03_step_notes.md)
            # (whatever was configured before -- parallel.attach()'s "0" of a multi-rank run, a caller's own choice -- is put back by result())
            self._branch_off = _env_unset("CLHIP_BRANCH_STREAM")
            if self._branch_off:
                self._branch_prev = _lib.lib().clhip_config_get(b"BRANCH_STREAM")
                _lib.lib().clhip_config(b"BRANCH_STREAM", b"0")
            # ... and the side pass keeps to one launch per unit: a stage-level TRAINING launch (stage_train.hip; teachers that `model.train()` returned to batch
            # statistics would take it) needs every workgroup resident at once, and two such grids on two streams would wait for each other's compute units.
            # Per-unit launches beside the student's stage-level ones are the better pairing anyway: they run in the waits of its in-launch all-reduces.
            st_prev = _lib.lib().clhip_config_get(b"STAGE_TRAIN")
            _lib.lib().clhip_config(b"STAGE_TRAIN", b"0")
            try:
                from .model.backbone import resnet as _resnet
                with torch.cuda.stream(side), torch.no_grad(), _resnet.no_backward_follows():
                    self._out = fn()
            finally:
                _lib.lib().clhip_config(b"STAGE_TRAIN", st_prev)
            self._side = side

    def result(self):
        if self._side is None:
            from .model.backbone import resnet as _resnet
            with torch.no_grad(), _resnet.no_backward_follows():
                return self._fn()
        main = torch.cuda.current_stream()
        main.wait_stream(self._side)
        if getattr(self, "_branch_off", False):
            _lib.lib().clhip_config(b"BRANCH_STREAM", self._branch_prev)   # back to what it was for the launches that follow (the backward)
            self._branch_off = False
        outs = self._out if isinstance(self._out, (tuple, list)) else (self._out,)
        for t in outs:
            if torch.is_tensor(t):
                t.record_stream(main)                                     # allocated on the side stream, consumed on this one
        return self._out
