"""EWC plugin (reference core/model/ewc.py:43-229) on the HIP hot path.

Same constructor, hooks, `fisher` / `ref_param` dicts (keys = `network.named_parameters()` names) and
quirks as the reference: Fisher of the *batch-mean* gradient scaled by
len(y), divisor batch_size*len(loader), BatchNorm in train mode during the Fisher pass, ref snapshot before
the Fisher pass, alpha merge applied even after task 0, head slice `p[:len(ref)]`.

MI355X design: the backbone's Fisher / reference parameters are single flat fp32 buffers aligned with the
backbone's flat parameter buffer, so the penalty is ONE streaming reduction (clhip_ewc_penalty) instead
of ~300 tiny torch kernels per step (ewc.py:221-225), its gradient ONE fused axpy into the flat gradient
buffer (clhip_ewc_grad), and the Fisher accumulation ONE launch per batch (clhip_fisher_accum).
"""
import torch
import torch.nn as nn

from .. import ops
from .._lib import call
from .finetune import Finetune
from .heads import HipLinear


class Model(nn.Module):
    """backbone + linear classifier (reference ewc.py:43-57)"""

    def __init__(self, backbone, feat_dim, num_class):
        super().__init__()
        self.backbone = backbone
        self.feat_dim = feat_dim
        self.num_class = num_class
        self.classifier = HipLinear(feat_dim, num_class)

    def forward(self, x):
        return self.get_logits(x)

    def get_logits(self, x):
        return self.classifier(self.backbone(x)["features"])


class _EwcLossFn(torch.autograd.Function):
    """logits = feat W^T + b;  CE(logits[:, lo:], y - lo) + lamda * sum F (p - p*)^2 / 2  as ONE autograd node (round 4: the head's linear layer is
    part of it).  Forward: linear_fwd + ce_slice + ewc_penalty (backbone flat buffer, head-weight and head-bias prefixes) into one scalar.  Backward:
    the head's input / weight / bias gradients by one launch, then lamda * F * (p - p*) added straight into the backbone's flat gradient buffer and into
    those two head gradients -- the node is the ONLY producer of the head's gradients, so autograd neither zero-fills nor adds anything (the
    separate-nodes form cost two torch fills and two torch adds per step: 19 us of a 0.82-ms 32-image step)."""

    @staticmethod
    def forward(ctx, feat, labels, lo, owner, anchor, head_w, head_b, aux):
        st = torch.cuda.current_stream().cuda_stream
        feat = feat.float().contiguous()
        B, D = feat.shape
        O = head_w.shape[0]
        logits = torch.empty(B, O, device=feat.device, dtype=torch.float32)
        call("clhip_linear_fwd", feat.data_ptr(), head_w.data_ptr(), head_b.data_ptr(), logits.data_ptr(), B, D, O, st)
        labels = labels.to(torch.int64).contiguous()
        loss = torch.empty(1, device=logits.device, dtype=torch.float32)
        dlog = torch.empty_like(logits)
        pred = torch.empty(B, device=logits.device, dtype=torch.int64)
        correct = torch.empty(1, device=logits.device, dtype=torch.int32)
        call("clhip_ce_slice", logits.data_ptr(), labels.data_ptr(), B, O, lo, O, O, 1.0, loss.data_ptr(), 0, dlog.data_ptr(), 0,
             pred.data_ptr(), correct.data_ptr(), st)
        flat, _ = owner.network.backbone.flat_parameters()
        lam = float(owner.lamda)
        nw, nb = owner._ref_head_w.numel(), owner._ref_head_b.numel()
        ops.ewc_penalty_multi([(flat, owner._ref_flat, owner._fisher_flat),
                               (head_w.detach().reshape(-1)[:nw], owner._ref_head_w.reshape(-1), owner._fisher_head_w.reshape(-1)),
                               (head_b.detach()[:nb], owner._ref_head_b, owner._fisher_head_b)], lam, loss, True)
        aux.pred, aux.correct, aux.batch = pred, correct, B
        ctx.owner = owner
        ctx.save_for_backward(dlog, feat, head_w, head_b)
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        dlog, feat, head_w, head_b = ctx.saved_tensors
        owner = ctx.owner
        unit = gout.is_cuda and gout.data_ptr() in ops.UNIT_GRAD_PTRS      # the trainer's cached unit root gradient: no scaling launch
        gout = gout.reshape(1).float().contiguous()
        st = torch.cuda.current_stream().cuda_stream
        if unit:
            dl = dlog
        else:
            dl = torch.empty_like(dlog)
            call("clhip_scale_dev", dlog.data_ptr(), dl.data_ptr(), dlog.numel(), 1.0, gout.data_ptr(), st)
        B, D = feat.shape
        O = head_w.shape[0]
        dfeat = torch.empty_like(feat)
        gw = torch.empty_like(head_w)
        gb = torch.empty_like(head_b)
        call("clhip_linear_bwd", feat.data_ptr(), head_w.data_ptr(), dl.data_ptr(), dfeat.data_ptr(), gw.data_ptr(), gb.data_ptr(), B, D, O, 0, st)
        bb = owner.network.backbone
        gflat = bb.begin_grad_write()              # zeroes the buffer if this is the first write after zero_grad()
        flat, _ = bb.flat_parameters()
        lam = float(owner.lamda)
        nw, nb = owner._ref_head_w.numel(), owner._ref_head_b.numel()
        ops.ewc_grad_multi([(flat, owner._ref_flat, owner._fisher_flat, gflat),
                            (head_w.detach().reshape(-1)[:nw], owner._ref_head_w.reshape(-1), owner._fisher_head_w.reshape(-1), gw.view(-1)[:nw]),
                            (head_b.detach()[:nb], owner._ref_head_b, owner._fisher_head_b, gb[:nb])], lam, gout)
        bb.attach_grads()
        return dfeat, None, None, None, None, gw, gb, None


class EWC(Finetune):
    # graph-safe on every task: the penalty node above launches flat-buffer kernels with device-pointer arguments only, and the batch-32 step of task 1 replays
    # (bench.py --workload ewc_resnet32_b50_task1 --batch 32: 0.74 ms replayed, 0.87-1.03 ms eager).  The aborted captures GPUTEST_r04 logged for EWC come
    # from tests/test_trainer_trace_gpu.py's Recorder, which reads every loss on the host inside observe() -- a wrapper the `auto` mode's fallback exists for; declaring the
    # method unsafe from task 1 on (tried early in round 5) cost the real loop its replay and was reverted.
    cuda_graph_safe = True

    def __init__(self, backbone, feat_dim, num_class, **kwargs):
        super().__init__(backbone, feat_dim, num_class, **kwargs)
        self.kwargs = kwargs
        self.network = Model(self.backbone, feat_dim, kwargs["init_cls_num"])
        self.lamda = self.kwargs["lamda"]
        self.task_idx = 0
        # snapshot at construction, like the reference (ewc.py:65-68): head entries have init_cls_num rows
        flat, _ = self.network.backbone.flat_parameters()
        cls = self.network.classifier
        self._ref_flat = flat.detach().clone()
        self._fisher_flat = torch.zeros_like(flat)
        self._ref_head_w, self._ref_head_b = cls.weight.detach().clone(), cls.bias.detach().clone()
        self._fisher_head_w, self._fisher_head_b = torch.zeros_like(self._ref_head_w), torch.zeros_like(self._ref_head_b)

    # -- name -> tensor views, the reference's public attributes (ewc.py:65-68)
    def _named_views(self, flat, hw, hb):
        bb = self.network.backbone
        out = {}
        for (nm, shp, off, is_conv) in bb._layout:
            out["backbone." + nm] = bb._view(flat, shp, off, is_conv)
        out["classifier.weight"] = hw
        out["classifier.bias"] = hb
        return out

    @property
    def fisher(self):
        self._ensure_state()
        return self._named_views(self._fisher_flat, self._fisher_head_w, self._fisher_head_b)

    @property
    def ref_param(self):
        self._ensure_state()
        return self._named_views(self._ref_flat, self._ref_head_w, self._ref_head_b)

    def _ensure_state(self):
        """the Fisher / reference tensors follow the network to its device (they are plain attributes, not buffers)"""
        flat, _ = self.network.backbone.flat_parameters()
        for n in ("_ref_flat", "_fisher_flat", "_ref_head_w", "_ref_head_b", "_fisher_head_w", "_fisher_head_b"):
            t = getattr(self, n)
            if t.device != flat.device:
                setattr(self, n, t.to(flat.device))

    def before_task(self, task_idx, buffer, train_loader, test_loaders):
        """grow the head to init + task_idx*inc outputs, old rows copied (ewc.py:71-80)"""
        self.task_idx = task_idx
        old = self.network.classifier
        new_fc = HipLinear(old.in_features, self.kwargs["init_cls_num"] + task_idx * self.kwargs["inc_cls_num"])
        new_fc = new_fc.to(old.weight.device)
        with torch.no_grad():
            new_fc.weight.data[: old.out_features] = old.weight.data
            new_fc.bias.data[: old.out_features] = old.bias.data
        self.network.classifier = new_fc
        self.network.to(self.device)

    def observe(self, data):
        x, y = self._xy(data)
        aux = ops.LossAux()
        if self.task_idx == 0:
            loss = ops.classify_loss(self.network(x), y, aux=aux)
        else:
            self._ensure_state()
            old_classes = self.network.classifier.out_features - self.kwargs["inc_cls_num"]
            cls = self.network.classifier
            feat = self.network.backbone(x)["features"]
            loss = _EwcLossFn.apply(feat, y, old_classes, self, self.network.backbone._params[0], cls.weight, cls.bias, aux)
        self._last_aux = aux
        return aux.pred, aux.acc(), loss

    def compute_ewc(self):
        """value of sum_n sum F_n (p_n[:len(ref_n)] - ref_n)^2 / 2 (no grad; ewc.py:207-225)"""
        self._ensure_state()
        flat, _ = self.network.backbone.flat_parameters()
        out = torch.empty(1, device=flat.device, dtype=torch.float32)
        cls = self.network.classifier
        nw, nb = self._ref_head_w.numel(), self._ref_head_b.numel()
        ops.ewc_penalty(flat, self._ref_flat, self._fisher_flat, 1.0, out, False)
        ops.ewc_penalty(cls.weight.detach().reshape(-1)[:nw], self._ref_head_w.reshape(-1), self._fisher_head_w.reshape(-1), 1.0, out, True)
        ops.ewc_penalty(cls.bias.detach()[:nb], self._ref_head_b, self._fisher_head_b, 1.0, out, True)
        return out.view(())

    def inference(self, data):
        x, y = self._xy(data)
        logit = self.network(x)
        pred, correct = ops.predict(logit, y)
        return pred, correct.item() / x.size(0)

    def getFisher(self, train_loader):
        """one pass over the task's loader: CE(all logits) -> backward -> fisher += grad^2 * len(y);
        / (batch_size*len(loader))  (ewc.py:147-205).  Returns (flat, head_w, head_b) Fisher tensors."""
        net = self.network
        bb, cls = net.backbone, net.classifier
        flat, gflat = bb.flat_parameters()
        f_flat = torch.zeros_like(flat)
        f_w = torch.zeros_like(cls.weight)
        f_b = torch.zeros_like(cls.bias)
        net.train()                                     # BN uses (and updates) batch statistics: quirk a9(iii)
        num_samples = train_loader.batch_size * len(train_loader)
        # The Fisher pass runs in fp32 whatever the training dtype (`fisher_dtype: bf16` in the classifier kwargs opts out): Fisher
        # diagonals are a stated output of the reference (ewc.py:147-205) and squares of bf16-path gradients are 20-40 % off per
        # entry on the fixtures, while the fp32 pass holds all 101 tensors to 1.3e-3 (tests/test_parity_gpu.py).  It runs once per
        # task over the task's data; the training steps keep the plan of the backbone's own dtype.
        # (round 5: replaying this loop as a HIP graph was built and measured -- 1.86 vs 1.84 ms per batch of 32: the pass is not host-bound, its fp32
        #  plan runs the generic implicit-GEMM / weight-gradient kernels, 1.92 ms of kernel time per batch; profiles/r05_notes.md)
        with bb.compute_dtype(self.kwargs.get("fisher_dtype", "f32")):
            for data in train_loader:
                x, y = self._xy(data)
                for p in net.parameters():
                    p.grad = None
                loss = ops.classify_loss(net(x), y)
                loss.backward()
                s = float(len(y)) / float(num_samples)
                ops.fisher_accum(f_flat, gflat, s)
                ops.fisher_accum(f_w.view(-1), cls.weight.grad.contiguous().view(-1), s)
                ops.fisher_accum(f_b, cls.bias.grad.contiguous(), s)
        for p in net.parameters():
            p.grad = None
        return f_flat, f_w, f_b

    def after_task(self, task_idx, buffer, train_loader, test_loaders):
        """ewc.py:110-133"""
        self._ensure_state()
        bb, cls = self.network.backbone, self.network.classifier
        flat, _ = bb.flat_parameters()
        ref_flat = flat.detach().clone()                # snapshot BEFORE the Fisher pass (quirk a9(vi))
        ref_w, ref_b = cls.weight.detach().clone(), cls.bias.detach().clone()
        nf, nw, nb = self.getFisher(train_loader)
        alpha = 1 - self.kwargs["inc_cls_num"] / cls.out_features
        ops.fisher_merge(nf, self._fisher_flat, alpha)
        ow, ob = self._fisher_head_w, self._fisher_head_b
        ops.fisher_merge(nw.view(-1)[: ow.numel()], ow.reshape(-1), alpha)
        ops.fisher_merge(nb[: ob.numel()], ob, alpha)
        self._fisher_flat, self._fisher_head_w, self._fisher_head_b = nf, nw, nb
        self._ref_flat, self._ref_head_w, self._ref_head_b = ref_flat, ref_w, ref_b

    def get_parameters(self, config):
        return [{"params": self.network.parameters()}]
