"""L2P plugin (reference core/model/l2p.py:36-122) on the HIP ViT executor.

Same constructor kwargs, hooks, trainable set (prompt pool, prompt keys, classifier), class-window masking
(`index_fill(-inf)` outside the current task == CE over the window [lo, hi)), pull-constraint term and the
backward + clip_grad_norm_(1.0) INSIDE observe (l2p.py:103-104; the trainer calls zero_grad before observe for this
method, trainer.py:593-596).  Per step the device runs: query forward (no grad) -> one selection kernel -> prompted
forward -> head + masked CE (fused loss/dlogits) -> head backward -> ONE backbone backward -> prompt scatter ->
norm + scale kernels.  No host sync: accuracy is returned as a deferred device value inside the trainer loop.
"""
import torch
import torch.nn as nn

from .. import ops
from .heads import HipLinear

_TRAINABLE_TAGS = ("prompt", "classifier")       # l2p.py:72-77: everything else of the network is frozen
_CLIP_NORM = 1.0                                 # l2p.py:104


class Model(nn.Module):
    """prompted backbone + one head over ALL classes (l2p.py:36-44)"""

    def __init__(self, backbone, embed_dim, total_cls_num):
        super().__init__()
        self.backbone = backbone
        self.classifier = HipLinear(embed_dim, total_cls_num, bias=True)

    def forward(self, x, train=True):
        feat, reduce_sim = self.backbone(x, train=train)
        return self.classifier(feat), reduce_sim


class L2P(nn.Module):
    reduces_own_gradients = True      # data parallel: the gradient is all-reduced INSIDE observe, before the norm clip
    grad_reducer = None

    def __init__(self, backbone, device, **kwargs):
        super().__init__()
        self.device = device
        for attr, key in (("init_cls_num", "init_cls_num"), ("inc_cls_num", "inc_cls_num"), ("total_cls_num", "num_class"),
                          ("task_num", "task_num"), ("embed_dim", "feat_dim"), ("pull_constraint_coeff", "pull_constraint_coeff")):
            setattr(self, attr, kwargs[key])
        self.cur_task_id, self._known_classes = 0, 0
        self.network = Model(backbone, self.embed_dim, self.total_cls_num)
        backbone.create_prompt(prompt_flag="l2p", length=kwargs["prompt_length"], prompt_init=nn.init.uniform_, pool_size=kwargs["pool_size"],
                               top_k=kwargs["top_k"], num_layers=1, embed_dim=self.embed_dim)
        self.network.to(device)
        self.unfrezeed_params = self._freeze_all_but(_TRAINABLE_TAGS)

    def _freeze_all_but(self, tags):
        kept = []
        for name, prm in self.network.named_parameters():
            on = any(t in name for t in tags)
            prm.requires_grad_(on)
            if on:
                kept.append(prm)
        return kept

    # ------------------------------------------------------------------------------------------ hooks
    def before_task(self, task_idx, buffer, train_loader, test_loaders):
        self.cur_task_id = task_idx

    def after_task(self, task_idx, buffer, train_loader, test_loaders):
        self._known_classes += self.inc_cls_num if task_idx else self.init_cls_num

    def _window(self):
        """[lo, hi): the classes of the current task -- the only logits left finite by the reference's -inf mask"""
        if self.cur_task_id == 0:
            return 0, self.init_cls_num
        return self._known_classes, self._known_classes + self.inc_cls_num

    def observe(self, data):
        x, y = data["image"].to(self.device), data["label"].to(self.device)
        lo, hi = self._window()
        logits, reduce_sim = self.network(x, train=True)
        aux = ops.LossAux()
        loss = ops.classify_loss(logits, y, lo=lo, hi=hi, pred_lo=lo, pred_hi=hi, aux=aux) - self.pull_constraint_coeff * reduce_sim
        loss.backward()
        if self.grad_reducer is not None:
            self.grad_reducer.reduce_mean(self.network)
        ops.clip_grad_norm_(self.unfrezeed_params, _CLIP_NORM)
        self._last_aux = aux
        return aux.pred, aux.acc(), loss.detach()

    def inference(self, data):
        x, y = data["image"].to(self.device), data["label"].to(self.device)
        with torch.no_grad():
            logits, _ = self.network(x, train=False)
        pred, correct = ops.predict(logits, y)
        return pred, correct.item() / x.size(0)

    def get_parameters(self, config):
        return self.unfrezeed_params
