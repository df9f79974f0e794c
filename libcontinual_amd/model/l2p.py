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


class Model(nn.Module):
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
        self.init_cls_num = kwargs["init_cls_num"]
        self.inc_cls_num = kwargs["inc_cls_num"]
        self.total_cls_num = kwargs["num_class"]
        self.task_num = kwargs["task_num"]
        self.embed_dim = kwargs["feat_dim"]
        self.pull_constraint_coeff = kwargs["pull_constraint_coeff"]
        self.cur_task_id = 0
        self._known_classes = 0
        self.network = Model(backbone, self.embed_dim, self.total_cls_num)
        self.network.backbone.create_prompt(prompt_flag="l2p", length=kwargs["prompt_length"], prompt_init=nn.init.uniform_,
                                            pool_size=kwargs["pool_size"], top_k=kwargs["top_k"], num_layers=1, embed_dim=self.embed_dim)
        self.network.to(self.device)
        self.unfrezeed_params = []
        for name, param in self.network.named_parameters():
            param.requires_grad_(False)
            if "prompt" in name or "classifier" in name:
                param.requires_grad_(True)
                self.unfrezeed_params.append(param)

    def before_task(self, task_idx, buffer, train_loader, test_loaders):
        self.cur_task_id = task_idx

    def after_task(self, task_idx, buffer, train_loader, test_loaders):
        self._known_classes += self.init_cls_num if task_idx == 0 else self.inc_cls_num

    def _window(self):
        if self.cur_task_id == 0:
            return 0, self.init_cls_num
        return self._known_classes, self._known_classes + self.inc_cls_num

    def observe(self, data):
        x, y = data["image"].to(self.device), data["label"].to(self.device)
        logits, reduce_sim = self.network(x, train=True)
        lo, hi = self._window()
        aux = ops.LossAux()
        # CE over logits masked to -inf outside [lo, hi) (l2p.py:92-101); argmax over the same window
        ce = ops.classify_loss(logits, y, lo=lo, hi=hi, pred_hi=hi, aux=aux, pred_lo=lo)
        loss = ce - self.pull_constraint_coeff * reduce_sim
        loss.backward()
        if self.grad_reducer is not None:
            self.grad_reducer.reduce_mean(self.network)
        ops.clip_grad_norm_(self.unfrezeed_params, 1.0)
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
