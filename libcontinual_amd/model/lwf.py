"""LwF plugin (reference core/model/lwf.py:9-81) on the HIP hot path.

Same surface and quirks: head grows to init + task_idx*inc outputs with old rows copied; the teacher is a
frozen deep copy of backbone + head taken in before_task; KD weight hard-coded to 3 with T=2 (lwf.py:64,
the YAML `lamda` is ignored); because the teacher is an nn.Module attribute, `model.train()` puts its
BatchNorm back in train mode -- reproduced simply by being an nn.Module too.
CE on the new-class slice and the KD term are fused into one loss node (ce_slice + kd kernels).
"""
import copy

import torch
import torch.nn as nn

from .. import ops
from .finetune import Finetune
from .heads import HipLinear


class LWF(Finetune):
    def __init__(self, backbone, feat_dim, num_class, **kwargs):
        super().__init__(backbone, feat_dim, num_class, **kwargs)
        self.kwargs = kwargs
        self.feat_dim = feat_dim
        self.classifier = HipLinear(self.feat_dim, kwargs["init_cls_num"])
        self.old_fc = None
        self.init_cls_num = kwargs["init_cls_num"]
        self.inc_cls_num = kwargs["inc_cls_num"]
        self.known_cls_num = 0
        self.total_cls_num = 0
        self.old_backbone = None

    def freeze(self, module):
        for p in module.parameters():
            p.requires_grad = False
        module.eval()
        return module

    def update_fc(self):
        fc = HipLinear(self.feat_dim, self.total_cls_num).to(self.device)
        if self.classifier is not None:
            self.old_fc = self.freeze(copy.deepcopy(self.classifier))
            old_out = self.classifier.out_features
            with torch.no_grad():
                fc.weight.data[:old_out] = self.classifier.weight.data
                fc.bias.data[:old_out] = self.classifier.bias.data
        self.classifier = fc

    def before_task(self, task_idx, buffer, train_loader, test_loaders):
        self.task_idx = task_idx
        self.known_cls_num = self.total_cls_num
        self.total_cls_num = self.init_cls_num + self.task_idx * self.inc_cls_num
        self.update_fc()
        self.loss_fn = nn.CrossEntropyLoss()
        if task_idx != 0:
            self.old_backbone = self.freeze(copy.deepcopy(self.backbone)).to(self.device)

    def observe(self, data):
        x, y = self._xy(data)
        logit = self.classifier(self.backbone(x)["features"])
        aux = ops.LossAux()
        if self.task_idx == 0:
            loss = ops.classify_loss(logit, y, aux=aux)
        else:
            k = self.known_cls_num
            with torch.no_grad():
                soft = self.old_fc(self.old_backbone(x)["features"])
            # loss = 3 * KD(logit[:, :k], soft, T=2) + CE(logit[:, k:], y - k)     (lwf.py:61-65)
            loss = ops.classify_loss(logit, y, lo=k, hi=logit.shape[1], w_ce=1.0, teacher=soft, k=k, T=2.0, w_kd=3.0, aux=aux)
        self._last_aux = aux
        return aux.pred, aux.acc(), loss

    def after_task(self, task_idx, buffer, train_loader, test_loaders):
        pass
