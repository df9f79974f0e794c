"""LwF plugin (reference core/model/lwf.py:9-81) on the HIP hot path.

Same surface and quirks: head grows to init + task_idx*inc outputs with old rows copied; the teacher is a
frozen deep copy of backbone + head taken in before_task; KD weight hard-coded to 3 with T=2 (lwf.py:64,
the YAML `lamda` is ignored); because the teacher is an nn.Module attribute, `model.train()` puts its
BatchNorm back in train mode -- reproduced simply by being an nn.Module too.
CE on the new-class slice and the KD term are fused into one loss node (ce_slice + kd kernels).
"""
import torch
import torch.nn as nn

from .. import ops
from .finetune import Finetune
from .heads import HipLinear, teacher_of, widened

_KD_WEIGHT, _KD_TEMPERATURE = 3.0, 2.0          # lwf.py:62-64 (not configurable there either)


class LWF(Finetune):
    def __init__(self, backbone, feat_dim, num_class, **kwargs):
        super().__init__(backbone, feat_dim, num_class, **kwargs)
        self.kwargs, self.feat_dim = kwargs, feat_dim
        self.init_cls_num, self.inc_cls_num = kwargs["init_cls_num"], kwargs["inc_cls_num"]
        self.known_cls_num = self.total_cls_num = 0
        self.classifier = HipLinear(feat_dim, self.init_cls_num)
        self.old_fc = self.old_backbone = None          # the teacher: previous head / previous backbone

    # kept for API compatibility (lwf.py:22-26)
    def freeze(self, module):
        for q in module.parameters():
            q.requires_grad = False
        return module.eval()

    def update_fc(self):
        """snapshot the current head as the teacher head, then widen it to `total_cls_num` outputs (lwf.py:28-40)"""
        self.old_fc = teacher_of(self.classifier)
        self.classifier = widened(self.classifier, self.total_cls_num, self.device)

    def before_task(self, task_idx, buffer, train_loader, test_loaders):
        self.task_idx = task_idx
        self.known_cls_num, self.total_cls_num = self.total_cls_num, self.init_cls_num + task_idx * self.inc_cls_num
        self.update_fc()
        self.loss_fn = nn.CrossEntropyLoss()
        if task_idx > 0:
            self.old_backbone = teacher_of(self.backbone, self.device)

    def observe(self, data):
        x, y = self._xy(data)
        teacher = ops.TeacherPass(x, lambda: self.old_fc(self.old_backbone(x)["features"])) if self.task_idx > 0 else None
        logit = self.classifier(self.backbone(x)["features"])
        aux = ops.LossAux()
        old = self.known_cls_num
        if self.task_idx == 0:
            loss = ops.classify_loss(logit, y, aux=aux)
        else:
            soft = teacher.result()
            # 3 * KD(logit[:, :old], soft, T=2) + CE(logit[:, old:], y - old) in one fused node
            loss = ops.classify_loss(logit, y, lo=old, hi=logit.shape[1], w_ce=1.0, teacher=soft, k=old, T=_KD_TEMPERATURE, w_kd=_KD_WEIGHT, aux=aux)
        self._last_aux = aux
        return aux.pred, aux.acc(), loss

    def after_task(self, task_idx, buffer, train_loader, test_loaders):
        pass
