"""WA plugin -- weight aligning (reference core/model/wa.py:43-243) on the HIP hot path.

Loss = (1-l)*CE(all seen logits) + l*KD(T=2, old logits vs frozen previous network), l = known/total: ONE fused loss node
(ce_slice + kd kernels), teacher forward = the same plan executor without gradients.  After every task > 0 the rows of the
newest classes are rescaled so that their mean L2 norm equals the old rows' (wa.py:96-109) and the network is snapshotted as the
next teacher; rehearsal goes through the herding buffer exactly like iCaRL.

Reference quirks kept because a drop-in must train the same thing:
  * `get_parameters` is Finetune's: it hands the optimizer the backbone and Finetune's own, otherwise unused, `classifier` --
    the head that produces the logits (`network.classifier`) is therefore NEVER stepped and only changes through the alignment;
  * the class count grows by `init_cls_num` every task (wa.py:222), whatever `inc_cls_num` says;
  * the frozen teacher is an nn.Module attribute, so `model.train()` returns its BatchNorm to batch statistics (like LwF, a10).
"""
import numpy as np
import torch
from torch import nn

from .. import ops
from .finetune import Finetune
from .heads import HipLinear, teacher_of, widened


class IncrementalModel(nn.Module):
    """backbone + a head that is re-allocated wider at every task (wa.py:43-139)"""

    def __init__(self, backbone, feat_dim, num_class):
        super().__init__()
        self.backbone, self.feat_dim, self.num_class = backbone, feat_dim, num_class
        self.classifier = None

    def extract_vector(self, x):
        return self.backbone(x)["features"]

    def get_logits(self, x):
        return self.classifier(self.extract_vector(x))

    forward = get_logits

    def update_classifier(self, number_classes):
        dev = next(self.backbone.parameters()).device
        if self.classifier is None:
            self.classifier = HipLinear(self.feat_dim, number_classes).to(dev)
        else:
            self.classifier = widened(self.classifier, number_classes, dev)

    @torch.no_grad()
    def classifier_weight_align(self, incremental_number):
        """new rows *= mean|old row| / mean|new row|  (wa.py:96-109); returns the factor"""
        w = self.classifier.weight
        row_norm = w.float().norm(dim=1)
        gamma = row_norm[:-incremental_number].mean() / row_norm[-incremental_number:].mean()
        w[-incremental_number:] *= gamma
        return gamma

    def freeze(self):
        for q in self.parameters():
            q.requires_grad = False
        return self.eval()


class WA(Finetune):
    cuda_graph_safe = False     # not audited for trainer.GraphedStep
    def __init__(self, backbone, feat_dim, num_class, **kwargs):
        super().__init__(backbone, feat_dim, num_class, **kwargs)
        self.network = IncrementalModel(self.backbone, feat_dim, kwargs["init_cls_num"])
        self.old_network = None
        self.known_classes = self.total_classes = 0
        self.task_idx = 0
        self.total_classes_indexes = 0          # class ids of the running task, read by the herding buffer

    def forward(self, x):
        return self.network(x)

    def observe(self, data):
        x, y = self._xy(data)
        teacher = ops.TeacherPass(x, lambda: self.old_network(x)) if self.task_idx > 0 else None
        logits = self.network(x)
        aux = ops.LossAux()
        if self.task_idx > 0:
            lam = self.known_classes / self.total_classes
            soft = teacher.result()
            loss = ops.classify_loss(logits, y, w_ce=1.0 - lam, teacher=soft, k=self.known_classes, T=2.0, w_kd=lam, aux=aux)
        else:
            loss = ops.classify_loss(logits, y, aux=aux)
        self._last_aux = aux
        return aux.pred, aux.acc(), loss

    def inference(self, data):
        x, y = self._xy(data)
        pred, correct = ops.predict(self.network(x), y)
        return pred, correct.item() / x.size(0)

    def before_task(self, task_idx, buffer, train_loader, test_loaders):
        self.total_classes += self.kwargs["init_cls_num"]
        self.network.update_classifier(self.total_classes)
        self.total_classes_indexes = np.arange(self.known_classes, self.total_classes)

    def after_task(self, task_idx, buffer, train_loader, test_loaders):
        if self.task_idx > 0:
            self.network.classifier_weight_align(self.total_classes - self.known_classes)
        self.old_network = teacher_of(self.network)
        self.known_classes = self.total_classes
        buffer.reduce_old_data(self.task_idx, self.total_classes)
        buffer.update(self.network, train_loader, test_loaders[0].dataset.trfms, self.task_idx, self.total_classes,
                      self.total_classes_indexes, self.device)
        self.task_idx += 1
