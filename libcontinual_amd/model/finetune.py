"""Finetune: the base plugin (reference core/model/finetune.py:4-51) on the HIP hot path -- backbone + one linear head,
plain cross entropy, no continual-learning mechanism.  The other ResNet methods derive from it for `device`, `_xy` and the
default hooks."""
import torch
from torch import nn

from .. import ops
from .heads import HipLinear


class Finetune(nn.Module):
    # a training step has no data-dependent host control flow and touches fixed buffers only: trainer.GraphedStep may capture it
    # (inherited by EWC and LWF; methods with host-side branching per batch leave it False)
    cuda_graph_safe = True

    def __init__(self, backbone, feat_dim, num_class, **kwargs):
        super().__init__()
        self.kwargs = kwargs
        self.device = kwargs["device"]
        self.backbone, self.feat_dim, self.num_class = backbone, feat_dim, num_class
        self.classifier = HipLinear(feat_dim, num_class)
        # attribute kept because reference plugins / user code read it; the fused kernel computes the same mean CE
        self.loss_fn = nn.CrossEntropyLoss(reduction="mean")

    # ---- helpers shared with the subclasses
    def _xy(self, data):
        """batch dict -> (images, labels) on the plugin's device"""
        return data["image"].to(self.device), data["label"].to(self.device)

    def _logits(self, x):
        return self.classifier(self.backbone(x)["features"])

    def forward(self, x):
        return self._logits(x)

    # ---- plugin surface
    def observe(self, data):
        x, y = self._xy(data)
        aux = ops.LossAux()
        loss = ops.classify_loss(self._logits(x), y, aux=aux)      # loss, dlogits, argmax and the correct count in one launch
        self._last_aux = aux
        return aux.pred, aux.acc(), loss

    def inference(self, data):
        x, y = self._xy(data)
        pred, correct = ops.predict(self._logits(x), y)
        return pred, correct.item() / x.size(0)

    def before_task(self, task_idx, buffer, train_loader, test_loaders):
        """nothing to prepare"""

    def after_task(self, task_idx, buffer, train_loader, test_loaders):
        """nothing to consolidate"""

    def get_parameters(self, config):
        return [{"params": self.backbone.parameters()}, {"params": self.classifier.parameters()}]
