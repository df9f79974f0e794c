"""Finetune: the base plugin (reference core/model/finetune.py:4-51) on the HIP hot path."""
import torch
from torch import nn

from .. import ops
from .heads import HipLinear


class Finetune(nn.Module):
    def __init__(self, backbone, feat_dim, num_class, **kwargs):
        super().__init__()
        self.backbone = backbone
        self.feat_dim = feat_dim
        self.num_class = num_class
        self.classifier = HipLinear(feat_dim, num_class)
        self.loss_fn = nn.CrossEntropyLoss(reduction="mean")   # kept for attribute compatibility; the fused kernel computes it
        self.device = kwargs["device"]
        self.kwargs = kwargs

    def _xy(self, data):
        return data["image"].to(self.device), data["label"].to(self.device)

    def observe(self, data):
        x, y = self._xy(data)
        logit = self.classifier(self.backbone(x)["features"])
        aux = ops.LossAux()
        loss = ops.classify_loss(logit, y, aux=aux)
        self._last_aux = aux
        return aux.pred, aux.acc(), loss

    def inference(self, data):
        x, y = self._xy(data)
        logit = self.classifier(self.backbone(x)["features"])
        pred, correct = ops.predict(logit, y)
        return pred, correct.item() / x.size(0)

    def forward(self, x):
        return self.classifier(self.backbone(x)["features"])

    def before_task(self, task_idx, buffer, train_loader, test_loaders):
        pass

    def after_task(self, task_idx, buffer, train_loader, test_loaders):
        pass

    def get_parameters(self, config):
        return [{"params": self.backbone.parameters()}, {"params": self.classifier.parameters()}]
