"""iCaRL plugin (reference core/model/icarl.py:24-287) on the HIP hot path.

CE over all seen classes + KD(T=2) against a deep copy of the previous network, herding rehearsal buffer,
nearest-class-mean inference.  KD + CE are one fused loss node; NCM distance/argmin and the herding greedy
loop are HIP kernels (clhip_ncm_classify, clhip_herding_select); herding and class means stay single-GPU.
The reference back-propagates into the (unused) teacher because its output is not detached
 -- those gradients are discarded there, so the teacher runs without grad here.
"""
import copy
import os

import numpy as np
import torch
from torch import nn
from torch.utils.data import DataLoader, Dataset

from .. import ops
from .heads import HipLinear


class Model(nn.Module):
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


class ICarl(nn.Module):
    cuda_graph_safe = True      # observe(): student forward, frozen teacher on its side stream (ops.TeacherPass), one fused CE + KD loss -- no host reads, fixed shapes per batch size

    def __init__(self, backbone, feat_dim, num_class, **kwargs):
        super().__init__()
        self.device = kwargs["device"]
        self.cur_task_id = 0
        self.cur_cls_indexes = None
        self.network = Model(backbone, feat_dim, num_class)
        self.old_network = None
        self.prev_cls_num = 0
        self.accu_cls_num = 0
        self.init_cls_num = kwargs["init_cls_num"]
        self.inc_cls_num = kwargs["inc_cls_num"]
        self.task_num = kwargs["task_num"]
        self.class_means = None

    def get_parameters(self, config):
        return self.network.parameters()

    def _xy(self, data):
        return data["image"].to(self.device), data["label"].to(self.device)

    def observe(self, data):
        x, y = self._xy(data)
        teacher = ops.TeacherPass(x, lambda: self.old_network(x)) if self.cur_task_id > 0 else None
        logits = self.network(x)
        aux = ops.LossAux()
        n = self.accu_cls_num
        if self.cur_task_id > 0:
            old_logits = teacher.result()
            # CE(logits[:, :n], y) + KD(logits[:, :prev], old[:, :prev], T=2)      (icarl.py:208-219)
            loss = ops.classify_loss(logits, y, lo=0, hi=n, pred_hi=n, w_ce=1.0, teacher=old_logits, k=self.prev_cls_num,
                                     T=2.0, w_kd=1.0, aux=aux)
        else:
            loss = ops.classify_loss(logits, y, lo=0, hi=n, pred_hi=n, aux=aux)
        self._last_aux = aux
        return aux.pred, aux.acc(), loss

    def inference(self, data):
        if self.class_means is not None and len(self.class_means) == self.accu_cls_num:
            return self.NCM_classify(data)
        x, y = self._xy(data)
        logits = self.network(x)
        pred, correct = ops.predict(logits, y, pred_hi=self.accu_cls_num)
        return pred, correct.item() / x.size(0)

    def NCM_classify(self, data):
        x, y = self._xy(data)
        feats = self.network.backbone(x)["features"]
        pred = ops.ncm_classify(feats, self.class_means)
        acc = torch.sum(pred == y).item()
        return pred, acc / x.size(0)

    def forward(self, x):
        return self.network(x)[:, : self.accu_cls_num]

    def before_task(self, task_idx, buffer, train_loader, test_loaders):
        if self.cur_task_id == 0:
            self.accu_cls_num = self.init_cls_num
        else:
            self.accu_cls_num += self.inc_cls_num
        self.cur_cls_indexes = np.arange(self.prev_cls_num, self.accu_cls_num)

    def after_task(self, task_idx, buffer, train_loader, test_loaders):
        self.old_network = copy.deepcopy(self.network)
        self.old_network.eval()
        self.prev_cls_num = self.accu_cls_num
        buffer.reduce_old_data(self.cur_task_id, self.accu_cls_num)
        val_transform = test_loaders[0].dataset.trfms
        buffer.update(self.network, train_loader, val_transform, self.cur_task_id, self.accu_cls_num, self.cur_cls_indexes,
                      self.device)
        self.class_means = self.calc_class_mean(buffer, train_loader, val_transform, self.device).to(self.device)
        self.cur_task_id += 1

    def calc_class_mean(self, buffer, train_loader, val_transform, device):
        """class prototypes from the BUFFER samples only: L2-normalised features, per-class mean, re-normalised
        (icarl.py:226-287)"""
        ds = copy.copy(train_loader.dataset)
        ds.images, ds.labels, ds.trfms = list(buffer.images), list(buffer.labels), val_transform
        loader = DataLoader(ds, batch_size=train_loader.batch_size, shuffle=False, num_workers=0)
        feats, targets = [], []
        with torch.no_grad():
            self.network.eval()
            for data in loader:
                images = data["image"].to(device)
                f = self.network.backbone(images)["features"]
                feats.append(ops.l2_normalize_rows(f))
                targets.append(data["label"].to(device))
        feats = torch.cat(feats)
        targets = torch.cat(targets)
        means = []
        for c in torch.unique(targets).tolist():
            m = feats[targets == c].mean(0, keepdim=True)
            means.append(ops.l2_normalize_rows(m)[0])
        return torch.stack(means)
