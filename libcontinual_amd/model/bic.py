"""BiC plugin -- "Large Scale Incremental Learning" (reference core/model/bic.py:72-340 + the trainer's second stage,
core/trainer.py:71-72, 297-303, 420-455, 534-561) on the HIP hot path.

Surface and quirks kept:

* `Model` = backbone + ONE full-width `nn.Linear(backbone.feat_dim, num_class)` (bic.py:72-81); the backbone is called for a
  feature TENSOR (the BiC configs use `cifar_resnet32_V2`, whose forward returns one; dict-returning backbones are accepted too).
* one `BiasLayer` (alpha, beta; backbone/resnet.py:579-587) per task; `bias_forward` applies EVERY layer to its slice of the logits
  whatever its `train` argument says (bic.py:129 overwrites it), in stage 1, stage 2 and inference alike.
* stage 1 = CE over the seen classes (task 0) or  alpha * T^2 * KD(T=2, old columns, teacher = bias-corrected previous model) +
  (1 - alpha) * CE,  alpha = old / seen, with the `cur_task / (cur_task + 1) == alpha` assertion (bic.py:193-217: it only holds
  when init_cls_num == inc_cls_num, like every shipped config).  `previous_model` is a deep copy kept as a sub-module: the trainer's
  `model.train()` puts its BatchNorm in batch-statistics mode.
* stage 2 (bic.py:219-232): the model frozen and in eval mode, the CURRENT task's bias layer trained with the plugin's own
  Adam(lr 1e-3) on the class-balanced validation split.
* `spilt_and_update` (bic.py:245-340): 9:1 class-wise split of the task data with the GLOBAL numpy RNG, train loader = 90 % +
  rehearsal train split (drop_last), validation loader = rehearsal validation split + the new 10 % (batch 100), then the buffer
  is re-cut to `buffer_size * count_c / total` exemplars per class, 9:1 again, keeping the OLDEST entries of each class.

The loss terms are one fused node (`ops.classify_loss`: CE slice + KD kernels); the bias correction itself is two tiny
element-wise torch ops on the [B, num_class] logits whose autograd yields d(alpha), d(beta) in stage 2.
"""
import copy
from collections import Counter

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .. import optim as fused_optim
from .heads import HipLinear

__all__ = ["bic", "BiasLayer"]

_T = 2.0            # bic.py:197
_VAL_RATIO = 0.1    # bic.py:247


class BiasLayer(nn.Module):
    """y = alpha * x + beta (backbone/resnet.py:579-587)"""

    def __init__(self):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones(1))
        self.beta = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        return self.alpha * x + self.beta


class Model(nn.Module):
    def __init__(self, backbone, num_class, device=None):
        super().__init__()
        self.backbone, self.num_class = backbone, num_class
        self.classifier = HipLinear(backbone.feat_dim, num_class)

    def forward(self, x):
        f = self.backbone(x)
        return self.classifier(f["features"] if isinstance(f, dict) else f)


def classwise_spilt(images, labels, test_size):
    """per class (ascending label): shuffle its positions with the global numpy RNG, first int(n * (1 - test_size)) go to the train
    side -- at least one when the class has more than one sample (bic.py:26-57)"""
    images, labels = np.array(images), np.array(labels)
    tr_i, tr_l, va_i, va_l = [], [], [], []
    for c in np.unique(labels):
        pos = np.where(labels == c)[0]
        np.random.shuffle(pos)
        cut = int(len(pos) * (1 - test_size))
        if cut == 0 and len(pos) > 1:
            cut = 1
        tr_i.extend(images[pos[:cut]]); tr_l.extend(labels[pos[:cut]])
        va_i.extend(images[pos[cut:]]); va_l.extend(labels[pos[cut:]])
    return tr_i, va_i, tr_l, va_l


class bic(nn.Module):
    def __init__(self, backbone, num_class, **kwargs):
        super().__init__()
        self.device = kwargs["device"]
        self.task_num = kwargs["task_num"]
        self.bias_layers = nn.ModuleList([BiasLayer().to(self.device) for _ in range(self.task_num)])
        self.bias_optimizer = fused_optim.Adam([q for layer in self.bias_layers for q in layer.parameters()], lr=1e-3)
        self.model = Model(backbone, num_class, self.device)
        self.init_cls_num, self.inc_cls_num = kwargs["init_cls_num"], kwargs["inc_cls_num"]
        self.seen_cls = self.cur_task = 0
        self.previous_model = None
        self.criterion = nn.CrossEntropyLoss()
        self.cls_count = {}
        self._last_aux = None

    # ------------------------------------------------------------------------------------------ hooks
    def before_task(self, task_idx, buffer, train_loader, test_loaders):
        self.previous_model = copy.deepcopy(self.model)
        for q in self.previous_model.parameters():
            q.requires_grad_(False)
        for q in self.model.parameters():
            q.requires_grad_(True)
        for q in self.bias_layers.parameters():
            q.requires_grad_(False)
        self.cur_task = task_idx
        self.seen_cls += self.init_cls_num if task_idx == 0 else self.inc_cls_num

    def after_task(self, task_idx, buffer, train_loader, test_loaders):
        for q in self.model.parameters():
            q.requires_grad_(False)
        for i, layer in enumerate(self.bias_layers):
            for q in layer.parameters():
                q.requires_grad_(i == task_idx)

    # ------------------------------------------------------------------------------------------ logits
    def _slices(self):
        """[lo, hi) of every task's logits (bic.py:132-136)"""
        return [(0, self.init_cls_num) if i == 0 else (self.init_cls_num + (i - 1) * self.inc_cls_num, self.init_cls_num + i * self.inc_cls_num)
                for i in range(self.task_num)]

    def bias_forward(self, input, train=True):
        """every task's slice through its own layer; columns past the last slice are dropped, as in the reference's `cat`"""
        spans = self._slices()
        width = spans[-1][1]
        a = torch.cat([layer.alpha.expand(hi - lo) for layer, (lo, hi) in zip(self.bias_layers, spans)])
        b = torch.cat([layer.beta.expand(hi - lo) for layer, (lo, hi) in zip(self.bias_layers, spans)])
        return input[:, :width] * a + b

    def _xy(self, data):
        return data["image"].to(self.device), data["label"].view(-1).to(self.device)

    def inference(self, data):
        x, y = self._xy(data)
        p = self.bias_forward(self.model(x), train=False)
        pred, correct = ops.predict(p, y, pred_hi=self.seen_cls)
        return pred, correct.item() / x.size(0)

    # ------------------------------------------------------------------------------------------ stage 1
    def stage1(self, data):
        x, y = self._xy(data)
        p = self.bias_forward(self.model(x))
        aux = ops.LossAux()
        loss = ops.classify_loss(p, y, lo=0, hi=self.seen_cls, pred_hi=self.seen_cls, aux=aux)
        self._last_aux = aux
        return aux.pred, aux.acc(), loss

    def stage1_distill(self, data):
        x, y = self._xy(data)
        old = self.seen_cls - self.inc_cls_num
        alpha = 1.0 * old / self.seen_cls
        assert 1.0 * self.cur_task / (self.cur_task + 1) == alpha
        teacher = ops.TeacherPass(x, lambda: self.bias_forward(self.previous_model(x), train=True))      # its first `old` columns are read
        p = self.bias_forward(self.model(x))
        aux = ops.LossAux()
        loss = ops.classify_loss(p, y, lo=0, hi=self.seen_cls, pred_hi=self.seen_cls, w_ce=1 - alpha, teacher=teacher.result(), k=old, T=_T,
                                 w_kd=alpha * _T * _T, aux=aux)
        self._last_aux = aux
        return aux.pred, aux.acc(), loss

    def observe(self, data):
        return self.stage1_distill(data) if self.cur_task > 0 else self.stage1(data)

    # ------------------------------------------------------------------------------------------ stage 2
    def stage2(self, data):
        x, y = self._xy(data)
        p = self.bias_forward(self.model(x))
        aux = ops.LossAux()
        loss = ops.classify_loss(p, y, lo=0, hi=self.seen_cls, pred_hi=self.seen_cls, aux=aux)
        self.bias_optimizer.zero_grad()
        loss.backward()
        self.bias_optimizer.step()
        self._last_aux = aux
        return aux.pred, aux.acc(), loss

    def get_parameters(self, config):
        return self.model.parameters()

    # ------------------------------------------------------------------------------------------ data
    def spilt_and_update(self, dataloader, buffer, task_idx, config):
        from ..data import make_loader
        cap = config["buffer"]["kwargs"]["buffer_size"]
        train_ds, val_ds = copy.deepcopy(dataloader.dataset), copy.deepcopy(dataloader.dataset)
        self.cls_count.update(Counter(train_ds.labels))
        tr_i, va_i, tr_l, va_l = classwise_spilt(train_ds.images, train_ds.labels, _VAL_RATIO)
        train_ds.images, train_ds.labels = tr_i + buffer.train_images, tr_l + buffer.train_labels
        dev = self.device if config.get("gpu_input_pipeline", True) else None
        train_loader = make_loader(train_ds, config["batch_size"], True, config["num_workers"], dev, drop_last=True)
        val_loader = None
        if task_idx > 0:
            val_ds.images, val_ds.labels = list(buffer.val_images) + list(va_i), list(buffer.val_labels) + list(va_l)
            val_loader = make_loader(val_ds, 100, True, config["num_workers"], dev)
        # the buffer takes everything, then is re-cut class by class to its share of the capacity (oldest entries first)
        buffer.train_images.extend(tr_i); buffer.train_labels.extend(tr_l)
        buffer.val_images.extend(va_i); buffer.val_labels.extend(va_l)
        buffer.total_classes += config["init_cls_num"] if task_idx == 0 else config["inc_cls_num"]
        seen = sum(self.cls_count.values())
        bt_i, bt_l = np.array(buffer.train_images), np.array(buffer.train_labels)
        bv_i, bv_l = np.array(buffer.val_images), np.array(buffer.val_labels)
        keep = {"ti": [], "tl": [], "vi": [], "vl": []}
        for c in range(buffer.total_classes):
            n_val = int(self.cls_count[c] * cap / seen * _VAL_RATIO)
            n_train = int(self.cls_count[c] * cap / seen * (1 - _VAL_RATIO))
            if n_val == 0 and n_train > 1:
                n_val, n_train = 1, n_train - 1
            t_pos, v_pos = np.where(bt_l == c)[0][:n_train], np.where(bv_l == c)[0][:n_val]
            keep["ti"].extend(bt_i[t_pos]); keep["tl"].extend(bt_l[t_pos])
            keep["vi"].extend(bv_i[v_pos]); keep["vl"].extend(bv_l[v_pos])
        buffer.train_images, buffer.train_labels, buffer.val_images, buffer.val_labels = keep["ti"], keep["tl"], keep["vi"], keep["vl"]
        return train_loader, val_loader
