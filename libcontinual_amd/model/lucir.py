"""LUCIR plugin (reference core/model/lucir.py:56-239) on the HIP hot path.

Cosine classifier, less-forget constraint (CosineEmbeddingLoss against the frozen previous model's
features), CE, and hard-negative margin ranking on the pre-sigma scores; fc2 of the split head is
imprinted from class-mean features in before_task; old-class embedding fc1 frozen through an lr=0
parameter group (lucir.py:229-236).  The reference's module-level forward hooks (lucir.py:33-51,
125-128) only capture (features, pre-sigma scores); here the heads expose them as `last_scores`.
Every loss term is a libclhip kernel (cosine_linear, sigma_scale, cos_embed_loss, ce_slice,
margin_rank_loss); there is no host sync for the data-dependent hard-sample count.
"""
import copy
import math

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import DataLoader

from .. import ops
from .backbone.resnet import CosineLinear, SplitCosineLinear, _Scratch, _ScratchMixin
from .finetune import Finetune


class Model(_ScratchMixin, nn.Module):
    def __init__(self, backbone, feat_dim, num_class):
        super().__init__()
        self._scratch = _Scratch()
        self.backbone = backbone
        self.feat_dim = feat_dim
        self.num_class = num_class
        self.classifier = CosineLinear(feat_dim, num_class)

    def forward(self, x):
        return self.get_logits(x)

    def get_logits(self, x):
        self._scratch.features = self.backbone(x)["features"]
        return self.classifier(self._scratch.features)


class LUCIR(Finetune):
    def __init__(self, backbone, feat_dim, num_class, **kwargs):
        super().__init__(backbone, feat_dim, num_class, **kwargs)
        self.kwargs = kwargs
        self.network = Model(self.backbone, feat_dim, kwargs["init_cls_num"])
        self.K = kwargs["K"]
        self.lw_mr = kwargs["lw_mr"]
        self.ref_model = None
        self.task_idx = 0

    def before_task(self, task_idx, buffer, train_loader, test_loaders):
        self.task_idx = task_idx
        net = self.network
        inc = self.kwargs["inc_cls_num"]
        if task_idx == 1:
            self.ref_model = copy.deepcopy(net)
            old = net.classifier
            new_fc = SplitCosineLinear(old.in_features, old.out_features, inc).to(old.weight.device)
            new_fc.fc1.weight.data = old.weight.data
            new_fc.sigma.data = old.sigma.data
            net.classifier = new_fc
            lamda_mult = old.out_features * 1.0 / inc
        elif task_idx > 1:
            self.ref_model = copy.deepcopy(net)
            old = net.classifier
            o1, o2 = old.fc1.out_features, old.fc2.out_features
            new_fc = SplitCosineLinear(old.in_features, o1 + o2, inc).to(self.device)
            new_fc.fc1.weight.data[:o1] = old.fc1.weight.data
            new_fc.fc1.weight.data[o1:] = old.fc2.weight.data
            new_fc.sigma.data = old.sigma.data
            net.classifier = new_fc
            lamda_mult = (o1 + o2) * 1.0 / inc
        if task_idx > 0:
            self.cur_lamda = self.kwargs["lamda"] * math.sqrt(lamda_mult)     # lucir.py:110
        else:
            self.cur_lamda = self.kwargs["lamda"]
        self._init_new_fc(task_idx, buffer, train_loader)
        if task_idx > 0:
            self.ref_model.eval()
            self.num_old_classes = self.ref_model.classifier.out_features
        self.network = self.network.to(self.device)
        if self.ref_model is not None:
            self.ref_model = self.ref_model.to(self.device)

    def _init_new_fc(self, task_idx, buffer, train_loader):
        """imprint fc2 from normalised class-mean features times the mean old-embedding norm (lucir.py:134-159)"""
        if task_idx == 0:
            return
        cls = self.network.classifier
        old_norm = cls.fc1.weight.data.norm(dim=1, keepdim=True)
        avg_old = torch.mean(old_norm, dim=0).to("cpu").type(torch.DoubleTensor)
        nfeat = cls.in_features
        novel = torch.zeros((self.kwargs["inc_cls_num"], nfeat))
        tmp = copy.deepcopy(train_loader.dataset)
        data, target = train_loader.dataset.images, train_loader.dataset.labels
        for cls_idx in range(cls.fc1.out_features, cls.fc1.out_features + cls.fc2.out_features):
            ind = np.where(np.array(target) == cls_idx)[0]
            tmp.images = np.array([data[i] for i in ind])
            tmp.labels = np.array([target[i] for i in ind])
            loader = DataLoader(tmp, batch_size=128, shuffle=False, num_workers=0)
            feats = self._compute_feature(self.network.backbone, loader, len(ind), nfeat)
            nf = torch.nn.functional.normalize(torch.from_numpy(feats), p=2, dim=1)
            emb = torch.mean(nf, dim=0)
            novel[cls_idx - cls.fc1.out_features] = torch.nn.functional.normalize(emb, p=2, dim=0) * avg_old
        self.network.to(self.device)
        cls.fc2.weight.data = novel.to(self.device)

    def _compute_feature(self, feature_model, loader, num_samples, num_features):
        feature_model.eval()
        feats = np.zeros([num_samples, num_features])
        s = 0
        with torch.no_grad():
            for batch in loader:
                x = batch["image"].to(self.device)
                feats[s:s + x.shape[0], :] = feature_model.feature(x).cpu().numpy()
                s += x.shape[0]
        assert s == num_samples
        return feats

    def observe(self, data):
        x, y = self._xy(data)
        logit = self.network(x)
        aux = ops.LossAux()
        loss = ops.classify_loss(logit, y, aux=aux)                                   # CE over all seen classes
        if self.task_idx > 0:
            with torch.no_grad():
                self.ref_model(x)
                ref_features = self.ref_model.last_features
            cur_features = self.network.last_features
            loss = loss + ops.cos_embed_loss(cur_features, ref_features, self.cur_lamda)          # lucir.py:182-183
            scores = self.network.classifier.last_scores                                          # pre-sigma, all classes
            loss = loss + ops.margin_rank_loss(scores, y, self.num_old_classes, self.K, self.kwargs["dist"], self.lw_mr)
        self._last_aux = aux
        return aux.pred, aux.acc(), loss

    def after_task(self, task_idx, buffer, train_loader, test_loaders):
        pass

    def inference(self, data):
        x, y = self._xy(data)
        logit = self.network(x)
        pred, correct = ops.predict(logit, y)
        return pred, correct.item() / x.size(0)

    def get_parameters(self, config):
        if self.task_idx > 0:
            ignored = list(map(id, self.network.classifier.fc1.parameters()))
            base = filter(lambda p: id(p) not in ignored, self.network.parameters())
            return [{"params": base, "lr": 0.1, "weight_decay": 5e-4},
                    {"params": self.network.classifier.fc1.parameters(), "lr": 0, "weight_decay": 0}]
        return self.network.parameters()
