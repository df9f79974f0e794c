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
    # round 4: audited for trainer.GraphedStep -- observe() has no host synchronisation (the hard-sample count of the margin ranking loss stays on the
    # device), the frozen model's pass forks / joins a side stream inside the capture like iCaRL's; eight replayed steps equal eight eager ones bit for
    # bit (tests/test_graph_step_gpu.py::test_lucir_replays_like_eager).  At 32 images per GPU: 1.33 -> 0.91 ms per step.
    cuda_graph_safe = True

    @property
    def cuda_graph_auto_max_batch(self):
        """round 5: LUCIR's step (cosine heads, three loss terms, the frozen model's pass) is host-enqueue-bound up to batch 256 on the CIFAR ResNet-32 backbones
        the reference configures it with (1.47-1.54 ms eager, 1.41-1.43 replayed); their plans have no weight-gradient stream, so a capture loses nothing.  Wider
        backbones (feat_dim > 64: ResNet-18's two-stream backward) keep the default cap."""
        return 256 if self.feat_dim <= 64 else 64

    def __init__(self, backbone, feat_dim, num_class, **kwargs):
        super().__init__(backbone, feat_dim, num_class, **kwargs)
        self.kwargs = kwargs
        self.network = Model(self.backbone, feat_dim, kwargs["init_cls_num"])
        self.K = kwargs["K"]
        self.lw_mr = kwargs["lw_mr"]
        self.ref_model = None
        self.task_idx = 0

    # ------------------------------------------------------------------------------------------ head growth
    def _split_head(self, n_new):
        """new SplitCosineLinear whose fc1 holds every embedding learned so far (one block at task 1, fc1 ++ fc2 later) and
        whose fc2 has `n_new` fresh rows; sigma is carried over (lucir.py:84-108).  Returns (head, number of old classes)."""
        old = self.network.classifier
        parts = [old.weight] if isinstance(old, CosineLinear) else [old.fc1.weight, old.fc2.weight]
        carried = torch.cat([w.data for w in parts], dim=0)
        head = SplitCosineLinear(old.in_features, carried.shape[0], n_new).to(carried.device)
        head.fc1.weight.data = carried
        head.sigma.data = old.sigma.data
        return head, carried.shape[0]

    def before_task(self, task_idx, buffer, train_loader, test_loaders):
        self.task_idx = task_idx
        self.cur_lamda = self.kwargs["lamda"]
        if task_idx > 0:
            inc = self.kwargs["inc_cls_num"]
            self.ref_model = copy.deepcopy(self.network)                     # frozen previous model (features + scores)
            self.network.classifier, n_old = self._split_head(inc)
            self.cur_lamda = self.kwargs["lamda"] * math.sqrt(n_old / inc)      # adaptive less-forget weight, lucir.py:110
            self._init_new_fc(task_idx, buffer, train_loader)
            self.ref_model.eval()
            self.num_old_classes = self.ref_model.classifier.out_features
            self.ref_model = self.ref_model.to(self.device)
        self.network = self.network.to(self.device)

    def _class_features(self, dataset, indices):
        """eval-mode backbone features [len(indices), feat_dim] (fp64 numpy) of the given samples, in order"""
        sub = copy.deepcopy(dataset)
        sub.images = np.array([dataset.images[i] for i in indices])
        sub.labels = np.array([dataset.labels[i] for i in indices])
        return self._compute_feature(self.network.backbone, DataLoader(sub, batch_size=128, shuffle=False, num_workers=0), len(indices),
                                     self.network.classifier.in_features)

    def _init_new_fc(self, task_idx, buffer, train_loader):
        """imprint fc2: row c = normalised mean of the L2-normalised features of class c, scaled to the mean norm of the old
        embeddings (lucir.py:134-159); classes are visited in label order like the reference (same RNG consumption)"""
        if task_idx == 0:
            return
        head = self.network.classifier
        first, count = head.fc1.out_features, head.fc2.out_features
        scale = head.fc1.weight.data.norm(dim=1, keepdim=True).mean(dim=0).cpu().double()
        labels = np.asarray(train_loader.dataset.labels)
        rows = []
        for c in range(first, first + count):
            f = torch.from_numpy(self._class_features(train_loader.dataset, np.flatnonzero(labels == c)))
            proto = torch.nn.functional.normalize(f, p=2, dim=1).mean(dim=0)
            rows.append(torch.nn.functional.normalize(proto, p=2, dim=0) * scale)
        self.network.to(self.device)
        head.fc2.weight.data = torch.stack(rows).float().to(self.device)

    def _compute_feature(self, feature_model, loader, num_samples, num_features):
        feature_model.eval()
        chunks = []
        with torch.no_grad():
            for batch in loader:
                chunks.append(feature_model.feature(batch["image"].to(self.device)).cpu().double())
        out = torch.cat(chunks).numpy() if chunks else np.zeros([0, num_features])
        assert out.shape == (num_samples, num_features)
        return out

    def observe(self, data):
        x, y = self._xy(data)

        def ref_pass():
            self.ref_model(x)
            return self.ref_model.last_features
        teacher = ops.TeacherPass(x, ref_pass) if self.task_idx > 0 else None
        logit = self.network(x)
        aux = ops.LossAux()
        loss = ops.classify_loss(logit, y, aux=aux)                                   # CE over all seen classes
        if self.task_idx > 0:
            ref_features = teacher.result()
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
        """task 0: everything at the YAML's settings; later the old-class embeddings fc1 sit in an lr = 0 group and the rest is
        hard-wired to lr 0.1 / wd 5e-4, overriding the YAML (lucir.py:229-236)"""
        if self.task_idx == 0:
            return self.network.parameters()
        frozen = list(self.network.classifier.fc1.parameters())
        held = {id(q) for q in frozen}
        return [{"params": [q for q in self.network.parameters() if id(q) not in held], "lr": 0.1, "weight_decay": 5e-4},
                {"params": frozen, "lr": 0, "weight_decay": 0}]
