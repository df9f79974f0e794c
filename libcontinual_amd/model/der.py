"""DER plugin -- dynamically expandable representation (reference core/model/der.py:66-226) on the HIP hot path.

One ResNet-18 (CIFAR stem) is added per task, initialised from the previous one; all earlier extractors are frozen.  A step runs
every extractor's plan forward (frozen ones without saved-activation backward), concatenates the 512-d features, and trains the
newest extractor + a full head over all seen classes + an auxiliary head (new classes vs "old") on the newest features:
loss = CE(logits, y) + CE(aux_logits, max(y - known + 1, 0)).  Both CE terms are the fused ce_slice kernel; the only torch ops
in the step are the feature concat / slice.

Reference quirks kept: the frozen extractors are nn.Module children, so the trainer's
`model.train()` puts their BatchNorm back on batch statistics (their running stats keep drifting); `_train()` is never called
by the trainer; the extractor type is fixed to resnet18 whatever backbone the YAML names; `weight_align` exists but nobody calls it.
"""
import torch
import torch.nn as nn

from .. import ops
from .backbone.resnet import resnet18, resnet34
from .finetune import Finetune
from .heads import HipLinear


def get_convnet(convnet_type, pretrained=False):
    name = convnet_type.lower()
    if name == "resnet18":
        return resnet18(num_classes=10, args={"dataset": "cifar100"})
    if name == "resnet34":
        return resnet34(num_classes=10, args={"dataset": "cifar100"})
    raise NotImplementedError(f"Unknown type {convnet_type}")


class SimpleLinear(HipLinear):
    """der.py:43-64: kaiming-uniform (linear gain) weight, zero bias, dict output"""

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, nonlinearity="linear")
        nn.init.constant_(self.bias, 0)

    def forward(self, input):
        return {"logits": super().forward(input)}


class DER(Finetune):
    cuda_graph_safe = False     # not audited for trainer.GraphedStep
    def __init__(self, backbone, feat_dim, num_class, **kwargs):
        super().__init__(backbone, feat_dim, num_class, **kwargs)
        self.convnets = nn.ModuleList()
        self.pretrained = None
        self.out_dim = self.fc = self.aux_fc = None
        self.task_sizes = []
        self.init_cls_num, self.inc_cls_num = kwargs["init_cls_num"], kwargs["inc_cls_num"]
        self.known_cls_num = self.total_cls_num = 0
        self.convnet_type = "resnet18"

    @property
    def feature_dim(self):
        return 0 if self.out_dim is None else self.out_dim * len(self.convnets)

    def _features(self, x):
        """frozen extractors have no backward: their forwards go to the second stream (ops.TeacherPass) and overlap the trainable one's"""
        nets = list(self.convnets)
        frozen = [n for n in nets[:-1] if not n._params[0].requires_grad]
        if not frozen or len(frozen) != len(nets) - 1 or not torch.is_grad_enabled():
            return torch.cat([net(x)["features"] for net in nets], 1)
        old = ops.TeacherPass(x, lambda: [net(x)["features"] for net in frozen])
        new = nets[-1](x)["features"]
        return torch.cat(list(old.result()) + [new], 1)

    def forward(self, x):
        features = self._features(x)
        out = self.fc(features)
        out.update(aux_logits=self.aux_fc(features[:, -self.out_dim:])["logits"], features=features)
        return out

    def observe(self, data):
        x, y = self._xy(data)
        features = self._features(x)
        aux = ops.LossAux()
        loss = ops.classify_loss(self.fc(features)["logits"], y, aux=aux)
        if self.task_idx > 0:
            # auxiliary head over the newest extractor's features: class 0 = "any old class", 1.. = the task's classes
            aux_y = torch.clamp(y - self.known_cls_num + 1, min=0)
            loss = ops.classify_loss(self.aux_fc(features[:, -self.out_dim:].contiguous())["logits"], aux_y) + loss
        self._last_aux = aux
        return aux.pred, aux.acc(), loss

    def inference(self, data):
        x, y = self._xy(data)
        pred, correct = ops.predict(self.fc(self._features(x))["logits"], y)
        return pred, correct.item() / x.size(0)

    def update_fc(self, nb_classes):
        """append an extractor (a copy of the last one), widen `fc` over the longer feature vector keeping the learned block,
        start a fresh auxiliary head (der.py:150-174)"""
        self.convnets.append(get_convnet(self.convnet_type))
        if len(self.convnets) > 1:
            self.convnets[-1].load_state_dict(self.convnets[-2].state_dict())
        if self.out_dim is None:
            self.out_dim = self.convnets[-1].out_dim
        fc = self.generate_fc(self.feature_dim, nb_classes)
        if self.fc is not None:
            rows = self.fc.out_features
            with torch.no_grad():
                fc.weight[:rows, : self.feature_dim - self.out_dim] = self.fc.weight.to(fc.weight.device)
                fc.bias[:rows] = self.fc.bias.to(fc.bias.device)
        self.fc = fc
        self.task_sizes.append(nb_classes - sum(self.task_sizes))
        self.aux_fc = self.generate_fc(self.out_dim, self.task_sizes[-1] + 1)

    def generate_fc(self, in_dim, out_dim):
        return SimpleLinear(in_dim, out_dim)

    def freeze_convnets(self):
        for q in self.convnets.parameters():
            q.requires_grad = False
        self.convnets.eval()

    @torch.no_grad()
    def weight_align(self, increment):
        w = self.fc.weight
        norms = w.float().norm(dim=1)
        gamma = norms[:-increment].mean() / norms[-increment:].mean()
        print("alignweights,gamma=", gamma)
        w[-increment:] *= gamma

    def before_task(self, task_idx, buffer, train_loader, test_loaders):
        self.task_idx = task_idx
        self.known_cls_num, self.total_cls_num = self.total_cls_num, self.init_cls_num + task_idx * self.inc_cls_num
        self.freeze_convnets()
        self.update_fc(self.total_cls_num)
        self.loss_fn = nn.CrossEntropyLoss()
        self.convnets, self.fc, self.aux_fc = self.convnets.to(self.device), self.fc.to(self.device), self.aux_fc.to(self.device)

    def _train(self):
        self.fc.train()
        self.aux_fc.train()
        for i in range(self.task_idx - 1):
            self.convnets[i].eval()
        self.convnets[-1].train()

    def get_parameters(self, config):
        groups = [{"params": self.convnets.parameters()}]
        groups += [{"params": head.parameters()} for head in (self.fc, self.aux_fc) if head is not None]
        return groups
