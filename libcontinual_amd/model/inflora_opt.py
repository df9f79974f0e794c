"""InfLoRA_OPT plugin (reference core/model/InfLoRA_opt.py:46-460, ViT branch) on the HIP ViT executor.

Same constructor kwargs, hooks and quirks: every task re-zeroes all `lora_B`, fixes
`lora_A` from the SVD of the (DualGPM-projected) input Gram, trains all `lora_B*` + the task's head
(`get_parameters` returns ALL network parameters; frozen ones keep `grad None` and are skipped by the optimizer),
labels offset by the known classes, CE on the current task's head only; after the task the LoRA branch is merged into
the qkv weight and the DualGPM bases are updated.

Hot loop (observe / backward / step) = HIP: effective-qkv refresh, one backbone forward, head + CE, one backbone
backward producing the 24 dB matrices through the rank-r shortcut.  Per-task host logic (SVD of 768x768 Grams,
DualGPM thresholds) stays on the host in torch / numpy exactly where the reference runs it (InfLoRA_opt.py:251-369);
the Gram itself (X^T X per layer, transformer.py:241-244) is accumulated on the device by the executor.
The CLIP branch and classifier alignment (`use_ca`) are outside the hot-path scope.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..utils import device_svd
from .backbone.vit import MultiHeadAttention_LoRA, ViTZoo
from .heads import HipLinear


class SiNet(nn.Module):
    def __init__(self, backbone, device, **kwargs):
        super().__init__()
        self._cur_task_id = -1
        self.backbone = backbone
        self.device = device
        if not isinstance(backbone, ViTZoo):
            raise NotImplementedError("only the ViT backbone is on the hot path (SURVEY.md section 8)")
        self.classifier_pool = nn.ModuleList(
            [HipLinear(kwargs["embd_dim"], kwargs["init_cls_num"], bias=True)] +
            [HipLinear(kwargs["embd_dim"], kwargs["inc_cls_num"], bias=True) for _ in range(kwargs["task_num"] - 1)])

    def update_fc(self, train_loader):
        self._cur_task_id += 1

    def get_feature(self, x):
        return self.backbone(x)

    def fc_only(self, x):
        return torch.cat([fc(x) for fc in self.classifier_pool[: self._cur_task_id + 1]], dim=1)

    def forward(self, x, inference=False):
        features = self.backbone(x)
        heads = self.classifier_pool[: self._cur_task_id + 1] if inference else [self.classifier_pool[self._cur_task_id]]
        return torch.cat([fc(features) for fc in heads], dim=1)

    def update_input_matrix(self, x):
        with torch.no_grad():
            self.backbone(x, get_input_matrix=True)


class InfLoRA_OPT(nn.Module):
    def __init__(self, backbone, device, **kwargs):
        super().__init__()
        self.device = device
        self.init_cls_num = kwargs["init_cls_num"]
        self.inc_cls_num = kwargs["inc_cls_num"]
        self.task_num = kwargs["task_num"]
        self.lame = kwargs["lame"]
        self.lamb = kwargs["lamb"]
        self._known_classes = 0
        self.feature_list = []
        self.project_type = []
        self._dataset = kwargs.get("dataset")
        if kwargs.get("use_ca", False):
            raise NotImplementedError("classifier alignment (use_ca) is outside the hot-path scope (SURVEY.md section 8)")
        self._network = SiNet(backbone, device, **kwargs).to(self.device)
        self.attention_modules = [m for m in self._network.modules() if isinstance(m, MultiHeadAttention_LoRA)]

    def observe(self, data):
        x, y = data["image"].to(self.device), data["label"].to(self.device) - self._known_classes
        logits = self._network(x)
        aux = ops.LossAux()
        loss = ops.classify_loss(logits, y, aux=aux)
        self._last_aux = aux
        return aux.pred, aux.acc(), loss

    def inference(self, data):
        x, y = data["image"].to(self.device), data["label"].to(self.device)
        with torch.no_grad():
            logits = self._network(x, inference=True)
        pred, correct = ops.predict(logits, y)
        return pred, correct.item() / y.size(0)

    @torch.no_grad()
    def before_task(self, task_idx, buffer, train_loader, test_loaders):
        if task_idx == 1:
            self._known_classes = self.init_cls_num
        elif task_idx > 1:
            self._known_classes += self.inc_cls_num
        self._network.update_fc(train_loader)
        for module in self.attention_modules:
            module.init_param()
        for name, param in self._network.named_parameters():
            param.requires_grad_(False)
            if f"classifier_pool.{task_idx}." in name or "lora_B" in name:
                param.requires_grad_(True)
        for batch in train_loader:
            self._network.update_input_matrix(x=batch["image"].to(self.device))
        for i, module in enumerate(self.attention_modules):
            assert module.n_cur_matrix > 0
            cur_matrix = module.cur_matrix
            if task_idx > 0:
                assert self.project_type[i] in ("remove", "retain")
                feature_mat = torch.as_tensor(self.feature_list[i] @ self.feature_list[i].T, dtype=cur_matrix.dtype)
                cur_matrix = cur_matrix - feature_mat @ cur_matrix if self.project_type[i] == "remove" else feature_mat @ cur_matrix
            U = torch.from_numpy(device_svd(cur_matrix, full_matrices=False)[0]).to(cur_matrix.dtype)      # fp64 on the GPU (utils.device_svd)
            A = (U[:, : module.lora_rank].T / math.sqrt(3)).to(module.lora_A_k.weight)
            module.lora_A_k.weight.copy_(A)          # in-place on the Parameter (not .data): bumps its version, which the
            module.lora_A_v.weight.copy_(A)          # executor watches to refresh its [A_k; A_v] copy
            module.reset_input_matrix()

    def after_task(self, task_idx, buffer, train_loader, test_loaders):
        for module in self.attention_modules:
            module.merge_weight()
        self._update_feature(task_idx, train_loader, None)

    @torch.no_grad()
    def _update_feature(self, task_idx, train_loader, test_trfms):
        """DualGPM update of the per-layer bases (InfLoRA_opt.py:290-369), host side as in the reference"""
        for batch in train_loader:
            self._network.update_input_matrix(x=batch["image"].to(self.device))
        threshold = (self.lame - self.lamb) * task_idx / self.task_num + self.lamb
        for i, module in enumerate(self.attention_modules):
            activation = module.cur_matrix.numpy().astype(np.float64)
            if task_idx == 0:
                U, S, _ = device_svd(activation, full_matrices=False)
                ratio = (S ** 2) / (S ** 2).sum()
                r = max(np.sum(np.cumsum(ratio) < threshold), 1)
                assert r < activation.shape[0] / 2
                self.feature_list.append(U[:, :r])
                self.project_type.append("remove")
            else:
                _, S, _ = device_svd(activation, full_matrices=False)
                total = (S ** 2).sum()
                fm = self.feature_list[i] @ self.feature_list[i].T
                if self.project_type[i] == "remove":
                    U, S, _ = device_svd(activation - fm @ activation, full_matrices=False)
                    ratio = (S ** 2) / total
                    acc = (total - (S ** 2).sum()) / total
                    if acc < threshold:
                        r = np.sum(np.cumsum(ratio) + acc < threshold) + 1
                        Ui = np.hstack((self.feature_list[i], U[:, :r]))
                        self.feature_list[i] = Ui[:, : min(Ui.shape[0], Ui.shape[1])]
                else:
                    U, S, _ = device_svd(fm @ activation, full_matrices=False)
                    ratio = (S ** 2) / total
                    acc = (S ** 2).sum() / total
                    if acc >= 1 - threshold:
                        r = np.sum(acc - np.cumsum(ratio) >= 1 - threshold) + 1
                        af = self.feature_list[i] - U[:, :r] @ U[:, :r].T @ self.feature_list[i]
                        U, _, _ = device_svd(af, full_matrices=True)
                        self.feature_list[i] = U[:, : self.feature_list[i].shape[1] - r]
            module.reset_input_matrix()
        for i in range(len(self.feature_list)):
            f = self.feature_list[i]
            if self.project_type[i] == "remove" and f.shape[1] > f.shape[0] / 2:
                U, _, _ = device_svd(f, full_matrices=True)
                self.feature_list[i] = U[:, f.shape[1]:]
                self.project_type[i] = "retain"
            elif self.project_type[i] == "retain":
                assert f.shape[1] <= f.shape[0] / 2

    def get_parameters(self, config):
        return self._network.parameters()
