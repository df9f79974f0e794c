"""InfLoRA plugin, the original multi-branch form (reference core/model/InfLoRA.py:36-317) on the HIP ViT executor.

One rank-r LoRA pair per task on k and v (backbone `SiNet_vit`): before a task the Gram of every attention layer's input is
accumulated over the task's data (on the device, by the executor), projected with the DualGPM bases and its top-r left
singular vectors become that task's `lora_A`; only `lora_B_{k,v}.{task}` and `classifier_pool.{task}` train; after the task the
bases are updated.  Differences to InfLoRA_OPT (model/inflora_opt.py) the fixtures pin: no merge into the qkv weight (all pairs
stay, the forward sums them), first-task rank r = #{cumulative ratio < threshold} without the +1, a first-task layer may start
as 'retain', inputs of the Gram passes are bilinearly resized to 224 (InfLoRA.py:147, 194), and inference concatenates all
heads on ONE feature vector computed with the pairs of all tasks so far.
Hot loop (observe / backward / step) = HIP; SVDs and thresholds stay host-side numpy / torch, where the reference runs them.
"""
import math
from copy import deepcopy

import numpy as np
import torch
import torch.nn.functional as F

from .. import ops
from .backbone.sinet import Attention_LoRA
from .finetune import Finetune
from ..utils import device_svd


class InfLoRA(Finetune):
    cuda_graph_safe = False     # not audited for trainer.GraphedStep
    def __init__(self, backbone, feat_dim, num_class, **kwargs):
        super().__init__(backbone, feat_dim, num_class, **kwargs)
        self._network = backbone
        self._attn = [m for m in self._network.modules() if isinstance(m, Attention_LoRA)]
        for module in self._attn:
            module.init_param()
        self.num_class = num_class
        self._total_classes = self._known_classes = 0
        self._cur_task = -1
        self.inc_cls_num = kwargs["inc_cls_num"]
        self.feature_list, self.project_type, self.feature_mat = [], [], []
        self.lame, self.lamb, self.total_sessions = kwargs["lame"], kwargs["lamb"], kwargs["total_sessions"]
        self.gram_size = kwargs.get("gram_size", 224)          # InfLoRA.py:147 hard-codes 224 (the ViT-B/16 input)

    def observe(self, data):
        x, y = self._xy(data)
        logits = self._network(x)["logits"]
        aux = ops.LossAux()
        loss = ops.classify_loss(logits, y - self._known_classes, aux=aux)       # the task head predicts 0 .. inc-1 (InfLoRA.py:79)
        self._last_aux = aux
        return aux.pred, aux.acc(), loss

    def inference(self, data):
        x, y = self._xy(data)
        with torch.no_grad():
            pred, correct = ops.predict(self._network.interface(x), y)
        return pred, correct.item() / y.size(0)

    # ------------------------------------------------------------------------------------------------ per-task host logic
    @torch.no_grad()
    def _accumulate_gram(self, train_loader):
        for batch in train_loader:
            x = batch["image"].to(self.device)
            if x.shape[-1] != self.gram_size:
                x = F.interpolate(x.float(), size=self.gram_size, mode="bilinear", align_corners=False)
            self._network(x, get_cur_feat=True)

    @torch.no_grad()
    def before_task(self, task_idx, buffer, train_loader, test_loaders):
        self._known_classes = self._total_classes
        self._cur_task += 1
        self._total_classes = self._known_classes + self.inc_cls_num
        self._network.update_fc(self._total_classes)
        self._network.to(self.device)
        t = self._network.numtask - 1
        for name, param in self._network.named_parameters():
            param.requires_grad_(any(f"{key}.{t}." in name for key in ("classifier_pool", "lora_B_k", "lora_B_v")))
        self._accumulate_gram(train_loader)
        for kk, module in enumerate(self._attn):
            cur = module.cur_matrix
            if self._cur_task > 0:
                fm = self.feature_mat[kk].to(cur.dtype)
                cur = cur - fm @ cur if self.project_type[kk] == "remove" else fm @ cur
            U = torch.from_numpy(device_svd(cur, full_matrices=self._cur_task == 0)[0]).to(cur.dtype)            # fp64 on the GPU (utils.device_svd)
            A = (U[:, : module.rank].T / math.sqrt(3)).to(module.lora_A_k[self._cur_task].weight)
            module.lora_A_k[self._cur_task].weight.copy_(A)
            module.lora_A_v[self._cur_task].weight.copy_(A)
            module.cur_matrix.zero_()
            module.n_cur_matrix = 0

    @torch.no_grad()
    def after_task(self, task_idx, buffer, train_loader, test_loaders):
        self._accumulate_gram(train_loader)
        mats = []
        for module in self._attn:
            mats.append(deepcopy(module.cur_matrix))
            module.cur_matrix.zero_()
            module.n_cur_matrix = 0
        self.update_DualGPM(mats)
        self.feature_mat = [torch.Tensor(f @ f.T) for f in self.feature_list]

    def update_DualGPM(self, mat_list):
        """per-layer bases of the directions to remove from / retain in the next task's Gram (InfLoRA.py:213-308)"""
        threshold = (self.lame - self.lamb) * self._cur_task / self.total_sessions + self.lamb
        first = len(self.feature_list) == 0
        for i, activation in enumerate(np.asarray(m) for m in mat_list):
            if first:
                U, S, _ = device_svd(activation, full_matrices=False)
                r = int(np.sum(np.cumsum(S ** 2 / (S ** 2).sum()) < threshold))
                self.feature_list.append(U[:, : max(r, 1)])
                self.project_type.append("remove" if r < activation.shape[0] / 2 else "retain")
                continue
            total = (device_svd(activation, compute_uv=False) ** 2).sum()
            f = self.feature_list[i]
            proj = f @ (f.T @ activation)
            if self.project_type[i] == "remove":
                U, S, _ = device_svd(activation - proj, full_matrices=False)
                ratio, acc, r = S ** 2 / total, (total - (S ** 2).sum()) / total, 0
                while r < ratio.shape[0] and acc < threshold:
                    acc += ratio[r]
                    r += 1
                if r:
                    Ui = np.hstack((f, U[:, :r]))
                    self.feature_list[i] = Ui[:, : Ui.shape[0]] if Ui.shape[1] > Ui.shape[0] else Ui
            else:
                U, S, _ = device_svd(proj, full_matrices=False)
                ratio, acc, r = S ** 2 / total, (S ** 2).sum() / total, 0
                while r < ratio.shape[0] and acc >= 1 - threshold:
                    acc -= ratio[r]
                    r += 1
                if r:
                    rest = f - U[:, :r] @ (U[:, :r].T @ f)
                    self.feature_list[i] = device_svd(rest, full_matrices=True)[0][:, : f.shape[1] - r]
        for i, f in enumerate(self.feature_list):
            if self.project_type[i] == "remove" and f.shape[1] > f.shape[0] / 2:
                self.feature_list[i] = device_svd(f, full_matrices=True)[0][:, f.shape[1]:]
                self.project_type[i] = "retain"
            elif self.project_type[i] == "retain":
                assert f.shape[1] <= f.shape[0] / 2

    def get_parameters(self, config):
        return [{"params": self.backbone.parameters()}, {"params": self.classifier.parameters()}]
