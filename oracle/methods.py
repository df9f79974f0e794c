"""CPU restatement of the in-scope continual-learning methods (TEST INFRASTRUCTURE, see oracle/__init__).

Each class mirrors the observable behaviour of one reference plugin *as the reference Trainer drives
it* (`core/trainer.py:563-614`): `model.train()` flips every sub-module -- including frozen teachers
-- to train mode, so teacher BatchNorm uses batch statistics and its running stats drift
(SURVEY.md section 8a quirks a10/a11/a12).  Backbones are the functional nets of `oracle/nets.py`.
"""
import copy
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import nets


def _clone_dict(d, grad=False):
    return {k: v.detach().clone().requires_grad_(grad and v.is_floating_point()) for k, v in d.items()}


def linear_default_init(out_f, in_f, generator=None):
    """nn.Linear default init: U(-1/sqrt(in), 1/sqrt(in)) for weight and bias."""
    b = 1.0 / math.sqrt(in_f)
    w = (torch.rand(out_f, in_f, generator=generator) * 2 - 1) * b
    bias = (torch.rand(out_f, generator=generator) * 2 - 1) * b
    return w, bias


def kd_loss(pred, soft, T=2.0):
    """`-(softmax(soft/T) * log_softmax(pred/T)).sum() / B`; reference lwf.py:75-78, icarl.py:198-206."""
    lp = torch.log_softmax(pred / T, dim=1)
    q = torch.softmax(soft / T, dim=1)
    return -(q * lp).sum() / pred.shape[0]


class Net:
    """backbone + linear head ("Model" of ewc.py:43-57 / icarl.py:24-38)."""

    def __init__(self, arch, P, Bf, head_w, head_b):
        self.arch = arch
        self.P = P            # backbone params (leaf tensors)
        self.Bf = Bf          # BN buffers
        self.head_w = head_w
        self.head_b = head_b

    def features(self, x, train):
        return nets.forward(self.arch, self.P, self.Bf, x, train)

    def logits(self, x, train):
        return F.linear(self.features(x, train), self.head_w, self.head_b)

    def named_parameters(self):
        for k, v in self.P.items():
            yield "backbone." + k, v
        yield "classifier.weight", self.head_w
        yield "classifier.bias", self.head_b

    def parameters(self):
        return [p for _, p in self.named_parameters()]

    def clone(self, grad=True):
        return Net(self.arch, _clone_dict(self.P, grad), _clone_dict(self.Bf),
                   self.head_w.detach().clone().requires_grad_(grad),
                   self.head_b.detach().clone().requires_grad_(grad))


def _acc(logit, y):
    pred = torch.argmax(logit, dim=1)
    return pred, (pred == y).sum().item() / y.shape[0]


# --------------------------------------------------------------------------------------------- EWC
class EWC:
    """reference core/model/ewc.py:59-229."""

    def __init__(self, net, init_cls_num, inc_cls_num, lamda):
        self.network = net                        # head has init_cls_num rows (ewc.py:63)
        self.init_cls_num, self.inc_cls_num, self.lamda = init_cls_num, inc_cls_num, lamda
        self.ref_param = {n: p.detach().clone() for n, p in net.named_parameters()}        # ewc.py:65-66
        self.fisher = {n: torch.zeros_like(p) for n, p in net.named_parameters()}         # ewc.py:67-68
        self.task_idx = 0

    def before_task(self, task_idx, new_rows=None):
        """Grow the head to init + task_idx*inc rows, old rows copied (ewc.py:71-80).
        `new_rows` = (w, b) full-size tensors supplying the fresh rows (default nn.Linear init)."""
        self.task_idx = task_idx
        net = self.network
        out_old = net.head_w.shape[0]
        n_new = self.init_cls_num + task_idx * self.inc_cls_num
        w, b = new_rows if new_rows is not None else linear_default_init(n_new, net.head_w.shape[1])
        w = w.detach().clone(); b = b.detach().clone()
        w[:out_old] = net.head_w.detach(); b[:out_old] = net.head_b.detach()
        net.head_w = w.requires_grad_(True); net.head_b = b.requires_grad_(True)

    def compute_ewc(self):
        """sum_n sum F_n * (p_n[:len(ref_n)] - ref_n)^2 / 2   (ewc.py:221-225)."""
        loss = 0
        for n, p in self.network.named_parameters():
            if n in self.fisher:
                r = self.ref_param[n]
                loss = loss + torch.sum(self.fisher[n] * (p[: len(r)] - r).pow(2)) / 2
        return loss

    def observe(self, x, y, train=True):
        logit = self.network.logits(x, train)
        if self.task_idx == 0:
            loss = F.cross_entropy(logit, y)                                             # ewc.py:87
        else:
            old = self.network.head_w.shape[0] - self.inc_cls_num                        # ewc.py:92
            loss = F.cross_entropy(logit[:, old:], y - old) + self.lamda * self.compute_ewc()   # ewc.py:99-100
        pred, acc = _acc(logit, y)
        return pred, acc, loss

    def inference(self, x, y):
        return _acc(self.network.logits(x, False), y)

    def get_fisher(self, batches, batch_size):
        """ewc.py:147-205: Fisher of the *batch-mean* gradient times len(y), BN in train mode,
        divided by batch_size*len(loader)."""
        fisher = {n: torch.zeros_like(p) for n, p in self.network.named_parameters()}
        for x, y in batches:
            for p in self.network.parameters():
                p.grad = None
            loss = F.cross_entropy(self.network.logits(x, True), y)
            loss.backward()
            for n, p in self.network.named_parameters():
                if p.grad is not None:
                    fisher[n] += p.grad.pow(2) * len(y)
        num = batch_size * len(batches)
        for p in self.network.parameters():
            p.grad = None
        return {n: f / num for n, f in fisher.items()}

    def after_task(self, batches, batch_size):
        """ewc.py:110-133 (ref snapshot BEFORE the Fisher pass; alpha merge even at task 0)."""
        self.ref_param = {n: p.detach().clone() for n, p in self.network.named_parameters()}
        new_fisher = self.get_fisher(batches, batch_size)
        alpha = 1 - self.inc_cls_num / self.network.head_w.shape[0]
        for n, p in self.fisher.items():
            new_fisher[n][: len(p)] = alpha * p + (1 - alpha) * new_fisher[n][: len(p)]
        self.fisher = new_fisher


# --------------------------------------------------------------------------------------------- LwF
class LWF:
    """reference core/model/lwf.py:9-81 (lamda hard-coded 3, T=2, teacher BN follows `train`)."""

    def __init__(self, net, init_cls_num, inc_cls_num):
        self.network = net            # net.head_* plays `self.classifier` (lwf.py:13)
        self.init_cls_num, self.inc_cls_num = init_cls_num, inc_cls_num
        self.known, self.total = 0, 0
        self.old = None               # frozen copy: old_backbone + old_fc
        self.task_idx = 0

    def before_task(self, task_idx, new_rows=None):
        self.task_idx = task_idx
        self.known = self.total
        self.total = self.init_cls_num + task_idx * self.inc_cls_num       # lwf.py:44-45
        net = self.network
        old_w, old_b = net.head_w.detach().clone(), net.head_b.detach().clone()
        w, b = new_rows if new_rows is not None else linear_default_init(self.total, net.head_w.shape[1])
        w = w.detach().clone(); b = b.detach().clone()
        n_old = old_w.shape[0]
        w[:n_old] = old_w; b[:n_old] = old_b                                 # lwf.py:35-37
        if task_idx != 0:
            # old_fc = copy of the classifier BEFORE growth; old_backbone = copy of the backbone (lwf.py:33,49)
            self.old = Net(net.arch, _clone_dict(net.P), _clone_dict(net.Bf), old_w, old_b)
        net.head_w = w.requires_grad_(True); net.head_b = b.requires_grad_(True)

    def observe(self, x, y, train=True):
        logit = self.network.logits(x, train)
        if self.task_idx == 0:
            loss = F.cross_entropy(logit, y)
        else:
            k = self.known
            loss_clf = F.cross_entropy(logit[:, k:], y - k)                 # lwf.py:61-62
            with torch.no_grad():
                soft = self.old.logits(x, train)                             # teacher in `train` mode: quirk a10
            loss = 3 * kd_loss(logit[:, :k], soft, 2.0) + loss_clf           # lwf.py:63-65
        pred, acc = _acc(logit, y)
        return pred, acc, loss

    def inference(self, x, y):
        return _acc(self.network.logits(x, False), y)


# --------------------------------------------------------------------------------------------- BiC
def classwise_split(images, labels, test_size):
    """bic.py:26-57: per class in ascending label order, positions shuffled with the GLOBAL numpy RNG, the first
    int(n * (1 - test_size)) (at least one of several) to the train side"""
    images, labels = np.array(images), np.array(labels)
    tr_i, tr_l, va_i, va_l = [], [], [], []
    for c in np.unique(labels):
        idx = np.where(labels == c)[0]
        np.random.shuffle(idx)
        k = int(len(idx) * (1 - test_size))
        if k == 0 and len(idx) > 1:
            k = 1
        tr_i += list(images[idx[:k]]); tr_l += list(labels[idx[:k]])
        va_i += list(images[idx[k:]]); va_l += list(labels[idx[k:]])
    return tr_i, va_i, tr_l, va_l


class BiC:
    """reference core/model/bic.py:83-340.  `net` = backbone + full-width head; one (alpha, beta) pair per task applied to that
    task's logit slice in EVERY pass (bic.py:129 forces the all-layers branch); stage 1: CE over the seen classes, from task 1 on
    alpha*T^2*KD(T=2) against the bias-corrected previous model + (1-alpha)*CE (:193-217); stage 2: the current task's pair under
    Adam(1e-3) with the net in eval mode (:219-232, trainer.py:545-547)."""

    def __init__(self, net, init_cls_num, inc_cls_num, task_num):
        self.network, self.init, self.inc, self.task_num = net, init_cls_num, inc_cls_num, task_num
        self.alphas = [torch.ones(1, dtype=net.head_w.dtype) for _ in range(task_num)]
        self.betas = [torch.zeros(1, dtype=net.head_w.dtype) for _ in range(task_num)]
        self.bias_opt = Adam([q for ab in zip(self.alphas, self.betas) for q in ab], 1e-3)
        self.seen, self.cur_task, self.old = 0, 0, None
        self.cls_count = {}

    def before_task(self, task_idx):
        self.old = self.network.clone(grad=False)                      # bic.py:111-114 (deepcopy, no eval())
        for q in self.alphas + self.betas:
            q.requires_grad_(False)                                    # :119-120
        self.cur_task = task_idx
        self.seen += self.init if task_idx == 0 else self.inc          # :123

    def after_task(self, task_idx):
        for i in range(self.task_num):                                 # :170-175
            self.alphas[i].requires_grad_(i == task_idx); self.betas[i].requires_grad_(i == task_idx)

    def bias_forward(self, logits):
        outs = []
        for i in range(self.task_num):                                 # :145-151
            lo, hi = (0, self.init) if i == 0 else (self.init + (i - 1) * self.inc, self.init + i * self.inc)
            outs.append(self.alphas[i] * logits[:, lo:hi] + self.betas[i])
        return torch.cat(outs, dim=1)

    def observe(self, x, y, train=True):
        p = self.bias_forward(self.network.logits(x, train))
        ce = F.cross_entropy(p[:, :self.seen], y)
        pred, acc = _acc(p[:, :self.seen], y)
        if self.cur_task == 0:
            return pred, acc, ce                                        # stage1, :180-191
        T, old = 2.0, self.seen - self.inc
        alpha = 1.0 * old / self.seen
        assert 1.0 * self.cur_task / (self.cur_task + 1) == alpha      # :199
        with torch.no_grad():
            pre = self.bias_forward(self.old.logits(x, train))          # previous model follows model.train(): quirk a10
        soft = -torch.mean(torch.sum(torch.softmax(pre[:, :old] / T, dim=1) * torch.log_softmax(p[:, :old] / T, dim=1), dim=1))
        return pred, acc, alpha * soft * T * T + (1 - alpha) * ce       # :212-215

    def stage2(self, x, y):
        with torch.no_grad():
            logits = self.network.logits(x, False)                     # trainer.py:545: model.eval(); the net is frozen (:167-168)
        p = self.bias_forward(logits)
        loss = F.cross_entropy(p[:, :self.seen], y)
        pred, acc = _acc(p[:, :self.seen], y)
        self.bias_opt.zero_grad(); loss.backward(); self.bias_opt.step()
        return pred, acc, loss

    def inference(self, x, y):
        return _acc(self.bias_forward(self.network.logits(x, False))[:, :self.seen], y)

    def split_and_update(self, images, labels, buffer, task_idx, buffer_size):
        """bic.py:245-340 on plain lists -> (train images, train labels, val images, val labels) of the two loaders' datasets
        (val lists None for task 0); `buffer` has train_images / train_labels / val_images / val_labels / total_classes"""
        from collections import Counter
        ratio = 0.1
        self.cls_count.update(Counter(labels))
        tr_i, va_i, tr_l, va_l = classwise_split(images, labels, ratio)
        train = (tr_i + buffer.train_images, tr_l + buffer.train_labels)
        val = (None, None)
        if task_idx > 0:
            vi, vl = list(buffer.val_images), list(buffer.val_labels)
            for c in np.unique(va_l):                                   # already grouped by class: appended in the same order
                pos = np.where(np.array(va_l) == c)[0]
                vi += list(np.array(va_i)[pos]); vl += list(np.array(va_l)[pos])
            val = (vi, vl)
        buffer.train_images += tr_i; buffer.train_labels += tr_l
        buffer.val_images += va_i; buffer.val_labels += va_l
        buffer.total_classes += self.init if task_idx == 0 else self.inc
        total = sum(self.cls_count.values())
        nt_i, nt_l, nv_i, nv_l = [], [], [], []
        for c in range(buffer.total_classes):
            n_val = int(self.cls_count[c] * buffer_size / total * ratio)
            n_tr = int(self.cls_count[c] * buffer_size / total * (1 - ratio))
            if n_val == 0 and n_tr > 1:
                n_val, n_tr = 1, n_tr - 1
            tp = np.where(np.array(buffer.train_labels) == c)[0][:n_tr]
            vp = np.where(np.array(buffer.val_labels) == c)[0][:n_val]
            nt_i += list(np.array(buffer.train_images)[tp]); nt_l += list(np.array(buffer.train_labels)[tp])
            nv_i += list(np.array(buffer.val_images)[vp]); nv_l += list(np.array(buffer.val_labels)[vp])
        buffer.train_images, buffer.train_labels, buffer.val_images, buffer.val_labels = nt_i, nt_l, nv_i, nv_l
        return train + val


# ---------------------------------------------------------------------------------------------- WA
def align_new_rows(w, n_new):
    """wa.py:96-109 / der.py:182-190: scale the last `n_new` rows by mean|old rows| / mean|new rows|; returns (w, gamma)."""
    norms = torch.norm(w, p=2, dim=1)
    gamma = norms[:-n_new].mean() / norms[-n_new:].mean()
    w = w.clone()
    w[-n_new:] *= gamma
    return w, gamma


class WA:
    """reference core/model/wa.py:141-243.  `total` grows by init_cls_num per task (wa.py:222); the logits head is not among
    the optimised parameters (Finetune.get_parameters, finetune.py:46-50, lists Finetune's own unused classifier instead) --
    `trainable()` returns what the reference optimizer really steps."""

    def __init__(self, net, init_cls_num):
        self.network = net                        # net.head_* plays network.classifier
        self.init_cls_num = init_cls_num
        self.known, self.total, self.task_idx = 0, 0, 0
        self.old = None

    def trainable(self):
        return list(self.network.P.values())

    def before_task(self, new_rows):
        self.total += self.init_cls_num
        net = self.network
        w, b = new_rows[0].detach().clone(), new_rows[1].detach().clone()
        if self.task_idx > 0:
            n_old = net.head_w.shape[0]
            w[:n_old] = net.head_w.detach(); b[:n_old] = net.head_b.detach()          # wa.py:84-91
        net.head_w, net.head_b = w.requires_grad_(True), b.requires_grad_(True)

    def observe(self, x, y, train=True):
        logits = self.network.logits(x, train)
        loss = F.cross_entropy(logits, y)
        if self.task_idx > 0:
            lam = self.known / self.total
            with torch.no_grad():
                soft = self.old.logits(x, train)                                      # teacher follows model.train()
            loss = (1 - lam) * loss + lam * kd_loss(logits[:, : self.known], soft, 2.0)   # wa.py:172-178
        pred, acc = _acc(logits, y)
        return pred, acc, loss

    def after_task(self):
        gamma = None
        if self.task_idx > 0:
            w, gamma = align_new_rows(self.network.head_w.detach(), self.total - self.known)
            self.network.head_w = w.requires_grad_(True)
        self.old = self.network.clone(grad=False)
        self.known = self.total
        self.task_idx += 1
        return gamma

    def inference(self, x, y):
        return _acc(self.network.logits(x, False), y)


# --------------------------------------------------------------------------------------------- DER
class DER:
    """reference core/model/der.py:66-226: one feature extractor per task (copied from the previous one, earlier ones frozen),
    `fc` over the concatenated features, `aux_fc` over the newest extractor's features."""

    def __init__(self, arch, init_cls_num, inc_cls_num):
        self.arch, self.init_cls_num, self.inc_cls_num = arch, init_cls_num, inc_cls_num
        self.P, self.Bf = [], []                  # per extractor
        self.fc_w = self.fc_b = self.aux_w = self.aux_b = None
        self.known = self.total = 0
        self.task_idx = 0

    def before_task(self, task_idx, fc_rows, aux_rows, first=None):
        """`fc_rows` / `aux_rows` = (w, b) supplying the fresh entries; `first` = (P, Bf) of extractor 0"""
        self.task_idx, self.known, self.total = task_idx, self.total, self.init_cls_num + task_idx * self.inc_cls_num
        self.P = [{k: v.detach() for k, v in P.items()} for P in self.P]              # freeze_convnets (der.py:176-179)
        if first is not None:
            P, Bf = first
        else:
            P, Bf = self.P[-1], self.Bf[-1]                                           # load_state_dict of the previous one
        self.P.append(_clone_dict(P, grad=True)); self.Bf.append(_clone_dict(Bf))
        w, b = fc_rows[0].detach().clone(), fc_rows[1].detach().clone()
        if self.fc_w is not None:
            r, c = self.fc_w.shape
            w[:r, :c] = self.fc_w.detach(); b[:r] = self.fc_b.detach()                # der.py:160-165
        self.fc_w, self.fc_b = w.requires_grad_(True), b.requires_grad_(True)
        self.aux_w, self.aux_b = aux_rows[0].detach().clone().requires_grad_(True), aux_rows[1].detach().clone().requires_grad_(True)

    def trainable(self):
        return list(self.P[-1].values()) + [self.fc_w, self.fc_b, self.aux_w, self.aux_b]

    def features(self, x, train):
        return torch.cat([nets.forward(self.arch, P, Bf, x, train) for P, Bf in zip(self.P, self.Bf)], 1)

    def observe(self, x, y, train=True):
        f = self.features(x, train)                # every extractor follows model.train(), frozen ones included
        logit = F.linear(f, self.fc_w, self.fc_b)
        loss = F.cross_entropy(logit, y)
        if self.task_idx > 0:
            aux_y = torch.clamp(y - self.known + 1, min=0)                            # der.py:117-123
            d = f.shape[1] // len(self.P)
            loss = F.cross_entropy(F.linear(f[:, -d:], self.aux_w, self.aux_b), aux_y) + loss
        pred, acc = _acc(logit, y)
        return pred, acc, loss

    def inference(self, x, y):
        return _acc(F.linear(self.features(x, False), self.fc_w, self.fc_b), y)


# ------------------------------------------------------------------------------------------- iCaRL
class ICarl:
    """reference core/model/icarl.py:42-287."""

    def __init__(self, net, init_cls_num, inc_cls_num):
        self.network = net            # head allocated with num_class rows up-front (icarl.py:56)
        self.init_cls_num, self.inc_cls_num = init_cls_num, inc_cls_num
        self.old_network = None
        self.prev_cls_num = 0
        self.accu_cls_num = 0
        self.cur_task_id = 0
        self.class_means = None

    def before_task(self, task_idx):
        if self.cur_task_id == 0:
            self.accu_cls_num = self.init_cls_num
        else:
            self.accu_cls_num += self.inc_cls_num                            # icarl.py:158-163

    def observe(self, x, y, train=True):
        cur = self.network.logits(x, train)[:, : self.accu_cls_num]         # icarl.py:208
        loss = F.cross_entropy(cur, y)
        if self.cur_task_id > 0:
            with torch.no_grad():                                            # grads into the teacher are discarded
                old = self.old_network.logits(x, train)
            loss = loss + kd_loss(cur[:, : self.prev_cls_num], old[:, : self.prev_cls_num], 2.0)
        pred, acc = _acc(cur, y)
        return pred, acc, loss

    def after_task_model(self):
        """icarl.py:169-176: snapshot teacher; buffer/herding handled by the caller."""
        self.old_network = self.network.clone(grad=False)
        self.prev_cls_num = self.accu_cls_num
        self.cur_task_id += 1

    def inference(self, x, y):
        if self.class_means is not None and len(self.class_means) == self.accu_cls_num:
            feats = self.network.features(x, False)
            return _acc(-ncm_distance(feats, self.class_means), y)          # argmin distance
        logits = self.network.logits(x, False)[:, : self.accu_cls_num]
        return _acc(logits, y)


def ncm_distance(feats, means):
    """squared L2 between every feature and every class mean (icarl.py:124-138)."""
    return (feats.unsqueeze(1) - means.unsqueeze(0)).pow(2).sum(2)


def class_means_from_buffer(feats, labels):
    """icarl.py:262-285: L2-normalise features, per-class mean, re-normalise."""
    f = feats / feats.norm(dim=1).view(-1, 1)
    out = []
    for c in np.unique(labels.numpy()):
        m = f[labels == int(c)].mean(0)
        out.append(m / m.norm())
    return torch.stack(out)


def herding_select(feats, labels, samples_per_class):
    """Greedy mean-matching on L2-normalised features; returns global indices
    (buffer/linearherdingbuffer.py:124-163; selected row is 'removed' by +1e6)."""
    f = feats / feats.norm(dim=1).view(-1, 1)
    f = f.clone()
    lab = labels.numpy()
    result = []
    for c in np.unique(lab):
        ind = np.where(lab == c)[0]
        cf = f[ind]
        mean = cf.mean(0, keepdim=True)
        run = torch.zeros_like(mean)
        i = 0
        while i < samples_per_class and i < cf.shape[0]:
            cost = (mean - (cf + run) / (i + 1)).norm(2, 1)
            j = cost.argmin().item()
            result.append(int(j + ind[0]))
            run += cf[j:j + 1]
            cf[j] = cf[j] + 1e6
            i += 1
    return result


# ------------------------------------------------------------------------------------------- LUCIR
def cosine_linear(x, w, sigma=None):
    """sigma * normalize(x) @ normalize(w)^T   (backbone/resnet.py:436-441)."""
    out = F.linear(F.normalize(x, p=2, dim=1), F.normalize(w, p=2, dim=1))
    return out if sigma is None else sigma * out


class LUCIR:
    """reference core/model/lucir.py:73-239, heads backbone/resnet.py:418-463."""

    def __init__(self, arch, P, Bf, w0, sigma, init_cls_num, inc_cls_num, lamda, K, lw_mr, dist):
        self.arch, self.P, self.Bf = arch, P, Bf
        self.fc1_w = w0            # CosineLinear at task 0; SplitCosineLinear.fc1 later
        self.fc2_w = None
        self.sigma = sigma
        self.init_cls_num, self.inc_cls_num = init_cls_num, inc_cls_num
        self.lamda, self.K, self.lw_mr, self.dist = lamda, K, lw_mr, dist
        self.ref = None
        self.task_idx = 0
        self.cur_lamda = lamda

    def _weights(self):
        return self.fc1_w if self.fc2_w is None else torch.cat([self.fc1_w, self.fc2_w], 0)

    def scores_and_feats(self, x, train):
        feats = nets.forward(self.arch, self.P, self.Bf, x, train)
        s = cosine_linear(feats, self._weights())
        return feats, s

    def before_task(self, task_idx, class_feats=None):
        """lucir.py:82-132.  `class_feats` = list (one per new class) of [n_i, D] eval-mode feature
        tensors used to imprint fc2 (lucir.py:134-159)."""
        self.task_idx = task_idx
        if task_idx >= 1:
            self.ref = dict(P=_clone_dict(self.P), Bf=_clone_dict(self.Bf), w=self._weights().detach().clone(),
                            sigma=self.sigma.detach().clone())
            old_w = self._weights().detach().clone()
            n_old = old_w.shape[0]
            self.cur_lamda = self.lamda * math.sqrt(n_old * 1.0 / self.inc_cls_num)
            avg_norm = old_w.norm(dim=1, keepdim=True).mean(0).double()
            self.fc1_w = old_w.clone().requires_grad_(True)
            novel = torch.zeros(self.inc_cls_num, old_w.shape[1])
            for j, cf in enumerate(class_feats):
                nf = F.normalize(cf.double(), p=2, dim=1)
                emb = nf.mean(0)
                novel[j] = (F.normalize(emb, p=2, dim=0) * avg_norm).to(novel.dtype)
            self.fc2_w = novel.requires_grad_(True)
            self.num_old = n_old
        else:
            self.cur_lamda = self.lamda

    def observe(self, x, y, train=True):
        feats, s = self.scores_and_feats(x, train)
        logit = self.sigma * s
        if self.task_idx == 0:
            loss = F.cross_entropy(logit, y)
        else:
            with torch.no_grad():
                ref_feats = nets.forward(self.arch, self.ref["P"], self.ref["Bf"], x, train)
            # CosineEmbeddingLoss(target=1) = mean(1 - cos)   lucir.py:182-183
            loss = (1 - F.cosine_similarity(feats, ref_feats, dim=1)).mean() * self.cur_lamda
            loss = loss + F.cross_entropy(logit, y)
            gt = s.gather(1, y.view(-1, 1)).squeeze(1)
            novel = s[:, self.num_old:].topk(self.K, dim=1)[0]
            hard = y < self.num_old
            if int(hard.sum()) > 0:
                g = gt[hard].view(-1, 1).repeat(1, self.K)
                n = novel[hard]
                # MarginRankingLoss(margin=dist)(g, n, 1) = mean(max(0, -(g-n)+dist))   lucir.py:204-205
                loss = loss + torch.clamp(-(g - n) + self.dist, min=0).mean() * self.lw_mr
        pred, acc = _acc(logit, y)
        return pred, acc, loss

    def inference(self, x, y):
        _, s = self.scores_and_feats(x, False)
        return _acc(self.sigma * s, y)

    def parameters(self):
        ps = list(self.P.values()) + [self.fc1_w]
        if self.fc2_w is not None:
            ps.append(self.fc2_w)
        ps.append(self.sigma)
        return ps


# ------------------------------------------------------------------------------------- optimisers
class SGD:
    """torch.optim.SGD semantics (trainer.py:159-166): d = g + wd*p; buf = m*buf + d (first step buf=d); p -= lr*buf.
    Params whose grad is None are skipped."""

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0):
        self.params = list(params)
        self.lr, self.momentum, self.wd = lr, momentum, weight_decay
        self.buf = [None] * len(self.params)

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        for i, p in enumerate(self.params):
            if p.grad is None:
                continue
            d = p.grad + self.wd * p if self.wd != 0 else p.grad.clone()
            if self.momentum != 0:
                if self.buf[i] is None:
                    self.buf[i] = d.clone()
                else:
                    self.buf[i].mul_(self.momentum).add_(d)
                d = self.buf[i]
            p.add_(d, alpha=-self.lr)


class Adam:
    """torch.optim.Adam (no amsgrad): m,v EMA, bias correction, p -= lr*mhat/(sqrt(vhat)+eps)."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = list(params)
        self.lr, self.b1, self.b2, self.eps, self.wd = lr, betas[0], betas[1], eps, weight_decay
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = [0] * len(self.params)      # per parameter, advanced only when it has a gradient (torch.optim.Adam's state["step"])

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        for i, p in enumerate(self.params):
            if p.grad is None:
                continue
            self.t[i] += 1
            g = p.grad + self.wd * p if self.wd != 0 else p.grad
            self.m[i].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[i].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            bc1 = 1 - self.b1 ** self.t[i]
            bc2 = 1 - self.b2 ** self.t[i]
            denom = (self.v[i].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(self.m[i], denom, value=-self.lr / bc1)
