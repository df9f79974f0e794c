"""Trainer host logic on CPU with a tiny pure-torch plugin (the product plugins need the GPU): hook order,
per-task optimizer reset, rehearsal merge, seed schedule, validation quirks and metric bookkeeping follow
core/trainer.py:259-532, 563-720."""
import types

import numpy as np
import torch
import torch.nn as nn

from libcontinual_amd import trainer as T
from libcontinual_amd.config import Config
from libcontinual_amd.data import get_dataloader
import libcontinual_amd.model as M


class TinyBackbone(nn.Module):
    def __init__(self, **kw):
        super().__init__()
        self.fc = nn.Linear(3 * 8 * 8, 16)

    def forward(self, x):
        return {"features": torch.relu(self.fc(x.flatten(1)))}


class TinyMethod(nn.Module):
    """reference-style plugin written against the documented contract (docs/tutorials/en/add_a_new_method_en.md)"""
    events = []

    def __init__(self, backbone, feat_dim, num_class, **kwargs):
        super().__init__()
        self.backbone, self.device = backbone, kwargs["device"]
        self.classifier = nn.Linear(feat_dim, num_class)
        self.seen = 0
        self.kw = kwargs

    def before_task(self, task_idx, buffer, train_loader, test_loaders):
        TinyMethod.events.append(("before", task_idx, len(train_loader.dataset), len(test_loaders)))
        self.seen = self.kw["init_cls_num"] + task_idx * self.kw["inc_cls_num"]

    def observe(self, data):
        x, y = data["image"], data["label"]
        logit = self.classifier(self.backbone(x)["features"])[:, : self.seen]
        loss = nn.functional.cross_entropy(logit, y)
        pred = logit.argmax(1)
        return pred, (pred == y).sum().item() / len(y), loss

    def inference(self, data):
        x, y = data["image"], data["label"]
        pred = self.classifier(self.backbone(x)["features"])[:, : self.seen].argmax(1)
        return pred, (pred == y).sum().item() / len(y)

    def after_task(self, task_idx, buffer, train_loader, test_loaders):
        TinyMethod.events.append(("after", task_idx, self.trainer_opt()))

    def get_parameters(self, config):
        return self.parameters()


def make_cfg(**over):
    cfg = Config().get_config_dict()
    cfg.update(dict(device="cpu", dataset="synthetic", image_size=8, init_cls_num=4, inc_cls_num=2, task_num=3, epoch=2, init_epoch=1,
                    batch_size=8, val_per_epoch=1, testing_times=2, num_workers=0, save_path="", synthetic_per_class=6, synthetic_test_per_class=3,
                    backbone={"name": "TinyBackbone", "kwargs": {}},
                    classifier={"name": "TinyMethod", "kwargs": {"feat_dim": 16, "num_class": 8, "init_cls_num": 4, "inc_cls_num": 2}},
                    optimizer={"name": "SGD", "kwargs": {"lr": 0.05, "momentum": 0.9}},
                    lr_scheduler={"name": "MultiStepLR", "kwargs": {"gamma": 0.1, "milestones": [1]}}))
    cfg.update(over)
    return cfg


def namespace(buffer_cls=None):
    ns = types.SimpleNamespace(TinyBackbone=TinyBackbone, TinyMethod=TinyMethod, LinearBuffer=M.LinearBuffer,
                               LinearHerdingBuffer=M.LinearHerdingBuffer)
    return ns


def test_hook_order_optimizer_reset_and_metrics():
    TinyMethod.events = []
    cfg = make_cfg()
    tr = T.Trainer(0, cfg, model_namespace=namespace(), optim_namespace=torch.optim, log=lambda *a, **k: None)
    opts = []
    TinyMethod.trainer_opt = lambda self: tr.optimizer
    res = tr.train_loop()
    ev = TinyMethod.events
    # before_task / after_task once per task, in order; the train loader of task t holds that task's samples, the test
    # loaders of tasks 0..t are handed over (trainer.py:288-289)
    assert [e[:2] for e in ev] == [("before", 0), ("after", 0), ("before", 1), ("after", 1), ("before", 2), ("after", 2)]
    assert [e[2:] for e in ev if e[0] == "before"] == [(24, 1), (12, 2), (12, 3)]
    # a fresh optimizer for every task (trainer.py:294)
    o = [e[2] for e in ev if e[0] == "after"]        # kept alive here, so identities are comparable
    assert o[0] is not o[1] and o[1] is not o[2] and o[0] is not o[2]
    # epochs: init_epoch for task 0, epoch afterwards; validation after every epoch (val_per_epoch=1) + testing_times
    tr_events = [e for e in tr.hook_trace if e[0] == "train_epoch"]
    assert [(e[1], e[2]) for e in tr_events] == [(0, 0), (1, 0), (1, 1), (2, 0), (2, 1)]
    assert sum(1 for e in tr.hook_trace if e[0] == "validate") == 5 + 3 * 2
    assert res["acc_table"].shape == (3, 3) and np.all(res["acc_table"][0, 1:] == 0)
    assert 0 <= res["batch_last_acc"] <= 100


def test_rehearsal_merge_and_random_buffer():
    """buffer_size > 0: the task dataset is concatenated with the buffer before training (trainer.py:305-322) and the
    buffer is refreshed by strategy after the task (:410-418)"""
    TinyMethod.events = []
    cfg = make_cfg(buffer={"name": "LinearBuffer", "kwargs": {"buffer_size": 10, "batch_size": 8, "strategy": "random"}}, testing_times=1)
    tr = T.Trainer(0, cfg, model_namespace=namespace(), optim_namespace=torch.optim, log=lambda *a, **k: None)
    TinyMethod.trainer_opt = lambda self: tr.optimizer
    tr.train_loop()
    assert len(tr.buffer.labels) == 10 and tr.buffer.total_classes == 8
    # task 1 trained on 12 new + 10 rehearsal samples
    assert len(tr.train_loader.get_loader(1).dataset) == 22


class TinyBic(TinyMethod):
    """pure-torch stand-in with BiC's trainer-facing surface (bic.py:83-340); the host-side split is the PRODUCT's method"""
    spilt_and_update = M.bic.spilt_and_update

    def __init__(self, backbone, feat_dim, num_class, **kwargs):
        super().__init__(backbone, feat_dim, num_class, **kwargs)
        self.bias_layers = nn.ModuleList([M.BiasLayer() for _ in range(kwargs["task_num"])])
        self.bias_opt = torch.optim.Adam(self.bias_layers.parameters(), lr=1e-3)
        self.cls_count, self.stage2_modes = {}, []

    def observe(self, data):
        TinyMethod.events.append(("observe_mode", self.training, self.bias_layers[0].training))
        return super().observe(data)

    def stage2(self, data):
        self.stage2_modes.append((self.training, self.backbone.training, self.bias_layers[0].training))
        x, y = data["image"], data["label"]
        with torch.no_grad():
            z = self.classifier(self.backbone(x)["features"])[:, : self.seen]
        loss = nn.functional.cross_entropy(self.bias_layers[0](z), y)
        self.bias_opt.zero_grad(); loss.backward(); self.bias_opt.step()
        pred = z.argmax(1)
        return pred, (pred == y).sum().item() / len(y), loss


def test_bic_flow_split_stage1_recipe_and_stage2():
    """the method named `bic` gets (core/trainer.py:297-303) a hard-wired SGD / MultiStepLR and its own train / validation loaders,
    keeps the trainer's buffer update away (:410) and, from task 1 on, a second stage on the validation loader between the buffer
    update and the final evaluations (:420-455) with the model in eval mode and the bias layers in train mode (:545-547)"""
    TinyMethod.events = []
    ns = namespace()
    ns.bic, ns.LinearSpiltBuffer = TinyBic, M.LinearSpiltBuffer
    cfg = make_cfg(classifier={"name": "bic", "kwargs": {"feat_dim": 16, "num_class": 8, "init_cls_num": 4, "inc_cls_num": 2, "task_num": 3}},
                   buffer={"name": "LinearSpiltBuffer", "kwargs": {"buffer_size": 40, "batch_size": 8, "strategy": "balance_random", "val_ratio": 0.1}},
                   stage2_epoch=2, testing_times=1, val_per_epoch=100, synthetic_per_class=10, gpu_input_pipeline=False)
    tr = T.Trainer(0, cfg, model_namespace=ns, optim_namespace=torch.optim, log=lambda *a, **k: None)
    seen_opts = []
    TinyMethod.trainer_opt = lambda self: seen_opts.append((tr.optimizer.param_groups[0]["lr"], tr.optimizer.param_groups[0]["momentum"],
                                                            tr.optimizer.param_groups[0]["weight_decay"], tr.scheduler.milestones)) or tr.optimizer
    tr.train_loop()
    want_wd = [2e-4 * 3 / (t + 1) for t in range(3)]
    assert [o[:2] for o in seen_opts] == [(0.1, 0.9)] * 3 and np.allclose([o[2] for o in seen_opts], want_wd)
    assert all(sorted(o[3]) == [100, 150, 200] for o in seen_opts)
    # order inside a task: stage-1 epochs, [stage-2 epochs], evaluation
    for t in range(3):
        ev = [e[0] for e in tr.hook_trace if e[1] == t and e[0] != "before_task"]
        want = ["train_epoch"] * (1 if t == 0 else 2) + ["after_task"] + (["stage2_epoch"] * 2 if t > 0 else [])
        assert [e for e in ev if e != "validate"] == want and ev[-1] == "validate", (t, ev)       # (the `epoch + 1 == inc_epoch` validation sits in between)
    assert all(m == (False, False, True) for m in tr.model.stage2_modes) and len(tr.model.stage2_modes) > 0
    assert all(e[1] is True for e in TinyMethod.events if e[0] == "observe_mode")                # stage 1 runs under model.train()
    # the split buffer was filled and re-cut by the plugin alone: 40 * count_c / total = 5 per class -> 3 train + 1 validation (bic.py:317-321)
    buf = tr.buffer
    assert buf.total_classes == 8 and len(buf.train_labels) == 24 and len(buf.val_labels) == 8
    assert sorted(int(v) for v in set(buf.train_labels)) == list(range(8)) and sorted(int(v) for v in buf.val_labels) == list(range(8))
    assert not hasattr(buf, "images")


def test_seed_schedule_reproducible():
    """init_seed(seed + epoch) before every epoch (trainer.py:584): two runs give identical parameters"""
    outs = []
    for _ in range(2):
        TinyMethod.events = []
        tr = T.Trainer(0, make_cfg(testing_times=1), model_namespace=namespace(), optim_namespace=torch.optim, log=lambda *a, **k: None)
        TinyMethod.trainer_opt = lambda self: tr.optimizer
        tr.train_loop()
        outs.append(torch.cat([p.detach().flatten() for p in tr.model.parameters()]))
    assert torch.equal(outs[0], outs[1])


def test_class_folder_dataset_and_class_order(tmp_path):
    """class-folder tree (docs/tutorials/en/data_module_en.md:15-39), per-task splits (dataset.py:81-92) and the
    seeded class-order permutation (dataloader.py:113-122)"""
    from PIL import Image
    root = tmp_path / "d"
    for mode, n in (("train", 3), ("test", 2)):
        for c in range(6):
            d = root / mode / f"c{c}"
            d.mkdir(parents=True)
            for k in range(n):
                Image.fromarray(np.full((32, 32, 3), c * 30 + k, np.uint8)).save(d / f"{k}.png")
    cfg = make_cfg(dataset="cifar100", data_root=str(root), image_size=32, init_cls_num=2, inc_cls_num=2, task_num=3)
    np.random.seed(1993)
    perm = np.random.permutation(6)
    np.random.seed(1993)
    tr = get_dataloader(cfg, "train")
    te = get_dataloader(cfg, "test", cls_map=tr.cls_map)
    assert [tr.cls_map[i] for i in range(6)] == [f"c{p}" for p in perm]
    ds1 = tr.get_loader(1).dataset
    assert sorted(set(ds1.labels)) == [2, 3] and len(ds1) == 6 and len(te.get_loader(2)) == 3
    b = next(iter(tr.get_loader(0)))
    assert b["image"].shape[1:] == (3, 32, 32) and b["image"].dtype == torch.float32 and b["label"].dtype == torch.int64
