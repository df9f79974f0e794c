"""Whole-trainer scenarios (TEST INFRASTRUCTURE, see oracle/__init__): a complete class-incremental run -- every task, epoch,
step, evaluation -- driven by the REFERENCE'S OWN `Trainer.train_loop` (core/trainer.py:259-532, 563-720) around the reference's
own method classes, on a small synthetic class-structured dataset held in memory.

    python -m oracle.gen_golden trainer_ewc trainer_lwf trainer_icarl      (build container only)

writes `tests/golden/trainer_<name>.npz`: the hook sequence, every per-step loss, the accuracy table and the averaged
accuracies the reference reports.  The product's `Trainer` then runs the same scenario on the MI355X
(`tests/test_trainer_trace_gpu.py`): identical hook sequence, per-step losses of the first steps at fp32 tolerance, final
accuracies within the band BASELINE.json asks for (0.3 points) or the measured gap reported.

What is shared by the two runs: the data (built from tags by `detrand`), the loaders (torch DataLoader over `MemDataset`, so both
sides draw the same shuffles from `init_seed(seed + epoch)`), the initial backbone state (`fixtures.det_backbone_state`) and the
rows a growing head gains in `before_task` (set right after the hook returns, from tags).  The reference's trainer module is
imported through `ref_shim` -- `core.data` (torchvision / continuum) is stubbed, the data loaders are handed in ready-made, the
model is built here instead of by `_init_model`, and `compute_fps` (a CUDA timing loop, utils.py:235-257) is skipped.
"""
import contextlib
import copy
import io
import os
import sys
import types

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import detrand
from . import fixtures as fx

COMMON = dict(init=20, inc=5, tasks=5, train_per_class=60, test_per_class=25, init_epoch=4, epoch=3, batch_size=32, seed=3,
              lr=0.02, momentum=0.9, wd=5e-4, milestones=[2, 3], gamma=0.2, testing_times=1)
SCENARIOS = {
    "ewc": dict(method="EWC", arch="cifar_resnet32", feat_dim=64, kwargs=dict(lamda=100.0), buffer=None),
    "lwf": dict(method="LWF", arch="resnet18", feat_dim=512, kwargs=dict(), buffer=None),
    "icarl": dict(method="ICarl", arch="cifar_resnet32", feat_dim=64, kwargs=dict(), buffer=("LinearHerdingBuffer", dict(buffer_size=200, batch_size=32)), png=True),
}


# ---- round 3: LOW-VARIANCE accuracy scenarios.  The three scenarios above are short, under-trained runs: the reference
# against ITSELF from 1e-6-perturbed weights spreads 1.8-4.1 points on the final accuracy, so they cannot resolve BASELINE.json's
# +-0.3-point band.  These train every task to convergence (learning rate decayed to 1e-3 of its start), on well-separated classes, and
# are scored on >= 100 test images per class over ALL seen classes (`avg_acc`, core/trainer.py:715-720) -- the reference's own spread
# over 10 runs is recorded in the fixture and has to be <= 0.3 points (std) for the scenario to be used as a gate.  `acc_icarl11` has the
# B50-5x10 shape: half of the classes in task 0, ten increments, rehearsal buffer + herding + NCM, exemplars read back from PNG files.
ACC_COMMON = dict(COMMON, train_per_class=60, test_per_class=100, lr=0.05)
ACC_SCENARIOS = {
    # iCaRL / CifarResNet-32, the B50-5x10 SHAPE: 20 + 10 x 2 classes, 11 tasks.  Scenario design (reference runs only, 4 each, from
    # 1e-6-perturbed starts; final / overall average accuracy, mean +- std): signal 1.0 / noise 0.8: 99.87 +- 0.15 / 99.97 +- 0.03;
    # signal 0.6 / noise 1.0: 99.28 +- 0.57 / 99.78 +- 0.11 -- the harder the data, the wider the reference's own spread.
    "acc_icarl11": dict(method="ICarl", arch="cifar_resnet32", feat_dim=64, kwargs=dict(), png=True,
                        buffer=("LinearHerdingBuffer", dict(buffer_size=200, batch_size=32)),
                        common=dict(ACC_COMMON, init=20, inc=2, tasks=11, train_per_class=50, init_epoch=12, epoch=8, milestones=[5, 7], gamma=0.1,
                                    signal=0.8, noise=0.9)),
    # the same at the HARDER data setting of the design notes above (signal 0.6 / noise 1.0: the reference's own runs are NOT saturated, 99.3 +- 0.6):
    # an unsaturated rehearsal scenario for the +-0.3-point gate
    "acc_icarl11_hard": dict(method="ICarl", arch="cifar_resnet32", feat_dim=64, kwargs=dict(), png=True,
                             buffer=("LinearHerdingBuffer", dict(buffer_size=200, batch_size=32)),
                             common=dict(ACC_COMMON, init=20, inc=2, tasks=11, train_per_class=50, init_epoch=12, epoch=8, milestones=[5, 7], gamma=0.1,
                                         signal=0.6, noise=1.0)),
    # LwF / ResNet-18 (BASELINE configs[1]).  WITHOUT rehearsal the reference's own class-incremental accuracy is chaotic: trained to
    # convergence on four tasks (10 + 3 x 5 classes) its final average accuracy over four 1e-6-perturbed runs was 25.6 / 36.7 / 39.9 / 43.7
    # (std 7.8 points; EWC the same way: 28.2 / 40.8 / 27.4 / 4.9, std 14.9) -- the new logits are trained on their own slice and their
    # calibration against the old ones is arbitrary.  What IS reproducible is the accuracy after task 0 (plain training of the benchmarked
    # step: no teacher, no forgetting) -- the gated quantity of this scenario; the two-task final figure is recorded and compared loosely.
    "acc_lwf": dict(method="LWF", arch="resnet18", feat_dim=512, kwargs=dict(), buffer=None,
                    common=dict(ACC_COMMON, init=20, inc=5, tasks=2, init_epoch=12, epoch=6, milestones=[6, 9, 11], gamma=0.1, signal=0.45, noise=1.0)),
    # round 5: a scenario whose reference accuracy ends in the 70-90 % band.  Weaker class signal alone does not give one the gate can use
    # (explored: signal 0.35 / noise 1.0 -> 96.1 +- 1.8, signal 0.28 -> 85.9 +- 1.9 over four reference runs: once the network stops separating the classes
    # cleanly, forgetting under the 200-exemplar buffer makes the reference's own runs scatter).  Real class OVERLAP does (`mix`, _mixed() below): every sample is a
    # blend of its class pattern and its partner class's, the blend weight uniform in [0, mix) -- the samples beyond 0.5 belong to the partner for any
    # classifier, the ones near 0.5 are decided by the details of the trained network.
    # (explored: mix 0.62 -> 79.2 +- 0.35 final / 80.0 +- 0.41 overall; mix 0.58 -> 84.1 +- 0.28 / 85.3 +- 0.18 over four reference runs each; the second is the scenario -- its fixture's 24 runs: 85.4 +- 0.56 / 85.3 +- 0.28)
    "acc_icarl11_overlap": dict(method="ICarl", arch="cifar_resnet32", feat_dim=64, kwargs=dict(), png=True,
                                buffer=("LinearHerdingBuffer", dict(buffer_size=200, batch_size=32)),
                                common=dict(ACC_COMMON, init=20, inc=2, tasks=11, train_per_class=50, init_epoch=12, epoch=8, milestones=[5, 7], gamma=0.1,
                                            signal=0.8, noise=0.9, mix=0.58)),
}
SCENARIOS.update(ACC_SCENARIOS)


def common_of(name):
    return SCENARIOS[name].get("common", COMMON)


def num_classes(c=COMMON):
    return c["init"] + c["inc"] * (c["tasks"] - 1)


# ------------------------------------------------------------------------------------------------------------ data
class MemDataset(Dataset):
    """The attributes the reference's trainer, buffers and plugins touch on a per-task dataset (`images`, `labels`, `trfms`,
    list semantics of the rehearsal merge, core/trainer.py:305-312), over an in-memory image store; `images` are row indices."""

    def __init__(self, store, images, labels, mode="train"):
        self.store, self.mode, self.trfms, self.data_root = store, mode, None, None
        self.images, self.labels = list(images), list(labels)

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, i):
        return {"image": self.store[int(self.images[i])], "label": int(self.labels[i])}


class TaskLoaders:
    """`ContinualDatasets.get_loader` (core/data/dataset.py:91-96): the task's loader for training, all loaders up to the task
    for testing; one DataLoader object per task for the whole run (the rehearsal merge edits its dataset in place)."""

    def __init__(self, mode, datasets, batch_size):
        self.mode, self.task_num = mode, len(datasets)
        self.dataloaders = [DataLoader(ds, shuffle=True, batch_size=batch_size, drop_last=False, num_workers=0, pin_memory=False) for ds in datasets]
        self.cls_map = None

    def get_loader(self, task_idx):
        assert 0 <= task_idx < self.task_num
        return self.dataloaders[task_idx] if self.mode == "train" else self.dataloaders[:task_idx + 1]


def _pattern(tag, cls, chw):
    """the class pattern of make_store (CHW) / make_png_loaders (HWC)"""
    yy, xx = np.meshgrid(np.linspace(0, 1, 32), np.linspace(0, 1, 32), indexing="ij")
    coef = detrand.uniform(f"{tag}/coef{cls}", (3, 6), -1.0, 1.0)
    return np.stack([coef[ch, 0] * np.sin(2 * np.pi * (coef[ch, 1] * 2 * xx + coef[ch, 2] * 2 * yy)) +
                     coef[ch, 3] * np.cos(2 * np.pi * (coef[ch, 4] * 3 * xx - coef[ch, 5] * 3 * yy)) for ch in range(3)], 0 if chw else -1)


def _mixed(tag, c, cls, pat, n, pattern, chw):
    """per-sample class pattern [n, ...].  Without `mix` in the scenario: the class's own pattern for every sample (the data of every earlier fixture,
    unchanged).  With mix = a_max: sample k of class c shows (1 - a_k) * pattern(c) + a_k * pattern(c ^ 1), a_k ~ U[0, a_max) -- REAL class overlap:
    for a_max > 0.5 the samples with a_k > 0.5 look more like the partner class (which shares the task: init and inc are even), the Bayes accuracy is
    about 0.5 / a_max, and the samples near a_k = 0.5 are decided by the details of the trained network."""
    a_max = c.get("mix", 0.0)
    if a_max <= 0.0:
        return np.broadcast_to(pat[None], (n,) + pat.shape)
    other = pattern(tag, cls ^ 1, chw)
    a = detrand.uniform(f"{tag}/mix{cls}", (n,), 0.0, a_max).reshape((n,) + (1,) * pat.ndim)
    return (1.0 - a) * pat[None] + a * other[None]


def make_store(tag, c=COMMON):
    """class-structured 32 x 32 images, already normalised: a smooth class pattern (low-frequency, shared by the class) plus
    per-sample noise, mixed so that a small ResNet separates the classes well but not perfectly after a few epochs"""
    C = num_classes(c)
    n_tr, n_te = c["train_per_class"], c["test_per_class"]
    rows, labels_tr, labels_te, idx_tr, idx_te = [], [], [], [], []
    yy, xx = np.meshgrid(np.linspace(0, 1, 32), np.linspace(0, 1, 32), indexing="ij")
    for cls in range(C):
        coef = detrand.uniform(f"{tag}/coef{cls}", (3, 6), -1.0, 1.0)
        pat = np.stack([coef[ch, 0] * np.sin(2 * np.pi * (coef[ch, 1] * 2 * xx + coef[ch, 2] * 2 * yy)) +
                        coef[ch, 3] * np.cos(2 * np.pi * (coef[ch, 4] * 3 * xx - coef[ch, 5] * 3 * yy)) for ch in range(3)])
        noise = detrand.uniform(f"{tag}/noise{cls}", (n_tr + n_te, 3, 32, 32), -1.0, 1.0)
        x = c.get("signal", 0.55) * _mixed(tag, c, cls, pat, n_tr + n_te, _pattern, chw=True) + c.get("noise", 1.1) * noise
        base = sum(r.shape[0] for r in rows)
        rows.append(x.astype(np.float32))
        idx_tr += list(range(base, base + n_tr)); labels_tr += [cls] * n_tr
        idx_te += list(range(base + n_tr, base + n_tr + n_te)); labels_te += [cls] * n_te
    store = torch.from_numpy(np.concatenate(rows, 0))
    return store, (idx_tr, labels_tr), (idx_te, labels_te)


def make_loaders(tag, c=COMMON, dtype=torch.float32):
    store, (itr, ltr), (ite, lte) = make_store(tag, c)
    store = store.to(dtype)
    bounds = [(0, c["init"])] + [(c["init"] + k * c["inc"], c["init"] + (k + 1) * c["inc"]) for k in range(c["tasks"] - 1)]

    def split(idx, lab, mode):
        return [MemDataset(store, [i for i, l in zip(idx, lab) if a <= l < b], [l for l in lab if a <= l < b], mode) for a, b in bounds]
    return TaskLoaders("train", split(itr, ltr, "train"), c["batch_size"]), TaskLoaders("test", split(ite, lte, "test"), c["batch_size"])


def make_png_loaders(root, tag, c=COMMON):
    """the same kind of data as class-folder PNG files (root/{train,test}/<class>/<k>.png, the reference's on-disk layout): the
    reference's iCaRL reads its exemplars back from disk by path (icarl.py:222-250), so its scenario needs real files"""
    import PIL.Image                               # noqa: F401  (icarl.py does `import PIL` and then uses PIL.Image)
    from PIL import Image
    from .scenarios import PngDataset, png_transform
    C = num_classes(c)
    yy, xx = np.meshgrid(np.linspace(0, 1, 32), np.linspace(0, 1, 32), indexing="ij")
    lists = {"train": ([], []), "test": ([], [])}
    # the files depend on (tag, c) only: a root that already holds them (the accuracy tests make up to 24 runs of one scenario from differently perturbed
    # initial weights) is re-used -- a third of such a run's wall-clock was writing the same PNGs again
    stamp = os.path.join(root, ".complete")
    key = repr((tag, sorted((k, repr(v)) for k, v in c.items())))
    have = os.path.exists(stamp) and open(stamp).read() == key
    for cls in range(C):
        n = c["train_per_class"] + c["test_per_class"]
        if have:
            for k in range(n):
                mode = "train" if k < c["train_per_class"] else "test"
                lists[mode][0].append(os.path.join(f"{cls:03d}", f"{k}.png")); lists[mode][1].append(cls)
            continue
        coef = detrand.uniform(f"{tag}/coef{cls}", (3, 6), -1.0, 1.0)
        pat = np.stack([coef[ch, 0] * np.sin(2 * np.pi * (coef[ch, 1] * 2 * xx + coef[ch, 2] * 2 * yy)) +
                        coef[ch, 3] * np.cos(2 * np.pi * (coef[ch, 4] * 3 * xx - coef[ch, 5] * 3 * yy)) for ch in range(3)], -1)      # HWC
        n = c["train_per_class"] + c["test_per_class"]
        noise = detrand.uniform(f"{tag}/noise{cls}", (n, 32, 32, 3), -1.0, 1.0)
        pats = _mixed(tag, c, cls, pat, n, _pattern, chw=False)
        for k in range(n):
            mode = "train" if k < c["train_per_class"] else "test"
            d = os.path.join(root, mode, f"{cls:03d}")
            os.makedirs(d, exist_ok=True)
            ks, kn = (0.2 * c["signal"], 0.2 * c["noise"]) if "signal" in c else (0.11, 0.22)
            a = np.clip((0.5 + ks * pats[k] + kn * noise[k]) * 255.0, 0, 255).astype(np.uint8)
            rel = os.path.join(f"{cls:03d}", f"{k}.png")
            Image.fromarray(a).save(os.path.join(root, mode, rel))
            lists[mode][0].append(rel); lists[mode][1].append(cls)
    if not have:
        with open(stamp, "w") as f:
            f.write(key)
    bounds = [(0, c["init"])] + [(c["init"] + k * c["inc"], c["init"] + (k + 1) * c["inc"]) for k in range(c["tasks"] - 1)]

    def split(mode):
        imgs, labs = lists[mode]
        return [PngDataset(root, mode, [i for i, l in zip(imgs, labs) if a <= l < b], [l for l in labs if a <= l < b], png_transform) for a, b in bounds]
    return TaskLoaders("train", split("train"), c["batch_size"]), TaskLoaders("test", split("test"), c["batch_size"])


def loaders_for(name, root, c=COMMON):
    """scenario data: in memory, or PNG files under `root` for the methods that read images back by path"""
    if SCENARIOS[name].get("png"):
        return make_png_loaders(root, f"trainer/{name}/data", c)
    return make_loaders(f"trainer/{name}/data", c)


def trainer_config(name, c=COMMON, **over):
    s = SCENARIOS[name]
    cfg = dict(classifier={"name": s["method"], "kwargs": dict(num_class=num_classes(c), feat_dim=s["feat_dim"], init_cls_num=c["init"], inc_cls_num=c["inc"],
                                                               task_num=c["tasks"], **s["kwargs"])},
               backbone={"name": s["arch"], "kwargs": {"num_classes": num_classes(c), "args": {"dataset": "cifar100"}}},
               buffer={"name": s["buffer"][0], "kwargs": dict(s["buffer"][1])} if s["buffer"] else {"name": "LinearBuffer", "kwargs": dict(buffer_size=0, batch_size=32, strategy="herding")},
               optimizer={"name": "SGD", "kwargs": dict(lr=c["lr"], momentum=c["momentum"], weight_decay=c["wd"])},
               lr_scheduler={"name": "MultiStepLR", "kwargs": dict(milestones=list(c["milestones"]), gamma=c["gamma"])},
               init_cls_num=c["init"], inc_cls_num=c["inc"], task_num=c["tasks"], epoch=c["epoch"], init_epoch=c["init_epoch"], batch_size=c["batch_size"],
               val_per_epoch=1000, testing_times=c["testing_times"], testing_per_task=True, setting="task-agnostic", seed=c["seed"], deterministic=True,
               n_gpu=1, num_workers=0, image_size=32, dataset="memory", save_path="", device_ids="auto", pin_memory=False)
    cfg.update(over)
    return cfg


def head_rows(name, task, rows, feat_dim):
    """deterministic values for the head rows that exist after `before_task(task)` (a growing head re-initialises / adds rows at
    random there): [rows, feat_dim] weight, [rows] bias"""
    return fx.det_linear(f"trainer/{name}/head{task}", rows, feat_dim)


# ------------------------------------------------------------------------------------- the reference's own trainer
def _reference_trainer_module():
    from . import ref_shim
    ref_shim.install()
    root = ref_shim.REF_ROOT
    if "core.utils" not in sys.modules or not hasattr(sys.modules["core.utils"], "AverageMeter"):
        pkg = types.ModuleType("core.utils")
        pkg.__path__ = [os.path.join(root, "core", "utils")]
        sys.modules["core.utils"] = pkg
        u = ref_shim.load("core.utils.utils")
        lg = ref_shim.load("core.utils.logger")
        for m in (u, lg):
            for k, v in vars(m).items():
                if not k.startswith("_"):
                    setattr(pkg, k, v)
    if "core.data" not in sys.modules:
        d = types.ModuleType("core.data")           # the real package needs torchvision / continuum; loaders are handed in ready-made
        d.get_dataloader = None
        sys.modules["core.data"] = d
    bufpkg = sys.modules["core.model.buffer"]
    for sub in ("linearbuffer", "update", "linearherdingbuffer"):
        m = ref_shim.load(f"core.model.buffer.{sub}")
        for k, v in vars(m).items():
            if not k.startswith("_"):
                setattr(bufpkg, k, v)
    if not hasattr(sys.modules["core.model"], "bic"):
        sys.modules["core.model"].bic = ref_shim.load("core.model.bic")
    tr = ref_shim.load("core.trainer")
    tr.compute_fps = lambda model, config: {"avg_fps": 0.0, "best_fps": 0.0}      # CUDA-only timing loop at the end of train_loop
    return tr


def build_reference(name, ns, root, c=COMMON, perturb=0):
    """-> trainer object of the reference's class, assembled without its __init__ (which needs CUDA, torchvision, a YAML, ...)"""
    s = SCENARIOS[name]
    trmod = _reference_trainer_module()
    cfg = trainer_config(name, c)
    P, Bf = fx.det_backbone_state(s["arch"], f"trainer/{name}")
    if s["arch"] == "resnet18":
        bb = ns.resnet18(num_classes=num_classes(c), args={"dataset": "cifar100"})
    else:
        bb = getattr(ns, s["arch"])()
    if perturb:
        # the SAME algorithm from initial weights moved by one part in 10^6 (fp32 rounding level): how far two runs of the reference
        # itself end up apart -- the noise floor any other fp32 implementation's accuracy has to be read against
        P = {k: v * (1.0 + 1e-6 * fx._t(detrand.uniform(f"trainer/{name}/perturb{perturb}/{k}", tuple(v.shape), -1.0, 1.0))) for k, v in P.items()}
    bb.load_state_dict({**{k: v.clone().float() for k, v in P.items()}, **{k: v.clone().float() if v.is_floating_point() else v.clone() for k, v in Bf.items()}})
    model = getattr(ns, s["method"])(bb, s["feat_dim"], num_classes(c), device=torch.device("cpu"), **{k: v for k, v in cfg["classifier"]["kwargs"].items() if k not in ("num_class", "feat_dim")})
    t = trmod.Trainer.__new__(trmod.Trainer)
    t.rank, t.config, t.distribute, t.device = 0, cfg, False, torch.device("cpu")
    t.init_cls_num, t.inc_cls_num, t.task_num = c["init"], c["inc"], c["tasks"]
    t.model = model
    t.train_loader, t.test_loader = loaders_for(name, root, c)
    if s["buffer"]:
        t.buffer = getattr(ns, s["buffer"][0])(**s["buffer"][1])
    else:
        t.buffer = sys.modules["core.model.buffer"].LinearBuffer(0, "herding", 32)
    t.task_idx = 0
    t.init_epoch, t.inc_epoch = c["init_epoch"], c["epoch"]
    t.val_per_epoch = cfg["val_per_epoch"]
    t.train_meter, t.test_meter = t._init_meter()
    t.writer = None
    return t


class Recorder:
    """hook sequence + per-step losses, captured by wrapping the methods the trainer calls (same wrappers on both sides)"""

    def __init__(self, name, trainer, model, head_of, to_item):
        self.trace, self.losses, self.task, self.epoch, self.validations = [], [], -1, -1, []
        s = SCENARIOS[name]
        rec = self
        orig_before, orig_after, orig_observe = model.before_task, getattr(model, "after_task", None), model.observe
        orig_train, orig_validate = trainer._train, trainer._validate

        def before_task(task_idx, *a, **k):
            rec.task = task_idx
            rec.trace.append(("before_task", task_idx, -1))
            old = head_of(model).weight.shape[0] if task_idx > 0 else 0      # rows trained so far stay as they are
            out = orig_before(task_idx, *a, **k)
            head = head_of(model)
            w, b = head_rows(name, task_idx, head.weight.shape[0], s["feat_dim"])
            with torch.no_grad():
                head.weight[old:].copy_(w[old:].to(head.weight))
                if head.bias is not None:
                    head.bias[old:].copy_(b[old:].to(head.bias))
            return out

        def after_task(task_idx, *a, **k):
            rec.trace.append(("after_task", task_idx, -1))
            return orig_after(task_idx, *a, **k)

        def observe(batch):
            out = orig_observe(batch)
            rec.losses.append(to_item(out[2]))
            return out

        def _train(epoch_idx, dataloader):
            rec.epoch = epoch_idx
            rec.trace.append(("train_epoch", rec.task, epoch_idx))
            n0 = len(rec.losses)
            out = orig_train(epoch_idx, dataloader)
            rec.trace.append(("steps", rec.task, len(rec.losses) - n0))
            return out

        def _validate(task_idx):
            rec.trace.append(("validate", task_idx, -1))
            out = orig_validate(task_idx)
            rec.validations.append((task_idx, float(out["avg_acc"]), [float(v) for v in out["per_task_acc"]]))
            return out
        model.before_task, model.observe = before_task, observe
        if orig_after is not None:
            model.after_task = after_task
        trainer._train, trainer._validate = _train, _validate


def reference_head(name):
    if SCENARIOS[name]["method"] == "LWF":
        return lambda m: m.classifier
    return lambda m: m.network.classifier


def encode_trace(trace):
    names = ["before_task", "train_epoch", "steps", "validate", "after_task"]
    return np.asarray([[names.index(e), t, k] for e, t, k in trace], dtype=np.int64)


def run_reference(name, ns, c=COMMON, perturb=0):
    import tempfile
    torch.manual_seed(c["seed"])
    with fx.use_dtype(torch.float32), tempfile.TemporaryDirectory() as root:     # the reference's own arithmetic: fp32 throughout
        t = build_reference(name, ns, root, c, perturb)
        rec = Recorder(name, t, t.model, reference_head(name), lambda l: float(l.item()))
        # mirror of Trainer.__init__'s seeding (core/trainer.py:99-114 -> init_seed) so that both sides start from the same RNG state
        sys.modules["core.utils"].init_seed(c["seed"], True)
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            t.train_loop()
        return pack(rec, t.buffer, c)


def scenario_trainer(name, ns, c=COMMON, n_perturbed=6):
    """run the reference's Trainer.train_loop on the scenario (+ `n_perturbed` runs from 1e-6-perturbed initial weights);
    -> dict for tests/golden/trainer_<name>.npz"""
    out = run_reference(name, ns, c)
    pert = [run_reference(name, ns, c, perturb=k + 1) for k in range(n_perturbed)]
    if pert:
        out["perturbed_batch_last_acc"] = np.stack([p["batch_last_acc"] for p in pert])
        out["perturbed_overall_avg_acc"] = np.concatenate([p["overall_avg_acc"] for p in pert])
        out["perturbed_losses_first_epoch"] = np.stack([p["losses"][:int(out["trace"][2][2])] for p in pert])
    return out


def pack(rec, buffer, c=COMMON):
    """the recorded run as arrays.  With testing_times = 1 the LAST validation of a task is the one whose figures enter the accuracy
    table and the per-task 'Last Average Acc' (core/trainer.py:457-489); the overall figure is their mean over the tasks (:509)."""
    T = c["tasks"]
    acc_table, last = np.zeros((T, T)), np.zeros(T)
    for task, avg, per in rec.validations:
        last[task] = avg
        acc_table[task, :len(per)] = per
    out = dict(trace=encode_trace(rec.trace), losses=np.asarray(rec.losses, np.float64), batch_last_acc=last, acc_table=acc_table,
               overall_avg_acc=np.asarray([last.mean()]), n_validations=np.asarray([len(rec.validations)]))
    labels = getattr(buffer, "labels", [])
    if len(labels):
        out["buffer_labels"] = np.asarray([int(v) for v in labels])
        out["buffer_images"] = np.asarray([str(v) for v in buffer.images])
    return out
