"""Accuracy side of BASELINE.json's metric on the REAL dataset, for a box that has it (this pool has no network and no CIFAR-100):

    python tools/accuracy_cifar100.py --data /path/to/cifar100 [--config lwf-resnet18-cifar100-b50-5-10] [--dtype bf16|f32] [--seeds 1993 1994 ...]
                                      [--reference-acc 61.2 60.8 ...] [--epochs N] [--tasks T]

`--data` is a class-folder tree (train/<class>/*.png, test/<class>/*.png: the layout the reference's `core/data/dataset.py:195-266` reads).
For every seed the product `Trainer` runs the YAML unchanged (optionally shortened with --epochs / --tasks) and its result dict
(`core/trainer.py:457-520`: last / overall average accuracy, forgetting, BWT, the accuracy table) goes to stdout as one JSON line per run.
`--reference-acc`: the "[Batch] Last Average Acc" figures of runs of the reference itself (`python run_trainer.py --config ...` in a LibContinual
checkout, same YAML / seeds / data); the two samples are then compared the way tests/test_trainer_trace_gpu.py compares them -- difference of
the mean final accuracies against 0.3 points + 3 standard errors (Welch).  Nothing here runs in CI: it needs the dataset."""
import argparse
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402


def load_cfg(name, a, seed):
    from libcontinual_amd.config import Config
    path = name if os.path.exists(name) else (glob.glob(os.path.join(ROOT, "config", "**", name + ".yaml"), recursive=True) or [None])[0]
    if path is None:
        raise FileNotFoundError(name)
    cfg = Config(path).get_config_dict()
    cfg["data_root"] = a.data
    cfg["seed"] = seed
    if a.dtype:
        cfg["backbone"]["kwargs"]["dtype"] = a.dtype
    if a.epochs:
        cfg["epoch"] = a.epochs
        if "init_epoch" in cfg:
            cfg["init_epoch"] = a.epochs
    if a.tasks:
        cfg["task_num"] = a.tasks
    return cfg


def summarise(res):
    return {k: (np.asarray(v).round(3).tolist() if k == "acc_table" else float(v)) for k, v in res.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", required=True)
    ap.add_argument("--config", default="lwf-resnet18-cifar100-b50-5-10")
    ap.add_argument("--dtype", default=None, choices=[None, "bf16", "f32"])
    ap.add_argument("--seeds", type=int, nargs="+", default=[1993])
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--tasks", type=int, default=None)
    ap.add_argument("--reference-acc", type=float, nargs="+", default=None, help="final accuracies of the reference's own runs")
    a = ap.parse_args()
    if not os.path.isdir(os.path.join(a.data, "train")):
        raise SystemExit(f"{a.data}/train not found: this tool needs the class-folder dataset (see the docstring)")
    from libcontinual_amd.trainer import Trainer
    prod = []
    for seed in a.seeds:
        res = Trainer(0, load_cfg(a.config, a, seed)).train_loop()
        prod.append(res)
        print(json.dumps(dict(side="product", seed=seed, **summarise(res))), flush=True)
    if a.reference_acc and len(prod) > 1 and len(a.reference_acc) > 1:
        p = np.array([r["batch_last_acc"] for r in prod]); q = np.array(a.reference_acc)
        gap = float(p.mean() - q.mean())
        band = 0.3 + 3.0 * float(np.sqrt(p.var(ddof=1) / len(p) + q.var(ddof=1) / len(q)))
        print(json.dumps(dict(mean_gap_points=gap, band=band, within=abs(gap) <= band)))


if __name__ == "__main__":
    main()
