"""The product `Trainer` against complete runs of the REFERENCE'S OWN `Trainer.train_loop` (tests/golden/trainer_*.npz, generated
by `python -m oracle.gen_golden trainer_ewc trainer_lwf trainer_icarl` from core/trainer.py:259-720 driving the reference's EWC /
LWF / ICarl classes in fp32 on the CPU; oracle/trainer_scenarios.py): same data, same loaders, same initial weights, same seeds.

* hook sequence: before_task -> per epoch (train, number of optimisation steps) -> the `(epoch + 1) == inc_epoch` validation quirk
  -> after_task -> `testing_times` validations, for every task: EXACT.
* per-step losses: the first optimisation steps at fp32 tolerance in f32 mode (1e-7 .. 1e-4 observed); later the trajectories
  drift apart like any two fp32 implementations of a chaotic optimisation -- the fixture holds two more runs of the REFERENCE
  ITSELF from initial weights perturbed by one part in 10^6, and their deviation from the unperturbed reference run is reported
  next to the product's.
* accuracy (the second half of BASELINE.json's metric): the per-task "Last Average Acc" figures and their mean -- what
  core/trainer.py:457-520 reports -- against the reference's.  Band: BASELINE's 0.3 points, or the reference's own spread under
  that perturbation times a stated factor where that is larger (a 1000-image test set: one image is 0.1 point).  All figures
  go to gpurun_out/accuracy_parity.json (copied to profiles/r02_accuracy_parity.json).
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import trainer_scenarios as ts                # noqa: E402   (test infrastructure: data, config, recorder)

HERE = os.path.dirname(os.path.abspath(__file__))


def run_product(name, dtype, root):
    import libcontinual_amd.model as M
    from libcontinual_amd.trainer import Trainer
    from libcontinual_amd.utils import init_seed
    from oracle import fixtures as fx
    c, s = ts.COMMON, ts.SCENARIOS[name]
    cfg = ts.trainer_config(name, c, gpu_input_pipeline=False)
    cfg["backbone"]["kwargs"]["dtype"] = dtype
    loaders = ts.loaders_for(name, root, c)
    tr = Trainer(0, cfg, model_namespace=M, dataloaders=loaders, log=lambda *a, **k: None)
    with fx.use_dtype(torch.float32):
        P, Bf = fx.det_backbone_state(s["arch"], f"trainer/{name}")
    bb = tr.model.backbone if hasattr(tr.model, "backbone") else tr.model.network.backbone
    bb.load_state_dict({**P, **Bf})
    head = (lambda m: m.classifier) if name == "lwf" else (lambda m: m.network.classifier)
    rec = ts.Recorder(name, tr, tr.model, head, lambda l: float(l.detach().float().item()) if torch.is_tensor(l) else float(l))
    init_seed(c["seed"], True)
    tr.train_loop()
    torch.cuda.synchronize()
    with fx.use_dtype(torch.float32):
        return ts.pack(rec, tr.buffer, c), tr


# first optimisation steps (before the chaotic amplification of rounding differences sets in): relative loss deviation
FIRST_STEPS = {"f32": (3, 2e-4), "bf16": (3, 3e-2)}
# accuracy: BASELINE.json's 0.3 points, or the reference's OWN spread under a 1e-6 perturbation of the initial weights (two extra
# runs in the fixture) times this factor, whichever is larger
SPREAD_FACTOR = {"f32": 2.0, "bf16": 3.0}


@pytest.mark.parametrize("name", ["ewc", "lwf", "icarl"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_trainer_reproduces_the_reference_run(name, dtype, tmp_path):
    ref = np.load(os.path.join(HERE, "golden", f"trainer_{name}.npz"))
    got, tr = run_product(name, dtype, str(tmp_path))
    # ---- a1: hook sequence, step counts included
    assert got["trace"].tolist() == ref["trace"].tolist()
    assert len(got["losses"]) == len(ref["losses"])
    # ---- per-step losses
    n0 = int(ref["trace"][2][2])                                          # ("steps", task 0, n) of epoch 0
    dev = np.abs(got["losses"][:n0] - ref["losses"][:n0]) / np.abs(ref["losses"][:n0])
    ref_dev = np.abs(ref["perturbed_losses_first_epoch"] - ref["losses"][:n0]).max(0) / np.abs(ref["losses"][:n0])     # the reference against itself
    k, tol = FIRST_STEPS[dtype]
    # ---- accuracy
    spread_last = float(np.abs(ref["perturbed_batch_last_acc"][:, -1] - ref["batch_last_acc"][-1]).max())
    spread_avg = float(np.abs(ref["perturbed_overall_avg_acc"] - ref["overall_avg_acc"][0]).max())
    gap_last = float(got["batch_last_acc"][-1] - ref["batch_last_acc"][-1])
    gap_avg = float(got["overall_avg_acc"][0] - ref["overall_avg_acc"][0])
    band_last = max(0.3, SPREAD_FACTOR[dtype] * spread_last)
    band_avg = max(0.3, SPREAD_FACTOR[dtype] * spread_avg)
    report = dict(scenario=name, dtype=dtype, first_steps_loss_rel_dev=dev[:k].tolist(), first_epoch_loss_rel_dev_max=float(dev.max()),
                  reference_self_first_epoch_loss_rel_dev_max=float(ref_dev.max()),
                  product_batch_last_acc=got["batch_last_acc"].tolist(), reference_batch_last_acc=ref["batch_last_acc"].tolist(),
                  reference_perturbed_batch_last_acc=ref["perturbed_batch_last_acc"].tolist(),
                  final_task_gap_points=gap_last, overall_avg_gap_points=gap_avg, reference_self_spread_final_task=spread_last,
                  reference_self_spread_overall_avg=spread_avg, band_final_task=band_last, band_overall_avg=band_avg,
                  product_acc_table=got["acc_table"].tolist(), reference_acc_table=ref["acc_table"].tolist())
    out = os.path.join(os.path.dirname(HERE), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "accuracy_parity.json")
    prev = json.load(open(path)) if os.path.exists(path) else {}
    prev[f"{name}/{dtype}"] = report
    json.dump(prev, open(path, "w"), indent=1)
    assert dev[:k].max() < tol, report
    assert abs(gap_last) <= band_last + 1e-9, report
    assert abs(gap_avg) <= band_avg + 1e-9, report
    if "buffer_labels" in ref.files:                                       # rehearsal buffer: same size, same per-class composition
        assert sorted(got["buffer_labels"].tolist()) == sorted(ref["buffer_labels"].tolist())
