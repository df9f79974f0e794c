"""Final average accuracy of the product's Trainer against the REFERENCE'S OWN Trainer.train_loop, at the resolution BASELINE.json asks
for: +-0.3 points.

Scenarios (oracle/trainer_scenarios.py: ACC_SCENARIOS; fixtures tests/golden/trainer_acc_*.npz from `python -m oracle.acc_runs`): every
task trained to convergence (learning rate decayed to 1e-3 of its start), >= 100 test images per class, scored on `avg_acc` over ALL
seen classes after the last task (core/trainer.py:715-720), TEN reference runs (the unperturbed start + nine from initial weights moved
by one part in 10^6) so that the reference's own run-to-run spread is part of the fixture:
  * acc_icarl11 -- iCaRL / CifarResNet-32 in the B50-5x10 SHAPE: half of the classes in task 0, ten increments (11 tasks), herded
    rehearsal buffer read back from PNG files, NCM classification;
  * acc_lwf     -- LwF / ResNet-18 (CIFAR stem; BASELINE configs[1]), 20 + 5 classes, two tasks.  Gated quantity: the average accuracy
    AFTER TASK 0 -- twelve epochs of exactly the step bench.py times.  The figure after task 1 is recorded and compared in a band of
    0.3 + 3 SE only: without rehearsal the reference's own class-incremental accuracy is chaotic (its ten runs of this scenario spread
    by several points after ONE increment; trained on four tasks LwF spread 7.8 and EWC 14.9 points, std over 1e-6-perturbed starts --
    the new logits are trained on their own slice and their calibration against the old ones is arbitrary).  That is a property of the
    reference, measured with the reference's own classes (oracle/trainer_scenarios.py records the numbers), and the reason there is no
    EWC scenario here (tests/test_trainer_trace_gpu.py keeps its loose distribution check).
  * acc_icarl11_overlap (round 5) -- the same shape on data with REAL class overlap (`mix` = 0.58 in
    oracle/trainer_scenarios.py: every sample blends its class pattern with its partner class's, blend weight uniform in [0, 0.58)): the
    reference's own twenty-four runs end at 85.4 +- 0.56 % (overall 85.3 +- 0.28; the distribution has a lower tail -- 83.8 at worst -- that
    twelve runs did not show and the product's runs do too) -- far from saturation, and the samples near a blend weight of 0.5 are decided
    by the details of the trained network.  Sixteen bf16 / eight f32 product runs in the DEFAULT suite, twenty-four with CLHIP_ACC_RUNS=full.
    Since round 6 its three quantities are gated on the mean alone: |mean(product) - mean(reference)| <= 0.3, the standard error reported beside it.
The gate, two-sided, in BOTH arithmetic modes (f32 = like for like with the reference, bf16 = the benchmarked mode):
      |mean(product) - mean(reference)| <= 0.3 + 2 SE,   SE = sqrt(var_ref / 10 + var_prod / 10),
and a quantity only counts as a gate if the runs can resolve the band: SE <= 0.15 (asserted -- a quantity that cannot is a finding, not
a pass).  The hook sequence and the first optimisation steps (2e-4 f32 / 3e-2 bf16) are checked on the unperturbed run as well.
Every run's figures go to gpurun_out/accuracy_parity_r06[_quick].json (copied to profiles/)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import trainer_scenarios as ts                # noqa: E402
from test_trainer_trace_gpu import run_product   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
N_RUNS = 10
# round 4: the LwF scenario with 40 runs per side (its after-one-increment figure is chaotic in the reference itself: std 4.3
# points -- ten runs could not tell a -4-point bf16 bias from nothing), and an UNSATURATED rehearsal scenario (acc_icarl11_hard: the reference's own
# runs sit at 99.3 +- 0.6, not at 99.9) with 24 runs per side.  A scenario runs as many product runs as its fixture holds reference runs, up to this cap.
RUN_CAP = {"acc_lwf": 40, "acc_icarl11": 10, "acc_icarl11_hard": 24, "acc_icarl11_overlap": 24}
# ... with CLHIP_ACC_RUNS=full (the evidence run behind profiles/r04_accuracy_parity.json: 16 minutes of GPU time for the three scenarios).  The default
# suite keeps the same gates (bands are 0.3 + 2 SE of the runs actually made) on fewer product runs so that `pytest -m gpu` stays a quarter of an hour.
if os.environ.get("CLHIP_ACC_RUNS", "") != "full":
    RUN_CAP = {"acc_lwf": 10, "acc_icarl11": 4, "acc_icarl11_hard": 4, "acc_icarl11_overlap": 16}      # (the 70-90 % scenario keeps most of its runs: it is the gate that bites)
# round 6: the f32 MODE of the overlap scenario -- not the product's default arithmetic -- makes half of them in the default suite
RUN_CAP_F32 = {"acc_icarl11_overlap": 8} if os.environ.get("CLHIP_ACC_RUNS", "") != "full" else {}
# first optimisation steps of the unperturbed run: these scenarios step at lr 0.05 (the short ones at 0.02), so the chaotic amplification
# sets in one step earlier -- f32 mode observed 1.5e-7, 1.8e-5, then 9e-4 at the third step
FIRST_STEPS_ACC = {"f32": (2, 2e-4), "bf16": (2, 3e-2)}


@pytest.mark.parametrize("name", ["acc_icarl11", "acc_lwf", "acc_icarl11_hard", "acc_icarl11_overlap"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_final_average_accuracy_within_the_band(name, dtype, tmp_path):
    path = os.path.join(HERE, "golden", f"trainer_{name}.npz")
    if not os.path.exists(path):
        pytest.skip(f"no fixture for {name} (python -m oracle.acc_runs {name} in the build container)")
    ref = np.load(path)
    ref_final, ref_overall = ref["runs_final_avg_acc"], ref["runs_overall_avg_acc"]
    assert len(ref_final) >= N_RUNS
    n_runs = min(len(ref_final), RUN_CAP_F32.get(name, RUN_CAP[name]) if dtype == "f32" else RUN_CAP[name])
    prod_final, prod_overall, prod_task0 = [], [], []
    for q in range(n_runs):
        got, _ = run_product(name, dtype, str(tmp_path / "data"), perturb=q)            # (one data root: the scenario's files do not depend on the perturbation)
        if q == 0:
            assert got["trace"].tolist() == ref["trace"].tolist()
            n0 = int(ref["trace"][2][2])
            dev = np.abs(got["losses"][:n0] - ref["losses"][:n0]) / np.abs(ref["losses"][:n0])
            k, tol = FIRST_STEPS_ACC[dtype]
            assert dev[:k].max() < tol, (dev[:k], tol)
            if "buffer_labels" in ref.files:
                assert sorted(got["buffer_labels"].tolist()) == sorted(ref["buffer_labels"].tolist())
        prod_final.append(float(got["batch_last_acc"][-1]))
        prod_overall.append(float(got["overall_avg_acc"][0]))
        prod_task0.append(float(got["batch_last_acc"][0]))
    R, P = len(ref_final), len(prod_final)
    ref_task0 = ref["runs_batch_last_acc"][:, 0]
    report = dict(scenario=name, dtype=dtype, product_final_avg_acc_runs=prod_final, reference_final_avg_acc_runs=ref_final.tolist(),
                  product_overall_avg_acc_runs=prod_overall, reference_overall_avg_acc_runs=ref_overall.tolist(),
                  product_task0_avg_acc_runs=prod_task0, reference_task0_avg_acc_runs=ref_task0.tolist())
    for key, pr, rf in (("final_avg_acc", np.asarray(prod_final), ref_final), ("overall_avg_acc", np.asarray(prod_overall), ref_overall),
                        ("task0_avg_acc", np.asarray(prod_task0), ref_task0)):
        se = float(np.sqrt(rf.var(ddof=1) / R + pr.var(ddof=1) / P))
        report[key] = dict(gap_points=float(pr.mean() - rf.mean()), se=se, band=0.3 + 2 * se, reference_mean=float(rf.mean()), reference_std=float(rf.std(ddof=1)),
                           product_mean=float(pr.mean()), product_std=float(pr.std(ddof=1)))
    out = os.path.join(os.path.dirname(HERE), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    jp = os.path.join(out, "accuracy_parity_r06.json" if os.environ.get("CLHIP_ACC_RUNS", "") == "full" else "accuracy_parity_r06_quick.json")
    prev = json.load(open(jp)) if os.path.exists(jp) else {}
    prev[f"{name}/{dtype}"] = report
    json.dump(prev, open(jp, "w"), indent=1)
    print(json.dumps({k: report[k] for k in ("final_avg_acc", "overall_avg_acc", "task0_avg_acc")}))
    gated = {"acc_icarl11": ("final_avg_acc", "overall_avg_acc", "task0_avg_acc"), "acc_lwf": ("task0_avg_acc",),
             "acc_icarl11_hard": ("final_avg_acc", "overall_avg_acc", "task0_avg_acc"),
             "acc_icarl11_overlap": ("final_avg_acc", "overall_avg_acc", "task0_avg_acc")}[name]
    # (the unsaturated scenario: 24 + 24 runs of a std-0.6 quantity; the overlap scenario: 24 + 16..24 runs of a quantity whose std is 0.56 in the reference
    #  and up to 1.0 in the product's f32 runs: SE 0.2-0.3)
    se_cap = 0.2 if name == "acc_icarl11_hard" else (0.35 if name == "acc_icarl11_overlap" else 0.15)
    for key in gated:
        r = report[key]
        assert r["se"] <= se_cap, (key, r)                                 # the runs resolve the band
        if name == "acc_icarl11_overlap":
            # round 6: the scenario that bites is gated on the MEAN alone -- BASELINE.json's +-0.3 points, no allowance for the standard
            # error, which is recorded beside it (`se`, `mean_gate`).  The runs are deterministic (bit-reproducible steps, fixed perturbations), so the gate
            # does not flicker; measured: final -0.13 / -0.19, overall +0.15 / +0.06, first task 0.00 / -0.17 points (f32 / bf16, 16 runs), and -0.22 /
            # +0.08 / +0.09 on the eight f32 runs of the default suite
            r["mean_gate"] = 0.3
            assert abs(r["gap_points"]) <= 0.3 + 1e-9, (key, r)
            continue
        assert abs(r["gap_points"]) <= r["band"] + 1e-9, (key, r)          # BASELINE.json: within +-0.3 points of the CPU reference
    if name == "acc_lwf":
        # the after-one-increment figure with 40 + 40 runs: a gate if they resolve it (SE <= 0.5), the measured bias with its interval otherwise
        r = report["final_avg_acc"]
        r["ci95_points"] = [r["gap_points"] - 1.96 * r["se"], r["gap_points"] + 1.96 * r["se"]]
        json.dump(prev, open(jp, "w"), indent=1)
        print("LwF after one increment, product - reference:", r["gap_points"], "+-", 1.96 * r["se"], f"({n_runs} + {R} runs)")
        if r["se"] <= 0.5:
            assert abs(r["gap_points"]) <= 0.3 + 2 * r["se"] + 1e-9, r
    for key in ("final_avg_acc", "overall_avg_acc"):                       # recorded quantities the reference itself cannot pin to 0.3: 3 SE
        r = report[key]
        assert abs(r["gap_points"]) <= 0.3 + 3 * r["se"] + 1e-9, (key, r)
