"""Runs of the REFERENCE'S OWN Trainer.train_loop on the low-variance accuracy scenarios (oracle/trainer_scenarios.py: ACC_SCENARIOS),
several at a time in worker processes (TEST INFRASTRUCTURE; build container only: needs /root/reference).

    python -m oracle.acc_runs acc_ewc acc_lwf acc_icarl11 [--runs 10] [--procs 4] [--explore]

writes tests/golden/trainer_<name>.npz: run 0 = the unperturbed start (hook trace, per-step losses, accuracy table), runs 1.. = the
same algorithm from initial weights moved by one part in 10^6 -- the reference's own run-to-run spread, which the product's runs are
compared with (tests/test_accuracy_parity_gpu.py).  `--explore` only prints the accuracies (scenario design)."""
import argparse
import os
import sys
import time

import numpy as np


def _cache_path(cache, name, perturb):
    return os.path.join(cache, f"{name}_run{perturb:03d}.npz")


def _one(job):
    name, perturb, threads, cache = job
    if cache and os.path.exists(_cache_path(cache, name, perturb)):          # an interrupted batch of runs resumes where it stopped
        d = np.load(_cache_path(cache, name, perturb), allow_pickle=False)
        return name, perturb, {k: d[k] for k in d.files}
    import torch
    torch.set_num_threads(threads)
    from . import fixtures, gen_golden, trainer_scenarios as ts
    t0 = time.time()
    with fixtures.use_dtype(torch.float32):
        out = ts.run_reference(name, gen_golden.reference_namespace(), ts.common_of(name), perturb=perturb)
    out["seconds"] = np.asarray([time.time() - t0])
    if cache:
        os.makedirs(cache, exist_ok=True)
        tmp = _cache_path(cache, name, perturb) + ".tmp.npz"
        np.savez_compressed(tmp, **{k: np.asarray(v) for k, v in out.items()})
        os.replace(tmp, _cache_path(cache, name, perturb))
    return name, perturb, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="+")
    ap.add_argument("--runs", type=int, default=10)
    ap.add_argument("--procs", type=int, default=4)
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--explore", action="store_true")
    ap.add_argument("--cache", default=None, help="directory of per-run results: finished runs are re-used, so an interrupted batch resumes")
    a = ap.parse_args()
    import multiprocessing as mp
    jobs = [(n, k, a.threads, a.cache) for n in a.names for k in range(a.runs)]
    res = {n: {} for n in a.names}
    with mp.get_context("spawn").Pool(a.procs) as pool:
        for name, k, out in pool.imap_unordered(_one, jobs):
            res[name][k] = out
            print(f"{name} run {k}: final avg acc {out['batch_last_acc'][-1]:.2f}  overall {out['overall_avg_acc'][0]:.3f}  per-task last {out['batch_last_acc'].round(2).tolist()}  ({out['seconds'][0]:.0f} s)", flush=True)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for n in a.names:
        runs = [res[n][k] for k in range(a.runs)]
        last = np.asarray([r["batch_last_acc"][-1] for r in runs])
        avg = np.asarray([r["overall_avg_acc"][0] for r in runs])
        print(f"== {n}: final avg acc mean {last.mean():.3f} std {last.std(ddof=1) if len(last) > 1 else 0:.3f}   overall mean {avg.mean():.3f} std {avg.std(ddof=1) if len(avg) > 1 else 0:.3f}")
        if a.explore:
            continue
        base = runs[0]
        n0 = int(base["trace"][2][2])
        fixture = dict(trace=base["trace"], losses=base["losses"], batch_last_acc=base["batch_last_acc"], acc_table=base["acc_table"],
                       overall_avg_acc=base["overall_avg_acc"], n_validations=base["n_validations"],
                       runs_final_avg_acc=last, runs_overall_avg_acc=avg, runs_batch_last_acc=np.stack([r["batch_last_acc"] for r in runs]),
                       runs_losses_first_epoch=np.stack([r["losses"][:n0] for r in runs]))
        for key in ("buffer_labels", "buffer_images"):
            if key in base:
                fixture[key] = base[key]
        path = os.path.join(root, "tests", "golden", f"trainer_{n}.npz")
        np.savez_compressed(path, **fixture)
        print(f"wrote {path}: {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
