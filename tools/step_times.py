"""GPU-side duration of every step of a bench.py workload (an event per batch on the compute stream): python tools/step_times.py [bench args]
-> percentiles of the per-step time, to tell a uniformly slower run from a few long steps"""
import sys
sys.path.insert(0, "/root/repo")
sys.argv = ["bench.py", "--cpu-steps", "0"] + (sys.argv[1:] or ["--workload", "ewc_resnet32_b50_task1", "--steps", "200"])
import numpy as np
import torch
import bench
import libcontinual_amd.trainer as T

orig, runs = T.train_steps, []


def wrapped(model, opt, batches, *a, **k):
    evs = []

    def gen():
        for b in batches:
            e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
            yield b
    r = orig(model, opt, gen(), *a, **k)
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    runs.append(evs)
    return r


T.train_steps = wrapped
bench.main()
torch.cuda.synchronize()
for evs in runs:
    d = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1)])
    if len(d) < 20:
        continue
    q = np.percentile(d, [0, 10, 50, 90, 99, 100])
    print("steps", len(d), "ms: min %.3f p10 %.3f p50 %.3f p90 %.3f p99 %.3f max %.3f mean %.3f" % (*q, d.mean()))
    print("first 40:", np.round(d[:40], 2).tolist())
