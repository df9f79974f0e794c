"""stand-alone timings of the small launches of a training step (HIP events around back-to-back launches on the compute stream):
the fused cross-entropy in its two single-workgroup forms (CE_ROWS switch), at the shapes of the bench workloads.
    python tools/small_kernels.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libcontinual_amd import _lib          # noqa: E402
from libcontinual_amd._lib import call     # noqa: E402


def timeit(fn, reps=300):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    st = torch.cuda.current_stream().cuda_stream
    L = _lib.lib()
    for B, O, lo, hi in ((256, 50, 0, 50), (256, 55, 50, 55), (32, 55, 50, 55), (128, 40, 20, 40), (512, 100, 0, 100), (256, 200, 180, 200)):
        logits = torch.randn(B, O, device="cuda")
        y = torch.randint(lo, hi, (B,), device="cuda")
        loss = torch.zeros(1, device="cuda"); dl = torch.empty_like(logits)
        pred = torch.empty(B, dtype=torch.int64, device="cuda"); corr = torch.zeros(1, dtype=torch.int32, device="cuda")
        run = lambda: call("clhip_ce_window", logits.data_ptr(), y.data_ptr(), B, O, lo, hi, 0, O, 1.0, loss.data_ptr(), 0, dl.data_ptr(), 0, pred.data_ptr(), corr.data_ptr(), st)
        res = []
        for rows in (b"1", b"0"):
            # CE_ROWS is read once per process: the A/B needs two processes -- this script times the form selected by $CLHIP_CE_ROWS
            res.append(timeit(run))
            break
        print(f"ce_window B={B} O={O} window [{lo},{hi}): {res[0]:.2f} us per launch (CLHIP_CE_ROWS={os.environ.get('CLHIP_CE_ROWS', '1')})")


if __name__ == "__main__":
    main()
