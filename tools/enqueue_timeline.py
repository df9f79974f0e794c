"""host-side enqueue timeline of a bench.py run: wall time between consecutive batches handed to train_steps (a stall shows up as one
long gap).  python tools/enqueue_timeline.py [bench args];  NOGC=1 disables the Python garbage collector (how the 85 ms generation-2
stalls were identified before libcontinual_amd.utils.quiesce_gc)."""
import sys, time
sys.path.insert(0, "/root/repo")
sys.argv = ["bench.py", "--no-cpu-baseline"] + (sys.argv[1:] or ["--workload", "lwf_resnet18_b50_task1"])
import bench, torch
import libcontinual_amd.trainer as T
orig = T.train_steps
log = []
def wrapped(model, opt, batches, *a, **k):
    st = []
    def gen():
        for b in batches:
            st.append(time.perf_counter()); yield b
    t0 = time.perf_counter(); r = orig(model, opt, gen(), *a, **k); t1 = time.perf_counter()
    log.append((t0, st, t1)); return r
T.train_steps = wrapped
import gc
if len(sys.argv) > 0 and __import__("os").environ.get("NOGC"): gc.disable()
bench.main()
for t0, st, t1 in log:
    d = [round((st[i + 1] - st[i]) * 1e3, 2) for i in range(len(st) - 1)]
    print("first", round((st[0] - t0) * 1e3, 2), d, "tail", round((t1 - st[-1]) * 1e3, 2))
