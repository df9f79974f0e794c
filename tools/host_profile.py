"""host-side cost of a training step: enqueue time vs total, and a cProfile of 40 steps.  python tools/host_profile.py <workload> <batch>"""
import sys, time, torch, cProfile, pstats
sys.path.insert(0, "/root/repo")
import bench
from libcontinual_amd.trainer import train_steps
from libcontinual_amd.utils import quiesce_gc
dev = torch.device("cuda:0")
wl, B = sys.argv[1], int(sys.argv[2])
m, opt, arch, teacher, (lo, hi) = bench.build_method(wl, "bf16", dev)
m.train()
batches = [bench.synthetic_batch(B, lo, hi, 100 + i, dev, 32) for i in range(4)]
def run(n): train_steps(m, opt, (batches[i % 4] for i in range(n)), None, type(m).__name__, None, dev)
run(5); torch.cuda.synchronize(); quiesce_gc()
t0 = time.perf_counter(); run(40); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(wl, "enqueue %.2f ms/step, total %.2f ms/step" % ((t1 - t0) / 40 * 1e3, (t2 - t0) / 40 * 1e3))
pr = cProfile.Profile(); pr.enable(); run(40); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
