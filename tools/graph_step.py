"""experiment: one training step (observe -> zero_grad -> backward -> step) captured into a HIP graph and replayed, against the eager
enqueue: python tools/graph_step.py [workload] [batch] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CLHIP_CUDA_GRAPH"] = "0"      # the eager leg really is eager; the graphed leg below is this tool's own capture
import torch
import bench
from libcontinual_amd.trainer import train_steps, _OBSERVE_DOES_BACKWARD
from libcontinual_amd import ops

wl = sys.argv[1] if len(sys.argv) > 1 else "ewc_resnet32_b50_task1"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
torch.manual_seed(1993)
model, opt, arch, teacher, (lo, hi) = bench.build_method(wl, "bf16", dev)
from libcontinual_amd import parallel
parallel.attach(model, opt, None)
model.train()
name = type(model).__name__
batches = [bench.synthetic_batch(B, lo, hi, 100 + i, dev, 32) for i in range(4)]


def eager(n):
    train_steps(model, opt, (batches[i % 4] for i in range(n)), None, name, None, dev)


eager(20); torch.cuda.synchronize()
t0 = time.perf_counter(); eager(steps); torch.cuda.synchronize(); te = (time.perf_counter() - t0) / steps
print(f"eager  : {te*1e3:.3f} ms/step  {B/te:.0f} img/s")

static = {k: v.clone() for k, v in batches[0].items() if torch.is_tensor(v)}


def one_step():
    with ops.deferred_metrics(True):
        if name in _OBSERVE_DOES_BACKWARD:
            opt.zero_grad()
            out, acc, loss = model.observe(static)
        else:
            out, acc, loss = model.observe(static)
            opt.zero_grad()
            loss.backward()
        opt.step()
    return loss


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        one_step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = one_step()
torch.cuda.synchronize()


def graphed(n):
    for i in range(n):
        b = batches[i % 4]
        for k in static:
            static[k].copy_(b[k], non_blocking=True)
        g.replay()


def lossval(l):
    t = l.tensor if hasattr(l, "tensor") else l
    return float(t.detach().float().reshape(-1)[0]) * (l.scale if hasattr(l, "scale") else 1.0)


for i in range(6):
    graphed(1); torch.cuda.synchronize()
    pn = sum(float(p.detach().float().pow(2).sum()) for p in model.parameters()) ** 0.5
    print(f"replay {i}: loss {lossval(loss):.5f} type {type(loss).__name__} |params| {pn:.4f}")
graphed(20); torch.cuda.synchronize()
t0 = time.perf_counter(); graphed(steps); torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / steps
print(f"graphed: {tg*1e3:.3f} ms/step  {B/tg:.0f} img/s   loss {lossval(loss):.4f}")
