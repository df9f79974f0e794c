"""the data-parallel step of the CIFAR backbones at 32 images per rank (BASELINE configs[2]: iCaRL / CifarResNet-32, 8 ranks), eager against
replayed from ONE HIP graph (backward + RCCL all-reduce + fused optimizer: trainer.GraphedStep with a reducer), on a 1-rank RCCL group with
the reducer told the world is 2:   python tools/dp_graph_micro.py   -> ms per step: plain replay | reduced eager | reduced replay"""
import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libcontinual_amd.model as M
import torch, torch.distributed as dist
from libcontinual_amd import optim, parallel
from libcontinual_amd.trainer import train_steps

s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
B = int(os.environ.get("DP_MICRO_BATCH", "32"))


def make(kind):
    torch.manual_seed(5)
    bb = M.cifar_resnet32(dtype="bf16")
    if kind == "ewc":
        m = M.EWC(bb, 64, 100, device="cuda", init_cls_num=50, inc_cls_num=5, lamda=100.0).to("cuda")
    else:
        m = M.ICarl(bb, 64, 100, device="cuda", init_cls_num=50, inc_cls_num=5).to("cuda") if hasattr(M, "ICarl") else None
    m.before_task(0, None, None, None)
    m.train()
    return m


def batch(seed):
    g = torch.Generator().manual_seed(seed)
    return {"image": torch.randn(B, 3, 32, 32, generator=g).cuda(), "label": torch.randint(0, 50, (B,), generator=g).cuda()}


def run(reduced, graph, steps=300):
    os.environ["CLHIP_CUDA_GRAPH"] = "1" if graph else "0"
    m = make("ewc")
    o = optim.SGD(m.get_parameters({}), lr=0.05, momentum=0.9)
    red = None
    if reduced:
        red = parallel.GradientReducer()
        red.world = 2
        parallel.attach(m, o, red)
    bs = [batch(20 + i) for i in range(4)]
    train_steps(m, o, (bs[i % 4] for i in range(20)), red, "EWC", None, "cuda")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    train_steps(m, o, (bs[i % 4] for i in range(steps)), red, "EWC", None, "cuda")
    e1.record(); torch.cuda.synchronize()
    gs = getattr(m, "_graphed_step", None)
    return e0.elapsed_time(e1) / steps, bool(gs is not None and gs.graphs and not gs.disabled)


for name, r, g in (("plain replay", False, True), ("reduced eager", True, False), ("reduced replay", True, True), ("plain eager", False, False),
                   ("reduced eager", True, False), ("reduced replay", True, True)):
    ms, rep = run(r, g)
    print(f"{name}: {ms:.3f} ms per step of {B} images (replayed: {rep})")
dist.destroy_process_group()
