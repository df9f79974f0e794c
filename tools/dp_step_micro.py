"""what the data-parallel machinery costs a batch-256 ResNet-18 step on ONE GPU (1-rank RCCL group, the reducer told the world is 2):
python tools/dp_step_micro.py   -> ms per step: plain | reducer without the early tail | reducer with the overlapped tail (the product's path)"""
import os, sys, socket, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libcontinual_amd.model as M
import torch, torch.distributed as dist
from libcontinual_amd import optim, parallel
from libcontinual_amd.trainer import train_steps

s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
import libcontinual_amd
print("hw_queue_cap", libcontinual_amd.hw_queue_cap_state())        # read at import: WORLD_SIZE > 1 in the caller's environment gives the cap of a rank (4)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)


def make():
    torch.manual_seed(5)
    bb = M.resnet18(args={"dataset": "cifar100"}, dtype="bf16")
    m = M.LWF(bb, 512, 100, device="cuda", init_cls_num=50, inc_cls_num=5).to("cuda")
    m.before_task(0, None, None, None)
    m.train()
    return m


def batch(seed, B=256):
    g = torch.Generator().manual_seed(seed)
    return {"image": torch.randn(B, 3, 32, 32, generator=g).cuda(), "label": torch.randint(0, 50, (B,), generator=g).cuda()}


def run(kind, steps=60):
    m = make()
    o = optim.SGD(m.get_parameters({}), lr=0.05, momentum=0.9)
    red = None
    if kind != "plain":
        red = parallel.GradientReducer()
        red.world = 2
        parallel.attach(m, o, red)
        o.grad_scale = 1.0
        if kind == "no_tail":
            red.overlap = lambda module, fraction=0.5: contextlib.nullcontext()
    bs = [batch(20 + i) for i in range(4)]
    train_steps(m, o, (bs[i % 4] for i in range(10)), red, "LWF", None, "cuda")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    train_steps(m, o, (bs[i % 4] for i in range(steps)), red, "LWF", None, "cuda")
    e1.record(); torch.cuda.synchronize()
    if os.environ.get("DP_MICRO_PTR"):
        bb = m.backbone
        print("   ws %x  shadow %x  flat %x  gflat %x" % (bb._ws.data_ptr(), bb._shadow.data_ptr(), bb._flat.data_ptr(), bb._gflat.data_ptr()),
              " reserved %.2f GB" % (torch.cuda.memory_reserved() / 2**30))
    ms = e0.elapsed_time(e1) / steps
    if os.environ.get("DP_MICRO_FREE"):                     # hand the memory back, so that the next model lands on the same addresses
        import gc
        del m, o, red, bs
        gc.collect(); torch.cuda.empty_cache()
    return ms


for kind in os.environ.get("DP_MICRO_KINDS", "plain,no_tail,tail,plain,no_tail,tail").split(","):
    print(kind, f"{run(kind):.3f} ms")
dist.destroy_process_group()
