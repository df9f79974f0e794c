"""bench.py -- images/sec of the hot path on N GPUs of one node (driver contract in the task statement).

Workload (BASELINE.json configs[1]): LwF, ResNet-18 with the CIFAR stem, CIFAR-100 B50-5x10, bf16, task-0
epoch step = forward + fused CE + backward + fused SGD on a synthetic batch already resident in HBM
([B,3,32,32] fp32 normalised images, labels in [0,50)).  One "step" = one pass of the inner loop
(`libcontinual_amd.trainer.train_steps`, the same function the Trainer runs) over one batch.
N > 1: one process per GPU (torchrun), per-GPU batch fixed (weak scaling), one all-reduce of the flat gradient
buffer over RCCL per step, 1/world folded into the SGD kernel.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, timed live with HIP events) and
`cpu_baseline` (the CPU oracle of the same step on the host cores, bounded sample) objects.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (must be set before the HIP runtime starts)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_FWD_PER_IMG = {"resnet18": 1.1108e9, "cifar_resnet32": 0.13825e9}     # 2*MAC, SURVEY.md section 8(d)
# whole-step algorithmic FLOP per image of the ViT-B/16 methods: L2P = query fwd (N=197) + prompted
# fwd (N=222) + activation-only bwd; InfLoRA_OPT = fwd + activation bwd + rank-10 dB (the reference's dense qkv dW not counted)
FLOP_STEP_PER_IMG = {"l2p_vitb16": 116.2e9, "inflora_vitb16": 71.5e9}
PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: 256 ResNet workloads, 16 L2P, 128 InfLoRA_OPT)")
    ap.add_argument("--workload", default="lwf_resnet18_b50_task0",
                    choices=["lwf_resnet18_b50_task0", "lwf_resnet18_b50_task1", "icarl_resnet32_b50_task1", "ewc_resnet32_b50_task1",
                             "l2p_vitb16_b10_task1", "inflora_vitb16_b20_task1"])
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)
    return ap.parse_args()


def build_method(workload, dtype, dev):
    """plugin + optimizer in the state of the named task (task>=1: teacher / Fisher active)"""
    import libcontinual_amd.model as M
    from libcontinual_amd import optim
    if workload.startswith("l2p_vitb16"):
        # config/l2p-vit-cifar100-b10-10-10.yaml: ViT-B/16, pool 10, top-5, length 5, Adam(1.875e-3), task 1 = classes 10..19
        bb = M.vit_pt_imnet(pretrained=False, dtype=dtype)
        m = M.L2P(bb, dev, init_cls_num=10, inc_cls_num=10, num_class=100, task_num=10, feat_dim=768, prompt_length=5, pool_size=10, top_k=5,
                  pull_constraint_coeff=1.0)
        m.before_task(0, None, None, None); m.after_task(0, None, None, None); m.before_task(1, None, None, None)
        opt = optim.Adam(m.get_parameters({}), lr=0.001875, betas=(0.9, 0.999), weight_decay=0)
        return m, opt, "l2p_vitb16", False, (10, 20)
    if workload.startswith("inflora_vitb16"):
        # config/InfLoRA_opt-vit-imagenetr-b20-20-10.yaml: ViT-B/16 + rank-10 LoRA on k,v; SGD(8e-3, m .9); task 1 = classes 20..39
        os.environ.setdefault("PYTHONHASHSEED", "0")
        bb = M.vit_pt_imnet(pretrained=False, attn_layer="MultiHeadAttention_LoRA", lora_rank=10, dtype=dtype)
        m = M.InfLoRA_OPT(bb, dev, init_cls_num=20, inc_cls_num=20, task_num=10, lame=1.0, lamb=0.95, dataset="imagenet-r", use_ca=False, embd_dim=768)
        m._network._cur_task_id, m._known_classes = 1, 20
        for a in m.attention_modules:
            a.init_param()
            torch.nn.init.normal_(a.lora_B_k.weight, std=1e-3); torch.nn.init.normal_(a.lora_B_v.weight, std=1e-3)
        for name, prm in m._network.named_parameters():
            prm.requires_grad_("classifier_pool.1." in name or "lora_B" in name)
        m._network.to(dev)
        opt = optim.SGD(m.get_parameters({}), lr=8e-3, momentum=0.9)
        return m, opt, "inflora_vitb16", False, (20, 40)
    if workload.startswith("lwf_resnet18"):
        bb = M.resnet18(args={"dataset": "cifar100"}, dtype=dtype)
        m = M.LWF(bb, 512, 100, device=dev, init_cls_num=50, inc_cls_num=5).to(dev)
        m.before_task(0, None, None, None)
        lo, hi = 0, 50
        if workload.endswith("task1"):
            m.before_task(1, None, None, None)
            lo, hi = 50, 55
        opt = optim.SGD(m.get_parameters({}), lr=0.1)                      # config/lwf.yaml:14-17
        arch, teacher = "resnet18", workload.endswith("task1")
    elif workload.startswith("icarl_resnet32"):
        bb = M.cifar_resnet32(dtype=dtype)
        m = M.ICarl(bb, 64, 100, device=dev, init_cls_num=50, inc_cls_num=5, task_num=11).to(dev)
        m.before_task(0, None, None, None)
        import copy
        m.old_network = copy.deepcopy(m.network).eval()
        m.prev_cls_num, m.cur_task_id = 50, 1
        m.before_task(1, None, None, None)
        lo, hi = 0, 55
        opt = optim.SGD(m.get_parameters({}), lr=0.1, momentum=0.9, weight_decay=5e-4)
        arch, teacher = "cifar_resnet32", True
    else:
        bb = M.cifar_resnet32(dtype=dtype)
        m = M.EWC(bb, 64, 100, device=dev, init_cls_num=50, inc_cls_num=5, lamda=1000).to(dev)
        m.before_task(0, None, None, None)
        m._ensure_state()
        m._fisher_flat.fill_(1e-4)
        m.before_task(1, None, None, None)
        lo, hi = 50, 55
        opt = optim.SGD(m.get_parameters({}), lr=0.1, momentum=0.9, weight_decay=5e-4)
        arch, teacher = "cifar_resnet32", False
    return m, opt, arch, teacher, (lo, hi)


def synthetic_batch(B, lo, hi, seed, dev, size=32):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, size, size, generator=g)
    mean = torch.tensor([0.5071, 0.4866, 0.4409]).view(1, 3, 1, 1)
    std = torch.tensor([0.2675, 0.2565, 0.2761]).view(1, 3, 1, 1)
    x = (x - mean) / std
    y = torch.randint(lo, hi, (B,), generator=g)
    return {"image": x.to(dev), "label": y.to(dev)}


def cpu_baseline(workload, steps, batch):
    """the CPU oracle (torch CPU fp32, all host cores) running the same step; bounded sample"""
    from oracle import methods as om, nets
    arch = "resnet18" if "resnet18" in workload else "cifar_resnet32"
    torch.manual_seed(0)
    P = {k: v.requires_grad_(True) for k, v in nets.init_params(arch).items()}
    Bf = nets.init_buffers(arch)
    fd = nets.arch(arch)[1]
    w, b = om.linear_default_init(50, fd)
    net = om.Net(arch, P, Bf, w.requires_grad_(True), b.requires_grad_(True))
    m = om.LWF(net, 50, 5)
    m.before_task(0, (w, b))
    opt = om.SGD(net.parameters(), 0.1)
    bt = synthetic_batch(batch, 0, 50, 1, "cpu")
    x, y = bt["image"], bt["label"]

    def step():
        _, _, loss = m.observe(x, y, True)
        opt.zero_grad(); loss.backward(); opt.step()
    step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = time.perf_counter() - t0
    return dict(value=batch * steps / dt, unit="images/sec", cores=torch.get_num_threads(), kind="port",
                sample=f"{steps} steps of batch {batch} (LwF task-0 step, {arch}, fp32 torch-CPU oracle), 1 warm-up")


def dominant_kernel_roofline(dev, dtype, B):
    """time the dominant kernel family of the ResNet-18 step -- the 3x3 stride-1 implicit-GEMM conv
    forward at the layer1 shape [B,32,32,64]->64 -- with HIP events on the launch stream and price it
    against the bf16 MFMA peak; algorithmic FLOPs = 2 * M * (9*Cin) * Cout per launch."""
    from libcontinual_amd import _lib
    code = _lib.BF16 if dtype == "bf16" else _lib.F32
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    N, H, W, C, K = B, 32, 32, 64, 64
    x = torch.randn(N, H, W, C, device=dev).to(tdt)
    w = (torch.randn(K, 9, C, device=dev) * 0.05).to(tdt)
    z = torch.empty(N, H, W, K, device=dev, dtype=tdt)
    tiles = _lib.lib().clhip_conv_fwd_tiles(N, H, W, C, K, 3, 1, 1)
    part = torch.empty(tiles, 2, K, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: _lib.call("clhip_conv_fwd", x.data_ptr(), w.data_ptr(), z.data_ptr(), part.data_ptr(), N, H, W, C, K, 3, 1, 1, code, st)
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * N * H * W * 9 * C * K
    alg_bytes = (N * H * W * C + N * H * W * K) * (2 if dtype == "bf16" else 4) + w.numel() * w.element_size()
    ach = flops / (ms * 1e-3) / 1e12
    # HBM traffic per launch: PMC passes (FETCH_SIZE x2 + WRITE_SIZE, tools/roofline_pmc.sh) of this same kernel / shape,
    # committed under profiles/; null if no matching measurement is on disk
    traffic, src = None, None
    pj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_roofline_pmc.json")
    if os.path.exists(pj):
        with open(pj) as f:
            m = json.load(f)
        if m.get("shape") == [N, H, W, C, K, 3, 1] and m.get("dtype") == dtype:
            traffic, src = m["traffic_bytes_per_launch"], "profiles/r01_roofline_pmc.json (rocprofv3 --pmc, separate passes)"
    kname = "conv3_kernel<4,1,0>" if dtype == "bf16" else "conv_igemm2_kernel<float>"
    return dict(bound="mfma", kernel=kname + " fwd 3x3/s1 + BN-stat epilogue @ [B,32,32,64]x[64,3,3,64]", achieved=ach,
                peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_BF16_TFLOPS, traffic=traffic, traffic_source=src,
                launch_ms=ms, algorithmic_flops_per_launch=flops, algorithmic_bytes_per_launch=alg_bytes,
                hbm_gbs_algorithmic=alg_bytes / (ms * 1e-3) / 1e9)


def gemm_roofline(dev, dtype, M):
    """dominant kernel of the ViT workloads: the fc1 GEMM [M,768] x [3072,768]^T with the bias+GELU epilogue, timed with HIP
    events on the launch stream; algorithmic FLOPs = 2*M*768*3072"""
    from libcontinual_amd import _lib
    code = _lib.BF16 if dtype == "bf16" else _lib.F32
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    N, K = 3072, 768
    A = torch.randn(M, K, device=dev).to(tdt)
    W = (torch.randn(N, K, device=dev) * 0.03).to(tdt)
    bias = torch.zeros(N, device=dev)
    Cc = torch.empty(M, N, device=dev, dtype=tdt)
    H = torch.empty(M, N, device=dev, dtype=tdt)
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: _lib.call("clhip_gemm_nt", A.data_ptr(), W.data_ptr(), Cc.data_ptr(), bias.data_ptr(), None, H.data_ptr(), M, N, K, K, K, N, 0, N, 3, code, st)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * M * N * K
    es = 2 if dtype == "bf16" else 4
    alg_bytes = (M * K + N * K + 2 * M * N) * es
    ach = flops / (ms * 1e-3) / 1e12
    # HBM traffic per launch: PMC passes (FETCH_SIZE x2 + WRITE_SIZE, tools/roofline_pmc_gemm.sh) of this kernel at this shape, committed
    # under profiles/; null if no matching measurement is on disk
    traffic, src = None, None
    pj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_roofline_pmc_gemm.json")
    if os.path.exists(pj):
        with open(pj) as f:
            for m in json.load(f):
                if m.get("shape") == [M, N, K] and m.get("dtype") == dtype:
                    traffic, src = m["traffic_bytes_per_launch"], "profiles/r01_roofline_pmc_gemm.json (rocprofv3 --pmc, separate passes)"
    return dict(bound="mfma", kernel=f"gemm_nt_kernel<{dtype}, bias+GELU> fc1 @ [{M},768]x[3072,768]^T", achieved=ach, peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                frac=ach / PEAK_BF16_TFLOPS, traffic=traffic, traffic_source=src, launch_ms=ms, algorithmic_flops_per_launch=flops, algorithmic_bytes_per_launch=alg_bytes,
                hbm_gbs_algorithmic=alg_bytes / (ms * 1e-3) / 1e9)


def main():
    a = parse()
    vit = "vitb16" in a.workload
    if a.batch is None:
        a.batch = 16 if a.workload.startswith("l2p") else (128 if vit else 256)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if os.environ.get("CLHIP_SHARED_GPU") else int(os.environ.get("LOCAL_RANK", "0"))     # test hook: ranks share cuda:0
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    import torch.distributed as dist
    from libcontinual_amd import parallel
    from libcontinual_amd.trainer import train_steps
    from libcontinual_amd.utils import AverageMeter
    reducer = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        parallel.init_distributed(True)
        reducer = parallel.GradientReducer()
    torch.manual_seed(1993)
    model, opt, arch, teacher, (lo, hi) = build_method(a.workload, a.dtype, dev)
    if world > 1:
        parallel.broadcast_module_state(model)
    parallel.attach(model, opt, reducer)
    model.train()
    batches = [synthetic_batch(a.batch, lo, hi, 100 + rank * 7 + i, dev, 224 if vit else 32) for i in range(4)]
    name = type(model).__name__
    meter = AverageMeter("train", ["loss", "acc1"])

    def run(n):
        train_steps(model, opt, (batches[i % len(batches)] for i in range(n)), reducer, name, meter, dev)

    # the dominant kernel's roofline measurement (HIP events around repeated launches of that kernel alone) runs first: it is
    # independent of the step, and the GPU enters the timed region at its sustained clocks instead of waking up in it
    roofline = gemm_roofline(dev, a.dtype, a.batch * (222 if a.workload.startswith("l2p") else 197)) if vit else dominant_kernel_roofline(dev, a.dtype, a.batch)
    from libcontinual_amd.utils import quiesce_gc
    quiesce_gc()          # what Trainer.train_loop does after building a task's optimizer (no 80 ms generation-2 GC stalls mid-epoch)
    run(a.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_avg = meter.avg("loss")
    if rank != 0:
        return
    ips = world * a.batch * a.steps / dt
    if vit:
        step_flops_per_img = FLOP_STEP_PER_IMG[arch]
    else:
        fwd = FLOP_FWD_PER_IMG[arch]
        step_flops_per_img = 3 * fwd + (fwd if teacher else 0.0)
    out = {
        "metric": ("images/sec/node (L2P task>=1 step), ViT-B/16, CIFAR-100 B10-10x10" if a.workload.startswith("l2p") else
                   "images/sec/node (InfLoRA_OPT task>=1 step), ViT-B/16, ImageNet-R B20-20x10" if vit else
                   "images/sec/node (task-0 epoch), CIFAR-100 B50-5x10" if a.workload.endswith("task0") else
                   "images/sec/node (task>=1 step), CIFAR-100 B50-5x10"),
        "value": ips, "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": a.workload, "method": name, "backbone": "vit_base_patch16_224" if vit else arch, "per_gpu_batch": a.batch,
                   "global_batch": a.batch * world, "image": "3x224x224" if vit else "3x32x32", "parallelism": f"dp{world}", "optimizer": "fused " + type(opt).__name__, "final_loss": loss_avg},
        "step_tflops_algorithmic": step_flops_per_img * ips / 1e12,
        "step_frac_of_bf16_mfma_peak": step_flops_per_img * ips / 1e12 / (PEAK_BF16_TFLOPS * world),
    }
    out["roofline"] = roofline
    if not a.no_cpu_baseline and not vit:
        out["cpu_baseline"] = cpu_baseline(a.workload, a.cpu_steps, 128)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
