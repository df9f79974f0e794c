"""bench.py -- images/sec of the hot path on N GPUs of one node (driver contract in the task statement).

Workload (BASELINE.json configs[1]): LwF, ResNet-18 with the CIFAR stem, CIFAR-100 B50-5x10, bf16, task-0
epoch step = forward + fused CE + backward + fused SGD on a synthetic batch already resident in HBM
([B,3,32,32] fp32 normalised images, labels in [0,50)).  One "step" = one pass of the inner loop
(`libcontinual_amd.trainer.train_steps`, the same function the Trainer runs) over one batch.
N > 1: one process per GPU (torchrun), one all-reduce of the flat gradient buffer over RCCL per step (tail of the buffer overlapped
with the backward), 1/world folded into the SGD kernel.  `--scaling weak` (default, what the driver runs): per-GPU batch fixed;
`--scaling strong`: the GLOBAL batch is fixed (256 by default, `--batch` = global) and every rank takes batch // N -- the reference's own
per-rank batch rule (core/trainer.py:229-241) and SURVEY.md section 8(d)'s first series.  With N > 1 the line carries a `dp` object:
backend, exchange, and per rank the step time plus a breakdown measured in a separate instrumented pass AFTER the timed region
(backward, exposed exchange = what the stream still waits for once the backward is done, optimizer, the bucket's all-reduce alone).

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
import libcontinual_amd  # noqa: E402,F401   (sets the hardware-queue cap -- 3 on one GPU, 4 for a rank of a multi-process job -- before the first HIP call)

FLOP_FWD_PER_IMG = {"resnet18": 1.1108e9, "cifar_resnet32": 0.13825e9}     # 2*MAC, SURVEY.md section 8(d)
# whole-step algorithmic FLOP per image of the ViT-B/16 methods: L2P = query fwd (N=197) + prompted
# fwd (N=222) + activation-only bwd; InfLoRA_OPT = fwd + activation bwd + rank-10 dB (the reference's dense qkv dW not counted)
FLOP_STEP_PER_IMG = {"l2p_vitb16": 116.2e9, "inflora_vitb16": 71.5e9}
# whole-step ALGORITHMIC HBM bytes per image, ideal-fused bf16 (every conv output written once and read once in the
# forward, read once + its gradient written / read once in the backward); the frozen teacher's forward adds one write + one read
HBM_BYTES_STEP_PER_IMG = {"cifar_resnet32": 3.2e6, "resnet18": 6.0e6}
HBM_BYTES_TEACHER_PER_IMG = {"cifar_resnet32": 1.3e6, "resnet18": 2.4e6}
PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: enough for a timed region of about a second)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (weak scaling) or GLOBAL batch (strong scaling); default: 256 ResNet workloads, 16 L2P, 128 InfLoRA_OPT")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="weak: per-GPU batch fixed as N grows; strong: global batch fixed, per-GPU batch = batch // N")
    ap.add_argument("--workload", default="lwf_resnet18_b50_task0",
                    choices=["lwf_resnet18_b50_task0", "lwf_resnet18_b50_task1", "icarl_resnet32_b50_task1", "ewc_resnet32_b50_task1",
                             "lucir_resnet32_b50_task1", "l2p_vitb16_b10_task1", "inflora_vitb16_b20_task1",
                             # the single-GPU per-task work north_star names (a "step" = one batch of the Fisher pass / one class of the herding)
                             "ewc_fisher_pass", "herding_b50"])
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the live lines of the other BASELINE workloads behind the headline")
    ap.add_argument("--roofline-only", action="store_true", help="run only the roofline kernels' launches (what the PMC passes sample) and print their block")
    ap.add_argument("--cpu-steps", type=int, default=None, help="steps of the CPU baseline sample (default: 5 ResNet, 1-2 ViT)")
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
    elif workload.startswith("lucir_resnet32"):
        # config/lucir-resnet32-cifar100-b50-5-10.yaml: resnet32_V2 + cosine head, task 1 = 50 old + 5 new classes, the frozen previous model's
        # features (less-forget), CE and the hard-negative margin ranking loss (core/model/lucir.py:175-210); fc2 keeps its fresh
        # initialisation here (the imprint of lucir.py:134-159 needs the task's data)
        bb = M.resnet32_V2(dtype=dtype)
        m = M.LUCIR(bb, 64, 100, device=dev, init_cls_num=50, inc_cls_num=5, dist=0.5, lamda=5, K=2, lw_mr=1).to(dev)
        m.before_task(0, None, None, None)
        m._init_new_fc = lambda *a, **k: None
        m.before_task(1, None, None, None)
        lo, hi = 0, 55
        pg = m.get_parameters({})
        opt = optim.SGD(pg, lr=0.1, momentum=0.9, weight_decay=5e-4)
        arch, teacher = "cifar_resnet32", True
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


def _physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return os.cpu_count() or 1


def cpu_baseline(workload, steps, batch):
    """the CPU oracle (torch CPU fp32 restatement of the reference's step, oracle/), the GPU line's own batch size; a bounded sample (about 10-30 s
    of CPU work).  ResNet workloads: timed at 8, 32 and all physical cores (one warm-up + two steps each) and the BEST thread count is reported with `steps`
    more steps at it -- every core of a 128-core host on one 32 x 32 convolution is an oversubscribed run (32-50 img/s on 128 threads, 205 on 8)."""
    cores = _physical_cores()
    torch.set_num_threads(cores)
    if "vitb16" in workload:
        return _cpu_baseline_vit(workload, steps, batch, cores)
    from oracle import methods as om, nets
    arch = "resnet18" if "resnet18" in workload else "cifar_resnet32"
    steps = steps or 5
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
    sweep = {}
    for nt in sorted({min(8, cores), min(32, cores), cores}):
        torch.set_num_threads(nt)
        step()
        t0 = time.perf_counter()
        step(); step()
        sweep[nt] = batch * 2 / (time.perf_counter() - t0)
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = time.perf_counter() - t0
    torch.set_num_threads(cores)
    return dict(value=batch * steps / dt, unit="images/sec", cores=best, kind="port", threads_tried={str(k): round(v, 1) for k, v in sweep.items()},
                sample=f"{steps} steps of batch {batch} (LwF task-0 step, {arch}, fp32 torch-CPU oracle) on {best} threads -- the best of {sorted(sweep)} on a host with "
                       f"{cores} physical cores, each tried with one warm-up + two steps")


def _cpu_baseline_vit(workload, steps, batch, cores):
    """ViT-B/16 workloads: the oracle's L2P / InfLoRA_OPT step (oracle/vit.py, restating l2p.py:46-122 / InfLoRA_opt.py:141-369 and
    backbone/transformer.py) at the full geometry, random weights, fp32"""
    import numpy as np
    from oracle import vit as ov
    cfg, D = ov.VIT_B16, 768
    g = torch.Generator().manual_seed(0)
    l2p = workload.startswith("l2p")
    steps = steps or (2 if l2p else 1)

    def rnd(shape, scale):
        return (torch.rand(shape, generator=g) * 2 - 1) * scale
    P = {n: rnd(shp, 0.02 if len(shp) > 1 else 0.01) for n, shp in ov.param_shapes(cfg, 0 if l2p else 10)}
    for n in P:
        if n.endswith("ln_1.weight") or n.endswith("ln_2.weight") or n.endswith("norm.weight"):
            P[n] = torch.ones_like(P[n])
    x = torch.rand(batch, 3, 224, 224, generator=g) * 2 - 1
    if l2p:
        P["prompt.prompt"], P["prompt.prompt_key"] = torch.rand(1, 10, 5, D, generator=g), torch.rand(10, D, generator=g)
        P["classifier.weight"], P["classifier.bias"] = rnd((100, D), 0.03), torch.zeros(100)
        m = ov.L2P(P, cfg, 10, 10, 100, 5, 1.0)
        m.before_task(0); m.after_task(0); m.before_task(1)
        opt = ov.Adam(m.parameters(), 0.001875)
        y = torch.randint(10, 20, (batch,), generator=g)

        def step():
            m.observe(x, y)
            opt.step()
    else:
        from oracle.methods import SGD
        for t in range(2):
            P[f"classifier_pool.{t}.weight"], P[f"classifier_pool.{t}.bias"] = rnd((20, D), 0.03), torch.zeros(20)
        m = ov.InfLoRA(P, cfg, 20, 20, 10, 1.0, 0.95, 10)
        m.cur_task, m.known, m.apply_lora = 1, 20, True
        m.trainable = [n for n in P if "lora_B" in n or n.startswith("classifier_pool.1.")]
        for n, prm in P.items():
            prm.requires_grad_(n in m.trainable)
        opt = SGD(m.parameters(), 8e-3, 0.9, 0.0)
        y = torch.randint(20, 40, (batch,), generator=g)

        def step():
            _, _, loss = m.observe(x, y)
            opt.zero_grad(); loss.backward(); opt.step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = time.perf_counter() - t0
    return dict(value=batch * steps / dt, unit="images/sec", cores=cores, kind="port",
                sample=f"{steps} step(s) of batch {batch} ({'L2P' if l2p else 'InfLoRA_OPT'} task-1 step, ViT-B/16 224x224, fp32 torch-CPU oracle on {cores} physical cores), no warm-up")


def _time_launches(run, reps, warm=5):
    """average duration of `run` (which launches on torch's current stream) with HIP events on that stream"""
    for _ in range(warm):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def _src_hash():
    """fingerprint of the kernel sources: a committed in-step profile is stale as soon as a kernel changes (the GPU box has no .git, so not a commit id)"""
    import glob
    import hashlib
    h = hashlib.sha1()
    for fn in sorted(glob.glob(os.path.join(ROOT, "libcontinual_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "libcontinual_amd", "csrc", "*.h"))):
        with open(fn, "rb") as f:
            h.update(os.path.basename(fn).encode() + b"\0" + f.read())
    return h.hexdigest()[:12]


_PROFILE_ROUNDS = ("r06", "r05", "r04", "r03", "r02")


def _profile_family_share(workload, prefixes):
    """share of the in-step kernel time of EVERY symbol of a family (all shapes), from the same committed table"""
    for r in _PROFILE_ROUNDS:
        pj = os.path.join(ROOT, "profiles", f"{r}_bench_kernel_stats.json")
        if os.path.exists(pj):
            with open(pj) as f:
                ent = json.load(f).get(workload, {}).get("kernels", {})
            return sum(v["pct"] for k, v in ent.items() if any(k.startswith(px) for px in prefixes)) / 100.0 if ent else None
    return None


def _profile_meta(workload):
    for r in _PROFILE_ROUNDS:
        pj = os.path.join(ROOT, "profiles", f"{r}_bench_kernel_stats.json")
        if os.path.exists(pj):
            with open(pj) as f:
                ent = json.load(f).get(workload, {})
            return dict(file=f"profiles/{r}_bench_kernel_stats.json", src_hash=ent.get("src_hash"), head=ent.get("head"))
    return None


def _profile_lookup(workload, symbol):
    """in-step average duration of a kernel symbol from the committed rocprofv3 --kernel-trace --stats summary of THIS command
    (profiles/r02_bench_kernel_stats.json, written by tools/bench_profile.sh); None if absent"""
    for name in tuple(f"{r}_bench_kernel_stats.json" for r in _PROFILE_ROUNDS):
        pj = os.path.join(ROOT, "profiles", name)
        if os.path.exists(pj):
            break
    else:
        return None
    with open(pj) as f:
        prof = json.load(f)
    ent = prof.get(workload, {}).get("kernels", {})
    hits = [v for k, v in ent.items() if k.startswith(symbol)]
    if not hits:
        return None
    calls = sum(h["calls"] for h in hits)
    return dict(avg_us=sum(h["avg_us"] * h["calls"] for h in hits) / calls, share_of_kernel_time=sum(h["pct"] for h in hits) / 100.0, calls=calls,
                source=f"profiles/{name} (rocprofv3 --kernel-trace --stats of bench.py)")


def _pmc_lookup(key):
    for name in tuple(f"{r}_roofline_pmc.json" for r in _PROFILE_ROUNDS):
        pj = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(pj):
            continue
        with open(pj) as f:
            m = json.load(f).get(key)
        if m:
            return m["traffic_bytes_per_launch"], f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH doubled per the gfx950 note)"
    return None, None


def conv_rooflines(dev, dtype, B, workload):
    """The convolution families of the ResNet-18 step, each timed live (HIP events on the launch stream, the kernel alone, layer shapes of
    the step) and priced against the roofline that bounds it; the first entry is the symbol with the largest share of the step's kernel
    time in the committed in-step trace -- since round 2 the 3x3 stride-1 weight gradient (its in-step figure averages the four layer shapes that share the symbol family).  Algorithmic FLOPs = 2 * M * 9 * Cin * Cout;
    algorithmic bytes = the two tensors read once + the result written once."""
    from libcontinual_amd import _lib
    code = _lib.BF16 if dtype == "bf16" else _lib.F32
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    es = 2 if dtype == "bf16" else 4
    st = torch.cuda.current_stream().cuda_stream
    L = _lib.lib()
    out = []

    def entry(kind, symbol, label, N, H, W, C, K, run, flops, alg_bytes, pmc_key, full_chip_run=None):
        ms = _time_launches(run, 50)
        ach = flops / (ms * 1e-3) / 1e12
        t_mfma, t_hbm = flops / (PEAK_BF16_TFLOPS * 1e12), alg_bytes / (PEAK_HBM_GBS * 1e9)
        bound = "mfma" if t_mfma >= t_hbm else "hbm"
        traffic, src = _pmc_lookup(pmc_key)
        e = dict(bound=bound, kernel=label, pmc_key=pmc_key, launch_ms=ms, algorithmic_flops_per_launch=flops, algorithmic_bytes_per_launch=alg_bytes,
                 traffic=traffic, traffic_source=src)
        if bound == "mfma":
            e.update(achieved=ach, peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_BF16_TFLOPS, hbm_gbs_algorithmic=alg_bytes / (ms * 1e-3) / 1e9)
        else:
            gbs = alg_bytes / (ms * 1e-3) / 1e9
            e.update(achieved=gbs, peak=PEAK_HBM_GBS, unit="GB/s", frac=gbs / PEAK_HBM_GBS, tflops_algorithmic=ach)
        if kind == "wgrad":
            # the in-step grid is 128 workgroups on purpose (half the chip: the launch runs beside the critical chain); the same kernel with
            # one workgroup per CU, for the kernel-quality reading of the fraction
            L.clhip_wgrad4_config(256)
            try:
                ms_full = _time_launches(full_chip_run(), 50)
            finally:
                L.clhip_wgrad4_config(0)
            e.update(grid_workgroups_in_step=160, full_chip=dict(grid_workgroups=256, launch_ms=ms_full, frac=e["frac"] * ms / ms_full))
        ins = [_profile_lookup(workload, sy) for sy in (symbol if isinstance(symbol, (list, tuple)) else [symbol])]
        if all(i is not None for i in ins):      # the same symbol(s) inside the step (two streams share the chip): the conservative figure
            t_in = sum(i["avg_us"] for i in ins) * 1e-3
            # a helper symbol shared with other kernels (the partial-block reduce runs once behind EVERY weight-gradient launch of the step)
            # counts with the launches it has behind THIS entry's first symbol only
            share = ins[0]["share_of_kernel_time"] + sum(i["share_of_kernel_time"] * min(1.0, ins[0]["calls"] / i["calls"]) for i in ins[1:])
            # `frac` / `achieved` = the IN-STEP figure (what follows from profiles/: the launch shares the chip with the other stream), the
            # live stand-alone measurement of this run is kept beside it
            e.update(standalone_frac=e["frac"], standalone_achieved=e["achieved"], standalone_launch_ms=ms)
            e.update(in_step_launch_ms=t_in, frac=e["standalone_frac"] * ms / t_in, achieved=e["standalone_achieved"] * ms / t_in, in_step_frac=e["standalone_frac"] * ms / t_in,
                     in_step_share_of_kernel_time=share, in_step_source=ins[0]["source"])
            if kind == "wgrad":      # every shape of the symbol family, not just this entry's (the family is the story, 34 % of the step)
                e["family_share"] = _profile_family_share(workload, ("conv_wgrad4_kernel", "wgrad3_reduce_kernel", "wgrad_multi_reduce_kernel"))
        out.append(e)

    def wgrad_full(x, dz, dw, N, H, W, C, K, stride):
        """-> the launch closure under the CURRENT clhip_wgrad4_config (its scratch is sized for it)"""
        nb = L.clhip_conv_wgrad_ws_bytes(N, H, W, C, C, K, 3, stride, 1, code)
        buf = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
        return lambda: _lib.call("clhip_conv_wgrad", x.data_ptr(), dz.data_ptr(), dw.data_ptr(), buf.data_ptr(), N, H, W, C, C, K, 3, stride, 1, code, st)

    r18 = "resnet18" in workload
    if not r18 and dtype == "bf16":
        # CifarResNet-32 / resnet32_V2 steps: the symbol with the largest share of the committed in-step traces is the fused backward of a
        # 16 -> 16-channel layer (bwd16_fused_kernel<true>: BatchNorm backward on the operand loads + dgrad + weight gradient, one launch,
        # conv3.hip) -- timed here through its C entry point at the stage-1 shape.  Algorithmic bytes: x, dy, z read once, dx written once
        # (+ the 9 KB of dW); FLOPs: dgrad + wgrad = 2 x 2 M 9 C K.  HBM-bound (AI = 18 FLOP/B).
        import ctypes as Ct

        class BnGrad(Ct.Structure):
            _fields_ = [("dy", Ct.c_void_p), ("z", Ct.c_void_p), ("sums", Ct.c_void_p), ("replicas", Ct.c_int), ("mean", Ct.c_void_p), ("invstd", Ct.c_void_p),
                        ("gamma", Ct.c_void_p), ("beta", Ct.c_void_p), ("dgamma", Ct.c_void_p), ("dbeta", Ct.c_void_p), ("relu_mask", Ct.c_void_p), ("dres", Ct.c_void_p),
                        ("dres_accumulate", Ct.c_int)]
        N, H, W, C = B, 32, 32, 16
        M = N * H * W
        if L.clhip_conv_bn_input_supported(N, H, W, C, C, 3, 1, 1, code) and L.clhip_conv_dgrad_wgrad_supported(N, H, W, C, C, C, 3, 1, 1, code):
            xin = torch.randn(N, H, W, C, device=dev).to(tdt)
            dy = (torch.randn(N, H, W, C, device=dev) * 0.5).to(tdt)
            zp = torch.randn(N, H, W, C, device=dev).to(tdt)
            wd = (torch.randn(C, 9, C, device=dev) * 0.1).to(tdt)
            dx = torch.empty(N, H, W, C, device=dev, dtype=tdt)
            dw = torch.zeros(C, 9, C, device=dev)
            ws_ = torch.empty(max(L.clhip_conv_wgrad_ws_bytes(N, H, W, C, C, C, 3, 1, 1, code), 16), dtype=torch.uint8, device=dev)
            mean, invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
            gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            dgam, dbet = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
            zf = zp.float().reshape(-1, C)
            g = (zf > 0).float() * dy.float().reshape(-1, C)
            gsum = torch.zeros(4, 2, C, dtype=torch.float64, device=dev)
            gsum[0, 0], gsum[0, 1] = g.double().sum(0), (g * zf).double().sum(0)
            bg = BnGrad(dy.data_ptr(), zp.data_ptr(), gsum.data_ptr(), 4, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dgam.data_ptr(), dbet.data_ptr(),
                        None, None, 0)
            keep = (xin, dy, zp, wd, dx, dw, ws_, mean, invstd, gamma, beta, dgam, dbet, gsum, bg)
            entry("bwd", "bwd16_fused_kernel<true>", f"bwd16_fused_kernel<true>: BatchNorm backward + dX + dW of 3x3/s1 16->16 in one launch @ [{N},{H},{W},{C}]", N, H, W, C, C,
                  lambda: _lib.call("clhip_conv_dgrad_wgrad_bn_grad", xin.data_ptr(), None, Ct.byref(bg), wd.data_ptr(), dx.data_ptr(), 0, dw.data_ptr(), ws_.data_ptr(), None, None,
                                    None, None, None, 1, N, H, W, C, C, C, 3, 1, 1, code, st),
                  2 * 2.0 * M * 9 * C * C, 4 * M * C * es + C * 9 * C * 4, f"bwd16/{N}x{H}x{W}x{C}")
            del keep
    # the 64 -> 64-channel forward runs on conv8.hip (round 5: two four-wave workgroups per CU) from 1024 tiles of 128 pixels up (batch >= 128 at 32 x 32)
    l1_sym = "conv8_kernel<0, 0, 0>" if B * 32 * 32 >= 1024 * 128 else "conv4_kernel<4, 1, 1, 32, 32, 0>"
    shapes = ((32, 64, l1_sym), (16, 128, "conv4_kernel<4, 2, 1, 64, 32, 0>")) if r18 else ((8, 64, "conv4_kernel<2, 1, 2, 64, 8, 0>"),)
    for i, (H, C, sym) in enumerate(shapes):
        N, W, K = B, H, C
        M = N * H * W
        x = torch.randn(N, H, W, C, device=dev).to(tdt)
        dz = torch.randn(N, H, W, K, device=dev).to(tdt)
        w = (torch.randn(K, 9, C, device=dev) * 0.05).to(tdt)
        z = torch.empty(N, H, W, K, device=dev, dtype=tdt)
        dw = torch.zeros(K, 9, C, device=dev)
        wsb = L.clhip_conv_wgrad_ws_bytes(N, H, W, C, C, K, 3, 1, 1, code)
        wsbuf = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        acc = torch.zeros(64, 2, K, dtype=torch.float64, device=dev)
        flops = 2.0 * M * 9 * C * K
        if i == 0 and r18:         # (the CifarResNet-32 steps contain no wgrad4 launch: their weight gradients are part of the fused backward kernels)
            _lib.call("clhip_conv_wgrad", x.data_ptr(), dz.data_ptr(), dw.data_ptr(), wsbuf.data_ptr(), N, H, W, C, C, K, 3, 1, 1, code, st)
            entry("wgrad", [f"conv_wgrad4_kernel<{W}, 1, 3>", "wgrad3_reduce_kernel"], f"conv_wgrad4_kernel<{W},1,3> + wgrad3_reduce_kernel: dW of 3x3/s1 @ [{N},{H},{W},{C}] x [{N},{H},{W},{K}]",
                  N, H, W, C, K,
                  lambda: _lib.call("clhip_conv_wgrad", x.data_ptr(), dz.data_ptr(), dw.data_ptr(), wsbuf.data_ptr(), N, H, W, C, C, K, 3, 1, 1, code, st),
                  flops, M * (C + K) * es + K * 9 * C * 4, f"wgrad/{N}x{H}x{W}x{C}x{K}", full_chip_run=lambda: wgrad_full(x, dz, dw, N, H, W, C, K, 1))
        entry("fwd", sym, f"{sym.replace(' ', '')} forward 3x3/s1 + BN-stat epilogue @ [{N},{H},{W},{C}] x [{K},3,3,{C}]", N, H, W, C, K,
              lambda: _lib.call("clhip_conv_fwd_acc", x.data_ptr(), w.data_ptr(), z.data_ptr(), acc.data_ptr(), 8, N, H, W, C, K, 3, 1, 1, code, st),
              flops, M * (C + K) * es + w.numel() * es, f"fwd/{N}x{H}x{W}x{C}x{K}")
    if r18:
        # the stride-2 layer entries (3x3 / stride 2), the largest of them
        N, H, W, C, K = B, 32, 32, 64, 128
        x = torch.randn(N, H, W, C, device=dev).to(tdt)
        dz = torch.randn(N, H // 2, W // 2, K, device=dev).to(tdt)
        dw = torch.zeros(K, 9, C, device=dev)
        M2 = N * (H // 2) * (W // 2)
        wsb = L.clhip_conv_wgrad_ws_bytes(N, H, W, C, C, K, 3, 2, 1, code)
        wsbuf = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        entry("wgrad", [f"conv_wgrad4_kernel<{W}, 2, 3>", "wgrad3_reduce_kernel"], f"conv_wgrad4_kernel<{W},2,3> + wgrad3_reduce_kernel: dW of 3x3/s2 @ [{N},{H},{W},{C}] x [{N},{H // 2},{W // 2},{K}]",
              N, H, W, C, K,
              lambda: _lib.call("clhip_conv_wgrad", x.data_ptr(), dz.data_ptr(), dw.data_ptr(), wsbuf.data_ptr(), N, H, W, C, C, K, 3, 2, 1, code, st),
              2.0 * M2 * 9 * C * K, (N * H * W * C + M2 * K) * es + K * 9 * C * 4, f"wgrad_s2/{N}x{H}x{W}x{C}x{K}",
              full_chip_run=lambda: wgrad_full(x, dz, dw, N, H, W, C, K, 2))
        # the input gradient of that block entry: dgrad of the 3x3 / s2 convolution + dgrad of the 1x1 / s2 shortcut in one launch (conv6.hip)
        if dtype == "bf16" and L.clhip_conv_dgrad_pair_supported(N, H, W, C, K, code):
            dzs = torch.randn(N, H // 2, W // 2, K, device=dev).to(tdt)
            w3 = (torch.randn(C, 9, K, device=dev) * 0.05).to(tdt)
            w1 = (torch.randn(C, 1, K, device=dev) * 0.05).to(tdt)
            dx = torch.empty(N, H, W, C, device=dev, dtype=tdt)
            pk = torch.empty(L.clhip_conv_dgrad_pair_packed_bytes(C, K), dtype=torch.uint8, device=dev)
            _lib.call("clhip_conv_dgrad_pair_pack", w3.data_ptr(), w1.data_ptr(), pk.data_ptr(), C, K, code, st)
            entry("dgrad", "dgrad6_kernel<2, true>", f"dgrad6_kernel<2,true> dX of 3x3/s2 + 1x1/s2 shortcut in one launch @ [{N},{H // 2},{W // 2},{K}] x2 -> [{N},{H},{W},{C}]",
                  N, H, W, C, K,
                  lambda: _lib.call("clhip_conv_dgrad_pair", dz.data_ptr(), pk.data_ptr(), dzs.data_ptr(), dx.data_ptr(), 0, N, H, W, C, K, code, st),
                  2.0 * M2 * 10 * C * K, (2 * M2 * K + N * H * W * C) * es + pk.numel(), f"dgrad_pair/{N}x{H}x{W}x{C}x{K}")
    # largest share of the step's kernel time first (committed in-step trace); without a trace, the order above
    out.sort(key=lambda e: -e.get("in_step_share_of_kernel_time", 0.0))
    return out


def gemm_roofline(dev, dtype, M, workload=None, default_batch=False):
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
    pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    pj = next((os.path.join(pdir, f) for f in ("r05_roofline_pmc_gemm.json", "r01_roofline_pmc_gemm.json") if os.path.exists(os.path.join(pdir, f))), "")
    if pj:
        with open(pj) as f:
            for m in json.load(f):
                if m.get("shape") == [M, N, K] and m.get("dtype") == dtype:
                    traffic, src = m["traffic_bytes_per_launch"], f"profiles/{os.path.basename(pj)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH doubled per the gfx950 note)"
    # which kernel clhip_gemm_nt picks for this shape: gemm8.hip (round 5) takes the row panels that fill whole rounds of its 256 workgroups (all rows when
    # the last round is >= 90 % full), the register-staged gemm_nt_kernel the rest -- and everything for fp32, for fewer than 256 tiles, or with CLHIP_GEMM8=0
    tiles = ((M + 255) // 256) * (N // 256)
    g8 = dtype == "bf16" and os.environ.get("CLHIP_GEMM8", "1") != "0" and tiles >= 256
    sym = "gemm8_kernel<bias+GELU> (+ gemm_nt_kernel for the rows behind the last whole round)" if g8 else f"gemm_nt_kernel<{dtype}, bias+GELU>"
    e = dict(bound="mfma", kernel=f"{sym} fc1 @ [{M},768]x[3072,768]^T", achieved=ach, peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
             frac=ach / PEAK_BF16_TFLOPS, traffic=traffic, traffic_source=src, launch_ms=ms, algorithmic_flops_per_launch=flops, algorithmic_bytes_per_launch=alg_bytes,
             hbm_gbs_algorithmic=alg_bytes / (ms * 1e-3) / 1e9)
    # `frac` / `achieved` = the IN-STEP figure when the committed kernel table of this very command holds the symbol (the profiles were taken at the default batch):
    # algorithmic FLOPs / the launch's average duration inside the training step; the live stand-alone measurement stays beside it
    if workload is not None and default_batch:
        ins = _profile_lookup(workload, "gemm8_kernel<3>" if g8 else "gemm_nt_kernel<unsigned short, 3,")
        if ins:
            t_in = ins["avg_us"] * 1e-3
            e.update(standalone_frac=e["frac"], standalone_achieved=ach, standalone_launch_ms=ms, in_step_launch_ms=t_in, achieved=flops / (t_in * 1e-3) / 1e12,
                     frac=flops / (t_in * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, in_step_frac=flops / (t_in * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                     in_step_share_of_kernel_time=ins["share_of_kernel_time"], in_step_source=ins["source"])
    return e


def dp_breakdown(model, opt, reducer, batches, method_name, dev, steps=12):
    """Where a data-parallel step spends its time, measured in an instrumented pass OUTSIDE the timed region (events on the compute stream
    around the pieces of trainer.train_steps' loop body, overlap context on): `backward_ms` (loss.backward(), the early tail of the flat
    gradient buffer already handed to the collective inside it), `exchange_exposed_ms` (reducer.reduce(): what is still being waited
    for once the backward is done -- 0 would be a fully hidden exchange), `optimizer_ms`, and the gradient bucket's all-reduce ALONE
    (`allreduce_alone_ms`, `bucket_mb`, `busbw_gbs` = 2 (N-1)/N x bytes / time: the ring's per-link rate) -- so a SCALE record says by
    itself whether a scaling loss is the links, a missing overlap, or the per-rank step."""
    import contextlib
    import torch.distributed as dist
    from libcontinual_amd import ops
    from libcontinual_amd.trainer import _OBSERVE_DOES_BACKWARD, _backward
    if method_name in _OBSERVE_DOES_BACKWARD or getattr(model, "grad_reducer", None) is not None:
        return {}                                                   # the plugin reduces inside observe: no separable pieces
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
    overlap = reducer.overlap(model) if hasattr(reducer, "overlap") else contextlib.nullcontext()
    with ops.deferred_metrics(True), overlap:
        for i in range(steps):
            _, _, loss = model.observe(dict(batches[i % len(batches)]))
            opt.zero_grad()
            ev[i][0].record()
            _backward(loss)
            ev[i][1].record()
            reducer.reduce(model)
            ev[i][2].record()
            opt.step()
            ev[i][3].record()
    torch.cuda.synchronize()
    use = ev[2:]                                                    # the first steps re-warm the collective
    mean = lambda a, b: sum(e[a].elapsed_time(e[b]) for e in use) / len(use)
    out = dict(backward_ms=mean(0, 1), exchange_exposed_ms=mean(1, 2), optimizer_ms=mean(2, 3))
    from libcontinual_amd.parallel import _flat_grad_buckets
    buckets, _ = _flat_grad_buckets(model)
    if buckets:
        b = max(buckets, key=lambda t: t.numel())
        scratch = torch.empty_like(b)
        for _ in range(3):
            dist.all_reduce(scratch)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        dist.barrier()
        e0.record()
        reps = 10
        for _ in range(reps):
            dist.all_reduce(scratch)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        nbytes = b.numel() * 4
        w = dist.get_world_size()
        out.update(allreduce_alone_ms=ms, bucket_mb=nbytes / 1e6, busbw_gbs=2.0 * (w - 1) / w * nbytes / (ms * 1e-3) / 1e9)
    return out


def task_work_main(a):
    """The single-GPU per-task work north_star names (Fisher-diagonal accumulation and buffer herding stay on one GPU): same contract,
    a "step" = one batch of EWC's Fisher pass (core/model/ewc.py:147-205: train-mode forward, CE over all logits, backward, fisher += g^2 * len(y),
    in fp32 whatever the training dtype) or one class of the herding after a task (core/model/buffer/linearherdingbuffer.py:77-161: eval-mode
    features of the class's 500 images, L2-normalised, greedy mean matching for buffer_size // classes = 40 exemplars)."""
    import libcontinual_amd.model as M
    from libcontinual_amd import _lib, ops
    assert a.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1, "the per-task passes are single-GPU by design (SURVEY.md section 8(e))"
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    torch.manual_seed(1993)
    fisher = a.workload == "ewc_fisher_pass"
    bb = M.cifar_resnet32(dtype=a.dtype)
    if fisher:
        B = a.batch or 32                                                  # config/ewc-resnet32-cifar100-b50-5-10.yaml: batch_size 32
        steps, warm = a.steps or 200, a.warmup if a.warmup is not None else 20
        m = M.EWC(bb, 64, 100, device=dev, init_cls_num=50, inc_cls_num=5, lamda=1000).to(dev)
        m.before_task(0, None, None, None)
        batches = [synthetic_batch(B, 0, 50, 100 + i, dev) for i in range(4)]

        class Loader:                                                        # what getFisher reads: batch_size, len(), iteration
            def __init__(self, n): self.n, self.batch_size = n, B
            def __len__(self): return self.n
            def __iter__(self): return (batches[i % 4] for i in range(self.n))
        run = lambda n: m.getFisher(Loader(n))
        per_step_imgs = B
        desc = dict(workload=a.workload, method="EWC.getFisher", backbone="cifar_resnet32", per_gpu_batch=B, global_batch=B, image="3x32x32", parallelism="dp1",
                    pass_images=25000, compute_dtype="f32 (the Fisher pass runs in fp32 whatever the training dtype)")
        metric = "images/sec (EWC Fisher pass after task 0: 25000 images), CIFAR-100 B50-5x10"
    else:
        B = a.batch or 500                                                   # images of one class
        steps, warm = a.steps or 50, a.warmup if a.warmup is not None else 50       # a "step" = one class; the default is one task's 50 classes
        net = M.ICarl(bb, 64, 100, device=dev, init_cls_num=50, inc_cls_num=5, task_num=11).to(dev)
        net.before_task(0, None, None, None)
        net.eval()
        xs = [synthetic_batch(B, 0, 50, 200 + i, dev)["image"] for i in range(4)]

        def run(n):
            # n classes: eval-mode features of their n * B images in the reference's DataLoader(batch_size=256) batches, L2-normalised, then the
            # greedy selection of every class in ONE launch (one block per class: ops.herding_select_classes)
            feats = []
            with torch.no_grad():
                for i in range(n):
                    for j in range(0, B, 256):
                        feats.append(ops.l2_normalize_rows(net.network.backbone(xs[i % 4][j:j + 256])["features"]))
            return ops.herding_select_classes(torch.cat(feats), [B] * n, 2000 // 50)
        per_step_imgs = B
        desc = dict(workload=a.workload, method="LinearHerdingBuffer.herding_select", backbone="cifar_resnet32", per_gpu_batch=B, global_batch=B, image="3x32x32",
                    parallelism="dp1", classes=50, exemplars_per_class=40)
        metric = "images/sec (herding after task 0: 50 classes x 500 images -> 40 exemplars each), CIFAR-100 B50-5x10"
    run(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ips = per_step_imgs * steps / dt
    # roofline of the pass's own elementwise / selection kernel, timed live (the convolution kernels are priced by the training workloads)
    st = torch.cuda.current_stream().cuda_stream
    if fisher:
        flat, gflat = bb.flat_parameters()
        f = torch.zeros_like(flat)
        ms = _time_launches(lambda: ops.fisher_accum(f, gflat, 1.0 / 25000), 50)
        nbytes = 12.0 * flat.numel()
        roof = dict(bound="hbm", kernel=f"fisher_accum_kernel: f += s * g^2 over the flat parameter buffer ({flat.numel()} fp32)", launch_ms=ms, algorithmic_bytes_per_launch=nbytes,
                    achieved=nbytes / (ms * 1e-3) / 1e9, peak=PEAK_HBM_GBS, unit="GB/s", frac=nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, traffic=None,
                    note="5.6 MB per launch: a launch at its latency floor, not a bandwidth-bound one")
        step_flops = 3 * FLOP_FWD_PER_IMG["cifar_resnet32"]
    else:
        cf = ops.l2_normalize_rows(torch.randn(B, 64, device=dev))
        cf = ops.l2_normalize_rows(torch.randn(50 * B, 64, device=dev))
        ms = _time_launches(lambda: ops.herding_select_classes(cf, [B] * 50, 40), 20)
        nbytes = 50.0 * B * 64 * 4                                           # the features read once (each class's rows live in its workgroup's LDS for the 40 picks)
        roof = dict(bound="hbm", kernel=f"herding_batched_kernel: 50 classes x 40 greedy picks over [{B}, 64] normalised features, one workgroup per class", launch_ms=ms,
                    algorithmic_bytes_per_launch=nbytes, achieved=nbytes / (ms * 1e-3) / 1e9, peak=PEAK_HBM_GBS, unit="GB/s", frac=nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                    traffic=None, note="40 sequential picks per class on LDS-resident rows: latency-bound by construction (50 workgroups; 6.4 MB read once)")
        step_flops = FLOP_FWD_PER_IMG["cifar_resnet32"]
    out = {"metric": metric, "value": ips, "unit": "images/sec", "n_gpus": 1, "steps": steps, "warmup": warm, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32" if fisher else a.dtype, "data": "synthetic", "config": desc,
           "whole_pass_seconds": (25000.0 / ips), "step_tflops_algorithmic": step_flops * ips / 1e12,
           "step_frac_of_hbm_peak": (HBM_BYTES_STEP_PER_IMG["cifar_resnet32"] * (2.0 if fisher else 0.4)) * ips / 1e9 / PEAK_HBM_GBS,
           "roofline": roof}
    print(json.dumps(out))


# the other BASELINE workloads, timed live in the same process behind the headline (20 steps each) so that they are in the driver's record
SECONDARY = (("ewc_resnet32_b50_task1", 256), ("icarl_resnet32_b50_task1", 32), ("inflora_vitb16_b20_task1", 128))


def secondary_lines(dtype, dev, steps=20):
    import gc
    from libcontinual_amd import parallel
    from libcontinual_amd.trainer import train_steps
    res = {}
    for workload, batch in SECONDARY:
        try:
            vit = "vitb16" in workload
            torch.manual_seed(1993)
            model, opt, arch, teacher, (lo, hi) = build_method(workload, dtype, dev)
            parallel.attach(model, opt, None)
            model.train()
            batches = [synthetic_batch(batch, lo, hi, 100 + i, dev, 224 if vit else 32) for i in range(4)]
            name = type(model).__name__

            def run(n):
                train_steps(model, opt, (batches[i % 4] for i in range(n)), None, name, None, dev)
            run(5 if vit else 30)                                   # (warm-up: plans, workspaces, the graph capture of the small batches and its probe)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(steps)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[f"{workload} --batch {batch}"] = dict(ms_per_step=dt / steps * 1e3, images_per_sec=batch * steps / dt, steps=steps)
            del model, opt, batches
        except Exception as e:                                      # (never at the price of the headline line)
            res[f"{workload} --batch {batch}"] = dict(error=f"{type(e).__name__}: {e}")
        gc.collect()
        torch.cuda.empty_cache()
    return res


def main():
    a = parse()
    if a.workload in ("ewc_fisher_pass", "herding_b50"):
        return task_work_main(a)
    vit = "vitb16" in a.workload
    if a.batch is None:
        a.batch = 16 if a.workload.startswith("l2p") else (128 if vit else 256)
    if a.steps is None:               # a timed region of about a second (the ResNet steps take 2-3 ms, the ViT steps 5-16 ms)
        a.steps = 80 if vit else 400
    if a.warmup is None:
        a.warmup = 5 if vit else 20
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    global_batch = a.batch * world if a.scaling == "weak" else a.batch
    if a.scaling == "strong":
        assert a.batch % world == 0, f"--scaling strong: the global batch {a.batch} does not divide over {world} ranks"
        a.batch = a.batch // world                         # per-rank batch = batch_size // n_gpu (core/trainer.py:229-241)
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
    more = []
    if vit:
        roofline = gemm_roofline(dev, a.dtype, a.batch * (222 if a.workload.startswith("l2p") else 197), a.workload, a.batch == (16 if a.workload.startswith("l2p") else 128))
    else:
        rl = conv_rooflines(dev, a.dtype, a.batch, a.workload)
        roofline, more = rl[0], rl[1:]
    if a.roofline_only:               # the launches the PMC passes of tools/bench_profile.py sample
        if rank == 0:
            print(json.dumps(dict(roofline=roofline, roofline_more=more)))
        return
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
    dp = None
    if world > 1:            # self-checking record of the process group: one entry per rank, gathered over the collective backend itself
        mine = dict(rank=rank, local_rank=local, device=torch.cuda.get_device_name(local), pci=torch.cuda.get_device_properties(local).pci_bus_id
                    if hasattr(torch.cuda.get_device_properties(local), "pci_bus_id") else None, ms_per_step=dt / a.steps * 1e3)
        mine.update(dp_breakdown(model, opt, reducer, batches, name, dev))
        every = [None] * world
        dist.all_gather_object(every, mine)
        import libcontinual_amd
        gs = getattr(model, "_graphed_step", None)
        dp = dict(backend=dist.get_backend(), world_size=dist.get_world_size(), exchange=reducer.exchange, ranks=every,
                  graph_replay=bool(gs is not None and gs.graphs and not gs.disabled),       # the timed steps replayed backward + all-reduce + optimizer from one HIP graph
                  hw_queue_cap=dict(zip(("state", "GPU_MAX_HW_QUEUES"), libcontinual_amd.hw_queue_cap_state())))
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
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": a.workload, "method": name, "backbone": "vit_base_patch16_224" if vit else arch, "per_gpu_batch": a.batch,
                   "global_batch": global_batch, "image": "3x224x224" if vit else "3x32x32", "parallelism": f"dp{world}", "optimizer": "fused " + type(opt).__name__, "final_loss": loss_avg},
        "step_tflops_algorithmic": step_flops_per_img * ips / 1e12,
        "step_frac_of_bf16_mfma_peak": step_flops_per_img * ips / 1e12 / (PEAK_BF16_TFLOPS * world),
    }
    if not vit:
        # the HBM-roofline reading of the whole step (primary for CifarResNet-32, AI ~ 130 FLOP/B; secondary for ResNet-18)
        hb = HBM_BYTES_STEP_PER_IMG[arch] + (HBM_BYTES_TEACHER_PER_IMG[arch] if teacher else 0.0)
        out["step_hbm_gbs_algorithmic"] = hb * ips / 1e9
        out["step_frac_of_hbm_peak"] = hb * ips / 1e9 / (PEAK_HBM_GBS * world)
        out["step_bound"] = "hbm" if arch == "cifar_resnet32" else "mfma"
    if dp is not None:
        out["dp"] = dp
    meta = _profile_meta(a.workload)
    if meta is not None and isinstance(roofline, dict):
        cur = _src_hash()
        roofline["profile_head"] = meta["head"]
        roofline["profile_src_hash"], roofline["src_hash"] = meta["src_hash"], cur
        roofline["profile_stale"] = meta["src_hash"] != cur
        if roofline["profile_stale"]:
            print(f"bench.py: the in-step figures of `roofline` come from {meta['file']} taken at kernel sources {meta['src_hash']} (commit {meta['head']}); "
                  f"the sources here hash to {cur}: re-run tools/bench_profile.py", file=sys.stderr)
    out["roofline"] = roofline
    if more:
        out["roofline_more"] = more
    if a.workload == "lwf_resnet18_b50_task0" and world == 1 and not a.no_secondary:
        out["secondary"] = secondary_lines(a.dtype, dev)
    if not a.no_cpu_baseline and world == 1:
        # the GPU line's own batch for the ResNet workloads; the ViT steps cost 2-3 CPU-seconds per image, so their bounded sample is a
        # smaller batch (the per-image cost of the CPU path does not depend on it)
        out["cpu_baseline"] = cpu_baseline(a.workload, a.cpu_steps, min(a.batch, 8) if vit else a.batch)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
