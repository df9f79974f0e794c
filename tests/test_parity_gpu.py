"""Network- and method-level parity on a real MI355X.

* backbone plan (forward, backward, running stats) vs the oracle nets on identical weights/inputs
  - f32 mode: fp32 MFMA, the exact-arithmetic parity mode  -> tight
  - bf16 mode: the performance mode                        -> tolerance stated per quantity
* the golden scenarios (tests/golden/*.npz = fp64 runs of the REFERENCE) driven through the product's
  plugin classes with the same adapter that drove the reference.
Tolerances.  Forward quantities and the FIRST optimisation step are held tight (f32 mode: 2e-4).  Later steps drift the
way any fp32 evaluation of these batch-4..8 scenarios drifts from the fp64 reference (tests/test_oracle_golden.py::*_fp32
shows the CPU fp32 oracle 3e-3 .. 5e-3 off at the fourth step of the EWC / WA scenarios): observed on the MI355X in f32
mode -- losses 2e-4 (LwF, DER, LUCIR), 1e-3 (iCaRL), 7e-3 .. 9e-3 (EWC, WA at step 4); final parameters <= 8e-3 of their
abs-sum; bounds below are ~3x the observed values.  The arithmetic itself is pinned kernel by kernel in test_kernels_gpu.py.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import libcontinual_amd.model as M          # noqa: E402
from libcontinual_amd import optim          # noqa: E402
from oracle import fixtures as fx           # noqa: E402
from oracle import nets                     # noqa: E402
from oracle import scenarios as sc          # noqa: E402

DEV = "cuda"


class ProductNS:
    """namespace handed to the shared PluginAdapter: product classes, compute dtype fixed per test"""

    def __init__(self, dtype):
        self.dtype = dtype
        for n in ("EWC", "LWF", "ICarl", "LUCIR", "WA", "DER", "bic", "Finetune", "LinearHerdingBuffer", "LinearSpiltBuffer", "CosineLinear",
                  "SplitCosineLinear"):
            setattr(self, n, getattr(M, n))

    def cifar_resnet32(self, **kw):
        return M.cifar_resnet32(dtype=self.dtype)

    def resnet32_V2(self, **kw):
        return M.resnet32_V2(dtype=self.dtype)

    def cifar_resnet32_V2(self, **kw):
        return M.cifar_resnet32_V2(dtype=self.dtype)

    def resnet18(self, **kw):
        return M.resnet18(dtype=self.dtype, **kw)


def adapter(dtype):
    return sc.PluginAdapter(ProductNS(dtype), DEV, sgd_factory=lambda params, **kw: optim.SGD(params, **kw))


def relnorm(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    v = float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))
    if os.environ.get("CLHIP_PRINT_DEVS"):          # bound-setting runs: every measured deviation with the line that asserts on it
        import inspect
        fr = inspect.stack()[1]
        with open(os.environ["CLHIP_PRINT_DEVS"], "a") as f:
            f.write(f"{fr.function}:{fr.lineno} {v:.3e}\n")
    return v


@pytest.mark.parametrize("arch", ["cifar_resnet32", "resnet18", "resnet32_V2"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_backbone_vs_oracle_random_init(arch, dtype):
    """reference-distribution random init against the fp64 oracle: batch 16 in the f32 mode, batch 64 in bf16 (a
    per-layer relative L2 norm at batch >= 64, not a cosine).

    What a bf16 bound can be.  The parameter gradient of a random-init ReLU network is ill-conditioned: the fp64 ORACLE ITSELF,
    evaluated on conv weights and an input that were merely rounded to bf16, lands 0.25 (ResNet-18) / 0.40 (the two 33-layer
    ResNet-32s) away from its own unrounded gradient in relative L2 norm (0.24 / 0.35 per-layer median, 0.31-0.73 worst layer).
    That operand-rounding yardstick is computed HERE, on the same weights and batch, and the bf16 mode -- which also rounds every
    activation and activation gradient -- is held to a multiple of it: whole gradient <= 1.45 x (1.29-1.34 observed over batch 16 /
    64 / 128 and the three backbones), per-layer median <= 1.5 x (1.27-1.36), worst layer <= 1.7 x the yardstick's worst layer
    (1.26-1.48).  Forward features <= 6e-2 of fp64 (2.0e-2 .. 4.1e-2 observed).  The arithmetic itself is pinned kernel by kernel
    in test_kernels_gpu.py (2^-7 of the output magnitude) and the wiring by the f32 run of this very test (per layer <= 2e-2)."""
    B = 16 if dtype == "f32" else 64
    g = torch.Generator().manual_seed(3)
    P = nets.init_params(arch, g)
    Bf = nets.init_buffers(arch)
    x = torch.randn(B, 3, 32, 32, generator=g)
    cw = torch.randn(B, nets.arch(arch)[1], generator=g)

    def oracle(Pin, xin):
        Pg_ = {k: v.double().requires_grad_(True) for k, v in Pin.items()}
        Bo_ = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in Bf.items()}
        f_ = nets.forward(arch, Pg_, Bo_, xin.double(), True)
        (f_ * cw.double()).sum().backward()
        return f_, Pg_, Bo_
    f_ref, Pg, Bo = oracle(P, x)
    bb = adapter(dtype).backbone(arch, P, Bf)
    bb.train()
    f = bb(x.to(DEV))["features"]
    (f * cw.to(DEV)).sum().backward()
    ftol = 2e-4 if dtype == "f32" else 6e-2
    assert relmax(f.detach().cpu(), f_ref.detach()) < ftol
    rels, a_all, b_all = {}, [], []
    for n, p in bb.named_parameters():
        if n.startswith("fc."):
            assert p.grad is None
            continue
        a, b = p.grad.cpu().double().reshape(-1), Pg[n].grad.reshape(-1)
        rels[n] = relnorm(a, b)
        a_all.append(a); b_all.append(b)
    whole = relnorm(torch.cat(a_all), torch.cat(b_all))
    if dtype == "f32":
        # gradients to ~1e-3: ReLU masks of the handful of activations within 1e-7 of zero flip between two fp32 evaluations
        assert max(rels.values()) < 2e-2 and whole < 1e-2, (max(rels.values()), whole)
    else:
        rb = lambda t: t.to(torch.bfloat16).float()
        _, Py, _ = oracle({k: (rb(v) if v.dim() == 4 else v) for k, v in P.items()}, rb(x))
        yard = {n: relnorm(Py[n].grad.reshape(-1), Pg[n].grad.reshape(-1)) for n in rels}
        yard_whole = relnorm(torch.cat([Py[n].grad.reshape(-1) for n in rels]), torch.cat(b_all))
        med, ymed = float(np.median(list(rels.values()))), float(np.median(list(yard.values())))
        print(f"{arch} bf16 B={B}: whole {whole:.3f} (yardstick {yard_whole:.3f}), per-layer median {med:.3f} ({ymed:.3f}), worst {max(rels.values()):.3f} ({max(yard.values()):.3f})")
        assert whole <= 1.45 * yard_whole, (whole, yard_whole)
        assert med <= 1.5 * ymed, (med, ymed)
        assert max(rels.values()) <= 1.7 * max(yard.values()), (max(rels.values()), max(yard.values()))
    for n, b in bb.named_buffers():
        if "running" in n:
            assert relmax(b.cpu(), Bo[n]) < (1e-4 if dtype == "f32" else 2e-2), n
        else:
            assert int(b) == 1
    bb.eval()
    with torch.no_grad():
        fe = bb(x.to(DEV))["features"]
    fe_ref = nets.forward(arch, Pg, Bo, x.double(), False).detach()
    assert relmax(fe.cpu(), fe_ref) < ftol
    # fmaps contract: list of NCHW fp32 stage outputs
    fm = bb(x.to(DEV))["fmaps"]
    assert [tuple(t.shape[1:]) for t in fm][-1] == (nets.arch(arch)[1], 32 // 2 ** (len(fm) - 1), 32 // 2 ** (len(fm) - 1))


@pytest.mark.parametrize("arch", ["cifar_resnet32", "resnet18"])
def test_bf16_gradient_sits_at_the_storage_format_floor(arch):
    """the bf16 mode's parameter gradient is 1.3 x the operand-rounding yardstick away from the fp64 oracle -- is the extra
    third a kernel defect or the price of the format?  `oracle/bf16_floor_study.py` evaluates the fp64 oracle with every tensor the HIP path
    STORES in bf16 rounded at its storage site (conv outputs z, activations y, their gradients; arithmetic fp64): that format floor is 1.30 x
    the yardstick on CifarResNet-32 (0.528 vs 0.406; the forward tensors z and y account for all of it, rounding dy / dz adds nothing --
    profiles/r04_bf16_floor_study.md).  The product must sit AT that floor: whole-gradient deviation <= 1.06 x the floor's (1.01 observed),
    per-layer median <= 1.1 x, and it must be closer to the format-faithful oracle than to the unrounded one."""
    from oracle import bf16_floor_study as fs
    B = 64
    g = torch.Generator().manual_seed(3)
    P = nets.init_params(arch, g)
    Bf = nets.init_buffers(arch)
    x = torch.randn(B, 3, 32, 32, generator=g)
    cw = torch.randn(B, nets.arch(arch)[1], generator=g).double()

    def oracle(sites, round_operands):
        Pg_ = {k: ((fs.rb(v.double()) if (round_operands and v.dim() == 4) else v.double()).clone().requires_grad_(True)) for k, v in P.items()}
        xin = fs.rb(x.double()) if round_operands else x.double()
        Bo_ = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in Bf.items()}
        f_ = fs.forward(arch, Pg_, Bo_, xin, sites)
        (f_ * cw).sum().backward()
        return {k: v.grad.reshape(-1) for k, v in Pg_.items() if v.grad is not None}
    ref = oracle((), False)
    floor = oracle(("z", "y", "dy", "dz"), True)
    bb = adapter("bf16").backbone(arch, P, Bf)
    bb.train()
    f = bb(x.to(DEV))["features"]
    (f * cw.float().to(DEV)).sum().backward()
    got = {n: p.grad.cpu().double().reshape(-1) for n, p in bb.named_parameters() if p.grad is not None}
    names = sorted(got)
    cat = lambda d: torch.cat([d[k] for k in names])       # noqa: E731
    w_prod, w_floor = relnorm(cat(got), cat(ref)), relnorm(cat(floor), cat(ref))
    to_floor = relnorm(cat(got), cat(floor))
    med_prod = float(np.median([relnorm(got[k], ref[k]) for k in names]))
    med_floor = float(np.median([relnorm(floor[k], ref[k]) for k in names]))
    cos = float(torch.dot(cat(got), cat(floor)) / (cat(got).norm() * cat(floor).norm()))
    print(f"{arch}: product vs fp64 oracle {w_prod:.3f}, format floor vs fp64 oracle {w_floor:.3f} (ratio {w_prod / w_floor:.3f}); per-layer median {med_prod:.3f} / {med_floor:.3f}; "
          f"product vs the format-faithful oracle {to_floor:.3f} (cosine {cos:.4f})")
    assert w_prod <= 1.06 * w_floor, (w_prod, w_floor)
    assert med_prod <= 1.1 * med_floor, (med_prod, med_floor)
    assert to_floor < w_prod, (to_floor, w_prod)


@pytest.mark.parametrize("size", [64, 32])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_preactivation_backbone_vs_oracle_random_init(dtype, size):
    """ResNet_BIC(32) (resnet.py:589-680): BN -> ReLU -> conv blocks, shortcuts on the raw sums, BN-free 1x1 shortcut convs,
    AvgPool2d(8) + flatten.  64 x 64 inputs = the 256 features its `feat_dim` is written for; 32 x 32 = one pooling window."""
    arch = "cifar_resnet32_V2"
    g = torch.Generator().manual_seed(4)
    P, Bf = nets.init_params(arch, g), nets.init_buffers(arch)
    x = torch.randn(8, 3, size, size, generator=g)
    nfeat = 64 * (size // 32) ** 2
    cw = torch.randn(8, nfeat, generator=g)
    Pg = {k: v.double().requires_grad_(True) for k, v in P.items()}
    Bo = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in Bf.items()}
    f_ref = nets.forward(arch, Pg, Bo, x.double(), True)
    (f_ref * cw.double()).sum().backward()
    bb = adapter(dtype).backbone(arch, P, Bf)
    assert bb.feat_dim == 256 and [n for n, _ in bb.named_parameters()].count("layer2.0.downsample.0.weight") == 1
    assert sorted(n for n, _ in bb.named_parameters()) == sorted(P) and sum(p.numel() for p in bb.parameters()) == sum(v.numel() for v in P.values())
    bb.train()
    f = bb(x.to(DEV))
    assert isinstance(f, torch.Tensor) and f.shape == (8, nfeat)
    (f * cw.to(DEV)).sum().backward()
    ftol = 2e-4 if dtype == "f32" else 6e-2
    assert relmax(f.detach().cpu(), f_ref.detach()) < ftol
    rels = {n: relnorm(p.grad.cpu().double().reshape(-1), Pg[n].grad.reshape(-1)) for n, p in bb.named_parameters()}
    names = sorted(rels)
    named = dict(bb.named_parameters())
    whole = relnorm(torch.cat([named[n].grad.cpu().double().reshape(-1) for n in names]), torch.cat([Pg[n].grad.reshape(-1) for n in names]))
    if dtype == "f32":
        assert max(rels.values()) < 2e-2 and whole < 1e-2, (max(rels.values()), whole)
    else:
        # bf16: against the operand-rounding yardstick of the oracle itself (see test_backbone_vs_oracle_random_init)
        rb = lambda t: t.to(torch.bfloat16).float()
        Py = {k: (rb(v) if v.dim() == 4 else v).double().requires_grad_(True) for k, v in P.items()}
        By = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in Bf.items()}
        (nets.forward(arch, Py, By, rb(x).double(), True) * cw.double()).sum().backward()
        yard_whole = relnorm(torch.cat([Py[n].grad.reshape(-1) for n in names]), torch.cat([Pg[n].grad.reshape(-1) for n in names]))
        yard = {n: relnorm(Py[n].grad.reshape(-1), Pg[n].grad.reshape(-1)) for n in names}
        med, ymed = float(np.median(list(rels.values()))), float(np.median(list(yard.values())))
        print(f"{arch} {size}px bf16: whole {whole:.3f} (yardstick {yard_whole:.3f}), per-layer median {med:.3f} ({ymed:.3f}), worst {max(rels.values()):.3f} ({max(yard.values()):.3f})")
        assert whole <= 1.6 * yard_whole, (whole, yard_whole)
        assert med <= 1.6 * ymed, (med, ymed)
        assert max(rels.values()) <= 2.0 * max(yard.values()), (max(rels.values()), max(yard.values()))
    for n, b in bb.named_buffers():
        if "running" in n:
            assert relmax(b.cpu(), Bo[n]) < (1e-4 if dtype == "f32" else 2e-2), n
        else:
            assert int(b) == 1
    bb.eval()
    with torch.no_grad():
        fe = bb(x.to(DEV))
    fe_ref = nets.forward(arch, Pg, Bo, x.double(), False).detach()
    assert relmax(fe.cpu(), fe_ref) < ftol
    # a second training step on the same module (persistent raw-gradient buffers, accumulators) and a deep copy
    import copy
    bb.train()
    for p in bb.parameters():
        p.grad = None
    twin = copy.deepcopy(bb)
    f1 = bb(x.to(DEV)); (f1 * cw.to(DEV)).sum().backward()
    f2 = twin(x.to(DEV)); (f2 * cw.to(DEV)).sum().backward()
    assert torch.equal(f1, f2)
    for (n, p), (_, q) in zip(bb.named_parameters(), twin.named_parameters()):
        assert relnorm(p.grad.cpu(), q.grad.cpu()) < 1e-5, n


def test_preactivation_backward_in_ranges_matches_the_whole_backward():
    """the data-parallel host runs the backward in unit ranges (parallel.GradientReducer): with raw-sum residuals the gradient of a
    block input crosses a range boundary through the persistent raw-gradient buffers"""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(8, 3, 32, 32, generator=g).to(DEV)
    cw = torch.randn(8, 64, generator=g).to(DEV)
    bb = M.cifar_resnet32_V2(dtype="f32").to(DEV)
    bb.train()
    (bb(x) * cw).sum().backward()
    whole = [p.grad.clone() for p in bb.parameters()]
    for cuts in ([17], [30, 22, 11, 3]):                     # inside a block, at shortcut units, several pieces
        for p in bb.parameters():
            p.grad = None
        seen = []
        bb._grad_segment_cuts, bb._grad_segment_hook = cuts, (lambda mod, lo, hi: seen.append((lo, hi)))
        (bb(x) * cw).sum().backward()
        bb._grad_segment_hook, bb._grad_segment_cuts = None, []
        assert len(seen) == len(cuts) + 1 and seen[0][1] == bb._nflat and seen[-1][0] == 0
        assert all(a[0] == b[1] for a, b in zip(seen, seen[1:]))             # contiguous ranges, tail of the flat buffer first
        for p, w in zip(bb.parameters(), whole):
            assert relnorm(p.grad.cpu(), w.cpu()) < 1e-5


def test_backbone_intermediate_activations_f32():
    """layer-by-layer: every unit's pre-BN conv output and post-activation against the oracle"""
    arch = "cifar_resnet32"
    g = torch.Generator().manual_seed(5)
    P, Bf = nets.init_params(arch, g), nets.init_buffers(arch)
    x = torch.randn(8, 3, 32, 32, generator=g)
    Bo = {k: v.clone() for k, v in Bf.items()}
    _, acts = nets.forward(arch, P, Bo, x, True, return_acts=True)
    bb = adapter("f32").backbone(arch, P, Bf)
    bb.train()
    with torch.no_grad():
        bb(x.to(DEV))
    for i, u in enumerate(nets.arch(arch)[0]):
        z = bb.debug_read(i + 1, 1).cpu()
        y = bb.debug_read(i + 1, 0).cpu()
        assert relmax(z, acts[u.dst + "#z"]) < 1e-4, (i, u.conv)
        assert relmax(y, acts[u.dst]) < 1e-4, (i, u.conv)


@pytest.mark.parametrize("arch", ["cifar_resnet32", "resnet18", "resnet32_V2", "cifar_resnet32_V2"])
def test_backbone_golden(golden, arch):
    """fixture = the REFERENCE's own module run in fp64 (oracle/gen_golden.py)"""
    want = golden(f"backbone_{arch}")
    got = sc.scenario_backbone(adapter("f32"), arch)
    assert relmax(got["features_train"], want["features_train"]) < 1e-4
    assert relmax(got["features_eval"], want["features_eval"]) < 1e-4
    assert relnorm(got["grad_stem"], want["grad_stem"]) < 0.1          # fp32 floor of this ill-conditioned case: 2e-2
    assert relmax(got["buf_rows"][:, :3], want["buf_rows"][:, :3]) < 1e-4
    got = sc.scenario_backbone(adapter("bf16"), arch)
    assert relmax(got["features_train"], want["features_train"]) < 5e-2
    assert relmax(got["features_eval"], want["features_eval"]) < 2.5e-2      # r05: 6.4e-3 observed (features_train of the worst architecture sits at 2.9e-2: its bound stays)


def _param_rel(got, want):
    g = dict(zip(got["param_names"], got["param_rows"]))
    w = dict(zip(want["param_names"], want["param_rows"]))
    assert set(map(str, g)) == set(map(str, w))
    return max(abs(g[n][0] - w[n][0]) / max(w[n][1], 1e-300) for n in w)


def test_ewc_golden(golden):
    want = golden("ewc")
    got = sc.scenario_ewc(adapter("f32"))
    assert relmax(got["losses"][:2], want["losses"][:2]) < 2e-4
    assert relmax(got["losses"], want["losses"]) < 2.5e-2           # 7e-3 observed at the 4th step (CPU fp32 oracle: 5e-3)
    np.testing.assert_array_equal(got["preds"][:2], want["preds"][:2])
    assert relnorm(got["fisher_head_w"], want["fisher_head_w"]) < 0.05
    assert _param_rel(got, want) < 2e-2
    assert relmax(got["rm_last"], want["rm_last"]) < 1e-2
    got = sc.scenario_ewc(adapter("bf16"))
    assert relmax(got["losses"][:3], want["losses"][:3]) < 2e-2      # r05: 5.5e-3 observed
    assert relnorm(got["fisher_head_w"], want["fisher_head_w"]) < 0.2


def test_ewc_fisher_all_rows_golden(golden, monkeypatch):
    """Fisher diagonals are a named output of BASELINE's north_star: ALL 101 tensors (conv weights, BatchNorm scales and shifts,
    head) of the product's Fisher pass against the reference's own getFisher (ewc.py:147-205) run in fp64 on the same fixed weights
    (tests/golden/ewc_fisher.npz; no optimisation step in front, so the comparison tests the pass, not a trajectory).  Per tensor:
    L2 error over 64 probed elements and error of the sum.  The product runs the Fisher pass in fp32 in BOTH training dtypes
    (EWC.getFisher -> HipResNet.compute_dtype): 2e-3 per tensor either way (1.3e-3 / 4e-4 observed; the fp32 CPU oracle meets 5e-3,
    tests/test_oracle_golden.py::test_ewc_fisher_all_rows).  With `fisher_dtype: bf16` the pass runs on the bf16 plan: squared
    gradients of bf16 activations at batch 8 are 40 % off per tensor in the median, up to 70 % (stated, measured, bounded at 85 %;
    the sum over all tensors within 10 %)."""
    from test_oracle_golden import fisher_deviation
    want = golden("ewc_fisher")
    for dt in ("f32", "bf16"):
        dev = fisher_deviation(sc.scenario_ewc_fisher(adapter(dt)), want)
        worst = max(dev.items(), key=lambda kv: max(kv[1]))
        # SURVEY section 8(d) wrote 1e-3 before anything was measured.  The reference's OWN arithmetic run in fp32 on the CPU (the oracle
        # restatement, bit-for-bit the reference in fp64) deviates from the fp64 fixture by 2.83e-3 in its worst tensor and 1.28e-3
        # in the median (tests/test_oracle_golden.py::test_ewc_fisher_all_rows, re-measured in round 3): 1e-3 is below what ANY
        # single-precision evaluation of getFisher attains, the product (1.3e-3 / 4e-4 worst tensor) is inside the fp32 floor
        assert max(max(v) for v in dev.values()) < 2e-3, (dt, worst, "CPU fp32 oracle's own figure: worst tensor 2.83e-3, median 1.28e-3")
    orig = M.EWC.__init__

    def init_bf16_fisher(self, *a, **kw):
        orig(self, *a, **kw)
        self.kwargs["fisher_dtype"] = "bf16"
    monkeypatch.setattr(M.EWC, "__init__", init_bf16_fisher)
    got = sc.scenario_ewc_fisher(adapter("bf16"))
    dev = fisher_deviation(got, want)
    worst = max(dev.items(), key=lambda kv: max(kv[1]))
    assert max(max(v) for v in dev.values()) < 0.85, worst
    total = abs(got["fisher_rows"][:, 0].sum() - want["fisher_rows"][:, 0].sum()) / want["fisher_rows"][:, 0].sum()      # order-independent
    assert total < 0.1, total


@pytest.mark.parametrize("name,cfg", [("lwf_resnet18", None),
                                      ("lwf_cifar_resnet32", dict(arch="cifar_resnet32", feat_dim=64, bs=8))])
def test_lwf_golden(golden, name, cfg):
    want = golden(name)
    got = sc.scenario_lwf(adapter("f32"), cfg)
    assert relmax(got["losses"][:2], want["losses"][:2]) < 2e-4     # task-0 step, then first task-1 step (teacher + KD)
    assert relmax(got["losses"], want["losses"]) < 1e-3             # 2e-4 observed
    np.testing.assert_array_equal(got["preds"], want["preds"])
    assert relmax(got["teacher_rm"], want["teacher_rm"]) < 1e-3      # teacher BN drifts in train mode (quirk a10)
    assert relmax(got["logits_eval"], want["logits_eval"]) < 3e-2    # eval-mode logits after the three updates (9e-3 observed)
    assert _param_rel(got, want) < 5e-3
    got = sc.scenario_lwf(adapter("bf16"), cfg)
    assert relmax(got["losses"], want["losses"]) < 1.5e-2      # r05: 4.6e-3 observed
    assert relmax(got["teacher_rm"], want["teacher_rm"]) < 3e-2


def test_lwf_long_trajectory_golden(golden):
    """12 optimisation steps through LwF (6 CE steps, then 6 CE + KD steps against the train-mode teacher) with momentum,
    weight decay, a learning-rate drop inside each task and a fresh optimizer per task, against the fp64 run of the
    reference: the whole trajectory, not just its first steps"""
    want = golden("lwf_long")
    got = sc.scenario_lwf(adapter("f32"), sc.LWF_LONG_CFG)
    d = np.abs(got["losses"] - want["losses"]) / np.abs(want["losses"])
    assert d[:2].max() < 2e-4, d
    assert d.max() < 5e-3, d                                          # 1.1e-3 observed (CPU fp32 oracle: 9e-4)
    assert (got["preds"] == want["preds"]).mean() > 0.95              # 98 % of the 384 decisions (near-ties flip)
    assert relmax(got["teacher_rm"], want["teacher_rm"]) < 5e-3       # 5e-4 observed
    assert _param_rel(got, want) < 3e-2                               # 1e-2 observed (early BN biases; CPU fp32 oracle: 3e-3)
    got = sc.scenario_lwf(adapter("bf16"), sc.LWF_LONG_CFG)
    d = np.abs(got["losses"] - want["losses"]) / np.abs(want["losses"])
    assert d.max() < 5e-2, d
    assert _param_rel(got, want) < 3e-2


def test_icarl_golden(golden, tmp_path):
    want = golden("icarl")
    got = sc.scenario_icarl(adapter("f32"), str(tmp_path))
    assert relmax(got["losses"][:2], want["losses"][:2]) < 5e-4
    assert relmax(got["losses"], want["losses"]) < 5e-3             # 1e-3 observed
    np.testing.assert_array_equal(got["preds"][:2], want["preds"][:2])
    assert _param_rel(got, want) < 2e-2
    # Herding runs on features recomputed after two SGD steps in fp32: the feature drift against the fp64 reference run shows in the
    # class means (9e-4 observed, bound 5e-3).  The selection KERNEL is exact on given features (test_kernels_gpu.py::
    # test_herding_at_benchmark_size, test_ncm_and_herding_match_reference_math); with that drift the picked SETS coincide (24 of 24
    # observed; the first two picks of one class, a near-tie, swap their order) and the NCM decisions differ only where two class means are at the
    # same distance to 1e-7 (3 of 16 test points, margins 6e-8).
    np.testing.assert_array_equal(got["buffer_labels0"], want["buffer_labels0"])
    np.testing.assert_array_equal(got["buffer_labels1"], want["buffer_labels1"])
    assert relmax(got["class_means0"], want["class_means0"]) < 5e-3
    # the slack is bounded by DECISION MARGINS, not by counts.  Herding: the fixture holds, for every class and every
    # greedy pick of the reference's own run, the relative gap between its best and second-best candidate (herding_margin0; smallest
    # gaps of this scenario: 0.7 %, 1.2 %, 1.7 %).  The product's pick sequence of a class may leave the reference's only AT a pick
    # whose gap is below 2 % -- four times the 5e-3 bound on the feature drift that can flip it -- and is identical before it.
    per = len(want["chosen0"]) // 4
    for cls in range(4):
        g, w = list(got["chosen0"][cls * per:(cls + 1) * per]), list(want["chosen0"][cls * per:(cls + 1) * per])
        if g != w:
            k = next(i for i in range(per) if g[i] != w[i])
            assert want["herding_margin0"][cls, k] < 2e-2, (cls, k, g, w, want["herding_margin0"][cls])
    assert len(set(got["chosen0"]) & set(want["chosen0"])) >= 22     # (and the sets still nearly coincide: 24 of 24 observed)
    # NCM: decisions may differ only where the two nearest class means are equidistant to rounding (relative gap < 1e-5 in BOTH runs'
    # own margins; this scenario's untrained features put 3 of 16 test points on such ties, gaps 6e-8)
    differ = got["ncm_pred0"] != want["ncm_pred0"]
    assert (got["ncm_margin0"][differ] < 1e-5).all() and (want["ncm_margin0"][differ] < 1e-5).all(), (got["ncm_margin0"], want["ncm_margin0"], differ)
    got = sc.scenario_icarl(adapter("bf16"), str(tmp_path / "b"))
    assert relmax(got["losses"][:2], want["losses"][:2]) < 2.5e-2      # r05: 6.9e-3 observed
    np.testing.assert_array_equal(got["buffer_labels1"], want["buffer_labels1"])


def test_lucir_golden(golden):
    want = golden("lucir")
    got = sc.scenario_lucir(adapter("f32"))
    assert relmax(got["fc2_imprint_norm"], want["fc2_imprint_norm"]) < 1e-9
    assert relmax(got["losses"][:2], want["losses"][:2]) < 2e-4     # CE on the cosine head; first step with all 3 LUCIR terms
    assert relmax(got["losses"], want["losses"]) < 5e-4             # 3e-5 observed
    np.testing.assert_array_equal(got["preds"], want["preds"])
    assert relmax(got["fc2_w"], want["fc2_w"]) < 2e-3
    assert _param_rel(got, want) < 2e-2
    got = sc.scenario_lucir(adapter("bf16"))
    assert relmax(got["losses"], want["losses"]) < 2.5e-2      # r05: 7.1e-3 observed


def test_wa_golden(golden):
    want = golden("wa")
    got = sc.scenario_wa(adapter("f32"))
    assert relmax(got["losses"][:2], want["losses"][:2]) < 2e-4     # the two CE steps of task 0
    assert relmax(got["losses"], want["losses"]) < 3e-2             # 9e-3 observed at the 4th step (CPU fp32 oracle: 4e-3)
    np.testing.assert_array_equal(got["preds"][:2], want["preds"][:2])
    # the logits head is outside the optimizer (reference quirk) and only the alignment rescales its new rows: exact
    np.testing.assert_allclose(got["head_before"], want["head_before"], rtol=1e-6)
    np.testing.assert_allclose(got["head_after"], want["head_after"], rtol=1e-5)
    assert abs(float(got["gamma"]) - float(want["gamma"])) < 1e-5
    assert str(got["buffer_calls"]) == str(want["buffer_calls"])
    assert relmax(got["teacher_rm"], want["teacher_rm"]) < 1e-2      # second teacher = snapshot after all four updates (3e-3 observed)
    # eval-mode logits of this 4-update net: running statistics from four batch-8 steps make the eval forward amplify the
    # parameter drift (the CPU fp32 oracle is 4.4 % off the fp64 reference here; 4.6 % observed on the GPU)
    assert relmax(got["logits_eval"], want["logits_eval"]) < 0.15
    assert _param_rel(got, want) < 2.5e-2
    got = sc.scenario_wa(adapter("bf16"))
    assert relmax(got["losses"], want["losses"]) < 2e-2      # r05: 6.5e-3 observed
    assert relmax(got["teacher_rm"], want["teacher_rm"]) < 3e-2


def test_der_golden(golden, monkeypatch):
    want = golden("der")
    monkeypatch.setenv("CLHIP_DTYPE", "f32")          # DER builds its own extractors (der.py:31-41): dtype comes from the env
    got = sc.scenario_der(adapter("f32"))
    assert relmax(got["losses"][:2], want["losses"][:2]) < 2e-4     # task-0 step, then the first two-extractor CE + aux CE step
    assert relmax(got["losses"], want["losses"]) < 2e-4             # 7e-6 observed
    np.testing.assert_array_equal(got["preds"], want["preds"])
    assert bool(got["frozen_same"])                                  # frozen extractor: parameters bit-identical ...
    assert relmax(got["rm_frozen"], want["rm_frozen"]) < 1e-3        # ... but its BN running stats move (train mode quirk)
    assert relmax(got["rm_new"], want["rm_new"]) < 1e-3
    assert relmax(got["logits_eval"], want["logits_eval"]) < 5e-3     # eval-mode logits after the three updates (9e-4 observed)
    assert _param_rel(got, want) < 2e-3
    monkeypatch.setenv("CLHIP_DTYPE", "bf16")
    got = sc.scenario_der(adapter("bf16"))
    assert relmax(got["losses"], want["losses"]) < 6e-3      # r05: 1.6e-3 observed
    assert bool(got["frozen_same"])
    assert relmax(got["rm_frozen"], want["rm_frozen"]) < 3e-2


def test_bic_golden(golden):
    """fixture = the reference's `bic` on its own ResNet_BIC backbone in fp64 (64 x 64 inputs): CE stage, distillation stage against
    the bias-corrected previous model, the Adam-trained bias layer of stage 2, inference, and the host-side split"""
    want = golden("bic")
    got = sc.scenario_bic(adapter("f32"))
    assert relmax(got["losses"], want["losses"]) < 2e-4
    assert relmax(got["losses2"], want["losses2"]) < 2e-4
    np.testing.assert_array_equal(got["preds"], want["preds"])
    np.testing.assert_array_equal(got["infer"], want["infer"])
    assert np.abs(got["bias"] - want["bias"]).max() < 2e-5                       # alpha, beta after three Adam steps of 1e-3
    assert np.array_equal(got["bias"][0], [1.0, 0.0]) and np.array_equal(got["bias"][2], [1.0, 0.0])
    assert relmax(got["teacher_rm"], want["teacher_rm"]) < 1e-3                   # previous model's BatchNorm ran on batch statistics
    assert _param_rel(got, want) < 2e-3
    for k in want:
        if k.startswith("split"):
            np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    got = sc.scenario_bic(adapter("bf16"))
    assert relmax(got["losses"], want["losses"]) < 1e-3      # r05: 1.3e-4 observed
    assert relmax(got["losses2"], want["losses2"]) < 8e-3      # r05: 2.1e-3 observed
    assert np.abs(got["bias"] - want["bias"]).max() < 2e-3


@pytest.mark.parametrize("order", ["zero_forward_backward", "forward_zero_backward"])
def test_gradient_buffer_restarts_after_zero_grad(order):
    """the flat gradient buffer holds THIS step's gradient whichever side of the forward `optimizer.zero_grad()` runs on
    (the reference trainer calls it between observe and backward, core/trainer.py:602-604), and accumulates when nobody
    zeroes (torch semantics)"""
    torch.manual_seed(0)
    bb = M.cifar_resnet32(dtype="f32").to(DEV)
    bb.train()
    opt = optim.SGD(bb.parameters(), lr=0.0)
    xs = [torch.randn(8, 3, 32, 32, device=DEV) for _ in range(3)]
    cw = torch.randn(8, 64, device=DEV)
    single = []
    for x in xs:                                   # each gradient on its own, from a clean buffer
        for p in bb.parameters():
            p.grad = None
        (bb(x)["features"] * cw).sum().backward()
        single.append(bb._gflat.clone())
    for p in bb.parameters():
        p.grad = None
    for x, want in zip(xs, single):
        if order == "zero_forward_backward":
            opt.zero_grad()
            f = bb(x)["features"]
        else:
            f = bb(x)["features"]
            opt.zero_grad()
        (f * cw).sum().backward()
        opt.step()
        assert relnorm(bb._gflat.cpu(), want.cpu()) < 1e-3      # fp32-atomic summation order only
    (bb(xs[0])["features"] * cw).sum().backward()               # no zero_grad: accumulates onto the last gradient
    assert relnorm(bb._gflat.cpu(), (single[2] + single[0]).cpu()) < 1e-3


def test_full_size_properties_bf16():
    """BASELINE size (ResNet-18, batch 256, bf16): size-independent properties instead of a CPU oracle run.
    (1) backward is linear in dfeat; (2) train-mode BN output statistics: the normalised stem pre-activation
    has per-channel mean 0 / variance 1 (gamma=1, beta=0 at init); (3) an SGD step with lr=0 and no weight
    decay leaves the parameters bit-identical; (4) eval forward is deterministic (bit-equal on repeat)."""
    torch.manual_seed(0)
    bb = M.resnet18(args={"dataset": "cifar100"}, dtype="bf16").to(DEV)
    bb.train()
    x = torch.randn(256, 3, 32, 32, device=DEV)
    cw = torch.randn(256, 512, device=DEV)
    f = bb(x)["features"]
    (f * cw).sum().backward()
    g1 = bb._gflat.clone()
    y1 = bb.debug_read(1, 0)                      # relu(bn(conv(x))) of the stem
    z1 = bb.debug_read(1, 1)
    for p in bb.parameters():
        p.grad = None
    f2 = bb(x)["features"]
    (f2 * (2 * cw)).sum().backward()
    g2 = bb._gflat.clone()
    assert torch.isfinite(g1).all() and torch.isfinite(f).all()
    assert relnorm(g2.cpu(), (2 * g1).cpu()) < 2e-2       # exact up to bf16 rounding of the scaled dy
    mu, var = z1.mean(dim=(0, 2, 3)), z1.var(dim=(0, 2, 3), unbiased=False)
    yn = (z1 - mu.view(1, -1, 1, 1)) / (var.view(1, -1, 1, 1) + 1e-5).sqrt()
    assert relmax(torch.relu(yn).cpu(), y1.cpu()) < 2e-2
    opt = optim.SGD(bb.parameters(), lr=0.0, momentum=0.9)
    before = bb._flat.clone()
    opt.step()
    assert torch.equal(before, bb._flat)
    bb.eval()
    with torch.no_grad():
        e1 = bb(x)["features"].clone()
        e2 = bb(x)["features"].clone()
    assert torch.equal(e1, e2)


@pytest.mark.parametrize("arch,batch", [("resnet18", 64), ("cifar_resnet32", 64)])
def test_fused_batchnorm_backward_reduction_at_network_level(arch, batch, monkeypatch):
    """The plan with the BatchNorm-backward reductions fused into the dgrad epilogues (13 of ResNet-18's 20 units) against the same plan
    without it (CLHIP_BN_FUSE=0: separate reduce passes everywhere; the default fuses only activations of at most 16384 pixels): same weights, same batch, bf16.  Every parameter gradient agrees to the rounding
    of the sums' inputs (fp32 results in the epilogue vs the bf16-rounded tensor in the separate pass)."""
    grads = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("CLHIP_BN_FUSE", fuse)
        torch.manual_seed(5)
        bb = (M.resnet18(args={"dataset": "cifar100"}, dtype="bf16") if arch == "resnet18" else M.cifar_resnet32(dtype="bf16")).to(DEV)
        bb.train()
        g = torch.Generator().manual_seed(9)
        x = torch.randn(batch, 3, 32, 32, generator=g).to(DEV)
        cw = torch.randn(batch, 512 if arch == "resnet18" else 64, generator=g).to(DEV)
        (bb(x)["features"] * cw).sum().backward()
        torch.cuda.synchronize()
        grads[fuse] = {k: p.grad.detach().float().cpu().clone() for k, p in bb.named_parameters() if p.grad is not None}
    devs = sorted(((relnorm(grads["1"][k], grads["0"][k]), k) for k in grads["0"]), reverse=True)
    print("largest fused-vs-separate gradient deviations:", devs[:6])
    worst = devs[0]
    assert worst[0] < 3e-2, devs[:6]
    flat1 = torch.cat([v.reshape(-1) for v in grads["1"].values()])
    flat0 = torch.cat([v.reshape(-1) for v in grads["0"].values()])
    assert relnorm(flat1, flat0) < 2e-2          # 0.8-1.2 % observed: the two arrangements round different intermediate values to bf16


def test_fused_stride2_input_gradient_inside_the_plan():
    """ResNet-18 plans run the input gradient of every down-sampling block entry as ONE launch (conv6.hip: the 3x3/s2 dgrad and the shortcut's
    1x1/s2 dgrad summed in fp32 accumulators, packed weights written by the plan's weight preparation).  Against the same plan with the two
    separate launches (CONV6_PAIR=0: the shortcut's gradient is rounded to bf16 before the 3x3 layer's is added): the forward is untouched
    (bit-identical features), the parameter gradients agree to the rounding of that one intermediate per block entry -- far inside the
    operand-rounding yardstick of test_bf16_gradients_against_f32_mode_at_batch_256 -- and both agree with the f32 parity mode equally well."""
    from libcontinual_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(33)
    x = torch.randn(64, 3, 32, 32, generator=g).to(DEV)
    cw = (torch.randn(64, 512, generator=g) / 16).to(DEV)
    out = {}
    try:
        for tag, dt, pair in (("pair", "bf16", b"1"), ("two", "bf16", b"0"), ("f32", "f32", b"1")):
            assert L.clhip_config(b"CONV6_PAIR", pair) == 0
            torch.manual_seed(7)
            bb = M.resnet18(args={"dataset": "cifar100"}, dtype=dt).to(DEV)
            bb.train()
            f = bb(x)["features"]
            (f * cw).sum().backward()
            torch.cuda.synchronize()
            out[tag] = (f.detach().float().cpu(), torch.cat([p.grad.detach().float().cpu().reshape(-1) for _, p in bb.named_parameters() if p.grad is not None]))
    finally:
        L.clhip_config(b"CONV6_PAIR", None)
    (fp, gp), (ft, gt), (f3, g3) = out["pair"], out["two"], out["f32"]
    assert torch.equal(fp, ft)
    d = relnorm(gp, gt)
    print("fused vs two-launch stride-2 input gradients: all parameter gradients differ by %.3e; against the f32 mode %.3f / %.3f" % (d, relnorm(gp, g3), relnorm(gt, g3)))
    assert 0.0 < d < 2e-2                                   # (0: the switch did nothing)
    assert relnorm(gp, g3) < 1.1 * relnorm(gt, g3) + 1e-3


def test_bf16_gradients_against_f32_mode_at_batch_256():
    """BASELINE size, the two arithmetic modes of the SAME plan on the same weights and batch (ResNet-18, batch 256, random init):
    relative L2 norm of the difference PER LAYER (not a cosine: a cosine of 0.95 hides a 30 % error vector).

    What the numbers mean.  The f32 mode is the parity mode (pinned against the fp64 reference at 2e-4).  A random-init ReLU network's
    parameter gradient is ill-conditioned: moving the f32 mode's OPERANDS by bf16 rounding alone (weights and input rounded to bf16,
    everything else fp32) already moves its own gradients by the yardstick printed below (observed 0.2-0.35 per layer: 5 % of the ReLU
    masks flip).  The bf16 mode additionally rounds every activation and every activation gradient; its
    deviation must stay within 2x that yardstick per layer and overall.  (The bf16 arithmetic itself is pinned kernel by kernel in
    test_kernels_gpu.py against fp64 on bf16-rounded operands.)"""
    out = {}
    g = torch.Generator().manual_seed(21)
    x = torch.randn(256, 3, 32, 32, generator=g).to(DEV)
    cw = (torch.randn(256, 512, generator=g) / 16).to(DEV)
    for tag, dt, rounded in (("f32", "f32", False), ("f32r", "f32", True), ("bf16", "bf16", False)):
        torch.manual_seed(7)
        bb = M.resnet18(args={"dataset": "cifar100"}, dtype=dt).to(DEV)
        xin = x
        if rounded:
            flat = bb.flat_parameters()[0]
            flat.copy_(flat.bfloat16().float())
            bb.mark_params_modified()
            xin = x.bfloat16().float()
        bb.train()
        f = bb(xin)["features"]
        (f * cw).sum().backward()
        torch.cuda.synchronize()
        out[tag] = (f.detach().float().cpu(), {k: p.grad.detach().float().cpu().clone() for k, p in bb.named_parameters() if p.grad is not None})
    (f32f, g32), (frf, gr), (f16f, g16) = out["f32"], out["f32r"], out["bf16"]
    rows = sorted(((relnorm(g16[k], g32[k]), relnorm(gr[k], g32[k]), k) for k in g32), reverse=True)
    cat = lambda d: torch.cat([d[k].reshape(-1) for k in g32])       # noqa: E731
    tot16, totr = relnorm(cat(g16), cat(g32)), relnorm(cat(gr), cat(g32))
    print("batch 256, relnorm against the f32 mode: features bf16 %.3e / f32-with-bf16-operands %.3e; all gradients %.3f / %.3f" % (
        relnorm(f16f, f32f), relnorm(frf, f32f), tot16, totr))
    print("worst layers (bf16, yardstick, name):", [(round(a, 3), round(b, 3), k) for a, b, k in rows[:6]], "median bf16 %.3f" % rows[len(rows) // 2][0])
    assert relnorm(f16f, f32f) < 2.5e-2                                # 20 layers of bf16 activations: 1.2e-2 observed
    assert tot16 < max(2.0 * totr, 5e-2), (tot16, totr)
    for a, b, k in rows:
        assert a < max(2.0 * b, 2.0 * totr, 5e-2), (k, a, b)


def test_bf16_loss_trajectory_against_f32_mode_over_50_steps():
    """50 LwF / ResNet-18 training steps (task 0, batch 128, SGD lr 0.01 momentum 0.9, the same 50 class-structured batches) in both
    arithmetic modes.  Two runs of the f32 mode give the yardstick (the atomic weight-gradient sums of the stride-2 layers make even
    that pair drift apart); the bf16 run must track the f32 run within 3x of it per step (and 2 % absolute), start within bf16 rounding
    of one forward, and end at the same loss level."""
    from libcontinual_amd.trainer import train_steps

    class Meter:
        def __init__(self):
            self.loss = []

        def update(self, k, v):
            if k == "loss":
                self.loss.append(v)

    def run(dt, perturb=0):
        torch.manual_seed(11)
        bb = M.resnet18(args={"dataset": "cifar100"}, dtype=dt)
        m = M.LWF(bb, 512, 100, device=DEV, init_cls_num=50, inc_cls_num=5).to(DEV)
        m.before_task(0, None, None, None)
        if perturb:                                           # the same start moved by one part in 10^6 (fp32 rounding level)
            flat = m.backbone.flat_parameters()[0]
            g = torch.Generator().manual_seed(900 + perturb)
            with torch.no_grad():
                flat.mul_((1.0 + 1e-6 * (torch.rand(flat.numel(), generator=g) * 2 - 1)).to(flat.device))
            m.backbone.mark_params_modified()
        m.train()
        o = optim.SGD(m.get_parameters({}), lr=0.01, momentum=0.9, weight_decay=5e-4)
        batches = []
        for i in range(50):
            g = torch.Generator().manual_seed(500 + i)
            y = torch.randint(0, 50, (128,), generator=g)
            pat = torch.nn.functional.one_hot(y, 50).float()[:, :48].reshape(128, 3, 4, 4).repeat_interleave(8, 2).repeat_interleave(8, 3)
            batches.append({"image": (torch.randn(128, 3, 32, 32, generator=g) + pat).to(DEV), "label": y.to(DEV)})
        meter = Meter()
        train_steps(m, o, batches, None, "LWF", meter, DEV)
        torch.cuda.synchronize()
        return np.array([float(v.float().item() if torch.is_tensor(v) else v) for v in meter.loss])

    fs, b = [run("f32", q) for q in range(3)], run("bf16")
    # The yardstick: how far f32 runs of this loop drift from EACH OTHER when their start differs at fp32 rounding level.  (Until round 3 two
    # f32 runs differed by themselves -- atomic weight-gradient sums -- by 7e-3 .. 5.5e-2 per step from invocation to invocation; since the
    # generic weight-gradient kernel has its deterministic form the f32 mode is bit-reproducible, so the perturbation is explicit: the
    # initial weights of runs 1 and 2 are moved by one part in 10^6.)  Yardstick = the largest deviation among the three pairs, and the
    # bf16 run is measured against the f32 run it stays closest to.
    pairs = [(0, 1), (0, 2), (1, 2)]
    self_dev = max(float((np.abs(fs[i] - fs[j]) / np.abs(fs[i])).max()) for i, j in pairs)
    devs = [np.abs(b - a) / np.abs(a) for a in fs]
    dev = min(devs, key=lambda d: float(d.max()))
    ends = [float(a[-10:].mean()) for a in fs]
    eb = float(b[-10:].mean())
    print("loss %.3f -> %s; f32 vs f32 per-step deviation max %.2e (3 runs); bf16 vs closest f32 max %.2e, first 5 %s; last-10 means f32 %s bf16 %.4f" % (
        fs[0][0], np.round(ends, 4).tolist(), self_dev, float(dev.max()), np.round(devs[0][:5], 5).tolist(), np.round(ends, 4).tolist(), eb))
    assert min(ends) < 0.9 * fs[0][:3].mean() and max(ends) < 0.95 * fs[0][:3].mean()      # the runs do learn (3.38 .. 3.59 from 3.98 observed)
    assert max(float(d[:3].max()) for d in devs) < 5e-3       # the first steps: bf16 rounding of one forward, against EVERY f32 run
    # floors: what chaos alone produces.  Eight differently initialised runs of this very loop end at 3.32 .. 3.94 in EVERY arithmetic (f32,
    # bf16 on conv4.hip, bf16 on conv5.hip: tools/scratch figures in profiles/r03_parity_notes.md), single bf16 runs sit up to 6.1e-2 off
    # a single f32 run per step and 3 % off at the end while two f32 runs sit up to 5.5e-2 / 4.6 % off each other
    assert float(dev.max()) < max(3 * self_dev, 8e-2), (float(dev.max()), self_dev)
    spread = max(ends) - min(ends)
    assert min(ends) - max(3 * spread, 5e-2 * ends[0]) < eb < max(ends) + max(3 * spread, 5e-2 * ends[0]), (eb, ends)




@pytest.mark.parametrize("arch,feat", [("resnet18", 512), ("cifar_resnet32", 64)])
def test_bf16_logits_of_trained_weights_at_batch_256_within_2e_2(arch, feat):
    """"bf16 path rtol 2e-2 on logits" -- held where it is meaningful: on TRAINED weights at the
    BASELINE batch size.  40 LwF training steps in the f32 parity mode on class-structured data (the random-init networks of the other tests
    are the ill-conditioned case: their logits are sums of near-cancelling terms), then the SAME fp32 master weights and running statistics
    in a bf16-mode model: logits of 256 fresh images, train-mode (batch statistics) and eval-mode (running statistics) forward, within 2e-2
    of the f32 mode's relative to the largest logit, argmax agreement >= 97 %."""
    from libcontinual_amd.trainer import train_steps

    def data(i, B):
        g = torch.Generator().manual_seed(700 + i)
        y = torch.randint(0, 50, (B,), generator=g)
        pat = torch.nn.functional.one_hot(y, 50).float()[:, :48].reshape(B, 3, 4, 4).repeat_interleave(8, 2).repeat_interleave(8, 3)
        return {"image": (torch.randn(B, 3, 32, 32, generator=g) + pat).to(DEV), "label": y.to(DEV)}

    def model(dt):
        torch.manual_seed(13)
        bb = M.resnet18(args={"dataset": "cifar100"}, dtype=dt) if arch == "resnet18" else M.cifar_resnet32(dtype=dt)
        m = M.LWF(bb, feat, 100, device=DEV, init_cls_num=50, inc_cls_num=5).to(DEV)
        m.before_task(0, None, None, None)
        return m
    m32 = model("f32")
    m32.train()
    o = optim.SGD(m32.get_parameters({}), lr=0.02, momentum=0.9, weight_decay=5e-4)
    train_steps(m32, o, [data(i, 128) for i in range(40)], None, "LWF", None, DEV)
    m16 = model("bf16")
    m16.load_state_dict(m32.state_dict())
    x = data(999, 256)
    res = {}
    for mode in ("train", "eval"):
        for tag, m in (("f32", m32), ("bf16", m16)):
            m.train() if mode == "train" else m.eval()
            with torch.no_grad():
                res[mode, tag] = m.classifier(m.backbone(x["image"])["features"]).float().cpu().numpy()
    torch.cuda.synchronize()
    for mode in ("train", "eval"):
        a, b = res[mode, "bf16"], res[mode, "f32"]
        agree = float((a.argmax(1) == b.argmax(1)).mean())
        acc = float((b.argmax(1) == x["label"].cpu().numpy()).mean())
        print(f"{arch} {mode}-mode logits at batch 256 after 40 steps: bf16 vs f32 relmax {relmax(a, b):.3e}, relnorm {relnorm(a, b):.3e}, argmax agreement {agree:.3f} (f32 accuracy {acc:.2f})")
        assert relmax(a, b) < 2e-2, (mode, relmax(a, b))
        assert agree >= 0.97, (mode, agree)


@pytest.mark.parametrize("arch", ["resnet18", "cifar_resnet32"])
def test_num_batches_tracked_counts_training_forwards_only(arch):
    """nn.BatchNorm2d increments `num_batches_tracked` once per training-mode forward (torch/nn/modules/batchnorm.py) and never in eval
    mode; here the increment rides on the plan's first launch (clhip_plan_forward_ex) instead of a torch kernel per step"""
    bb = (M.resnet18(args={"dataset": "cifar100"}, dtype="bf16") if arch == "resnet18" else M.cifar_resnet32(dtype="bf16")).to(DEV)
    x = torch.randn(4, 3, 32, 32, device=DEV)
    bb.train()
    for _ in range(3):
        bb(x)
    bb.eval()
    with torch.no_grad():
        bb(x)
    torch.cuda.synchronize()
    counters = {k: int(v) for k, v in bb.state_dict().items() if k.endswith("num_batches_tracked")}
    assert len(counters) >= 20 and set(counters.values()) == {3}, counters
