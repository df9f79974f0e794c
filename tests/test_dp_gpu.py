"""Data-parallel plumbing on ONE MI355X: the segmented backward (clhip_plan_backward_range) with the overlapped all-reduce of
the finished tail of the flat gradient buffer, over a real RCCL process group of world size 1 (the reducer is told the
world is 2 so that every code path runs; a 1-rank all-reduce returns its input, so the result must equal the plain
single-call backward).  The multi-rank arithmetic is covered on CPU by tests/test_dp_gloo.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

import libcontinual_amd.model as M                     # noqa: E402
from libcontinual_amd import ops, optim, parallel      # noqa: E402
from libcontinual_amd.trainer import train_steps       # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def rccl_group():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    yield
    dist.destroy_process_group()


def _make(seed):
    torch.manual_seed(seed)
    bb = M.resnet18(args={"dataset": "cifar100"}, dtype="bf16")
    m = M.LWF(bb, 512, 100, device="cuda", init_cls_num=50, inc_cls_num=5).to("cuda")
    m.before_task(0, None, None, None)
    return m


def _batch(seed, B=32):
    g = torch.Generator().manual_seed(seed)
    return {"image": torch.randn(B, 3, 32, 32, generator=g).cuda(), "label": torch.randint(0, 50, (B,), generator=g).cuda()}


def test_segmented_backward_matches_single_call():
    m1, m2 = _make(3), _make(3)
    b = _batch(1)
    for m in (m1, m2):
        m.train()
    bb2 = m2.backbone
    seen = []
    bb2._grad_segment_cuts = [bb2.grad_cut_for_fraction(0.5), 5]
    bb2._grad_segment_hook = lambda bb, lo, hi: seen.append((lo, hi))
    for m in (m1, m2):
        _, _, loss = m.observe(b)
        loss.backward()
    torch.cuda.synchronize()
    g1, g2 = m1.backbone.flat_parameters()[1], bb2.flat_parameters()[1]
    # same kernels in the same order; the stride-2 / 1x1 weight gradients use fp32 atomics, so equality is up to their
    # summation order
    assert float((g1 - g2).abs().max()) <= 2e-3 * float(g1.abs().max())
    assert seen[0][1] == bb2._nflat and seen[-1][0] == 0 and all(seen[i][0] == seen[i + 1][1] for i in range(len(seen) - 1))
    assert bb2._nflat - seen[0][0] >= 0.5 * bb2._nflat            # ResNet-18: layer4 holds > 50 % of the parameters


def test_overlapped_reduce_on_rccl(rccl_group):
    m1, m2 = _make(4), _make(4)
    batches = [_batch(10)]
    o1 = optim.SGD(m1.get_parameters({}), lr=0.05, momentum=0.9)
    o2 = optim.SGD(m2.get_parameters({}), lr=0.05, momentum=0.9)
    for m in (m1, m2):
        m.train()
    train_steps(m1, o1, batches, None, "LWF", None, "cuda")
    red = parallel.GradientReducer()
    red.world = 2                                                 # exercise every DP code path on a 1-rank RCCL group
    parallel.attach(m2, o2, red)
    assert o2.grad_scale == 0.5
    o2.grad_scale = 1.0                                           # ... whose "sum over ranks" is the local gradient
    calls = []
    orig = red._on_segment
    red._on_segment = lambda bb, lo, hi: (calls.append((lo, hi)), orig(bb, lo, hi))[1]
    train_steps(m2, o2, batches, red, "LWF", None, "cuda")
    torch.cuda.synchronize()
    assert len(calls) == 2 * len(batches) and m2.backbone._grad_segment_hook is None     # hooks removed after the loop
    f1, f2 = m1.backbone.flat_parameters()[0], m2.backbone.flat_parameters()[0]
    assert float((f1 - f2).abs().max()) <= 2e-3 * float(f1.abs().max())        # one step; fp32-atomic summation order only
    assert float((m1.classifier.weight - m2.classifier.weight).detach().abs().max()) <= 1e-4


def test_reduced_step_does_not_stall_under_the_queue_cap():
    """the batch-256 ResNet-18 step with the weight-gradient stream AND the reducer's overlapped all-reduce on RCCL's own stream
    (1-rank group, the reducer told the world is 2), under the hardware-queue cap a RANK gets (4: libcontinual_amd/__init__.py) -- the collective's
    streams must neither re-create the multi-stream stall of profiles/r04_stream_stall.md (4.2 ms at a cap of 6) nor push the weight-gradient stream
    onto a shared hardware queue (2.6 ms in some stream-creation orders at a cap of 2 or 3, plain steps included: round 5, profiles/r05_notes.md).
    Runs tools/dp_step_micro.py in a process of its own (the cap is read when the HIP runtime starts): several models one after the other, so that
    later plans' streams land on every queue; every reduced step stays within 15 % of the fastest plain step and plain steps within 8 % of each other
    (the stalls this guards against are +27 % and +100 %; a busy box moved a plain step by 5 % once)."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(WORLD_SIZE="2", DP_MICRO_KINDS="plain,tail,tail,plain,tail,tail,plain")       # WORLD_SIZE > 1 at import -> the rank's cap; the script itself builds a 1-rank group
    # (a timing test on a shared box: ONE retry -- the stalls this guards against are reproducible, a hiccup of the box is not)
    for attempt in range(2):
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "dp_step_micro.py")], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        rows = re.findall(r"^(plain|tail) ([0-9.]+) ms", out.stdout, re.M)
        cap = re.findall(r"^hw_queue_cap (.*)$", out.stdout, re.M)
        assert cap and "'ok', '4'" in cap[0], out.stdout[-800:]
        plain = [float(v) for k, v in rows if k == "plain"]
        tail = [float(v) for k, v in rows if k == "tail"]
        print(f"batch-256 ResNet-18 LwF step in a rank's configuration (GPU_MAX_HW_QUEUES=4, 1-rank RCCL group): plain {plain} ms, with the overlapped all-reduce {tail} ms")
        assert len(plain) == 3 and len(tail) == 4
        if max(plain) < 1.08 * min(plain) and max(tail) < 1.15 * min(plain):
            break
    assert max(plain) < 1.08 * min(plain), plain
    assert max(tail) < 1.15 * min(plain), (plain, tail)


def _make_r32(seed):
    torch.manual_seed(seed)
    bb = M.cifar_resnet32(dtype="bf16")
    m = M.EWC(bb, 64, 100, device="cuda", init_cls_num=50, inc_cls_num=5, lamda=100.0).to("cuda")
    m.before_task(0, None, None, None)
    m.train()
    return m


def test_reduced_step_replays_from_a_graph_bit_for_bit(rccl_group, monkeypatch):
    """the data-parallel step -- backward, the RCCL all-reduce of the flat gradient buffer and of the head's bucket, the fused
    optimizer -- is captured into ONE HIP graph and replayed (trainer.GraphedStep with a reducer); on a 1-rank RCCL group (the reducer told the
    world is 2, so every collective is really enqueued) ten steps of which eight are replays end bit-identical to ten eager steps of the same
    reduced loop (CifarResNet-32 at 32 images: every kernel of the step is deterministic, tests/test_graph_step_gpu.py), and CLHIP_DP_GRAPH=0 keeps
    a reduced loop eager."""
    monkeypatch.setenv("CLHIP_SGD_ZERO", "0")
    res = []
    for mode, dpg in (("0", "1"), ("1", "1"), ("1", "0")):
        monkeypatch.setenv("CLHIP_CUDA_GRAPH", mode)
        monkeypatch.setenv("CLHIP_DP_GRAPH", dpg)
        m = _make_r32(11)
        o = optim.SGD(m.get_parameters({}), lr=0.05, momentum=0.9, weight_decay=5e-4)
        red = parallel.GradientReducer()
        red.world = 2
        parallel.attach(m, o, red)
        assert o.grad_scale == 0.5                                  # kept: both runs scale the ("summed") gradient the same way
        batches = [_batch(40 + i) for i in range(10)]
        train_steps(m, o, batches, red, "EWC", None, "cuda")
        torch.cuda.synchronize()
        res.append((m.network.backbone.flat_parameters()[0].clone(), m.network.classifier.weight.detach().clone(), getattr(m, "_graphed_step", None)))
    (p0, h0, g0), (p1, h1, g1), (p2, h2, g2) = res
    assert g0 is None and g2 is None
    assert g1 is not None and g1.reducer is not None and len(g1.graphs) == 1 and not g1.disabled
    assert torch.equal(p0, p1) and torch.equal(h0, h1)
    assert torch.equal(p0, p2)
