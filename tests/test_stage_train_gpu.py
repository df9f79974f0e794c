"""stage_train.hip: the TRAINING forward / backward of a run of BasicBlocks of the CIFAR ResNet-32s as one launch per direction (one workgroup per image, the
batch statistics of every convolution through the in-launch all-reduce of xch.h) against the per-unit launches it replaces (STAGE_TRAIN=0): same z, same
saved / running statistics, same block outputs up to the summation order of the statistics; bit-reproducible; either backward follows either forward."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import libcontinual_amd.model as M                     # noqa: E402
from libcontinual_amd import _lib                      # noqa: E402


def _backbone(kind, seed):
    torch.manual_seed(seed)
    bb = (M.cifar_resnet32 if kind == "cifar" else M.resnet32_V2)(dtype="bf16").to("cuda")
    bb.train()
    return bb


def _trained_like(bb, seed):
    """move the BatchNorm parameters and the running statistics away from (1, 0) / (0, 1) so that a wrong coefficient shows"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in bb.named_parameters():
            if p.dim() == 1:
                p.copy_((1.0 + 0.3 * torch.randn(p.shape, generator=g)).to(p.device) if n.endswith("weight") else (0.2 * torch.randn(p.shape, generator=g)).to(p.device))
    bb.mark_params_modified()


def _status(bb):
    L = _lib.lib()
    return [int(L.clhip_plan_stage_status(p)) for p, _ in bb._handle.plans.values()]


def _launches(bb):
    """(units inside training runs, stage-level forward launches, backward launches) over the backbone's plans"""
    L = _lib.lib()
    return tuple(sum(int(L.clhip_plan_stage_info(p, w)) for p, _ in bb._handle.plans.values()) for w in (0, 1, 2))


def _forward_reads(bb, x, flag):
    L = _lib.lib()
    assert L.clhip_config(b"STAGE_TRAIN", flag) == 0
    try:
        stats0 = bb._stats.clone()
        f = bb(x)["features"].clone()
        n_act = len(bb._units)
        z = [bb.debug_read(a, 1).clone() for a in range(1, n_act + 1)]
        y = [bb.debug_read(a, 0).clone() for a in range(1, n_act + 1)]
        stats1 = bb._stats.clone()
        bb._stats.copy_(stats0)                          # every variant starts from the same running statistics
        torch.cuda.synchronize()
    finally:
        L.clhip_config(b"STAGE_TRAIN", None)
    return f, z, y, stats1


@pytest.mark.parametrize("kind,batch", [("cifar", 32), ("cifar", 256), ("cifar", 100), ("v2", 64)])
def test_stage_level_training_forward_matches_the_per_unit_launches(kind, batch):
    bb = _backbone(kind, 3)
    _trained_like(bb, 4)
    x = torch.randn(batch, 3, 32, 32, generator=torch.Generator().manual_seed(5)).cuda()
    with torch.no_grad():
        bb(x)                                             # plans, workspaces, weight copies
    n0 = _launches(bb)
    f1, z1, y1, s1 = _forward_reads(bb, x, b"1")
    n1 = _launches(bb)
    f0, z0, y0, s0 = _forward_reads(bb, x, b"0")
    n2 = _launches(bb)
    f1b, z1b, y1b, s1b = _forward_reads(bb, x, b"1")
    assert _status(bb) == [0] * len(bb._handle.plans)
    assert n0[0] == (26 if kind == "cifar" else 24) and n1[1] - n0[1] == 3 and n2[1] == n1[1], (n0, n1, n2)          # three runs (10 + 8 + 8 units); none with the switch off
    assert torch.isfinite(f1).all() and float(f1.detach().abs().max()) > 0
    # bit-reproducible from run to run
    assert torch.equal(f1, f1b) and torch.equal(s1, s1b) and all(torch.equal(a, b) for a, b in zip(z1, z1b)) and all(torch.equal(a, b) for a, b in zip(y1, y1b))
    # the per-unit path sums its statistics in fp64 atomics (arrival order, per-tile fp32 partial sums), this one in a fixed tree of per-image partial sums: the same
    # numbers to ~1e-7, so a bf16 value flips its last bit here and there -- and a network that STORES bf16 amplifies a flipped bit by ~1.5-2 x per layer (the
    # storage-format floor of profiles/r04_bf16_floor_study.md): tight where the paths first meet, the floor's level at the end
    worst = 0.0
    for a, (u, v) in enumerate(zip(z1, z0)):
        d = float((u - v).abs().max()) / max(float(v.abs().max()), 1e-6)
        worst = max(worst, d)
        l2 = float((u - v).norm()) / max(float(v.norm()), 1e-12)
        ly = float((y1[a] - y0[a]).norm()) / max(float(y0[a].norm()), 1e-12)
        assert l2 <= (1e-4 if a <= 4 else 6e-2) and ly <= (1e-4 if a <= 4 else 6e-2), f"unit {a}: relative L2 distance z {l2:.2e} y {ly:.2e}"
        assert d <= 8e-2, f"z of unit {a}: {d}"
    df = float((f1 - f0).abs().max()) / float(f0.abs().max())
    ds = float((s1 - s0).abs().max()) / float(s0.abs().max())
    print(f"{kind} batch {batch}: features {df:.2e}, running statistics {ds:.2e}, worst z {worst:.2e}")
    assert df <= 5e-2 and ds <= 2e-3

def _step(bb, x, w, fwd, bwd):
    """one forward + backward of sum(features * w) with the stage-level launches on / off per direction; -> (features, flat gradient)"""
    L = _lib.lib()
    stats0 = bb._stats.clone()
    for prm in bb.parameters():
        prm.grad = None
    try:
        assert L.clhip_config(b"STAGE_TRAIN", fwd) == 0
        f = bb(x)["features"]
        assert L.clhip_config(b"STAGE_TRAIN", bwd) == 0
        (f * w).sum().backward()
        torch.cuda.synchronize()
    finally:
        L.clhip_config(b"STAGE_TRAIN", None)
    g = bb.flat_parameters()[1].clone()
    bb._stats.copy_(stats0)
    return f.detach().clone(), g


def _rel(a, b):
    return float((a - b).norm()) / max(float(b.norm()), 1e-12)


@pytest.mark.parametrize("kind,batch", [("cifar", 32), ("cifar", 256), ("cifar", 72), ("v2", 128)])
def test_stage_level_training_backward_matches_the_per_unit_launches(kind, batch):
    bb = _backbone(kind, 7)
    _trained_like(bb, 8)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(batch, 3, 32, 32, generator=g).cuda()
    w = torch.randn(batch, bb.out_dim, generator=g).cuda() / batch
    n0 = _launches(bb)
    f11, g11 = _step(bb, x, w, b"1", b"1")
    n1 = _launches(bb)
    assert n1[1] - n0[1] == 3 and n1[2] - n0[2] == 3, (n0, n1)
    f11b, g11b = _step(bb, x, w, b"1", b"1")
    f00, g00 = _step(bb, x, w, b"0", b"0")
    f10, g10 = _step(bb, x, w, b"1", b"0")               # per-unit backward behind the stage-level forward
    f01, g01 = _step(bb, x, w, b"0", b"1")               # ... and the other way round
    assert _status(bb) == [0] * len(bb._handle.plans)
    assert torch.isfinite(g11).all() and float(g11.abs().max()) > 0
    assert torch.equal(f11, f11b) and torch.equal(g11, g11b), "a stage-level step is not bit-reproducible"
    # behind the SAME forward the two backward paths round dy / dz at the same places and differ by summation order only; behind different forwards the
    # gradients sit the storage-format floor apart (a flipped bf16 bit of an activation grows ~2 x per layer: 0.2-0.5 relative at this depth, see the forward test)
    worst = ("", 0.0)
    for name, shape, off, _ in bb._layout:
        cnt = 1
        for d_ in shape:
            cnt *= int(d_)
        for a, b in ((g11[off:off + cnt], g10[off:off + cnt]), (g01[off:off + cnt], g00[off:off + cnt])):
            if float(b.norm()) == 0.0:
                assert float(a.norm()) == 0.0, name
                continue
            d = _rel(a, b)
            if d > worst[1]:
                worst = (name, d)
    print(f"{kind} batch {batch}: whole gradient, stage vs per-unit backward: behind the stage forward {_rel(g11, g10):.2e}, behind the per-unit forward {_rel(g01, g00):.2e}; "
          f"worst tensor {worst[0]} {worst[1]:.2e}; stage step vs per-unit step {_rel(g11, g00):.2e}")
    assert _rel(g11, g10) <= 3e-2 and _rel(g01, g00) <= 3e-2
    assert worst[1] <= 6e-2, worst
    assert _rel(g11, g00) <= 0.6


@pytest.mark.parametrize("kind,batch", [("cifar", 256), ("v2", 72)])
def test_the_pooling_inside_the_last_runs_launches_changes_no_bit(kind, batch):
    """STAGE_POOL (default on): the last run's forward launch also averages its output, its backward launch forms the output gradient from the feature gradient --
    same sums in the same order, same bf16 rounding of the gradient as the two pooling launches they replace"""
    bb = _backbone(kind, 11)
    _trained_like(bb, 12)
    g = torch.Generator().manual_seed(13)
    x = torch.randn(batch, 3, 32, 32, generator=g).cuda()
    w = torch.randn(batch, bb.out_dim, generator=g).cuda() / batch
    L = _lib.lib()
    out = {}
    for flag in (b"1", b"0", b"1"):
        assert L.clhip_config(b"STAGE_POOL", flag) == 0
        try:
            out.setdefault(flag, []).append(_step(bb, x, w, b"1", b"1"))
        finally:
            L.clhip_config(b"STAGE_POOL", None)
    assert _status(bb) == [0] * len(bb._handle.plans)
    (f1, g1), (f1b, g1b) = out[b"1"]
    (f0, g0), = out[b"0"]
    assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    assert torch.equal(f1, f1b) and torch.equal(g1, g1b)
    assert torch.equal(f1, f0), float((f1 - f0).abs().max())
    assert torch.equal(g1, g0), _rel(g1, g0)


@pytest.mark.parametrize("kind,batch", [("cifar", 256), ("cifar", 100), ("v2", 128)])
def test_the_three_level_exchange_agrees_with_the_two_hop_form(kind, batch):
    """STAGE_XCH3 (default on where the device's workgroup -> XCD rule holds): the statistics exchange of big grids goes XCD-local first (plain stores, one sweeping
    member per XCD, the XCDs' partial sums across).  Another fixed summation order than the two-hop form's: the same totals to fp64 rounding, so the first units agree
    to 1e-4 and the rest to the storage-format floor; bit-reproducible from run to run; no wait ever timed out"""
    bb = _backbone(kind, 21)
    _trained_like(bb, 22)
    g = torch.Generator().manual_seed(23)
    x = torch.randn(batch, 3, 32, 32, generator=g).cuda()
    w = torch.randn(batch, bb.out_dim, generator=g).cuda() / batch
    L = _lib.lib()
    res = {}
    for flag in (b"1", b"0", b"1"):
        assert L.clhip_config(b"STAGE_XCH3", flag) == 0
        try:
            f, z, y, s = _forward_reads(bb, x, b"1")
            fg = _step(bb, x, w, b"1", b"1")
        finally:
            L.clhip_config(b"STAGE_XCH3", None)
        res.setdefault(flag, []).append((f, z, s, fg))
    assert _status(bb) == [0] * len(bb._handle.plans)
    (f1, z1, s1, (ff1, g1)), (f1b, z1b, s1b, (ff1b, g1b)) = res[b"1"]
    (f0, z0, s0, (ff0, g0)), = res[b"0"]
    assert torch.equal(f1, f1b) and torch.equal(s1, s1b) and torch.equal(g1, g1b) and all(torch.equal(a, b) for a, b in zip(z1, z1b))
    for a, (u, v) in enumerate(zip(z1, z0)):
        l2 = float((u - v).norm()) / max(float(v.norm()), 1e-12)
        assert l2 <= (1e-4 if a <= 4 else 6e-2), f"unit {a}: {l2:.2e}"
    assert float((s1 - s0).abs().max()) / float(s0.abs().max()) <= 2e-3
    assert torch.isfinite(g1).all() and _rel(g1, g0) <= 0.6


@pytest.mark.parametrize("kind,batch", [("cifar", 256), ("v2", 96)])
def test_a_forward_no_backward_follows_leaves_the_same_features_and_statistics(kind, batch):
    """clhip_plan_forward_ex(training = 2) -- what ops.TeacherPass asks for around a teacher that runs on batch statistics: the stage-level launches skip the z /
    activation stores inside a run; features and running statistics are bit-identical to the saving forward's"""
    from libcontinual_amd.model.backbone import resnet as R
    bb = _backbone(kind, 31)
    _trained_like(bb, 32)
    x = torch.randn(batch, 3, 32, 32, generator=torch.Generator().manual_seed(33)).cuda()
    stats0 = bb._stats.clone()
    with torch.no_grad():
        f_save = bb(x)["features"].clone()
    s_save = bb._stats.clone()
    bb._stats.copy_(stats0)
    n0 = _launches(bb)
    with torch.no_grad(), R.no_backward_follows():
        f_ns = bb(x)["features"].clone()
    s_ns = bb._stats.clone()
    n1 = _launches(bb)
    torch.cuda.synchronize()
    assert n1[1] - n0[1] == 3                              # still the three stage-level launches
    assert _status(bb) == [0] * len(bb._handle.plans)
    assert torch.isfinite(f_ns).all() and float(f_ns.abs().max()) > 0
    assert torch.equal(f_save, f_ns) and torch.equal(s_save, s_ns)
    # and a training step right after it is unaffected (the workspace holds nothing of the no-save forward that the step would read)
    w = torch.randn(batch, bb.out_dim, generator=torch.Generator().manual_seed(34)).cuda() / batch
    bb._stats.copy_(stats0)
    fa, ga = _step(bb, x, w, b"1", b"1")
    with torch.no_grad(), R.no_backward_follows():
        bb(x)
    bb._stats.copy_(stats0)
    fb, gb = _step(bb, x, w, b"1", b"1")
    assert torch.equal(fa, fb) and torch.equal(ga, gb)
