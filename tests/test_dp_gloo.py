"""Data-parallel path on CPU with gloo, world_size 2 (the GPU path uses the same code over RCCL): one all-reduce per
flat gradient buffer + one packed bucket for the head; 1/world folded into the optimizer; identical terms added
on every rank (EWC) come out exact; parameters broadcast from rank 0."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import libcontinual_amd.model as M
    from libcontinual_amd import parallel
    parallel.init_distributed(device_is_cuda=False)
    torch.manual_seed(100 + rank)                   # different init per rank on purpose
    bb = M.cifar_resnet32()
    head = torch.nn.Linear(64, 10)
    net = torch.nn.ModuleDict({"backbone": bb, "classifier": head})
    parallel.broadcast_module_state(net)
    flat, gflat = bb.flat_parameters()
    ref = flat.clone()
    # every rank fabricates its local gradient: g_r = (r+1) * pattern  (+ a rank-independent "EWC" term)
    pattern = torch.arange(flat.numel(), dtype=torch.float32) % 7 - 3
    gflat.copy_((rank + 1) * pattern + 5.0)
    bb.attach_grads()
    head.weight.grad = torch.full_like(head.weight, float(rank + 1))
    head.bias.grad = torch.full_like(head.bias, 2.0 * (rank + 1))
    red = parallel.GradientReducer()
    red.reduce(net)
    scale = 1.0 / world
    want = (sum(r + 1 for r in range(world)) * pattern + 5.0 * world) * scale
    ok = torch.allclose(gflat * scale, want) and torch.allclose(head.weight.grad * scale, torch.full_like(head.weight, 1.5)) \
        and torch.allclose(head.bias.grad * scale, torch.full_like(head.bias, 3.0))
    # gradients are still the views the optimizer reads
    ok = ok and bb._params[0].grad.data_ptr() == gflat.data_ptr()
    gathered = [torch.zeros_like(ref) for _ in range(world)]
    dist.all_gather(gathered, ref)
    ok = ok and all(torch.equal(g, gathered[0]) for g in gathered)          # broadcast made the ranks identical
    # overlapped exchange: the tail of the flat buffer is handed over early (as the segmented backward does), reduce() finishes
    gflat.copy_((rank + 1) * pattern + 5.0)
    head.weight.grad = torch.full_like(head.weight, float(rank + 1))
    with red.overlap(net, fraction=0.5):
        k = bb.grad_cut_for_fraction(0.5)
        cut = bb._unit_off[k]
        ok = ok and k > 0 and bb._grad_segment_hook is not None and bb._nflat - cut >= 0.5 * bb._nflat
        bb._grad_segment_hook(bb, cut, bb._nflat)          # "units >= k are done"
        bb._grad_segment_hook(bb, 0, cut)                  # the head of the buffer is left to reduce()
        red.reduce(net)
    ok = ok and bb._grad_segment_hook is None and not red._early
    ok = ok and torch.allclose(gflat * scale, want) and torch.allclose(head.weight.grad * scale, torch.full_like(head.weight, 1.5))
    # plugins that clip inside observe (L2P) own the reduction: mean gradient first, then the clip, optimizer unscaled
    class SelfReducing(torch.nn.Module):
        reduces_own_gradients = True
        grad_reducer = None

        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(4))

    class Opt:
        grad_scale = 7.0

    sr, opt2 = SelfReducing(), Opt()
    ok = ok and parallel.attach(sr, opt2, red) and sr.grad_reducer is red and opt2.grad_scale == 1.0
    sr.w.grad = torch.full((4,), 3.0 * (rank + 1))
    sr.grad_reducer.reduce_mean(sr)
    ok = ok and torch.allclose(sr.w.grad, torch.full((4,), 4.5))                 # mean of 3 and 6
    norm = torch.nn.utils.clip_grad_norm_([sr.w], 1.0)
    ok = ok and abs(float(norm) - 9.0) < 1e-5 and abs(float(sr.w.grad.norm()) - 1.0) < 1e-4
    plain, opt3 = torch.nn.Linear(2, 2), Opt()
    ok = ok and (not parallel.attach(plain, opt3, red)) and opt3.grad_scale == 0.5
    m = red.mean_scalar(float(rank), "cpu")
    ok = ok and abs(m - 0.5) < 1e-12
    # ---- the sharded exchange: reduce-scatter -> update of this rank's shard only -> in-place all-gather of the flat parameters
    from libcontinual_amd import optim as fused
    rs = parallel.GradientReducer(exchange="reduce_scatter")
    n = flat.numel()
    # (a) ADVICE r2: an optimizer that does NOT step the backbone as one flat range (torch.optim fallback, or no optimizer attached)
    #     would step on its local gradients after a reduce-scatter: those buckets take the in-place all-reduce instead
    gflat.copy_((rank + 1) * pattern + 5.0)
    bb.attach_grads()
    parallel.attach(net, torch.optim.SGD(net.parameters(), lr=0.1), rs)
    rs.reduce(net)
    ok = ok and getattr(bb, "_dp_shard", None) is None and torch.allclose(gflat * scale, want)
    # (b) half of the backbone frozen-by-group: split across two param groups -> not "whole" -> all-reduce as well
    gflat.copy_((rank + 1) * pattern + 5.0)
    ps = list(bb.parameters())
    split_opt = fused.SGD([{"params": ps[:10]}, {"params": ps[10:] + list(head.parameters())}], lr=0.1)
    parallel.attach(net, split_opt, rs)
    ok = ok and split_opt.whole_backbones() == []
    rs.reduce(net)
    ok = ok and getattr(bb, "_dp_shard", None) is None and torch.allclose(gflat * scale, want)
    # (c) the optimizer refuses a shard record it would not consume (second line of defence)
    bb._dp_shard = dict(lo=0, hi=4, prefix=4, grad=gflat[:4].clone(), reducer=rs)
    try:
        split_opt._check_unconsumed_shards(ps)
        ok = False
    except RuntimeError:
        pass
    bb._dp_shard = None
    # (d) the fused optimizer with the whole backbone in one group: the sharded path
    whole_opt = fused.SGD(net.parameters(), lr=0.1)
    parallel.attach(net, whole_opt, rs)
    ok = ok and [id(o) for o in whole_opt.whole_backbones()] == [id(bb)]
    per, prefix = rs.shard_bounds(n)
    ok = ok and per % 4 == 0 and prefix == per * world and 0 <= n - prefix < 4 * world
    gflat.copy_((rank + 1) * pattern + 5.0)
    head.weight.grad = torch.full_like(head.weight, float(rank + 1))
    flat.copy_(ref)
    with rs.overlap(net, fraction=0.5):                                   # no early segments in this mode
        ok = ok and bb._grad_segment_hook is None
        rs.reduce(net)
    d = bb._dp_shard
    total = sum(r + 1 for r in range(world)) * pattern + 5.0 * world
    ok = ok and d["lo"] == rank * per and d["hi"] == (rank + 1) * per and d["prefix"] == prefix
    ok = ok and torch.allclose(d["grad"], total[d["lo"]:d["hi"]]) and torch.allclose(gflat[prefix:], total[prefix:])
    ok = ok and torch.allclose(head.weight.grad * scale, torch.full_like(head.weight, 1.5))            # the head still goes through the small all-reduce
    # what the fused optimizer does with the record (optim._dp_plan): its slices, then the gather
    from libcontinual_amd.optim import _dp_plan
    parts, shard = _dp_plan(bb)
    ok = ok and shard is d and bb._dp_shard is None and [sfx for _, _, sfx in parts] == (["_shard", "_tail"] if prefix < n else ["_shard"])
    for pslice, gslice, _ in parts:
        pslice.sub_(0.1 * scale * gslice)                                  # a plain SGD step on the slices this rank owns
    bb._dp_shard = shard
    rs.gather_params(bb)
    ok = ok and torch.allclose(flat, ref - 0.1 * scale * total, atol=1e-6)  # every rank now holds the full updated buffer
    # plugins that post-process the whole gradient get the all-reduce whatever the mode
    sr.grad_reducer = rs
    sr.w.grad = torch.full((4,), 3.0 * (rank + 1))
    rs.reduce_mean(sr)
    ok = ok and torch.allclose(sr.w.grad, torch.full((4,), 4.5))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gradient_reducer_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def _worker_w(rank, world, port, q):
    """world-size-generic checks (run at world 8): shard arithmetic with a ragged tail (n % (4 * world) != 0), the all-reduce with the early tail
    hand-over, the reduce-scatter -> shard update -> all-gather exchange, the BatchNorm-buffer broadcast that precedes after_task"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), OMP_NUM_THREADS="1")
    torch.set_num_threads(1)
    import libcontinual_amd.model as M
    from libcontinual_amd import optim as fused, parallel
    from libcontinual_amd.optim import _dp_plan
    parallel.init_distributed(device_is_cuda=False)
    torch.manual_seed(100 + rank)
    bb = M.cifar_resnet32()
    head = torch.nn.Linear(64, 55)
    net = torch.nn.ModuleDict({"backbone": bb, "classifier": head})
    bb._stats.copy_(torch.full_like(bb._stats, float(rank)))             # per-rank running statistics (DDP-faithful BatchNorm)
    parallel.broadcast_module_state(net)
    flat, gflat = bb.flat_parameters()
    ref = flat.clone()
    ok = bool((bb._stats == 0).all())                                       # rank 0's buffers everywhere
    n = flat.numel()
    pattern = torch.arange(n, dtype=torch.float32) % 11 - 5
    ranks_sum = float(sum(r + 1 for r in range(world)))
    total = ranks_sum * pattern + 5.0 * world
    scale = 1.0 / world
    # ---- all-reduce, the tail of the buffer handed over early
    red = parallel.GradientReducer()
    gflat.copy_((rank + 1) * pattern + 5.0)
    bb.attach_grads()
    head.weight.grad = torch.full_like(head.weight, float(rank + 1))
    head.bias.grad = torch.full_like(head.bias, 2.0 * (rank + 1))
    with red.overlap(net, fraction=0.5):
        k = bb.grad_cut_for_fraction(0.5)
        cut = bb._unit_off[k]
        bb._grad_segment_hook(bb, cut, bb._nflat)
        bb._grad_segment_hook(bb, 0, cut)
        red.reduce(net)
    ok = ok and torch.allclose(gflat, total) and torch.allclose(head.weight.grad, torch.full_like(head.weight, ranks_sum))
    ok = ok and torch.allclose(head.bias.grad, torch.full_like(head.bias, 2.0 * ranks_sum))
    # ---- reduce-scatter with a ragged tail
    rs = parallel.GradientReducer(exchange="reduce_scatter")
    per, prefix = rs.shard_bounds(n)
    ok = ok and per % 4 == 0 and prefix == per * world and 0 < n - prefix < 4 * world          # 466256 = 8 * 58280 + 16
    whole_opt = fused.SGD(net.parameters(), lr=0.1)
    parallel.attach(net, whole_opt, rs)
    gflat.copy_((rank + 1) * pattern + 5.0)
    head.weight.grad = torch.full_like(head.weight, float(rank + 1))
    flat.copy_(ref)
    rs.reduce(net)
    d = bb._dp_shard
    ok = ok and d is not None and d["lo"] == rank * per and d["hi"] == (rank + 1) * per and d["prefix"] == prefix
    ok = ok and torch.allclose(d["grad"], total[d["lo"]:d["hi"]]) and torch.allclose(gflat[prefix:], total[prefix:])
    parts, shard = _dp_plan(bb)
    ok = ok and [sfx for _, _, sfx in parts] == ["_shard", "_tail"]
    for pslice, gslice, _ in parts:
        pslice.sub_(0.1 * scale * gslice)
    bb._dp_shard = shard
    rs.gather_params(bb)
    ok = ok and torch.allclose(flat, ref - 0.1 * scale * total, atol=1e-6)
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    ok = ok and all(torch.equal(g, gathered[0]) for g in gathered)
    ok = ok and abs(rs.mean_scalar(float(rank), "cpu") - (world - 1) / 2.0) < 1e-12
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gradient_reducer_world8():
    """the world-size > 2 arithmetic (8 ranks = the node BASELINE configs[2] / [4] name) on CPU over gloo"""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_w, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=420) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(r, True) for r in range(world)]
