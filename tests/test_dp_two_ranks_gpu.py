"""The real N>1 code path on a 1-GPU box: two processes share cuda:0 and exchange through gloo (RCCL refuses two ranks on
one device; the env hooks CLHIP_DIST_BACKEND / CLHIP_SHARED_GPU exist for exactly this).  Everything except the transport is
what an 8-GPU run executes: rank-0 broadcast, per-rank batches, segmented backward with the overlapped all-reduce of the flat
gradient tail, the small bucket for the head, 1/world folded into the fused optimizer step, bench.py's barrier / max-over-ranks
timing contract."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(script_args, timeout=600, nproc=2, **extra):
    env = dict(os.environ, CLHIP_DIST_BACKEND="gloo", CLHIP_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **dict({"OMP_NUM_THREADS": "4"}, **extra))
    def cmd_with(port):
        return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    # the port is free when it is picked, not necessarily when torchrun's store binds it a second later (an ephemeral port can be handed to another
    # connection in between: one EADDRINUSE in ~800 launches on the GPU boxes) -- retry with a fresh one
    for attempt in range(4):
        r = subprocess.run(cmd_with(_free_port()), cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
        if r.returncode == 0 or "EADDRINUSE" not in r.stderr:
            break
    return r


def test_two_ranks_train_in_lockstep_and_match_the_emulation(tmp_path):
    steps = 3
    r = _launch([os.path.join(ROOT, "tests", "dp_worker.py"), str(tmp_path), str(steps)])
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    # ranks end with the same parameters: bit-identical (same reduced gradient, same fused step)
    np.testing.assert_array_equal(a["flat"], b["flat"])
    np.testing.assert_array_equal(a["head"], b["head"])
    assert np.abs(a["rm"] - b["rm"]).max() > 0          # ... but per-rank BatchNorm statistics (DDP-faithful)
    np.testing.assert_array_equal(a["rm_synced"], b["rm_synced"])      # until the end-of-task broadcast makes the replicas one model
    np.testing.assert_array_equal(a["rm_synced"], a["rm"])             # (rank 0's statistics)
    # the head's bucket: the reduced bucket IS the gradient afterwards (views of it: no copy_ per tensor behind the collective)
    assert int(a["head_grad_in_bucket"]) == 1

    # single-process emulation: two replicas, each on its rank's batches, gradients averaged by hand, same optimizer
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dp_worker as W
    from libcontinual_amd import optim
    reps = [W.make(100), W.make(100)]
    opts = [optim.SGD(m.get_parameters({}), lr=0.05, momentum=0.9, weight_decay=5e-4) for m in reps]
    for m in reps:
        m.train()
    for i in range(steps):
        for rank, (m, o) in enumerate(zip(reps, opts)):
            _, _, loss = m.observe(W.batch(1000 * rank + i))
            o.zero_grad()
            loss.backward()
        gs = [m.backbone.flat_parameters()[1] for m in reps]
        mean = (gs[0] + gs[1]) / 2
        hw = (reps[0].classifier.weight.grad + reps[1].classifier.weight.grad) / 2
        hb = (reps[0].classifier.bias.grad + reps[1].classifier.bias.grad) / 2
        for m, o in zip(reps, opts):
            m.backbone.flat_parameters()[1].copy_(mean)
            m.classifier.weight.grad.copy_(hw)
            m.classifier.bias.grad.copy_(hb)
            o.step()
    torch.cuda.synchronize()
    flat = reps[0].backbone.flat_parameters()[0].cpu().numpy()
    # equal up to the summation order of the fp32-atomic weight gradients (and sum-then-scale vs scale-in-step rounding)
    assert np.abs(flat - a["flat"]).max() <= 2e-3 * np.abs(flat).max()
    assert np.abs(reps[0].classifier.weight.detach().cpu().numpy() - a["head"]).max() <= 1e-4
    np.testing.assert_allclose(reps[0].backbone._stats.cpu().numpy(), a["rm"], rtol=2e-3, atol=1e-4)


def test_sharded_exchange_matches_the_all_reduce(tmp_path):
    """reduce-scatter -> fused SGD on this rank's shard of the flat buffer -> all-gather: same parameters as the all-reduce path
    (the two collectives sum in a different order: fp32 rounding only), identical on both ranks, half the momentum state per rank"""
    steps = 3
    (tmp_path / "ar").mkdir(); (tmp_path / "rs").mkdir()
    r1 = _launch([os.path.join(ROOT, "tests", "dp_worker.py"), str(tmp_path / "ar"), str(steps)])
    assert r1.returncode == 0, r1.stderr[-3000:]
    r2 = _launch([os.path.join(ROOT, "tests", "dp_worker.py"), str(tmp_path / "rs"), str(steps)], CLHIP_DP_EXCHANGE="reduce_scatter")
    assert r2.returncode == 0, r2.stderr[-3000:]
    ar = np.load(tmp_path / "ar" / "rank0.npz")
    a, b = np.load(tmp_path / "rs" / "rank0.npz"), np.load(tmp_path / "rs" / "rank1.npz")
    np.testing.assert_array_equal(a["flat"], b["flat"])
    np.testing.assert_array_equal(a["head"], b["head"])
    assert np.abs(a["flat"] - ar["flat"]).max() <= 2e-3 * np.abs(ar["flat"]).max()
    assert np.abs(a["head"] - ar["head"]).max() <= 1e-4
    n = int(a["nflat"])
    assert int(ar["momentum_elems"]) == n and n // 2 <= int(a["momentum_elems"]) <= n // 2 + 8


def test_trainer_two_ranks_on_the_gpu_loader(tmp_path):
    """the Trainer itself under data parallelism with the GPU batch loader (no `sampler` attribute; 111 training images in task
    0, 74 + 21 rehearsal exemplars later: neither a multiple of 2): both ranks run the same number of steps, end with identical
    parameters, buffers and learning rates (the patience schedule sees the rank-averaged loss)"""
    r = _launch([os.path.join(ROOT, "tests", "dp_trainer_worker.py"), str(tmp_path)])
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = np.load(tmp_path / "trainer_rank0.npz"), np.load(tmp_path / "trainer_rank1.npz")
    np.testing.assert_array_equal(a["flat"], b["flat"])
    np.testing.assert_array_equal(a["head"], b["head"])
    np.testing.assert_array_equal(a["buffer"], b["buffer"])
    np.testing.assert_array_equal(a["lr"], b["lr"])
    np.testing.assert_array_equal(a["acc"], b["acc"])
    assert len(a["buffer"]) > 0 and np.isfinite(a["flat"]).all()


@pytest.mark.parametrize("workload,batch", [("lwf_resnet18_b50_task0", 64), ("icarl_resnet32_b50_task1", 64), ("ewc_resnet32_b50_task1", 64),
                                            ("l2p_vitb16_b10_task1", 8), ("inflora_vitb16_b20_task1", 8)])
def test_bench_contract_at_two_ranks(workload, batch):
    """every bench workload through the N=2 path: the trainer-side reduce (ResNet methods, InfLoRA_OPT's lora_B bucket) and
    the plugin-side reduce-then-clip of L2P"""
    r = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", workload, "--batch", str(batch),
                 "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 prints ONE json line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 2 * batch and out["config"]["parallelism"] == "dp2"
    assert out["scaling"] == "weak" and out["value"] > 0 and np.isfinite(out["config"]["final_loss"])
    assert abs(out["value"] - 2 * batch * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]
    # the self-checking record of the process group: one entry per rank, the reported time is the maximum over them
    dp = out["dp"]
    assert dp["world_size"] == 2 and dp["backend"] == "gloo" and sorted(r_["rank"] for r_ in dp["ranks"]) == [0, 1]
    assert dp["exchange"] in ("all_reduce", "reduce_scatter")


def test_bench_strong_scaling_and_step_breakdown_at_two_ranks():
    """`--scaling strong`: the GLOBAL batch is fixed and every rank takes batch // N (core/trainer.py:229-241); the `dp` object carries
    the per-rank breakdown of the step (backward, exposed exchange, optimizer, the bucket's all-reduce alone) that makes a SCALE record
    self-diagnosing"""
    r = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "64", "--scaling", "strong", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["scaling"] == "strong" and out["config"]["global_batch"] == 64 and out["config"]["per_gpu_batch"] == 32
    assert abs(out["value"] - 64 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]          # whole-job images per second
    for rk in out["dp"]["ranks"]:
        for key in ("backward_ms", "exchange_exposed_ms", "optimizer_ms", "allreduce_alone_ms", "bucket_mb", "busbw_gbs"):
            assert key in rk and rk[key] > 0, (key, rk)
        assert abs(rk["bucket_mb"] - 44.7) < 0.2                                                      # ResNet-18's flat gradient buffer
    r = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "65", "--scaling", "strong", "--no-cpu-baseline"])
    assert r.returncode != 0 and "does not divide" in r.stderr


def test_eight_ranks_train_in_lockstep(tmp_path):
    """the world-8 path (the node BASELINE configs[2] / [4] name) with eight processes sharing the one GPU over gloo --
    rank-0 broadcast, per-rank batches, the segmented backward with the early tail hand-over, 1/8 folded into the fused SGD step: all eight
    replicas end with bit-identical parameters, per-rank BatchNorm statistics until the end-of-task broadcast"""
    r = _launch([os.path.join(ROOT, "tests", "dp_worker.py"), str(tmp_path), "2"], timeout=900, nproc=8, OMP_NUM_THREADS="1")
    assert r.returncode == 0, r.stderr[-3000:]
    ranks = [np.load(tmp_path / f"rank{k}.npz") for k in range(8)]
    for k in range(1, 8):
        np.testing.assert_array_equal(ranks[0]["flat"], ranks[k]["flat"])
        np.testing.assert_array_equal(ranks[0]["head"], ranks[k]["head"])
        np.testing.assert_array_equal(ranks[0]["rm_synced"], ranks[k]["rm_synced"])
        assert np.abs(ranks[0]["rm"] - ranks[k]["rm"]).max() > 0
    np.testing.assert_array_equal(ranks[0]["rm_synced"], ranks[0]["rm"])
    assert np.isfinite(ranks[0]["flat"]).all()


@pytest.mark.parametrize("exchange", ["all_reduce", "reduce_scatter"])
def test_bench_icarl_strong_scaling_at_eight_ranks(exchange):
    """BASELINE configs[2] as the driver will launch it on an 8-GPU node, strong-scaling series (x[256 global -> 32 / GPU], SURVEY.md section
    8(d), core/trainer.py:229-241): eight ranks (sharing cuda:0 over gloo here), 32 images each, the flat 1.9-MB gradient bucket reduced per
    step -- with the ragged reduce-scatter shards of a 466 256-element buffer (n % 32 = 16) in the second form; ONE json line, whole-job rate,
    one `dp` entry per rank with the step breakdown"""
    r = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "2", "--workload", "icarl_resnet32_b50_task1", "--batch", "256",
                 "--scaling", "strong", "--no-cpu-baseline"], timeout=900, nproc=8, OMP_NUM_THREADS="1", CLHIP_DP_EXCHANGE=exchange)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["config"]["global_batch"] == 256 and out["config"]["per_gpu_batch"] == 32
    assert out["config"]["parallelism"] == "dp8" and np.isfinite(out["config"]["final_loss"])
    assert abs(out["value"] - 256 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]
    dp = out["dp"]
    assert dp["world_size"] == 8 and sorted(r_["rank"] for r_ in dp["ranks"]) == list(range(8)) and dp["exchange"] == exchange
    for rk in dp["ranks"]:
        assert rk["backward_ms"] > 0 and abs(rk["bucket_mb"] - 1.865) < 0.01, rk
