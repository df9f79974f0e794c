"""Data parallelism over a REAL RCCL group: one process per GPU, backend "nccl" (= RCCL over xGMI).  Needs two visible GPUs and is
skipped on the 1-GPU boxes (tests/test_dp_two_ranks_gpu.py runs the same code path there with two ranks sharing the GPU over gloo,
tests/test_dp_gpu.py a 1-rank RCCL group).  What only a real ring can show: the asynchronous all-reduce of the flat gradient buffer's
tail running on RCCL's stream WHILE the rest of the backward runs on the compute stream (GradientReducer.overlap), the reduce-scatter /
all-gather exchange between two devices, and bench.py's `dp` breakdown with real link rates."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the driver's multi-GPU node)")]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(script_args, timeout=900, **extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", **extra)
    env.pop("CLHIP_DIST_BACKEND", None)
    env.pop("CLHIP_SHARED_GPU", None)
    def cmd_with(port):
        return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    # the port is free when it is picked, not necessarily when torchrun's store binds it a second later (an ephemeral port can be handed to another
    # connection in between: one EADDRINUSE in ~800 launches on the GPU boxes) -- retry with a fresh one
    for attempt in range(4):
        r = subprocess.run(cmd_with(_free_port()), cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
        if r.returncode == 0 or "EADDRINUSE" not in r.stderr:
            break
    return r


@pytest.mark.parametrize("exchange", ["all_reduce", "reduce_scatter"])
def test_two_gpus_train_in_lockstep_over_rccl(tmp_path, exchange):
    steps = 4
    r = _launch([os.path.join(ROOT, "tests", "dp_worker.py"), str(tmp_path), str(steps)], CLHIP_DP_EXCHANGE=exchange)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(a["flat"], b["flat"])                  # same reduced gradient, same fused step: bit-identical replicas
    np.testing.assert_array_equal(a["head"], b["head"])
    assert np.abs(a["rm"] - b["rm"]).max() > 0                           # per-rank BatchNorm statistics until the broadcast
    np.testing.assert_array_equal(a["rm_synced"], b["rm_synced"])
    # the single-process emulation (two replicas, gradients averaged by hand)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dp_worker as W
    from libcontinual_amd import optim
    torch.cuda.set_device(0)
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
    assert np.abs(flat - a["flat"]).max() <= 2e-3 * np.abs(flat).max()
    if exchange == "reduce_scatter":
        n = int(a["nflat"])
        assert n // 2 <= int(a["momentum_elems"]) <= n // 2 + 8            # optimizer state for this rank's shard only


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_over_rccl_reports_the_overlap(scaling):
    """bench.py --gpus 2 on two devices: backend nccl, and the exposed part of the exchange (what the compute stream still waits for after
    the backward) is shorter than the bucket's all-reduce alone -- the tail handed over early ran under the backward"""
    r = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "5", "--scaling", scaling, "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    dp = out["dp"]
    assert dp["backend"] == "nccl" and dp["world_size"] == 2 and out["scaling"] == scaling
    assert out["config"]["global_batch"] == (512 if scaling == "weak" else 256)
    assert len({rk["pci"] for rk in dp["ranks"]}) == 2 or all(rk["pci"] is None for rk in dp["ranks"])      # two different devices
    for rk in dp["ranks"]:
        print(scaling, rk)
        assert rk["exchange_exposed_ms"] < rk["allreduce_alone_ms"], rk
        assert rk["busbw_gbs"] > 5.0, rk                                   # a device-to-device ring, not a host bounce
