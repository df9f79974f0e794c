"""Rank body for tests/test_dp_two_ranks_gpu.py (launched through torch.distributed.run; not a test module itself).
Every rank trains the same LwF / ResNet-18 model for a few steps on ITS OWN batches through the product's data-parallel path
(broadcast, segmented backward + overlapped all-reduce, 1/world folded into the fused SGD step) and dumps the result."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libcontinual_amd.model as M                     # noqa: E402
from libcontinual_amd import optim, parallel           # noqa: E402
from libcontinual_amd.trainer import train_steps       # noqa: E402


def make(seed):
    torch.manual_seed(seed)
    bb = M.resnet18(args={"dataset": "cifar100"}, dtype="f32")
    m = M.LWF(bb, 512, 100, device="cuda", init_cls_num=50, inc_cls_num=5).to("cuda")
    m.before_task(0, None, None, None)
    return m


def batch(seed, B=16):
    g = torch.Generator().manual_seed(seed)
    return {"image": torch.randn(B, 3, 32, 32, generator=g).cuda(), "label": torch.randint(0, 50, (B,), generator=g).cuda()}


def main(out_dir, steps):
    # every rank on cuda:0 under the shared-GPU test hook (gloo); one GPU per rank otherwise (RCCL: tests/test_dp_rccl_gpu.py)
    torch.cuda.set_device(0 if os.environ.get("CLHIP_SHARED_GPU") else int(os.environ.get("LOCAL_RANK", "0")))
    rank, world = parallel.init_distributed(True)
    m = make(100 + rank)                               # different initial weights per rank: the broadcast must fix that
    parallel.broadcast_module_state(m)
    opt = optim.SGD(m.get_parameters({}), lr=0.05, momentum=0.9, weight_decay=5e-4)
    red = parallel.GradientReducer()                   # CLHIP_DP_EXCHANGE picks the exchange (all_reduce | reduce_scatter)
    parallel.attach(m, opt, red)
    m.train()
    train_steps(m, opt, [batch(1000 * rank + i) for i in range(steps)], red, "LWF", None, "cuda")
    torch.cuda.synchronize()
    flat = m.backbone.flat_parameters()[0]
    rm = m.backbone._stats.cpu().numpy()
    parallel.broadcast_module_state(m)                 # what the Trainer does before after_task: rank 0's running statistics everywhere
    st = opt.state[m.backbone._params[0]]
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), flat=flat.cpu().numpy(), head=m.classifier.weight.detach().cpu().numpy(),
             rm=rm, rm_synced=m.backbone._stats.cpu().numpy(), momentum_elems=np.array(sum(v.numel() for k, v in st.items() if k.startswith("flat_momentum"))),
             nflat=np.array(flat.numel()),
             head_grad_in_bucket=np.array(int(red._rest.get("flat") is not None and m.classifier.weight.grad is not None and
                                              red._rest["flat"].data_ptr() <= m.classifier.weight.grad.data_ptr() < red._rest["flat"].data_ptr() + 4 * red._rest["flat"].numel())))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
