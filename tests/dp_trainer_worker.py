"""Rank body for tests/test_dp_two_ranks_gpu.py::test_trainer_two_ranks_on_the_gpu_loader (launched through torch.distributed.run).
The product Trainer with n_gpu = 2 on the resident-store GPU input pipeline: per-rank shards of every permutation (padded to
the same length on every rank), per-step gradient all-reduce, rank-0 BatchNorm statistics before after_task, herding on every
rank -- on a dataset whose sizes are NOT multiples of the world size once the rehearsal buffer is merged in."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libcontinual_amd.config import Config             # noqa: E402
from libcontinual_amd.trainer import Trainer           # noqa: E402


def main(out_dir):
    cfg = Config().get_config_dict()
    kw = {"num_class": 7, "feat_dim": 64, "init_cls_num": 3, "inc_cls_num": 2, "task_num": 3}
    cfg.update(dict(dataset="synthetic", image_size=32, init_cls_num=3, inc_cls_num=2, task_num=3, epoch=2, init_epoch=2, batch_size=32,
                    val_per_epoch=10, testing_times=1, num_workers=0, save_path="", synthetic_per_class=37, synthetic_test_per_class=9, seed=11, n_gpu=2,
                    backbone={"name": "cifar_resnet32", "kwargs": {"num_classes": 7, "dtype": "f32", "args": {"dataset": "cifar100"}}},
                    classifier={"name": "ICarl", "kwargs": kw},
                    buffer={"name": "LinearHerdingBuffer", "kwargs": {"buffer_size": 21, "batch_size": 32}},
                    optimizer={"name": "SGD", "kwargs": {"lr": 0.02, "momentum": 0.9, "weight_decay": 5e-4}},
                    lr_scheduler={"name": "PatienceSchedule", "kwargs": {"patience": 1, "factor": 2, "stopping_lr": 1e-6}}))
    tr = Trainer(int(os.environ.get("RANK", 0)), cfg, log=lambda *a, **k: None)
    out = tr.train_loop()
    torch.cuda.synchronize()
    rank = dist.get_rank()
    flat = tr.model.network.backbone.flat_parameters()[0]
    np.savez(os.path.join(out_dir, f"trainer_rank{rank}.npz"), flat=flat.cpu().numpy(), head=tr.model.network.classifier.weight.detach().cpu().numpy(),
             acc=np.asarray(out["acc_table"]), buffer=np.asarray([int(v) for v in tr.buffer.labels]), lr=np.asarray([tr.optimizer.param_groups[0]["lr"]]),
             steps=np.asarray([e[2] for e in tr.hook_trace if e[0] == "train_epoch"]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
