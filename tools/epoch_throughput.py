"""images/sec of a REAL Trainer epoch (data pipeline + step) on a CIFAR-sized synthetic dataset, GPU input pipeline vs the CPU
DataLoader path: python tools/epoch_throughput.py [gpu|cpu] [method] [backbone] [batch] [workers]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libcontinual_amd.config import Config
from libcontinual_amd.trainer import Trainer
mode = sys.argv[1] if len(sys.argv) > 1 else "gpu"
method = sys.argv[2] if len(sys.argv) > 2 else "LWF"
backbone = sys.argv[3] if len(sys.argv) > 3 else "resnet18"
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 256
workers = int(sys.argv[5]) if len(sys.argv) > 5 else 0
cfg = Config().get_config_dict()
feat = 512 if backbone == "resnet18" else 64
cfg.update(dict(dataset="synthetic", image_size=32, init_cls_num=50, inc_cls_num=5, task_num=2, epoch=1, init_epoch=3, batch_size=batch,
                val_per_epoch=100, testing_times=1, num_workers=workers, save_path="", synthetic_per_class=500, synthetic_test_per_class=20, seed=1,
                gpu_input_pipeline=(mode == "gpu"),
                backbone={"name": backbone, "kwargs": {"num_classes": 100, "dtype": "bf16", "args": {"dataset": "cifar100"}}},
                classifier={"name": method, "kwargs": {"num_class": 100, "feat_dim": feat, "init_cls_num": 50, "inc_cls_num": 5, "lamda": 1000}},
                optimizer={"name": "SGD", "kwargs": {"lr": 0.05}}, lr_scheduler={"name": "Constant"}))
lines = []
tr = Trainer(0, cfg, log=lambda *a, **k: lines.append(" ".join(str(x) for x in a)))
tr.train_loop()
for l in lines:
    if l.startswith("Epoch"):
        print(mode, l.replace("\t", " "))
