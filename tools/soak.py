"""multi-task soak run of the product Trainer on the synthetic class-structured dataset: accuracy table, throughput and device-memory
high-water mark per task (a leak shows up as monotone growth).  python tools/soak.py [method] [backbone] [tasks] [epochs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libcontinual_amd.config import Config
from libcontinual_amd.trainer import Trainer
method = sys.argv[1] if len(sys.argv) > 1 else "ICarl"
backbone = sys.argv[2] if len(sys.argv) > 2 else "cifar_resnet32"
tasks = int(sys.argv[3]) if len(sys.argv) > 3 else 5
epochs = int(sys.argv[4]) if len(sys.argv) > 4 else 6
feat = 512 if backbone == "resnet18" else 64
init, inc = 20, 10
total = init + inc * (tasks - 1)
kw = {"num_class": total, "feat_dim": feat, "init_cls_num": init, "inc_cls_num": inc, "task_num": tasks, "lamda": 100, "K": 2, "lw_mr": 1, "dist": 0.5}
buf = {"ICarl": {"name": "LinearHerdingBuffer", "kwargs": {"buffer_size": 600, "batch_size": 128}},
       "WA": {"name": "LinearHerdingBuffer", "kwargs": {"buffer_size": 600, "batch_size": 128}},
       "DER": {"name": "LinearBuffer", "kwargs": {"buffer_size": 600, "batch_size": 128, "strategy": "herding"}},
       "LUCIR": {"name": "LinearBuffer", "kwargs": {"buffer_size": 600, "batch_size": 128, "strategy": "herding"}}}.get(
           method, {"name": "LinearBuffer", "kwargs": {"buffer_size": 0, "batch_size": 128, "strategy": "random"}})
if method == "WA":
    kw["init_cls_num"] = init = inc
    total = inc * tasks; kw["num_class"] = total
cfg = Config().get_config_dict()
cfg.update(dict(dataset="synthetic", image_size=32, init_cls_num=init, inc_cls_num=inc, task_num=tasks, epoch=epochs, init_epoch=epochs, batch_size=128,
                val_per_epoch=1000, testing_times=1, num_workers=0, save_path="", synthetic_per_class=300, synthetic_test_per_class=30, seed=3,
                backbone={"name": backbone, "kwargs": {"num_classes": total, "dtype": "bf16", "args": {"dataset": "cifar100"}}},
                classifier={"name": method, "kwargs": kw}, buffer=buf,
                optimizer={"name": "SGD", "kwargs": {"lr": 0.05, "momentum": 0.9, "weight_decay": 5e-4}},
                lr_scheduler={"name": "MultiStepLR", "kwargs": {"milestones": [4], "gamma": 0.2}}))
mem = []
def log(*a, **k):
    s = " ".join(str(x) for x in a)
    if "Result of Task" in s:
        mem.append((torch.cuda.memory_allocated() / 2**20, torch.cuda.max_memory_allocated() / 2**20))
t0 = time.time()
tr = Trainer(0, cfg, log=log)
out = tr.train_loop()
print(method, backbone, "wall %.1f s" % (time.time() - t0))
print(np.round(out["acc_table"], 1))
print("allocated / peak MiB per task:", [(round(a), round(b)) for a, b in mem])
print("finite:", bool(np.isfinite(out["acc_table"]).all()), "last avg acc %.1f" % out["batch_last_acc"])
