import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_trainer_gpu import cfg_for
from libcontinual_amd.trainer import Trainer
method, backbone, dtype = sys.argv[1], sys.argv[2], sys.argv[3]
over = {}
if len(sys.argv) > 4:
    over = eval(sys.argv[4])
tr = Trainer(0, cfg_for(method, backbone, dtype, **over))
out = tr.train_loop()
print(out["acc_table"])
