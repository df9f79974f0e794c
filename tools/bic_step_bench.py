"""throughput of one BiC stage-1 distillation step (task >= 1: student + bias-corrected teacher forward, KD + CE, backward, fused SGD) on the
pre-activation backbone:  python tools/bic_step_bench.py [batch] [image size] [steps] [dtype]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import libcontinual_amd.model as M
from libcontinual_amd import optim, trainer, utils

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dtype = sys.argv[4] if len(sys.argv) > 4 else "bf16"
dev = torch.device("cuda")
torch.manual_seed(0)
m = M.bic(M.cifar_resnet32_V2(dtype=dtype), 200, device=dev, task_num=10, init_cls_num=20, inc_cls_num=20).to(dev)
m.before_task(0, None, None, None); m.after_task(0, None, None, None)
m.before_task(1, None, None, None)
opt = optim.SGD(m.get_parameters({}), lr=0.1, momentum=0.9, weight_decay=1e-3)
m.train()
batches = [{"image": torch.randn(B, 3, S, S, device=dev), "label": torch.randint(0, 40, (B,), device=dev)} for _ in range(4)]
utils.quiesce_gc()
trainer.train_steps(m, opt, [batches[i % 4] for i in range(5)], device=dev)
torch.cuda.synchronize()
t0 = time.time()
trainer.train_steps(m, opt, [batches[i % 4] for i in range(steps)], device=dev)
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print(f"BiC task-1 step, ResNet_BIC(32) {dtype}, batch {B}, {S}x{S}: {dt * 1e3:.2f} ms/step, {B / dt:.0f} img/s")
