"""conv4 configuration sweep over the four 3x3 / stride-1 layer shapes of ResNet-18 at a given batch:  python tools/conv4_cfg_sweep.py [batch]
(forward through clhip_conv_fwd_acc and dgrad, every instantiated (WM, WN, KG, CK) through the clhip_conv4_set_cfg tuning hook; first column = pick4's choice)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libcontinual_amd import _lib
dev, tdt, code = "cuda", torch.bfloat16, _lib.BF16
st = torch.cuda.current_stream().cuda_stream
L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
def timed(fn, reps=40):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
cfgs = [(0,0,0,0), (2,1,4,64), (4,1,1,32), (2,1,1,32), (2,1,1,64), (2,1,2,32), (2,1,2,64), (2,2,1,32), (2,2,2,32), (4,2,1,64), (1,1,4,32), (1,1,4,64), (2,1,4,32), (2,1,4,64), (1,2,2,32), (1,2,4,32)]
print("batch", B, "cfgs", cfgs)
for (H, C) in ((32, 64), (16, 128), (8, 256), (4, 512)):
    N, K = B, C
    x = torch.randn(N, H, H, C, device=dev).to(tdt); w = (torch.randn(K, 9, C, device=dev) * 0.05).to(tdt)
    z = torch.empty(N, H, H, K, device=dev, dtype=tdt)
    acc = torch.zeros(16, 2, K, device=dev, dtype=torch.float64)
    for mode in (0, 1):
        row = []
        for c in cfgs:
            L.clhip_conv4_set_cfg(*c)
            try:
                if mode == 0:
                    us = timed(lambda: _lib.call("clhip_conv_fwd_acc", x.data_ptr(), w.data_ptr(), z.data_ptr(), acc.data_ptr(), 16, N, H, H, C, K, 3, 1, 1, code, st))
                else:
                    us = timed(lambda: _lib.call("clhip_conv_dgrad", z.data_ptr(), w.data_ptr(), x.data_ptr(), 0, N, H, H, K, C, 3, 1, 1, code, st))
                row.append(f"{us:6.1f}")
            except Exception as e:
                row.append("   n/a")
        L.clhip_conv4_set_cfg(0, 0, 0, 0)
        print(f"{H}x{H}x{C} {'fwd ' if mode == 0 else 'dgrd'} " + " ".join(row))
