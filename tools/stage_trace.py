"""(numbers are hundreds of s_memtime ticks = shader-clock cycles / 100: divide by ~24 for microseconds)
phase times inside the stage-level training launches (stage_train.hip): workgroup 0's s_memtime stamps of one convolution per geometry.
    python tools/stage_trace.py [batch]"""
import ctypes
import sys

import torch

import libcontinual_amd.model as M
from libcontinual_amd import _lib

FWD = ["MFMA loop", "z store + sums", "(sync) publish + reduce role", "wait totals", "coefficients", "epilogue (+ y, mask)"]
BWD = ["A: g + sums", "red / vals", "publish + reduce role", "weight gradient of the unit above", "input image -> LDS, filters", "wait totals", "ctab + dz -> LDS", "dgrad MFMA"]

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = _lib.lib()
torch.manual_seed(0)
bb = M.cifar_resnet32(dtype="bf16").to("cuda")
bb.train()
x = torch.randn(batch, 3, 32, 32, device="cuda")
for C, cv in ((16, 4), (16, 5), (32, 4), (32, 5), (64, 4), (64, 5)):
    L.clhip_config(b"STAGE_TRACE", f"{C}:{cv}".encode())
    for _ in range(3):
        for p in bb.parameters():
            p.grad = None
        f = bb(x)["features"]
        f.sum().backward()
    torch.cuda.synchronize()
    plan = list(bb._handle.plans.values())[0][0]
    out = (ctypes.c_ulonglong * 24)()
    L.clhip_plan_stage_trace(plan, out)
    t = [int(v) for v in out]
    print(f"C {C} convolution {cv} ({'second' if cv & 1 else 'first'} of its block), batch {batch}")
    print("   forward : " + ", ".join(f"{n} {(t[i + 1] - t[i]) / 100:.2f}" for i, n in enumerate(FWD)) + f"  = {(t[6] - t[0]) / 100:.2f} us")
    print("   backward: " + ", ".join(f"{n} {(t[9 + i] - t[8 + i]) / 100:.2f}" for i, n in enumerate(BWD)) + f"  = {(t[16] - t[8]) / 100:.2f} us")
    if C != 16:
        GRP = ["fetch wait + LDS store", "MFMA", "slab"]
        print(f"   group fetch issue {(t[22] - t[21]) / 100:.2f}")
        print("   group wgrad: publish->start " + f"{(t[17] - t[11]) / 100:.2f}, " + ", ".join(f"{n} {(t[18 + i] - t[17 + i]) / 100:.2f}" for i, n in enumerate(GRP)))
L.clhip_config(b"STAGE_TRACE", None)
