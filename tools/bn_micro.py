"""stand-alone BatchNorm launches at ResNet-18 layer1's shape ([256 x 32 x 32] x 64, bf16): forward apply (train), backward (reduce + apply), with fresh and warm caches
   python tools/bn_micro.py [C=64] [HW=32] [N=256]"""
import ctypes as C
import sys
import torch
from libcontinual_amd import _lib

L = _lib.lib()
Cc = int(sys.argv[1]) if len(sys.argv) > 1 else 64
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = int(sys.argv[3]) if len(sys.argv) > 3 else 256
M = N * HW * HW
dev = "cuda"
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
z = torch.randn(M, Cc, device=dev).to(torch.bfloat16)
y = torch.relu(z)
dy = torch.randn(M, Cc, device=dev).to(torch.bfloat16)
dz = torch.empty_like(z)
dres = torch.empty_like(z)
mean, invstd, gamma, beta = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev), torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
dg, db = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
REP = 8
acc = torch.zeros(REP * 2 * Cc, device=dev, dtype=torch.float64)
flush = torch.empty(512 * 1024 * 1024 // 4, device=dev)            # > L2 + Infinity Cache


def t_us(fn, iters, cold):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(iters):
        if cold:
            flush.add_(1.0)
        acc.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) * 1e3
    return tot / iters


def bwd_res():      # block output: mask from y, residual gradient written
    assert L.clhip_bn_bwd_acc(p(dy), p(y), p(z), p(mean), p(invstd), p(gamma), p(dg), p(db), p(dz), p(dres), C.c_int(0), C.c_int64(M), C.c_int(Cc), C.c_int(1), p(acc), C.c_int(REP),
                              C.c_int(0), st) == 0


def bwd_z():        # first convolution of a block: mask from z
    assert L.clhip_bn_bwd_acc_zmask(p(dy), p(z), p(mean), p(invstd), p(gamma), p(beta), p(dg), p(db), p(dz), C.c_int64(M), C.c_int(Cc), p(acc), C.c_int(REP), C.c_int(0), st) == 0


mb = M * Cc * 2 / 1e6
for name, fn, nb in (("bn_bwd (mask from y, + residual gradient): reads dy, y, z twice-ish, writes dz, dres", bwd_res, 7 * mb), ("bn_bwd (mask from z): dy, z twice, dz", bwd_z, 5 * mb)):
    for cold in (False, True):
        try:
            us = t_us(fn, 20, cold)
            print(f"{name} [{M} x {Cc}] {'cold' if cold else 'warm'}: {us:.1f} us for both launches, {nb / us * 1e3:.0f} GB/s")
        except Exception as e:
            print(name, "failed:", e)
            break
