"""stand-alone LayerNorm forward / backward launches at the ViT-B/16 shapes (rows x 768, bf16):  python tools/ln_micro.py [rows=25216] [iters=200] [lib.so]"""
import ctypes as C
import sys
import torch

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 25216
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
if len(sys.argv) > 3:
    L = C.CDLL(sys.argv[3])
else:
    from libcontinual_amd import _lib
    L = _lib.lib()
D = 768
dev = "cuda"
x = torch.randn(rows, D, device=dev).to(torch.bfloat16)
dy = torch.randn(rows, D, device=dev).to(torch.bfloat16)
g = torch.zeros(rows, D, device=dev, dtype=torch.bfloat16)
y = torch.empty_like(x)
gamma, beta = torch.ones(D, device=dev), torch.zeros(D, device=dev)
mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
fwd = lambda: L.clhip_ln_fwd(p(x), p(gamma), p(beta), p(y), p(mean), p(rstd), C.c_int(rows), C.c_int(D), C.c_float(1e-6), C.c_int(0), st)
bwd = lambda: L.clhip_ln_bwd(p(dy), p(x), p(gamma), p(mean), p(rstd), p(g), C.c_int(rows), C.c_int(D), C.c_int(0), st)
for name, fn, nbytes in (("ln_fwd", fwd, 2 * rows * D * 2), ("ln_bwd", bwd, 4 * rows * D * 2)):
    for _ in range(5):
        assert fn() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{name} {rows} x {D}: {us:.1f} us, {nbytes / us * 1e-3:.0f} GB/s")
