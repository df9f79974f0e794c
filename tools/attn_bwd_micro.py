"""stand-alone launches of the attention backward at the InfLoRA shape (B x 197 tokens x 12 heads x 64): timing, or a PMC target
   python tools/attn_bwd_micro.py [B=128] [iters=50]        (CLHIP_ATTN_BWD=1 / 2 picks the kernel)"""
import ctypes as C
import sys
import torch
from libcontinual_amd import _lib
from libcontinual_amd._lib import call

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
N, H, D = 197, 12, 768
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
qkv = (torch.randn(B * N, 3 * D, device=dev, generator=g) * 1.5).to(torch.bfloat16)
dout = torch.randn(B * N, D, device=dev, generator=g).to(torch.bfloat16)
out = torch.empty(B * N, D, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, H, N, device=dev)
dqkv = torch.empty(B * N, 3 * D, device=dev, dtype=torch.bfloat16)
dsum = torch.empty(B, H, N, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
call("clhip_attn_fwd", p(qkv), p(out), p(lse), B, N, H, D, _lib.BF16, st)
for _ in range(3):
    call("clhip_attn_bwd", p(qkv), p(out), p(lse), p(dout), p(dqkv), p(dsum), B, N, H, D, _lib.BF16, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    call("clhip_attn_bwd", p(qkv), p(out), p(lse), p(dout), p(dqkv), p(dsum), B, N, H, D, _lib.BF16, st)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / iters
flops = 7 * 2.0 * N * N * 64 * B * H          # the kernel's own MFMA work: S and dP twice, dQ, dK, dV
print(f"attn_bwd B={B}: {us:.1f} us per launch, {flops / us * 1e-6:.0f} TFLOP/s issued, {8 * B * N * D * 2 / us * 1e-3:.0f} GB/s algorithmic")
