"""micro-benchmark of clhip_attn_fwd / bwd: python tools/attn_micro.py B N H reps"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libcontinual_amd import _lib
B, N, H, reps = map(int, sys.argv[1:5])
D = 64 * H
dev = "cuda"
qkv = torch.randn(B * N, 3 * D, device=dev).bfloat16()
dout = torch.randn(B * N, D, device=dev).bfloat16()
out = torch.empty(B * N, D, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, H, N, device=dev)
dqkv = torch.empty_like(qkv)
dsum = torch.empty(B, H, N, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(reps):
    _lib.call("clhip_attn_fwd", qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, N, H, D, _lib.BF16, st)
    _lib.call("clhip_attn_bwd", qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), dsum.data_ptr(), B, N, H, D, _lib.BF16, st)
torch.cuda.synchronize()
print("done")
