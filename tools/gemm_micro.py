"""micro-benchmark of clhip_gemm_nt for PMC / kernel-trace passes: python tools/gemm_micro.py M N K epi reps [dtype]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libcontinual_amd import _lib
M, N, K, epi, reps = map(int, sys.argv[1:6])
dt = sys.argv[6] if len(sys.argv) > 6 else "bf16"
td, code = (torch.bfloat16, _lib.BF16) if dt == "bf16" else (torch.float32, _lib.F32)
dev = "cuda"
A = torch.randn(M, K, device=dev).to(td)
B = (torch.randn(N, K, device=dev) * 0.03).to(td)
C = torch.empty(M, N, device=dev, dtype=td)
bias = torch.zeros(N, device=dev)
R = torch.randn(M, N, device=dev).to(td)
H = torch.randn(M, N, device=dev).to(td)
st = torch.cuda.current_stream().cuda_stream
def run():
    _lib.call("clhip_gemm_nt", A.data_ptr(), B.data_ptr(), C.data_ptr(), bias.data_ptr() if epi in (1, 2, 3) else None, R.data_ptr() if epi == 2 else None,
              H.data_ptr() if epi in (3, 4) else None, M, N, K, K, K, N, N, N, epi, code, st)
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"gemm M{M} N{N} K{K} epi{epi} {dt}: {ms*1e3:.1f} us  {2.0*M*N*K/ms/1e9:.0f} TFLOP/s")
