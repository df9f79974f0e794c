"""clhip_gemm_nt against the vendor library (torch.nn.functional.linear -> hipBLASLt / rocBLAS) on the ViT-B/16 GEMM shapes of the two
ViT workloads: python tools/gemm_vs_blas.py [reps]  (C = A @ B^T, bf16, no epilogue; HIP events, rotating operand sets)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from libcontinual_amd import _lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
SHAPES = []
for tag, M in (("inflora b128", 128 * 197), ("l2p b16", 16 * 222)):
    for name, N, K in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
        SHAPES.append((f"{tag} {name} fwd", M, N, K))
    # weight gradients: dW[N,K] = dY^T X  -> "NT" with the reduction over M (operands transposed copies): M' = N, N' = K, K' = M
    if M % 64 == 0:
        SHAPES.append((f"{tag} fc1 dW", 3072, 768, M))
    SHAPES.append((f"{tag} fc2 dX", M, 3072, 768))
SHAPES.append(("square 4096", 4096, 4096, 4096))
SHAPES.append(("square 8192", 8192, 8192, 8192))


def timeit(fn, n):
    for _ in range(3):
        fn(0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, M, N, K in SHAPES:
    nset = max(2, min(8, int(300e6 / ((M * K + N * K + M * N) * 2)) + 1))
    A = [torch.randn(M, K, device=dev).bfloat16() for _ in range(nset)]
    B = [(torch.randn(N, K, device=dev) * 0.03).bfloat16() for _ in range(nset)]
    C = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(nset)]
    ours = timeit(lambda i: _lib.call("clhip_gemm_nt", A[i % nset].data_ptr(), B[i % nset].data_ptr(), C[i % nset].data_ptr(), None, None, None,
                                      M, N, K, K, K, N, N, N, 0, _lib.BF16, st), reps)
    ref = F.linear(A[0], B[0])
    err = float((C[0].float() - ref.float()).abs().max() / (ref.float().abs().max() + 1e-30))
    blas = timeit(lambda i: torch.mm(A[i % nset], B[i % nset].t(), out=C[i % nset]), reps)
    fl = 2.0 * M * N * K
    print(f"{name:22s} M{M:6d} N{N:5d} K{K:6d}  ours {ours*1e3:7.1f} us {fl/ours/1e9:6.0f} TF/s | vendor {blas*1e3:7.1f} us {fl/blas/1e9:6.0f} TF/s | ours/vendor time {ours/blas:5.2f}  maxrel {err:.1e}", flush=True)
