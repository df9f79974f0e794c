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


# ---- the step's FUSED forms against the vendor path for the same operation (what torch runs: addmm / linear through hipBLASLt, then the
#      elementwise kernels of the epilogue).  Same operands, same outputs; the vendor side is given every fusion torch offers (bias inside
#      addmm's epilogue, in-place residual add, one fused-GELU kernel for the activation and one for its derivative).
print("\nfused epilogues (ours: one launch) vs vendor GEMM + torch elementwise", flush=True)
SQRT1_2, INV_SQRT_2PI = 0.7071067811865476, 0.3989422804014327
EPI = {"plain": 0, "bias": 1, "bias+res": 2, "bias+GELU+GELU'": 3, "x H": 4}
FUSED = []
for tag, M in (("inflora b128", 128 * 197), ("l2p b16", 16 * 222)):
    FUSED += [(f"{tag} qkv (+bias)", M, 2304, 768, "bias"), (f"{tag} proj (+bias+res)", M, 768, 768, "bias+res"),
              (f"{tag} fc1 (+bias,GELU,GELU')", M, 3072, 768, "bias+GELU+GELU'"), (f"{tag} fc2 (+bias+res)", M, 768, 3072, "bias+res"),
              (f"{tag} d fc2 (x GELU')", M, 3072, 768, "x H"), (f"{tag} d fc1", M, 768, 3072, "plain"),
              (f"{tag} d proj", M, 768, 768, "plain"), (f"{tag} d qkv", M, 768, 2304, "plain")]
for name, M, N, K, kind in FUSED:
    nset = max(2, min(6, int(300e6 / ((M * K + N * K + 3 * M * N) * 2)) + 1))
    A = [torch.randn(M, K, device=dev).bfloat16() for _ in range(nset)]
    B = [(torch.randn(N, K, device=dev) * 0.03).bfloat16() for _ in range(nset)]
    C = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(nset)]
    bias = torch.randn(N, device=dev) * 0.1
    bias_bf = bias.bfloat16()
    R = [torch.randn(M, N, device=dev).bfloat16() for _ in range(nset)] if kind == "bias+res" else None
    Hb = [torch.rand(M, N, device=dev).bfloat16() for _ in range(nset)] if kind in ("x H", "bias+GELU+GELU'") else None
    ones = torch.ones(M, N, device=dev, dtype=torch.bfloat16) if kind == "bias+GELU+GELU'" else None

    def ours_fn(i):
        j = i % nset
        _lib.call("clhip_gemm_nt", A[j].data_ptr(), B[j].data_ptr(), C[j].data_ptr(), bias.data_ptr() if "bias" in kind else None,
                  R[j].data_ptr() if R else None, Hb[j].data_ptr() if Hb else None, M, N, K, K, K, N, N, N, EPI[kind], _lib.BF16, st)

    def vendor_fn(i):
        j = i % nset
        if kind == "plain":
            torch.mm(A[j], B[j].t(), out=C[j])
        elif kind == "bias":
            torch.addmm(bias_bf, A[j], B[j].t(), out=C[j])
        elif kind == "bias+res":
            torch.addmm(bias_bf, A[j], B[j].t(), out=C[j]); C[j].add_(R[j])
        elif kind == "x H":
            torch.mm(A[j], B[j].t(), out=C[j]); C[j].mul_(Hb[j])
        else:   # pre-activation -> GELU (saved for fc2 and its backward) and GELU' (saved for the backward)
            pre = torch.addmm(bias_bf, A[j], B[j].t())
            torch.ops.aten.gelu.out(pre, approximate="none", out=C[j])
            # d/dx gelu = Phi(x) + x phi(x): torch has no forward kernel for it; autograd's gelu_backward with grad 1 is the one fused kernel that computes it
            torch.ops.aten.gelu_backward.grad_input(ones, pre, approximate="none", grad_input=Hb[j])

    ours = timeit(ours_fn, reps)
    vend = timeit(vendor_fn, reps)
    fl = 2.0 * M * N * K
    print(f"{name:34s} M{M:6d} N{N:5d} K{K:5d}  ours {ours*1e3:7.1f} us {fl/ours/1e9:6.0f} TF/s | vendor path {vend*1e3:7.1f} us {fl/vend/1e9:6.0f} TF/s | ours/vendor time {ours/vend:5.2f}", flush=True)
