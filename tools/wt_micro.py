"""Stand-alone timing of the write-through lazy BatchNorm input (clhip_conv_fwd_acc_bn_input_wt) against the two launches it replaces
(clhip_bn_apply_train[_mask] + clhip_conv_fwd_acc), HIP events around 50 launches each:  python tools/wt_micro.py [N=256]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libcontinual_amd import _lib  # noqa: E402
from libcontinual_amd._lib import call  # noqa: E402


class BnInput(C.Structure):
    _fields_ = [("stat_acc", C.c_void_p), ("replicas", C.c_int), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("running_mean", C.c_void_p),
                ("running_var", C.c_void_p), ("momentum", C.c_float), ("eps", C.c_float), ("mean", C.c_void_p), ("invstd", C.c_void_p), ("coef", C.c_void_p)]


class BnRes(C.Structure):
    _fields_ = [("res", C.c_void_p), ("y", C.c_void_p), ("relu_mask", C.c_void_p)]


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = "cuda:0"
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    code = _lib.BF16
    print("| shape | conv alone | apply + conv | fused | apply (+res, mask) + conv | fused (+res, mask) |")
    print("|---|---|---|---|---|---|")
    for H, Cc in ((32, 64), (16, 128), (8, 256), (4, 512)):
        W, K = H, Cc
        if not L.clhip_conv_bn_input_wt_supported(N, H, W, Cc, K, 3, 1, 1, code):
            print(f"| {N}x{H}x{W}x{Cc} | unsupported |")
            continue
        M = N * H * W
        bufs = []
        for _ in range(3):      # rotate over buffer sets (no Infinity-Cache residency across launches of one variant)
            z = torch.randn(N, H, W, Cc, device=dev).to(torch.bfloat16)
            r = torch.randn(N, H, W, Cc, device=dev).to(torch.bfloat16)
            y = torch.empty(N, H, W, Cc, device=dev, dtype=torch.bfloat16)
            zo = torch.empty(N, H, W, K, device=dev, dtype=torch.bfloat16)
            mask = torch.empty(M * Cc // 8, dtype=torch.uint8, device=dev)
            bufs.append((z, r, y, zo, mask))
        w = (torch.randn(K, 9, Cc, device=dev) * 0.05).to(torch.bfloat16)
        zf = bufs[0][0].float().reshape(-1, Cc).double()
        acc_in = torch.zeros(4, 2, Cc, dtype=torch.float64, device=dev)
        acc_in[0, 0], acc_in[0, 1] = zf.sum(0), (zf * zf).sum(0)
        gamma, beta = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
        rm, rv, mean, invstd, coef = (torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev), torch.empty(Cc, device=dev), torch.empty(Cc, device=dev),
                                      torch.empty(2, Cc, device=dev))
        acc = torch.zeros(8, 2, K, dtype=torch.float64, device=dev)
        it = [0]

        def nxt():
            it[0] = (it[0] + 1) % 3
            return bufs[it[0]]

        def conv_only():
            z, r, y, zo, mask = nxt()
            call("clhip_conv_fwd_acc", y.data_ptr(), w.data_ptr(), zo.data_ptr(), acc.data_ptr(), 8, N, H, W, Cc, K, 3, 1, 1, code, st)

        def sep(res):
            z, r, y, zo, mask = nxt()
            if res:
                call("clhip_bn_apply_train_mask", z.data_ptr(), acc_in.data_ptr(), 4, M, Cc, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5, mean.data_ptr(),
                     invstd.data_ptr(), r.data_ptr(), y.data_ptr(), mask.data_ptr(), code, st)
            else:
                call("clhip_bn_apply_train", z.data_ptr(), acc_in.data_ptr(), 4, M, Cc, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5, mean.data_ptr(),
                     invstd.data_ptr(), None, y.data_ptr(), 1, code, st)
            call("clhip_conv_fwd_acc", y.data_ptr(), w.data_ptr(), zo.data_ptr(), acc.data_ptr(), 8, N, H, W, Cc, K, 3, 1, 1, code, st)

        def fused(res):
            z, r, y, zo, mask = nxt()
            bi = BnInput(acc_in.data_ptr(), 4, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5, mean.data_ptr(), invstd.data_ptr(), coef.data_ptr())
            br = BnRes(r.data_ptr() if res else None, y.data_ptr(), mask.data_ptr() if res else None)
            call("clhip_conv_fwd_acc_bn_input_wt", z.data_ptr(), C.byref(bi), C.byref(br), w.data_ptr(), zo.data_ptr(), acc.data_ptr(), 8, N, H, W, Cc, K, 3, 1, 1, code, st)
        t = [timeit(conv_only), timeit(lambda: sep(False)), timeit(lambda: fused(False)), timeit(lambda: sep(True)), timeit(lambda: fused(True))]
        print(f"| {N}x{H}x{W}x{Cc}->{K} | " + " | ".join(f"{v:.1f} us" for v in t) + " |")


if __name__ == "__main__":
    main()
