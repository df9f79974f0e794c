"""BatchNorm kernels per ResNet-18 layer shape (batch 256, bf16), through the C ABI:  python tools/bn_bench.py [reps]

Rows: the forward apply from the fp64 sums (clhip_bn_apply_train: reads z (+ residual), writes y), the backward pair
(clhip_bn_bwd_acc = reduce + apply, and the zmask form of the residual-free units), and the apply pass alone
(clhip_bn_bwd_apply_acc with ready sums).  Bytes = tensors read + written once; buffers rotate over more than the 256 MB
Infinity Cache when `rot` says so.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from libcontinual_amd import _lib

if os.environ.get("BN_BENCH_LIB"):                                    # A/B against another build of the library
    _lib.LIB_PATH = os.environ["BN_BENCH_LIB"]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev, tdt, code = "cuda", torch.bfloat16, _lib.BF16
st = torch.cuda.current_stream().cuda_stream


def timed(fn, nset):
    for i in range(3):
        fn(i % nset)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i % nset)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def shape_rows(M, C, nset):
    t = lambda: [torch.randn(M, C, device=dev).to(tdt) for _ in range(nset)]
    z, y, dy, dz, res = t(), t(), t(), t(), t()
    g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    mean, invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    rep = 1
    while rep < 32 and _lib.lib().clhip_bn_bwd_blocks(M, C) > 64 * rep:      # plan.hip's choice: ~64 producer workgroups per replica
        rep <<= 1
    acc = torch.zeros(rep, 2, C, device=dev, dtype=torch.float64)
    acc[0, 1] = M
    tb = M * C * 2 / 1e3                                               # KB per tensor
    out = []

    def add(name, ntens, fn):
        us = timed(fn, nset)
        out.append(f"| {M}x{C} rep {rep} {name} | {us:.1f} | {ntens} | {ntens * tb / us / 1e3:.0f} |")

    add("fwd apply relu", 2, lambda i: _lib.call("clhip_bn_apply_train", z[i].data_ptr(), acc.data_ptr(), rep, M, C, g.data_ptr(), b.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                                 0.1, 1e-5, mean.data_ptr(), invstd.data_ptr(), None, y[i].data_ptr(), 1, code, st))
    add("fwd apply relu +res", 3, lambda i: _lib.call("clhip_bn_apply_train", z[i].data_ptr(), acc.data_ptr(), rep, M, C, g.data_ptr(), b.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                                      0.1, 1e-5, mean.data_ptr(), invstd.data_ptr(), res[i].data_ptr(), y[i].data_ptr(), 1, code, st))
    mean.zero_(); invstd.fill_(1.0)
    add("bwd zmask (reduce 2 + apply 3)", 5, lambda i: (acc.zero_(), _lib.call("clhip_bn_bwd_acc_zmask", dy[i].data_ptr(), z[i].data_ptr(), mean.data_ptr(), invstd.data_ptr(), g.data_ptr(),
                                                                              b.data_ptr(), dg.data_ptr(), db.data_ptr(), dz[i].data_ptr(), M, C, acc.data_ptr(), rep, code, st)))
    add("bwd relu +dres (reduce 3 + apply 5)", 8, lambda i: (acc.zero_(), _lib.call("clhip_bn_bwd_acc", dy[i].data_ptr(), y[i].data_ptr(), z[i].data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                                                                   g.data_ptr(), dg.data_ptr(), db.data_ptr(), dz[i].data_ptr(), res[i].data_ptr(), 0, M, C, 1, acc.data_ptr(), rep, code, st)))
    add("bwd apply only, zmask", 3, lambda i: _lib.call("clhip_bn_bwd_apply_acc", dy[i].data_ptr(), None, z[i].data_ptr(), mean.data_ptr(), invstd.data_ptr(), g.data_ptr(), b.data_ptr(),
                                                        dg.data_ptr(), db.data_ptr(), dz[i].data_ptr(), None, 0, M, C, 2, acc.data_ptr(), rep, code, st))
    add("bwd apply only, relu +dres", 5, lambda i: _lib.call("clhip_bn_bwd_apply_acc", dy[i].data_ptr(), y[i].data_ptr(), z[i].data_ptr(), mean.data_ptr(), invstd.data_ptr(), g.data_ptr(),
                                                             None, dg.data_ptr(), db.data_ptr(), dz[i].data_ptr(), res[i].data_ptr(), 0, M, C, 1, acc.data_ptr(), rep, code, st))
    add("memset of the accumulator alone", 0, lambda i: acc.zero_())
    return out


print("| shape, launch | us | tensors moved | TB/s |\n|---|---|---|---|")
for nset, tag in ((1, "same buffers (Infinity-Cache resident where they fit)"), (6, "rotating over 6 buffer sets")):
    print(f"| **{tag}** | | | |")
    for M, C in ((262144, 64), (65536, 128), (16384, 256), (4096, 512)):
        print("\n".join(shape_rows(M, C, nset)))
