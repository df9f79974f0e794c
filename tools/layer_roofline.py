"""Per-shape roofline table of the hot kernels (markdown on stdout):  python tools/layer_roofline.py [reps]

Every distinct convolution of the two CIFAR backbones the bench lines use (ResNet-18 with the CIFAR stem and CifarResNet-32, batch
256, bf16) in its three directions, and every GEMM shape of a ViT-B/16 block at the InfLoRA_OPT (128 x 197 rows) and L2P (16 x 222 rows)
batch sizes, timed through the C ABI with HIP events.  For each launch: duration, achieved TFLOP/s against the dense bf16 MFMA peak
(2.5 PFLOP/s), algorithmic bytes (operands read once + result written once) against 8 TB/s of HBM, and the larger of the two
fractions = how close the launch is to ITS roofline.  Back-to-back repetitions keep operands in the 256 MB Infinity Cache, so the
byte column is an upper bound on what HBM saw.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from libcontinual_amd import _lib

PEAK_TF, PEAK_GB = 2500.0, 8000.0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev, tdt, code = "cuda", torch.bfloat16, _lib.BF16
st = torch.cuda.current_stream().cuda_stream


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3          # us


def row(name, us, flops, nbytes):
    tf, gb = flops / us / 1e6, nbytes / us / 1e3
    f_c, f_m = tf / PEAK_TF, gb / PEAK_GB
    bound = "mfma" if flops / (PEAK_TF * 1e12) >= nbytes / (PEAK_GB * 1e9) else "hbm"
    print(f"| {name} | {us:.1f} | {tf:.0f} | {f_c:.2f} | {gb:.0f} | {f_m:.2f} | {bound} | {max(f_c, f_m):.2f} |")


def conv_rows(tag, N, H, W, C, K, ks, stride, count):
    pad = 1 if ks == 3 else 0
    Cp = max(8, C)
    if Cp & (Cp - 1):
        Cp = 1 << Cp.bit_length()
    Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    x = torch.randn(N, H, W, Cp, device=dev).to(tdt)
    w = (torch.randn(K, ks * ks, Cp, device=dev) * 0.05).to(tdt)
    wd = (torch.randn(Cp, ks * ks, K, device=dev) * 0.05).to(tdt)
    z = torch.randn(N, Ho, Wo, K, device=dev).to(tdt)
    dx = torch.empty(N, H, W, Cp, device=dev, dtype=tdt)
    dw = torch.zeros(K, ks * ks, C, device=dev)
    lib = _lib.lib()
    part = torch.empty(lib.clhip_conv_fwd_tiles(N, H, W, Cp, K, ks, stride, pad), 2, K, device=dev)
    wsb = lib.clhip_conv_wgrad_ws_bytes(N, H, W, Cp, C, K, ks, stride, pad, code)
    wsbuf = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    fl = 2.0 * N * Ho * Wo * ks * ks * C * K
    act_in, act_out, wb = x.numel() * 2, z.numel() * 2, w.numel() * 2
    if ks == 1 and stride == 2:
        act_in //= 4                                                # only every other row and column is touched
    shape = f"{tag} {C}->{K} k{ks} s{stride} {H}x{W} (x{count})"
    acc = torch.zeros(16, 2, K, device=dev, dtype=torch.float64)    # the entry point the training step uses: sums into fp64 accumulators
    us = timed(lambda: _lib.call("clhip_conv_fwd_acc", x.data_ptr(), w.data_ptr(), z.data_ptr(), acc.data_ptr(), 16, N, H, W, Cp, K, ks, stride, pad, code, st))
    row(shape + " fwd", us, fl, act_in + act_out + wb)
    if C >= 16:
        # (the shortcut's input gradient is the FIRST writer of dx in the reverse sweep: it writes all of dx, zeros on the untouched pixels)
        us = timed(lambda: _lib.call("clhip_conv_dgrad", z.data_ptr(), wd.data_ptr(), dx.data_ptr(), 0, N, H, W, Cp, K, ks, stride, pad, code, st))
        row(shape + " dgrad", us, fl, dx.numel() * 2 + act_out + wb)
    us = timed(lambda: _lib.call("clhip_conv_wgrad", x.data_ptr(), z.data_ptr(), dw.data_ptr(), wsbuf.data_ptr() if wsb else None, N, H, W, Cp, C, K, ks,
                                 stride, pad, code, st))
    row(shape + " wgrad", us, fl, act_in + act_out + dw.numel() * 4)


def pair_row(tag, N, H, W, C, K):
    """the step's launch for a down-sampling block entry: dgrad of the 3x3/s2 conv + dgrad of the 1x1/s2 shortcut, one launch (conv6.hip)"""
    Ho, Wo = H // 2, W // 2
    dz = torch.randn(N, Ho, Wo, K, device=dev).to(tdt)
    dzs = torch.randn(N, Ho, Wo, K, device=dev).to(tdt)
    w3 = (torch.randn(C, 9, K, device=dev) * 0.05).to(tdt)
    w1 = (torch.randn(C, 1, K, device=dev) * 0.05).to(tdt)
    dx = torch.empty(N, H, W, C, device=dev, dtype=tdt)
    pk = torch.empty(_lib.lib().clhip_conv_dgrad_pair_packed_bytes(C, K), dtype=torch.uint8, device=dev)
    _lib.call("clhip_conv_dgrad_pair_pack", w3.data_ptr(), w1.data_ptr(), pk.data_ptr(), C, K, code, st)
    us = timed(lambda: _lib.call("clhip_conv_dgrad_pair", dz.data_ptr(), pk.data_ptr(), dzs.data_ptr(), dx.data_ptr(), 0, N, H, W, C, K, code, st))
    row(f"{tag} {C}->{K} {H}x{W}: dgrad of k3 s2 + shortcut k1 s2, ONE launch (the step's form)", us, 2.0 * N * Ho * Wo * K * C * 10,
        (dz.numel() + dzs.numel() + dx.numel()) * 2 + pk.numel())


def entry_rows(tag, N, H, W, C, K):
    """the step's three launches for a FEW-channel down-sampling entry (conv7.hip): forward pair, input-gradient pair, weight-gradient pair"""
    Ho, Wo = H // 2, W // 2
    x = torch.randn(N, H, W, C, device=dev).to(tdt)
    w3f = (torch.randn(K, 9, C, device=dev) * 0.05).to(tdt)
    w1f = (torch.randn(K, 1, C, device=dev) * 0.05).to(tdt)
    z3 = torch.empty(N, Ho, Wo, K, device=dev, dtype=tdt)
    z1 = torch.empty_like(z3)
    a3 = torch.zeros(8, 2, K, dtype=torch.float64, device=dev)
    a1 = torch.zeros(8, 2, K, dtype=torch.float64, device=dev)
    act = (x.numel() + 2 * z3.numel()) * 2
    us = timed(lambda: _lib.call("clhip_conv_fwd_acc_pair", x.data_ptr(), w3f.data_ptr(), w1f.data_ptr(), z3.data_ptr(), z1.data_ptr(), a3.data_ptr(), 8, a1.data_ptr(), 8,
                                 N, H, W, C, K, code, st))
    row(f"{tag} {C}->{K} {H}x{W}: fwd of k3 s2 + shortcut k1 s2 with both BatchNorm sums, ONE launch (the step's form)", us, 2.0 * N * Ho * Wo * K * C * 10, act)
    pair_row(tag, N, H, W, C, K)
    dz = torch.randn(N, Ho, Wo, K, device=dev).to(tdt)
    dzs = torch.randn(N, Ho, Wo, K, device=dev).to(tdt)
    dw3 = torch.zeros(K, 9, C, device=dev)
    dw1 = torch.zeros(K, 1, C, device=dev)
    L = _lib.lib()
    ws3 = torch.empty(L.clhip_conv_wgrad_ws_bytes(N, H, W, C, C, K, 3, 2, 1, code), dtype=torch.uint8, device=dev)
    ws1 = torch.empty(L.clhip_conv_wgrad_ws_bytes(N, H, W, C, C, K, 1, 2, 0, code), dtype=torch.uint8, device=dev)
    us = timed(lambda: _lib.call("clhip_conv_wgrad_pair", x.data_ptr(), dz.data_ptr(), dzs.data_ptr(), dw3.data_ptr(), dw1.data_ptr(), ws3.data_ptr(), ws1.data_ptr(),
                                 N, H, W, C, K, code, st))
    row(f"{tag} {C}->{K} {H}x{W}: wgrad of k3 s2 + shortcut k1 s2, ONE launch + the two reduces (the step's form)", us, 2.0 * N * Ho * Wo * K * C * 10,
        act + (dw3.numel() + dw1.numel()) * 4)


def gemm_row(tag, M, N, K, epi):
    A = torch.randn(M, K, device=dev).to(tdt)
    B = (torch.randn(N, K, device=dev) * 0.03).to(tdt)
    Cm = torch.empty(M, N, device=dev, dtype=tdt)
    bias = torch.zeros(N, device=dev)
    R = torch.randn(M, N, device=dev).to(tdt)
    Hh = torch.randn(M, N, device=dev).to(tdt)
    us = timed(lambda: _lib.call("clhip_gemm_nt", A.data_ptr(), B.data_ptr(), Cm.data_ptr(), bias.data_ptr() if epi in (1, 2, 3) else None,
                                 R.data_ptr() if epi == 2 else None, Hh.data_ptr() if epi in (3, 4) else None, M, N, K, K, K, N, N, N, epi, code, st))
    extra = {0: 0, 1: 0, 2: M * N * 2, 3: M * N * 2, 4: M * N * 2}[epi]       # residual read / GELU' written / GELU' read
    row(f"{tag} [{M}x{K}]·[{N}x{K}]ᵀ epi{epi}", us, 2.0 * M * N * K, (M * K + N * K + M * N) * 2 + extra)


HEAD = "| launch | us | TFLOP/s | /2500 | GB/s (algorithmic) | /8000 | bound | frac of its roofline |\n|---|---|---|---|---|---|---|---|"
print("## ResNet-18 (CIFAR stem), batch 256, bf16\n\n" + HEAD)
conv_rows("stem", 256, 32, 32, 3, 64, 3, 1, 1)
conv_rows("layer1", 256, 32, 32, 64, 64, 3, 1, 4)
conv_rows("layer2.0", 256, 32, 32, 64, 128, 3, 2, 1)
conv_rows("layer2.0 shortcut", 256, 32, 32, 64, 128, 1, 2, 1)
pair_row("layer2.0", 256, 32, 32, 64, 128)
conv_rows("layer2", 256, 16, 16, 128, 128, 3, 1, 3)
conv_rows("layer3.0", 256, 16, 16, 128, 256, 3, 2, 1)
conv_rows("layer3.0 shortcut", 256, 16, 16, 128, 256, 1, 2, 1)
pair_row("layer3.0", 256, 16, 16, 128, 256)
conv_rows("layer3", 256, 8, 8, 256, 256, 3, 1, 3)
conv_rows("layer4.0", 256, 8, 8, 256, 512, 3, 2, 1)
conv_rows("layer4.0 shortcut", 256, 8, 8, 256, 512, 1, 2, 1)
pair_row("layer4.0", 256, 8, 8, 256, 512)
conv_rows("layer4", 256, 4, 4, 512, 512, 3, 1, 3)
print("\n## CifarResNet-32, batch 256, bf16\n\n" + HEAD)
conv_rows("stem", 256, 32, 32, 3, 16, 3, 1, 1)
conv_rows("stage1", 256, 32, 32, 16, 16, 3, 1, 10)
conv_rows("stage2.0", 256, 32, 32, 16, 32, 3, 2, 1)
conv_rows("stage2.0 shortcut", 256, 32, 32, 16, 32, 1, 2, 1)
entry_rows("stage2.0", 256, 32, 32, 16, 32)
conv_rows("stage2", 256, 16, 16, 32, 32, 3, 1, 9)
conv_rows("stage3.0", 256, 16, 16, 32, 64, 3, 2, 1)
conv_rows("stage3.0 shortcut", 256, 16, 16, 32, 64, 1, 2, 1)
entry_rows("stage3.0", 256, 16, 16, 32, 64)
conv_rows("stage3", 256, 8, 8, 64, 64, 3, 1, 9)
for tag, rows in (("InfLoRA_OPT b128", 128 * 197), ("L2P b16", 16 * 222)):
    print(f"\n## ViT-B/16 block GEMMs, {tag} ({rows} token rows), bf16\n\n" + HEAD)
    gemm_row("qkv", rows, 2304, 768, 1)
    gemm_row("proj (+bias +residual)", rows, 768, 768, 2)
    gemm_row("fc1 (+bias, GELU, GELU')", rows, 3072, 768, 3)
    gemm_row("fc2 (+bias +residual)", rows, 768, 3072, 2)
    gemm_row("d fc2 (x GELU')", rows, 3072, 768, 4)
    gemm_row("d fc1", rows, 768, 3072, 0)
    gemm_row("d proj", rows, 768, 768, 0)
    gemm_row("d qkv", rows, 768, 2304, 0)


def attn_rows(tag, B, N):
    """softmax(q k^T / 8) v and its backward on the packed qkv (attn.hip).  Algorithmic bytes: qkv read once + the output written (forward);
    qkv, out, d out read + d qkv written (backward).  AI = 99 / 125 FLOP per byte at 197 tokens: HBM-bound, not MFMA-bound."""
    H, D = 12, 768
    M = B * N
    qkv = torch.randn(M, 3 * D, device=dev).to(tdt)
    out = torch.empty(M, D, device=dev, dtype=tdt)
    lse = torch.empty(B * H * N, device=dev)
    dout = torch.randn(M, D, device=dev).to(tdt)
    dqkv = torch.empty(M, 3 * D, device=dev, dtype=tdt)
    dsum = torch.empty(B * H * N, device=dev)
    us = timed(lambda: _lib.call("clhip_attn_fwd", qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, N, H, D, code, st))
    row(f"{tag} attention forward, {B} x {H} heads x {N} tokens", us, 4.0 * B * H * N * N * 64, (M * 3 * D + M * D) * 2)
    us = timed(lambda: _lib.call("clhip_attn_bwd", qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), dsum.data_ptr(), B, N, H, D, code, st))
    row(f"{tag} attention backward", us, 10.0 * B * H * N * N * 64, (M * 3 * D * 2 + M * D * 2) * 2)


def gram_row(tag, B, N):
    M, D, Lr = B * N, 768, 12
    h = torch.randn(Lr, M, D, device=dev).to(tdt)
    G = torch.zeros(Lr, D, D, device=dev)
    us = timed(lambda: _lib.call("clhip_gram_accum_batched", h.data_ptr(), M * D, Lr, G.data_ptr(), M, D, code, st))
    row(f"{tag} Gram X^T X of the 12 attention inputs, one launch", us, Lr * 2.0 * M * D * D, Lr * (M * D * 2 + 2 * D * D * 4))


print("\n## ViT-B/16 attention and the InfLoRA Gram launch, bf16\n\n" + HEAD)
attn_rows("InfLoRA_OPT b128", 128, 197)
attn_rows("L2P b16", 16, 222)
gram_row("InfLoRA_OPT b128", 128, 197)
