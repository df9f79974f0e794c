"""micro-benchmark of one conv launch shape (used for PMC passes): python tools/conv_micro.py N H W C K ks stride which[fwd|dgrad|wgrad] reps"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libcontinual_amd import _lib
N, H, W, C, K, ks, stride = map(int, sys.argv[1:8]); which = sys.argv[8]; reps = int(sys.argv[9])
pad = 1 if ks == 3 else 0
dev = "cuda"; tdt = torch.bfloat16; code = _lib.BF16
Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
x = torch.randn(N, H, W, C, device=dev).to(tdt)
w = (torch.randn(K, ks * ks, C, device=dev) * 0.05).to(tdt)
wd = (torch.randn(C, ks * ks, K, device=dev) * 0.05).to(tdt)
z = torch.randn(N, Ho, Wo, K, device=dev).to(tdt)
dx = torch.empty(N, H, W, C, device=dev, dtype=tdt)
dw = torch.zeros(K, ks * ks, C, device=dev)
tiles = _lib.lib().clhip_conv_fwd_tiles(N, H, W, C, K, ks, stride, pad)
part = torch.empty(tiles, 2, K, device=dev)
st = torch.cuda.current_stream().cuda_stream
wsb = _lib.lib().clhip_conv_wgrad_ws_bytes(N, H, W, C, C, K, ks, stride, pad, code)
wsbuf = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
wsp = wsbuf.data_ptr() if (wsb and not os.environ.get('WGRAD_ATOMIC')) else None
def run():
    if which == "fwd":
        _lib.call("clhip_conv_fwd", x.data_ptr(), w.data_ptr(), z.data_ptr(), part.data_ptr(), N, H, W, C, K, ks, stride, pad, code, st)
    elif which == "dgrad":
        _lib.call("clhip_conv_dgrad", z.data_ptr(), wd.data_ptr(), dx.data_ptr(), 0, N, H, W, C, K, ks, stride, pad, code, st)
    else:
        _lib.call("clhip_conv_wgrad", x.data_ptr(), z.data_ptr(), dw.data_ptr(), wsp, N, H, W, C, C, K, ks, stride, pad, code, st)
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * N * Ho * Wo * ks * ks * C * K
print(f"{which} N{N} {H}x{W} C{C} K{K} k{ks} s{stride}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TFLOP/s  in {x.numel()*2/1e6:.1f} MB out {z.numel()*2/1e6:.1f} MB")
