"""TEST INFRASTRUCTURE (oracle): the anti-aliased bilinear resize the reference's ImageNet-R input pipeline performs.

The reference declares `RandomResizedCrop` / `Resize` in its YAML (config/InfLoRA_opt-vit-imagenetr-b20-20-10.yaml:28-43) and
builds them from torchvision.transforms (core/data/dataloader.py:17-37) on PIL images, so the arithmetic is Pillow's
ImagingResample (third-party dependency, absent from /root/reference; this image ships Pillow 12.2.0, requirements.txt pins none).
Published algorithm restated here (src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
ImagingResampleHorizontal_8bpc / Vertical_8bpc):
  * per output position a triangle filter of support max(scale, 1) centred at (o + 0.5) * scale, taps [xmin, xmax) clamped to
    the image, weights normalised to sum 1 in double precision;
  * weights -> fixed point with 22 fractional bits, round half away from zero;
  * horizontal pass first, accumulate from 2**21, shift right by 22, saturate to uint8; then the vertical pass on that uint8 image.
Pinned against Pillow itself (tests/test_host_cpu.py::test_resample_oracle_matches_pillow): bit-exact.
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def coefficients(in_size, out_size):
    """-> (xmin[out], count[out], kk[out, ksize] int64): Resample.c precompute_coeffs + normalize_coeffs_8bpc for a full-image box"""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int64)
    cnt = np.zeros(out_size, np.int64)
    kk = np.zeros((out_size, ksize), np.int64)
    ss = 1.0 / filterscale
    for o in range(out_size):
        center = (o + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), in_size)
        n = hi - lo
        w = np.zeros(n)
        for t in range(n):
            v = abs((t + lo - center + 0.5) * ss)
            w[t] = 1.0 - v if v < 1.0 else 0.0
        ww = 0.0
        for t in range(n):                     # Resample.c adds in tap order (numpy's pairwise sum would differ for wide filters)
            ww += w[t]
        if ww != 0.0:
            w = w / ww
        q = w * float(1 << PRECISION_BITS)
        kk[o, :n] = np.where(q < 0, (q - 0.5).astype(np.int64), (q + 0.5).astype(np.int64))
        xmin[o], cnt[o] = lo, n
    return xmin, cnt, kk


def _pass(img, out_size, axis):
    """one resampling pass over `axis` of a uint8 [H, W, C] image"""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    xmin, cnt, kk = coefficients(src.shape[0], out_size)
    out = np.empty((out_size,) + src.shape[1:], np.int64)
    for o in range(out_size):
        n = int(cnt[o])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for t in range(n):
            acc = acc + src[xmin[o] + t] * kk[o, t]
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out.astype(np.uint8), 0, axis)


def resize_bilinear(img, out_h, out_w):
    """PIL `Image.resize((out_w, out_h), BILINEAR)` of a uint8 [H, W, 3] array (horizontal pass, then vertical; a pass whose
    size is unchanged is the identity)"""
    a = np.ascontiguousarray(img)
    if a.shape[1] != out_w:
        a = _pass(a, out_w, 1)
    if a.shape[0] != out_h:
        a = _pass(a, out_h, 0)
    return a


def resized_crop(img, y0, x0, h, w, out_h, out_w):
    """torchvision F.resized_crop: crop first (the filter clamps at the crop border), then resize"""
    return resize_bilinear(img[y0:y0 + h, x0:x0 + w], out_h, out_w)
