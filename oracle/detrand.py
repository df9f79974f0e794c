"""Portable deterministic pseudo-random tensors for fixtures (test infrastructure).

Golden fixtures store a *tag* instead of the tensor: both the generator script (which runs the
reference) and the tests rebuild identical inputs / weights from the tag with integer arithmetic
only (splitmix64 over the element index), so they are bit-identical on any machine, numpy or torch
version.
"""
import zlib
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(tag, shape, lo=-1.0, hi=1.0):
    """float32 array of `shape`, i.i.d.-looking uniform in [lo, hi), a pure function of (tag, shape)."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(zlib.crc32(tag.encode()) + 1)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + seed * np.uint64(0x100000001B3)
    bits = _splitmix64(idx) >> np.uint64(40)          # top 24 bits
    u = bits.astype(np.float32) / np.float32(1 << 24)  # exact in fp32
    return (np.float32(lo) + u * np.float32(hi - lo)).reshape(shape)


def randint(tag, shape, lo, hi):
    """int64 array in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(zlib.crc32(tag.encode()) + 7)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + seed * np.uint64(0x100000001B3)
    bits = _splitmix64(idx) >> np.uint64(33)
    return (lo + (bits % np.uint64(hi - lo)).astype(np.int64)).reshape(shape)
