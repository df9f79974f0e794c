"""libcontinual_amd -- MI355X-native continual-learning training engine with the plugin surface of
RL-VIG/LibContinual (`core.model` / `core.trainer`): method `.observe()/.inference()/.before_task()/
.after_task()/.get_parameters()`, name-based backbone registry, rehearsal-buffer API, YAML config loader.

The per-task inner loop (ResNet forward/backward, BatchNorm, EWC / KD / LUCIR terms, Fisher accumulation,
fused SGD/Adam) runs as hand-written gfx950 HIP kernels in `libclhip.so` behind the C ABI of
`include/clhip.h`; this Python package is host orchestration only and has no CPU fallback.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
