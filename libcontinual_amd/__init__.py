"""libcontinual_amd -- MI355X-native continual-learning training engine with the plugin surface of
RL-VIG/LibContinual (`core.model` / `core.trainer`): method `.observe()/.inference()/.before_task()/
.after_task()/.get_parameters()`, name-based backbone registry, rehearsal-buffer API, YAML config loader.

The per-task inner loop (ResNet forward/backward, BatchNorm, EWC / KD / LUCIR terms, Fisher accumulation,
fused SGD/Adam) runs as hand-written gfx950 HIP kernels in `libclhip.so` behind the C ABI of
`include/clhip.h`; this Python package is host orchestration only and has no CPU fallback.
"""
import os as _os

# Hardware-queue cap (round 4, the cause behind DESIGN.md lesson 11 "count the streams"): the HIP runtime gives every stream a hardware queue of its
# own until GPU_MAX_HW_QUEUES (default 4) are in use and multiplexes the rest.  A step that keeps FIVE queues busy at once -- LwF task >= 1 with
# the shortcut-branch streams on: caller + weight-gradient + branch + teacher + teacher-branch -- runs 6.15 ms instead of 2.61 ms whenever the cap
# allows five queues (GPU_MAX_HW_QUEUES >= 5; 2.90 ms at 4, 2.61 ms at 3 -- the time of the same step WITHOUT the extra streams): the chip's
# command processor serves four compute pipes, a fifth busy queue is time-sliced against the others and every cross-stream event wait behind it
# pays the slice.  Three queues cost nothing measurable on any workload (kernels of different streams still overlap inside one hardware queue --
# only same-stream packets carry the barrier bit: ResNet-18 task-0 step 2.1135 / 2.1025 / 2.098 / 2.096 ms at 1 / 3 / 4 / 8 queues), so the cap is
# set here for every process that imports the package before the HIP runtime starts; an explicit GPU_MAX_HW_QUEUES in the environment wins.
# Measurements: profiles/r04_stream_stall.md.
# Round 5: THREE queues for a single-GPU process, FOUR for a rank of a multi-process job (WORLD_SIZE > 1 in the environment, as torch.distributed.run sets it).
# RCCL brings streams of its own; with three queues the step's weight-gradient stream then shares a hardware queue with one of them (which one depends on
# the order the streams were created in) and a queue whose head packet waits for an event holds back the other stream's packets behind it: the batch-256
# ResNet-18 step on a 1-rank RCCL group ran 2.19 OR 2.61 ms at a cap of 2 or 3 -- plain steps without a reducer included, 2.10 vs 2.56-2.60 -- and
# 2.15-2.19 / 2.10 ms every time at 4; 4.2-4.3 ms at 6 (tools/dp_step_micro.py, profiles/r05_notes.md).  Without RCCL in the process three queues are the
# faster setting by 1.5 % (2.053 vs 2.087 ms, two alternating runs) and the other workloads do not care (iCaRL, LwF with the teacher, InfLoRA: within 1 %).
def _default_queue_cap():
    try:
        return "4" if int(_os.environ.get("WORLD_SIZE", "1")) > 1 else "3"
    except ValueError:
        return "3"


_QUEUE_CAP_PRESET = "GPU_MAX_HW_QUEUES" in _os.environ
_QUEUE_CAP_DEFAULT = _default_queue_cap()
_os.environ.setdefault("GPU_MAX_HW_QUEUES", _QUEUE_CAP_DEFAULT)


def hw_queue_cap_state():
    """-> ("ok" | "late" | "user", detail).  The cap only takes effect if it is in the environment BEFORE the HIP runtime initialises (ADVICE r4 /
    VERDICT r4 item 7a): "late" = this package was imported after torch had already started the runtime without the variable set -- the runtime
    keeps its default of four queues, which the two-network steps still tolerate (2.90 vs 2.61 ms) but which is not what was measured; "user" =
    the caller chose a value itself."""
    if _QUEUE_CAP_PRESET:
        return "user", _os.environ.get("GPU_MAX_HW_QUEUES")
    return ("late" if _HIP_WAS_UP else "ok"), _QUEUE_CAP_DEFAULT


def _hip_already_up():
    import sys
    t = sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:
        return False


_HIP_WAS_UP = _hip_already_up()
if _HIP_WAS_UP and not _QUEUE_CAP_PRESET:
    import warnings as _warnings
    _warnings.warn("libcontinual_amd was imported after the HIP runtime had started: its GPU_MAX_HW_QUEUES setting cannot take effect any more (the runtime keeps its "
                   "default of 4 hardware queues).  Steps that run a frozen teacher beside the student keep their branch streams off, so nothing stalls, "
                   "but import libcontinual_amd (or set GPU_MAX_HW_QUEUES yourself: 3 on one GPU, 4 for a rank of a multi-GPU job) before the first torch.cuda call to get the measured configuration.")

from . import _lib  # noqa: E402,F401

__version__ = "0.1.0"
