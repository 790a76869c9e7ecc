"""amdq3d — MI355X-native quantized CenterPoint inference path (Python host side).

The directory name follows the project naming scheme and is not a valid Python identifier; import the
package through the top-level alias module `amdq3d` (repo root), which points here.
"""
__version__ = "0.1.0"

import os as _os
import sys as _sys

# ROCm 7.2's hipGraph "packet capture" fast path (pre-built AQL packets + kernel-argument blocks made at instantiation)
# keeps those argument blocks in memory that later hipMalloc calls hand out again: any allocation + write after a capture
# (a second device plan, a user tensor) silently corrupts the arguments of the captured kernels — wrong detections, or a
# multi-second stall (measured, reproducible, scratch/dbg_async5.py). The flag below selects the runtime's ordinary
# per-node graph launch, which is unaffected and measured equally fast for this workload (359 vs 359 frames/s). The HIP
# runtime reads it once, at its first call, so it has to be in the environment before CUDA is initialised; when that is
# already too late (the process holds /dev/kfd, i.e. the runtime has started — torch.cuda.is_initialized() does not
# tell), GRAPH_REPLAY_SAFE is False and the engine launches eagerly instead of replaying graphs.
_FLAG = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"


def _hip_runtime_started():
    """Has the HIP/HSA runtime of this process started (and therefore read its DEBUG_CLR_* / GPU_* settings)? torch state does
    not tell: torch.cuda.is_available() or any other HIP user starts the runtime while torch.cuda.is_initialized() stays False.
    What does tell is the driver handle: the runtime opens /dev/kfd during its one-time initialisation, right after it has
    read the settings. None = cannot tell (no /proc)."""
    try:
        for fd in _os.listdir("/proc/self/fd"):
            try:
                if _os.readlink("/proc/self/fd/" + fd) == "/dev/kfd":
                    return True
            except OSError:
                continue
        return False
    except OSError:
        return None


def _guard_graph_replay():
    v = _os.environ.get(_FLAG)
    started = _hip_runtime_started()
    if v is not None:
        # a launcher-provided value counts; a value somebody put into os.environ after the runtime started does not, but that
        # cannot be told apart here — the engine's replay self-check (tests/test_engine_gpu.py) is the backstop
        return v == "0"
    if started or (started is None and "torch" in _sys.modules):
        return False        # too late: the runtime has read (or may have read) its settings without the flag
    _os.environ[_FLAG] = "0"
    # forward_async keeps four frames in flight on four HIP streams (one per device plan; a frame is one linear hipGraph): every
    # stream gets a hardware queue of its own and the graphs' internal queues the rest (GPU_MAX_HW_QUEUES = 2 / 4 / 8 / 16 with
    # four plans: 504 / 608 / 754 / 747 frames/s in round 2; the runtime's default is 4)
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    return True


GRAPH_REPLAY_SAFE = _guard_graph_replay()
