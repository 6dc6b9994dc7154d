"""geoformer_amd -- MI355X (gfx950) native layer for GeoFormer's per-scene hot path.

Only what the path needs lives here: ``csrc/`` (hand-written HIP kernels + the C ABI of
``include/geoformer_hip.h``), the ctypes binding, and the host-side mirrors of the
reference's operator interfaces (spconv / PG_OP / pointnet2._ext / faiss shapes).
"""
import os as _os

__version__ = "0.1.0"

# Hardware queues.  A training batch runs every scene's sampling and every scene's geodesic BFS on a stream of its own
# (eight latency-bound kernels that should run beside each other, plus the main / side / aux streams), and the eval
# forward uses four streams.  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels
# of streams that share a queue run one after the other (rocprofv3 trace of a batch-4 step: the third scene's sampling
# started when the first scene's BFS ended; eval forward 183 -> 143 scenes/s when streams alias).  The runtime reads the
# variable when it initialises (the process's first HIP call).  Importing this package does NOT touch the process
# environment: a harness calls ``configure_runtime()`` before its first HIP call (bench.py, tools/train_dp.py,
# tests/conftest.py and __graft_entry__.py do), or exports the variable itself; ``_lib.load()`` warns when the library
# is loaded into a process whose runtime is already up with fewer queues.
HW_QUEUES_WANTED = 16
HW_QUEUES_MIN = 8


def hw_queues_setting():
    """The value the HIP runtime will read / has read (None: unset, the runtime's default of 4)."""
    v = _os.environ.get("GPU_MAX_HW_QUEUES")
    try:
        return int(v) if v is not None else None
    except ValueError:
        return None


def configure_runtime(hw_queues: int = HW_QUEUES_WANTED) -> bool:
    """Ask the HIP runtime for enough hardware queues; must run BEFORE the process's first HIP call (a value the user
    exported wins).  Returns False -- and changes nothing -- when torch has already initialised the GPU: the variable
    would not be read any more."""
    import sys

    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_initialized():
        return hw_queues_setting() is not None and hw_queues_setting() >= HW_QUEUES_MIN
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", str(int(hw_queues)))
    return True
