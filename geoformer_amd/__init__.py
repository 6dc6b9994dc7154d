"""geoformer_amd -- MI355X (gfx950) native layer for GeoFormer's per-scene hot path.

Only what the path needs lives here: ``csrc/`` (hand-written HIP kernels + the C ABI of
``include/geoformer_hip.h``), the ctypes binding, and the host-side mirrors of the
reference's operator interfaces (spconv / PG_OP / pointnet2._ext / faiss shapes).
"""
import os as _os

__version__ = "0.1.0"

# A training batch runs every scene's sampling and every scene's geodesic BFS on a stream of its own (eight latency-bound
# kernels that should run beside each other, plus the main / side / aux streams).  The HIP runtime maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels of streams that share a queue run one after the other
# (rocprofv3 trace of a batch-4 step: the third scene's sampling started when the first scene's BFS ended).  The
# runtime reads the variable when it initialises (the process's first HIP call), so the package has to be imported
# before that; a value the user exported wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
