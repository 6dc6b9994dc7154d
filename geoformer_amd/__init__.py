"""geoformer_amd -- MI355X (gfx950) native layer for GeoFormer's per-scene hot path.

Only what the path needs lives here: ``csrc/`` (hand-written HIP kernels + the C ABI of
``include/geoformer_hip.h``), the ctypes binding, and the host-side mirrors of the
reference's operator interfaces (spconv / PG_OP / pointnet2._ext / faiss shapes).
"""
__version__ = "0.1.0"
