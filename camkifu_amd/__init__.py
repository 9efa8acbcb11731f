"""camkifu_amd -- MI355X-native implementation of CamKifu's per-frame vision hot path.

Compute lives in camkifu_amd/libck_hip.so (hand-written HIP for gfx950, C-ABI in
include/camkifu_amd.h); this package holds the ctypes binding and the host-side mirror of the
reference's VidProcessor / BoardFinder / StonesFinder plugin interface.
"""
__version__ = "0.1.0"
