"""camkifu_amd -- MI355X-native implementation of CamKifu's per-frame vision hot path.

Compute lives in camkifu_amd/libck_hip.so (hand-written HIP for gfx950, C-ABI in
include/camkifu_amd.h); this package holds the ctypes binding and the host-side mirror of the
reference's VidProcessor / BoardFinder / StonesFinder plugin interface.
"""
import os as _os

# Every ck_ctx owns a HIP stream, and the design leans on several of them being in flight at once (board path and stones
# path of each lane, the background-model stream, upload streams).  The HIP runtime multiplexes streams onto
# GPU_MAX_HW_QUEUES hardware queues (4 by default); streams that share a queue run one after the other.  Measured on an
# MI355X: five contexts created after other GPU work has claimed queues -> 18.4 ms per 256-frame step with 4 queues,
# 14.0 ms with 6 or more.  The variable is read when the HIP runtime initialises, so it is set (if the user has not)
# as early as this package is imported; a process that initialised HIP earlier keeps whatever it had.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.2.0"
