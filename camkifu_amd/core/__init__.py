from .exceptions import ControllerWarning, CorrectionWarning, DeletedError  # noqa: F401
from .video import VidProcessor  # noqa: F401
from .vmanager import VManagerBase, VManagerSeq, ArrayCapture  # noqa: F401
