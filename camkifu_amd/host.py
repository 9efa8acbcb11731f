"""Where the drop-in finders take their base classes from.

"Drop into vmanager.py unchanged" (north_star) means: registered as `(module, class)` pairs in the host
application's `cvconf.bfinders / sfinders`, instantiated as `cls(vmanager)` (reference core/vmanager.py:163-198,
347, 367) and driven through the host's own `VidProcessor` loop.  So when CamKifu is importable the finders of
this package inherit `camkifu.board.boardfinder.BoardFinder` / `camkifu.stone.stonesfinder.StonesFinder`
(their frame loop, display, user-correction learning and controller sink stay the host's) and only override the
hooks where the arithmetic lives.  Otherwise they inherit the standalone protocol bases of this package, which
offer the same hooks to the headless harnesses.  CAMKIFU_AMD_STANDALONE=1 forces the latter."""
import importlib
import os


def _host_class(module, name):
    if os.environ.get("CAMKIFU_AMD_STANDALONE") == "1":
        return None
    try:
        return getattr(importlib.import_module(module), name)
    except Exception:                      # CamKifu (or its cv2 / golib / keras dependencies) is not installed
        return None


def board_finder_base():
    found = _host_class("camkifu.board.boardfinder", "BoardFinder")
    if found is None:
        from .board.boardfinder import BoardFinder as found
    return found


def stones_finder_base():
    found = _host_class("camkifu.stone.stonesfinder", "StonesFinder")
    if found is None:
        from .stone.stonesfinder import StonesFinder as found
    return found


def in_host():
    return _host_class("camkifu.board.boardfinder", "BoardFinder") is not None
