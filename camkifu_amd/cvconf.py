"""Vision configuration constants with the same names the reference keeps in
camkifu/config/cvconf.py (canonical_size, frame_period, unsynced, file_fps, bfinders, sfinders)."""
from . import golib_shim

canonical_size = 20 * golib_shim.gsize      # 380: side of the straightened goban image
frame_period = 0.2                          # min seconds between two iterations (live input)
unsynced = "unsynced"                       # marker returned by lock-step file readers
file_fps = 5                                # target read rate for video files

# (module, class) pairs resolved by VManagerBase._reflect; first importable entry is the default
bfinders = [
    ("camkifu_amd.board.bf_auto", "BoardFinderAuto"),
    ("None", "None"),
]
sfinders = [
    ("camkifu_amd.stone.sf_neural", "SfNeural"),
    ("camkifu_amd.stone.sf_contours", "SfContours"),
    ("None", "None"),
]
bf_loc = None
sf_loc = None
