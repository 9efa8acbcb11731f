"""Frame sources and the capture readers of the reference's vision manager.

The reference reads frames with cv2.VideoCapture (FFmpeg) and wraps it in CaptureReaderBase /
CaptureReader (core/vmanager.py:461-635): when the input is a file, frames are skipped so that
only `cvconf.file_fps` frames per second of video are analysed, and every active VidProcessor
receives the same sequence of frames (lock-step).  Video decoding itself is a third-party
library there and is not rebuilt here; the captures below quack like cv2.VideoCapture over
uncompressed containers:

  Y4MCapture   .y4m (YUV4MPEG2, 4:2:0 planar): frames stay I420 on the host (1.5 B/px) and are
               converted to BGR on the GPU (ck_i420_to_bgr) -- read() for one frame,
               read_raw_batch() for the fast-file pipeline
  ArrayCapture an (n, h, w, 3) uint8 array or .npy file of BGR frames (memory mapped)

cv2 property ids are kept so CaptureReaderBase.skip reads like the reference's.
"""
import os
import threading
import time

import numpy as np

from .. import cvconf

CAP_PROP_POS_FRAMES = 1
CAP_PROP_POS_AVI_RATIO = 2
CAP_PROP_FRAME_WIDTH = 3
CAP_PROP_FRAME_HEIGHT = 4
CAP_PROP_FPS = 5
CAP_PROP_FRAME_COUNT = 7


class ArrayCapture:
    """Frame source over an (n, h, w, 3) uint8 array (or a path to one saved with np.save).
    read() hands each consumer a private copy, like CaptureReader.read_file does."""

    def __init__(self, frames, fps=30.0):
        if isinstance(frames, str):
            frames = np.load(frames, mmap_mode="r")
        self.frames = frames
        self.pos = 0
        self.fps = float(fps)

    def isOpened(self):
        return self.frames is not None

    def read(self, caller=None):
        if self.frames is None or self.pos >= len(self.frames):
            return False, None
        frame = np.array(self.frames[self.pos], copy=True)
        self.pos += 1
        return True, frame

    def get(self, prop):
        if prop == CAP_PROP_POS_FRAMES:
            return float(self.pos)
        if prop == CAP_PROP_FRAME_COUNT:
            return float(len(self.frames))
        if prop == CAP_PROP_FPS:
            return self.fps
        if prop == CAP_PROP_POS_AVI_RATIO:
            return self.progress()
        if prop == CAP_PROP_FRAME_WIDTH:
            return float(self.frames.shape[2])
        if prop == CAP_PROP_FRAME_HEIGHT:
            return float(self.frames.shape[1])
        return 0.0

    def set(self, prop, value):
        if prop == CAP_PROP_POS_FRAMES:
            self.pos = int(value)          # assumption: the FFmpeg backend truncates a fractional frame index
            return True
        if prop == CAP_PROP_POS_AVI_RATIO:
            self.seek(value)
            return True
        return False

    def progress(self):
        return self.pos / max(1, len(self.frames))

    def seek(self, ratio):
        self.pos = int(ratio * len(self.frames))

    def release(self):
        pass


class Y4MError(ValueError):
    pass


def write_y4m(path, frames_i420, h, w, fps=(30, 1)):
    """frames_i420: iterable of flat I420 frames (h*w*3/2 bytes each)."""
    with open(path, "wb") as f:
        f.write(("YUV4MPEG2 W%d H%d F%d:%d Ip A1:1 C420jpeg\n" % (w, h, fps[0], fps[1])).encode())
        for fr in frames_i420:
            buf = np.ascontiguousarray(fr, np.uint8).reshape(-1)
            if buf.size != h * w * 3 // 2:
                raise Y4MError("frame has %d bytes, expected %d" % (buf.size, h * w * 3 // 2))
            f.write(b"FRAME\n")
            f.write(buf.tobytes())


class Y4MCapture:
    """cv2.VideoCapture look-alike over a YUV4MPEG2 4:2:0 file.  The file is memory mapped; frames are
    handed out as I420 (read_raw / read_raw_batch, zero copy) or as BGR through `convert`, a callable
    (i420_flat, h, w) -> (h, w, 3) BGR -- by default the GPU conversion of a camkifu_amd Context."""

    def __init__(self, path, convert=None):
        self.path = path
        self.convert = convert
        self.pos = 0
        self._mm = None
        self._offsets = []
        try:
            self._open(path)
        except (OSError, Y4MError) as exc:
            self.error = exc
            self._mm = None

    def _open(self, path):
        mm = np.memmap(path, dtype=np.uint8, mode="r")
        end = min(len(mm), 4096)
        head = bytes(mm[:end])
        nl = head.find(b"\n")
        if not head.startswith(b"YUV4MPEG2") or nl < 0:
            raise Y4MError("not a YUV4MPEG2 file: " + path)
        self.w = self.h = None
        self.fps = 30.0
        chroma = "420"
        for tok in head[:nl].split(b" ")[1:]:
            tag, val = tok[:1], tok[1:].decode()
            if tag == b"W":
                self.w = int(val)
            elif tag == b"H":
                self.h = int(val)
            elif tag == b"F":
                num, den = val.split(":")
                self.fps = float(num) / float(den) if float(den) else 30.0
            elif tag == b"C":
                chroma = val
        if not self.w or not self.h:
            raise Y4MError("missing W/H in the stream header")
        if not chroma.startswith("420") or "p1" in chroma:      # 420jpeg / 420mpeg2 / 420paldv; no high bit depth
            raise Y4MError("only 8-bit 4:2:0 is supported, got C" + chroma)
        if (self.w | self.h) & 1:
            raise Y4MError("odd dimensions are not supported")
        self.fsize = self.h * self.w * 3 // 2
        # frame index: every frame starts with a "FRAME[ params]\n" line
        off, offsets = nl + 1, []
        n = len(mm)
        while off + 6 <= n:
            line_end = off + 5
            if bytes(mm[off:off + 5]) != b"FRAME":
                raise Y4MError("corrupt frame header at byte %d" % off)
            while line_end < n and mm[line_end] != 0x0A:
                line_end += 1
            data = line_end + 1
            if data + self.fsize > n:
                break                                        # truncated last frame: ignored
            offsets.append(data)
            off = data + self.fsize
        self._mm, self._offsets = mm, offsets

    def isOpened(self):
        return self._mm is not None

    def __len__(self):
        return len(self._offsets)

    def read_raw(self):
        """next frame as a flat I420 view into the mapping (no copy), or None at the end"""
        if self._mm is None or self.pos >= len(self._offsets):
            return None
        o = self._offsets[self.pos]
        self.pos += 1
        return self._mm[o:o + self.fsize]

    def read_raw_batch(self, indices, out=None):
        """frames `indices` stacked into an (len, fsize) uint8 array (`out` may be a reusable, e.g.
        pinned, buffer).  Does not move the read position."""
        if out is None:
            out = np.empty((len(indices), self.fsize), np.uint8)
        for k, i in enumerate(indices):
            o = self._offsets[i]
            out[k] = self._mm[o:o + self.fsize]
        return out[:len(indices)]

    def read(self, caller=None):
        raw = self.read_raw()
        if raw is None:
            return False, None
        convert = self.convert
        if convert is None:
            from .. import capi
            convert = capi.get_context().i420_to_bgr
            self.convert = convert
        return True, np.asarray(convert(np.ascontiguousarray(raw), self.h, self.w))

    def get(self, prop):
        if prop == CAP_PROP_POS_FRAMES:
            return float(self.pos)
        if prop == CAP_PROP_FRAME_COUNT:
            return float(len(self._offsets))
        if prop == CAP_PROP_FPS:
            return self.fps
        if prop == CAP_PROP_POS_AVI_RATIO:
            return self.pos / max(1, len(self._offsets))
        if prop == CAP_PROP_FRAME_WIDTH:
            return float(self.w)
        if prop == CAP_PROP_FRAME_HEIGHT:
            return float(self.h)
        return 0.0

    def set(self, prop, value):
        if prop == CAP_PROP_POS_FRAMES:
            self.pos = int(value)
            return True
        if prop == CAP_PROP_POS_AVI_RATIO:
            self.pos = int(value * len(self._offsets))
            return True
        return False

    def progress(self):
        return self.pos / max(1, len(self._offsets))

    def seek(self, ratio):
        self.pos = int(ratio * len(self._offsets))

    def release(self):
        self._mm = None


def open_capture(video, convert=None):
    """the capture for a controller's `video` attribute: an array, a .npy path or a .y4m path"""
    if isinstance(video, str) and video.lower().endswith(".y4m"):
        return Y4MCapture(video, convert=convert)
    return ArrayCapture(video)


def file_frame_indices(nframes, fps, rate=None, start=0):
    """the frame numbers CaptureReaderBase.skip() makes a file reader visit (core/vmanager.py:511-525):
    before EVERY read the position advances by max(1, fps / rate) and the read itself advances by one,
    so the period is fps/rate + 1 and the first frame analysed is not frame 0."""
    rate = cvconf.file_fps if rate is None else rate
    out, idx = [], float(start)
    while True:
        idx += max(1, fps / rate)
        idx = min(idx, nframes)
        if int(idx) >= nframes:
            return out
        out.append(int(idx))
        idx = int(idx) + 1.0


class CaptureReaderBase:
    """What a manager puts between its finders and the capture: `read(caller)` is taken over, every other
    attribute is the capture's.  A video FILE is thinned to `fps` analysed frames per second of footage with
    the reference's arithmetic (core/vmanager.py:510-525; see file_frame_indices); other sources pass through."""

    def __init__(self, capture, vmanager, fps=None):
        self.__dict__.update(capture=capture, vmanager=vmanager,
                             frame_rate=cvconf.file_fps if fps is None else fps)

    def __getattr__(self, attr):                       # only reached for names this object does not define
        return getattr(self.__dict__["capture"], attr)

    def is_file(self):
        video = getattr(getattr(self.vmanager, "controller", None), "video", None)
        return isinstance(video, str) and os.path.isfile(video)

    def advance(self):
        """move the read position forward by the thinning stride (never beyond the last frame)"""
        cap = self.capture
        target = cap.get(CAP_PROP_POS_FRAMES) + max(1, cap.get(CAP_PROP_FPS) / self.frame_rate)
        cap.set(CAP_PROP_POS_FRAMES, min(target, cap.get(CAP_PROP_FRAME_COUNT)))

    skip = advance

    def read(self, caller=None):
        if self.is_file():
            self.advance()
        return self.capture.read()


class CaptureReader(CaptureReaderBase):
    """Lock-step serving for the threaded manager (what core/vmanager.py:528-635 is for): every finder that is
    currently able to read gets frame k before anybody gets frame k+1.

    Built as a generation counter under one condition variable: the reader holds (generation, frame); a
    consumer remembers the last generation it took and sleeps on the condition while that is still the current
    one; whoever completes the set of consumers for a generation fetches the next frame and wakes the others.
    A consumer therefore sees every frame exactly once -- the reference's reader lets the thread served last
    walk away with the following frame (and sometimes see it twice), a timing artefact that is not mirrored."""

    def __init__(self, capture, vmanager, fps=None):
        super().__init__(capture, vmanager, fps)
        self.__dict__.update(_cv=threading.Condition(), _gen=0, _frame=None, _taken={}, unsync=False,
                             sleep_time=0.05)

    def _consumers(self):
        ready = []
        for thread in getattr(self.vmanager, "processes", ()):
            try:
                if thread.ready_to_read():
                    ready.append(thread.processor)
            except AttributeError:
                continue
        return ready

    def _fetch(self, first):
        """under the lock: load the next generation"""
        if not first:
            self.advance()
        self._frame = self.capture.read()
        self._gen += 1
        if not first:
            self.vmanager.vid_progress(self.capture.get(CAP_PROP_POS_AVI_RATIO) * 100)
        self._cv.notify_all()

    def _maybe_turn_over(self):
        if all(self._taken.get(c) == self._gen for c in self._consumers()):
            self._fetch(first=False)

    def read(self, caller=None):
        if not self.is_file():
            return self.capture.read()
        with self._cv:
            if self._gen == 0:
                self._fetch(first=True)
            while not self.unsync and self._taken.get(caller) == self._gen:
                self._cv.wait(self.sleep_time)
                self._maybe_turn_over()                # a finder may have stopped being ready meanwhile
            if self.unsync:
                return False, cvconf.unsynced
            ok, img = self._frame
            self._taken[caller] = self._gen
            self._maybe_turn_over()
        return (ok, img.copy()) if ok else (ok, img)   # consumers may write on their image

    def unsync_threads(self, unsync):
        """True: wake everybody up with the `unsynced` marker (the manager is stopping its finders)"""
        with self._cv:
            self.unsync = bool(unsync)
            self._cv.notify_all()
