"""Headless vision managers -- only what the harnesses and tests need around the finders.

The reference's managers (core/vmanager.py) own the capture, pick the finder classes by name and run
them; the drop-in finders of this package are meant to be registered with THAT manager unchanged
(INTEGRATION.md).  Standalone, this module supplies the same seams: `(module, class)` registration
resolved by `VManagerBase._reflect` (core/vmanager.py:163-198), `read(caller)`, `error_raised`,
`confirm_stop`, `board_finder` / `stones_finder` / `capt` / `controller` / `imqueue` / `current_video`,
a threaded manager (one daemon thread per finder, frames served in lock step) and the sequential
driver of the reference's test harness (test/objects/vmanager_test.py)."""
from importlib import import_module
from threading import Event, Thread

from .. import cvconf
from .capture import ArrayCapture, CaptureReader, CaptureReaderBase, open_capture     # noqa: F401  (re-exported)
from .video import VisionThread


def resolve_finder(name, registry):
    """`registry`: [(module path, class name), ...].  The default is the first entry whose module imports;
    `name` picks another importable entry; "None" (as a name, or as the only usable entry) means no finder."""
    usable = []
    if name == "None":
        return None
    for module_path, class_name in registry:
        if class_name == "None":
            continue
        try:
            usable.append((class_name, getattr(import_module(module_path), class_name)))
        except ImportError as why:
            print("finder %s not available: %s" % (class_name, why))
            continue
        if class_name == name:
            return usable[-1][1]
    return usable[0][1] if usable else None


class VManagerBase(Thread):
    _reflect = staticmethod(resolve_finder)
    reader_class = CaptureReaderBase

    def __init__(self, controller, imqueue=None, bf=None, sf=None):
        super().__init__(name="Vision", daemon=True)
        self.controller, self.imqueue = controller, imqueue
        self.capt = self.current_video = None
        self.board_finder = self.stones_finder = None
        self.bf_class, self.sf_class = resolve_finder(bf, cvconf.bfinders), resolve_finder(sf, cvconf.sfinders)
        self.full_speed, self.error = False, None
        # the controller calls back into the vision side for user corrections and single-stepping
        controller.corrected = self.corrected
        controller.next = self.next

    # ---- capture ------------------------------------------------------------------------------
    def init_capt(self):
        old, self.capt = self.capt, None
        if old is not None:
            old.release()
        self.current_video = video = self.controller.video
        source = open_capture(video)
        if not source.isOpened():
            print("cannot open %r: %s" % (video, getattr(source, "error", "")))
            return
        self.capt = self.reader_class(source, self)
        self.full_speed = True                       # files / arrays: as fast as the finders go
        self.capt.seek(self.controller.bounds[0])

    def read(self, finder):
        return self.capt.read(finder)

    def vid_progress(self, percent):
        """progress listeners (a GUI) hook in here"""

    # ---- finders ------------------------------------------------------------------------------
    def finders(self):
        return [p for p in (self.board_finder, self.stones_finder) if p is not None]

    def next(self):
        for p in self.finders():
            p.next()

    def corrected(self, wrong, right):
        """a user correction on the goban: (removed move or None, added move or None)"""
        sf = self.stones_finder
        if sf is not None:
            sf.corrected(wrong, right)

    def error_raised(self, finder, error):
        print("%s stops: %s in %s" % (type(self).__name__, type(error).__name__, type(finder).__name__))
        self.error = error
        type(self).interrupt(self)

    def confirm_stop(self, finder):
        """a finder's loop has ended"""

    def stop_processing(self):
        for p in self.finders():
            p.interrupt()

    def interrupt(self):
        VManagerBase.stop_processing(self)


class VManager(VManagerBase):
    """threads: the manager polls, creates each finder once a capture exists and runs it on its own
    daemon thread; over a file the reader hands every frame to all running finders before moving on"""
    reader_class = CaptureReader
    POLL_SECONDS = 0.02

    def __init__(self, controller, imqueue=None, bf=None, sf=None, active=True):
        VManagerBase.__init__(self, controller, imqueue, bf, sf)
        self.processes, self.active, self.hasrun = [], active, False      # processes: VisionThreads still running
        self._over = Event()

    def run(self):
        VManagerBase.init_capt(self)
        while not self._over.is_set():
            self._poll()
            self._over.wait(self.POLL_SECONDS)
            self.hasrun = True

    def _poll(self):
        if not self.active:
            return
        wanted = self.controller.video
        if wanted is not self.current_video and wanted != self.current_video:
            VManager.stop_processing(self)
            VManagerBase.init_capt(self)
            self.board_finder = self.stones_finder = None
            self.controller.pipe("video_changed")
        if self.capt is not None:
            self._ensure("board_finder", self.bf_class)
            self._ensure("stones_finder", self.sf_class)

    def _ensure(self, slot, cls):
        if cls is None or getattr(self, slot) is not None:
            return
        finder = cls(self)
        setattr(self, slot, finder)
        thread = VisionThread(finder)
        self.processes.append(thread)
        self._lock_step(True)
        thread.start()

    def _lock_step(self, on):
        release = getattr(self.capt, "unsync_threads", None)
        if release is not None:
            release(not on)

    def is_processing(self):
        return len(self.processes) > 0

    def next(self):
        for t in self.processes:
            t.next()

    def confirm_stop(self, finder):
        self.processes = [t for t in self.processes if t.processor is not finder]

    def stop_processing(self):
        for t in list(self.processes):
            t.interrupt()
        self._lock_step(False)                       # nobody may stay asleep inside the reader

    def interrupt(self):
        self._over.set()
        VManager.stop_processing(self)


class VManagerSeq(VManagerBase):
    """caller's thread only: the board finder runs until it has a transform, then the stones finder runs
    to the end of the video"""

    def __init__(self, controller=None, bf=None, sf=None):
        VManagerBase.__init__(self, controller, None, bf, sf)
        self._cancelled = False

    def run(self):
        VManagerBase.init_capt(self)
        board, stones = self.bf_class(self), self.sf_class(self)
        self.board_finder, self.stones_finder = board, stones
        board.full_speed = stones.full_speed = True
        per_frame = board._doframe

        def stop_once_located(frame):
            per_frame(frame)
            if board.mtx is not None:
                board.interrupt()
        board._doframe = stop_once_located
        try:
            board.execute()
        finally:
            board._doframe = per_frame
        if board.mtx is not None and not self._cancelled:
            stones.execute()

    def stop_processing(self):
        self._cancelled = True
        VManagerBase.stop_processing(self)
