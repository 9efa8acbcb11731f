"""Frame-loop protocol of a finder.

When the host application (CamKifu) is importable the drop-in finders inherit ITS loop (see
camkifu_amd/host.py); this module is the standalone stand-in used by the headless harnesses, the tests
and the batch tools.  It offers the attributes SURVEY 8(b) lists as touched by callers -- execute,
_doframe, interrupt, pause, next, ready_to_read, full_speed, total_f_processed, bindings, _show,
metadata, _window_name (reference: core/video.py:60-332) -- built on threading.Event objects rather
than polled flags: a paused loop sleeps on an event, an interrupt wakes every wait at once.
Display output goes to the manager's image queue if it has one; HighGUI is out of scope."""
import time
from collections import Counter, defaultdict
from threading import Event, Thread
from traceback import print_exc

from .. import cvconf


class VidProcessor:
    RETRY_SECONDS = 2.0          # after a failed camera read

    def __init__(self, manager):
        self.vmanager, self.total_f_processed, self.full_speed = manager, 0, False
        self.frame_period, self.metadata = cvconf.frame_period, defaultdict(list)
        self.bindings = {"p": self.pause, "q": self.interrupt, "f": self.next}
        self._quit, self._go, self._one = Event(), Event(), Event()      # _go is cleared while paused; _one = "next"
        self._go.set()
        self._read_at, self._shown_at, self.dropped_images = float("-inf"), {}, Counter()

    # ------------------------------------------------------------------ to be provided by finders
    def _doframe(self, frame):
        raise NotImplementedError("a finder implements _doframe(frame)")

    def ready_to_read(self):
        return not self._quit.is_set()

    def _window_name(self):
        return type(self).__name__

    # ------------------------------------------------------------------ loop
    def execute(self):
        vm = self.vmanager
        self._quit.clear()
        try:
            while not (self._quit.is_set() or self._video_over()):
                self._hold_while_paused()
                if not (self.ready_to_read() and self._is_due()):
                    self._quit.wait(self.frame_period / 10)
                    continue
                ok, frame = vm.read(self)
                if ok:
                    self._read_at = time.monotonic()
                    self._doframe(frame)
                    self._count_frame()
                elif isinstance(frame, str) and frame == cvconf.unsynced:
                    continue                                  # the reader let go of its consumers: ask again
                elif self._video_over():
                    break
                else:
                    print("%s: no frame from the input, retrying" % type(self).__name__)
                    self._quit.wait(self.RETRY_SECONDS)
        except BaseException as failure:                      # the manager stops everything (core/video.py:115-120)
            vm.error_raised(self, failure)
            print_exc()
        finally:
            vm.confirm_stop(self)

    def _count_frame(self):
        self.total_f_processed = self.total_f_processed + 1

    def _is_due(self):
        return self.full_speed or time.monotonic() - self._read_at > self.frame_period

    def _video_over(self):
        capt = getattr(self.vmanager, "capt", None)
        return capt is not None and capt.progress() >= self.vmanager.controller.bounds[1]

    terminated_video = _video_over

    def _hold_while_paused(self):
        while not (self._go.is_set() or self._quit.is_set()):
            if self._one.is_set():
                self._one.clear()
                return
            self._go.wait(0.05)

    # ------------------------------------------------------------------ controls (key bindings)
    def interrupt(self):
        self._quit.set()
        self._go.set()

    def pause(self, dopause=None):
        want = self._go.is_set() if dopause is None else bool(dopause)
        (self._go.clear if want else self._go.set)()

    @property
    def pausedflag(self):
        return not self._go.is_set()

    def next(self):
        self._one.set()

    # ------------------------------------------------------------------ display
    def _show(self, img, name=None, loc=None, max_frequ=2, **_ignored):
        """hand `img` to the manager's image queue, at most `max_frequ` times per second per window"""
        name = name or self._window_name()
        sink = getattr(self.vmanager, "imqueue", None)
        now = time.monotonic()
        if sink is not None and now - self._shown_at.get(name, float("-inf")) > 1.0 / max_frequ:
            try:
                sink.put_nowait((name, img, self, loc))
                self._shown_at[name] = now
            except Exception:
                self.dropped_images[name] += 1
        self.metadata = defaultdict(list)


class VisionThread(Thread):
    """one daemon thread per finder; unknown attributes resolve on the finder itself"""

    def __init__(self, finder):
        super().__init__(name=type(finder).__name__, daemon=True, target=finder.execute)
        self.__dict__["processor"] = finder

    def __getattr__(self, attr):
        return getattr(self.__dict__["processor"], attr)
