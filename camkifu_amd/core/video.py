"""VidProcessor: the frame loop every finder inherits (mirror of the reference's
core/video.py:14-332 for the parts callers touch: execute, _doframe, interrupt, pause, next,
ready_to_read, full_speed, total_f_processed, bindings, metadata, _show, _window_name).
Display goes to the manager's image queue when there is one and is otherwise dropped: HighGUI
is out of scope."""
import collections
import threading
import time
import traceback

from .. import cvconf


class VidProcessor:
    def __init__(self, vmanager):
        self.vmanager = vmanager
        self.bindings = {'p': self.pause, 'q': self.interrupt, 'f': self.next}
        self.key = None
        self.total_f_processed = 0
        self.frame_period = cvconf.frame_period
        self.full_speed = False
        self.last_read = 0.0
        self._interruptflag = False
        self.pausedflag = False
        self.next_flag = False
        self.own_images = {}
        self.last_shown = collections.defaultdict(lambda: 0)
        self.ignored_show = collections.defaultdict(lambda: 0)
        self.metadata = collections.defaultdict(list)

    # ---- main loop ---------------------------------------------------------------------------
    def execute(self):
        try:
            self._interruptflag = False
            while not self._interrupt_mainloop():
                self._checkpause()
                due = self.full_speed or (self.frame_period < time.time() - self.last_read)
                if self.ready_to_read() and due:
                    ret, frame = self.vmanager.read(self)
                    if ret:
                        self.last_read = time.time()
                        self._doframe(frame)
                        self.total_f_processed += 1
                    elif not (isinstance(frame, str) and frame == cvconf.unsynced):
                        if self.terminated_video():
                            break
                        print("Could not read camera for {0}.".format(type(self)))
                        time.sleep(2)
                else:
                    time.sleep(self.frame_period / 10)
        except BaseException as exc:
            self.vmanager.error_raised(self, exc)
            traceback.print_exc()
        finally:
            self.vmanager.confirm_stop(self)

    def _interrupt_mainloop(self):
        return self.terminated_video() or self._interruptflag

    def terminated_video(self):
        capt = getattr(self.vmanager, "capt", None)
        if capt is None:
            return False
        return self.vmanager.controller.bounds[1] <= capt.progress()

    def ready_to_read(self):
        return not self._interruptflag

    def _doframe(self, frame):
        raise NotImplementedError("Abstract method meant to be extended")

    def interrupt(self):
        self._interruptflag = True

    def pause(self, dopause=None):
        self.pausedflag = (not self.pausedflag) if dopause is None else bool(dopause)

    def next(self):
        self.next_flag = True

    def _checkpause(self):
        while self.pausedflag and not self._interruptflag:
            if self.next_flag:
                self.next_flag = False
                break
            time.sleep(0.05)

    # ---- display (queue only) ------------------------------------------------------------------
    def _window_name(self):
        return type(self).__name__

    def _show(self, img, name=None, frame=True, latency=True, thread=False, loc=None, max_frequ=2):
        name = name or self._window_name()
        q = getattr(self.vmanager, "imqueue", None)
        if q is None:
            self.metadata.clear()
            return
        now = time.time()
        if 1 / max_frequ < now - self.last_shown[name]:
            try:
                q.put_nowait((name, img, self, loc))
                self.own_images[name] = img
                self.last_shown[name] = now
            except Exception:
                self.ignored_show[name] += 1
        self.metadata.clear()


class VisionThread(threading.Thread):
    """Daemon-thread wrapper of a VidProcessor (core/video.py:335-355): run() is the processor's
    execute(); every other attribute is delegated to the processor."""

    def __init__(self, processor):
        super().__init__(name=processor.__class__.__name__)
        self.daemon = True
        self.processor = processor
        self.run = processor.execute

    def __getattr__(self, item):
        return getattr(self.processor, item)
