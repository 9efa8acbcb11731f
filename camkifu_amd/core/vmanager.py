"""Vision managers: finder registry/reflection (core/vmanager.py:16-198 `VManagerBase`), the
multi-threaded `VManager` with lock-step file reading (core/vmanager.py:201-458) and the sequential
headless driver of the reference's tests (test/objects/vmanager_test.py `VManagerSeq`).
Frames come from core/capture.py: an in-memory array, a .npy file or an uncompressed .y4m file."""
import importlib
import os
import threading
import time

from .. import cvconf
from .capture import ArrayCapture, CaptureReader, CaptureReaderBase, open_capture     # noqa: F401  (ArrayCapture re-exported)
from .video import VisionThread


class VManagerBase(threading.Thread):
    def __init__(self, controller, imqueue=None, bf=None, sf=None):
        threading.Thread.__init__(self, name="Vision")
        self.controller = controller
        self.punch_controller()
        self.imqueue = imqueue
        self.capt = None
        self.current_video = None
        self.bf_class = self._reflect(bf, cvconf.bfinders)
        self.sf_class = self._reflect(sf, cvconf.sfinders)
        self.board_finder = None
        self.stones_finder = None
        self.full_speed = False

    def punch_controller(self):
        self.controller.corrected = self.corrected
        self.controller.next = self.next

    def init_capt(self):
        self.current_video = self.controller.video
        if self.capt is not None:
            self.capt.release()
        self.capt = self._get_capture()
        if self.capt is not None:
            # arrays and files are read as fast as the finders go; only a live camera is rate limited
            self.full_speed = True
            self.capt.seek(self.controller.bounds[0])

    def _get_capture(self):
        """the capture for controller.video; a file path gets the frame-skipping reader (file_fps)"""
        cap = open_capture(self.controller.video)
        if not cap.isOpened():
            print("Could not open video: {}".format(getattr(cap, "error", self.controller.video)))
            return None
        return CaptureReaderBase(cap, self)

    def error_raised(self, processor, error):
        print("{} terminating due to {} in {}.".format(type(self).__name__, type(error).__name__,
                                                       type(processor).__name__))
        self.error = error
        self.interrupt()

    def read(self, caller):
        return self.capt.read(caller)

    def next(self):
        for p in (self.board_finder, self.stones_finder):
            if p is not None:
                p.next()

    def corrected(self, err_move, exp_move):
        if self.stones_finder is not None:
            self.stones_finder.corrected(err_move, exp_move)

    def confirm_stop(self, process):
        pass

    def vid_progress(self, progress):
        pass

    def interrupt(self):
        raise NotImplementedError

    def stop_processing(self):
        raise NotImplementedError

    @staticmethod
    def _reflect(name, classes):
        """first importable (module, class) entry is the default; `name` selects another one"""
        if name == "None":
            return None
        chosen = None
        for m, c in classes:
            if c == "None":
                continue
            try:
                importlib.import_module(m)
            except ImportError as err:
                print("Can't load {}: {}".format(c, err))
                continue
            if chosen is None:
                chosen = (m, c)
            if c == name:
                chosen = (m, c)
                break
        if chosen is None:
            return None
        return getattr(importlib.import_module(chosen[0]), chosen[1])


class VManager(VManagerBase):
    """Multi-threaded manager (core/vmanager.py:201-458, headless part): one daemon thread per finder;
    when the input is a file both finders receive the same frames in lock step (CaptureReader)."""

    def __init__(self, controller, imqueue=None, bf=None, sf=None, active=True):
        super().__init__(controller, imqueue=imqueue, bf=bf, sf=sf)
        self.daemon = True
        self.processes = []
        self._interrupt_flag = False
        self.active = active
        self.hasrun = False
        self.error = None

    def _get_capture(self):
        cap = open_capture(self.controller.video)
        if not cap.isOpened():
            print("Could not open video: {}".format(getattr(cap, "error", self.controller.video)))
            return None
        return CaptureReader(cap, self)

    def next(self):
        for proc in self.processes:
            proc.next()

    def run(self):
        self.init_capt()
        while not self._interrupt_flag:
            if self.active:
                self.check_video()
                if self.capt is not None:
                    self.check_bf()
                    self.check_sf()
            time.sleep(0.02)
            self.hasrun = True

    def interrupt(self):
        self.stop_processing()
        self._interrupt_flag = True

    def stop_processing(self):
        for proc in list(self.processes):
            proc.interrupt()
        try:
            self.capt.unsync_threads(True)        # release threads a CaptureReader keeps sleeping
        except AttributeError:
            pass

    def check_video(self):
        if self.current_video is not self.controller.video and self.current_video != self.controller.video:
            self.stop_processing()
            self.init_capt()
            self.board_finder = None
            self.stones_finder = None
            self.controller.pipe("video_changed")

    def check_bf(self):
        if self.board_finder is None and self.bf_class is not None:
            self.board_finder = self.bf_class(self)
            self._spawn(self.board_finder)

    def check_sf(self):
        if self.stones_finder is None and self.sf_class is not None:
            self.stones_finder = self.sf_class(self)
            self._spawn(self.stones_finder)

    def is_processing(self):
        return len(self.processes).__bool__()

    def confirm_stop(self, process):
        for vt in list(self.processes):
            if vt.processor is process:
                self.processes.remove(vt)

    def _spawn(self, process):
        vt = VisionThread(process)
        self.processes.append(vt)
        try:
            self.capt.unsync_threads(False)       # processes wait for each other again when reading frames
        except AttributeError:
            pass
        vt.start()


class VManagerSeq(VManagerBase):
    """Board detection until a transform exists, then stones detection to the end of the video;
    everything on the caller's thread."""

    def __init__(self, controller=None, bf=None, sf=None):
        super().__init__(controller, bf=bf, sf=sf)
        self.error = None
        self._stop = False

    def run(self):
        self.init_capt()
        self.board_finder = self.bf_class(self)
        self.board_finder.full_speed = True
        self.stones_finder = self.sf_class(self)
        self.stones_finder.full_speed = True
        bf = self.board_finder
        orig = bf._doframe

        def until_found(frame):
            orig(frame)
            if bf.mtx is not None:
                bf.interrupt()
        bf._doframe = until_found
        bf.execute()
        bf._doframe = orig
        if self._stop or bf.mtx is None:
            return
        self.stones_finder.execute()

    def interrupt(self):
        self.stop_processing()

    def stop_processing(self):
        self._stop = True
        for p in (self.board_finder, self.stones_finder):
            if p is not None:
                p.interrupt()
