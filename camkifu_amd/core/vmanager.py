"""Vision managers: finder registry/reflection and a sequential headless driver (mirror of the
reference's core/vmanager.py:16-198 `VManagerBase` and test/objects/vmanager_test.py `VManagerSeq`).
Video decoding is out of scope: frames come from an in-memory array or a
.npy file through ArrayCapture, which plays the role of CaptureReaderBase."""
import importlib
import threading

import numpy as np

from .. import cvconf


class ArrayCapture:
    """Frame source over an (n, h, w, 3) uint8 array (or a path to one saved with np.save).
    read() hands each consumer a private copy, like CaptureReader.read_file does."""

    def __init__(self, frames):
        if isinstance(frames, str):
            frames = np.load(frames, mmap_mode="r")
        self.frames = frames
        self.pos = 0

    def read(self, caller=None):
        if self.pos >= len(self.frames):
            return False, None
        frame = np.array(self.frames[self.pos], copy=True)
        self.pos += 1
        return True, frame

    def progress(self):
        return self.pos / max(1, len(self.frames))

    def seek(self, ratio):
        self.pos = int(ratio * len(self.frames))

    def release(self):
        pass


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
        self.capt = ArrayCapture(self.controller.video)
        self.full_speed = True
        self.capt.seek(self.controller.bounds[0])

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
