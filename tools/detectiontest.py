#!/usr/bin/env python3
"""Headless DetectionTest (mirror of the reference's test/mains/detectiontest.py / benchmark.py):
a video (frames saved with np.save, or a synthetic clip) + a reference SGF -> move-sequence
match ratio.  Runs the drop-in finders (BoardFinderAuto + SfNeural) on the HIP library.

    python tools/detectiontest.py --synthetic 640x480 --frames 120
    python tools/detectiontest.py -v clip.npy --sgf game.sgf [--bf BoardFinderAuto --sf SfNeural]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camkifu_amd import synth  # noqa: E402
from camkifu_amd.controller import ControllerHeadless  # noqa: E402
from camkifu_amd.core.vmanager import VManagerSeq  # noqa: E402
from camkifu_amd.golib_shim import Kifu, Move, NP_TYPE  # noqa: E402
from camkifu_amd.kifu_checker import KifuChecker, report  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-v", "--video", help=".npy file holding (n,h,w,3) uint8 BGR frames")
    ap.add_argument("--sgf", help="reference SGF")
    ap.add_argument("--synthetic", default=None, help="WxH: render a clip and its reference game instead")
    ap.add_argument("--frames", type=int, default=120)
    ap.add_argument("--bf", default=None)
    ap.add_argument("--sf", default=None)
    ap.add_argument("--failfast", action="store_true")
    args = ap.parse_args()
    if args.synthetic:
        w, h = map(int, args.synthetic.lower().split("x"))
        rng = np.random.default_rng(synth.SEED)
        corners = synth.random_corners(h, w, rng)
        stones = synth.random_stones(rng, density=0.3)
        frames = np.stack([synth.render(h, w, stones, corners, seed=synth.SEED + f).numpy() for f in range(args.frames)])
        ref = Kifu()
        for r in range(19):                       # reference = the position, in the order predict_all reports it
            for c in range(19):
                if stones[r, c]:
                    ref.append(Move(NP_TYPE, ("EBW"[stones[r, c]], r, c)))
        name = "synthetic-%dx%d" % (w, h)
    else:
        frames, ref, name = np.load(args.video, mmap_mode="r"), Kifu(sgffile=args.sgf), os.path.basename(args.video)
    ctrl = ControllerHeadless(video=frames)
    vm = VManagerSeq(ctrl, bf=args.bf, sf=args.sf)
    t0 = time.time()
    vm.run()
    if getattr(vm, "error", None) is not None:
        raise vm.error
    matcher = KifuChecker(ref, failfast=args.failfast).check(ctrl.kifu)
    print(report(name, matcher, time.time() - t0))
    return matcher.ratio()


if __name__ == "__main__":
    main()
