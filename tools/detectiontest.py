#!/usr/bin/env python3
"""Headless DetectionTest (mirror of the reference's test/mains/detectiontest.py / benchmark.py):
a video (frames saved with np.save, or a synthetic clip) + a reference SGF -> move-sequence
match ratio.  Runs the drop-in finders (BoardFinderAuto + SfNeural) on the HIP library.

    python tools/detectiontest.py --synthetic 640x480 --frames 200
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
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--bf", default=None)
    ap.add_argument("--sf", default=None)
    ap.add_argument("--failfast", action="store_true")
    args = ap.parse_args()
    if args.synthetic:
        # a filmed game: a position, then a move every 30 frames with the player's hand over the point first
        # (synth.film); the reference game = that position in the order the first assessment reports it, then the moves
        w, h = map(int, args.synthetic.lower().split("x"))
        film, corners, truth, moves, hands = synth.film(args.frames, h, w, seed=synth.SEED, quiet=62, move_every=30,
                                                         hand_frames=12)
        frames = film.numpy()
        ref = Kifu()
        for r in range(19):
            for c in range(19):
                if truth[0][r, c]:
                    ref.append(Move(NP_TYPE, ("EBW"[truth[0][r, c]], r, c)))
        for col, r, c, f in moves:
            if f + 14 < args.frames:                    # the finder needs the stone to settle into the background
                ref.append(Move(NP_TYPE, ("EBW"[col], r, c)))
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
    got, want = set(matcher.b), set(matcher.a)
    print("moves: %d recorded, %d of the %d reference moves among them, %d not in the reference"
          % (len(matcher.b), len(got & want), len(want), len(got - want)))
    return matcher.ratio()


if __name__ == "__main__":
    main()
