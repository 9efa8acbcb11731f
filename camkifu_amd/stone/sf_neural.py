"""SfNeural: the CNN stones finder (mirror of the reference's stone/sf_neural.py:26-244).

Phase machine of `_find`: frame 0 loads the net; frames < bg_init_frames only feed the
background model; then one full-board assessment (predict_all); afterwards foreground
agitation marks targets, calm 2x2 regions with hot targets are re-predicted, and recent
predictions are re-checked (lookback).  All numeric work (K10..K12) is one GPU call per frame
through NNCache; the policy code below is the reference's, quirks included."""
import math

import numpy as np

from ..core.exceptions import DeletedError
from ..golib_shim import gsize, E, B, W, Move, NP_TYPE, KGS_TYPE
from .nn_cache import NNCache
from .nn_manager import NNManager
from .stonesfinder import StonesFinder

COLD = 'cold'
WIN_NAME = 'Neural'
MIN_CONFIDENCE = 0.6
TARGET_THRESH = 15
TARGET_INCR = 5
NB_LOOKBACK = 3


class SfNeural(StonesFinder):
    def __init__(self, vmanager, ctx=None):
        super().__init__(vmanager, learn_bg=True, ctx=ctx)
        self.manager = NNManager()
        self.cache = None
        self.has_sampled = False
        self.indices = self.manager.class_indices()
        self.targets = np.zeros((gsize, gsize), dtype=np.uint8)
        self.heatmap = np.full((gsize, gsize), None, dtype=object)
        self._net_loaded = False

    def _load_net(self):
        if not self._net_loaded:
            self.ctx.cnn_set_weights(self.manager.get_net(download=True))
            self._net_loaded = True

    def _find(self, goban_img):
        self.cache = NNCache(self.manager, goban_img, self.ctx)
        if self.total_f_processed == 0:
            self._load_net()
        elif self.total_f_processed < self.bg_init_frames:
            pass                                   # background sampling
        elif not self.has_sampled:
            self._load_net()
            self.predict_all()
            self.has_sampled = True
        else:
            self.mark_targets()
            self.process_targets()
            self.lookback()

    def predict_all(self):
        stones = self.cache.predict_all_stones()
        moves = []
        for r in range(gsize):
            for c in range(gsize):
                color, confidence = stones[r, c]
                if color != E and confidence > MIN_CONFIDENCE:
                    moves.append((color, r, c))
                    self.heatmap[r, c] = HeatPoint(color, confidence, self.total_f_processed)
        self.bulk_update(moves)

    def mark_targets(self):
        fg = self.get_foreground()
        for r in range(gsize):
            for c in range(gsize):
                if self.heatmap[r, c] is None and self.is_agitated(r, c, fg):
                    self.targets[r, c] += TARGET_INCR
        self.targets[np.where(self.targets > 0)] -= 1        # decay

    def process_targets(self):
        moves = self.predict_moves(self.select_targets())
        if len(moves) and self.get_color_ratio(moves) < 1:
            for (color, r, c, confidence) in moves:
                self.heatmap[r, c] = HeatPoint(color, confidence, self.total_f_processed)
            if len(moves) == 1:
                try:
                    self.suggest(*moves.pop()[0:3], doprint=False)
                except DeletedError as de:
                    print(de)
            else:
                self.bulk_update([m[0:3] for m in moves])

    def predict_moves(self, targets):
        moves = set()
        if not len(targets):
            return moves
        stones = self.get_stones()
        for i, j in targets:
            new_stones, confidence = self.cache.predict_4_stones(i, j)
            if confidence < MIN_CONFIDENCE:
                continue
            rs, re, cs, ce = self.manager._subregion(i, j)
            for a, b in np.transpose(np.where(new_stones != E)):
                r, c = a + rs, b + cs
                prev_color, new_color = stones[r, c], new_stones[a, b]
                if prev_color == E:
                    moves.add((new_color, r, c, confidence))
                elif prev_color != new_color:
                    loc = Move(NP_TYPE, (prev_color, r, c)).get_coord(KGS_TYPE)
                    print("Err.. hum. Now seeing {} instead of {} at {}".format(new_color, prev_color, loc))
        return moves

    def select_targets(self):
        """regions holding a hot target and no agitated intersection right now"""
        fg = self.get_foreground()
        targets = []
        for i in range(self.manager.split):
            for j in range(self.manager.split):
                rs, re, cs, ce = self.manager._subregion(i, j)
                if not (self.targets[rs:re, cs:ce] > TARGET_THRESH).any():
                    continue
                agitated = any(self.is_agitated(a, b, fg, ratio=0.5)
                               for a in range(rs, re) for b in range(cs, ce))
                if not agitated:
                    targets.append((i, j))
                    self.targets[rs:re, cs:ce] = 0
        return targets

    def lookback(self):
        stones = self.get_stones()
        to_del = []
        for r in range(gsize):
            for c in range(gsize):
                hpoint = self.heatmap[r, c]
                if hpoint is None or not (0 < hpoint.energy):
                    continue
                if hpoint.color != stones[r, c]:
                    self.heatmap[r, c] = None            # somebody else changed this location
                    continue
                if 10 < self.total_f_processed - hpoint.stamp:
                    hpoint.stamp = self.total_f_processed
                    hpoint.check(*self.cache.predict_stone(r, c))
                    if not hpoint.is_valid():
                        to_del.append((E, r, c))
        if len(to_del):
            self.bulk_update(to_del)
        for r in range(gsize):
            for c in range(gsize):
                hp = self.heatmap[r, c]
                if hp is not None and hp.is_cold():
                    self.heatmap[r, c] = None
        # the reference then draws str(value) of every heat point on its debug canvas, and
        # HeatPoint.__repr__ ages exhausted points (sf_neural.py:238-244): keep that ageing
        for r in range(gsize):
            for c in range(gsize):
                if self.heatmap[r, c] is not None:
                    repr(self.heatmap[r, c])

    def is_agitated(self, r, c, fg, ratio=0.7):
        a0, b0, a1, b1 = self.getrect(r, c)
        return (a1 - a0) * (b1 - b0) * ratio < np.sum(fg[a0:a1, b0:b1]) / 255

    @staticmethod
    def get_color_ratio(moves):
        count = {B: 0, W: 0}
        for m in moves:
            if m[0] != E:
                count[m[0]] += 1
        if 0 in count.values():
            count[B] += 1
            count[W] += 1
        return abs(math.log(count[B] / count[W], 3))

    def _window_name(self):
        return WIN_NAME


class HeatPoint:
    """a recent prediction under watch: re-checked NB_LOOKBACK times, cancelled as soon as it
    can no longer pass two thirds of them"""

    def __init__(self, color, confidence, stamp, energy=NB_LOOKBACK):
        self.target = energy
        self.energy = energy
        self.color = color
        self.confidence = confidence
        self.stamp = stamp
        self.nb_checks = 0
        self.nb_passed = 0

    def check(self, color, confidence):
        self.nb_checks += 1
        self.energy -= 1
        new_conf = 0
        if color == self.color:
            self.nb_passed += 1
            new_conf = confidence
        self.confidence = (self.confidence * self.nb_checks + new_conf) / (self.nb_checks + 1)

    def is_valid(self):
        can_pass = 2 * self.target / 3 <= self.nb_passed + self.energy
        if not can_pass:
            self.energy = 0
            self.confidence = 0.0
        return can_pass

    def is_cold(self):
        return self.energy < -5

    def __repr__(self):
        if self.energy <= 0:
            self.energy -= 1
        return '{:d}'.format(int(self.confidence * 10)) if not self.is_cold() else ''
