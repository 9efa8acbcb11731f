"""Headless controller: the sink the finders talk to (same `pipe` instruction names and board
queries as the reference's vgui/ControllerV and test/objects/ControllerVDev: "append",
"delete", "bulk", "auto_save"; is_empty_blocking, locate, get_stones).  Instructions run
synchronously on the caller's thread.  With rules=True every appended stone goes through the
rule engine (captures are taken off the goban, suicide / ko / occupied raise StateError), as the
reference's controller does through Golib; the default keeps the goban a plain mirror of what the
finders report."""
import numpy as np

from .golib_shim import gsize, E, Kifu, Rule


class ControllerHeadless:
    def __init__(self, video=None, bounds=(0, 1), autosave_path=None, rules=False):
        self.rules = Rule() if rules else None
        self.last_captured = []            # (color, x, y) taken by the most recent append
        self.video = video
        self.bounds = bounds
        self.kifu = Kifu()
        self.board = [[None] * gsize for _ in range(gsize)]      # board[x][y] -> Move
        self._stones = np.full((gsize, gsize), E, dtype=object)  # the same goban in numpy coordinates, kept in step
        self.autosave_path = autosave_path
        self.ignored = set()
        self.api = {"append": self._append, "delete": self._delete, "bulk": self._bulk, "auto_save": self._auto_save}

    def pipe(self, instruction, *args):
        fn = self.api.get(instruction)
        if fn is None:
            self.ignored.add(instruction)
            return
        fn(*args)

    def _append(self, move):
        if self.board[move.x][move.y] is not None:
            raise ValueError("occupied: %s" % move)
        self.last_captured = []
        if self.rules is not None:
            self.last_captured = self.rules.put(move)          # raises StateError when illegal
            for _, cx, cy in self.last_captured:
                self.board[cx][cy] = None                      # prisoners leave the goban, not the record
                self._stones[cy, cx] = E
        self.board[move.x][move.y] = move
        self._stones[move.y, move.x] = move.color
        self.kifu.append(move)

    def _delete(self, x, y):
        mv = self.board[x][y]
        if mv is not None:
            self.board[x][y] = None
            self._stones[y, x] = E
            self.kifu.pop_at(x, y)
            if self.rules is not None:
                self.rules.remove(x, y)
        return mv

    def _bulk(self, moves):
        for mv in moves:
            if mv.color == E:
                self._delete(mv.x, mv.y)
            else:
                self._append(mv)

    def _auto_save(self):
        if self.autosave_path:
            self.kifu.save(self.autosave_path)

    def is_empty_blocking(self, x, y):
        return self.board[x][y] is None

    def locate(self, x, y):
        return self.board[x][y]

    def get_stones(self):
        """copy of the goban in numpy coordinates: stones[r][c] in {'E','B','W'}"""
        return self._stones.copy()
