"""Tiny stand-in for the parts of Golib the hot path touches (the reference imports them from
a sibling repository that is not vendored: golib.config.golib_conf.{gsize,E,B,W},
golib.model.Move, a move list with SGF I/O, and the rule engine behind controller.rules:
captures, suicide and simple ko).  SURVEY.md section 2 and 8f rank 4."""
gsize = 19
E, B, W = 'E', 'B', 'W'
NP_TYPE, KGS_TYPE, SGF_TYPE = 'np', 'kgs', 'sgf'
_KGS_COLS = "ABCDEFGHJKLMNOPQRST"


class Move:
    """A stone event.  ctuple for NP_TYPE is (color, row, col) in numpy coordinates; internally
    x = column, y = row (screen coordinates), like the reference's Move('np', (color, r, c))."""

    def __init__(self, ctype, ctuple=None, number=-1):
        if ctype != NP_TYPE:
            raise NotImplementedError("only numpy-coordinate moves are needed on the hot path")
        color, r, c = ctuple
        self.color = color
        self.x, self.y = int(c), int(r)
        self.number = number

    def get_coord(self, ctype=SGF_TYPE):
        if ctype == NP_TYPE:
            return self.y, self.x
        if ctype == KGS_TYPE:
            return "%s%d" % (_KGS_COLS[self.x], gsize - self.y)
        return chr(97 + self.x) + chr(97 + self.y)

    def sgf(self):
        return ";%s[%s]" % (self.color, self.get_coord(SGF_TYPE)) if self.color in (B, W) else ""

    def __eq__(self, o):
        return isinstance(o, Move) and (self.color, self.x, self.y) == (o.color, o.x, o.y)

    def __hash__(self):
        return hash((self.color, self.x, self.y))

    def __repr__(self):
        return "%s[%s]" % (self.color, self.get_coord(KGS_TYPE))


class Kifu:
    """Main-line move sequence with minimal SGF output/input (no variations, no captures)."""

    def __init__(self, sgffile=None):
        self.moves = []
        if sgffile:
            self.load(sgffile)

    def append(self, move):
        move.number = len(self.moves) + 1
        self.moves.append(move)

    def pop_at(self, x, y):
        for i in range(len(self.moves) - 1, -1, -1):
            if (self.moves[i].x, self.moves[i].y) == (x, y):
                return self.moves.pop(i)
        return None

    def get_move_seq(self, first=0, last=1000):
        return self.moves[max(first - 1, 0):last]

    def to_sgf(self):
        return "(;GM[1]FF[4]SZ[%d]%s)" % (gsize, "".join(m.sgf() for m in self.moves))

    def save(self, path):
        with open(path, "w") as f:
            f.write(self.to_sgf())

    def load(self, path):
        import re
        txt = open(path).read()
        for col, xy in re.findall(r";\s*([BW])\[([a-s]{2})\]", txt):
            self.append(Move(NP_TYPE, (col, ord(xy[1]) - 97, ord(xy[0]) - 97)))


class StateError(Exception):
    """an instruction that the rules of Go do not allow in the current position"""


class Rule:
    """Board state under the rules of Go (what the reference gets from Golib's rule object behind
    controller.rules, vgui/controllerv.py + test/objects/controllerv_test.py:40-45 `rules.stones`):
    a move captures the opponent groups it leaves without liberties, suicide and the immediate
    retaking of a simple ko are refused.  stones[x][y] in {E, B, W}, x = column, y = row."""

    def __init__(self):
        self.stones = [[E] * gsize for _ in range(gsize)]
        self.ko = None                 # (x, y) the next move may not retake
        self.deads = {B: 0, W: 0}      # prisoners taken FROM each colour

    @staticmethod
    def _neighbours(x, y):
        if x > 0:
            yield x - 1, y
        if x < gsize - 1:
            yield x + 1, y
        if y > 0:
            yield x, y - 1
        if y < gsize - 1:
            yield x, y + 1

    def group(self, x, y):
        """-> (stones of the group holding (x, y), its liberties) as two sets"""
        color = self.stones[x][y]
        grp, libs, todo = {(x, y)}, set(), [(x, y)]
        while todo:
            cx, cy = todo.pop()
            for nx, ny in self._neighbours(cx, cy):
                s = self.stones[nx][ny]
                if s == E:
                    libs.add((nx, ny))
                elif s == color and (nx, ny) not in grp:
                    grp.add((nx, ny))
                    todo.append((nx, ny))
        return grp, libs

    def put(self, move):
        """play `move`; returns the list of captured (color, x, y); raises StateError when illegal"""
        x, y, color = move.x, move.y, move.color
        if color not in (B, W):
            raise StateError("not a stone: %r" % (move,))
        if self.stones[x][y] != E:
            raise StateError("occupied: %r" % (move,))
        if self.ko == (x, y):
            raise StateError("ko: %r" % (move,))
        enemy = W if color == B else B
        self.stones[x][y] = color
        captured = []
        for nx, ny in self._neighbours(x, y):
            if self.stones[nx][ny] == enemy:
                grp, libs = self.group(nx, ny)
                if not libs:
                    for gx, gy in sorted(grp):
                        self.stones[gx][gy] = E
                        captured.append((enemy, gx, gy))
        grp, libs = self.group(x, y)
        if not libs:
            self.stones[x][y] = E
            raise StateError("suicide: %r" % (move,))
        # simple ko: one stone taken by a lone stone that is itself left in atari on that point
        self.ko = None
        if len(captured) == 1 and len(grp) == 1 and libs == {(captured[0][1], captured[0][2])}:
            self.ko = (captured[0][1], captured[0][2])
        self.deads[enemy] += len(captured)
        return captured

    def remove(self, x, y):
        """take a stone off the board (a user correction, not a capture)"""
        if self.stones[x][y] == E:
            raise StateError("empty: (%d, %d)" % (x, y))
        self.stones[x][y] = E
        self.ko = None
