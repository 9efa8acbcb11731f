"""Tiny stand-in for the parts of Golib the hot path touches (the reference imports them from
a sibling repository that is not vendored: golib.config.golib_conf.{gsize,E,B,W},
golib.model.Move, a move list with SGF I/O, and a capture-free board).  SURVEY.md section 2."""
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
