"""Move-sequence scoring against a reference SGF, as the reference's test harness does
(test/objects/kifu_checker.py:8-40: difflib.SequenceMatcher over the two move sequences;
test/mains/benchmark.py:89-101 prints `[name: X% in Y s]`).  No Go rules: captures are not
replayed."""
import difflib

from .golib_shim import Kifu


class KifuChecker:
    def __init__(self, ref_sgf, failfast=False, bounds=(0, 1000)):
        self.ref = ref_sgf if isinstance(ref_sgf, Kifu) else Kifu(sgffile=ref_sgf)
        self.failfast = failfast
        self.bounds = bounds

    def check(self, found):
        """found: Kifu recorded by the controller -> difflib.SequenceMatcher(a=reference, b=found)"""
        f, last = self.bounds
        ref = [repr(m) for m in self.ref.get_move_seq(first=f, last=last)]
        seq = [repr(m) for m in found.get_move_seq(first=f, last=last)]
        if self.failfast:
            for i, (a, b) in enumerate(zip(ref, seq), 1):
                if a != b:
                    raise AssertionError("At move {0}: expected {1}, got {2}.  (failfast mode is on)".format(i, a, b))
        return difflib.SequenceMatcher(a=ref, b=seq)


def report(name, matcher, seconds):
    return "[{}: {:.1f}% in {:.1f} s]".format(name, 100 * matcher.ratio(), seconds)
