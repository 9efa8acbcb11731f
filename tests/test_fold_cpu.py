"""The ordered halves of the finders, C++ (camkifu_amd/csrc/ck_fold.cpp through the C-ABI) against the
oracle's scalar Python restatement (oracle/ora_logic.py) on the same seeded inputs: identical corner
decisions and identical request streams, bit for bit.  No GPU: both entry points are host-only."""
import numpy as np
import pytest

from camkifu_amd import capi
from oracle import ora_logic as ol


# ------------------------------------------------------------------------------------------------ board
def _sides(rng, h, w):
    s = np.array([[0.16 * w, 0.05], [0.84 * w, 0.08], [0.12 * h, 1.55], [0.88 * h, 1.62]])
    s[:, 0] += rng.uniform(-0.03, 0.03, 4) * min(h, w)
    s[:, 1] += rng.integers(-2, 3, 4) * np.pi / 180
    return s


def _hough_like(rng, sides, h, w, n_noise):
    """four bundles of near-duplicate (rho, theta) pairs around a slanted quadrilateral + stray lines"""
    out = []
    for s in sides:
        for _ in range(int(rng.integers(0, 4))):
            out.append((np.float32(s[0] + rng.integers(-1, 2)), np.float32(s[1])))
    for _ in range(n_noise):
        out.append((np.float32(rng.uniform(-w, w)), np.float32(rng.integers(0, 180) * np.pi / 180)))
    rng.shuffle(out)
    return np.array(out, np.float32).reshape(-1, 2)


@pytest.mark.parametrize("seed,h,w,noise", [(1, 480, 640, 0), (2, 1080, 1920, 1), (3, 480, 640, 3), (4, 2160, 3840, 2),
                                            (5, 46, 68, 1), (6, 1080, 1920, 6)])
def test_boardfold_matches_oracle(seed, h, w, noise):
    rng = np.random.default_rng(seed)
    core, ora = capi.BoardFoldCore(), ol.BoardLogic()
    hull, n_found = None, 0
    sides = _sides(rng, h, w)
    for f in range(40):
        if f == 20:
            sides = _sides(rng, h, w)                          # the camera is bumped: corners must move
        status = int(rng.choice([0, 0, 0, 0, 0, 0, 1, 2]))
        lines = _hough_like(rng, sides, h, w, noise)
        got = core.step(h, w, status, lines, f, hull)
        try:
            want = ora.step(h, w, status, [tuple(l) for l in lines], f, hull)
        except IndexError:
            want = "IndexError"
        assert got == want, (f, got, want)
        if got[0] and got[1] and len(got[2]) == 4:
            hull, n_found = got[2], n_found + 1
    if noise <= 1 and h >= 480:
        assert n_found >= 1


def test_boardfold_counts_and_degenerate_inputs():
    core = capi.BoardFoldCore()
    assert core.step(480, 640, 1, np.zeros((0, 2), np.float32), 0, None) == (False, False, [], None)
    assert core.step(480, 640, 0, np.zeros((0, 2), np.float32), 1, None) == (False, False, [], None)
    # a frame counter that is a multiple of 4 groups whatever has accumulated, even nothing
    assert core.step(480, 640, 0, np.zeros((0, 2), np.float32), 4, None) == (False, False, [], (0, 0))
    with pytest.raises(capi.CkError):
        core.step(0, 640, 0, np.zeros((0, 2), np.float32), 0, None)


def test_hull_ordering_known_answers():
    """the reference's doctests (core/imgutil.py:244-249, 279-284) through the oracle AND through the C++ fold:
    four single-point groups come back as the ordered hull"""
    import json, os
    ka = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))
    for case in ka["get_ordered_hull"]:
        pts, want = [tuple(p) for p in case["in"]], [tuple(p) for p in case["out"]]
        assert ol.ordered_hull(pts) == want
    for case in ka["cyclic_permute"]:
        assert ol.cyclic_permute([tuple(p) for p in case["in"]]) == [tuple(p) for p in case["out"]]


# ------------------------------------------------------------------------------------------------ stones
class _Goban:
    """the sink both policies talk to: StonesFinder.bulk_update / suggest semantics on a bare array"""

    def __init__(self):
        self.b = np.zeros((19, 19), np.uint8)
        self.log = []

    def apply(self, kind, moves, frame=None):
        self.log.append((frame, kind, tuple(moves)))
        for col, r, c in moves:
            self.b[r, c] = col


def _script(rng, n, bg):
    """a plausible run: a fixed position, stones appearing after a burst of agitation, and mis-reads"""
    truth = np.zeros((19, 19), np.uint8)
    m = rng.random((19, 19)) < 0.25
    truth[m] = rng.integers(1, 3, int(m.sum()))
    rl = np.zeros((n, 10, 10), np.uint8)
    rc = np.zeros((n, 10, 10), np.float64)
    fg = np.zeros((n, 19, 19), np.int32)
    rows = [ol.region_rows(i)[0] for i in range(10)]
    hand_until, hand_cells = -1, []
    for f in range(n):
        if f > bg and rng.random() < 0.08 and f > hand_until:
            r, c = rng.integers(0, 19, 2)
            hand_cells = [(a, b) for a in range(max(0, r - 1), min(19, r + 2)) for b in range(max(0, c - 1), min(19, c + 2))]
            hand_until = f + int(rng.integers(3, 14))
            if truth[r, c] == 0:
                truth[r, c] = rng.integers(1, 3)
        if f <= hand_until:
            for a, b in hand_cells:
                fg[f, a, b] = rng.integers(200, 401)
        elif hand_cells and f <= hand_until + 6:
            a, b = hand_cells[len(hand_cells) // 2]
            fg[f, a, b] = rng.integers(150, 290)                    # the new stone is still foreground for a while
        seen = truth.copy()
        flip = rng.random((19, 19)) < 0.01                          # mis-reads
        seen[flip] = rng.integers(0, 3, int(flip.sum()))
        for i in range(10):
            for j in range(10):
                blk = seen[rows[i]:rows[i] + 2, rows[j]:rows[j] + 2].reshape(4)
                rl[f, i, j] = sum(int(blk[k]) * 3 ** k for k in range(4))
        rc[f] = np.where(rng.random((10, 10)) < 0.1, rng.uniform(0.3, 0.7, (10, 10)), rng.uniform(0.7, 1.0, (10, 10)))
        rc[f, rng.integers(0, 10), rng.integers(0, 10)] = 0.6       # the boundary value itself
    return rl, rc, fg


@pytest.mark.parametrize("seed,bg", [(0, 50), (1, 5), (2, 0), (3, 20), (4, 50)])
def test_policy_matches_oracle(seed, bg):
    rng = np.random.default_rng(100 + seed)
    n = 400
    rl, rc, fg = _script(rng, n, bg)
    # C++ over the whole run at once (the batch fold's shape) ...
    g1 = _Goban()
    core = capi.PolicyCore(bg)
    core.run(0, rl, rc, fg, lambda: g1.b, g1.apply)
    # ... in ragged chunks (the per-frame finder's shape is chunk == 1) ...
    g2 = _Goban()
    core2 = capi.PolicyCore(bg)
    f0 = 0
    while f0 < n:
        k = int(rng.integers(1, 40))
        core2.run(f0, rl[f0:f0 + k], rc[f0:f0 + k], fg[f0:f0 + k], lambda: g2.b,
                  lambda kind, mv, fr, base=f0: g2.apply(kind, mv, fr + base))
        f0 += k
    # ... and the oracle frame by frame
    g3 = _Goban()
    pol = ol.StonePolicy(bg)
    for f in range(n):
        pol.frame(f, rl[f].tolist(), rc[f].tolist(), fg[f].tolist(), lambda: g3.b,
                  lambda req, f=f: g3.apply({"suggest": 1, "bulk": 2}[req[0]], [req[1]] if req[0] == "suggest" else req[1], f))
    assert g1.log == g2.log
    assert g1.log == [(f, k, tuple(tuple(int(x) for x in m) for m in mv)) for f, k, mv in g3.log]
    assert len(g1.log) >= 3 and any(k == 1 for _, k, _ in g1.log)
    # the scripts contain mis-reads: at least one run must have cancelled a stone through lookback
    st = core.state()
    tg = np.array(pol.targets, np.uint8)
    assert np.array_equal(st["targets"], tg) and st["has_sampled"] == pol.sampled
    heat = np.array([[0 if h is None else h.color for h in row] for row in pol.heat], np.uint8)
    assert np.array_equal(st["heat_color"], heat)


def test_lookback_cancels_a_misread():
    """a stone read once with high confidence and never again: checked at f+11 and f+22, cancelled at the second
    failed check (2 <= 0 + 1 is false) -- sf_neural.py:156-176, 219-225"""
    rl = np.zeros((60, 10, 10), np.uint8)
    rc = np.full((60, 10, 10), 0.9)
    rl[3, 1, 1] = 1                                    # frame 3 (the assessment frame): black at (2, 2)
    g = _Goban()
    core = capi.PolicyCore(3)
    core.run(0, rl, rc, None, lambda: g.b, g.apply)
    assert g.log == [(3, 2, ((1, 2, 2),)), (25, 2, ((0, 2, 2),))]
    assert g.b.sum() == 0
    st = core.state()
    assert st["heat_color"].sum() == 0                 # exhausted, aged for 6 frames, forgotten


def test_colour_ratio_veto_and_single_suggest():
    core = capi.PolicyCore(0)
    core.set_sampled(True)
    t = np.zeros((19, 19), np.uint8)
    t[0:2, 0:2] = 20
    core.set_targets(t)
    rl = np.zeros((1, 10, 10), np.uint8)
    rc = np.full((1, 10, 10), 0.9)
    rl[0, 0, 0] = 1 + 3 + 9 + 2 * 27                   # B B B W in one region: |log3(3/1)| = 1 is not < 1
    g = _Goban()
    core.run(5, rl, rc, None, lambda: g.b, g.apply)
    assert g.log == [] and core.state()["targets"].sum() == 0
    t[:] = 0
    t[4, 6] = 17                                       # decays to 16 before the selection
    core.set_targets(t)
    rl[:] = 0
    rl[0, 2, 3] = 2                                    # white at (4, 6)
    core.run(6, rl, rc, None, lambda: g.b, g.apply)
    assert g.log == [(0, 1, ((2, 4, 6),))]             # ONE move -> suggest
    # confidence exactly 0.6 passes the region test (`< 0.6` rejects), below does not
    for conf, expect in ((0.6, 1), (0.5999, 0)):
        g2, c2 = _Goban(), capi.PolicyCore(0)
        c2.set_sampled(True)
        c2.set_targets(t)
        c2.run(6, rl, np.full((1, 10, 10), conf), None, lambda: g2.b, g2.apply)
        assert len(g2.log) == expect


# ------------------------------------------------------------------------------------------------ round 6: the folds over gathered records
def test_round10_in_integers_equals_cpython_round():
    """line_angle rounds its cosine with Python's round(x, 10) (core/imgutil.py:510).  The fold computes it in 128-bit
    integers; held equal to the long way (decimal string and back) and to CPython's own round on random values, on the
    half-way cases k + 0.5 (in units of 1e-10) and their neighbouring doubles, and at the ends of the fast path's range."""
    L = capi.lib()
    rng = np.random.default_rng(1)
    vals = list(rng.uniform(-1.0000001, 1.0000001, 40000)) + list(rng.uniform(-1e-9, 1e-9, 4000)) + list(rng.uniform(-1e6, 1e6, 4000))
    for k in rng.integers(0, 10 ** 10, 8000):
        x = (int(k) + 0.5) / 1e10
        for d in (-2, -1, 0, 1, 2):
            y = x
            for _ in range(abs(d)):
                y = np.nextafter(y, np.inf if d > 0 else -np.inf)
            vals += [float(y), -float(y)]
    vals += [0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 0.99999999995, 0.5e-10, 1.5e-10, 2.5e-10, 5e-324, 1e-300, 524287.99999999995,
             524288.0, 1048576.0, 1e15, 1e22, -1e22]
    for x in vals:
        x = float(x)
        a, b, c = L.ck_round10(x), L.ck_round10_reference(x), round(x, 10)
        assert a == b == c and np.signbit(a) == np.signbit(c), (x, a, b, c)
    assert L.ck_round10(float("inf")) == float("inf") and np.isnan(L.ck_round10(float("nan")))


def _board_records(seed, n, h, w, noise=1):
    from camkifu_amd.pipeline import REC, LMAX, fill_board
    rng = np.random.default_rng(seed)
    sides = _sides(rng, h, w)
    res, lines = np.zeros(n, capi.BOARD_DTYPE), np.zeros((n, LMAX, 2), np.float32)
    for f in range(n):
        if f == n // 2:
            sides = _sides(rng, h, w)
        ls = _hough_like(rng, sides, h, w, noise)[:LMAX]
        res["status"][f] = int(rng.choice([0, 0, 0, 0, 1, 2]))
        res["n_lines"][f] = len(ls)
        lines[f, :len(ls)] = ls
    return fill_board(np.zeros(n, REC), (res, lines))


@pytest.mark.parametrize("seed,h,w,refresh", [(1, 1080, 1920, 50), (2, 480, 640, 3), (3, 1080, 1920, 0), (4, 2160, 3840, 17)])
def test_boardfold_run_over_records_equals_the_frame_by_frame_fold(seed, h, w, refresh):
    """ck_boardfold_run (the loop over a batch's records, the hold-off and the step in the library; Python entered only
    where the corners change) against BoardFold.step frame by frame -- the per-frame finder's own _detect: same corners,
    same transform, same counters after every batch, and the same IndexError where the reference raises one"""
    from camkifu_amd.pipeline import BoardFold
    a, b = BoardFold(h, w, refresh_frames=refresh), BoardFold(h, w, refresh_frames=refresh)
    found = 0
    for batch in range(6):
        recs = _board_records(10 * seed + batch, 97 + 31 * batch, h, w)
        out = []
        for fold, data in ((a, recs), (b, [dict(status=int(r["status"]), n_lines=int(r["n_lines"]), lines=r["lines"]) for r in recs])):
            try:
                fold.run(data)
                out.append("ok")
            except IndexError as why:
                out.append(str(why))
        assert out[0] == out[1]
        assert (a.mtx is None) == (b.mtx is None) and (a.mtx is None or np.array_equal(a.mtx, b.mtx))
        assert a.finder.corners.hull == b.finder.corners.hull
        assert (a.hold, a.seen, a.looked, a.finder.total_f_processed) == (b.hold, b.seen, b.looked, b.finder.total_f_processed)
        if out[0] != "ok":                                   # after the reference's failure both folds stand at the same frame;
            a, b = BoardFold(h, w, refresh_frames=refresh), BoardFold(h, w, refresh_frames=refresh)   # (start over)
        found += a.mtx is not None
    assert found >= 1


def test_policy_over_records_equals_the_policy_over_arrays():
    """ck_policy_run_records reads the classifier's answers from the stones halves of the gathered records where they lie;
    same requests, same state as ck_policy_run on contiguous copies"""
    from camkifu_amd.pipeline import REC
    rng = np.random.default_rng(77)
    n, bg = 300, 20
    rl, rc, fg = _script(rng, n, bg)
    recs = np.zeros(n, REC)
    recs["region_label"], recs["region_conf"] = rl, rc
    recs["lines"] = rng.random((n, 64, 2))                   # (the board half is somebody else's)
    g1, g2 = _Goban(), _Goban()
    c1, c2 = capi.PolicyCore(bg), capi.PolicyCore(bg)
    c1.run(0, rl, rc, fg, lambda: g1.b, g1.apply)
    c2.run(0, None, None, fg, lambda: g2.b, g2.apply, records=recs)
    assert g1.log == g2.log and len(g1.log) >= 3
    s1, s2 = c1.state(), c2.state()
    assert all(np.array_equal(s1[k], s2[k]) for k in s1)


def test_quiet_frames_leave_the_policy_early_with_the_same_state():
    """round 6: a frame with no agitated zone, no live target and no watched prediction costs one pass over its 361
    counts.  Counts just below / at the agitation thresholds of every cell size (400, 380, 361 pixels; 0.5 and 0.7 of the
    area in the reference's float arithmetic) on every cell: the library's integer thresholds against the oracle's floats."""
    bg = 0
    n = 40
    rl = np.zeros((n, 10, 10), np.uint8)
    rc = np.full((n, 10, 10), 0.9)
    fg = np.zeros((n, 19, 19), np.int32)
    levels = [179, 180, 181, 189, 190, 191, 199, 200, 201, 252, 253, 265, 266, 267, 279, 280, 281]
    for f in range(2, 2 + len(levels)):
        fg[f] = levels[f - 2]
    g1, g3 = _Goban(), _Goban()
    core, pol = capi.PolicyCore(bg), ol.StonePolicy(bg)
    for f in range(n):                                       # frame by frame: the targets are compared after every frame
        core.run(f, rl[f:f + 1], rc[f:f + 1], fg[f:f + 1], lambda: g1.b, lambda kind, mv, fr, f=f: g1.apply(kind, mv, fr + f))
        pol.frame(f, rl[f].tolist(), rc[f].tolist(), fg[f].tolist(), lambda: g3.b,
                  lambda req, f=f: g3.apply({"suggest": 1, "bulk": 2}[req[0]], [req[1]] if req[0] == "suggest" else req[1], f))
        assert np.array_equal(core.state()["targets"], np.array(pol.targets, np.uint8)), f
    assert core.state()["targets"].max() == 0 and len(g1.log) == len(g3.log)
