"""SURVEY 8f rank 3 without a GPU: the oracle restatement of SfContours.find_stones (oracle/ora_stones.py) checked
piece by piece against definitions written a second way (brute force), and the SfContours finder class driven through
the sequential harness with every library call answered by the oracle (tests/stub_ctx.py)."""
import heapq

import numpy as np
import pytest

from oracle import ora_stones as S
from oracle import oracle as O


def test_opening_is_the_four_row_min_then_max():
    rng = np.random.default_rng(3)
    fg = (rng.random((40, 33)) < 0.7).astype(np.uint8) * 255
    fg[10:30, 5:20] = 255
    got = S.morph_open_rows(fg)
    h = fg.shape[0]
    er = np.empty_like(fg)
    for y in range(h):
        er[y] = fg[max(0, y - 3):y + 1].min(0)
    want = np.empty_like(fg)
    for y in range(h):
        want[y] = er[max(0, y - 3):y + 1].max(0)
    assert np.array_equal(got, want)
    assert got[10:13, 5:20].min() == 0 or fg[7:10, 5:20].min() == 255       # the opening shifts blobs down, it does not centre them
    assert not S.morph_open_rows(np.zeros((8, 8), np.uint8)).any()
    assert S.morph_open_rows(np.full((8, 8), 255, np.uint8)).all()          # borders are ignored, not zero


@pytest.mark.parametrize("seed", range(6))
def test_filled_hull_against_the_geometric_definition(seed):
    """every pixel centre strictly inside the polygon is painted, nothing farther than one pixel outside is, the
    vertices and the bounding rows / columns are touched"""
    rng = np.random.default_rng(seed)
    pts = rng.integers(2, 60, (int(rng.integers(3, 25)), 2))
    hull = S.convex_hull(pts)
    if len(hull) < 3:
        pytest.skip("degenerate draw")
    img = np.zeros((64, 64), np.uint8)
    S.fill_polygon(img, hull, 1)
    hx, hy = hull[:, 0].astype(float), hull[:, 1].astype(float)
    nx, ny = np.roll(hx, -1), np.roll(hy, -1)
    ys, xs = np.mgrid[0:64, 0:64]
    cross = (nx - hx)[:, None, None] * (ys[None] - hy[:, None, None]) - (ny - hy)[:, None, None] * (xs[None] - hx[:, None, None])
    length = np.hypot(nx - hx, ny - hy)[:, None, None]
    signed = cross / length                                    # distance to each side's line, same sign inside
    if signed[:, int(hy.mean()), int(hx.mean())].mean() < 0:
        signed = -signed
    inside = (signed > 1e-9).all(0)
    far_out = (signed < -1.0).any(0)
    assert img[inside].all()
    assert not img[far_out].any()
    assert all(img[y, x] for x, y in hull)
    x0, y0, w, h = S.bounding_rect(hull)
    on = np.argwhere(img)
    assert on[:, 0].min() == y0 and on[:, 0].max() == y0 + h - 1 and on[:, 1].min() == x0 and on[:, 1].max() == x0 + w - 1
    shifted = np.zeros((64, 64), np.uint8)                     # the offset argument is a pure translation
    S.fill_polygon(shifted, hull, 1, offset=(-x0, -y0))
    assert np.array_equal(shifted[:h, :w], img[y0:y0 + h, x0:x0 + w]) and shifted.sum() == img.sum()


def test_line_iterator_endpoints_and_symmetry():
    rng = np.random.default_rng(0)
    for _ in range(200):
        a, b = tuple(rng.integers(0, 40, 2)), tuple(rng.integers(0, 40, 2))
        px = S.line_pixels(a, b)
        assert set(px) == set(S.line_pixels(b, a))             # always walked left to right
        assert tuple(map(int, a)) in px and tuple(map(int, b)) in px
        assert len(px) == max(abs(a[0] - b[0]), abs(a[1] - b[1])) + 1
        for (x0, y0), (x1, y1) in zip(px, px[1:]):
            assert max(abs(x1 - x0), abs(y1 - y0)) == 1        # 8-connected


def _chamfer_exact(img):
    """shortest paths with the 5x5 chamfer steps (Dijkstra): the metric the two raster passes approximate from above"""
    h, w = img.shape
    a, b, c = 65536, 91750, 143976
    steps = [(0, 1, a), (1, 0, a), (0, -1, a), (-1, 0, a), (1, 1, b), (1, -1, b), (-1, 1, b), (-1, -1, b)]
    steps += [(dy, dx, c) for dy, dx in ((1, 2), (2, 1), (-1, 2), (-2, 1), (1, -2), (2, -1), (-1, -2), (-2, -1))]
    dist = np.full((h, w), 1 << 40, np.int64)
    heap = [(0, y, x) for y, x in np.argwhere(img == 0)]
    for _, y, x in heap:
        dist[y, x] = 0
    heapq.heapify(heap)
    while heap:
        d, y, x = heapq.heappop(heap)
        if d > dist[y, x]:
            continue
        for dy, dx, cost in steps:
            yy, xx = y + dy, x + dx
            if 0 <= yy < h and 0 <= xx < w and d + cost < dist[yy, xx]:
                dist[yy, xx] = d + cost
                heapq.heappush(heap, (d + cost, yy, xx))
    return dist


@pytest.mark.parametrize("seed", range(4))
def test_distance_transform_bounds(seed):
    rng = np.random.default_rng(seed)
    img = np.full((28, 35), 255, np.uint8)
    for y, x in rng.integers(0, 28, (4, 2)):
        img[y, x] = 0
    img[5, 3:20] = 0
    got = S.distance_transform_5x5(img)
    fixed = np.round(got.astype(np.float64) * 65536).astype(np.int64)
    assert np.array_equal(fixed / 65536.0, got.astype(np.float64))                # exact multiples of 2^-16
    exact = _chamfer_exact(img)
    assert (fixed >= exact).all() and (fixed == exact).mean() > 0.98              # two passes: never below, almost always equal
    ys, xs = np.argwhere(img == 0).T
    yy, xx = np.mgrid[0:28, 0:35]
    euclid = np.sqrt(((yy[..., None] - ys) ** 2 + (xx[..., None] - xs) ** 2).min(-1))
    assert np.abs(got - euclid).max() <= 0.03 * euclid.max() + 0.5                # a 2 % chamfer metric
    assert (got[img == 0] == 0).all()


def test_find_centers_cells():
    d = np.zeros((40, 40), np.float32)
    d[10, 9] = 5.0                                   # the maximum of the top-left cell sits near that cell's centre
    assert S.find_centers(d, 10.0) and S.find_centers(d, 10.0)[0] == (9, 10)
    e = np.zeros((40, 40), np.float32)
    e[0, 0] = e[0, 39] = e[39, 0] = e[39, 39] = 5.0  # every cell's maximum hugs a wall
    assert S.find_centers(e, 10.0) == []
    with pytest.raises(ZeroDivisionError):           # what the reference does with a box thinner than a stone radius
        S.find_centers(np.zeros((8, 40), np.float32), 10.0)


def test_min_area_rect_box_conventions():
    w, h, a = S.min_area_rect_box([(0, 0), (10, 0), (10, 5), (0, 5)])
    assert sorted((w, h)) == [5.0, 10.0] and a in (-90.0, 0.0, -0.0)
    w, h, a = S.min_area_rect_box([(0, 0), (10, 10), (5, 15), (-5, 5)])
    assert abs(min(w, h) - 50 ** 0.5) < 1e-5 and abs(max(w, h) - 200 ** 0.5) < 1e-5 and abs(abs(a) - 45) < 1e-4
    assert S.min_area_rect_box([(3, 3)]) == (0.0, 0.0, 0.0)
    w, h, a = S.min_area_rect_box([(0, 0), (3, 4)])
    assert (w, h) == (5.0, 0.0) and abs(a - 53.13010235) < 1e-4


def test_find_color_rules():
    z = np.zeros((3, 3, 4), np.int16)
    z[:, :, 1:] = 100                                # bare wood everywhere, nothing masked
    z[1, 1] = (1, 20, 20, 20)                        # a dark zone under a hull
    st = np.zeros((3, 3), np.uint8)
    S.find_color(1, 1, z, st)
    assert st[1, 1] == S.B                           # three empty neighbours, each > 100 brighter
    z[1, 1, 1:] = 240
    st[:] = 0
    S.find_color(1, 1, z, st)
    assert st[1, 1] == S.W
    z[1, 1, 1:] = 110                                # close to an empty neighbour: E, and the search stops
    st[:] = 0
    S.find_color(1, 1, z, st)
    assert st[1, 1] == S.E
    z[1, 1, 1:] = 20                                 # an already found ally to the west, within 10 %
    z[1, 0] = (1, 21, 20, 20)
    z[0, :, 0] = 1
    z[0, :, 1:] = 20
    st[:] = 0
    st[0, :] = S.B
    st[1, 0] = S.B
    S.find_color(1, 1, z, st)
    assert st[1, 1] == S.B


@pytest.fixture(scope="module")
def quiet_game():
    from camkifu_amd import synth
    film, corners, truth, moves, hands = synth.film(60, 480, 640, seed=synth.SEED + 1, quiet=60, move_every=1000)
    frames = film.numpy()
    M = O.get_perspective_transform(corners, np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32))
    model = O.MOG2(380, 380, 3)
    for f in range(55):
        gob = O.warp_perspective(frames[f], M)
        fg = model.apply(gob, 0.01)
    return frames, corners, truth[54], gob, fg


def test_the_method_finds_the_stones_of_a_still_position(quiet_game):
    _, _, truth, gob, fg = quiet_game
    stones, zones, mask, info = S.find_stones(gob, fg, want_all=True)
    assert (truth > 0).sum() > 20 and (stones == truth).mean() >= 0.98
    assert mask.shape == (379, 379) and zones.shape == (19, 19, 4) and len(info["img"]) >= 20      # (touching stones share a contour)
    sub = S.find_stones(gob, fg, 6, 13, 6, 13)
    assert not sub[:6].any() and not sub[:, 13:].any()
    assert (sub[6:13, 6:13] == truth[6:13, 6:13]).mean() >= 0.95


def test_sfcontours_through_the_sequential_harness(quiet_game, monkeypatch):
    from camkifu_amd import capi, cvconf
    from camkifu_amd.controller import ControllerHeadless
    from camkifu_amd.core.vmanager import VManagerBase, VManagerSeq
    from camkifu_amd.golib_shim import B, W, E
    from .stub_ctx import OracleCtx
    frames, corners, truth, _, _ = quiet_game
    assert VManagerBase._reflect("SfContours", cvconf.sfinders).__name__ == "SfContours"
    ctx = OracleCtx()
    monkeypatch.setattr(capi, "Context", lambda device=0: ctx)
    monkeypatch.setattr(capi, "get_perspective_transform", O.get_perspective_transform)
    controller = ControllerHeadless(video=frames)
    vm = VManagerSeq(controller, sf="SfContours")
    vm.run()
    assert getattr(vm, "error", None) is None
    sf = vm.stones_finder
    assert type(sf).__name__ == "SfContours" and sf.total_f_processed > sf.bg_init_frames
    got = sf.last_stones
    assert got is not None and got.shape == (19, 19) and set(np.unique(got)) <= {B, W, E}
    want = np.array([E, B, W], dtype=object)[truth]
    assert (got == want).mean() >= 0.97
    assert not controller.kifu.moves                  # as in the reference, the finder on its own submits nothing


# ---- the grid-line search: find_intersections / update_grid / PosGrid.learn -------------------------------------
def test_cv_rng_first_draws():
    """cv::RNG((uint64)-1): multiply-with-carry with 4164903690; the first state is computed by hand"""
    from oracle import ora_grid as G
    r = G.CvRNG()
    first = r.next()
    assert r.state == (0xFFFFFFFF * 4164903690 + 0xFFFFFFFF) & 0xFFFFFFFFFFFFFFFF and first == r.state & 0xFFFFFFFF
    draws = [G.CvRNG().uniform(0, k) for k in (1, 2, 10)]
    assert draws[0] == 0 and all(0 <= d < k for d, k in zip(draws, (1, 2, 10)))


def test_hough_lines_p_on_simple_zones():
    from oracle import ora_grid as G
    z = np.zeros((20, 20), np.uint8)
    z[7, :] = 255
    assert G.hough_lines_p(z, 15, 13) == [(0, 7, 19, 7)]
    z[:] = 0
    z[:, 12] = 255
    (line,) = G.hough_lines_p(z, 15, 13)
    assert {line[:2], line[2:]} == {(12, 0), (12, 19)}
    z[:] = 0
    z[7, 3:12] = 255                                    # too short for the vote threshold
    assert G.hough_lines_p(z, 15, 13) == []
    z[:] = 0
    z[5, :] = 255
    z[:, 5] = 255                                        # a cross: the second line lost its crossing pixel to the first
    lines = G.hough_lines_p(z, 15, 13)
    assert 1 <= len(lines) <= 2 and all(abs(l[2] - l[0]) >= 13 or abs(l[3] - l[1]) >= 13 for l in lines)


def test_update_grid_library_matches_oracle():
    """ck_update_grid (host only: no GPU needed) against the oracle on random line sets, every branch taken"""
    from camkifu_amd import capi
    from oracle import ora_grid as G
    rng = np.random.default_rng(4)
    box = (40, 60, 60, 80)
    branches = set()
    for _ in range(400):
        k = int(rng.integers(1, 7))
        lines = []
        for _ in range(k):
            kind = rng.integers(0, 4)
            a, b = int(rng.integers(0, 6)), int(rng.integers(14, 20))
            t = int(rng.integers(0, 20))
            if kind == 0:
                lines.append((a, t, b, t + int(rng.integers(-1, 2))))
            elif kind == 1:
                lines.append((t, a, t + int(rng.integers(-1, 2)), b))
            elif kind == 2:
                lines.append((a, a, b, b))
            else:
                lines.append(tuple(int(v) for v in rng.integers(0, 20, 4)))
        lines = [l for l in lines if (l[0], l[1]) != (l[2], l[3])]
        if not lines:
            continue
        want = np.array([50, 70], np.int16)
        G.update_grid(lines, box, want)
        got = np.array([50, 70], np.int16)
        capi.update_grid(np.array(lines, np.int32).reshape(-1, 1, 4), box, got)
        assert np.array_equal(got, want), (lines, got, want)
        branches.add("same" if (want == (50, 70)).all() else "negated" if (want == (-50, -70)).all() else "moved")
    assert branches == {"same", "negated", "moved"}
    with pytest.raises(ZeroDivisionError):
        capi.update_grid([(3, 3, 3, 3)], box, np.array([50, 70], np.int16))


def test_posgrid_learn_matches_oracle():
    from camkifu_amd.stone.stonesfinder import PosGrid
    from oracle import ora_grid as G
    pg, st = PosGrid(380), G.GridState(O.posgrid(380))
    assert np.array_equal(pg.mtx, st.mtx)
    rng = np.random.default_rng(2)
    for step in range(12):
        grid = pg.mtx.copy()
        pick = rng.random((19, 19)) < 0.03 * (1 + step % 3)
        grid[pick] += rng.integers(-3, 14, (int(pick.sum()), 2)).astype(np.int16)      # a drift, mostly one way
        pg.learn(grid, 0.2)
        st.learn(grid, 0.2)
        assert np.array_equal(pg.mtx, st.mtx) and np.array_equal(pg.adjust_vect, st.adjust_vect) and pg.adjust_contribs == st.adjust_contribs
    assert not np.array_equal(pg.mtx, O.posgrid(380))                # the grid did move at some point
    far = pg.mtx.copy()
    far[0, 0] -= 300
    with pytest.raises(ValueError):
        pg.learn(far)


def test_get_intersections_through_the_finder(quiet_game, monkeypatch):
    """StonesFinder.get_intersections: cached per frame, the grid learns from it (oracle-backed context)"""
    from camkifu_amd import capi
    from camkifu_amd.stone.sf_contours import SfContours
    from oracle import ora_grid as G
    from .stub_ctx import OracleCtx
    _, _, truth, gob, _ = quiet_game

    class Manager:
        device, current_video, controller = 0, None, None

    sf = SfContours(Manager(), ctx=OracleCtx())
    first = sf.get_intersections(gob)
    assert first is sf.get_intersections(gob)
    rects = np.array([[O.sf_getrect(r, c) for c in range(19)] for r in range(19)], np.int32)
    assert np.array_equal(first, G.find_intersections(gob, O.posgrid(380), rects))
    empty_seen = first[:, :, 0] < 0
    assert empty_seen.sum() > 100 and not (empty_seen & (truth > 0)).any()


def test_hull_and_min_area_rect_against_independent_geometry():
    """the strictly convex hull against qhull (scipy), the rotating-calipers box against a float64 search over the hull's
    edge directions (the minimum-area rectangle has a side collinear with a hull edge)"""
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(21)
    for trial in range(40):
        pts = rng.integers(0, 60, (int(rng.integers(5, 80)), 2))
        hull = S.convex_hull(pts)
        if len(hull) < 3:
            continue
        q = ConvexHull(pts.astype(np.float64))
        assert set(map(tuple, hull)) == set(map(tuple, pts[q.vertices])), trial
        w, h, angle = S.min_area_rect_box(pts)
        hv = hull.astype(np.float64)
        best = None
        for a, b in zip(hv, np.roll(hv, -1, 0)):
            d = (b - a) / np.hypot(*(b - a))
            along, across = hv @ d, hv @ np.array([-d[1], d[0]])
            cand = (along.max() - along.min()) * (across.max() - across.min())
            if best is None or cand < best[0]:
                best = (cand, sorted((along.max() - along.min(), across.max() - across.min())))
        assert abs(w * h - best[0]) <= 1e-3 * max(1.0, best[0]), trial
        assert np.allclose(sorted((w, h)), best[1], rtol=1e-4, atol=1e-3), trial
        assert -90.0 <= angle <= 0.0 or angle == 0.0
