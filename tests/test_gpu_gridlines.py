"""StonesFinder.find_intersections on the GPU (ck_find_intersections; stone/stonesfinder.py:516-552, 888-947) against
the oracle restatement (oracle/ora_grid.py): the Canny map of the grey image, the lines of every zone in the order the
progressive Hough transform finds them, and the resulting grid, all bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DST = np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32)


@pytest.fixture(scope="module")
def ck():
    from camkifu_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="module")
def gobans(ora):
    from camkifu_amd import synth
    out = []
    for seed, density in ((1, 0.0), (2, 0.2), (3, 0.5)):
        rng = np.random.default_rng(seed)
        corners = synth.random_corners(480, 640, rng)
        stones = synth.random_stones(rng, density=density)
        frame = synth.render(480, 640, stones, corners, seed=seed).numpy()
        out.append((ora.warp_perspective(frame, ora.get_perspective_transform(corners, DST)), stones))
    return out


def _tables(ora):
    return ora.posgrid(380), np.array([[ora.sf_getrect(r, c) for c in range(19)] for r in range(19)], np.int32)


def test_find_intersections_match_oracle(ck, ora, gobans):
    from oracle import ora_grid as G
    mtx, rects = _tables(ora)
    batch = np.stack([g for g, _ in gobans])
    grid, found, edges = ck.find_intersections(batch, mtx, rects, want_lines=True)
    total = 0
    for k, (gob, stones) in enumerate(gobans):
        g, f, e = G.find_intersections(gob, mtx, rects, want_lines=True)
        assert np.array_equal(edges[k], e), k
        assert found[k] == f, k
        assert np.array_equal(grid[k], g), k
        total += len(f)
        assert not ((g[:, :, 0] < 0) & (stones > 0)).any()          # no grid line is ever seen through a stone
    assert total > 300
    assert (grid[0][:, :, 0] < 0).mean() > 0.5                       # the empty board shows most of its lines
    # single image, device memory
    import torch
    assert np.array_equal(ck.find_intersections(gobans[1][0], mtx, rects), grid[1])
    assert np.array_equal(ck.find_intersections(torch.from_numpy(batch).cuda(), mtx, rects), grid)


def test_find_intersections_on_a_shifted_grid(ck, ora, gobans):
    """a learnt (displaced) grid: other zone rectangles, the same agreement"""
    from oracle import ora_grid as G
    from camkifu_amd.stone.stonesfinder import PosGrid
    pg = PosGrid(380)
    pg.mtx = pg.mtx + np.array([3, -2], np.int16)
    rects = pg.zones(1.0)
    gob = gobans[1][0]
    grid, found, _ = ck.find_intersections(gob, pg.mtx, rects, want_lines=True)
    g, f, _ = G.find_intersections(gob, pg.mtx, rects, want_lines=True)
    assert found == f and np.array_equal(grid, g)


def test_hough_zone_lines_on_patterns(ck, ora):
    """hand-made zones: a cross, a diagonal, two parallel strokes, noise -- lines and their order against the oracle"""
    from oracle import ora_grid as G
    mtx, rects = _tables(ora)
    rng = np.random.default_rng(9)
    img = np.zeros((380, 380, 3), np.uint8)
    img[:] = 90
    for r in range(19):
        for c in range(19):
            x0, y0, x1, y1 = rects[r, c]
            kind = (r * 19 + c) % 5
            z = img[x0:x1, y0:y1]
            if kind == 0:
                z[4:6, :] = 230                     # an off-centre cross: its long arms survive the crossing
                z[:, 4:6] = 230
            elif kind == 1:
                for t in range(min(z.shape[:2])):
                    z[t, t] = 230
            elif kind == 2:
                z[4:6, :] = 20
                z[13:15, :] = 20
            elif kind == 3:
                z[rng.random(z.shape[:2]) < 0.15] = 240
    grid, found, edges = ck.find_intersections(img, mtx, rects, want_lines=True)
    g, f, e = G.find_intersections(img, mtx, rects, want_lines=True)
    assert np.array_equal(edges, e) and found == f and np.array_equal(grid, g)
    assert len(f) > 60
    # every zone an off-centre cross whose arms run on into the next zone: both lines survive, the intersections move
    crosses = np.full((380, 380, 3), 90, np.uint8)
    for r in range(19):
        for c in range(19):
            x0, y0, x1, y1 = rects[r, c]
            crosses[x0 + 4:x0 + 6, y0:y1] = 230
            crosses[x0:x1, y0 + 4:y0 + 6] = 230
    grid, found, edges = ck.find_intersections(crosses, mtx, rects, want_lines=True)
    g, f, e = G.find_intersections(crosses, mtx, rects, want_lines=True)
    assert np.array_equal(edges, e) and found == f and np.array_equal(grid, g)
    assert (np.abs(g) != mtx).any(-1).sum() > 200


def test_find_intersections_argument_errors(ck, ora):
    from camkifu_amd import capi
    mtx, rects = _tables(ora)
    g = np.zeros((380, 380, 3), np.uint8)
    assert np.array_equal(ck.find_intersections(g, mtx, rects), mtx)  # nothing found: the grid comes back as it went in
    bad = rects.copy()
    bad[3, 3] = (60, 60, 160, 160)
    with pytest.raises(capi.CkError, match="error 1"):
        ck.find_intersections(g, mtx, bad)
    with pytest.raises(ValueError):
        ck.find_intersections(g[:100], mtx, rects)
