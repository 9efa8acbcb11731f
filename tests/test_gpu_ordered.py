"""The ordered stones path on the GPU (ck_stones_run = K8 -> K9 in frame order -> K10..K12 in one call,
ck_cnn_regions, ck_zone_counts) against the oracle: foreground counts per intersection zone bit-exact
through a whole sequence (MOG2 is stateful), region labels bit-exact, region confidences within 1e-4
(they are ratios of softmax outputs, north_star's float tolerance), and the run form against the
per-frame form of the same library (bit-exact, including the state carried from one run to the next)."""
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


def _clip(n, seed):
    """a fixed camera, a stone every few frames, and a dark blob (a hand) wandering over the board"""
    from camkifu_amd import synth
    rng = np.random.default_rng(seed)
    corners = synth.random_corners(240, 320, rng)
    stones = synth.random_stones(rng, density=0.2)
    frames = []
    for f in range(n):
        if f and f % 7 == 0:
            r, c = rng.integers(1, 18, 2)
            stones[r, c] = 1 + (f // 7) % 2
        fr = synth.render(240, 320, stones, corners, seed=seed * 1000 + f).numpy().copy()
        if 10 <= f % 25 < 18:
            cx, cy = 100 + 5 * (f % 25), 80 + 3 * (f % 25)
            fr[cy:cy + 50, cx:cx + 40] = (60, 90, 140)
        frames.append(fr)
    return np.stack(frames), corners


def test_zone_counts_match_oracle(ck, ora):
    rng = np.random.default_rng(5)
    masks = (rng.random((3, 380, 380)) < np.array([0.05, 0.5, 0.95])[:, None, None]).astype(np.uint8) * 255
    masks[1, 360:, :] = 255
    got = ck.zone_counts(masks)
    for k in range(3):
        assert np.array_equal(got[k], ora.zone_counts(masks[k])), k
    assert np.array_equal(ck.zone_counts(masks[2]), got[2])


def test_stones_run_matches_oracle_sequence(ck, ora):
    from camkifu_amd.stone.nn_manager import NNManager
    W = NNManager.init_net()
    ck.cnn_set_weights(W)
    n = 36
    frames, corners = _clip(n, seed=8)
    M = ora.get_perspective_transform(corners, DST)
    lr = np.where(np.arange(n) < 20, 0.01, 0.005)
    h = ck.mog2_create(380, 380)
    # two runs back to back: the mixture state must carry over
    a = ck.stones_run(frames[:15], M, mog2=h, learning_rates=lr[:15], want_grid=True)
    b = ck.stones_run(frames[15:], M, mog2=h, learning_rates=lr[15:], want_grid=True)
    rl = np.concatenate([a["region_label"], b["region_label"]])
    rc = np.concatenate([a["region_conf"], b["region_conf"]])
    fg = np.concatenate([a["fgcount"], b["fgcount"]])
    grid = np.concatenate([a["labels"], b["labels"]])
    # the same library frame by frame: warp, mog2_apply, zone_counts, cnn_regions
    h2 = ck.mog2_create(380, 380)
    model = ora.MOG2(380, 380, 3)
    some_fg = 0
    for f in range(n):
        gob = ck.warp_perspective(frames[f], M)
        mask = ck.mog2_apply(h2, gob, float(lr[f]))
        assert np.array_equal(ck.zone_counts(mask), fg[f]), f
        r1, c1 = ck.cnn_regions(gob)
        assert np.array_equal(r1[0], rl[f]) and np.array_equal(c1[0], rc[f]), f
        # the oracle
        gob_o = ora.warp_perspective(frames[f], M)
        assert np.array_equal(gob, gob_o)
        assert np.array_equal(ora.zone_counts(model.apply(gob_o, float(lr[f]))), fg[f]), f
        y = ora.cnn_predict_regions(W, gob_o)
        lab_o, conf_o = ora.decode_regions(y)
        assert np.array_equal(lab_o.reshape(10, 10), rl[f]), f
        assert np.abs(conf_o.reshape(10, 10) - rc[f]).max() <= 1e-4
        assert np.array_equal(ora.decode_all(y)[0], grid[f])
        some_fg += int(fg[f].sum())
    assert some_fg > 0
    ck.mog2_destroy(h)
    ck.mog2_destroy(h2)


def test_stones_run_without_background_model(ck, ora):
    from camkifu_amd.stone.nn_manager import NNManager
    ck.cnn_set_weights(NNManager.init_net())
    frames, corners = _clip(3, seed=3)
    M = ora.get_perspective_transform(corners, DST)
    out = ck.stones_run(frames, M)
    assert out["fgcount"] is None and out["region_label"].shape == (3, 10, 10)
    lab, conf = ck.stones_detect(frames, M)
    again = ck.stones_run(frames, M, want_grid=True)
    assert np.array_equal(again["labels"], lab) and np.array_equal(again["conf"], conf)
    # the grid is the regions' answers with the later region winning on row / column 17
    for f in range(3):
        assert again["labels"][f][16, 16] == again["region_label"][f][8, 8] % 3
        assert again["conf"][f][18, 18] == again["region_conf"][f][9, 9]


@pytest.mark.parametrize("world", [2, 8])
def test_band_models_equal_the_whole_image_model(ck, ora, world):
    """the multi-GPU form of K9: the goban split into `world` bands of intersection rows, one model per band run over the
    whole sequence (ck_mog2_band_run, host and device inputs) == the whole-image run of ck_stones_run, count for count"""
    import torch
    from camkifu_amd import pipeline
    from camkifu_amd.stone.nn_manager import NNManager
    ck.cnn_set_weights(NNManager.init_net())
    n = 30
    frames, corners = _clip(n, seed=21)
    M = ora.get_perspective_transform(corners, DST)
    lr = np.where(np.arange(n) < 10, 0.01, 0.005)
    h = ck.mog2_create(380, 380)
    whole = ck.stones_run(frames, M, mog2=h, learning_rates=lr)["fgcount"]
    gobans = ck.warp_perspective(frames, M)
    parts = []
    for k, (a, b) in enumerate(pipeline.band_rows(world)):
        lo, hi = 20 * a, min(20 * b, 380)
        hb = ck.mog2_create(hi - lo, 380)
        band = np.ascontiguousarray(gobans[:, lo:hi])
        if k % 2:                                            # device-resident band, in two runs (state carries over)
            t = torch.from_numpy(band).cuda()
            got = torch.cat([ck.mog2_band_run(hb, t[:11], lr[:11], last_band=(b == 19)),
                             ck.mog2_band_run(hb, t[11:], lr[11:], last_band=(b == 19))]).cpu().numpy()
        else:
            got = ck.mog2_band_run(hb, band, lr, last_band=(b == 19))
        assert got.shape == (n, b - a, 19)
        parts.append(got)
        ck.mog2_destroy(hb)
    assert np.array_equal(np.concatenate(parts, 1), whole) and whole.sum() > 0
    ck.mog2_destroy(h)


def test_ordered_entry_points_report_errors(ora):
    """status codes, never exceptions across the boundary: missing weights, bad model handles, a learning rate that
    would reset the model inside a run, single-frame and empty-ish calls"""
    from camkifu_amd import capi
    ctx = capi.Context(0)
    try:
        frames, corners = _clip(2, seed=1)
        M = ora.get_perspective_transform(corners, DST)
        with pytest.raises(capi.CkError, match="error 4"):                  # CK_ERR_STATE: no weights yet
            ctx.stones_run(frames, M)
        from camkifu_amd.stone.nn_manager import NNManager
        ctx.cnn_set_weights(NNManager.init_net())
        with pytest.raises(capi.CkError, match="error 1"):
            ctx.stones_run(frames, M, mog2=7, learning_rates=[0.01, 0.01])  # no such model
        h = ctx.mog2_create(380, 380)
        with pytest.raises(capi.CkError, match="error 1"):
            ctx.stones_run(frames, M, mog2=h, learning_rates=[0.01, 1.0])   # a reset inside a run is refused
        one = ctx.stones_run(frames[:1], M, mog2=h, learning_rates=[0.01])
        assert one["fgcount"].shape == (1, 19, 19) and one["region_label"].shape == (1, 10, 10)
        small = ctx.mog2_create(40, 380)
        with pytest.raises(capi.CkError, match="error 1"):
            ctx.stones_run(frames, M, mog2=small, learning_rates=[0.01, 0.01])   # a band model is not a goban model
        with pytest.raises(capi.CkError, match="error 1"):
            ctx.mog2_band_run(99, np.zeros((1, 40, 380, 3), np.uint8), [0.01], last_band=False)
        out = ctx.mog2_band_run(small, np.zeros((3, 40, 380, 3), np.uint8), [0.01] * 3, last_band=False)
        assert out.shape == (3, 2, 19) and out[0].sum() == 40 * 379 and out[1:].sum() == 0     # first frame: all foreground
    finally:
        ctx.close()
