"""Per-frame result records written in place (round 6): ck_board_detect_records / ck_cnn_regions_records against the plain
entry points they wrap (ck_board_detect, ck_cnn_regions -- themselves held to the oracle bit for bit by
tests/test_gpu_parity.py), records in host memory and in HBM, each half leaving the other untouched.

Reference side: the board half is what BoardFinderAuto._detect reads after the image chain (board/bf_auto.py:76-84,
125-135), the stones half what NNCache.predict_4_stones reads (stone/nn_cache.py:16-31)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

H, W = 480, 640


@pytest.fixture(scope="module")
def setup():
    import torch
    from camkifu_amd import capi, synth
    from camkifu_amd.stone.nn_manager import NNManager
    ctx = capi.Context(0)
    ctx.cnn_set_weights(NNManager.init_net())
    frames = synth.film(12, H, W, seed=5, quiet=2, move_every=4, hand_frames=2)[0]
    frames[3] = 90                                              # a frame with nothing in it: status != CK_BOARD_LINES
    yield ctx, frames, torch
    ctx.close()


def _filled(n, where, torch):
    from camkifu_amd import capi
    a = np.full((n, capi.REC_BYTES), 0xAB, np.uint8)
    return torch.from_numpy(a).cuda() if where == "hbm" else a.view(capi.REC_DTYPE).reshape(n)


def _host(rec, torch):
    from camkifu_amd import capi
    if hasattr(rec, "is_cuda"):
        return rec.cpu().numpy().view(capi.REC_DTYPE).reshape(len(rec))
    return rec


@pytest.mark.parametrize("where", ["host", "hbm"])
@pytest.mark.parametrize("frames_on", ["host", "hbm"])
def test_board_half_equals_board_detect_and_leaves_the_stones_half_alone(setup, where, frames_on):
    from camkifu_amd import capi
    ctx, frames, torch = setup
    fr = frames.cuda() if frames_on == "hbm" else frames.numpy()
    n = len(frames)
    res, lines = ctx.board_detect(fr, -1, capi.REC_LMAX, raw=True)
    assert (res["status"] == capi.CK_BOARD_LINES).any() and (res["status"] != capi.CK_BOARD_LINES).any()
    rec = _filled(n, where, torch)
    ctx.board_detect_records(fr, rec)
    got = _host(rec, torch)
    for name in ("status", "n_contours", "n_lines", "biggest_area"):
        assert np.array_equal(got[name], res[name]), name
    kept = np.minimum(res["n_lines"], capi.REC_LMAX)
    for f in range(n):
        assert np.array_equal(got["lines"][f, :kept[f]], lines[f, :kept[f]])
        assert not got["lines"][f, kept[f]:].any()              # beyond the lines found: zero, whatever the scratch held
    assert np.array_equal(got["flags"], np.where(res["n_lines"] > capi.REC_LMAX, capi.REC_LINES_CUT, 0))
    raw = got.view(np.uint8).reshape(n, capi.REC_BYTES)
    assert (raw[:, capi.REC_DTYPE.fields["region_conf"][1]:] == 0xAB).all()   # the stones half: untouched


def test_more_lines_than_a_record_holds_are_flagged_and_the_strongest_kept(setup):
    """a low Hough threshold gives hundreds of lines: the record keeps the first CK_REC_LMAX (OpenCV's order = most votes
    first), says how many there were, and raises CK_REC_LINES_CUT"""
    from camkifu_amd import capi
    ctx, frames, torch = setup
    fr = frames[:2].cuda()
    for thresh in (24, 12, 6):
        res, lines = ctx.board_detect(fr, thresh, 4096, raw=True)
        if res["n_lines"].max() > capi.REC_LMAX:
            break
    assert res["n_lines"].max() > capi.REC_LMAX
    rec = _filled(2, "hbm", torch)
    ctx.board_detect_records(fr, rec, hough_thresh=thresh)
    got = _host(rec, torch)
    assert np.array_equal(got["n_lines"], res["n_lines"])
    cut = res["n_lines"] > capi.REC_LMAX
    assert np.array_equal(got["flags"] & capi.REC_LINES_CUT, np.where(cut, capi.REC_LINES_CUT, 0))
    for f in range(2):
        k = min(int(res["n_lines"][f]), capi.REC_LMAX)
        assert np.array_equal(got["lines"][f, :k], lines[f, :k])


@pytest.mark.parametrize("where", ["host", "hbm"])
@pytest.mark.parametrize("mode", ["f16x2", "bf16"])
def test_stones_half_equals_cnn_regions_and_leaves_the_board_half_alone(setup, where, mode):
    from camkifu_amd import capi
    ctx, frames, torch = setup
    ctx.cnn_set_mode(dict(f16x2=capi.CK_CNN_F16X2, bf16=capi.CK_CNN_BF16)[mode])
    try:
        sc_corners = np.array([[120, 60], [520, 70], [530, 430], [110, 420]], np.float32)
        M = capi.get_perspective_transform(sc_corners, np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32))
        gob = ctx.warp_perspective(frames.cuda(), M)
        n = len(gob)
        lab, conf = ctx.cnn_regions(gob)
        rec = _filled(n, where, torch)
        ctx.cnn_regions_records(gob, rec)
        got = _host(rec, torch)
        assert np.array_equal(got["region_label"], lab.cpu().numpy())
        assert np.array_equal(got["region_conf"], conf.cpu().numpy())        # the same doubles, bit for bit
        raw = got.view(np.uint8).reshape(n, capi.REC_BYTES)
        assert (raw[:, :capi.REC_DTYPE.fields["region_conf"][1]] == 0xAB).all() and (raw[:, -4:] == 0xAB).all()
    finally:
        ctx.cnn_set_mode(capi.CK_CNN_DEFAULT)


def test_both_halves_from_two_contexts_at_once_fill_one_buffer(setup):
    """the pipeline's use: a board context and a stones context, each on its own thread and stream, write their halves of
    the same records in HBM at the same time; slices of one buffer per lane"""
    import threading
    from camkifu_amd import capi
    ctx, frames, torch = setup
    other = capi.Context(0)
    try:
        fr = frames.cuda()
        n = len(fr)
        M = capi.get_perspective_transform(np.array([[120, 60], [520, 70], [530, 430], [110, 420]], np.float32),
                                           np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32))
        gob = ctx.warp_perspective(fr, M)
        res, lines = other.board_detect(fr, -1, capi.REC_LMAX, raw=True)
        lab, conf = ctx.cnn_regions(gob)
        buf = torch.zeros((n + 2, capi.REC_BYTES), dtype=torch.uint8, device="cuda")
        errs = []

        def run(fn, *a):
            try:
                fn(*a)
            except Exception as why:                             # pragma: no cover
                errs.append(why)
        ts = [threading.Thread(target=run, args=(other.board_detect_records, fr[:5], buf[1:6])),
              threading.Thread(target=run, args=(ctx.cnn_regions_records, gob, buf[1:1 + n]))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        other.board_detect_records(fr[5:], buf[6:1 + n])
        assert not errs, errs
        got = _host(buf, torch)
        assert not got[0].tobytes().strip(b"\0") and not got[-1].tobytes().strip(b"\0")   # header and spare rows: zero
        body = got[1:1 + n]
        assert np.array_equal(body["n_lines"], res["n_lines"]) and np.array_equal(body["status"], res["status"])
        assert np.array_equal(body["region_label"], lab.cpu().numpy()) and np.array_equal(body["region_conf"], conf.cpu().numpy())
    finally:
        other.close()


def test_bad_record_arguments_are_refused(setup):
    from camkifu_amd import capi
    ctx, frames, torch = setup
    with pytest.raises(ValueError):
        ctx.board_detect_records(frames[:2].numpy(), np.zeros(3, capi.REC_DTYPE))                 # wrong count
    with pytest.raises(ValueError):
        ctx.board_detect_records(frames[:2].numpy(), np.zeros(4, capi.REC_DTYPE)[::2])            # not contiguous
    with pytest.raises(ValueError):
        ctx.cnn_regions_records(np.zeros((2, 380, 380, 3), np.uint8), torch.zeros((2, 100), dtype=torch.uint8))
