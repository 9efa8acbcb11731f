"""Independent brute-force checks of the oracle's image stages (K1..K11).

The reference holds no fixture for these stages ("parity unpinned"), so each
oracle routine is checked against a second, deliberately naive restatement (numpy / scipy /
torch-CPU) of the same published algorithm."""
import math

import numpy as np
import pytest
from scipy import ndimage


def _rand_img(rng, h, w, cn=3, smooth=True):
    img = rng.integers(0, 256, (h, w, cn), dtype=np.uint8)
    if smooth:
        img = ndimage.uniform_filter(img.astype(np.float32), (5, 5, 1)).astype(np.uint8)
        img[h // 4: h // 2, w // 3: 2 * w // 3] //= 3
    return img


# ---------------------------------------------------------------- K1 median
@pytest.mark.parametrize("ksize", [3, 5, 15])
def test_median_vs_numpy(ora, ksize):
    rng = np.random.default_rng(ksize)
    img = rng.integers(0, 256, (23, 37, 3), dtype=np.uint8)
    r = ksize // 2
    pad = np.pad(img, ((r, r), (r, r), (0, 0)), mode="edge")
    exp = np.empty_like(img)
    for y in range(img.shape[0]):
        for x in range(img.shape[1]):
            win = pad[y:y + ksize, x:x + ksize].reshape(-1, 3)
            exp[y, x] = np.sort(win, axis=0)[(ksize * ksize) // 2]
    assert np.array_equal(ora.median(img, ksize), exp)


def test_median_small_image_and_constant(ora):
    img = np.full((4, 6, 3), 77, np.uint8)
    assert np.array_equal(ora.median(img, 15), img)
    img = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3)
    out = ora.median(img, 15)           # window larger than the image: replicate border
    pad = np.pad(img, ((7, 7), (7, 7), (0, 0)), mode="edge")
    exp = np.empty_like(img)
    for y in range(2):
        for x in range(3):
            exp[y, x] = np.sort(pad[y:y + 15, x:x + 15].reshape(-1, 3), axis=0)[112]
    assert np.array_equal(out, exp)


# ---------------------------------------------------------------- K2 canny
def _canny_naive(img, low, high):
    h, w, cn = img.shape
    f = img.astype(np.int32)
    kx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]])
    ky = kx.T
    dx = np.stack([ndimage.correlate(f[..., c], kx, mode="nearest") for c in range(cn)], -1)
    dy = np.stack([ndimage.correlate(f[..., c], ky, mode="nearest") for c in range(cn)], -1)
    n = np.abs(dx) + np.abs(dy)
    ch = np.argmax(n, -1)                       # first maximal channel
    yy, xx = np.mgrid[0:h, 0:w]
    dx, dy, mag = dx[yy, xx, ch], dy[yy, xx, ch], n[yy, xx, ch]
    mp = np.pad(mag, 1)
    tg22 = int(0.4142135623730950488016887242097 * (1 << 15) + 0.5)
    cand = np.zeros((h, w), bool)
    for y in range(h):
        for x in range(w):
            m = int(mag[y, x])
            if m <= low:
                continue
            xs, ys = int(dx[y, x]), int(dy[y, x])
            ax, ay = abs(xs), abs(ys) << 15
            t = ax * tg22
            Y, X = y + 1, x + 1
            if ay < t:
                ok = m > mp[Y, X - 1] and m >= mp[Y, X + 1]
            elif ay > t + (ax << 16):
                ok = m > mp[Y - 1, X] and m >= mp[Y + 1, X]
            else:
                s = -1 if (xs ^ ys) < 0 else 1
                ok = m > mp[Y - 1, X - s] and m > mp[Y + 1, X + s]
            cand[y, x] = ok
    lab, nl = ndimage.label(cand, structure=np.ones((3, 3)))
    strong = np.unique(lab[cand & (mag > high)])
    out = np.isin(lab, strong[strong > 0])
    return (out * 255).astype(np.uint8), cand, mag, dx, dy


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_canny_vs_naive(ora, seed):
    rng = np.random.default_rng(seed)
    img = _rand_img(rng, 40, 56)
    edges, m, mag, dx, dy = ora.canny(img, 25, 75, want_map=True)
    e2, cand, mag2, dx2, dy2 = _canny_naive(img, 25, 75)
    assert np.array_equal(mag, mag2) and np.array_equal(dx, dx2) and np.array_equal(dy, dy2)
    assert np.array_equal(m != 1, cand)
    assert np.array_equal(edges, e2)
    assert edges.any() and not edges.all()


def test_canny_single_channel_and_flat(ora):
    flat = np.full((16, 16, 3), 100, np.uint8)
    assert not ora.canny(flat).any()
    step = np.zeros((16, 16), np.uint8)
    step[:, 8:] = 200
    e = ora.canny(step)
    # a vertical step gives one column of edges (asymmetric tie rule: > left, >= right)
    assert e[:, 7].all() and not e[:, 8].any() and e.sum() == 255 * 16


# ---------------------------------------------------------------- K3 contours
def _shapes_image(rng, h=48, w=64):
    e = np.zeros((h, w), np.uint8)
    e[5:40, 6:50] = 255
    e[6:39, 7:49] = 0            # thin ring
    e[10:20, 12:25] = 255        # blob nested in the ring
    e[12:18, 14:22] = 0
    e[14:16, 16:19] = 255        # nested deeper
    e[25:30, 30:35] = 255
    e[42:46, 3:60] = 255         # outside the ring
    e[0, :] = 255                # frame pixels must be ignored
    e[:, -1] = 255
    return e


def _check_suzuki_vs_sets(ora, e):
    conts = ora.find_external_suzuki(e)
    n, labels, starts = ora.find_external_sets(e)
    assert n == len(conts)
    for k, c in enumerate(conts):
        assert tuple(starts[k]) == tuple(c["start"])
        traced = set(map(tuple, c["pix"]))
        ys, xs = np.nonzero(labels == k)
        assert traced == set(zip(xs.tolist(), ys.tolist()))
        # CHAIN_APPROX_SIMPLE vertices span the same convex hull as the traced pixels
        assert ora.min_area_rect(c["vert"]) == ora.min_area_rect(c["pix"])


def test_suzuki_matches_set_definition_shapes(ora):
    rng = np.random.default_rng(0)
    e = _shapes_image(rng)
    conts = ora.find_external_suzuki(e)
    # ring + bar are top level; everything inside the closed ring is not returned
    assert len(conts) == 2
    _check_suzuki_vs_sets(ora, e)


@pytest.mark.parametrize("density", [0.05, 0.2, 0.45, 0.6, 0.8])
def test_suzuki_matches_set_definition_random(ora, density):
    rng = np.random.default_rng(int(density * 100))
    for _ in range(12):
        e = (rng.random((28, 33)) < density).astype(np.uint8) * 255
        _check_suzuki_vs_sets(ora, e)


def test_suzuki_edge_cases(ora):
    assert ora.find_external_suzuki(np.zeros((8, 8), np.uint8)) == []
    e = np.zeros((8, 8), np.uint8)
    e[3, 3] = 255
    c = ora.find_external_suzuki(e)
    assert len(c) == 1 and c[0]["pix"].tolist() == [[3, 3]] and c[0]["vert"].tolist() == [[3, 3]]
    e = np.full((8, 8), 255, np.uint8)
    c = ora.find_external_suzuki(e)          # solid block, frame cleared -> 6x6 square
    assert len(c) == 1 and c[0]["vert"].tolist() == [[1, 1], [1, 6], [6, 6], [6, 1]]


# ---------------------------------------------------------------- K4 minAreaRect / top-3
def _min_area_brute(pts):
    pts = np.unique(np.asarray(pts, np.float64), axis=0)
    if len(pts) < 3:
        return 0.0
    from scipy.spatial import ConvexHull, QhullError
    try:
        hull = pts[ConvexHull(pts).vertices]
    except QhullError:
        return 0.0
    best = np.inf
    for i in range(len(hull)):
        d = hull[(i + 1) % len(hull)] - hull[i]
        d /= np.linalg.norm(d)
        nrm = np.array([-d[1], d[0]])
        u, v = hull @ d, hull @ nrm
        best = min(best, (u.max() - u.min()) * (v.max() - v.min()))
    return best


def test_min_area_rect(ora):
    w, h = ora.min_area_rect([[0, 0], [10, 0], [10, 4], [0, 4], [5, 2]])
    assert sorted((w, h)) == [4.0, 10.0]
    assert ora.min_area_rect([[3, 3]]) == (0.0, 0.0)
    w, h = ora.min_area_rect([[0, 0], [3, 4]])
    assert (w, h) == (5.0, 0.0)
    assert ora.min_area_rect([[0, 0], [2, 2], [4, 4], [1, 1]])[1] == 0.0      # collinear
    rng = np.random.default_rng(3)
    for _ in range(50):
        pts = rng.integers(0, 200, (rng.integers(3, 60), 2))
        w, h = ora.min_area_rect(pts)
        ref = _min_area_brute(pts)
        assert abs(w * h - ref) <= 1e-3 * max(1.0, ref)


def test_top3_is_insort_order(ora):
    import bisect
    rng = np.random.default_rng(1)
    for n in (1, 2, 3, 7, 30):
        areas = rng.integers(0, 6, n).astype(np.float64)      # many ties

        class B:
            def __init__(self, a, p): self.area, self.pos = a, p
            def __lt__(self, o): return self.area < o.area
        s = []
        for i, a in enumerate(areas):
            bisect.insort(s, B(a, i))
        pos, big = ora.top3(areas)
        assert pos == [b.pos for b in s[-3:]] and big == s[-1].area


# ---------------------------------------------------------------- K6 hough
def _hough_naive(img, thr):
    h, w = img.shape
    theta = np.float32(math.pi / 180)
    numangle, numrho = 180, 2 * (w + h) + 1
    ang = np.float32(0)
    ts, tc = [], []
    for _ in range(numangle):
        ts.append(np.float32(math.sin(float(ang))))
        tc.append(np.float32(math.cos(float(ang))))
        ang = np.float32(ang + theta)
    ts, tc = np.array(ts, np.float32), np.array(tc, np.float32)
    acc = np.zeros((numangle + 2, numrho + 2), np.int32)
    ys, xs = np.nonzero(img)
    for y, x in zip(ys, xs):
        v = (np.float32(x) * tc + np.float32(y) * ts).astype(np.float32)
        r = np.rint(v.astype(np.float64)).astype(np.int64) + (numrho - 1) // 2
        acc[np.arange(1, numangle + 1), r + 1] += 1
    peaks = []
    for r in range(numrho):
        for n in range(numangle):
            a = acc[n + 1, r + 1]
            if a > thr and a > acc[n + 1, r] and a >= acc[n + 1, r + 2] and a > acc[n, r + 1] and a >= acc[n + 2, r + 1]:
                peaks.append((-int(a), (n + 1) * (numrho + 2) + r + 1, r, n))
    peaks.sort()
    lines = [((np.float32(r) - np.float32(numrho - 1) * np.float32(0.5)), np.float32(0) + np.float32(n) * theta)
             for _, _, r, n in peaks]
    return np.array(lines, np.float32).reshape(-1, 2), acc


def test_hough_vs_naive(ora):
    img = np.zeros((60, 80), np.uint8)
    img[10, 5:70] = 255
    img[5:55, 20] = 255
    for i in range(45):
        img[8 + i, 12 + i] = 255
        img[50 - i, 30 + i // 2] = 255
    lines, acc = ora.hough_lines(img, 12, want_accum=True)
    l2, acc2 = _hough_naive(img, 12)
    assert np.array_equal(acc, acc2)
    assert lines.shape == l2.shape and np.array_equal(lines, l2)
    assert len(lines) >= 3
    # the horizontal line y=10 is (rho=10, theta=pi/2); the vertical x=20 is (20, 0)
    as_set = {(float(r), round(float(t), 4)) for r, t in lines}
    assert (10.0, round(math.pi / 2, 4)) in as_set and (20.0, 0.0) in as_set


def test_hough_empty(ora):
    assert len(ora.hough_lines(np.zeros((20, 20), np.uint8), 3)) == 0


# ---------------------------------------------------------------- K7 / K8
def test_perspective_transform_and_warp(ora):
    rng = np.random.default_rng(2)
    src = np.array([[120, 80], [510, 95], [600, 400], [40, 380]], np.float32)
    dst = np.array([[0, 0], [380, 0], [380, 380], [0, 380]], np.float32)
    M = ora.get_perspective_transform(src, dst)
    for s, d in zip(src, dst):
        v = M @ np.array([s[0], s[1], 1.0])
        assert np.allclose(v[:2] / v[2], d, atol=1e-8)
    img = ndimage.uniform_filter(rng.integers(0, 256, (480, 640, 3)).astype(np.float32), (9, 9, 1)).astype(np.uint8)
    out = ora.warp_perspective(img, M)
    assert out.shape == (380, 380, 3)
    # float bilinear reference (coords quantised to 1/32 px like the library) within 1 LSB
    Mi = np.linalg.inv(M)
    ys, xs = np.mgrid[0:380, 0:380]
    den = Mi[2, 0] * xs + Mi[2, 1] * ys + Mi[2, 2]
    fx = np.rint((Mi[0, 0] * xs + Mi[0, 1] * ys + Mi[0, 2]) / den * 32) / 32
    fy = np.rint((Mi[1, 0] * xs + Mi[1, 1] * ys + Mi[1, 2]) / den * 32) / 32
    x0, y0 = np.floor(fx).astype(int), np.floor(fy).astype(int)
    ax, ay = (fx - x0)[..., None], (fy - y0)[..., None]
    pad = np.pad(img.astype(np.float64), ((1, 1), (1, 1), (0, 0)))
    g = lambda yy, xx: pad[np.clip(yy + 1, 0, 481), np.clip(xx + 1, 0, 641)]
    ref = (g(y0, x0) * (1 - ax) * (1 - ay) + g(y0, x0 + 1) * ax * (1 - ay) +
           g(y0 + 1, x0) * (1 - ax) * ay + g(y0 + 1, x0 + 1) * ax * ay)
    assert np.abs(out.astype(np.float64) - ref).max() <= 1.0


def test_warp_identity_translation_and_border(ora):
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (50, 60, 3), dtype=np.uint8)
    out = ora.warp_perspective(img, np.eye(3), (60, 50))
    assert np.array_equal(out, img)
    T = np.array([[1, 0, 7], [0, 1, -3], [0, 0, 1]], np.float64)     # dst = src shifted (+7, -3)
    out = ora.warp_perspective(img, T, (60, 50))
    exp = np.zeros_like(img)
    exp[:47, 7:] = img[3:, :53]
    assert np.array_equal(out, exp)


# ---------------------------------------------------------------- K9
def test_mog2_behaviour(ora):
    rng = np.random.default_rng(6)
    base = rng.integers(60, 180, (24, 24, 3)).astype(np.int16)
    m = ora.MOG2(24, 24, 3)
    fg = None
    for f in range(60):
        frame = np.clip(base + rng.integers(-2, 3, base.shape), 0, 255).astype(np.uint8)
        fg = m.apply(frame, 0.01 if f < 50 else 0.005)
        if f == 0:
            assert (fg == 255).all()        # first frame: every pixel opens a new mode
    assert (fg == 0).mean() > 0.99
    frame = np.clip(base + rng.integers(-2, 3, base.shape), 0, 255).astype(np.uint8)
    frame[5:12, 5:12] = 255 - frame[5:12, 5:12]
    fg = m.apply(frame, 0.005)
    assert (fg[5:12, 5:12] == 255).all() and (fg[14:, 14:] == 0).all()
    assert set(np.unique(fg)) <= {0, 255}


def _mog2_reference(frames, rates):
    """Zivkovic's adaptive mixture written out pixel by pixel in numpy float32 scalars (library defaults: 5 modes, Tb 16,
    Tg 9, TB 0.9, initial variance 15 clipped to [4, 75], complexity reduction 0.05, no shadows) -- a second, structurally
    different statement of K9 (lists of modes that are re-sorted, instead of in-place bubbling over fixed arrays)"""
    f32 = np.float32
    h, w = frames[0].shape[:2]
    modes = [[[] for _ in range(w)] for _ in range(h)]           # each mode: [weight, variance, mean(3)]
    masks = []
    for t, (img, rate) in enumerate(zip(frames, rates)):
        nframes = t + 1
        lr = rate if (rate >= 0 and nframes > 1) else 1.0 / min(2 * nframes, 500)
        alpha = f32(lr)
        one_minus = f32(1.0) - alpha
        prune = f32(-lr * f32(0.05))
        mask = np.zeros((h, w), np.uint8)
        for y in range(h):
            for x in range(w):
                px = img[y, x].astype(np.float32)
                ms = modes[y][x]
                background, fits, total = False, False, f32(0)
                k = 0
                while k < len(ms):
                    wgt = one_minus * ms[k][0] + prune
                    moved_to = k
                    if not fits:
                        var = ms[k][1]
                        d = ms[k][2] - px
                        dist2 = f32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]
                        if total < f32(0.9) and dist2 < f32(16) * var:
                            background = True
                        if dist2 < f32(9) * var:
                            fits = True
                            wgt = wgt + alpha
                            kk = alpha / wgt
                            ms[k][2] = ms[k][2] - kk * d
                            nv = var + kk * (dist2 - var)
                            ms[k][1] = min(max(nv, f32(4)), f32(75))
                            while moved_to > 0 and not (wgt < ms[moved_to - 1][0]):     # keep the modes sorted by weight
                                ms[moved_to], ms[moved_to - 1] = ms[moved_to - 1], ms[moved_to]
                                moved_to -= 1
                    if wgt < -prune:
                        wgt = f32(0)
                        ms[moved_to][0] = wgt
                        ms.pop()                                   # sorted: the mode that falls away is the last one
                        if moved_to >= len(ms):
                            continue
                    else:
                        ms[moved_to][0] = wgt
                    total = total + wgt
                    k += 1
                inv = f32(1) / total if len(ms) else f32(0)
                for m in ms:
                    m[0] = m[0] * inv
                if not fits and alpha > 0:
                    if len(ms) == 5:
                        ms.pop()
                    if not ms:
                        ms.append([f32(1), f32(15), px.copy()])
                    else:
                        for m in ms:
                            m[0] = m[0] * one_minus
                        ms.append([alpha, f32(15), px.copy()])
                        j = len(ms) - 1
                        while j > 0 and not (alpha < ms[j - 1][0]):
                            ms[j], ms[j - 1] = ms[j - 1], ms[j]
                            j -= 1
                mask[y, x] = 0 if background else 255
        masks.append(mask)
    return masks


def test_mog2_vs_numpy_restatement(ora):
    """the C oracle against the list-of-modes numpy statement above, mask for mask, over a sequence with sensor noise, an
    object that appears and stays (absorbed into the background), one that passes, and both learning rates"""
    rng = np.random.default_rng(31)
    h, w, n = 12, 14, 90
    base = rng.integers(40, 200, (h, w, 3)).astype(np.int16)
    frames, rates = [], []
    for t in range(n):
        fr = np.clip(base + rng.integers(-3, 4, base.shape), 0, 255).astype(np.uint8)
        if t >= 30:
            fr[2:6, 3:8] = np.clip(220 + rng.integers(-2, 3, (4, 5, 3)), 0, 255)       # a stone is put down and stays
        if 50 <= t < 58:
            fr[7:11, (t - 50):(t - 46)] = 15                                           # something dark passes
        frames.append(fr)
        rates.append(0.01 if t < 20 else 0.005)
    want = _mog2_reference(frames, rates)
    m = ora.MOG2(h, w, 3)
    seen_fg = 0
    for t in range(n):
        got = m.apply(frames[t], rates[t])
        assert np.array_equal(got, want[t]), t
        seen_fg += int((got[2:6, 3:8] == 255).sum())
    assert seen_fg > 0 and (want[-1][2:6, 3:8] == 0).all()                              # seen, then absorbed


# ---------------------------------------------------------------- K10-K12
def _weights(seed=20161001, scale1=1.0 / 128):
    rng = np.random.default_rng(seed)
    from oracle.oracle import WEIGHT_SHAPES
    W = {}
    for k, shp in WEIGHT_SHAPES.items():
        if k.endswith("b"):
            W[k] = (rng.standard_normal(shp) * 0.05).astype(np.float32)
        else:
            fan_in = int(np.prod(shp[:-1]))
            W[k] = (rng.standard_normal(shp) * math.sqrt(2.0 / fan_in)).astype(np.float32)
    W["c1w"] *= np.float32(scale1)
    return W


def test_cnn_vs_torch_cpu(ora):
    import torch
    import torch.nn.functional as F
    W = _weights()
    rng = np.random.default_rng(7)
    patches = rng.integers(0, 256, (3, 40, 40, 3), dtype=np.uint8)
    y, lg = ora.cnn_forward(W, patches, want_logits=True)

    def conv(x, w, b):                      # true convolution == correlation with flipped kernel
        wt = torch.from_numpy(np.ascontiguousarray(w[::-1, ::-1].transpose(3, 2, 0, 1))).double()
        return F.relu(F.conv2d(x, wt, torch.from_numpy(b).double()))
    x = torch.from_numpy(patches.astype(np.float64)).permute(0, 3, 1, 2)
    x = conv(x, W["c1w"], W["c1b"])
    x = conv(x, W["c2w"], W["c2b"])
    x = F.max_pool2d(x, 2)
    x = conv(x, W["c3w"], W["c3b"])
    x = conv(x, W["c4w"], W["c4b"])
    x = F.max_pool2d(x, 2)
    x = x.permute(0, 2, 3, 1).reshape(3, -1)             # Flatten on channels-last
    x = F.relu(x @ torch.from_numpy(W["d1w"]).double() + torch.from_numpy(W["d1b"]).double())
    lg2 = x @ torch.from_numpy(W["d2w"]).double() + torch.from_numpy(W["d2b"]).double()
    y2 = torch.softmax(lg2, 1).numpy()
    assert np.abs(lg - lg2.numpy()).max() <= 1e-4 * max(1.0, np.abs(lg2.numpy()).max())
    assert np.abs(y - y2).max() <= 1e-4
    assert np.array_equal(np.argmax(y, 1), np.argmax(y2, 1))


def test_cnn_region_slicing(ora):
    W = _weights()
    rng = np.random.default_rng(8)
    goban = rng.integers(0, 256, (380, 380, 3), dtype=np.uint8)
    y = ora.cnn_predict_regions(W, goban)
    idx = [(0, 0), (3, 7), (9, 9), (9, 2)]
    patches = []
    for i, j in idx:
        x0, x1, y0, y1 = ora.nn_rect(*ora.subregion(i, j))
        patches.append(goban[x0:x1, y0:y1])
    y2 = ora.cnn_forward(W, np.stack(patches))
    for k, (i, j) in enumerate(idx):
        assert np.array_equal(y[i * 10 + j], y2[k])

# ---------------------------------------------------------------- SfContours.get_canny pieces
def test_bgr2gray_matches_float_formula_within_rounding(ora):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)
    g = ora.bgr2gray(img).astype(np.float64)
    f = 0.114 * img[..., 0] + 0.587 * img[..., 1] + 0.299 * img[..., 2]
    assert np.abs(g - f).max() <= 0.51              # 14-bit coefficients, round to nearest
    assert (ora.bgr2gray(np.full((2, 2, 3), 255, np.uint8)) == 255).all() and (ora.bgr2gray(np.zeros((2, 2, 3), np.uint8)) == 0).all()


def test_otsu_level_maximises_between_class_variance(ora):
    """independent restatement: brute force over all 256 cuts with exact class statistics"""
    rng = np.random.default_rng(6)
    for k in range(6):
        a = np.clip(rng.normal(70 + 10 * k, 12, 3000), 0, 255)
        b = np.clip(rng.normal(180 - 8 * k, 20, 2000 + 300 * k), 0, 255)
        g = np.concatenate([a, b]).astype(np.uint8).reshape(50, -1)
        lvl = int(ora.otsu_level(g))
        v = g.ravel().astype(np.float64)
        best, arg = -1.0, 0
        for t in range(256):
            lo, hi = v[v <= t], v[v > t]
            if len(lo) == 0 or len(hi) == 0:
                continue
            s = len(lo) * len(hi) * (lo.mean() - hi.mean()) ** 2
            if s > best * (1 + 1e-12):
                best, arg = s, t
        assert lvl == arg, (k, lvl, arg)
    assert ora.otsu_level(np.full((8, 8), 77, np.uint8)) == 0.0          # one class only: nothing beats sigma = 0


def test_goban_canny_chain_is_its_parts(ora):
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (64, 70, 3), dtype=np.uint8)
    img[16:48, 20:50] = (30, 40, 50)
    e, otsu = ora.goban_canny(img, want_otsu=True)
    m = ora.median(ora.median(img, 13), 7)
    assert otsu == ora.otsu_level(ora.bgr2gray(m))
    assert np.array_equal(e, ora.canny(m, int(otsu // 2), int(otsu)))
