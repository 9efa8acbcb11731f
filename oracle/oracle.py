"""ctypes/numpy front-end of the CPU ORACLE (oracle/*.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; the product package (camkifu_amd/) must never import it.

Every function restates one call site of the reference hot path; see oracle/ck_oracle.h
for the per-function reference citations (file:line under /root/reference).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libck_oracle.so")
_lib = None

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the oracle with gcc (make).  Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.ora_find_external_suzuki.restype = C.c_int
        _lib.ora_find_external_sets.restype = C.c_int
        _lib.ora_top3.restype = C.c_int
        _lib.ora_board_lines.restype = C.c_int
        _lib.ora_hough_lines.restype = C.c_int
        _lib.ora_get_perspective_transform.restype = C.c_int
        _lib.ora_mog2_create.restype = C.c_void_p
    return _lib


def num_threads():
    return int(lib().ora_num_threads())


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def median(img, ksize=15):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    cn = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty_like(img)
    lib().ora_median(_vp(img), h, w, cn, ksize, _vp(out))
    return out


def bgr2gray(img):
    """cv2.cvtColor(img, COLOR_BGR2GRAY) for 8-bit images: 14-bit fixed point, rounded
    (reference call site: stone/sf_contours.py:338)"""
    a = np.asarray(img, np.uint8).astype(np.int64)
    return ((1868 * a[..., 0] + 9617 * a[..., 1] + 4899 * a[..., 2] + (1 << 13)) >> 14).astype(np.uint8)


def otsu_level(gray):
    """the level cv2.threshold(gray, _, 255, THRESH_OTSU) returns: the library's getThreshVal_Otsu_8u restated
    (double arithmetic, FLT_EPSILON guards, the first maximum of the between-class variance wins)"""
    hist = np.bincount(np.asarray(gray, np.uint8).ravel(), minlength=256).astype(np.float64)
    scale = 1.0 / gray.size
    mu = float((np.arange(256) * hist).sum()) * scale
    eps = float(np.finfo(np.float32).eps)
    mu1 = q1 = 0.0
    max_sigma, max_val = 0.0, 0
    for i in range(256):
        p_i = hist[i] * scale
        mu1 *= q1
        q1 += p_i
        q2 = 1.0 - q1
        if min(q1, q2) < eps or max(q1, q2) > 1.0 - eps:
            continue
        mu1 = (mu1 + i * p_i) / q1
        mu2 = (mu - q1 * mu1) / q2
        sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2)
        if sigma > max_sigma:
            max_sigma, max_val = sigma, i
    return float(max_val)


def goban_canny(img, want_otsu=False):
    """SfContours.get_canny (stone/sf_contours.py:332-340): medianBlur 13, medianBlur 7, Otsu level of the grey
    image, Canny(median, otsu / 2, otsu) with the thresholds floored as cv2.Canny does for the L1 gradient"""
    m = median(median(img, 13), 7)
    otsu = otsu_level(bgr2gray(m))
    e = canny(m, int(np.floor(otsu / 2)), int(np.floor(otsu)))
    return (e, otsu) if want_otsu else e


def canny(img, low=25, high=75, want_map=False):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    cn = 1 if img.ndim == 2 else img.shape[2]
    edges = np.empty((h, w), np.uint8)
    if not want_map:
        lib().ora_canny(_vp(img), h, w, cn, int(low), int(high), _vp(edges), None, None, None, None)
        return edges
    m = np.empty((h, w), np.uint8)
    mag = np.empty((h, w), np.int32)
    dx = np.empty((h, w), np.int16)
    dy = np.empty((h, w), np.int16)
    lib().ora_canny(_vp(img), h, w, cn, int(low), int(high), _vp(edges), _vp(m), _vp(mag), _vp(dx), _vp(dy))
    return edges, m, mag, dx, dy


def find_external_suzuki(edges):
    """-> list of dicts {start:(x,y), pix:(n,2) int32, vert:(m,2) int32} in discovery order."""
    edges = np.ascontiguousarray(edges, np.uint8)
    h, w = edges.shape
    maxc = h * w // 2 + 4
    maxp = 4 * h * w + 16
    starts = np.zeros((maxc, 2), np.int32)
    poff = np.zeros(maxc + 1, np.int32)
    voff = np.zeros(maxc + 1, np.int32)
    pix = np.zeros((maxp, 2), np.int32)
    vert = np.zeros((maxp, 2), np.int32)
    n = lib().ora_find_external_suzuki(_vp(edges), h, w, maxc, _vp(starts), _vp(poff), _vp(voff),
                                       maxp, _vp(pix), maxp, _vp(vert))
    assert n >= 0
    return [dict(start=tuple(starts[k]), pix=pix[poff[k]:poff[k + 1]].copy(),
                 vert=vert[voff[k]:voff[k + 1]].copy()) for k in range(n)]


def find_external_sets(edges):
    edges = np.ascontiguousarray(edges, np.uint8)
    h, w = edges.shape
    labels = np.empty((h, w), np.int32)
    maxc = h * w // 2 + 4
    starts = np.zeros((maxc, 2), np.int32)
    n = lib().ora_find_external_sets(_vp(edges), h, w, _vp(labels), maxc, _vp(starts))
    return n, labels, starts[:n].copy()


def min_area_rect(pts):
    pts = np.ascontiguousarray(pts, np.int32).reshape(-1, 2)
    wh = np.zeros(2, np.float32)
    lib().ora_min_area_rect(_vp(pts), len(pts), _vp(wh))
    return float(wh[0]), float(wh[1])


def top3(areas):
    areas = np.ascontiguousarray(areas, np.float64)
    pos = np.zeros(3, np.int32)
    big = C.c_double(0)
    k = lib().ora_top3(_vp(areas), len(areas), _vp(pos), C.byref(big))
    return list(pos[:k]), big.value


def hough_lines(img, threshold, cap=4096, want_accum=False):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    lines = np.zeros((cap, 2), np.float32)
    acc = None
    if want_accum:
        acc = np.zeros((182, 2 * (w + h) + 3), np.int32)
    n = lib().ora_hough_lines(_vp(img), h, w, int(threshold), _vp(lines), cap, _vp(acc))
    out = lines[:min(n, cap)].copy()
    return (out, acc) if want_accum else out


def board_lines(edges, hough_thresh=None, cap=4096):
    """K3..K6.  -> dict(status, lines, ghost, biggest_area, n_contours)"""
    edges = np.ascontiguousarray(edges, np.uint8)
    h, w = edges.shape
    if hough_thresh is None:
        hough_thresh = int(min(h, w) / 5)
    ghost = np.zeros((h, w), np.uint8)
    lines = np.zeros((cap, 2), np.float32)
    big = C.c_double(0)
    nc = C.c_int(0)
    n = lib().ora_board_lines(_vp(edges), h, w, int(hough_thresh), _vp(ghost), _vp(lines), cap,
                              C.byref(big), C.byref(nc))
    return dict(status=n, lines=lines[:max(0, min(n, cap))].copy(), ghost=ghost,
                biggest_area=big.value, n_contours=nc.value)


def get_perspective_transform(src, dst):
    src = np.ascontiguousarray(src, np.float32).reshape(4, 2)
    dst = np.ascontiguousarray(dst, np.float32).reshape(4, 2)
    M = np.zeros(9, np.float64)
    rc = lib().ora_get_perspective_transform(_vp(src), _vp(dst), _vp(M))
    if rc != 0:
        raise ValueError("singular")
    return M.reshape(3, 3)


def warp_perspective(img, M, dsize=(380, 380)):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    cn = 1 if img.ndim == 2 else img.shape[2]
    M = np.ascontiguousarray(M, np.float64).reshape(9)
    out = np.zeros((dsize[1], dsize[0]) + ((cn,) if img.ndim == 3 else ()), np.uint8)
    lib().ora_warp_perspective(_vp(img), h, w, cn, _vp(M), int(dsize[0]), int(dsize[1]), _vp(out), None)
    return out


def i420_to_bgr(i420, h, w):
    """one frame: flat I420 buffer of h*w*3/2 bytes -> (h, w, 3) BGR"""
    buf = np.ascontiguousarray(i420, np.uint8).reshape(-1)
    assert buf.size == h * w * 3 // 2 and h % 2 == 0 and w % 2 == 0
    out = np.empty((h, w, 3), np.uint8)
    lib().ora_i420_to_bgr(_vp(buf), h, w, _vp(out))
    return out


class MOG2:
    def __init__(self, h, w, cn=3):
        self.h, self.w, self.cn = h, w, cn
        self._p = C.c_void_p(lib().ora_mog2_create(h, w, cn))

    def apply(self, img, learning_rate):
        img = np.ascontiguousarray(img, np.uint8)
        fg = np.empty((self.h, self.w), np.uint8)
        lib().ora_mog2_apply(self._p, _vp(img), C.c_double(learning_rate), _vp(fg))
        return fg

    def __del__(self):
        try:
            lib().ora_mog2_destroy(self._p)
        except Exception:
            pass


class _W(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("c1w", "c1b", "c2w", "c2b", "c3w", "c3b", "c4w", "c4b", "d1w", "d1b", "d2w", "d2b")]


WEIGHT_SHAPES = dict(c1w=(5, 5, 3, 32), c1b=(32,), c2w=(5, 5, 32, 32), c2b=(32,),
                     c3w=(3, 3, 32, 90), c3b=(90,), c4w=(3, 3, 90, 90), c4b=(90,),
                     d1w=(3240, 160), d1b=(160,), d2w=(160, 81), d2b=(81,))


def _wstruct(weights):
    keep = {}
    s = _W()
    for k, shp in WEIGHT_SHAPES.items():
        a = np.ascontiguousarray(weights[k], np.float32)
        assert a.shape == shp, (k, a.shape, shp)
        keep[k] = a
        setattr(s, k, a.ctypes.data)
    return s, keep


def cnn_forward(weights, patches, want_logits=False):
    patches = np.ascontiguousarray(patches, np.uint8).reshape(-1, 40, 40, 3)
    n = len(patches)
    s, keep = _wstruct(weights)
    y = np.zeros((n, 81), np.float32)
    lg = np.zeros((n, 81), np.float32)
    lib().ora_cnn_forward(C.byref(s), _vp(patches), n, _vp(y), _vp(lg))
    return (y, lg) if want_logits else y


def cnn_region_maps(weights, goban):
    """the 100 regions of a goban image through the network -> (y (100, 81), pool2 (100, 16, 16, 32), pool4 (100, 6, 6, 90)):
    the softmax and the outputs of the two MaxPooling2D layers (nn_manager.py:286, 292)"""
    goban = np.ascontiguousarray(goban, np.uint8)
    assert goban.shape == (380, 380, 3)
    s, keep = _wstruct(weights)
    patches = np.zeros((100, 40, 40, 3), np.uint8)
    lib().ora_cnn_region_patches(_vp(goban), _vp(patches))
    y = np.zeros((100, 81), np.float32)
    p2 = np.zeros((100, 16, 16, 32), np.float32)
    p4 = np.zeros((100, 6, 6, 90), np.float32)
    lib().ora_cnn_forward_maps(C.byref(s), _vp(patches), 100, _vp(y), None, _vp(p2), _vp(p4))
    return y, p2, p4


def cnn_predict_regions(weights, goban, want_logits=False):
    goban = np.ascontiguousarray(goban, np.uint8)
    assert goban.shape == (380, 380, 3)
    s, keep = _wstruct(weights)
    y = np.zeros((100, 81), np.float32)
    lg = np.zeros((100, 81), np.float32)
    lib().ora_cnn_predict_regions(C.byref(s), _vp(goban), _vp(y), _vp(lg))
    return (y, lg) if want_logits else y


def baseline_frames(weights, frames, M, threads):
    """bench.py's cpu_baseline leg: the hot path of n frames with OpenMP ACROSS frames, one frame per
    thread.  weights None: board path + warp only, the goban images are returned for the caller's classifier
    -> (per-frame ora_board_lines status / line count, labels (n, 19, 19) or gobans (n, 380, 380, 3), threads that took part)"""
    frames = np.ascontiguousarray(frames, np.uint8)
    n, h, w, _ = frames.shape
    M = np.ascontiguousarray(M, np.float64).reshape(9)
    nl = np.zeros(n, np.int32)
    if weights is None:
        out = np.zeros((n, 380, 380, 3), np.uint8)
        used = lib().ora_baseline_frames(_vp(frames), n, h, w, _vp(M), None, int(threads), _vp(nl), None, _vp(out))
    else:
        s, keep = _wstruct(weights)
        out = np.zeros((n, 19, 19), np.uint8)
        used = lib().ora_baseline_frames(_vp(frames), n, h, w, _vp(M), C.byref(s), int(threads), _vp(nl), _vp(out), None)
    if used < 0:
        raise MemoryError("ora_baseline_frames")
    return nl, out, used


def decode_all(y):
    y = np.ascontiguousarray(y, np.float32).reshape(100, 81)
    labels = np.zeros((19, 19), np.uint8)
    conf = np.zeros((19, 19), np.float64)
    lib().ora_decode_all(_vp(y), _vp(labels), _vp(conf))
    return labels, conf


def decode_regions(y):
    """per region: label = argmax(y) (first maximum) and confidence = max(y) / sum(y), the sum running over the 81
    values in index order in float64 (NNCache.predict_4_stones / predict_stone, nn_cache.py:16-31)"""
    y = np.ascontiguousarray(y, np.float32).reshape(100, 81)
    lab = np.argmax(y, axis=1).astype(np.uint8)
    conf = np.zeros(100, np.float64)
    for t in range(100):
        tot = 0.0
        for v in y[t]:
            tot = tot + float(v)
        conf[t] = float(y[t, lab[t]]) / tot
    return lab, conf


def zone_counts(mask):
    """foreground pixels of a 380x380 mask per StonesFinder.getrect(r, c) zone -> int32 (19, 19):
    np.sum(fg[x0:x1, y0:y1]) / 255 of SfNeural.is_agitated (sf_neural.py:178-180)"""
    out = np.zeros((GSIZE, GSIZE), np.int32)
    for r in range(GSIZE):
        for c in range(GSIZE):
            x0, y0, x1, y1 = sf_getrect(r, c)
            out[r, c] = int(np.count_nonzero(mask[x0:x1, y0:y1]))
    return out


# ---- pure-python restatements of the codec / geometry helpers (pinned by the reference's
# ---- own known-answer tests, tests/golden/reference_known_answers.json) ------------------

GSIZE = 19


def compute_stones(label, dimension=4):
    """nn_manager.py:246-254 -> list of 'E'/'B'/'W'."""
    k = label
    stones = [None] * dimension
    for i in reversed(range(dimension)):
        digit = int(k / (3 ** i))
        stones[i] = 'E' if digit == 0 else 'B' if digit == 1 else 'W'
        k %= 3 ** i
    return stones


def compute_label(stones4):
    """nn_manager.py:236-244 for a flat list of 4 colours in (r, c) raster order."""
    colors = {'E': 0, 'B': 1, 'W': 2}
    return sum(colors[s] * 3 ** p for p, s in enumerate(stones4))


def class_indices(nb_classes=81):
    """nn_manager.py:360-382 -> array (4, 3, 27)."""
    import math
    dimension = int(math.log(nb_classes, 3))
    binar = [compute_stones(c, dimension) for c in range(nb_classes)]
    out = np.zeros((dimension, 3, nb_classes // 3), np.uint8)
    for d in range(dimension):
        for ci, col in enumerate("EBW"):
            out[d, ci] = [c for c in range(nb_classes) if binar[c][d] == col]
    return out


def subregion(i, j, split=10):
    """nn_manager.py:92-126."""
    step = (GSIZE + 1) // split
    rs, re = i * step, (i + 1) * step
    if GSIZE - rs < step:
        rs, re = GSIZE - step, GSIZE
    cs, ce = j * step, (j + 1) * step
    if GSIZE - cs < step:
        cs, ce = GSIZE - step, GSIZE
    return rs, re, cs, ce


def nn_rect(rs, re, cs, ce, size=380, width=40):
    """nn_manager.py:256-275 (getrect + _get_rect_nn) -> x0, x1, y0, y1."""
    x0, y0 = int(rs * size / GSIZE), int(cs * size / GSIZE)
    x1, y1 = int(re * size / GSIZE), int(ce * size / GSIZE)
    if x1 - x0 != width:
        x0 = x1 - width
    if y1 - y0 != width:
        y0 = y1 - width
    return x0, x1, y0, y1


def posgrid(size=380):
    """stonesfinder.py:964-981 PosGrid.__init__ -> (19,19,2) int16."""
    mtx = np.zeros((GSIZE, GSIZE, 2), np.int16)
    start = size / GSIZE / 2
    end = size - start
    hull = [(start, start), (end, start), (end, end), (start, end)]
    g = GSIZE
    for i in range(g):
        xup = (hull[0][0] * (g - 1 - i) + hull[1][0] * i) / (g - 1)
        xdown = (hull[3][0] * (g - 1 - i) + hull[2][0] * i) / (g - 1)
        for j in range(g):
            mtx[i][j][0] = (xup * (g - 1 - j) + xdown * j) / (g - 1)
            yleft = (hull[0][1] * (g - 1 - j) + hull[3][1] * j) / (g - 1)
            yright = (hull[1][1] * (g - 1 - j) + hull[2][1] * j) / (g - 1)
            mtx[i][j][1] = (yleft * (g - 1 - i) + yright * i) / (g - 1)
    return mtx


def sf_getrect(r, c, cursor=1.0, size=380):
    """stonesfinder.py:412-450 StonesFinder.getrect."""
    grid = posgrid(size)
    p = grid[r][c]
    pbefore = grid[r - 1][c - 1].copy()
    pafter = grid[min(r + 1, GSIZE - 1)][min(c + 1, GSIZE - 1)].copy()
    if r == 0:
        pbefore[0] = -p[0]
    elif r == GSIZE - 1:
        pafter[0] = 2 * size - p[0] - 2
    if c == 0:
        pbefore[1] = -p[1]
    elif c == GSIZE - 1:
        pafter[1] = 2 * size - p[1] - 2
    w = cursor / 2
    x0 = max(0, int(w * pbefore[0] + (1 - w) * p[0]))
    y0 = max(0, int(w * pbefore[1] + (1 - w) * p[1]))
    x1 = min(size, int((1 - w) * p[0] + w * pafter[0]))
    y1 = min(size, int((1 - w) * p[1] + w * pafter[1]))
    return x0, y0, x1, y1
