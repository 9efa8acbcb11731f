"""ctypes binding of libck_hip.so -- the C-ABI declared in include/camkifu_amd.h.

This is the ONLY compute backend of the package: there is no CPU fallback.  If the shared
library has not been built, or no HIP device is present, the calls raise.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CK_HIP_LIB: developer override used to A/B kernel variants (tools/ab_variant.sh); never set in production
SO_PATH = os.environ.get("CK_HIP_LIB") or os.path.join(_HERE, "libck_hip.so")
CSRC = os.path.join(_HERE, "csrc")

CK_HOST, CK_DEVICE = 0, 1
CK_CNN_FP32, CK_CNN_BF16, CK_CNN_F16X2 = 0, 1, 2
CK_CNN_DEFAULT = CK_CNN_F16X2            # what a new context computes in
CK_BOARD_LINES, CK_BOARD_NO_CONTOUR, CK_BOARD_TOO_SMALL = 0, 1, 2

EXPORTS = [
    "ck_ctx_create", "ck_ctx_destroy", "ck_last_error", "ck_backend", "ck_version", "ck_stream",
    "ck_timing_enable", "ck_timing_reset", "ck_timing_get",
    "ck_median15", "ck_median", "ck_canny", "ck_goban_canny", "ck_board_edges", "ck_board_lines", "ck_board_detect",
    "ck_i420_to_bgr", "ck_get_perspective_transform", "ck_warp_perspective",
    "ck_mog2_create", "ck_mog2_apply", "ck_mog2_destroy",
    "ck_cnn_set_weights", "ck_cnn_set_mode", "ck_cnn_predict", "ck_stones_detect",
]

WEIGHT_ORDER = ("c1w", "c1b", "c2w", "c2b", "c3w", "c3b", "c4w", "c4b", "d1w", "d1b", "d2w", "d2b")
WEIGHT_SHAPES = dict(c1w=(5, 5, 3, 32), c1b=(32,), c2w=(5, 5, 32, 32), c2b=(32,),
                     c3w=(3, 3, 32, 90), c3b=(90,), c4w=(3, 3, 90, 90), c4b=(90,),
                     d1w=(3240, 160), d1b=(160,), d2w=(160, 81), d2b=(81,))


class BoardResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_contours", C.c_int32), ("n_lines", C.c_int32),
                ("reserved", C.c_int32), ("biggest_area", C.c_double)]


BOARD_DTYPE = np.dtype([("status", "<i4"), ("n_contours", "<i4"), ("n_lines", "<i4"), ("reserved", "<i4"),
                        ("biggest_area", "<f8")])          # ck_board_result, 24 bytes


class CkError(RuntimeError):
    pass


def build(force=False):
    """Compile every HIP source for gfx950 into camkifu_amd/libck_hip.so (in-tree)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean", "-s"])
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j8"])
    return SO_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise CkError("libck_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                          "camkifu_amd has no CPU fallback")
        # PyTorch-ROCm wheels bundle their own HIP runtime; two HIP runtimes in one process
        # cannot both own the GPU.  Importing torch first makes the loader resolve
        # libck_hip.so's libamdhip64.so.7 dependency to the copy torch already mapped, so
        # torch tensors and this library share one runtime (and one device context).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(SO_PATH)
        L.ck_last_error.restype = C.c_char_p
        L.ck_last_error.argtypes = [C.c_void_p]
        L.ck_stream.restype = C.c_void_p
        L.ck_stream.argtypes = [C.c_void_p]
        L.ck_ctx_destroy.argtypes = [C.c_void_p]
        L.ck_ctx_destroy.restype = None
        _lib = L
    return _lib


def _is_torch(a):
    return type(a).__module__.startswith("torch")


def _in(a, dtype=np.uint8):
    """-> (pointer, space, keepalive) for a numpy array or a torch tensor."""
    if _is_torch(a):
        assert a.is_contiguous()
        if a.is_cuda:
            # the library works on its own HIP stream: whatever torch still has queued that produces
            # this tensor must have finished before the pointer is handed over
            import torch
            torch.cuda.current_stream(a.device).synchronize()
            return C.c_void_p(a.data_ptr()), CK_DEVICE, a
        a = a.numpy()
    a = np.ascontiguousarray(a, dtype)
    return a.ctypes.data_as(C.c_void_p), CK_HOST, a


class Context:
    """One ck_ctx: a HIP stream plus scratch buffers.  One per finder instance / thread."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = lib().ck_ctx_create(int(device), C.byref(self._h))
        if rc != 0:
            raise CkError("ck_ctx_create failed: " + (lib().ck_last_error(None) or b"").decode())
        self.device = device

    def close(self):
        if self._h:
            lib().ck_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise CkError("libck_hip error %d: %s" % (rc, (lib().ck_last_error(self._h) or b"").decode()))

    # ---- timing -------------------------------------------------------------------------
    def timing_enable(self, on=True):
        self._chk(lib().ck_timing_enable(self._h, int(on)))

    def timing_reset(self):
        self._chk(lib().ck_timing_reset(self._h))

    def timing_get(self, name):
        ms, cnt = C.c_double(0), C.c_int(0)
        self._chk(lib().ck_timing_get(self._h, name.encode(), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    # ---- helpers ---------------------------------------------------------------------------
    @staticmethod
    def _shape(img, cn):
        shp = tuple(img.shape)
        if len(shp) == 2 + (cn > 1):
            shp = (1,) + shp
        n, h, w = shp[0], shp[1], shp[2]
        if cn > 1:
            assert shp[3] == cn, shp
        return n, h, w

    def _out(self, like, shape, dtype):
        """allocate an output in the same memory space as `like`"""
        if _is_torch(like) and like.is_cuda:
            import torch
            t = torch.empty(shape, dtype=getattr(torch, np.dtype(dtype).name), device=like.device)
            return t, C.c_void_p(t.data_ptr()), CK_DEVICE
        a = np.empty(shape, dtype)
        return a, a.ctypes.data_as(C.c_void_p), CK_HOST

    # ---- K1 ---------------------------------------------------------------------------------
    def median15(self, bgr):
        n, h, w = self._shape(bgr, 3)
        p, sp, keep = _in(bgr)
        out, op, osp = self._out(bgr, tuple(bgr.shape), np.uint8)
        self._chk(lib().ck_median15(self._h, p, n, h, w, sp, op, osp))
        return out

    def median(self, bgr, ksize):
        """cv2.medianBlur(bgr, ksize) for odd ksize in 3..17 (same matrix-core kernel, other window)"""
        n, h, w = self._shape(bgr, 3)
        p, sp, keep = _in(bgr)
        out, op, osp = self._out(bgr, tuple(bgr.shape), np.uint8)
        self._chk(lib().ck_median(self._h, p, n, h, w, int(ksize), sp, op, osp))
        return out

    def goban_canny(self, bgr, want_otsu=False):
        """SfContours.get_canny (stone/sf_contours.py:332-340): medianBlur 13 then 7, Otsu level of the grey image,
        Canny(median, otsu / 2, otsu) -> edges (and the Otsu levels)"""
        n, h, w = self._shape(bgr, 3)
        p, sp, keep = _in(bgr)
        edges, ep, osp = self._out(bgr, tuple(bgr.shape[:-1]), np.uint8)
        otsu = np.zeros(n, np.float64)
        self._chk(lib().ck_goban_canny(self._h, p, n, h, w, sp, ep, osp, otsu.ctypes.data_as(C.c_void_p)))
        return (edges, otsu) if want_otsu else edges

    # ---- K2 ---------------------------------------------------------------------------------
    def canny(self, img3, low=25, high=75, want_map=False):
        n, h, w = self._shape(img3, 3)
        p, sp, keep = _in(img3)
        oshape = tuple(img3.shape[:-1])
        edges, ep, osp = self._out(img3, oshape, np.uint8)
        m, mp = None, None
        if want_map:
            m, mp, _ = self._out(img3, oshape, np.uint8)
        self._chk(lib().ck_canny(self._h, p, n, h, w, sp, int(low), int(high), ep, mp, osp))
        return (edges, m) if want_map else edges

    def board_edges(self, bgr):
        n, h, w = self._shape(bgr, 3)
        p, sp, keep = _in(bgr)
        edges, ep, osp = self._out(bgr, tuple(bgr.shape[:-1]), np.uint8)
        self._chk(lib().ck_board_edges(self._h, p, n, h, w, sp, ep, osp))
        return edges

    # ---- K3..K6 -------------------------------------------------------------------------------
    def _lines_out(self, n, cap, lines, res):
        out = []
        for f in range(n):
            k = min(res[f].n_lines, cap)
            out.append(dict(status=res[f].status, n_contours=res[f].n_contours, n_lines=res[f].n_lines,
                            biggest_area=res[f].biggest_area, lines=lines[f, :k].copy()))
        return out

    def board_lines(self, edges, hough_thresh=-1, cap=1024, want_ghost=False):
        n, h, w = self._shape(edges, 1)
        p, sp, keep = _in(edges)
        lines = np.zeros((n, cap, 2), np.float32)
        res = (BoardResult * n)()
        ghost, gp, gsp = (None, None, CK_HOST)
        if want_ghost:
            ghost, gp, gsp = self._out(edges, tuple(edges.shape), np.uint8)
        self._chk(lib().ck_board_lines(self._h, p, n, h, w, sp, int(hough_thresh),
                                       lines.ctypes.data_as(C.c_void_p), cap, res, gp, gsp))
        out = self._lines_out(n, cap, lines, res)
        return (out, ghost) if want_ghost else out

    def board_detect(self, bgr, hough_thresh=-1, cap=1024, raw=False):
        """raw=True: (structured array of BOARD_DTYPE, lines (n, cap, 2)) without per-frame python objects"""
        n, h, w = self._shape(bgr, 3)
        p, sp, keep = _in(bgr)
        lines = np.zeros((n, cap, 2), np.float32)
        res = np.zeros(n, BOARD_DTYPE)
        self._chk(lib().ck_board_detect(self._h, p, n, h, w, sp, int(hough_thresh),
                                        lines.ctypes.data_as(C.c_void_p), cap, res.ctypes.data_as(C.c_void_p)))
        if raw:
            return res, lines
        return [dict(status=int(r["status"]), n_contours=int(r["n_contours"]), n_lines=int(r["n_lines"]),
                     biggest_area=float(r["biggest_area"]), lines=lines[f, :min(int(r["n_lines"]), cap)].copy())
                for f, r in enumerate(res)]

    # ---- frame source ------------------------------------------------------------------------
    def i420_to_bgr(self, i420, h, w, to_device=None):
        """planar YUV 4:2:0 frames (n, h*w*3/2) or one flat frame -> BGR (n, h, w, 3) / (h, w, 3).
        `to_device`: a torch device to leave the BGR frames in HBM even when the I420 bytes come from
        host memory (the fast-file path uploads 1.5 B/px and converts on the GPU)."""
        fsz = h * w * 3 // 2
        single = len(i420.shape) == 1
        n = 1 if single else int(i420.shape[0])
        assert int(i420.shape[-1]) == fsz, (tuple(i420.shape), fsz)
        p, sp, keep = _in(i420)
        oshape = (h, w, 3) if single else (n, h, w, 3)
        if to_device is not None and sp == CK_HOST:
            import torch
            out = torch.empty(oshape, dtype=torch.uint8, device=to_device)
            op, osp = C.c_void_p(out.data_ptr()), CK_DEVICE
        else:
            out, op, osp = self._out(i420, oshape, np.uint8)
        self._chk(lib().ck_i420_to_bgr(self._h, p, n, int(h), int(w), sp, op, osp))
        return out

    # ---- K8 ---------------------------------------------------------------------------------
    def warp_perspective(self, bgr, M, dsize=380):
        n, h, w = self._shape(bgr, 3)
        p, sp, keep = _in(bgr)
        M = np.ascontiguousarray(M, np.float64).reshape(-1, 9)
        oshape = (dsize, dsize, 3) if len(bgr.shape) == 3 else (n, dsize, dsize, 3)
        out, op, osp = self._out(bgr, oshape, np.uint8)
        self._chk(lib().ck_warp_perspective(self._h, p, n, h, w, sp, M.ctypes.data_as(C.c_void_p), len(M),
                                            int(dsize), op, osp))
        return out

    # ---- K9 ---------------------------------------------------------------------------------
    def mog2_create(self, h=380, w=380):
        hd = C.c_int(-1)
        self._chk(lib().ck_mog2_create(self._h, h, w, C.byref(hd)))
        return hd.value

    def mog2_apply(self, handle, img3, learning_rate):
        p, sp, keep = _in(img3)
        fg, fp_, osp = self._out(img3, tuple(img3.shape[:-1]), np.uint8)
        self._chk(lib().ck_mog2_apply(self._h, int(handle), p, sp, C.c_double(learning_rate), fp_, osp))
        return fg

    def mog2_destroy(self, handle):
        self._chk(lib().ck_mog2_destroy(self._h, int(handle)))

    # ---- K10..K12 ----------------------------------------------------------------------------
    def cnn_set_weights(self, weights):
        """weights: dict name -> float32 array (numpy, or torch tensors all on cuda / all on cpu)."""
        ptrs = (C.c_void_p * 12)()
        keep, space = [], None
        for i, k in enumerate(WEIGHT_ORDER):
            a = weights[k]
            assert tuple(a.shape) == WEIGHT_SHAPES[k], (k, tuple(a.shape))
            if _is_torch(a):
                import torch
                assert a.dtype == torch.float32
            p, sp, ka = _in(a, np.float32)
            assert space in (None, sp), "weights must live in one memory space"
            space = sp
            ptrs[i] = p
            keep.append(ka)
        self._chk(lib().ck_cnn_set_weights(self._h, ptrs, space))

    def cnn_set_mode(self, mode):
        self._chk(lib().ck_cnn_set_mode(self._h, int(mode)))

    def cnn_predict(self, goban, want_y=True):
        shp = tuple(goban.shape)
        n = 1 if len(shp) == 3 else shp[0]
        p, sp, keep = _in(goban)
        y, yp = None, None
        if want_y:
            y, yp, _ = self._out(goban, (n, 100, 81), np.float32)
        labels, lp, osp = self._out(goban, (n, 19, 19), np.uint8)
        conf, cp, _ = self._out(goban, (n, 19, 19), np.float64)
        self._chk(lib().ck_cnn_predict(self._h, p, n, sp, yp, lp, cp, osp))
        return (y, labels, conf) if want_y else (labels, conf)

    def stones_detect(self, bgr, M):
        n, h, w = self._shape(bgr, 3)
        p, sp, keep = _in(bgr)
        M = np.ascontiguousarray(M, np.float64).reshape(-1, 9)
        labels, lp, osp = self._out(bgr, (n, 19, 19), np.uint8)
        conf, cp, _ = self._out(bgr, (n, 19, 19), np.float64)
        self._chk(lib().ck_stones_detect(self._h, p, n, h, w, sp, M.ctypes.data_as(C.c_void_p), len(M), lp, cp, osp))
        return labels, conf


def get_perspective_transform(src4, dst4):
    """K7, host only (board/boardfinder.py:43-45)."""
    src = np.ascontiguousarray(src4, np.float32).reshape(4, 2)
    dst = np.ascontiguousarray(dst4, np.float32).reshape(4, 2)
    M = np.zeros(9, np.float64)
    rc = lib().ck_get_perspective_transform(src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p),
                                            M.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise CkError("degenerate quadrilateral")
    return M.reshape(3, 3)


_default_ctx = {}


def get_context(device=0):
    """Process-wide default context per device (finders create their own)."""
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
