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
CK_CNN_FP32, CK_CNN_BF16, CK_CNN_F16X2, CK_CNN_F16Q8 = 0, 1, 2, 3
CK_CNN_DEFAULT = CK_CNN_F16X2            # what a new context computes in
CK_BOARD_LINES, CK_BOARD_NO_CONTOUR, CK_BOARD_TOO_SMALL = 0, 1, 2

ZONE_LINES = 32        # CK_ZONE_LINES

EXPORTS = [
    "ck_ctx_create", "ck_ctx_create_prio", "ck_ctx_destroy", "ck_ctx_destroy2", "ck_stream_wait", "ck_last_error", "ck_backend", "ck_version", "ck_stream",
    "ck_timing_enable", "ck_timing_reset", "ck_timing_get",
    "ck_median15", "ck_median", "ck_canny", "ck_goban_canny", "ck_board_edges", "ck_board_lines", "ck_board_detect",
    "ck_i420_to_bgr", "ck_get_perspective_transform", "ck_warp_perspective",
    "ck_mog2_create", "ck_mog2_apply", "ck_mog2_destroy",
    "ck_cnn_set_weights", "ck_cnn_set_mode", "ck_cnn_predict", "ck_cnn_maps", "ck_stones_detect",
    "ck_cnn_regions", "ck_stones_run", "ck_zone_counts", "ck_mog2_band_run",
    "ck_board_detect_records", "ck_cnn_regions_records",
    "ck_contour_stones", "ck_contours_external", "ck_find_intersections", "ck_update_grid",
    "ck_ordered_hull", "ck_boardfold_create", "ck_boardfold_destroy", "ck_boardfold_reset", "ck_boardfold_step", "ck_boardfold_run", "ck_round10", "ck_round10_reference",
    "ck_policy_create", "ck_policy_destroy", "ck_policy_run", "ck_policy_run_records", "ck_policy_get_state", "ck_policy_set_state",
    "ck_policy_watch",
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


REC_LMAX = 64                                               # CK_REC_LMAX
REC_LINES_CUT, REC_FAILED = 1, 2                            # CK_REC_*
REC_DTYPE = np.dtype([("status", "<i4"), ("n_contours", "<i4"), ("n_lines", "<i4"), ("flags", "<i4"),
                      ("biggest_area", "<f8"), ("lines", "<f4", (REC_LMAX, 2)),
                      ("region_conf", "<f8", (10, 10)), ("region_label", "u1", (10, 10)), ("pad", "u1", (4,))])   # ck_frame_record
REC_BYTES = REC_DTYPE.itemsize
assert REC_BYTES == 1440


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
        L.ck_ctx_destroy2.argtypes = [C.c_void_p]
        L.ck_stream_wait.argtypes = [C.c_void_p, C.c_void_p]
        for fn in (L.ck_boardfold_destroy, L.ck_policy_destroy):
            fn.argtypes = [C.c_void_p]
            fn.restype = None
        L.ck_boardfold_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_longlong,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ck_policy_run.argtypes = [C.c_void_p, C.c_int, C.c_longlong] + [C.c_void_p] * 8 + [C.c_int, C.c_void_p]
        L.ck_policy_run_records.argtypes = [C.c_void_p, C.c_int, C.c_longlong] + [C.c_void_p] * 8 + [C.c_int, C.c_void_p]
        for fn in (L.ck_round10, L.ck_round10_reference):
            fn.argtypes, fn.restype = [C.c_double], C.c_double
        L.ck_boardfold_run.argtypes = ([C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int]
                                       + [C.c_void_p] * 5)
        L.ck_policy_get_state.argtypes = [C.c_void_p] * 6
        L.ck_policy_set_state.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.ck_policy_watch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_longlong]
        _lib = L
    return _lib


def _is_torch(a):
    return type(a).__module__.startswith("torch")


def _in(a, dtype=np.uint8):
    """-> (pointer, space, keepalive) for a numpy array or a torch tensor.  A DEVICE tensor must go through Context._in /
    Context._out instead (they order the context's stream behind torch's): this function refuses it."""
    if _is_torch(a):
        assert a.is_contiguous()
        if a.is_cuda:
            raise CkError("a device tensor is handed to the library through Context._in / Context._out (stream-ordered hand-over)")
        a = a.numpy()
    a = np.ascontiguousarray(a, dtype)
    return a.ctypes.data_as(C.c_void_p), CK_HOST, a


class Context:
    """One ck_ctx: a HIP stream plus scratch buffers.  One per finder instance / thread."""

    def __init__(self, device=0, priority=0):
        """priority: HIP stream priority of the context's stream -- 0 normal, 1 highest, -1 lowest"""
        self._h = C.c_void_p()
        rc = lib().ck_ctx_create_prio(int(device), int(priority), C.byref(self._h))
        if rc != 0:
            raise CkError("ck_ctx_create failed: " + (lib().ck_last_error(None) or b"").decode())
        self.device = device

    def close(self):
        """free the context.  A context another thread is still inside (for more than 5 s) is NOT freed and keeps its
        handle: CkError says so, and a later close() -- or the finaliser -- tries again (the handle used to be
        dropped regardless, the stream and all scratch HBM leaked with no way to retry)."""
        if self._h:
            if lib().ck_ctx_destroy2(self._h) != 0:
                raise CkError("the context is still inside a call on another thread: not freed, close() it again later")
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

    # Hand-over of DEVICE memory (torch tensors) -- the ONE place where it happens.  The library launches on the context's
    # own HIP stream; torch queues its kernels on the calling thread's current stream and reuses a freed block at once for
    # the next allocation on that stream, from any thread.  So before a pointer crosses the C-ABI the context's stream is
    # ordered behind torch's current stream (ck_stream_wait: an event recorded there, waited for here -- no host wait):
    # an input's producer has run, and so has the last kernel torch queued on a recycled output block (round 4: a memory
    # fault in a gather whose index tensor had been recycled into an output buffer).  Results need nothing in the other
    # direction: every entry point returns with its work complete.
    def _order_behind_torch(self, device):
        import torch
        self._chk(lib().ck_stream_wait(self._h, C.c_void_p(torch.cuda.current_stream(device).cuda_stream)))

    def _in(self, a, dtype=np.uint8):
        """-> (pointer, space, keepalive) for a numpy array or a torch tensor"""
        if _is_torch(a) and a.is_cuda:
            assert a.is_contiguous()
            self._order_behind_torch(a.device)
            return C.c_void_p(a.data_ptr()), CK_DEVICE, a
        return _in(a, dtype)

    def _out(self, like, shape, dtype):
        """allocate an output in the same memory space as `like`"""
        if _is_torch(like) and like.is_cuda:
            return self._out_on(like.device, shape, dtype)
        a = np.empty(shape, dtype)
        return a, a.ctypes.data_as(C.c_void_p), CK_HOST

    def _out_on(self, device, shape, dtype):
        """a fresh device tensor the library may write: allocated, THEN the context's stream ordered behind torch's"""
        import torch
        t = torch.empty(shape, dtype=getattr(torch, np.dtype(dtype).name), device=device)
        self._order_behind_torch(t.device)
        return t, C.c_void_p(t.data_ptr()), CK_DEVICE

    # ---- K1 ---------------------------------------------------------------------------------
    def median15(self, bgr):
        n, h, w = self._shape(bgr, 3)
        p, sp, keep = self._in(bgr)
        out, op, osp = self._out(bgr, tuple(bgr.shape), np.uint8)
        self._chk(lib().ck_median15(self._h, p, n, h, w, sp, op, osp))
        return out

    def median(self, bgr, ksize):
        """cv2.medianBlur(bgr, ksize) for odd ksize in 3..17 (same matrix-core kernel, other window)"""
        n, h, w = self._shape(bgr, 3)
        p, sp, keep = self._in(bgr)
        out, op, osp = self._out(bgr, tuple(bgr.shape), np.uint8)
        self._chk(lib().ck_median(self._h, p, n, h, w, int(ksize), sp, op, osp))
        return out

    def goban_canny(self, bgr, want_otsu=False):
        """SfContours.get_canny (stone/sf_contours.py:332-340): medianBlur 13 then 7, Otsu level of the grey image,
        Canny(median, otsu / 2, otsu) -> edges (and the Otsu levels)"""
        n, h, w = self._shape(bgr, 3)
        p, sp, keep = self._in(bgr)
        edges, ep, osp = self._out(bgr, tuple(bgr.shape[:-1]), np.uint8)
        otsu = np.zeros(n, np.float64)
        self._chk(lib().ck_goban_canny(self._h, p, n, h, w, sp, ep, osp, otsu.ctypes.data_as(C.c_void_p)))
        return (edges, otsu) if want_otsu else edges

    # ---- K2 ---------------------------------------------------------------------------------
    def canny(self, img3, low=25, high=75, want_map=False):
        n, h, w = self._shape(img3, 3)
        p, sp, keep = self._in(img3)
        oshape = tuple(img3.shape[:-1])
        edges, ep, osp = self._out(img3, oshape, np.uint8)
        m, mp = None, None
        if want_map:
            m, mp, _ = self._out(img3, oshape, np.uint8)
        self._chk(lib().ck_canny(self._h, p, n, h, w, sp, int(low), int(high), ep, mp, osp))
        return (edges, m) if want_map else edges

    def board_edges(self, bgr):
        n, h, w = self._shape(bgr, 3)
        p, sp, keep = self._in(bgr)
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
        p, sp, keep = self._in(edges)
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
        p, sp, keep = self._in(bgr)
        lines = np.zeros((n, cap, 2), np.float32)
        res = np.zeros(n, BOARD_DTYPE)
        self._chk(lib().ck_board_detect(self._h, p, n, h, w, sp, int(hough_thresh),
                                        lines.ctypes.data_as(C.c_void_p), cap, res.ctypes.data_as(C.c_void_p)))
        if raw:
            return res, lines
        return [dict(status=int(r["status"]), n_contours=int(r["n_contours"]), n_lines=int(r["n_lines"]),
                     biggest_area=float(r["biggest_area"]), lines=lines[f, :min(int(r["n_lines"]), cap)].copy())
                for f, r in enumerate(res)]

    # ---- per-frame result records, written in place (host memory or HBM) -------------------------------------------
    def wait_stream(self, stream):
        """everything queued on `stream` (a torch.cuda.Stream) up to now runs before anything this context launches from now
        on -- for a tensor another thread produced on a stream that is not this thread's current one (pipeline._take)"""
        self._chk(lib().ck_stream_wait(self._h, C.c_void_p(stream.cuda_stream)))

    def _rec(self, rec, n):
        """-> (pointer, space, keepalive) of n records: a C-contiguous numpy array of REC_DTYPE, or a contiguous torch uint8
        tensor (n, REC_BYTES) in HBM"""
        if _is_torch(rec):
            if tuple(rec.shape) != (n, REC_BYTES) or not rec.is_contiguous() or rec.element_size() != 1:
                raise ValueError("records: expected a contiguous uint8 tensor of shape (%d, %d), got %r" % (n, REC_BYTES, tuple(rec.shape)))
            if rec.is_cuda:
                return self._in(rec)
            rec = rec.numpy().view(REC_DTYPE).reshape(n)
        if rec.dtype != REC_DTYPE or rec.shape != (n,) or not rec.flags.c_contiguous or not rec.flags.writeable:
            raise ValueError("records: expected a writeable C-contiguous array of %d REC_DTYPE rows" % n)
        return rec.ctypes.data_as(C.c_void_p), CK_HOST, rec

    def board_detect_records(self, bgr, rec, hough_thresh=-1):
        """K1..K6 of the frames -> the board half of their records, in place (ck_board_detect_records)"""
        n, h, w = self._shape(bgr, 3)
        p, sp, keep = self._in(bgr)
        rp, rsp, keep2 = self._rec(rec, n)
        self._chk(lib().ck_board_detect_records(self._h, p, n, h, w, sp, int(hough_thresh), rp, rsp))
        return rec

    def cnn_regions_records(self, goban, rec):
        """K10..K12 of the goban images -> the stones half of their records, in place (ck_cnn_regions_records)"""
        shp = tuple(goban.shape)
        n = 1 if len(shp) == 3 else shp[0]
        p, sp, keep = self._in(goban)
        rp, rsp, keep2 = self._rec(rec, n)
        self._chk(lib().ck_cnn_regions_records(self._h, p, n, sp, rp, rsp))
        return rec

    # ---- frame source ------------------------------------------------------------------------
    def i420_to_bgr(self, i420, h, w, to_device=None, out=None):
        """planar YUV 4:2:0 frames (n, h*w*3/2) or one flat frame -> BGR (n, h, w, 3) / (h, w, 3).
        `to_device`: a torch device to leave the BGR frames in HBM even when the I420 bytes come from host memory (the
        fast-file path uploads 1.5 B/px and converts on the GPU); `out`: a preallocated BGR tensor / array to fill."""
        fsz = h * w * 3 // 2
        single = len(i420.shape) == 1
        n = 1 if single else int(i420.shape[0])
        assert int(i420.shape[-1]) == fsz, (tuple(i420.shape), fsz)
        p, sp, keep = self._in(i420)
        oshape = (h, w, 3) if single else (n, h, w, 3)
        if out is not None:
            assert tuple(out.shape) == oshape
            op, osp, _ = self._in(out)
        elif to_device is not None and sp == CK_HOST:
            out, op, osp = self._out_on(to_device, oshape, np.uint8)
        else:
            out, op, osp = self._out(i420, oshape, np.uint8)
        self._chk(lib().ck_i420_to_bgr(self._h, p, n, int(h), int(w), sp, op, osp))
        return out

    # ---- K8 ---------------------------------------------------------------------------------
    def warp_perspective(self, bgr, M, dsize=380, out=None):
        """`out`: an (n, dsize, dsize, 3) uint8 array / tensor in the same memory space as `bgr` to write into"""
        n, h, w = self._shape(bgr, 3)
        p, sp, keep = self._in(bgr)
        M = np.ascontiguousarray(M, np.float64).reshape(-1, 9)
        oshape = (dsize, dsize, 3) if len(bgr.shape) == 3 else (n, dsize, dsize, 3)
        if out is None:
            out, op, osp = self._out(bgr, oshape, np.uint8)
        else:
            assert tuple(out.shape) == oshape and (out.is_contiguous() if _is_torch(out) else out.flags.c_contiguous)
            op, osp, _ = self._in(out)
            assert osp == sp, "out must live where the frames live"
        self._chk(lib().ck_warp_perspective(self._h, p, n, h, w, sp, M.ctypes.data_as(C.c_void_p), len(M),
                                            int(dsize), op, osp))
        return out

    # ---- K9 ---------------------------------------------------------------------------------
    def mog2_create(self, h=380, w=380):
        hd = C.c_int(-1)
        self._chk(lib().ck_mog2_create(self._h, h, w, C.byref(hd)))
        return hd.value

    def mog2_apply(self, handle, img3, learning_rate):
        p, sp, keep = self._in(img3)
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
            p, sp, ka = self._in(a, np.float32)
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
        p, sp, keep = self._in(goban)
        y, yp = None, None
        if want_y:
            y, yp, _ = self._out(goban, (n, 100, 81), np.float32)
        labels, lp, osp = self._out(goban, (n, 19, 19), np.uint8)
        conf, cp, _ = self._out(goban, (n, 19, 19), np.float64)
        self._chk(lib().ck_cnn_predict(self._h, p, n, sp, yp, lp, cp, osp))
        return (y, labels, conf) if want_y else (labels, conf)

    def cnn_maps(self, goban):
        """the classifier's pooled filter maps as the context's mode computes them (host float32, channels-last):
        (pool2 (n, 100, 16, 16, 32), pool4 (n, 100, 6, 6, 90)) -- the outputs of the two MaxPooling2D layers of
        create_net (stone/nn_manager.py:280-292); n <= 128"""
        shp = tuple(goban.shape)
        n = 1 if len(shp) == 3 else shp[0]
        p, sp, keep = self._in(goban)
        p2 = np.empty((n, 100, 16, 16, 32), np.float32)
        p4 = np.empty((n, 100, 6, 6, 90), np.float32)
        self._chk(lib().ck_cnn_maps(self._h, p, n, sp, p2.ctypes.data_as(C.c_void_p), p4.ctypes.data_as(C.c_void_p)))
        return p2, p4

    def stones_detect(self, bgr, M):
        n, h, w = self._shape(bgr, 3)
        p, sp, keep = self._in(bgr)
        M = np.ascontiguousarray(M, np.float64).reshape(-1, 9)
        labels, lp, osp = self._out(bgr, (n, 19, 19), np.uint8)
        conf, cp, _ = self._out(bgr, (n, 19, 19), np.float64)
        self._chk(lib().ck_stones_detect(self._h, p, n, h, w, sp, M.ctypes.data_as(C.c_void_p), len(M), lp, cp, osp))
        return labels, conf

    # ---- ordered stones path (feeds PolicyCore) ---------------------------------------------------
    def cnn_regions(self, goban):
        """-> (region_label (n, 10, 10) uint8, region_conf (n, 10, 10) float64)"""
        shp = tuple(goban.shape)
        n = 1 if len(shp) == 3 else shp[0]
        p, sp, keep = self._in(goban)
        rl, rlp, osp = self._out(goban, (n, 10, 10), np.uint8)
        rc, rcp, _ = self._out(goban, (n, 10, 10), np.float64)
        self._chk(lib().ck_cnn_regions(self._h, p, n, sp, rlp, rcp, osp))
        return rl, rc

    def stones_run(self, bgr, M, mog2=None, learning_rates=None, want_grid=False):
        """ordered run of n consecutive frames of one stream -> dict(region_label, region_conf, fgcount[, labels, conf]);
        fgcount is None without a background model"""
        n, h, w = self._shape(bgr, 3)
        p, sp, keep = self._in(bgr)
        M = np.ascontiguousarray(M, np.float64).reshape(-1, 9)
        rl, rlp, osp = self._out(bgr, (n, 10, 10), np.uint8)
        rc, rcp, _ = self._out(bgr, (n, 10, 10), np.float64)
        fg, fgp, lr, lrp = None, None, None, None
        if mog2 is not None:
            fg, fgp, _ = self._out(bgr, (n, 19, 19), np.int32)
            lr = np.ascontiguousarray(learning_rates, np.float64).reshape(n)
            lrp = lr.ctypes.data_as(C.c_void_p)
        lab, labp, cf, cfp = None, None, None, None
        if want_grid:
            lab, labp, _ = self._out(bgr, (n, 19, 19), np.uint8)
            cf, cfp, _ = self._out(bgr, (n, 19, 19), np.float64)
        self._chk(lib().ck_stones_run(self._h, p, n, h, w, sp, M.ctypes.data_as(C.c_void_p), len(M),
                                      -1 if mog2 is None else int(mog2), lrp, rlp, rcp, fgp, labp, cfp, osp))
        return dict(region_label=rl, region_conf=rc, fgcount=fg, labels=lab, conf=cf)

    def mog2_band_run(self, handle, band, learning_rates, last_band):
        """band (n, band_h, 380, 3) of consecutive goban images -> int32 (n, zone rows, 19) foreground counts"""
        n, bh = int(band.shape[0]), int(band.shape[1])
        p, sp, keep = self._in(band)
        lr = np.ascontiguousarray(learning_rates, np.float64).reshape(n)
        out, op, osp = self._out(band, (n, (bh + 19) // 20, 19), np.int32)
        self._chk(lib().ck_mog2_band_run(self._h, int(handle), p, n, sp, lr.ctypes.data_as(C.c_void_p), int(bool(last_band)),
                                         op, osp))
        return out

    def zone_counts(self, mask):
        """(n, 380, 380) or (380, 380) mask -> int32 (n, 19, 19) / (19, 19) foreground pixels per intersection zone"""
        single = len(mask.shape) == 2
        n = 1 if single else int(mask.shape[0])
        p, sp, keep = self._in(mask)
        out, op, osp = self._out(mask, (19, 19) if single else (n, 19, 19), np.int32)
        self._chk(lib().ck_zone_counts(self._h, p, n, sp, op, osp))
        return out

    # ---- SfContours.find_stones -------------------------------------------------------------
    def contour_stones(self, goban, fg, rects, rs=0, re=19, cs=0, ce=19, want_all=False):
        """SfContours.find_stones (stone/sf_contours.py:48-111) for one goban image (side, side, 3) + its foreground mask
        (side, side), or a batch (n, ...) of them, host or device; rects = (19, 19, 4) StonesFinder.getrect table.
        -> stones uint8 (19, 19) / (n, 19, 19) of 0 E, 1 B, 2 W (and with want_all: zones int16 (.., re-rs, ce-cs, 4),
        the hull mask uint8 (.., hs, ws) of the analysed view)"""
        single = len(goban.shape) == 3
        n = 1 if single else int(goban.shape[0])
        side = int(goban.shape[-2])
        if tuple(goban.shape[-3:]) != (side, side, 3) or tuple(fg.shape[-2:]) != (side, side) or len(fg.shape) != len(goban.shape) - 1:
            raise ValueError("goban %r / foreground %r: expected (.., s, s, 3) and (.., s, s)" % (tuple(goban.shape), tuple(fg.shape)))
        rects = np.ascontiguousarray(rects, np.int32).reshape(19, 19, 4)
        p, sp, keep = self._in(goban)
        q, sq, keep2 = self._in(fg)
        if sp != sq:
            raise ValueError("goban image and foreground mask must live in the same space (both host or both device)")
        stones = np.zeros((n, 19, 19), np.uint8)
        zones = mask = None
        zp = mp = None
        if want_all:
            hs = int(rects[re - 1, ce - 1, 2] - rects[rs, cs, 0])
            ws = int(rects[re - 1, ce - 1, 3] - rects[rs, cs, 1])
            zones = np.zeros((n, re - rs, ce - cs, 4), np.int16)
            mask = np.zeros((n, max(hs, 0), max(ws, 0)), np.uint8)
            zp, mp = zones.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(C.c_void_p)
        self._chk(lib().ck_contour_stones(self._h, p, q, n, side, sp, rects.ctypes.data_as(C.c_void_p), int(rs), int(re), int(cs),
                                          int(ce), stones.ctypes.data_as(C.c_void_p), zp, mp))
        if single:
            return (stones[0], zones[0], mask[0]) if want_all else stones[0]
        return (stones, zones, mask) if want_all else stones

    def contours_external(self, edges, want_points=False):
        """cv2.findContours(edges, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) as SfContours reads it: for one (h, w) edge map or a
        batch (n, h, w) -> per map a list of dicts {start: (x, y), nvert, pix: (k, 2) int32 or None} in cv2's order"""
        single = len(edges.shape) == 2
        n = 1 if single else int(edges.shape[0])
        h, w = int(edges.shape[-2]), int(edges.shape[-1])
        p, sp, keep = self._in(edges)
        cap = n * (h * w // 4 + 1)
        counts = np.zeros(n, np.int32)
        table = np.zeros((cap, 4), np.int32)
        pts = np.zeros((n * h * w, 2), np.int32) if want_points else None
        self._chk(lib().ck_contours_external(self._h, p, n, h, w, sp, counts.ctypes.data_as(C.c_void_p),
                                             table.ctypes.data_as(C.c_void_p), cap,
                                             pts.ctypes.data_as(C.c_void_p) if want_points else None, n * h * w))
        out, k, o = [], 0, 0
        for f in range(n):
            lst = []
            for _ in range(int(counts[f])):
                x, y, nv, npx = (int(v) for v in table[k])
                lst.append(dict(start=(x, y), nvert=nv, pix=pts[o:o + npx].copy() if want_points else None))
                k += 1
                o += npx
            out.append(lst)
        return out[0] if single else out

    # ---- StonesFinder.find_intersections ----------------------------------------------------
    def find_intersections(self, goban, mtx, rects, want_lines=False):
        """StonesFinder.find_intersections (stone/stonesfinder.py:516-552) for one goban image (side, side, 3) or a batch,
        host or device; mtx (19, 19, 2) int16 PosGrid.mtx, rects (19, 19, 4) getrect table -> grid int16 (.., 19, 19, 2)
        (with want_lines also: per image a dict {(r, c): [(x0, y0, x1, y1), ..]} of the lines found, and the Canny map)"""
        single = len(goban.shape) == 3
        n = 1 if single else int(goban.shape[0])
        side = int(goban.shape[-2])
        if tuple(goban.shape[-3:]) != (side, side, 3):
            raise ValueError("goban %r: expected (.., s, s, 3)" % (tuple(goban.shape),))
        mtx = np.ascontiguousarray(mtx, np.int16).reshape(19, 19, 2)
        rects = np.ascontiguousarray(rects, np.int32).reshape(19, 19, 4)
        p, sp, keep = self._in(goban)
        grid = np.zeros((n, 19, 19, 2), np.int16)
        lines = counts = edges = None
        lp = cp = ep = None
        if want_lines:
            lines = np.zeros((n, 361, ZONE_LINES, 4), np.int16)
            counts = np.zeros((n, 361), np.int32)
            edges = np.zeros((n, side, side), np.uint8)
            lp, cp, ep = (a.ctypes.data_as(C.c_void_p) for a in (lines, counts, edges))
        self._chk(lib().ck_find_intersections(self._h, p, n, side, sp, mtx.ctypes.data_as(C.c_void_p), rects.ctypes.data_as(C.c_void_p),
                                              grid.ctypes.data_as(C.c_void_p), lp, cp, ep))
        if not want_lines:
            return grid[0] if single else grid
        found = [{divmod(z, 19): [tuple(int(v) for v in lines[f, z, k]) for k in range(counts[f, z])]
                  for z in np.nonzero(counts[f])[0]} for f in range(n)]
        return (grid[0], found[0], edges[0]) if single else (grid, found, edges)


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


def update_grid(lines, box, slot):
    """update_grid (stone/stonesfinder.py:888-947), host only: lines as cv2.HoughLinesP returns them ((k, 1, 4) or (k, 4)),
    box = getrect of the zone, slot = the int16 pair of the intersection, updated in place"""
    seg = np.ascontiguousarray(np.asarray(lines).reshape(-1, 4), np.int32)
    b = np.ascontiguousarray(box, np.int32)
    out = np.ascontiguousarray(slot, np.int16).copy()
    rc = lib().ck_update_grid(seg.ctypes.data_as(C.c_void_p), len(seg), b.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ZeroDivisionError("float division by zero") if rc == 1 else CkError("ck_update_grid: error %d" % rc)
    slot[:] = out


def ordered_hull(points):
    """imgutil.get_ordered_hull: convex hull of integer (x, y) points, clockwise on screen, nearest to the origin
    first -> list of (x, y) tuples (host only)"""
    pts = np.ascontiguousarray(np.asarray(points).reshape(-1, 2), np.int32)
    out, n = np.zeros((max(1, len(pts)), 2), np.int32), C.c_int32(0)
    rc = lib().ck_ordered_hull(pts.ctypes.data_as(C.c_void_p), len(pts), out.ctypes.data_as(C.c_void_p), C.byref(n))
    if rc != 0:
        raise CkError("ck_ordered_hull: error %d" % rc)
    return [(int(x), int(y)) for x, y in out[:n.value]]


class BoardFoldCore:
    """ck_boardfold: the ordered half of the automatic board finder (host only, no GPU needed)."""

    def __init__(self):
        self._h = C.c_void_p()
        if lib().ck_boardfold_create(C.byref(self._h)) != 0:
            raise CkError("ck_boardfold_create failed")
        self._out = np.zeros(16, np.int32)          # found, update, n_centers, pad, centers[8], stats[2]
        self._io32, self._io64 = np.zeros(2, np.int32), np.zeros(3, np.int64)      # ck_boardfold_run: (k, hold), (counter, seen, looked)

    def __del__(self):
        try:
            if self._h:
                lib().ck_boardfold_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def reset(self):
        lib().ck_boardfold_reset(self._h)

    def step(self, h, w, status, lines, counter, cur_hull):
        """-> (found, update, centers [(x, y), ...], stats (clusters, intersections) or None)"""
        lines = np.ascontiguousarray(lines, np.float32).reshape(-1, 2)
        hull = None if cur_hull is None else np.ascontiguousarray(cur_hull, np.int32).reshape(8)
        o = self._out
        base = o.ctypes.data
        rc = lib().ck_boardfold_step(self._h, int(h), int(w), int(status), lines.ctypes.data_as(C.c_void_p), len(lines),
                                     int(counter), None if hull is None else hull.ctypes.data_as(C.c_void_p),
                                     C.c_void_p(base), C.c_void_p(base + 4), C.c_void_p(base + 16), C.c_void_p(base + 8),
                                     C.c_void_p(base + 48))
        if rc == 4:
            raise IndexError("corner hull has fewer than 4 vertices")      # what the reference raises (bf_auto.py:206)
        if rc != 0:
            raise CkError("ck_boardfold_step: error %d" % rc)
        cen = [(int(o[4 + 2 * i]), int(o[5 + 2 * i])) for i in range(int(o[2]))]
        return bool(o[0]), bool(o[1]), cen, (None if o[12] < 0 else (int(o[12]), int(o[13])))


    def run(self, h, w, recs, k, counter, hold, seen, looked, cur_hull, order=None, hold_after_same_hit=-1):
        """ck_boardfold_run over the REC_DTYPE array `recs` (frame f at recs[order[f]], or recs[f]) from frame k ->
        (k, counter, hold, seen, looked, found, update, centers, stats): stops after the first frame that changes the corners
        (with hold_after_same_hit < 0: after every hit), or at the end of the batch"""
        assert recs.dtype == REC_DTYPE and recs.flags.c_contiguous
        n = len(recs)
        if order is not None:
            assert order.dtype == np.int32 and order.flags.c_contiguous
            n = len(order)
        hull = None if cur_hull is None else np.ascontiguousarray(cur_hull, np.int32).reshape(8)
        io32, io64 = self._io32, self._io64
        io32[:] = (k, hold)
        io64[:] = (counter, seen, looked)
        o = self._out
        base, b32, b64 = o.ctypes.data, io32.ctypes.data, io64.ctypes.data
        rc = lib().ck_boardfold_run(self._h, int(h), int(w), recs.ctypes.data_as(C.c_void_p),
                                    None if order is None else order.ctypes.data_as(C.c_void_p), n, C.c_void_p(b32),
                                    C.c_void_p(b64), C.c_void_p(b32 + 4), C.c_void_p(b64 + 8),
                                    None if hull is None else hull.ctypes.data_as(C.c_void_p), int(hold_after_same_hit),
                                    C.c_void_p(base), C.c_void_p(base + 4), C.c_void_p(base + 16), C.c_void_p(base + 8),
                                    C.c_void_p(base + 48))
        state = (int(io32[0]), int(io64[0]), int(io32[1]), int(io64[1]), int(io64[2]))
        if rc == 4:
            err = IndexError("corner hull has fewer than 4 vertices")      # what the reference raises (bf_auto.py:206)
            err.fold_state = state
            raise err
        if rc != 0:
            raise CkError("ck_boardfold_run: error %d" % rc)
        cen = [(int(o[4 + 2 * i]), int(o[5 + 2 * i])) for i in range(int(o[2]))]
        return state + (bool(o[0]), bool(o[1]), cen, (None if o[12] < 0 else (int(o[12]), int(o[13]))))


class PolicyCore:
    """ck_policy: SfNeural's emission policy over ordered runs of frames (host only, no GPU needed)."""
    SUGGEST, BULK = 1, 2

    def __init__(self, bg_init_frames=50):
        self._h = C.c_void_p()
        if lib().ck_policy_create(int(bg_init_frames), C.byref(self._h)) != 0:
            raise CkError("ck_policy_create failed")
        self._io = np.zeros(4, np.int32)            # frame, phase, kind, n_moves
        self._moves = np.zeros((2 * 361, 3), np.int32)

    def __del__(self):
        try:
            if self._h:
                lib().ck_policy_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def run(self, first_counter, region_label, region_conf, fgcount, board_of, apply, records=None, order=None):
        """Ordered run over n frames.  `board_of()` -> uint8 (19,19) goban as the controller holds it now;
        `apply(kind, [(color, r, c), ...], frame_index)` executes a request; if it raises, the run stops there and
        the rest of that frame is never done -- what an exception out of SfNeural._find means in the reference.
        `records`: a C-contiguous REC_DTYPE array to read the classifier's answers from (region_label / region_conf are
        then ignored: ck_policy_run_records, no copy of the two strided fields), frame f at records[order[f]] when `order`
        (int32) is given"""
        if records is not None:
            assert records.dtype == REC_DTYPE and records.flags.c_contiguous
            assert order is None or (order.dtype == np.int32 and order.flags.c_contiguous)
            n = len(records) if order is None else len(order)
        else:
            rl = np.ascontiguousarray(region_label, np.uint8).reshape(-1, 100)
            rc = np.ascontiguousarray(region_conf, np.float64).reshape(-1, 100)
            n = len(rl)
            assert len(rc) == n
        fg = None
        if fgcount is not None:
            fg = np.ascontiguousarray(fgcount, np.int32).reshape(n, 361)
        io, mv = self._io, self._moves
        io[:] = 0
        b = io.ctypes.data
        while True:
            board = np.ascontiguousarray(board_of(), np.uint8).reshape(361)
            tail = (None if fg is None else fg.ctypes.data_as(C.c_void_p), board.ctypes.data_as(C.c_void_p), C.c_void_p(b),
                    C.c_void_p(b + 4), C.c_void_p(b + 8), mv.ctypes.data_as(C.c_void_p), len(mv), C.c_void_p(b + 12))
            if records is not None:
                rcode = lib().ck_policy_run_records(self._h, n, int(first_counter), records.ctypes.data_as(C.c_void_p),
                                                    None if order is None else order.ctypes.data_as(C.c_void_p), *tail)
            else:
                rcode = lib().ck_policy_run(self._h, n, int(first_counter), rl.ctypes.data_as(C.c_void_p),
                                            rc.ctypes.data_as(C.c_void_p), *tail)
            if rcode != 0:
                raise CkError("ck_policy_run: error %d" % rcode)
            if io[2] == 0:
                return
            # a request made in the first half of frame k leaves (k, 1); a lookback request leaves (k + 1, 0)
            frame = int(io[0]) if io[1] == 1 else int(io[0]) - 1
            apply(int(io[2]), [(int(c), int(r), int(k)) for c, r, k in mv[:io[3]]], frame)

    def state(self):
        t, hc = np.zeros((19, 19), np.uint8), np.zeros((19, 19), np.uint8)
        he, cf, fl = np.zeros((19, 19), np.int32), np.zeros((19, 19), np.float64), np.zeros(2, np.int32)
        lib().ck_policy_get_state(self._h, *(a.ctypes.data_as(C.c_void_p) for a in (t, hc, he, cf, fl)))
        return dict(targets=t, heat_color=hc, heat_energy=he, heat_conf=cf, has_sampled=bool(fl[0]), recolour_seen=int(fl[1]))

    def set_targets(self, targets):
        t = np.ascontiguousarray(targets, np.uint8).reshape(19, 19)
        lib().ck_policy_set_state(self._h, t.ctypes.data_as(C.c_void_p), -1)

    def set_sampled(self, flag=True):
        lib().ck_policy_set_state(self._h, None, int(bool(flag)))

    def watch(self, r, c, color, confidence=0.0, stamp=0):
        """start (or with color 0: drop) the watch on a prediction, as a fresh HeatPoint would"""
        if lib().ck_policy_watch(self._h, int(r), int(c), int(color), float(confidence), int(stamp)) != 0:
            raise CkError("ck_policy_watch: bad arguments")


_default_ctx = {}


def get_context(device=0):
    """Process-wide default context per device (finders create their own)."""
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
