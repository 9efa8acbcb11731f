"""SfContours: the image-processing front end of the reference's contour-based stones finder --
`get_canny`, stone/sf_contours.py:332-340:

    median = cv2.medianBlur(cv2.medianBlur(img, 13), 7)
    otsu, _ = cv2.threshold(cv2.cvtColor(median, cv2.COLOR_BGR2GRAY), 12, 255, cv2.THRESH_OTSU)
    return cv2.Canny(median, otsu / 2, otsu)

Here it is ONE library call (`ck_goban_canny`): the two medians run on the matrix-core kernel of K1 with other
windows, the grey histogram is an LDS-atomic kernel, the Otsu level is the library's double-precision scan on the
256 bins, Canny is K2's kernel with that frame's thresholds.  `find_stones` (contour hulls filled into a mask, zone
colours, the foreground's distance transform: sf_contours.py:48-300) is NOT built -- its pixel-exact pieces
(fillConvexPoly's raster rule, distanceTransform's 5x5 chamfer, minAreaRect's angle convention) need a machine with
OpenCV to be pinned; it stays on the reference's own code path."""
from numpy import ascontiguousarray, uint8

from .. import capi

_shared_ctx = []


def get_canny(img, ctx=None):
    """Smooth with two median blurs, then Canny with Otsu thresholds: an (h, w, 3) uint8 BGR image -> (h, w) uint8 edge
    map of 0 / 255 (same argument and result as the reference's static method)"""
    if ctx is None:
        if not _shared_ctx:
            _shared_ctx.append(capi.Context(0))
        ctx = _shared_ctx[0]
    return ctx.goban_canny(ascontiguousarray(img, uint8)[None])[0]


class SfContours:
    """the front end only; not registered in cvconf.sfinders (it is not a complete finder)"""
    get_canny = staticmethod(get_canny)

    def find_stones(self, *a, **kw):
        raise NotImplementedError("SfContours.find_stones is not part of the MI355X path (see the module docstring)")
