"""SfContours: only the image-processing front end of the reference's contour-based stones finder is built so far
 -- `get_canny`, stone/sf_contours.py:332-340:

    median = cv2.medianBlur(cv2.medianBlur(img, 13), 7)
    otsu, _ = cv2.threshold(cv2.cvtColor(median, cv2.COLOR_BGR2GRAY), 12, 255, cv2.THRESH_OTSU)
    return cv2.Canny(median, otsu / 2, otsu)

Here it is ONE library call (`ck_goban_canny`): the two medians run on the matrix-core kernel of K1 with other
windows, the grey histogram is an LDS-atomic kernel, the Otsu level is the library's double-precision scan on the
256 bins, Canny is K2's kernel with that frame's thresholds.  `find_stones` (contour hulls, zone colours:
sf_contours.py:48-230) is NOT built: it stays on the reference's own code path."""
import numpy as np

from .. import capi


class SfContours:
    """static front end only; not registered in cvconf.sfinders (it is not a complete finder)"""

    _ctx = None

    @staticmethod
    def get_canny(img, ctx=None):
        """Smooth image using median blur, then call Canny with Otsu thresholds (same name, argument and result as
        the reference's static method: an (h, w, 3) uint8 BGR image -> (h, w) uint8 edge map of 0 / 255)."""
        if ctx is None:
            if SfContours._ctx is None:
                SfContours._ctx = capi.Context(0)
            ctx = SfContours._ctx
        return ctx.goban_canny(np.ascontiguousarray(img, np.uint8)[None])[0]

    def find_stones(self, *a, **kw):
        raise NotImplementedError("SfContours.find_stones is not part of the MI355X path yet (SURVEY 8f rank 3)")
