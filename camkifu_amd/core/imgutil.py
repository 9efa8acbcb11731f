"""Geometry helpers of the board path -- the subset of the reference's core/imgutil.py that
BoardFinderAuto and GobanCorners use (reference: src/camkifu/core/imgutil.py:38-68, 216-357,
464-530).  Pure Python on purpose: these run once per frame on a handful of lines and carry
the reference's quirks (int() truncation, x-only distance tests, round(dot, 10) before acos),
which are what parity is defined by."""
import math
import sys


def norm(p1, p2):
    return math.sqrt((p1[0] - p2[0]) ** 2 + (p1[1] - p2[1]) ** 2)


def within_margin(p, box, margin):
    return box[0] + margin < p[0] < box[2] - margin and box[1] + margin < p[1] < box[3] - margin


class Segment:
    """Two integer end points (x0, y0, x1, y1); theta = angle with the horizontal in [0, pi]."""

    def __init__(self, coordinates):
        self.coords = coordinates
        self.theta = math.acos((coordinates[2] - coordinates[0]) / self.norm())
        self.offset = 0, 0

    def __getitem__(self, item):
        return self.coords[item]

    def p1(self):
        return self.coords[0], self.coords[1]

    def p2(self):
        return self.coords[2], self.coords[3]

    def norm(self):
        return math.sqrt((self.coords[0] - self.coords[2]) ** 2 + (self.coords[1] - self.coords[3]) ** 2)

    def line_angle(self, other):
        """smallest angle between the two supporting lines, in [0, pi/2]"""
        n0, n1 = self.norm(), other.norm()
        x0, y0 = (self.coords[2] - self.coords[0]) / n0, (self.coords[3] - self.coords[1]) / n0
        x1, y1 = (other.coords[2] - other.coords[0]) / n1, (other.coords[3] - other.coords[1]) / n1
        theta = math.acos(round(x0 * x1 + y0 * y1, 10))
        return theta if theta <= math.pi / 2 else math.pi - theta

    def intersection(self, other):
        """intersection of the two infinite lines, truncated to ints; None when parallel"""
        dx, dy = other[0] - self[0], other[1] - self[1]
        d1 = (self[2] - self[0], self[3] - self[1])
        d2 = (other[2] - other[0], other[3] - other[1])
        cross = float(d1[0] * d2[1] - d1[1] * d2[0])
        if abs(cross) < sys.float_info.epsilon:
            return None
        t1 = (dx * d2[1] - dy * d2[0]) / cross
        return int(self[0] + t1 * d1[0]), int(self[1] + t1 * d1[1])

    def __str__(self):
        return "Seg(%s + %s, %.2frad)" % (self.coords, self.offset, self.theta)


def segment_from_hough(hough_line, img_shape):
    """(rho, theta) as cv2.HoughLines returns it -> a long Segment lying on that line."""
    rho, theta = float(hough_line[0]), float(hough_line[1])
    a, b = math.cos(theta), math.sin(theta)
    x0, y0 = a * rho, b * rho
    extent = max(img_shape[0], img_shape[1])
    pt1 = int(x0 + extent * (-b)), int(y0 + extent * a)
    pt2 = int(x0 - extent * (-b)), int(y0 - extent * a)
    return Segment((pt1[0], pt1[1], pt2[0], pt2[1]))


def cyclic_permute(points):
    """rotate the sequence so that the point closest to the image origin comes first"""
    best, idx = sys.maxsize, 0
    for i, p in enumerate(points):
        d = p[0] ** 2 + p[1] ** 2
        if d < best:
            best, idx = d, i
    n = len(points)
    return [(points[i % n][0], points[i % n][1]) for i in range(idx, idx + n)]


def convex_hull(points):
    """Convex hull, vertices in the orientation cv2.convexHull defaults to (clockwise on screen,
    y pointing down), collinear points dropped.  Andrew's monotone chain on integer points."""
    pts = sorted(set((int(p[0]), int(p[1])) for p in points))
    if len(pts) <= 2:
        return pts

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])
    lower, upper = [], []
    for p in pts:                       # small-y side, left to right
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 0:
            lower.pop()
        lower.append(p)
    for p in reversed(pts):             # large-y side, right to left
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 0:
            upper.pop()
        upper.append(p)
    return lower[:-1] + upper[:-1]


def get_ordered_hull(points):
    """convex hull, clockwise on screen, first point = closest to the upper-left corner"""
    return cyclic_permute(convex_hull(points))


def connect_clusters(groups, dist):
    """one merging pass of connectivity clustering; the distance test looks at x only, twice --
    a quirk of the reference (imgutil.py:59) that parity keeps"""
    todel = []
    for g0 in groups:
        merge = None
        for p0 in g0:
            for g1 in groups:
                if g0 is not g1 and not any(g1 is d for d in todel):
                    for p1 in g1:
                        if (p0[0] - p1[0]) ** 2 + (p0[0] - p1[0]) ** 2 < dist:
                            merge = g1
                            break
                if merge:
                    break
            if merge:
                break
        if merge:
            merge.extend(g0)
            todel.append(g0)
    for gdel in todel:
        for i, g in enumerate(groups):
            if g is gdel:
                del groups[i]
                break
