"""The two geometry helpers the host-side Python still needs (everything else the reference keeps in
core/imgutil.py for the board path -- Segment, segment_from_hough, connect_clusters -- lives in the C++
fold, camkifu_amd/csrc/ck_fold.cpp, and in the oracle's restatement)."""
from numpy import hypot

from .. import capi


def norm(p, q):
    """euclidean distance between two points"""
    return float(hypot(float(p[0]) - float(q[0]), float(p[1]) - float(q[1])))


def get_ordered_hull(pts):
    """convex hull of integer points, clockwise on screen, first vertex = nearest to the upper-left corner
    (same name and result as the reference's helper; computed by ck_ordered_hull)"""
    return capi.ordered_hull(pts)
