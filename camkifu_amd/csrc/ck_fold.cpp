// ck_fold.cpp -- the two ORDERED (stateful) parts of the hot path, host only, one implementation for the
// per-frame finders and for the batch pipeline's fold over gathered records:
//
//   ck_boardfold_*  what BoardFinderAuto does with a frame's Hough lines: 4-frame accumulation, pairwise
//                   intersections of non-parallel lines, greedy grouping, cluster merging, the 4 corners
//                   (reference: src/camkifu/board/bf_auto.py:76-102, 143-217; core/imgutil.py:38-68, 216-288,
//                   464-530).
//   ck_policy_*     what SfNeural does with a frame's classifier outputs: the one-off full assessment, agitation
//                   targets, calm-region re-prediction with the colour-ratio veto, the three-check lookback
//                   (reference: src/camkifu/stone/sf_neural.py:37-244; stone/nn_cache.py:16-41;
//                   stone/nn_manager.py:92-126, 246-254).
//
// Everything the reference computes in Python floats is computed here in the same IEEE double operations in the
// same order (the library is built with -ffp-contract=off; libm's acos/cos/sin/log/sqrt are the ones CPython
// calls), and Python's int() / round(x, 10) are restated exactly, so the results are the reference's bit for bit.
// State is flat arrays: targets uint8[19][19], the watched predictions ("heat points") as a struct of arrays.
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include <vector>

#include "../../include/camkifu_amd.h"

namespace {

constexpr int G = 19;            // goban size
constexpr int NREG = 10;         // 10 x 10 regions of 2 x 2 intersections (the last one overlaps: rows 17, 18)
constexpr double PI = 3.141592653589793;

// ---------------------------------------------------------------------------------------------------------
// small exact restatements of Python numerics
// ---------------------------------------------------------------------------------------------------------
// round(x, 10): CPython rounds the exact binary value to 10 decimals (correctly, ties to even) and converts the decimal
// string back to the nearest double.  Here for |x| < 2^19 (every caller passes a cosine) in integers: x = m * 2^e exactly,
// so N = round_half_even(m * 10^10 * 2^e) is the 10-decimal value times 10^10 (m * 10^10 < 2^87 fits 128 bits), and the
// IEEE quotient N / 1e10 -- both operands exact doubles -- is the double nearest to N / 10^10, i.e. what strtod returns for
// that decimal string.  Anything larger goes the long way: glibc's printf does the same exact decimal conversion.
inline double py_round10_slow(double x)
{
    char buf[400];
    std::snprintf(buf, sizeof buf, "%.10f", x);
    return std::strtod(buf, nullptr);
}

inline double py_round10(double x)
{
    if (!(x == x) || x == HUGE_VAL || x == -HUGE_VAL) return x;
    const double ax = std::fabs(x);
    if (!(ax < 524288.0)) return py_round10_slow(x);      // N = ax * 10^10 must stay below 2^53
    if (ax == 0.0) return x;
    int e2;
    const double fr = std::frexp(ax, &e2);                        // ax = fr * 2^e2, fr in [0.5, 1)
    const unsigned long long m = (unsigned long long)std::ldexp(fr, 53);   // 53-bit integer mantissa (subnormals: fewer bits, still exact)
    const int e = e2 - 53;                                        // ax = m * 2^e
    unsigned __int128 prod = (unsigned __int128)m * 10000000000ULL;
    unsigned long long N;
    if (e >= 0) N = (unsigned long long)(prod << e);              // (ax < 2^19: e <= -34, never taken; kept for completeness)
    else {
        const int sh = -e;
        if (sh >= 128) N = 0;
        else {
            const unsigned __int128 q = prod >> sh, rem = prod - (q << sh), half = (unsigned __int128)1 << (sh - 1);
            N = (unsigned long long)q;
            if (rem > half || (rem == half && (N & 1))) N++;
        }
    }
    const double r = (double)N / 1e10;
    return x < 0 ? -r : r;
}

inline long long py_int(double x) { return (long long)x; }      // int(): truncation toward zero

struct Pt { long long x, y; };

struct Seg {
    long long x0, y0, x1, y1;
    double len;          // sqrt((x0-x1)^2 + (y0-y1)^2)
    double theta;        // acos((x1 - x0) / len)
    double ux, uy;       // (x1 - x0) / len, (y1 - y0) / len: the unit vector line_angle divides out for every pair
};

inline Seg make_seg(long long x0, long long y0, long long x1, long long y1)
{
    Seg s{ x0, y0, x1, y1, 0., 0., 0., 0. };
    const long long dx = x0 - x1, dy = y0 - y1;
    s.len = std::sqrt((double)(dx * dx + dy * dy));
    s.ux = (double)(x1 - x0) / s.len;
    s.uy = (double)(y1 - y0) / s.len;
    s.theta = std::acos(s.ux);
    return s;
}

// (rho, theta) of cv2.HoughLines -> two far points on that line, int() truncated      core/imgutil.py:216-233
inline Seg seg_from_hough(float rho_f, float theta_f, int h, int w)
{
    const double rho = rho_f, theta = theta_f;
    const double a = std::cos(theta), b = std::sin(theta);
    const double x0 = a * rho, y0 = b * rho;
    const double extent = (double)std::max(h, w);
    const double mb = -b;
    return make_seg(py_int(x0 + extent * mb), py_int(y0 + extent * a), py_int(x0 - extent * mb), py_int(y0 - extent * a));
}

// smallest angle between the supporting lines, [0, pi/2]                               core/imgutil.py:504-513
inline double line_angle(const Seg& s, const Seg& o)
{
    const double t = std::acos(py_round10(s.ux * o.ux + s.uy * o.uy));
    return t <= PI / 2 ? t : PI - t;
}

// PI / 3 < line_angle(s, o), the test group_intersections makes for every pair, decided without the acos wherever the
// cosine is not within 1e-10 of +-0.5: the angle's argument is a multiple of 1e-10 (round(dot, 10)), so away from that
// one value acos is further than 1e-10 from PI / 3 resp. 2 PI / 3 -- a million times libm's error -- and the comparison
// can only come out one way.  At the boundary (and for NaN: a zero-length segment) the reference's expression is evaluated.
inline bool wider_than_60_degrees(const Seg& s, const Seg& o)
{
    const double a = std::fabs(s.ux * o.ux + s.uy * o.uy);
    if (a < 0.4999999998) return true;
    if (a > 0.5000000002) return false;
    return PI / 3 < line_angle(s, o);
}

// intersection of the infinite lines, int() truncated; false when parallel            core/imgutil.py:515-530
inline bool intersect(const Seg& s, const Seg& o, Pt* out)
{
    const long long qx = o.x0 - s.x0, qy = o.y0 - s.y0;
    const long long d1x = s.x1 - s.x0, d1y = s.y1 - s.y0, d2x = o.x1 - o.x0, d2y = o.y1 - o.y0;
    const double cross = (double)(d1x * d2y - d1y * d2x);
    if (std::fabs(cross) < DBL_EPSILON) return false;
    const double t1 = (double)(qx * d2y - qy * d2x) / cross;
    out->x = py_int((double)s.x0 + t1 * (double)d1x);
    out->y = py_int((double)s.y0 + t1 * (double)d1y);
    return true;
}

inline double pt_norm(const Pt& a, const Pt& b)
{
    const long long dx = a.x - b.x, dy = a.y - b.y;
    return std::sqrt((double)(dx * dx + dy * dy));
}

// convex hull, clockwise on screen (what cv2.convexHull defaults to), collinear points dropped, then rotated so
// that the vertex closest to the image origin comes first                               core/imgutil.py:236-288
std::vector<Pt> ordered_hull(std::vector<Pt> p)
{
    std::sort(p.begin(), p.end(), [](const Pt& a, const Pt& b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
    p.erase(std::unique(p.begin(), p.end(), [](const Pt& a, const Pt& b) { return a.x == b.x && a.y == b.y; }), p.end());
    std::vector<Pt> hull;
    if (p.size() <= 2) hull = p;
    else {
        auto cross = [](const Pt& o, const Pt& a, const Pt& b) { return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x); };
        std::vector<Pt> lo, up;
        for (const Pt& q : p) {                                    // small-y side, left to right
            while (lo.size() >= 2 && cross(lo[lo.size() - 2], lo.back(), q) <= 0) lo.pop_back();
            lo.push_back(q);
        }
        for (size_t i = p.size(); i-- > 0;) {                      // large-y side, right to left
            const Pt& q = p[i];
            while (up.size() >= 2 && cross(up[up.size() - 2], up.back(), q) <= 0) up.pop_back();
            up.push_back(q);
        }
        lo.pop_back(); up.pop_back();
        hull = lo;
        hull.insert(hull.end(), up.begin(), up.end());
    }
    size_t first = 0;
    long long best = -1;
    for (size_t i = 0; i < hull.size(); i++) {
        const long long d = hull[i].x * hull[i].x + hull[i].y * hull[i].y;
        if (best < 0 || d < best) { best = d; first = i; }
    }
    std::rotate(hull.begin(), hull.begin() + (long)first, hull.end());
    return hull;
}

}  // namespace

// =========================================================================================================
// board: ordered part of BoardFinderAuto._detect
// =========================================================================================================
struct ck_boardfold {
    std::vector<Seg> lines;                      // lines_accu
    std::vector<std::vector<Pt>> groups;         // groups_accu
    std::vector<int32_t> first_at_x;             // scratch of group_intersections
    long long pairs_tested = 0;
};

namespace {

// bf_auto.py:143-172.  Both loops run over the theta-sorted lines, the inner one backwards until the pair gets
// "too parallel"; the proximity test looks at x twice and never at y (the reference's quirk), so a group accepts
// a point iff some member's x is within dmax of the point's: `first[x]` holds the lowest group index with a member at
// that x, and the group a point joins -- the FIRST group, in creation order, with a close member -- is the minimum of
// `first` over the 2 dmax + 1 columns around it.  Points outside the grown image never get here, so x is bounded.
void group_intersections(ck_boardfold* bf, int h, int w)
{
    const double length_ref = (double)std::min(h, w);
    const double margin = -length_ref / 15;
    const double thresh = (length_ref / 80) * (length_ref / 80);       // ** 2 of a float: exact product
    std::vector<const Seg*> order(bf->lines.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = &bf->lines[i];
    std::stable_sort(order.begin(), order.end(), [](const Seg* a, const Seg* b) { return a->theta < b->theta; });
    const double x_lo = 0 + margin, x_hi = (double)w - margin, y_lo = 0 + margin, y_hi = (double)h - margin;
    long long dmax = -1;                                               // largest |dx| that passes (double)(dx^2 + dx^2) < thresh
    while ((double)((dmax + 1) * (dmax + 1) + (dmax + 1) * (dmax + 1)) < thresh) dmax++;
    const long long x_min = (long long)std::floor(x_lo) - dmax - 1, x_max = (long long)std::ceil(x_hi) + dmax + 1;
    constexpr int32_t NONE = INT32_MAX;
    std::vector<int32_t>& first = bf->first_at_x;
    first.assign((size_t)(x_max - x_min + 1), NONE);
    for (size_t g = 0; g < bf->groups.size(); g++)                     // (empty here: every grouping round ends with a clear)
        for (const Pt& q : bf->groups[g])
            if (q.x >= x_min && q.x <= x_max) first[(size_t)(q.x - x_min)] = std::min(first[(size_t)(q.x - x_min)], (int32_t)g);
    for (const Seg* s1 : order) {
        for (size_t k = order.size(); k-- > 0;) {
            const Seg* s2 = order[k];
            bf->pairs_tested++;
            if (!wider_than_60_degrees(*s1, *s2)) break;
            Pt p0;
            if (!intersect(*s1, *s2, &p0)) continue;       // cannot happen for lines more than pi/3 apart
            if (!(x_lo < (double)p0.x && (double)p0.x < x_hi && y_lo < (double)p0.y && (double)p0.y < y_hi)) continue;
            int32_t g = NONE;
            if (dmax >= 0) {
                const int32_t* col = &first[(size_t)(p0.x - dmax - x_min)];
                for (long long d = 0; d <= 2 * dmax; d++) g = std::min(g, col[d]);
            }
            if (g == NONE) { g = (int32_t)bf->groups.size(); bf->groups.push_back({ p0 }); }
            else bf->groups[(size_t)g].push_back(p0);
            int32_t& slot = first[(size_t)(p0.x - x_min)];
            slot = std::min(slot, g);
        }
    }
}

// one merging pass, x-only distance again                                              core/imgutil.py:38-68
// The reference compares every member of a group with every member of every other group until one is close; the test
// depends on |dx| only, so per group the NEAREST x decides: each group keeps its x values sorted (as above) and a member
// costs a binary search per group instead of a scan of all points -- same first (member, group) hit, same merges.
void connect_clusters(std::vector<std::vector<Pt>>& groups, double dist)
{
    const size_t n = groups.size();
    std::vector<char> gone(n, 0);
    std::vector<std::vector<long long>> xs(n);
    for (size_t g = 0; g < n; g++) {
        xs[g].reserve(groups[g].size());
        for (const Pt& q : groups[g]) xs[g].push_back(q.x);
        std::sort(xs[g].begin(), xs[g].end());
    }
    auto near = [dist](const std::vector<long long>& v, long long x) {
        const auto it = std::lower_bound(v.begin(), v.end(), x);
        if (it != v.end()) { const long long d = *it - x; if ((double)(d * d + d * d) < dist) return true; }
        if (it != v.begin()) { const long long d = x - *(it - 1); if ((double)(d * d + d * d) < dist) return true; }
        return false;
    };
    std::vector<long long> merged;
    for (size_t a = 0; a < n; a++) {
        long merge = -1;
        for (size_t ia = 0; ia < groups[a].size() && merge < 0; ia++) {
            const long long x = groups[a][ia].x;
            for (size_t b = 0; b < n && merge < 0; b++)
                if (b != a && !gone[b] && near(xs[b], x)) merge = (long)b;
        }
        if (merge >= 0) {
            std::vector<Pt> moved = groups[a];             // copy: the target may be the same storage after growth
            groups[(size_t)merge].insert(groups[(size_t)merge].end(), moved.begin(), moved.end());
            merged.resize(xs[(size_t)merge].size() + xs[a].size());
            std::merge(xs[(size_t)merge].begin(), xs[(size_t)merge].end(), xs[a].begin(), xs[a].end(), merged.begin());
            xs[(size_t)merge].swap(merged);
            gone[a] = 1;
        }
    }
    size_t k = 0;
    for (size_t a = 0; a < n; a++) if (!gone[a]) { if (k != a) groups[k] = std::move(groups[a]); k++; }
    groups.resize(k);
}

}  // namespace

extern "C" {

int ck_ordered_hull(const int32_t* pts, int n, int32_t* out, int32_t* n_out)
{
    if (!pts || !out || !n_out || n < 0) return CK_ERR_ARG;
    try {
        std::vector<Pt> p((size_t)n);
        for (int i = 0; i < n; i++) p[(size_t)i] = { pts[2 * i], pts[2 * i + 1] };
        const std::vector<Pt> hull = ordered_hull(std::move(p));
        *n_out = (int32_t)hull.size();
        for (size_t i = 0; i < hull.size(); i++) { out[2 * i] = (int32_t)hull[i].x; out[2 * i + 1] = (int32_t)hull[i].y; }
        return CK_OK;
    } catch (const std::bad_alloc&) {
        return CK_ERR_STATE;
    }
}

int ck_boardfold_create(ck_boardfold** out)
{
    if (!out) return CK_ERR_ARG;
    *out = new (std::nothrow) ck_boardfold();
    return *out ? CK_OK : CK_ERR_STATE;
}

void ck_boardfold_destroy(ck_boardfold* bf) { delete bf; }

int ck_boardfold_reset(ck_boardfold* bf)
{
    if (!bf) return CK_ERR_ARG;
    bf->lines.clear(); bf->groups.clear();
    return CK_OK;
}

int ck_boardfold_step(ck_boardfold* bf, int h, int w, int status, const float* lines, int n_lines,
                      long long frame_counter, const int32_t* cur_hull, int32_t* found, int32_t* update,
                      int32_t* centers, int32_t* n_centers, int32_t* stats)
{
    if (!bf || !found || !update || !centers || !n_centers || h <= 0 || w <= 0 || n_lines < 0 || (n_lines && !lines))
        return CK_ERR_ARG;
    *found = 0; *update = 0; *n_centers = 0;
    if (stats) { stats[0] = -1; stats[1] = -1; }
    if (status != CK_BOARD_LINES) return CK_OK;              // no contour / biggest contour too small: nothing accumulates
    try {
        for (int i = 0; i < n_lines; i++) bf->lines.push_back(seg_from_hough(lines[2 * i], lines[2 * i + 1], h, w));
        if (frame_counter % 4) return CK_OK;                 // 4 frames of lines before looking for corners
        const double length_ref = (double)std::min(h, w);
        group_intersections(bf, h, w);
        while (bf->groups.size() > 4) {
            const size_t before = bf->groups.size();
            connect_clusters(bf->groups, (length_ref / 50) * (length_ref / 50));
            if (bf->groups.size() == before) break;
        }
        // bf_auto.py:174-217
        int rc = CK_OK;
        if (bf->groups.size() == 4) {
            std::vector<Pt> cen;
            for (const auto& g : bf->groups) {
                long long sx = 0, sy = 0;
                for (const Pt& q : g) { sx += q.x; sy += q.y; }
                cen.push_back({ py_int((double)sx / (double)g.size()), py_int((double)sy / (double)g.size()) });
            }
            cen = ordered_hull(cen);
            bool ok = true;
            for (size_t i = 0; i < cen.size() && ok; i++)
                if (pt_norm(cen[(i + cen.size() - 1) % cen.size()], cen[i]) < length_ref / 3) ok = false;
            bool upd = cur_hull == nullptr;
            if (ok && !upd) {
                if (cen.size() < 4) rc = CK_ERR_STATE;       // the reference indexes 4 corners of a 3-vertex hull: IndexError
                else for (int i = 0; i < 4 && !upd; i++) {
                    const Pt cur{ cur_hull[2 * i], cur_hull[2 * i + 1] };
                    if (5 < pt_norm(cen[(size_t)i], cur)) upd = true;
                }
            }
            *found = ok; *update = upd;
            *n_centers = (int32_t)cen.size();
            for (size_t i = 0; i < cen.size(); i++) { centers[2 * i] = (int32_t)cen[i].x; centers[2 * i + 1] = (int32_t)cen[i].y; }
        }
        if (stats) {
            stats[0] = (int32_t)bf->groups.size();
            long long tot = 0;
            for (const auto& g : bf->groups) tot += (long long)g.size();
            stats[1] = (int32_t)tot;
        }
        bf->lines.clear();
        bf->groups.clear();
        return rc;
    } catch (const std::bad_alloc&) {
        return CK_ERR_STATE;
    }
}

int ck_boardfold_run(ck_boardfold* bf, int h, int w, const ck_frame_record* recs, const int32_t* order, int n, int32_t* k_io,
                     long long* counter_io, int32_t* hold_io, long long* seen_looked_io, const int32_t* cur_hull,
                     int hold_after_same_hit, int32_t* found, int32_t* update, int32_t* centers, int32_t* n_centers,
                     int32_t* stats)
{
    if (!bf || !k_io || !counter_io || !hold_io || !seen_looked_io || !found || !update || !centers || !n_centers ||
        n < 0 || (n && !recs) || *k_io < 0 || *hold_io < 0)
        return CK_ERR_ARG;
    *found = 0; *update = 0; *n_centers = 0;
    int k = *k_io;
    while (k < n) {
        if (*hold_io > 0) {                                  // nothing is looked at during the hold-off      bf_auto.py:43-49
            const int skip = std::min(*hold_io, n - k);
            *hold_io -= skip; seen_looked_io[0] += skip; *counter_io += skip; k += skip;
            continue;
        }
        const ck_frame_record& r = recs[order ? order[k] : k];
        const int kept = std::max(0, std::min(r.n_lines, (int32_t)CK_REC_LMAX));
        seen_looked_io[0]++; seen_looked_io[1]++;
        *k_io = k;                                           // (where an error leaves the fold: this frame is not counted)
        const int rc = ck_boardfold_step(bf, h, w, r.status, &r.lines[0][0], kept, *counter_io, cur_hull, found, update,
                                         centers, n_centers, stats);
        if (rc != CK_OK) return rc;
        (*counter_io)++; k++;
        if (*update) break;                                  // the caller's corners change: it comes back with the new hull
        if (*found) {
            // a hit that leaves the corners where they are: the transform is the one the caller already holds (same hull,
            // same K7), so only the hold-off starts -- unless the caller wants to see every hit
            if (hold_after_same_hit < 0) break;
            *hold_io = hold_after_same_hit;
        }
    }
    *k_io = k;
    return CK_OK;
}

}  // extern "C"

// =========================================================================================================
// stones: SfNeural's emission policy
// =========================================================================================================
struct ck_policy {
    int bg_init_frames = 50;
    bool has_sampled = false, pending_sampled = false;
    uint8_t targets[G][G] = {};
    // heat points, struct of arrays; color 0 = no point
    uint8_t hp_color[G][G] = {};
    double hp_conf[G][G] = {};
    long long hp_stamp[G][G] = {};
    int hp_energy[G][G] = {}, hp_goal[G][G] = {}, hp_checks[G][G] = {}, hp_passed[G][G] = {};
    long long recolour_seen = 0;     // "now seeing X instead of Y" events (the reference prints them)
    // what lets a quiet frame leave after ONE pass over its 361 counts: how many targets are non-zero, how many watched
    // predictions exist (both recounted wherever the arrays change)
    int live_targets = 0, live_points = 0;
    // agitation thresholds per intersection: the smallest count v with area * ratio < v in the reference's double
    // arithmetic (sf_neural.py:178-180), for ratio 0.7 (mark) and 0.5 (select); min_thr5 = the smallest of the latter
    int32_t thr7[G][G], thr5[G][G], min_thr5;
    ck_policy();
};

namespace {

constexpr double MIN_CONFIDENCE = 0.6;
constexpr int TARGET_THRESH = 15, TARGET_INCR = 5, NB_LOOKBACK = 3;

inline int reg_start(int i) { return i < NREG - 1 ? 2 * i : G - 2; }             // nn_manager.py:92-126
inline int digit(int label, int k) { static const int p3[4] = { 1, 3, 9, 27 }; return (label / p3[k]) % 3; }
inline int cell_extent(int r) { return r == G - 1 ? 19 : 20; }                    // stonesfinder.py:412-450

inline int32_t first_above(int area, double ratio)
{
    int32_t v = (int32_t)((double)area * ratio);
    while (v > 0 && (double)area * ratio < (double)(v - 1)) v--;
    while (!((double)area * ratio < (double)v)) v++;
    return v;
}

inline void heat_drop(ck_policy* p, int r, int c)
{
    if (p->hp_color[r][c]) p->live_points--;
    p->hp_color[r][c] = 0;
}

inline void heat_set(ck_policy* p, int r, int c, int color, double conf, long long stamp)
{
    if (!p->hp_color[r][c] && color) p->live_points++;
    p->hp_color[r][c] = (uint8_t)color; p->hp_conf[r][c] = conf; p->hp_stamp[r][c] = stamp;
    p->hp_energy[r][c] = NB_LOOKBACK; p->hp_goal[r][c] = NB_LOOKBACK; p->hp_checks[r][c] = 0; p->hp_passed[r][c] = 0;
}

inline bool agitated(const int32_t* fgc, int r, int c, double ratio)
{
    if (!fgc) return false;
    const int area = cell_extent(r) * cell_extent(c);
    return (double)area * ratio < (double)fgc[r * G + c];
}

struct Req { int kind = 0; std::vector<int32_t> m; };       // triples (color, r, c)

// no intersection of the frame reaches even the lower (select) threshold: nothing is agitated
inline bool quiet_frame(const ck_policy* p, const int32_t* fgc)
{
    if (!fgc) return true;
    int32_t top = 0;
    for (int i = 0; i < G * G; i++) top = std::max(top, fgc[i]);
    return top < p->min_thr5;
}

// |log3(#B / #W)|, both counts bumped when one is zero                                 sf_neural.py:186-195
inline double colour_ratio(int nb, int nw)
{
    if (nb == 0 || nw == 0) { nb++; nw++; }
    return std::fabs(std::log((double)nb / (double)nw) / std::log(3.0));
}

void phase_assess(ck_policy* p, long long f, const uint8_t* rl, const double* rc, Req* q)
{
    uint8_t grid[G][G]; double cf[G][G];
    for (int i = 0; i < NREG; i++) for (int j = 0; j < NREG; j++) {              // nn_cache.py:33-41: later regions overwrite
        const int rs = reg_start(i), cs = reg_start(j), lab = rl[i * NREG + j];
        for (int k = 0; k < 4; k++) { grid[rs + k / 2][cs + k % 2] = (uint8_t)digit(lab, k); cf[rs + k / 2][cs + k % 2] = rc[i * NREG + j]; }
    }
    q->kind = 2;
    for (int r = 0; r < G; r++) for (int c = 0; c < G; c++)
        if (grid[r][c] && cf[r][c] > MIN_CONFIDENCE) {
            q->m.insert(q->m.end(), { grid[r][c], r, c });
            heat_set(p, r, c, grid[r][c], cf[r][c], f);
        }
}

void phase_targets(ck_policy* p, long long f, const uint8_t* rl, const double* rc, const int32_t* fgc,
                   const uint8_t* board, Req* q)
{
    const bool quiet = quiet_frame(p, fgc);
    if (quiet && p->live_targets == 0) return;          // nothing to mark, nothing to decay, no region can be hot
    // mark, then decay                                                                 sf_neural.py:72-83
    // (one branch-free pass over the 361 cells as flat arrays: the compiler vectorises it)
    uint8_t hot[G][G], busy[G][G];
    int live = 0, any_hot = 0;
    {
        uint8_t* __restrict__ tg = &p->targets[0][0];
        const uint8_t* __restrict__ hp = &p->hp_color[0][0];
        const int32_t* __restrict__ t7 = &p->thr7[0][0];
        const int32_t* __restrict__ t5 = &p->thr5[0][0];
        uint8_t* __restrict__ ht = &hot[0][0];
        uint8_t* __restrict__ bs = &busy[0][0];
        for (int i = 0; i < G * G; i++) {
            const int32_t v = quiet ? 0 : fgc[i];
            const uint8_t marked = (uint8_t)((hp[i] == 0) & (v >= t7[i]));
            uint8_t t = (uint8_t)(tg[i] + (uint8_t)(marked * TARGET_INCR));
            t = (uint8_t)(t - (uint8_t)(t != 0));
            tg[i] = t;
            live += t != 0;
            const uint8_t h = (uint8_t)(t > TARGET_THRESH);
            ht[i] = h;
            any_hot |= h;
            bs[i] = (uint8_t)(v >= t5[i]);
        }
    }
    p->live_targets = live;
    if (!any_hot) return;                               // no region can be selected
    // select + predict                                                                 sf_neural.py:101-154
    struct Mv { int color, r, c; double conf; };
    std::vector<Mv> mv;
    for (int i = 0; i < NREG; i++) for (int j = 0; j < NREG; j++) {
        const int rs = reg_start(i), cs = reg_start(j);
        if (!(hot[rs][cs] | hot[rs][cs + 1] | hot[rs + 1][cs] | hot[rs + 1][cs + 1])) continue;
        if (busy[rs][cs] | busy[rs][cs + 1] | busy[rs + 1][cs] | busy[rs + 1][cs + 1]) continue;
        for (int k = 0; k < 4; k++) {
            uint8_t& t = p->targets[rs + k / 2][cs + k % 2];
            p->live_targets -= t != 0;
            t = 0;
            hot[rs + k / 2][cs + k % 2] = 0;            // (an overlapping region of the last row / column sees the zero, as the reference does)
        }
        const double conf = rc[i * NREG + j];
        if (conf < MIN_CONFIDENCE) continue;
        for (int k = 0; k < 4; k++) {
            const int col = digit(rl[i * NREG + j], k), r = rs + k / 2, c = cs + k % 2;
            if (!col) continue;
            const int prev = board[r * G + c];
            if (prev == 0) {
                bool dup = false;                       // the reference collects into a set
                for (const Mv& m : mv) dup |= m.color == col && m.r == r && m.c == c && m.conf == conf;
                if (!dup) mv.push_back({ col, r, c, conf });
            } else if (prev != col) p->recolour_seen++;
        }
    }
    if (mv.empty()) return;
    int nb = 0, nw = 0;
    for (const Mv& m : mv) (m.color == 1 ? nb : nw)++;
    if (!(colour_ratio(nb, nw) < 1)) return;            // colour-ratio veto                sf_neural.py:89
    for (const Mv& m : mv) heat_set(p, m.r, m.c, m.color, m.conf, f);
    q->kind = mv.size() == 1 ? 1 : 2;
    for (const Mv& m : mv) q->m.insert(q->m.end(), { m.color, m.r, m.c });
}

void phase_lookback(ck_policy* p, long long f, const uint8_t* rl, const double* rc, const uint8_t* board, Req* q)
{
    for (int r = 0; r < G; r++) for (int c = 0; c < G; c++) {                     // sf_neural.py:156-176
        if (!p->hp_color[r][c] || !(0 < p->hp_energy[r][c])) continue;
        if (p->hp_color[r][c] != board[r * G + c]) { heat_drop(p, r, c); continue; }         // changed by somebody else
        if (!(10 < f - p->hp_stamp[r][c])) continue;
        p->hp_stamp[r][c] = f;
        // NNCache.predict_stone (nn_cache.py:16-23): region (r // 2, c // 2), entry 2 * (r % 2) + c % 2 of its decode
        // -- on row / column 18 that is the region's FIRST row / column (17): the reference's indexing, kept
        const int i = r / 2, j = c / 2;
        const int seen = digit(rl[i * NREG + j], 2 * (r % 2) + c % 2);
        const double conf = rc[i * NREG + j];
        // HeatPoint.check                                                              sf_neural.py:209-217
        p->hp_checks[r][c]++; p->hp_energy[r][c]--;
        double add = 0;
        if (seen == p->hp_color[r][c]) { p->hp_passed[r][c]++; add = conf; }
        p->hp_conf[r][c] = (p->hp_conf[r][c] * p->hp_checks[r][c] + add) / (p->hp_checks[r][c] + 1);
        // HeatPoint.is_valid                                                           sf_neural.py:219-225
        if (!(2.0 * p->hp_goal[r][c] / 3 <= (double)(p->hp_passed[r][c] + p->hp_energy[r][c]))) {
            p->hp_energy[r][c] = 0; p->hp_conf[r][c] = 0.0;
            q->kind = 2;
            q->m.insert(q->m.end(), { 0, r, c });
        }
    }
}

void phase_age(ck_policy* p)
{
    for (int r = 0; r < G; r++) for (int c = 0; c < G; c++) {
        if (!p->hp_color[r][c]) continue;
        if (p->hp_energy[r][c] < -5) { heat_drop(p, r, c); continue; }              // cold: forgotten  sf_neural.py:182-184
        if (p->hp_energy[r][c] <= 0) p->hp_energy[r][c]--;                         // drawn once per frame: __repr__ ages it
    }
}

}  // namespace

ck_policy::ck_policy()
{
    min_thr5 = INT32_MAX;
    for (int r = 0; r < G; r++) for (int c = 0; c < G; c++) {
        const int area = cell_extent(r) * cell_extent(c);
        thr7[r][c] = first_above(area, 0.7);
        thr5[r][c] = first_above(area, 0.5);
        min_thr5 = std::min(min_thr5, thr5[r][c]);
    }
}

namespace {

// frames [*frame_io, n) of an ordered run; frame k's classifier answers are at rl + k * rl_stride bytes (100 labels) and
// rc + k * rc_stride bytes (100 doubles): contiguous arrays (ck_policy_run) or the stones halves of gathered records
int policy_run(ck_policy* p, int n, long long first_counter, const uint8_t* region_label, size_t rl_stride,
               const uint8_t* region_conf, size_t rc_stride, const int32_t* order, const int32_t* fgcount, const uint8_t* board,
               int32_t* frame_io, int32_t* phase_io, int32_t* kind, int32_t* moves, int cap, int32_t* n_moves)
{
    if (!p || n < 0 || !frame_io || !phase_io || !kind || !moves || !n_moves || cap < 2 * G * G || !board ||
        (n && (!region_label || !region_conf)) || *frame_io < 0 || *phase_io < 0 || *phase_io > 1)
        return CK_ERR_ARG;
    *kind = 0; *n_moves = 0;
    try {
        int k = *frame_io, ph = *phase_io;
        auto hand_over = [&](const Req& q, int kind_, int next_frame, int next_phase) {
            *kind = kind_; *n_moves = (int32_t)(q.m.size() / 3);
            std::memcpy(moves, q.m.data(), q.m.size() * sizeof(int32_t));
            *frame_io = next_frame; *phase_io = next_phase;
        };
        Req q, d;
        for (; k < n; k++, ph = 0) {
            const long long f = first_counter + k;
            const size_t row = order ? (size_t)order[k] : (size_t)k;
            const uint8_t* rl = region_label + row * rl_stride;
            const double* rc = (const double*)(region_conf + row * rc_stride);
            const int32_t* fgc = fgcount ? fgcount + (size_t)k * (G * G) : nullptr;
            if (ph == 0) {                                   // sf_neural.py:37-55, first half of the frame
                q.kind = 0; q.m.clear();
                p->pending_sampled = false;
                if (f == 0 || f < p->bg_init_frames) { /* net loading frame / background sampling */ }
                else if (!p->has_sampled) { phase_assess(p, f, rl, rc, &q); p->pending_sampled = true; }
                else phase_targets(p, f, rl, rc, fgc, board, &q);
                if (!q.m.empty()) { hand_over(q, q.kind, k, 1); return CK_OK; }
            }
            // second half: the caller has applied the request, `board` is the goban as it is now
            if (p->pending_sampled) { p->has_sampled = true; p->pending_sampled = false; continue; }
            if (f == 0 || f < p->bg_init_frames || !p->has_sampled) continue;
            if (p->live_points == 0) continue;               // no watched prediction: nothing to look back at, nothing ages
            d.kind = 0; d.m.clear();
            phase_lookback(p, f, rl, rc, board, &d);
            phase_age(p);
            if (!d.m.empty()) { hand_over(d, 2, k + 1, 0); return CK_OK; }
        }
        *frame_io = n; *phase_io = 0;
        return CK_OK;
    } catch (const std::bad_alloc&) {
        return CK_ERR_STATE;
    }
}

}  // namespace

extern "C" {

double ck_round10(double x) { return py_round10(x); }
double ck_round10_reference(double x) { return py_round10_slow(x); }

int ck_policy_create(int bg_init_frames, ck_policy** out)
{
    if (!out || bg_init_frames < 0) return CK_ERR_ARG;
    *out = new (std::nothrow) ck_policy();
    if (!*out) return CK_ERR_STATE;
    (*out)->bg_init_frames = bg_init_frames;
    return CK_OK;
}

void ck_policy_destroy(ck_policy* p) { delete p; }

int ck_policy_run(ck_policy* p, int n, long long first_counter, const uint8_t* region_label, const double* region_conf,
                  const int32_t* fgcount, const uint8_t* board, int32_t* frame_io, int32_t* phase_io,
                  int32_t* kind, int32_t* moves, int cap, int32_t* n_moves)
{
    return policy_run(p, n, first_counter, region_label, 100, (const uint8_t*)region_conf, 100 * sizeof(double), nullptr, fgcount, board,
                      frame_io, phase_io, kind, moves, cap, n_moves);
}

int ck_policy_run_records(ck_policy* p, int n, long long first_counter, const ck_frame_record* recs, const int32_t* order,
                          const int32_t* fgcount, const uint8_t* board, int32_t* frame_io, int32_t* phase_io,
                          int32_t* kind, int32_t* moves, int cap, int32_t* n_moves)
{
    if (n && !recs) return CK_ERR_ARG;
    return policy_run(p, n, first_counter, recs ? recs->region_label : nullptr, sizeof(ck_frame_record),
                      recs ? (const uint8_t*)recs->region_conf : nullptr, sizeof(ck_frame_record), order, fgcount, board,
                      frame_io, phase_io, kind, moves, cap, n_moves);
}

int ck_policy_get_state(const ck_policy* p, uint8_t* targets, uint8_t* heat_color, int32_t* heat_energy,
                        double* heat_conf, int32_t* flags)
{
    if (!p) return CK_ERR_ARG;
    if (targets) std::memcpy(targets, p->targets, sizeof p->targets);
    if (heat_color) std::memcpy(heat_color, p->hp_color, sizeof p->hp_color);
    if (heat_energy) std::memcpy(heat_energy, p->hp_energy, sizeof p->hp_energy);
    if (heat_conf) std::memcpy(heat_conf, p->hp_conf, sizeof p->hp_conf);
    if (flags) { flags[0] = p->has_sampled; flags[1] = (int32_t)std::min<long long>(p->recolour_seen, INT32_MAX); }
    return CK_OK;
}

int ck_policy_set_state(ck_policy* p, const uint8_t* targets, int has_sampled)
{
    if (!p) return CK_ERR_ARG;
    if (targets) {
        std::memcpy(p->targets, targets, sizeof p->targets);
        p->live_targets = 0;
        for (int r = 0; r < G; r++) for (int c = 0; c < G; c++) p->live_targets += p->targets[r][c] != 0;
    }
    if (has_sampled >= 0) p->has_sampled = has_sampled != 0;
    return CK_OK;
}

int ck_policy_watch(ck_policy* p, int r, int c, int color, double confidence, long long stamp)
{
    if (!p || r < 0 || r >= G || c < 0 || c >= G || color < 0 || color > 2) return CK_ERR_ARG;
    if (color == 0) heat_drop(p, r, c);
    else heat_set(p, r, c, color, confidence, stamp);
    return CK_OK;
}

}  // extern "C"
