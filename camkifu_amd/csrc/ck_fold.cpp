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
// round(x, 10): CPython rounds the exact binary value to 10 decimals (correctly, ties to even) and converts
// back; glibc's printf does the same exact decimal conversion.
inline double py_round10(double x)
{
    if (!(x == x) || x == HUGE_VAL || x == -HUGE_VAL) return x;
    char buf[64];
    std::snprintf(buf, sizeof buf, "%.10f", x);
    return std::strtod(buf, nullptr);
}

inline long long py_int(double x) { return (long long)x; }      // int(): truncation toward zero

struct Pt { long long x, y; };

struct Seg {
    long long x0, y0, x1, y1;
    double len;          // sqrt((x0-x1)^2 + (y0-y1)^2)
    double theta;        // acos((x1 - x0) / len)
};

inline Seg make_seg(long long x0, long long y0, long long x1, long long y1)
{
    Seg s{ x0, y0, x1, y1, 0., 0. };
    const long long dx = x0 - x1, dy = y0 - y1;
    s.len = std::sqrt((double)(dx * dx + dy * dy));
    s.theta = std::acos((double)(x1 - x0) / s.len);
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
    const double ax = (double)(s.x1 - s.x0) / s.len, ay = (double)(s.y1 - s.y0) / s.len;
    const double bx = (double)(o.x1 - o.x0) / o.len, by = (double)(o.y1 - o.y0) / o.len;
    const double t = std::acos(py_round10(ax * bx + ay * by));
    return t <= PI / 2 ? t : PI - t;
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
    long long pairs_tested = 0;
};

namespace {

// bf_auto.py:143-172.  Both loops run over the theta-sorted lines, the inner one backwards until the pair gets
// "too parallel"; the proximity test looks at x twice and never at y (the reference's quirk), so a group accepts
// a point iff some member's x is close: each group keeps its x values sorted and the nearest one decides.
void group_intersections(ck_boardfold* bf, int h, int w)
{
    const double length_ref = (double)std::min(h, w);
    const double margin = -length_ref / 15;
    const double thresh = (length_ref / 80) * (length_ref / 80);       // ** 2 of a float: exact product
    std::vector<const Seg*> order(bf->lines.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = &bf->lines[i];
    std::stable_sort(order.begin(), order.end(), [](const Seg* a, const Seg* b) { return a->theta < b->theta; });
    std::vector<std::vector<long long>> xs(bf->groups.size());
    for (size_t g = 0; g < bf->groups.size(); g++) {
        for (const Pt& q : bf->groups[g]) xs[g].push_back(q.x);
        std::sort(xs[g].begin(), xs[g].end());
    }
    const double x_lo = 0 + margin, x_hi = (double)w - margin, y_lo = 0 + margin, y_hi = (double)h - margin;
    for (const Seg* s1 : order) {
        for (size_t k = order.size(); k-- > 0;) {
            const Seg* s2 = order[k];
            bf->pairs_tested++;
            if (!(PI / 3 < line_angle(*s1, *s2))) break;
            Pt p0;
            if (!intersect(*s1, *s2, &p0)) continue;       // cannot happen for lines more than pi/3 apart
            if (!(x_lo < (double)p0.x && (double)p0.x < x_hi && y_lo < (double)p0.y && (double)p0.y < y_hi)) continue;
            bool placed = false;
            for (size_t g = 0; g < bf->groups.size() && !placed; g++) {
                std::vector<long long>& v = xs[g];
                const auto it = std::lower_bound(v.begin(), v.end(), p0.x);
                bool near = false;
                if (it != v.end()) { const long long d = *it - p0.x; near = (double)(d * d + d * d) < thresh; }
                if (!near && it != v.begin()) { const long long d = p0.x - *(it - 1); near = (double)(d * d + d * d) < thresh; }
                if (near) { bf->groups[g].push_back(p0); v.insert(it, p0.x); placed = true; }
            }
            if (!placed) { bf->groups.push_back({ p0 }); xs.push_back({ p0.x }); }
        }
    }
}

// one merging pass, x-only distance again                                              core/imgutil.py:38-68
void connect_clusters(std::vector<std::vector<Pt>>& groups, double dist)
{
    const size_t n = groups.size();
    std::vector<char> gone(n, 0);
    for (size_t a = 0; a < n; a++) {
        long merge = -1;
        for (size_t ia = 0; ia < groups[a].size() && merge < 0; ia++) {
            const long long x = groups[a][ia].x;
            for (size_t b = 0; b < n && merge < 0; b++) {
                if (b == a || gone[b]) continue;
                for (const Pt& q : groups[b]) {
                    const long long d = x - q.x;
                    if ((double)(d * d + d * d) < dist) { merge = (long)b; break; }
                }
            }
        }
        if (merge >= 0) {
            std::vector<Pt> moved = groups[a];             // copy: the target may be the same storage after growth
            groups[(size_t)merge].insert(groups[(size_t)merge].end(), moved.begin(), moved.end());
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
};

namespace {

constexpr double MIN_CONFIDENCE = 0.6;
constexpr int TARGET_THRESH = 15, TARGET_INCR = 5, NB_LOOKBACK = 3;

inline int reg_start(int i) { return i < NREG - 1 ? 2 * i : G - 2; }             // nn_manager.py:92-126
inline int digit(int label, int k) { static const int p3[4] = { 1, 3, 9, 27 }; return (label / p3[k]) % 3; }
inline int cell_extent(int r) { return r == G - 1 ? 19 : 20; }                    // stonesfinder.py:412-450

inline void heat_set(ck_policy* p, int r, int c, int color, double conf, long long stamp)
{
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
    // mark                                                                             sf_neural.py:72-83
    for (int r = 0; r < G; r++) for (int c = 0; c < G; c++)
        if (!p->hp_color[r][c] && agitated(fgc, r, c, 0.7)) p->targets[r][c] = (uint8_t)(p->targets[r][c] + TARGET_INCR);
    for (int r = 0; r < G; r++) for (int c = 0; c < G; c++) if (p->targets[r][c]) p->targets[r][c]--;
    // select + predict                                                                 sf_neural.py:101-154
    struct Mv { int color, r, c; double conf; };
    std::vector<Mv> mv;
    for (int i = 0; i < NREG; i++) for (int j = 0; j < NREG; j++) {
        const int rs = reg_start(i), cs = reg_start(j);
        bool hot = false, busy = false;
        for (int k = 0; k < 4; k++) {
            hot |= p->targets[rs + k / 2][cs + k % 2] > TARGET_THRESH;
            busy |= agitated(fgc, rs + k / 2, cs + k % 2, 0.5);
        }
        if (!hot || busy) continue;
        for (int k = 0; k < 4; k++) p->targets[rs + k / 2][cs + k % 2] = 0;
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
        if (p->hp_color[r][c] != board[r * G + c]) { p->hp_color[r][c] = 0; continue; }     // changed by somebody else
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
        if (p->hp_energy[r][c] < -5) { p->hp_color[r][c] = 0; continue; }           // cold: forgotten  sf_neural.py:182-184
        if (p->hp_energy[r][c] <= 0) p->hp_energy[r][c]--;                         // drawn once per frame: __repr__ ages it
    }
}

}  // namespace

extern "C" {

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
        for (; k < n; k++, ph = 0) {
            const long long f = first_counter + k;
            const uint8_t* rl = region_label + (size_t)k * 100;
            const double* rc = region_conf + (size_t)k * 100;
            const int32_t* fgc = fgcount ? fgcount + (size_t)k * (G * G) : nullptr;
            if (ph == 0) {                                   // sf_neural.py:37-55, first half of the frame
                Req q;
                p->pending_sampled = false;
                if (f == 0 || f < p->bg_init_frames) { /* net loading frame / background sampling */ }
                else if (!p->has_sampled) { phase_assess(p, f, rl, rc, &q); p->pending_sampled = true; }
                else phase_targets(p, f, rl, rc, fgc, board, &q);
                if (!q.m.empty()) { hand_over(q, q.kind, k, 1); return CK_OK; }
            }
            // second half: the caller has applied the request, `board` is the goban as it is now
            if (p->pending_sampled) { p->has_sampled = true; p->pending_sampled = false; continue; }
            if (f == 0 || f < p->bg_init_frames || !p->has_sampled) continue;
            Req d;
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
    if (targets) std::memcpy(p->targets, targets, sizeof p->targets);
    if (has_sampled >= 0) p->has_sampled = has_sampled != 0;
    return CK_OK;
}

int ck_policy_watch(ck_policy* p, int r, int c, int color, double confidence, long long stamp)
{
    if (!p || r < 0 || r >= G || c < 0 || c >= G || color < 0 || color > 2) return CK_ERR_ARG;
    if (color == 0) p->hp_color[r][c] = 0;
    else heat_set(p, r, c, color, confidence, stamp);
    return CK_OK;
}

}  // extern "C"
