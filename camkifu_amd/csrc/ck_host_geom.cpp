// ck_host_geom.cpp -- host-side float geometry of the board path (tiny, latency-bound work
// that the reference also does on the CPU through cv2):
//   K4  cv2.minAreaRect(contour)            (reference: src/camkifu/core/imgutil.py:423-428)
//   K7  cv2.getPerspectiveTransform         (reference: src/camkifu/board/boardfinder.py:43-45)
//       + the 3x3 inverse cv2.warpPerspective applies to M (stone/stonesfinder.py:140)
// Compiled with -ffp-contract=off: the float32 sequence of the rotating-calipers search is
// kept exactly as written so that areas (and therefore the choice of the three biggest
// contours) are reproducible.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/camkifu_amd.h"

void ck_min_area_rect_box(const int32_t* pts, int n, float* out_wha);

void ck_invert3x3(const double* s, double* d)
{
    double det = s[0] * (s[4] * s[8] - s[5] * s[7]) - s[1] * (s[3] * s[8] - s[5] * s[6]) +
                 s[2] * (s[3] * s[7] - s[4] * s[6]);
    if (det == 0) { std::memset(d, 0, 9 * sizeof(double)); return; }
    det = 1. / det;
    double t[9];
    t[0] = (s[4] * s[8] - s[5] * s[7]) * det;
    t[1] = (s[2] * s[7] - s[1] * s[8]) * det;
    t[2] = (s[1] * s[5] - s[2] * s[4]) * det;
    t[3] = (s[5] * s[6] - s[3] * s[8]) * det;
    t[4] = (s[0] * s[8] - s[2] * s[6]) * det;
    t[5] = (s[2] * s[3] - s[0] * s[5]) * det;
    t[6] = (s[3] * s[7] - s[4] * s[6]) * det;
    t[7] = (s[1] * s[6] - s[0] * s[7]) * det;
    t[8] = (s[0] * s[4] - s[1] * s[3]) * det;
    std::memcpy(d, t, sizeof t);
}

namespace {

struct P2 { int x, y; };

inline long long turn(const P2& o, const P2& a, const P2& b)
{
    return (long long)(a.x - o.x) * (b.y - o.y) - (long long)(a.y - o.y) * (b.x - o.x);
}

// convex hull in the vertex order cv::convexHull(clockwise=true) produces: start at the
// leftmost point, follow the large-y chain to the rightmost point, return on the small-y
// chain; collinear points dropped.
std::vector<P2> hull_ordered(const int32_t* pts, int n)
{
    std::vector<P2> p((size_t)n);
    for (int i = 0; i < n; i++) p[i] = { pts[2 * i], pts[2 * i + 1] };
    std::sort(p.begin(), p.end(), [](const P2& a, const P2& b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
    p.erase(std::unique(p.begin(), p.end(), [](const P2& a, const P2& b) { return a.x == b.x && a.y == b.y; }), p.end());
    const int m = (int)p.size();
    if (m <= 2) return p;
    std::vector<P2> st;
    st.reserve(2 * (size_t)m);
    for (int i = 0; i < m; i++) {
        while (st.size() >= 2 && turn(st[st.size() - 2], st.back(), p[i]) >= 0) st.pop_back();
        st.push_back(p[i]);
    }
    const size_t lower = st.size() + 1;
    for (int i = m - 2; i >= 0; i--) {
        while (st.size() >= lower && turn(st[st.size() - 2], st.back(), p[i]) >= 0) st.pop_back();
        st.push_back(p[i]);
    }
    st.pop_back();
    if (st.size() < 3) return { p.front(), p.back() };
    return st;
}

}  // namespace

std::vector<int32_t> ck_hull_points(const int32_t* pts, int n)
{
    std::vector<int32_t> out;
    if (n <= 0) return out;
    for (const P2& p : hull_ordered(pts, n)) { out.push_back(p.x); out.push_back(p.y); }
    return out;
}

void ck_min_area_rect(const int32_t* pts, int n, float* out_wh)
{
    float wha[3];
    ck_min_area_rect_box(pts, n, wha);
    out_wh[0] = wha[0]; out_wh[1] = wha[1];
}

// (w, h, angle in degrees) as the cv2 binding returns box[1], box[2]: the angle is the direction of the rectangle's first
// side vector (atan2 in double -> float, then float * 180 / pi in double -> float)
void ck_min_area_rect_box(const int32_t* pts, int n, float* out_wh)
{
    out_wh[0] = out_wh[1] = out_wh[2] = 0.f;
    if (n <= 0) return;
    const std::vector<P2> hull = hull_ordered(pts, n);
    const int hn = (int)hull.size();
    auto degrees = [](float a) { return (float)((double)(a * 180.f) / 3.1415926535897932384626433832795); };
    if (hn == 2) {
        const double dx = (double)((float)hull[1].x - (float)hull[0].x);
        const double dy = (double)((float)hull[1].y - (float)hull[0].y);
        out_wh[0] = (float)std::sqrt(dx * dx + dy * dy);
        out_wh[2] = degrees((float)std::atan2(dy, dx));
        return;
    }
    if (hn < 3) return;
    std::vector<float> px((size_t)hn), py((size_t)hn), vx((size_t)hn), vy((size_t)hn), inv_len((size_t)hn);
    for (int i = 0; i < hn; i++) { px[i] = (float)hull[i].x; py[i] = (float)hull[i].y; }

    int left = 0, bottom = 0, right = 0, top = 0;
    float left_x = px[0], right_x = px[0], top_y = py[0], bottom_y = py[0];
    float p0x = px[0], p0y = py[0];
    for (int i = 0; i < hn; i++) {
        if (p0x < left_x) { left_x = p0x; left = i; }
        if (p0x > right_x) { right_x = p0x; right = i; }
        if (p0y > top_y) { top_y = p0y; top = i; }
        if (p0y < bottom_y) { bottom_y = p0y; bottom = i; }
        const int j = (i + 1 < hn) ? i + 1 : 0;
        const double dx = (double)px[j] - (double)p0x;
        const double dy = (double)py[j] - (double)p0y;
        vx[i] = (float)dx; vy[i] = (float)dy;
        inv_len[i] = (float)(1. / std::sqrt(dx * dx + dy * dy));
        p0x = px[j]; p0y = py[j];
    }
    float orientation = 0.f;
    {
        double ax = vx[hn - 1], ay = vy[hn - 1];
        for (int i = 0; i < hn; i++) {
            const double bx = vx[i], by = vy[i];
            const double convexity = ax * by - ay * bx;
            if (convexity != 0) { orientation = convexity > 0 ? 1.f : -1.f; break; }
            ax = bx; ay = by;
        }
    }
    float base_a = orientation, base_b = 0.f;
    int seq[4] = { bottom, right, top, left };
    float minarea = FLT_MAX;
    float best_a = 0, best_b = 0, best_w = 0, best_h = 0;
    for (int k = 0; k < hn; k++) {
        const float dp[4] = {
            +base_a * vx[seq[0]] + base_b * vy[seq[0]],
            -base_b * vx[seq[1]] + base_a * vy[seq[1]],
            -base_a * vx[seq[2]] - base_b * vy[seq[2]],
            +base_b * vx[seq[3]] - base_a * vy[seq[3]],
        };
        float maxcos = dp[0] * inv_len[seq[0]];
        int main_element = 0;
        for (int i = 1; i < 4; i++) {
            const float cosalpha = dp[i] * inv_len[seq[i]];
            if (cosalpha > maxcos) { main_element = i; maxcos = cosalpha; }
        }
        const int pindex = seq[main_element];
        const float lead_x = vx[pindex] * inv_len[pindex];
        const float lead_y = vy[pindex] * inv_len[pindex];
        switch (main_element) {
        case 0: base_a = lead_x; base_b = lead_y; break;
        case 1: base_a = lead_y; base_b = -lead_x; break;
        case 2: base_a = -lead_x; base_b = -lead_y; break;
        default: base_a = -lead_y; base_b = lead_x; break;
        }
        seq[main_element] = seq[main_element] + 1 == hn ? 0 : seq[main_element] + 1;
        float dx = px[seq[1]] - px[seq[3]];
        float dy = py[seq[1]] - py[seq[3]];
        const float width = dx * base_a + dy * base_b;
        dx = px[seq[2]] - px[seq[0]];
        dy = py[seq[2]] - py[seq[0]];
        const float height = -dx * base_b + dy * base_a;
        const float area = width * height;
        if (area <= minarea) { minarea = area; best_a = base_a; best_b = base_b; best_w = width; best_h = height; }
    }
    const float v1x = best_a * best_w, v1y = best_b * best_w;
    const float v2x = -best_b * best_h, v2y = best_a * best_h;
    out_wh[0] = (float)std::sqrt((double)v1x * v1x + (double)v1y * v1y);
    out_wh[1] = (float)std::sqrt((double)v2x * v2x + (double)v2y * v2y);
    out_wh[2] = degrees((float)std::atan2((double)v1y, (double)v1x));
}

extern "C" int ck_get_perspective_transform(const float* src, const float* dst, double* M)
{
    if (!src || !dst || !M) return CK_ERR_ARG;
    double a[8][9];
    for (int i = 0; i < 4; i++) {
        const double sx = src[2 * i], sy = src[2 * i + 1], dx = dst[2 * i], dy = dst[2 * i + 1];
        const double r0[9] = { sx, sy, 1, 0, 0, 0, -sx * dx, -sy * dx, dx };
        const double r1[9] = { 0, 0, 0, sx, sy, 1, -sx * dy, -sy * dy, dy };
        std::memcpy(a[i], r0, sizeof r0);
        std::memcpy(a[i + 4], r1, sizeof r1);
    }
    for (int c = 0; c < 8; c++) {
        int piv = c;
        for (int r = c + 1; r < 8; r++) if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
        if (std::fabs(a[piv][c]) < 1e-300) return CK_ERR_ARG;
        if (piv != c) for (int k = 0; k < 9; k++) std::swap(a[c][k], a[piv][k]);
        for (int r = 0; r < 8; r++) {
            if (r == c) continue;
            const double f = a[r][c] / a[c][c];
            for (int k = c; k < 9; k++) a[r][k] -= f * a[c][k];
        }
    }
    for (int i = 0; i < 8; i++) M[i] = a[i][8] / a[i][i];
    M[8] = 1.;
    return CK_OK;
}
