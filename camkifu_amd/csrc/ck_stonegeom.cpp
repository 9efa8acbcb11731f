// ck_stonegeom.cpp -- host-side geometry of SfContours.find_stones (reference: src/camkifu/stone/sf_contours.py:48-330).
// The dense pixel work of that method (medians, Canny, contour labelling, the hull mask and the zone sums) runs on the
// GPU (k_stonefind.hip); what is left is per-contour work on a few dozen points, ordered and branchy, which the
// reference also does on the CPU through cv2:
//   * the raster of a filled convex hull        cv2.drawContours(img, [hull], 0, c, thickness=-1)      :85, 291
//   * the 5x5 chamfer distance of a small box   cv2.distanceTransform(negative, DIST_L2, DIST_MASK_5)  :240
//   * _find_centers                              :302-330
//   * find_color                                 :128-184 (raster order matters: it reads the stones found so far)
// Integer arithmetic throughout, except the few double comparisons the Python code makes, kept in its operation order.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "ck_stonegeom.h"

// ---- filled polygon ----------------------------------------------------------------------------------------------
// The library draws every side with its 8-connected line iterator (always walked left to right), then fills the
// scanlines ymin .. ymax-1 between pairs of active edges held in 16.16 fixed point: an edge starts at its upper vertex
// and moves by dx = ((x1 - x0) << 16) / (y1 - y0) (C division) per scanline, a span covers ceil(left) .. floor(right).
// Both are written here in closed form (the offset after i steps, the edge position on scanline y) instead of the
// library's running sums; the result is a bitmap of the polygon's bounding box.
void ck_raster_polygon(const int32_t* v, int nv, int* bx, int* by, int* bw, int* bh, std::vector<uint8_t>& bits)
{
    int x0 = v[0], x1 = v[0], y0 = v[1], y1 = v[1];
    for (int i = 1; i < nv; i++) {
        x0 = std::min(x0, v[2 * i]); x1 = std::max(x1, v[2 * i]);
        y0 = std::min(y0, v[2 * i + 1]); y1 = std::max(y1, v[2 * i + 1]);
    }
    const int w = x1 - x0 + 1, h = y1 - y0 + 1;
    *bx = x0; *by = y0; *bw = w; *bh = h;
    bits.assign((size_t)w * h, 0);
    struct Edge { int ya, yb; long long x, dx; };
    std::vector<Edge> edges;
    for (int i = 0; i < nv; i++) {
        int ax = v[2 * i] - x0, ay = v[2 * i + 1] - y0;
        int cx = v[2 * ((i + 1) % nv)] - x0, cy = v[2 * ((i + 1) % nv) + 1] - y0;
        if (ay != cy) {
            const long long num = ((long long)cx - ax) * 65536, den = cy - ay;
            const long long q = (num < 0 ? -num : num) / (den < 0 ? -den : den);
            const long long dx = ((num < 0) == (den < 0)) ? q : -q;
            if (ay < cy) edges.push_back({ ay, cy, (long long)ax * 65536, dx });
            else edges.push_back({ cy, ay, (long long)cx * 65536, dx });
        }
        // outline: start from the left end; the major axis advances every step, the minor one has moved
        // ceil((2 * minor * i - major) / (2 * major)) pixels after i steps
        if (cx < ax) { std::swap(ax, cx); std::swap(ay, cy); }
        const int ddx = cx - ax, ddy = cy - ay, ady = ddy < 0 ? -ddy : ddy, sy = ddy < 0 ? -1 : 1;
        const bool steep = ady > ddx;
        const int major = steep ? ady : ddx, minor = steep ? ddx : ady;
        for (int i = 0; i <= major; i++) {
            const int m = major ? (2 * minor * i - major + 2 * major - 1) / (2 * major) : 0;
            const int px = steep ? ax + m : ax + i, py = steep ? ay + sy * i : ay + sy * m;
            bits[(size_t)py * w + px] = 1;
        }
    }
    if (edges.size() < 2) return;
    std::vector<long long> xs;
    for (int y = 0; y < h - 1; y++) {                  // the last scanline belongs to the outline alone
        xs.clear();
        for (const Edge& e : edges)
            if (e.ya <= y && y < e.yb) xs.push_back(e.x + (long long)(y - e.ya) * e.dx);
        std::sort(xs.begin(), xs.end());
        for (size_t k = 0; k + 1 < xs.size(); k += 2) {
            long long a = (xs[k] + 65535) >> 16, b = xs[k + 1] >> 16;
            if (a < 0) a = 0;
            if (b > w - 1) b = w - 1;
            for (long long x = a; x <= b; x++) bits[(size_t)y * w + x] = 1;
        }
    }
}

// ---- 5x5 chamfer distance --------------------------------------------------------------------------------------
// Two raster passes with the weights (1, 1.4, 2.1969) in 16.16 fixed point (65536, 91750, 143976), zero pixels are
// the sources, a two-pixel frame of "infinity" around the image.  Result in fixed point (the library's float is this
// integer * 2^-16, exactly; only the position of maxima is used downstream).
void ck_chamfer5(const uint8_t* img, int h, int w, std::vector<int32_t>& dist)
{
    const int HV = 65536, DIAG = 91750, LONGD = 143976, INIT = 0x7fffffff >> 2;
    const int W = w + 4;
    std::vector<int32_t> t((size_t)(h + 4) * W, INIT);
    auto at = [&](int i, int j) -> int32_t& { return t[(size_t)(i + 2) * W + j + 2]; };
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            if (!img[(size_t)i * w + j]) { at(i, j) = 0; continue; }
            int d = at(i - 2, j - 1) + LONGD;
            d = std::min(d, at(i - 2, j + 1) + LONGD);
            d = std::min(d, at(i - 1, j - 2) + LONGD);
            d = std::min(d, at(i - 1, j - 1) + DIAG);
            d = std::min(d, at(i - 1, j) + HV);
            d = std::min(d, at(i - 1, j + 1) + DIAG);
            d = std::min(d, at(i - 1, j + 2) + LONGD);
            d = std::min(d, at(i, j - 1) + HV);
            at(i, j) = d;
        }
    dist.resize((size_t)h * w);
    for (int i = h - 1; i >= 0; i--)
        for (int j = w - 1; j >= 0; j--) {
            int d = at(i, j);
            if (d > HV) {
                d = std::min(d, at(i + 2, j + 1) + LONGD);
                d = std::min(d, at(i + 2, j - 1) + LONGD);
                d = std::min(d, at(i + 1, j + 2) + LONGD);
                d = std::min(d, at(i + 1, j + 1) + DIAG);
                d = std::min(d, at(i + 1, j) + HV);
                d = std::min(d, at(i + 1, j - 1) + DIAG);
                d = std::min(d, at(i + 1, j - 2) + LONGD);
                d = std::min(d, at(i, j + 1) + HV);
                at(i, j) = d;
            }
            dist[(size_t)i * w + j] = d;
        }
}

// ---- _find_centers: is any cell's farthest point near the cell centre? ------------------------------------------
// returns 1 / 0, or -1 where the reference would divide by zero (a box thinner than one stone radius)
int ck_has_stone_center(const int32_t* dist, int dx /* rows */, int dy /* cols */, double radius)
{
    const int nb_rows = (int)std::nearbyint((double)dx / 2 / radius);      // Python's round(): ties to even
    const int nb_cols = (int)std::nearbyint((double)dy / 2 / radius);
    if (nb_rows <= 0 || nb_cols <= 0) return -1;
    const int row_width = (int)((double)dx / nb_rows), col_width = (int)((double)dy / nb_cols);
    for (int row = 0; row < nb_rows; row++) {
        const int rs = row * row_width, re = (row + 1) * row_width;
        for (int col = 0; col < nb_cols; col++) {
            const int cs = col * col_width, ce = (col + 1) * col_width;
            int best = -1, mx = 0, my = 0;
            for (int i = rs; i < std::min(re + 1, dx); i++)
                for (int j = cs; j < std::min(ce + 1, dy); j++)
                    if (dist[(size_t)i * dy + j] > best) { best = dist[(size_t)i * dy + j]; mx = j - cs; my = i - rs; }
            if (best < 0) continue;                                        // empty slice: nothing to yield
            const double ex = mx - (double)(ce - cs) / 2, ey = my - (double)(re - rs) / 2;
            if ((double)std::min(row_width, col_width) / 3 < std::sqrt(ex * ex + ey * ey)) continue;
            return 1;
        }
    }
    return 0;
}

// ---- find_color over a block of zones, raster order -----------------------------------------------------------
// zones: R x C x 4 int16 (flag, B, G, R means); stones: R x C view (row stride `stride`) of 0 E / 1 B / 2 W
static void find_color(int r, int c, const int16_t* zones, int R, int C, uint8_t* stones, int stride)
{
    enum { cE = 0, cB = 1, cW = 2 };
    bool seen[3] = { false, false, false };
    int added = 0;
    const int16_t* me = zones + ((size_t)r * C + c) * 4;
    for (int i = -1; i < 2; i++) {
        if (0 <= r + i && r + i < R) {
            for (int j = -1; j < 2; j++) {
                if (i == 0 && j == 0) continue;
                if (0 <= c + j && c + j < C) {
                    const int16_t* ng = zones + ((size_t)(r + i) * C + c + j) * 4;
                    int sum = 0, sumabs = 0;
                    for (int k = 1; k < 4; k++) { const int d = me[k] - ng[k]; sum += d; sumabs += d < 0 ? -d : d; }
                    const int diff = sum < 0 ? -sumabs : sumabs;
                    if (!ng[0]) {
                        if (100 < sumabs) { seen[diff < 0 ? cB : cW] = true; added += 1; }
                        else if (sumabs < 70) { seen[cE] = true; added = 3; }
                    } else {
                        const int min_val = std::min(me[1] + me[2] + me[3], ng[1] + ng[2] + ng[3]);
                        if (i < 1 && j < 1) {
                            const int ns = stones[(size_t)(r + i) * stride + c + j];
                            if (ns != cB && ns != cW) continue;
                            if ((double)sumabs < min_val * 0.1) { seen[ns] = true; added += 1; }
                            else if (min_val < sumabs) { seen[ns == cW ? cB : cW] = true; added += 1; }
                        }
                    }
                    if (added == 3) break;
                }
            }
        }
        if (added == 3) {
            const int ncol = (int)seen[0] + (int)seen[1] + (int)seen[2];
            if (ncol == 1) stones[(size_t)r * stride + c] = seen[cB] ? cB : (seen[cW] ? cW : cE);
            break;
        }
    }
}

void ck_find_colors(const int16_t* zones, int R, int C, uint8_t* stones, int stride)
{
    for (int r = 0; r < R; r++)
        for (int c = 0; c < C; c++)
            if (zones[((size_t)r * C + c) * 4]) find_color(r, c, zones, R, C, stones, stride);
}

// ---- update_grid (stonesfinder.py:888-947) ----------------------------------------------------------------------
// lines: k x (x0, y0, x1, y1) as cv2.HoughLinesP returns them for the zone `box` = (x0, y0, x1, y1) of
// StonesFinder.getrect (x along rows); slot: the int16 (x, y) of the intersection, updated in place.  Python float
// arithmetic restated in its operation order (true divisions in double, int() truncation, math.acos / cos / sin).
void ck_update_grid_host(const int32_t* lines, int k, const int32_t* box, int16_t* slot)
{
    const double margin = (double)std::min(box[2] - box[0], box[3] - box[1]) / 7;
    auto inside = [&](double px, double py) {
        return box[0] + margin < px && px < box[2] - margin && box[1] + margin < py && py < box[3] - margin;
    };
    const int32_t* kept_buf[64];
    struct { const int32_t** p; int n; bool empty() const { return n == 0; } size_t size() const { return (size_t)n; }
             void push_back(const int32_t* c) { if (n < 64) p[n++] = c; } const int32_t* operator[](size_t i) const { return p[i]; } } kept = { kept_buf, 0 };
    for (int i = 0; i < k; i++) {
        const int32_t* c = lines + 4 * (size_t)i;
        const long long ddx = (long long)c[0] - c[2], ddy = (long long)c[1] - c[3];
        const double theta = std::acos((double)(c[2] - c[0]) / std::sqrt((double)(ddx * ddx + ddy * ddy)));
        double px, py;
        if (0.995 < std::fabs(std::cos(theta))) {            // level line: does it pass the middle columns?
            px = (double)(box[0] + box[2]) / 2;
            py = (double)(c[0] + c[2]) / 2 + box[1];
        } else if (0.995 < std::fabs(std::sin(theta))) {     // upright line: the middle rows?
            px = (double)(c[1] + c[3]) / 2 + box[0];
            py = (double)(box[3] + box[1]) / 2;
        } else continue;
        if (inside(px, py)) kept.push_back(c);
    }
    if (kept.empty()) return;
    slot[0] = (int16_t)-slot[0];
    slot[1] = (int16_t)-slot[1];
    if (!(1 < kept.size() && kept.size() < 5)) return;
    long long sx = 0, sy = 0, hits = 0;
    for (size_t a = 0; a < kept.size(); a++)
        for (size_t b = 0; b < kept.size(); b++) {
            if (a == b) continue;
            const int32_t *s = kept[a], *o = kept[b];
            const long long d1x = s[2] - s[0], d1y = s[3] - s[1], d2x = o[2] - o[0], d2y = o[3] - o[1];
            const double cross = (double)(d1x * d2y - d1y * d2x);
            if (std::fabs(cross) < 2.220446049250313e-16) continue;          // sys.float_info.epsilon
            const double t1 = (double)((long long)(o[0] - s[0]) * d2y - (long long)(o[1] - s[1]) * d2x) / cross;
            const int ix = (int)(s[0] + t1 * (double)d1x), iy = (int)(s[1] + t1 * (double)d1y);
            const long long qx = (long long)iy + box[0], qy = (long long)ix + box[1];    // crossing comes as (column, row)
            if (inside((double)qx, (double)qy)) { sx += qx; sy += qy; hits++; }
        }
    if (hits) {
        slot[0] = (int16_t)(int)((double)-sx / (double)hits);
        slot[1] = (int16_t)(int)((double)-sy / (double)hits);
    }
}
