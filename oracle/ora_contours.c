/*
 * ora_contours.c -- ORACLE (test infrastructure only; see ck_oracle.h).
 *
 * K3  cv2.findContours(canny, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)
 *     (/root/reference/src/camkifu/board/bf_auto.py:75)
 * K4  imgutil.sort_contours_box -> BoundingBox(cv2.minAreaRect) -> bisect.insort
 *     (/root/reference/src/camkifu/core/imgutil.py:291-315, 409-434)
 * K5  cv2.drawContours(ghost, contours, pos, 255, thickness=1) for the 3 biggest
 *     (/root/reference/src/camkifu/board/bf_auto.py:78-84, 125-129)
 *
 * The library algorithms restated: Suzuki & Abe (1985) border following as OpenCV 3.1
 * runs it for RETR_EXTERNAL (frame cleared, only outer borders traced, the "last
 * non-zero border on this row" acceptance test); cv::convexHull ordering
 * (clockwise=true: leftmost -> max-y chain -> rightmost -> min-y chain); rotating
 * calipers in float32.  "parity unpinned" (no reference fixture); pinned against the
 * independent set-based restatement below and brute force in tests/.
 */
#include "ck_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* direction codes: 0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE (y grows downwards) */
static const int DX8[8] = { 1, 1, 0, -1, -1, -1, 0, 1 };
static const int DY8[8] = { 0, -1, -1, -1, 0, 1, 1, 1 };

int ora_find_external_suzuki(const uint8_t* edges, int h, int w,
                             int max_contours, int* starts,
                             int* pix_off, int* vert_off,
                             int max_pix, int* pix_xy, int max_vert, int* vert_xy)
{
    int8_t* img = (int8_t*)malloc((size_t)h * w);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            img[(size_t)y * w + x] =
                (y == 0 || y == h - 1 || x == 0 || x == w - 1) ? 0 : (edges[(size_t)y * w + x] != 0);
    int delta[16];
    for (int k = 0; k < 16; k++) delta[k] = DY8[k & 7] * w + DX8[k & 7];

    const int8_t NBD = 2, NBD_NEG = (int8_t)(2 | -128);
    int ncont = 0, npix = 0, nvert = 0, overflow = 0;
    pix_off[0] = 0; vert_off[0] = 0;

    for (int y = 1; y < h - 1 && !overflow; y++) {
        int8_t* row = img + (size_t)y * w;
        int prev = row[0];
        int lnbd_x = 0;
        int x = 1;
        while (x < w) {
            int p = 0;
            for (; x < w && (p = row[x]) == prev; x++) ;
            if (x >= w) break;
            int is_hole = 0;
            if (!(prev == 0 && p == 1)) {
                if (p != 0 || prev < 1) goto resume_scan;
                if (prev & -2) lnbd_x = x - 1;
                is_hole = 1;
            }
            if (is_hole || row[lnbd_x] > 0) goto resume_scan;   /* RETR_EXTERNAL */
            {
                /* follow the outer border starting at (x, y) */
                if (ncont >= max_contours) { overflow = 1; break; }
                starts[ncont * 2] = x; starts[ncont * 2 + 1] = y;
                int8_t* i0 = row + x;
                int8_t *i1, *i3, *i4 = 0;
                int s, s_end, prev_s;
                int px = x, py = y;
                s_end = s = 4;
                do {
                    s = (s - 1) & 7;
                    i1 = i0 + delta[s];
                    if (*i1 != 0) break;
                } while (s != s_end);
                if (s == s_end) {                /* isolated pixel */
                    *i0 = NBD_NEG;
                    if (npix < max_pix) { pix_xy[npix * 2] = px; pix_xy[npix * 2 + 1] = py; npix++; } else overflow = 1;
                    if (nvert < max_vert) { vert_xy[nvert * 2] = px; vert_xy[nvert * 2 + 1] = py; nvert++; } else overflow = 1;
                } else {
                    i3 = i0;
                    prev_s = s ^ 4;
                    for (;;) {
                        s_end = s;
                        for (;;) {
                            i4 = i3 + delta[++s];
                            if (*i4 != 0) break;
                        }
                        s &= 7;
                        /* east neighbour examined and found 0 -> right-hand exit mark */
                        if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = NBD_NEG;
                        else if (*i3 == 1) *i3 = NBD;
                        if (npix < max_pix) { pix_xy[npix * 2] = px; pix_xy[npix * 2 + 1] = py; npix++; } else { overflow = 1; break; }
                        if (s != prev_s) {
                            if (nvert < max_vert) { vert_xy[nvert * 2] = px; vert_xy[nvert * 2 + 1] = py; nvert++; } else { overflow = 1; break; }
                            prev_s = s;
                        }
                        px += DX8[s]; py += DY8[s];
                        if (i4 == i0 && i3 == i1) break;
                        i3 = i4;
                        s = (s + 4) & 7;
                    }
                }
                ncont++;
                pix_off[ncont] = npix; vert_off[ncont] = nvert;
                p = row[x];
            }
        resume_scan:
            prev = p;
            if (prev & -2) lnbd_x = x;
            x++;   /* OpenCV re-enters its skip loop at the same x; p == prev there, so it advances */
        }
    }
    free(img);
    return overflow ? -1 : ncont;
}

int ora_find_external_sets(const uint8_t* edges, int h, int w, int32_t* labels_out,
                           int max_contours, int* starts)
{
    const size_t npx = (size_t)h * w;
    uint8_t* e = (uint8_t*)malloc(npx);
    uint8_t* s0 = (uint8_t*)calloc(npx, 1);
    int32_t* lab = (int32_t*)malloc(npx * sizeof(int32_t));
    size_t* q = (size_t*)malloc(npx * sizeof(size_t));
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            e[(size_t)y * w + x] =
                (y == 0 || y == h - 1 || x == 0 || x == w - 1) ? 0 : (edges[(size_t)y * w + x] != 0);
    /* S0: background 4-connected to the (cleared) frame */
    size_t qh = 0, qt = 0;
    s0[0] = 1; q[qt++] = 0;
    while (qh < qt) {
        size_t i = q[qh++];
        int y = (int)(i / w), x = (int)(i % w);
        const int ddx[4] = { 1, -1, 0, 0 }, ddy[4] = { 0, 0, 1, -1 };
        for (int k = 0; k < 4; k++) {
            int yy = y + ddy[k], xx = x + ddx[k];
            if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
            size_t j = (size_t)yy * w + xx;
            if (!e[j] && !s0[j]) { s0[j] = 1; q[qt++] = j; }
        }
    }
    /* 8-connected components in raster order of first pixel */
    for (size_t i = 0; i < npx; i++) lab[i] = -1;
    int ncomp = 0, ntop = 0;
    int* top_id = (int*)malloc(sizeof(int) * (npx / 2 + 2));
    for (size_t i0 = 0; i0 < npx; i0++) {
        if (!e[i0] || lab[i0] >= 0) continue;
        int id = ncomp++;
        int is_top = 0;
        qh = qt = 0; q[qt++] = i0; lab[i0] = id;
        while (qh < qt) {
            size_t i = q[qh++];
            int y = (int)(i / w), x = (int)(i % w);
            for (int k = 0; k < 8; k++) {
                int yy = y + DY8[k], xx = x + DX8[k];
                size_t j = (size_t)yy * w + xx;     /* frame is cleared, so in range */
                if (e[j]) { if (lab[j] < 0) { lab[j] = id; q[qt++] = j; } }
                else if ((k & 1) == 0 && s0[j]) is_top = 1;
            }
        }
        if (is_top) {
            if (ntop < max_contours) { starts[ntop * 2] = (int)(i0 % w); starts[ntop * 2 + 1] = (int)(i0 / w); }
            top_id[id] = ntop++;
        } else top_id[id] = -1;
    }
    for (size_t i = 0; i < npx; i++) {
        int32_t out = -1;
        if (e[i] && top_id[lab[i]] >= 0) {
            int y = (int)(i / w), x = (int)(i % w);
            if (s0[i + 1] || s0[i - 1] || s0[i + w] || s0[i - w]) out = top_id[lab[i]];
            (void)y; (void)x;
        }
        labels_out[i] = out;
    }
    free(top_id); free(q); free(lab); free(s0); free(e);
    return ntop;
}

/* ---- K4: convex hull in cv::convexHull(clockwise=true) order, then calipers ---------- */

typedef struct { int x, y; } ipt;
static int cmp_ipt(const void* a, const void* b)
{
    const ipt *p = (const ipt*)a, *q = (const ipt*)b;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    return (p->y > q->y) - (p->y < q->y);
}
static long long cross3(ipt o, ipt a, ipt b)
{
    return (long long)(a.x - o.x) * (b.y - o.y) - (long long)(a.y - o.y) * (b.x - o.x);
}

/* Returns hull vertex count; hull written in the order OpenCV emits for clockwise=true:
 * start at the leftmost point (smallest x, then smallest y), walk the chain on the
 * larger-y side to the rightmost point, come back on the smaller-y side.
 * Collinear and duplicate points are dropped. */
static int hull_cv_order(const int32_t* pts, int n, ipt* hull)
{
    ipt* p = (ipt*)malloc(sizeof(ipt) * (size_t)n);
    for (int i = 0; i < n; i++) { p[i].x = pts[2 * i]; p[i].y = pts[2 * i + 1]; }
    qsort(p, (size_t)n, sizeof(ipt), cmp_ipt);
    int m = 0;
    for (int i = 0; i < n; i++) if (m == 0 || p[i].x != p[m - 1].x || p[i].y != p[m - 1].y) p[m++] = p[i];
    if (m <= 2) { for (int i = 0; i < m; i++) hull[i] = p[i]; free(p); return m; }
    ipt* st = (ipt*)malloc(sizeof(ipt) * (size_t)(2 * m + 2));
    int k = 0;
    /* chain on the larger-y side, left to right: keep turns with cross < 0 when going
     * left->right with y up...  in image coordinates the larger-y chain turns so that
     * cross(o,a,b) < 0 is rejected for the "upper" (visually lower) boundary. */
    for (int i = 0; i < m; i++) {                      /* larger-y side */
        while (k >= 2 && cross3(st[k - 2], st[k - 1], p[i]) >= 0) k--;
        st[k++] = p[i];
    }
    int t = k + 1;
    for (int i = m - 2; i >= 0; i--) {                 /* smaller-y side, right to left */
        while (k >= t && cross3(st[k - 2], st[k - 1], p[i]) >= 0) k--;
        st[k++] = p[i];
    }
    k--;                                               /* last == first */
    if (k < 3) {                                       /* all collinear: two extremes */
        hull[0] = p[0]; hull[1] = p[m - 1];
        free(st); free(p); return 2;
    }
    for (int i = 0; i < k; i++) hull[i] = st[i];
    free(st); free(p);
    return k;
}

/* rotating calipers, minimum-area rectangle; float32 arithmetic throughout as in the
 * library routine; returns out[0..5] = corner, vec1, vec2 */
static void rotating_calipers_minarea(const float* px, const float* py, int n, float* out)
{
    float minarea = FLT_MAX;
    int left = 0, bottom = 0, right = 0, top = 0;
    int seq[4];
    float* inv_len = (float*)malloc(sizeof(float) * (size_t)n);
    float* vx = (float*)malloc(sizeof(float) * (size_t)n);
    float* vy = (float*)malloc(sizeof(float) * (size_t)n);
    float orientation = 0, base_a, base_b = 0;
    float left_x, right_x, top_y, bottom_y;
    float p0x = px[0], p0y = py[0];
    int   b_left = 0, b_bottom = 0;
    float b_a = 0, b_w = 0, b_b = 0, b_h = 0;

    left_x = right_x = p0x; top_y = bottom_y = p0y;
    for (int i = 0; i < n; i++) {
        if (p0x < left_x) { left_x = p0x; left = i; }
        if (p0x > right_x) { right_x = p0x; right = i; }
        if (p0y > top_y) { top_y = p0y; top = i; }
        if (p0y < bottom_y) { bottom_y = p0y; bottom = i; }
        int j = (i + 1 < n) ? i + 1 : 0;
        double dx = (double)px[j] - (double)p0x;
        double dy = (double)py[j] - (double)p0y;
        vx[i] = (float)dx; vy[i] = (float)dy;
        inv_len[i] = (float)(1. / sqrt(dx * dx + dy * dy));
        p0x = px[j]; p0y = py[j];
    }
    {
        double ax = vx[n - 1], ay = vy[n - 1];
        for (int i = 0; i < n; i++) {
            double bx = vx[i], by = vy[i];
            double convexity = ax * by - ay * bx;
            if (convexity != 0) { orientation = (convexity > 0) ? 1.f : -1.f; break; }
            ax = bx; ay = by;
        }
    }
    base_a = orientation;
    seq[0] = bottom; seq[1] = right; seq[2] = top; seq[3] = left;
    for (int k = 0; k < n; k++) {
        float dp[4] = {
            +base_a * vx[seq[0]] + base_b * vy[seq[0]],
            -base_b * vx[seq[1]] + base_a * vy[seq[1]],
            -base_a * vx[seq[2]] - base_b * vy[seq[2]],
            +base_b * vx[seq[3]] - base_a * vy[seq[3]],
        };
        float maxcos = dp[0] * inv_len[seq[0]];
        int main_element = 0;
        for (int i = 1; i < 4; i++) {
            float cosalpha = dp[i] * inv_len[seq[i]];
            if (cosalpha > maxcos) { main_element = i; maxcos = cosalpha; }
        }
        {
            int pindex = seq[main_element];
            float lead_x = vx[pindex] * inv_len[pindex];
            float lead_y = vy[pindex] * inv_len[pindex];
            switch (main_element) {
            case 0: base_a = lead_x;  base_b = lead_y;  break;
            case 1: base_a = lead_y;  base_b = -lead_x; break;
            case 2: base_a = -lead_x; base_b = -lead_y; break;
            default: base_a = -lead_y; base_b = lead_x; break;
            }
        }
        seq[main_element] += 1;
        if (seq[main_element] == n) seq[main_element] = 0;
        {
            float dx = px[seq[1]] - px[seq[3]];
            float dy = py[seq[1]] - py[seq[3]];
            float width = dx * base_a + dy * base_b;
            dx = px[seq[2]] - px[seq[0]];
            dy = py[seq[2]] - py[seq[0]];
            float height = -dx * base_b + dy * base_a;
            float area = width * height;
            if (area <= minarea) {
                minarea = area;
                b_left = seq[3]; b_a = base_a; b_w = width; b_b = base_b; b_h = height;
                b_bottom = seq[0];
            }
        }
    }
    {
        float A1 = b_a, B1 = b_b, A2 = -b_b, B2 = b_a;
        float C1 = A1 * px[b_left] + py[b_left] * B1;
        float C2 = A2 * px[b_bottom] + py[b_bottom] * B2;
        float idet = 1.f / (A1 * B2 - A2 * B1);
        out[0] = (C1 * B2 - C2 * B1) * idet;
        out[1] = (A1 * C2 - A2 * C1) * idet;
        out[2] = A1 * b_w; out[3] = B1 * b_w;
        out[4] = A2 * b_h; out[5] = B2 * b_h;
    }
    free(vy); free(vx); free(inv_len);
}

void ora_min_area_rect(const int32_t* pts, int n, float* out_wh)
{
    out_wh[0] = out_wh[1] = 0.f;
    if (n <= 0) return;
    ipt* hull = (ipt*)malloc(sizeof(ipt) * (size_t)(n + 2));
    int hn = hull_cv_order(pts, n, hull);
    if (hn > 2) {
        float* hx = (float*)malloc(sizeof(float) * (size_t)hn);
        float* hy = (float*)malloc(sizeof(float) * (size_t)hn);
        for (int i = 0; i < hn; i++) { hx[i] = (float)hull[i].x; hy[i] = (float)hull[i].y; }
        float out[6];
        rotating_calipers_minarea(hx, hy, hn, out);
        out_wh[0] = (float)sqrt((double)out[2] * out[2] + (double)out[3] * out[3]);
        out_wh[1] = (float)sqrt((double)out[4] * out[4] + (double)out[5] * out[5]);
        free(hy); free(hx);
    } else if (hn == 2) {
        double dx = (double)((float)hull[1].x - (float)hull[0].x);
        double dy = (double)((float)hull[1].y - (float)hull[0].y);
        out_wh[0] = (float)sqrt(dx * dx + dy * dy);
        out_wh[1] = 0.f;
    }
    free(hull);
}

/* cv2.minAreaRect as the Python binding returns it: (w, h, angle in degrees); the centre is not used by any caller
 * (sf_contours.py:201-203, 268-277).  n > 2: the calipers' first vector gives the width and the angle
 * (atan2 in double, cast to float, then float * 180 / CV_PI in double, cast to float); n == 2: the segment. */
void ora_min_area_rect_box(const int32_t* pts, int n, float* out_wha)
{
    out_wha[0] = out_wha[1] = out_wha[2] = 0.f;
    if (n <= 0) return;
    ipt* hull = (ipt*)malloc(sizeof(ipt) * (size_t)(n + 2));
    int hn = hull_cv_order(pts, n, hull);
    float angle = 0.f;
    if (hn > 2) {
        float* hx = (float*)malloc(sizeof(float) * (size_t)hn);
        float* hy = (float*)malloc(sizeof(float) * (size_t)hn);
        for (int i = 0; i < hn; i++) { hx[i] = (float)hull[i].x; hy[i] = (float)hull[i].y; }
        float out[6];
        rotating_calipers_minarea(hx, hy, hn, out);
        out_wha[0] = (float)sqrt((double)out[2] * out[2] + (double)out[3] * out[3]);
        out_wha[1] = (float)sqrt((double)out[4] * out[4] + (double)out[5] * out[5]);
        angle = (float)atan2((double)out[3], (double)out[2]);
        free(hy); free(hx);
    } else if (hn == 2) {
        double dx = (double)((float)hull[1].x - (float)hull[0].x);
        double dy = (double)((float)hull[1].y - (float)hull[0].y);
        out_wha[0] = (float)sqrt(dx * dx + dy * dy);
        angle = (float)atan2(dy, dx);
    }
    out_wha[2] = (float)((double)(angle * 180.f) / 3.1415926535897932384626433832795);
    free(hull);
}

/* cv2.convexHull(points) as a vertex set (strictly convex: collinear and duplicate points dropped); the order is
 * the one hull_cv_order documents.  Returns the vertex count. */
int ora_convex_hull(const int32_t* pts, int n, int32_t* out_xy)
{
    if (n <= 0) return 0;
    ipt* hull = (ipt*)malloc(sizeof(ipt) * (size_t)(n + 2));
    int hn = hull_cv_order(pts, n, hull);
    for (int i = 0; i < hn; i++) { out_xy[2 * i] = hull[i].x; out_xy[2 * i + 1] = hull[i].y; }
    free(hull);
    return hn;
}

int ora_top3(const double* areas, int n, int* out_pos, double* biggest)
{
    /* bisect.insort == insort_right on BoundingBox.__lt__ (area <): an element is placed
     * after every element whose area is <= its own. */
    int* order = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int m = 0;
    for (int i = 0; i < n; i++) {
        int lo = 0, hi = m;
        while (lo < hi) {
            int mid = (lo + hi) / 2;
            if (areas[i] < areas[order[mid]]) hi = mid; else lo = mid + 1;
        }
        memmove(order + lo + 1, order + lo, sizeof(int) * (size_t)(m - lo));
        order[lo] = i; m++;
    }
    int k = n < 3 ? n : 3;
    for (int i = 0; i < k; i++) out_pos[i] = order[n - k + i];
    if (n > 0 && biggest) *biggest = areas[order[n - 1]];
    free(order);
    return k;
}

int ora_board_lines(const uint8_t* edges, int h, int w, int hough_thresh,
                    uint8_t* ghost, float* lines, int cap, double* biggest_area,
                    int* n_contours)
{
    const size_t npx = (size_t)h * w;
    int maxc = (int)(npx / 2 + 4);
    int* starts = (int*)malloc(sizeof(int) * 2 * (size_t)maxc);
    int* pix_off = (int*)malloc(sizeof(int) * (size_t)(maxc + 1));
    int* vert_off = (int*)malloc(sizeof(int) * (size_t)(maxc + 1));
    int maxp = (int)(4 * npx + 16);
    int* pix = (int*)malloc(sizeof(int) * 2 * (size_t)maxp);
    int* vert = (int*)malloc(sizeof(int) * 2 * (size_t)maxp);
    int nc = ora_find_external_suzuki(edges, h, w, maxc, starts, pix_off, vert_off, maxp, pix, maxp, vert);
    int ret;
    memset(ghost, 0, npx);
    if (n_contours) *n_contours = nc;
    if (nc <= 0) { ret = -1; goto done; }
    {
        /* cv2 hands the contours back in reverse discovery order */
        double* areas = (double*)malloc(sizeof(double) * (size_t)nc);
        for (int k = 0; k < nc; k++) {
            int d = nc - 1 - k;            /* discovery index of cv2 position k */
            float wh[2];
            ora_min_area_rect(vert + 2 * vert_off[d], vert_off[d + 1] - vert_off[d], wh);
            areas[k] = (double)wh[0] * (double)wh[1];
        }
        int pos[3]; double biggest = 0;
        int np = ora_top3(areas, nc, pos, &biggest);
        if (biggest_area) *biggest_area = biggest;
        double frame_area = (double)h * (double)w;
        if (!(frame_area / 3 < biggest)) { free(areas); ret = -2; goto done; }
        for (int k = 0; k < np; k++) {
            int d = nc - 1 - pos[k];
            for (int i = pix_off[d]; i < pix_off[d + 1]; i++)
                ghost[(size_t)pix[2 * i + 1] * w + pix[2 * i]] = 255;
        }
        free(areas);
        ret = ora_hough_lines(ghost, h, w, hough_thresh, lines, cap, 0);
    }
done:
    free(vert); free(pix); free(vert_off); free(pix_off); free(starts);
    return ret;
}
