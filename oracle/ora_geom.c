/*
 * ora_geom.c -- ORACLE (test infrastructure only; see ck_oracle.h).
 *
 * K6  cv2.HoughLines(ghost, 1, math.pi / 180, threshold=int(length_ref / 5))
 *     (/root/reference/src/camkifu/board/bf_auto.py:131-133)
 * K7  cv2.getPerspectiveTransform(hull, transform_dst)
 *     (/root/reference/src/camkifu/board/boardfinder.py:43-45)
 * K8  cv2.warpPerspective(frame, mtx, (380, 380))
 *     (/root/reference/src/camkifu/stone/stonesfinder.py:140)
 *
 * Library semantics restated (OpenCV 3.1.0, standard Hough transform / remap):
 * float32 sin/cos tables built by repeated float addition of theta, votes at
 * round-half-even(x*cos + y*sin) with each product and the sum rounded to float32,
 * 4-neighbour peak test with mixed strict / non-strict comparisons, sort by votes then
 * index; warp with double-precision homography evaluated in 64x16 blocks, coordinates
 * quantised to 1/32 px, 15-bit integer bilinear weights.  "parity unpinned".
 */
#include "ck_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORA_PI 3.1415926535897932384626433832795

static inline int round_half_even(double v) { return (int)nearbyint(v); /* default FE_TONEAREST */ }

static const int32_t* g_sort_accum;
static int cmp_peaks(const void* a, const void* b)
{
    int l1 = *(const int*)a, l2 = *(const int*)b;
    int v1 = g_sort_accum[l1], v2 = g_sort_accum[l2];
    if (v1 != v2) return v1 > v2 ? -1 : 1;
    return (l1 > l2) - (l1 < l2);
}

int ora_hough_lines(const uint8_t* img, int h, int w, int threshold,
                    float* lines, int cap, int32_t* accum_out)
{
    const float rho = 1.f;
    const float theta = (float)(ORA_PI / 180);       /* python passes math.pi/180; C++ narrows to float */
    const double min_theta = 0, max_theta = ORA_PI;
    const float irho = 1 / rho;
    const int numangle = round_half_even((max_theta - min_theta) / theta);
    const int numrho = round_half_even(((w + h) * 2 + 1) / rho);
    const size_t asz = (size_t)(numangle + 2) * (numrho + 2);
    int32_t* accum = (int32_t*)calloc(asz, sizeof(int32_t));
    float* tabSin = (float*)malloc(sizeof(float) * (size_t)numangle);
    float* tabCos = (float*)malloc(sizeof(float) * (size_t)numangle);

    float ang = (float)min_theta;
    for (int n = 0; n < numangle; ang += theta, n++) {
        tabSin[n] = (float)(sin((double)ang) * irho);
        tabCos[n] = (float)(cos((double)ang) * irho);
    }
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++)
            if (img[(size_t)i * w + j] != 0)
                for (int n = 0; n < numangle; n++) {
                    float a = (float)j * tabCos[n];
                    float b = (float)i * tabSin[n];
                    float s = a + b;
                    int r = round_half_even((double)s);
                    r += (numrho - 1) / 2;
                    accum[(size_t)(n + 1) * (numrho + 2) + r + 1]++;
                }
    int* peaks = (int*)malloc(sizeof(int) * (size_t)numangle * (size_t)numrho / 4 + 64);
    int np = 0;
    for (int r = 0; r < numrho; r++)
        for (int n = 0; n < numangle; n++) {
            int base = (n + 1) * (numrho + 2) + r + 1;
            if (accum[base] > threshold &&
                accum[base] > accum[base - 1] && accum[base] >= accum[base + 1] &&
                accum[base] > accum[base - numrho - 2] && accum[base] >= accum[base + numrho + 2])
                peaks[np++] = base;
        }
    g_sort_accum = accum;
    qsort(peaks, (size_t)np, sizeof(int), cmp_peaks);
    double scale = 1. / (numrho + 2);
    for (int i = 0; i < np && i < cap; i++) {
        int idx = peaks[i];
        int n = (int)floor(idx * scale) - 1;
        int r = idx - (n + 1) * (numrho + 2) - 1;
        lines[2 * i] = (r - (numrho - 1) * 0.5f) * rho;
        lines[2 * i + 1] = (float)min_theta + n * theta;
    }
    if (accum_out) memcpy(accum_out, accum, asz * sizeof(int32_t));
    free(peaks); free(tabCos); free(tabSin); free(accum);
    return np;
}

int ora_get_perspective_transform(const float* src, const float* dst, double* M)
{
    double a[8][9];
    for (int i = 0; i < 4; i++) {
        double sx = src[2 * i], sy = src[2 * i + 1], dx = dst[2 * i], dy = dst[2 * i + 1];
        double r0[9] = { sx, sy, 1, 0, 0, 0, -sx * dx, -sy * dx, dx };
        double r1[9] = { 0, 0, 0, sx, sy, 1, -sx * dy, -sy * dy, dy };
        memcpy(a[i], r0, sizeof r0);
        memcpy(a[i + 4], r1, sizeof r1);
    }
    for (int c = 0; c < 8; c++) {
        int piv = c;
        for (int r = c + 1; r < 8; r++) if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
        if (fabs(a[piv][c]) < 1e-300) return -1;
        if (piv != c) for (int k = 0; k < 9; k++) { double t = a[c][k]; a[c][k] = a[piv][k]; a[piv][k] = t; }
        for (int r = 0; r < 8; r++) {
            if (r == c) continue;
            double f = a[r][c] / a[c][c];
            for (int k = c; k < 9; k++) a[r][k] -= f * a[c][k];
        }
    }
    for (int i = 0; i < 8; i++) M[i] = a[i][8] / a[i][i];
    M[8] = 1.;
    return 0;
}

static void invert3x3(const double* s, double* d)
{
    /* closed-form adjugate / determinant, as the library does for 3x3 */
    double det = s[0] * (s[4] * s[8] - s[5] * s[7]) - s[1] * (s[3] * s[8] - s[5] * s[6]) +
                 s[2] * (s[3] * s[7] - s[4] * s[6]);
    if (det == 0) { memset(d, 0, 9 * sizeof(double)); return; }
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
    memcpy(d, t, sizeof t);
}

static inline int sat_int(double v)
{
    if (v < -2147483648.0) v = -2147483648.0;
    if (v > 2147483647.0) v = 2147483647.0;
    return round_half_even(v);
}
static inline int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

void ora_warp_perspective(const uint8_t* src, int h, int w, int cn, const double* Min,
                          int dw, int dh, uint8_t* dst, double* Minv_out)
{
    enum { INTER_BITS = 5, TAB = 1 << INTER_BITS, BLOCK_SZ = 32 };
    double M[9];
    invert3x3(Min, M);
    if (Minv_out) memcpy(Minv_out, M, sizeof M);
    int bh0 = BLOCK_SZ / 2 < dh ? BLOCK_SZ / 2 : dh;
    int bw0 = BLOCK_SZ * BLOCK_SZ / bh0 < dw ? BLOCK_SZ * BLOCK_SZ / bh0 : dw;
    bh0 = BLOCK_SZ * BLOCK_SZ / bw0 < dh ? BLOCK_SZ * BLOCK_SZ / bw0 : dh;

    for (int y = 0; y < dh; y += bh0)
        for (int x = 0; x < dw; x += bw0) {
            int bw = bw0 < dw - x ? bw0 : dw - x;
            int bh = bh0 < dh - y ? bh0 : dh - y;
            for (int y1 = 0; y1 < bh; y1++) {
                double X0 = M[0] * x + M[1] * (y + y1) + M[2];
                double Y0 = M[3] * x + M[4] * (y + y1) + M[5];
                double W0 = M[6] * x + M[7] * (y + y1) + M[8];
                for (int x1 = 0; x1 < bw; x1++) {
                    double W = W0 + M[6] * x1;
                    W = W ? TAB / W : 0;
                    double fX = (X0 + M[0] * x1) * W;
                    double fY = (Y0 + M[3] * x1) * W;
                    int X = sat_int(fX), Y = sat_int(fY);
                    int sx = sat_short(X >> INTER_BITS), sy = sat_short(Y >> INTER_BITS);
                    int fx = X & (TAB - 1), fy = Y & (TAB - 1);
                    /* 15-bit bilinear weights; (fy,fx)=(0,0) is {32767,0,0,1} in the
                     * library table, which is arithmetically the same as {32768,0,0,0}. */
                    int w00 = (TAB - fy) * (TAB - fx) * 32, w01 = (TAB - fy) * fx * 32;
                    int w10 = fy * (TAB - fx) * 32, w11 = fy * fx * 32;
                    uint8_t* d = dst + ((size_t)(y + y1) * dw + (x + x1)) * cn;
                    for (int c = 0; c < cn; c++) {
                        int v00 = ((unsigned)sx < (unsigned)w && (unsigned)sy < (unsigned)h) ? src[((size_t)sy * w + sx) * cn + c] : 0;
                        int v01 = ((unsigned)(sx + 1) < (unsigned)w && (unsigned)sy < (unsigned)h) ? src[((size_t)sy * w + sx + 1) * cn + c] : 0;
                        int v10 = ((unsigned)sx < (unsigned)w && (unsigned)(sy + 1) < (unsigned)h) ? src[((size_t)(sy + 1) * w + sx) * cn + c] : 0;
                        int v11 = ((unsigned)(sx + 1) < (unsigned)w && (unsigned)(sy + 1) < (unsigned)h) ? src[((size_t)(sy + 1) * w + sx + 1) * cn + c] : 0;
                        int v = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
                        d[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
                    }
                }
            }
        }
}
