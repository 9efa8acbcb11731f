/*
 * ora_filter.c -- ORACLE (test infrastructure only; see ck_oracle.h).
 * K1 median blur and K2 Canny as BoardFinderAuto._detect calls them
 * (/root/reference/src/camkifu/board/bf_auto.py:72-73):
 *     median = cv2.medianBlur(frame, 15)
 *     canny  = cv2.Canny(median, 25, 75)
 * OpenCV 3.1.0 is not vendored in the reference; the semantics restated here are the
 * library's published behaviour: exact median with BORDER_REPLICATE; Canny with 3x3 Sobel
 * (BORDER_REPLICATE, int16), per-pixel choice of the channel with the largest L1
 * magnitude, non-maximum suppression with the fixed-point tan(22.5 deg) sector test and
 * 8-connected hysteresis.  "parity unpinned": no reference fixture pins these stages.
 */
#include "ck_oracle.h"
#include <stdlib.h>
#include <string.h>

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* K1: Huang sliding histogram per row, two-level (16 coarse + 256 fine) lookup.
 * The median of ksize*ksize samples is the element of rank (ksize*ksize)/2 (0-based). */
void ora_median(const uint8_t* src, int h, int w, int cn, int ksize, uint8_t* dst)
{
    const int r = ksize / 2;
    const int rank = (ksize * ksize) / 2;
    for (int c = 0; c < cn; c++) {
#pragma omp parallel for schedule(static)
        for (int y = 0; y < h; y++) {
            int fine[256];
            int coarse[16];
            memset(fine, 0, sizeof fine);
            memset(coarse, 0, sizeof coarse);
            /* window centred on x = 0 */
            for (int dy = -r; dy <= r; dy++) {
                const uint8_t* row = src + (size_t)clampi(y + dy, 0, h - 1) * w * cn;
                for (int dx = -r; dx <= r; dx++) {
                    int v = row[clampi(dx, 0, w - 1) * cn + c];
                    fine[v]++; coarse[v >> 4]++;
                }
            }
            for (int x = 0; x < w; x++) {
                if (x > 0) {
                    int xo = clampi(x - r - 1, 0, w - 1), xn = clampi(x + r, 0, w - 1);
                    for (int dy = -r; dy <= r; dy++) {
                        const uint8_t* row = src + (size_t)clampi(y + dy, 0, h - 1) * w * cn;
                        int vo = row[xo * cn + c], vn = row[xn * cn + c];
                        fine[vo]--; coarse[vo >> 4]--;
                        fine[vn]++; coarse[vn >> 4]++;
                    }
                }
                int acc = 0, cb = 0;
                while (acc + coarse[cb] <= rank) { acc += coarse[cb]; cb++; }
                int v = cb << 4;
                while (acc + fine[v] <= rank) { acc += fine[v]; v++; }
                dst[((size_t)y * w + x) * cn + c] = (uint8_t)v;
            }
        }
    }
}

/* K2 */
void ora_canny(const uint8_t* src, int h, int w, int cn, int low, int high,
               uint8_t* edges, uint8_t* map_out, int32_t* mag_out,
               int16_t* dx_out, int16_t* dy_out)
{
    const size_t npx = (size_t)h * w;
    int32_t* mag = (int32_t*)malloc(npx * sizeof(int32_t));
    int16_t* gx = (int16_t*)malloc(npx * sizeof(int16_t));
    int16_t* gy = (int16_t*)malloc(npx * sizeof(int16_t));
    uint8_t* map = (uint8_t*)malloc(npx);
    if (low > high) { int t = low; low = high; high = t; }

    /* Sobel 3x3, replicate border, per channel; keep the channel with the largest
     * |dx|+|dy| (first such channel on ties). */
    for (int y = 0; y < h; y++) {
        const uint8_t* r0 = src + (size_t)clampi(y - 1, 0, h - 1) * w * cn;
        const uint8_t* r1 = src + (size_t)y * w * cn;
        const uint8_t* r2 = src + (size_t)clampi(y + 1, 0, h - 1) * w * cn;
        for (int x = 0; x < w; x++) {
            int xm = clampi(x - 1, 0, w - 1) * cn, xc = x * cn, xp = clampi(x + 1, 0, w - 1) * cn;
            int best = -1, bdx = 0, bdy = 0;
            for (int c = 0; c < cn; c++) {
                int dx = (r0[xp + c] + 2 * r1[xp + c] + r2[xp + c]) -
                         (r0[xm + c] + 2 * r1[xm + c] + r2[xm + c]);
                int dy = (r2[xm + c] + 2 * r2[xc + c] + r2[xp + c]) -
                         (r0[xm + c] + 2 * r0[xc + c] + r0[xp + c]);
                int m = abs(dx) + abs(dy);
                if (m > best) { best = m; bdx = dx; bdy = dy; }
            }
            size_t i = (size_t)y * w + x;
            mag[i] = best; gx[i] = (int16_t)bdx; gy[i] = (int16_t)bdy;
        }
    }

    /* Non-maximum suppression.  Outside the image the magnitude is 0. */
#define MAG(yy, xx) (((yy) < 0 || (yy) >= h || (xx) < 0 || (xx) >= w) ? 0 : mag[(size_t)(yy) * w + (xx)])
    const int TG22 = (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5);
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            size_t i = (size_t)y * w + x;
            int m = mag[i];
            int keep = 0;
            if (m > low) {
                int xs = gx[i], ys = gy[i];
                int ax = abs(xs), ay = abs(ys) << 15;
                int tg22x = ax * TG22;
                if (ay < tg22x) {
                    keep = (m > MAG(y, x - 1) && m >= MAG(y, x + 1));
                } else {
                    int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x) {
                        keep = (m > MAG(y - 1, x) && m >= MAG(y + 1, x));
                    } else {
                        int s = ((xs ^ ys) < 0) ? -1 : 1;
                        keep = (m > MAG(y - 1, x - s) && m > MAG(y + 1, x + s));
                    }
                }
            }
            map[i] = keep ? (m > high ? 2 : 0) : 1;
        }
    }
#undef MAG
    if (map_out) memcpy(map_out, map, npx);
    if (mag_out) memcpy(mag_out, mag, npx * sizeof(int32_t));
    if (dx_out) memcpy(dx_out, gx, npx * sizeof(int16_t));
    if (dy_out) memcpy(dy_out, gy, npx * sizeof(int16_t));

    /* Hysteresis: flood from every strong pixel through candidates, 8-connected. */
    size_t* stack = (size_t*)malloc(npx * sizeof(size_t));
    size_t top = 0;
    for (size_t i = 0; i < npx; i++) if (map[i] == 2) stack[top++] = i;
    while (top) {
        size_t i = stack[--top];
        int y = (int)(i / w), x = (int)(i % w);
        for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
            int yy = y + dy, xx = x + dx;
            if ((dy | dx) == 0 || yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
            size_t j = (size_t)yy * w + xx;
            if (map[j] == 0) { map[j] = 2; stack[top++] = j; }
        }
    }
    for (size_t i = 0; i < npx; i++) edges[i] = (uint8_t)-(map[i] >> 1);
    free(stack); free(map); free(gy); free(gx); free(mag);
}

#ifdef _OPENMP
#include <omp.h>
int ora_num_threads(void) { return omp_get_max_threads(); }
#else
int ora_num_threads(void) { return 1; }
#endif
