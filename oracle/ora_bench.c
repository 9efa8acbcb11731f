/* ora_bench.c -- TEST / BENCH INFRASTRUCTURE ONLY (never linked or loaded by the product).
 *
 * The cpu_baseline leg of bench.py, as BASELINE.md section 3 prescribes it: the per-frame hot path of the reference
 * (board/bf_auto.py:72-84, 105-133: medianBlur 15 -> Canny 25/75 -> findContours -> minAreaRect sort -> drawContours x3
 * -> HoughLines; stone/stonesfinder.py:140 + stone/nn_cache.py:33-41: warpPerspective -> 100 patches through the
 * classifier -> 19x19 labels) restated by this oracle, with OpenMP ACROSS frames: one frame per thread, every routine
 * inside runs single-threaded (nested parallel regions are inactive by default). */
#include <stdlib.h>
#include <string.h>

#include "ck_oracle.h"

#ifdef _OPENMP
#include <omp.h>
#endif

/* frames: n x h x w x 3 BGR; M: 3x3 transform (src -> 380 x 380); n_lines_out: n (ora_board_lines' return value per
 * frame); W NULL: the classifier is left to the caller (BASELINE.md 3 times it with torch on the CPU) and the goban
 * images go to gobans_out (n x 380 x 380 x 3); otherwise labels_out: n x 361.  Returns the number of threads that took
 * part, or -1 when memory ran out. */
int ora_baseline_frames(const uint8_t* frames, int n, int h, int w, const double* M, const ora_cnn_weights* W,
                        int threads, int32_t* n_lines_out, uint8_t* labels_out, uint8_t* gobans_out)
{
    int used = 1, failed = 0;
    const size_t px = (size_t)h * w;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
#ifdef _OPENMP
#pragma omp single
        used = omp_get_num_threads();
#endif
        uint8_t* med = (uint8_t*)malloc(px * 3);
        uint8_t* edges = (uint8_t*)malloc(px);
        uint8_t* ghost = (uint8_t*)malloc(px);
        uint8_t* goban = (uint8_t*)malloc(380 * 380 * 3);
        float* lines = (float*)malloc(4096 * 2 * sizeof(float));
        float* y = (float*)malloc(100 * 81 * sizeof(float));
        double conf[361];
        if (!med || !edges || !ghost || !goban || !lines || !y) {
#pragma omp atomic write
            failed = 1;
        }
#pragma omp barrier
        if (!failed) {
#pragma omp for schedule(dynamic, 1)
            for (int f = 0; f < n; f++) {
                const uint8_t* fr = frames + (size_t)f * px * 3;
                double biggest = 0;
                int ncont = 0;
                ora_median(fr, h, w, 3, 15, med);
                ora_canny(med, h, w, 3, 25, 75, edges, NULL, NULL, NULL, NULL);
                n_lines_out[f] = ora_board_lines(edges, h, w, (int)((h < w ? h : w) / 5.0), ghost, lines, 4096, &biggest, &ncont);
                uint8_t* g = gobans_out ? gobans_out + (size_t)f * 380 * 380 * 3 : goban;
                ora_warp_perspective(fr, h, w, 3, M, 380, 380, g, NULL);
                if (W) {
                    ora_cnn_predict_regions(W, g, y, NULL);
                    ora_decode_all(y, labels_out + (size_t)f * 361, conf);
                }
            }
        }
        free(med); free(edges); free(ghost); free(goban); free(lines); free(y);
    }
    return failed ? -1 : used;
}
