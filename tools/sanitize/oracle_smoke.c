/* Runs every oracle routine once on small random inputs under ASan/UBSan (tools/sanitize/run.sh). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ck_oracle.h"

static uint32_t s = 12345u;
static uint32_t rnd(void) { s = s * 1664525u + 1013904223u; return s >> 8; }

int main(void)
{
    const int sizes[][2] = { {16, 16}, {37, 53}, {64, 48}, {1, 1}, {3, 40}, {97, 131} };
    for (unsigned t = 0; t < sizeof sizes / sizeof sizes[0]; t++) {
        const int h = sizes[t][0], w = sizes[t][1];
        uint8_t* img = malloc((size_t)h * w * 3);
        uint8_t* med = malloc((size_t)h * w * 3);
        uint8_t* edges = malloc((size_t)h * w);
        uint8_t* map = malloc((size_t)h * w);
        uint8_t* ghost = malloc((size_t)h * w);
        for (int i = 0; i < h * w * 3; i++) img[i] = (uint8_t)(rnd() % 256 < 40 ? 250 : 30 + rnd() % 20);
        ora_median(img, h, w, 3, 15, med);
        ora_canny(med, h, w, 3, 25, 75, edges, map, NULL, NULL, NULL);
        for (int i = 0; i < h * w; i++) edges[i] = (rnd() % 100 < 30) ? 255 : 0;      /* dense random edge map */
        if (h >= 3 && w >= 3) {
            float lines[64 * 2];
            double big = 0;
            int nc = 0;
            (void)ora_board_lines(edges, h, w, 4, ghost, lines, 64, &big, &nc);
            int32_t* lab = malloc((size_t)h * w * 4);
            int* starts = malloc((size_t)h * w * 2 * sizeof(int));
            (void)ora_find_external_sets(edges, h, w, lab, h * w, starts);
            free(lab); free(starts);
        }
        (void)ora_hough_lines(edges, h, w, 3, (float[32]){0}, 16, NULL);
        if (h % 2 == 0 && w % 2 == 0) {
            uint8_t* yuv = malloc((size_t)h * w * 3 / 2);
            for (int i = 0; i < h * w * 3 / 2; i++) yuv[i] = (uint8_t)rnd();
            ora_i420_to_bgr(yuv, h, w, img);
            free(yuv);
        }
        free(img); free(med); free(edges); free(map); free(ghost);
    }
    {   /* warp + mog2 + cnn on one goban-sized image */
        const int h = 120, w = 160;
        uint8_t* img = malloc((size_t)h * w * 3);
        for (int i = 0; i < h * w * 3; i++) img[i] = (uint8_t)rnd();
        const float src[8] = {20, 10, 140, 14, 150, 110, 12, 100}, dst[8] = {0, 0, 380, 0, 380, 380, 0, 380};
        double M[9];
        if (ora_get_perspective_transform(src, dst, M)) return 2;
        uint8_t* gob = malloc(380 * 380 * 3);
        ora_warp_perspective(img, h, w, 3, M, 380, 380, gob, NULL);
        ora_mog2* m = ora_mog2_create(380, 380, 3);
        uint8_t* fg = malloc(380 * 380);
        for (int k = 0; k < 3; k++) ora_mog2_apply(m, gob, 0.01, fg);
        ora_mog2_destroy(m);
        const size_t cnt[12] = {5 * 5 * 3 * 32, 32, 5 * 5 * 32 * 32, 32, 3 * 3 * 32 * 90, 90, 3 * 3 * 90 * 90, 90, 3240 * 160, 160, 160 * 81, 81};
        float* wts[12];
        for (int i = 0; i < 12; i++) {
            wts[i] = malloc(cnt[i] * 4);
            for (size_t k = 0; k < cnt[i]; k++) wts[i][k] = ((int)(rnd() % 2001) - 1000) * (i == 0 ? 1e-6f : 3e-5f);
        }
        ora_cnn_weights W = {wts[0], wts[1], wts[2], wts[3], wts[4], wts[5], wts[6], wts[7], wts[8], wts[9], wts[10], wts[11]};
        float* y = malloc(100 * 81 * 4);
        ora_cnn_predict_regions(&W, gob, y, NULL);
        uint8_t labels[361];
        double conf[361];
        ora_decode_all(y, labels, conf);
        for (int i = 0; i < 12; i++) free(wts[i]);
        free(y); free(fg); free(gob); free(img);
    }
    printf("oracle: every routine clean under ASan/UBSan\n");
    return 0;
}
