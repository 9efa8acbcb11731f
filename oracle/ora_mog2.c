/*
 * ora_mog2.c -- ORACLE (test infrastructure only; see ck_oracle.h).
 *
 * K9  self.bg_model = cv2.createBackgroundSubtractorMOG2(detectShadows=False)
 *     self._fg = self.bg_model.apply(self.goban_img, learningRate=learning)
 *     (/root/reference/src/camkifu/stone/stonesfinder.py:113-115, 171-176)
 *
 * Zivkovic's adaptive GMM with the library defaults: history 500, 5 mixtures,
 * varThreshold (Tb) 16, varThresholdGen (Tg) 9, backgroundRatio (TB) 0.9, varInit 15,
 * varMin 4, varMax 75, complexity reduction CT 0.05, no shadow detection.
 * float32 arithmetic, one model per pixel, modes kept sorted by weight.
 * "parity unpinned": the reference holds no MOG2 vector and OpenCV cannot be run here, so this is a
 * restatement of the published algorithm -- Z. Zivkovic, "Improved adaptive Gaussian mixture model for
 * background subtraction" (ICPR 2004) and Zivkovic & van der Heijden, "Efficient adaptive density estimation
 * per image pixel ..." (PRL 2006), whose reference implementation OpenCV adopted as bgfg_gaussmix2.
 * What is ASSUMED beyond the papers' equations (they define the update, not the bookkeeping) is marked
 * ASSUMPTION below: (1) the order of the per-mode loop body (decay, match test on the first fitting mode only,
 * sort by bubbling the matched mode up, prune); (2) the prune test `w < -prune` with prune = -alpha * CT sits at
 * the END of the loop body and shortens the loop bound itself (see PRUNE NOTE); (3) a new mode replaces the
 * weakest one when all five are in use.  tests/test_gpu_parity.py::test_cv2_live_crosscheck compares masks with
 * the real library whenever a machine has it.
 */
#include "ck_oracle.h"
#include <stdlib.h>
#include <string.h>

enum { NMIX = 5 };

struct ora_mog2 {
    int h, w, cn, nframes;
    float* weight;    /* h*w*NMIX */
    float* variance;  /* h*w*NMIX */
    float* mean;      /* h*w*NMIX*cn */
    uint8_t* nmodes;  /* h*w */
};

ora_mog2* ora_mog2_create(int h, int w, int cn)
{
    ora_mog2* m = (ora_mog2*)calloc(1, sizeof(*m));
    size_t n = (size_t)h * w;
    m->h = h; m->w = w; m->cn = cn; m->nframes = 0;
    m->weight = (float*)calloc(n * NMIX, sizeof(float));
    m->variance = (float*)calloc(n * NMIX, sizeof(float));
    m->mean = (float*)calloc(n * NMIX * (size_t)cn, sizeof(float));
    m->nmodes = (uint8_t*)calloc(n, 1);
    return m;
}

void ora_mog2_destroy(ora_mog2* m)
{
    if (!m) return;
    free(m->weight); free(m->variance); free(m->mean); free(m->nmodes); free(m);
}

void ora_mog2_apply(ora_mog2* m, const uint8_t* img, double learning_rate, uint8_t* fgmask)
{
    const int history = 500;
    const float Tb = 16.f, Tg = 9.f, TB = 0.9f;
    const float varInit = 15.f, varMin = 4.f, varMax = 5 * 15.f, fCT = 0.05f;
    const int cn = m->cn;

    if (learning_rate >= 1) {   /* re-initialise */
        size_t n = (size_t)m->h * m->w;
        memset(m->nmodes, 0, n);
        m->nframes = 0;
    }
    ++m->nframes;
    int lim = 2 * m->nframes < history ? 2 * m->nframes : history;
    double lr = (learning_rate >= 0 && m->nframes > 1) ? learning_rate : 1. / lim;
    const float alphaT = (float)lr;
    const float alpha1 = 1.f - alphaT;
    const float prune = (float)(-lr * fCT);

    size_t npx = (size_t)m->h * m->w;
    for (size_t px = 0; px < npx; px++) {
        float* gw = m->weight + px * NMIX;
        float* gv = m->variance + px * NMIX;
        float* mean = m->mean + px * NMIX * (size_t)cn;
        float data[4], dData[4];
        for (int c = 0; c < cn; c++) data[c] = (float)img[px * (size_t)cn + c];

        int background = 0, fitsPDF = 0;
        int nmodes = m->nmodes[px], nNewModes = nmodes;
        float totalWeight = 0.f;
        float* mean_m = mean;
        for (int mode = 0; mode < nmodes; mode++, mean_m += cn) {
            float weight = alpha1 * gw[mode] + prune;
            int swap_count = 0;
            if (!fitsPDF) {
                float var = gv[mode];
                float dist2 = 0.f;
                if (cn == 3) {
                    dData[0] = mean_m[0] - data[0];
                    dData[1] = mean_m[1] - data[1];
                    dData[2] = mean_m[2] - data[2];
                    dist2 = dData[0] * dData[0] + dData[1] * dData[1] + dData[2] * dData[2];
                } else {
                    for (int c = 0; c < cn; c++) { dData[c] = mean_m[c] - data[c]; dist2 += dData[c] * dData[c]; }
                }
                if (totalWeight < TB && dist2 < Tb * var) background = 1;
                if (dist2 < Tg * var) {
                    fitsPDF = 1;
                    weight += alphaT;
                    float k = alphaT / weight;
                    for (int c = 0; c < cn; c++) mean_m[c] -= k * dData[c];
                    float varnew = var + k * (dist2 - var);
                    varnew = varnew > varMin ? varnew : varMin;
                    varnew = varnew < varMax ? varnew : varMax;
                    gv[mode] = varnew;
                    for (int i = mode; i > 0; i--) {
                        if (weight < gw[i - 1]) break;
                        swap_count++;
                        float t;
                        t = gw[i]; gw[i] = gw[i - 1]; gw[i - 1] = t;
                        t = gv[i]; gv[i] = gv[i - 1]; gv[i - 1] = t;
                        for (int c = 0; c < cn; c++) {
                            t = mean[i * cn + c]; mean[i * cn + c] = mean[(i - 1) * cn + c]; mean[(i - 1) * cn + c] = t;
                        }
                    }
                }
            }
            /* PRUNE NOTE (ASSUMPTION 2).  Zivkovic's complexity-reduction prior subtracts alpha * CT from every
             * weight (eq. 14 of the 2006 paper: w <- w + alpha (o - w) - alpha c_T) and discards a component whose
             * weight becomes negative.  Here, as in his implementation, `prune` = -alpha * CT is already added to
             * `weight` above, the test is weight < -prune (i.e. below +alpha * CT: the component could not survive
             * the next subtraction), and discarding = zero weight + one fewer live mode; because `nmodes` is the
             * bound of this very loop and the modes are sorted by weight, only trailing modes are ever cut. */
            if (weight < -prune) { weight = 0.f; nmodes--; }
            gw[mode - swap_count] = weight;
            totalWeight += weight;
        }
        totalWeight = 1.f / totalWeight;
        for (int mode = 0; mode < nmodes; mode++) gw[mode] *= totalWeight;
        (void)nNewModes;

        if (!fitsPDF && alphaT > 0.f) {
            int mode = (nmodes == NMIX) ? NMIX - 1 : nmodes++;
            if (nmodes == 1) gw[mode] = 1.f;
            else {
                gw[mode] = alphaT;
                for (int i = 0; i < nmodes - 1; i++) gw[i] *= alpha1;
            }
            for (int c = 0; c < cn; c++) mean[mode * cn + c] = data[c];
            gv[mode] = varInit;
            for (int i = nmodes - 1; i > 0; i--) {
                if (alphaT < gw[i - 1]) break;
                float t;
                t = gw[i]; gw[i] = gw[i - 1]; gw[i - 1] = t;
                t = gv[i]; gv[i] = gv[i - 1]; gv[i - 1] = t;
                for (int c = 0; c < cn; c++) {
                    t = mean[i * cn + c]; mean[i * cn + c] = mean[(i - 1) * cn + c]; mean[(i - 1) * cn + c] = t;
                }
            }
        }
        m->nmodes[px] = (uint8_t)nmodes;
        fgmask[px] = background ? 0 : 255;
    }
}
