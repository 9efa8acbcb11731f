// k_mog2.hip -- K9: BackgroundSubtractorMOG2(detectShadows=False).apply(goban_img, lr)
// (reference: src/camkifu/stone/stonesfinder.py:113-115, 171-176).
//
// One thread per pixel of the 380x380 board image; the 5-mode mixture lives in HBM in
// structure-of-arrays form ([mode][pixel]) so every access is coalesced.  The model is
// per-stream state: frames of one stream are applied strictly in order, so this stage is
// never sharded by frame.  float32 arithmetic without contraction, same operation order as
// the scalar formulation (Zivkovic's update with the library defaults).
#include "ck_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int NMIX = 5;

__global__ __launch_bounds__(256) void mog2_kernel(const uint8_t* __restrict__ img, int npx,
                                                   float* __restrict__ gw_, float* __restrict__ gv_,
                                                   float* __restrict__ mean_, uint8_t* __restrict__ nmodes_,
                                                   float alphaT, float prune, uint8_t* __restrict__ fg)
{
#pragma clang fp contract(off)
    const int px = blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= npx) return;
    const float Tb = 16.f, Tg = 9.f, TB = 0.9f;
    const float varInit = 15.f, varMin = 4.f, varMax = 75.f;
    const float alpha1 = 1.f - alphaT;

    float gw[NMIX], gv[NMIX], mean[NMIX][3];
    int nmodes = nmodes_[px];
#pragma unroll
    for (int k = 0; k < NMIX; k++) {
        if (k < nmodes) {
            gw[k] = gw_[(size_t)k * npx + px];
            gv[k] = gv_[(size_t)k * npx + px];
#pragma unroll
            for (int c = 0; c < 3; c++) mean[k][c] = mean_[((size_t)k * 3 + c) * npx + px];
        } else {
            gw[k] = 0.f; gv[k] = 0.f; mean[k][0] = mean[k][1] = mean[k][2] = 0.f;
        }
    }
    float data[3];
#pragma unroll
    for (int c = 0; c < 3; c++) data[c] = (float)img[(size_t)px * 3 + c];

    bool background = false, fitsPDF = false;
    float totalWeight = 0.f;
    int bound = nmodes;
#pragma unroll
    for (int mode = 0; mode < NMIX; mode++) {
        if (mode < bound) {
            float weight = alpha1 * gw[mode] + prune;
            int dst = mode;
            if (!fitsPDF) {
                const float var = gv[mode];
                const float d0 = mean[mode][0] - data[0];
                const float d1 = mean[mode][1] - data[1];
                const float d2 = mean[mode][2] - data[2];
                const float dist2 = d0 * d0 + d1 * d1 + d2 * d2;
                if (totalWeight < TB && dist2 < Tb * var) background = true;
                if (dist2 < Tg * var) {
                    fitsPDF = true;
                    weight += alphaT;
                    const float k = alphaT / weight;
                    mean[mode][0] -= k * d0;
                    mean[mode][1] -= k * d1;
                    mean[mode][2] -= k * d2;
                    float varnew = var + k * (dist2 - var);
                    varnew = varnew > varMin ? varnew : varMin;
                    varnew = varnew < varMax ? varnew : varMax;
                    gv[mode] = varnew;
                    // bubble the matched mode up while its new weight is not smaller
#pragma unroll
                    for (int i = NMIX - 1; i > 0; i--) {
                        if (i <= mode && i == dst && !(weight < gw[i - 1])) {
                            float t;
                            t = gw[i]; gw[i] = gw[i - 1]; gw[i - 1] = t;
                            t = gv[i]; gv[i] = gv[i - 1]; gv[i - 1] = t;
#pragma unroll
                            for (int c = 0; c < 3; c++) { t = mean[i][c]; mean[i][c] = mean[i - 1][c]; mean[i - 1][c] = t; }
                            dst = i - 1;
                        }
                    }
                }
            }
            if (weight < -prune) { weight = 0.f; bound--; }
            gw[dst] = weight;
            totalWeight += weight;
        }
    }
    nmodes = bound;
    totalWeight = 1.f / totalWeight;
#pragma unroll
    for (int mode = 0; mode < NMIX; mode++)
        if (mode < nmodes) gw[mode] *= totalWeight;

    if (!fitsPDF && alphaT > 0.f) {
        const int mode = (nmodes == NMIX) ? NMIX - 1 : nmodes++;
        float nw;
        if (nmodes == 1) nw = 1.f;
        else {
            nw = alphaT;
#pragma unroll
            for (int i = 0; i < NMIX; i++)
                if (i < nmodes - 1) gw[i] *= alpha1;
        }
        // write into slot `mode`, then bubble up while alphaT is not smaller than the one above
        int dst = mode;
#pragma unroll
        for (int k = 0; k < NMIX; k++)
            if (k == mode) { gw[k] = nw; gv[k] = varInit; mean[k][0] = data[0]; mean[k][1] = data[1]; mean[k][2] = data[2]; }
#pragma unroll
        for (int i = NMIX - 1; i > 0; i--) {
            if (i <= nmodes - 1 && i == dst && !(alphaT < gw[i - 1])) {
                float t;
                t = gw[i]; gw[i] = gw[i - 1]; gw[i - 1] = t;
                t = gv[i]; gv[i] = gv[i - 1]; gv[i - 1] = t;
#pragma unroll
                for (int c = 0; c < 3; c++) { t = mean[i][c]; mean[i][c] = mean[i - 1][c]; mean[i - 1][c] = t; }
                dst = i - 1;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NMIX; k++) {
        gw_[(size_t)k * npx + px] = gw[k];
        gv_[(size_t)k * npx + px] = gv[k];
#pragma unroll
        for (int c = 0; c < 3; c++) mean_[((size_t)k * 3 + c) * npx + px] = mean[k][c];
    }
    nmodes_[px] = (uint8_t)nmodes;
    fg[px] = background ? 0 : 255;
}

}  // namespace

int k_mog2_apply(ck_ctx* ctx, Mog2State& st, const uint8_t* d_img, double learning_rate, uint8_t* d_fg)
{
    TimeScope ts(ctx, "mog2");
    const int history = 500;
    const int npx = st.h * st.w;
    if (learning_rate >= 1) {
        CK_HIP(ctx, hipMemsetAsync(st.nmodes.p, 0, (size_t)npx, ctx->stream));
        st.nframes = 0;
    }
    ++st.nframes;
    const int lim = 2 * st.nframes < history ? 2 * st.nframes : history;
    const double lr = (learning_rate >= 0 && st.nframes > 1) ? learning_rate : 1. / lim;
    const float alphaT = (float)lr;
    const float prune = (float)(-lr * 0.05f);
    hipLaunchKernelGGL(mog2_kernel, dim3((npx + 255) / 256), dim3(256), 0, ctx->stream, d_img, npx,
                       (float*)st.weight.p, (float*)st.variance.p, (float*)st.mean.p, (uint8_t*)st.nmodes.p,
                       alphaT, prune, d_fg);
    CK_HIP(ctx, hipGetLastError());
    return CK_OK;
}
