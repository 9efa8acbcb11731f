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

// the per-pixel mixture, held in registers while a pixel is being updated
struct Mix {
    float gw[NMIX], gv[NMIX], mean[NMIX][3];
    int nmodes;
};

__device__ __forceinline__ void mix_load(Mix& m, const float* __restrict__ gw_, const float* __restrict__ gv_,
                                         const float* __restrict__ mean_, const uint8_t* __restrict__ nmodes_, int npx, int px)
{
    m.nmodes = nmodes_[px];
#pragma unroll
    for (int k = 0; k < NMIX; k++) {
        if (k < m.nmodes) {
            m.gw[k] = gw_[(size_t)k * npx + px];
            m.gv[k] = gv_[(size_t)k * npx + px];
#pragma unroll
            for (int c = 0; c < 3; c++) m.mean[k][c] = mean_[((size_t)k * 3 + c) * npx + px];
        } else {
            m.gw[k] = 0.f; m.gv[k] = 0.f; m.mean[k][0] = m.mean[k][1] = m.mean[k][2] = 0.f;
        }
    }
}

__device__ __forceinline__ void mix_store(const Mix& m, float* __restrict__ gw_, float* __restrict__ gv_,
                                          float* __restrict__ mean_, uint8_t* __restrict__ nmodes_, int npx, int px)
{
#pragma unroll
    for (int k = 0; k < NMIX; k++) {
        gw_[(size_t)k * npx + px] = m.gw[k];
        gv_[(size_t)k * npx + px] = m.gv[k];
#pragma unroll
        for (int c = 0; c < 3; c++) mean_[((size_t)k * 3 + c) * npx + px] = m.mean[k][c];
    }
    nmodes_[px] = (uint8_t)m.nmodes;
}

// one frame's update of one pixel (Zivkovic's update with the library defaults); returns "background"
__device__ __forceinline__ bool mix_update(Mix& m, const float x0, const float x1, const float x2, float alphaT, float prune)
{
#pragma clang fp contract(off)
    const float Tb = 16.f, Tg = 9.f, TB = 0.9f;
    const float varInit = 15.f, varMin = 4.f, varMax = 75.f;
    const float alpha1 = 1.f - alphaT;
    float (&gw)[NMIX] = m.gw;
    float (&gv)[NMIX] = m.gv;
    float (&mean)[NMIX][3] = m.mean;
    int nmodes = m.nmodes;

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
                const float d0 = mean[mode][0] - x0;
                const float d1 = mean[mode][1] - x1;
                const float d2 = mean[mode][2] - x2;
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
            // `dst` is a run-time value: a plain gw[dst] would send the whole mixture to scratch memory
#pragma unroll
            for (int k = 0; k < NMIX; k++)
                if (k <= mode) gw[k] = (k == dst) ? weight : gw[k];
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
        for (int k = 0; k < NMIX; k++) {              // selects, not branches: the compiler turns an if-chain on a
            const bool here = k == mode;              // run-time slot into a pointer to a scratch copy of the mixture
            gw[k] = here ? nw : gw[k];
            gv[k] = here ? varInit : gv[k];
            mean[k][0] = here ? x0 : mean[k][0];
            mean[k][1] = here ? x1 : mean[k][1];
            mean[k][2] = here ? x2 : mean[k][2];
        }
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
    m.nmodes = nmodes;
    return background;
}

__global__ __launch_bounds__(256) void mog2_kernel(const uint8_t* __restrict__ img, int npx,
                                                   float* __restrict__ gw_, float* __restrict__ gv_,
                                                   float* __restrict__ mean_, uint8_t* __restrict__ nmodes_,
                                                   float alphaT, float prune, uint8_t* __restrict__ fg)
{
    const int px = blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= npx) return;
    Mix m;
    mix_load(m, gw_, gv_, mean_, nmodes_, npx, px);
    const uint8_t* q = img + (size_t)px * 3;
    const bool background = mix_update(m, (float)q[0], (float)q[1], (float)q[2], alphaT, prune);
    mix_store(m, gw_, gv_, mean_, nmodes_, npx, px);
    fg[px] = background ? 0 : 255;
}

// An ORDERED RUN of n frames of one stream in one launch (the batch pipeline's shape).  A workgroup owns one
// 20x20-pixel block of the 380x380 goban = the zone of one intersection (StonesFinder.getrect, stonesfinder.py:412-450;
// the zones of the last row / column stop at pixel 379: that pixel is updated but not counted); a thread owns a pixel,
// keeps its mixture in registers across the whole run (state traffic: once per run instead of once per frame) and
// prefetches the next frame's pixel while updating with the current one.  What leaves the kernel per frame is the
// number of foreground pixels of the zone -- all SfNeural.is_agitated ever asks of the mask (sf_neural.py:178-180):
// wave ballots, one LDS add per wave and frame, no global atomics.
constexpr int ZONE = 20, ZTHREADS = 448;     // 400 pixels, 7 waves
// The image may be a horizontal BAND of the goban (multi-GPU: the background model is sharded by pixel, each rank
// keeps a band of intersection rows): h x w pixels, zones of 20 x 20, skip_row / skip_col = the pixel row / column
// that is updated but not counted (379 of the full image; -1 for a band that does not hold it).
__global__ __launch_bounds__(ZTHREADS) void mog2_run_kernel(const uint8_t* __restrict__ gobans, int nframes, int h, int w,
                                                            int skip_row, int skip_col,
                                                            float* __restrict__ gw_, float* __restrict__ gv_,
                                                            float* __restrict__ mean_, uint8_t* __restrict__ nmodes_,
                                                            const float2* __restrict__ rates, int32_t* __restrict__ fgcount,
                                                            uint8_t* __restrict__ last_fg)
{
    extern __shared__ int zcount[];                     // one counter per frame
    const int zcols = (w + ZONE - 1) / ZONE, zrows = (h + ZONE - 1) / ZONE, nzones = zrows * zcols;
    // XCD-aware zone order (round 4): workgroups go round-robin over the 8 XCDs by their id, so with zone = id the 19 zones
    // of a zone row -- whose 60-byte pixel segments share 128-byte lines -- sat on 8 different L2s and every line came from
    // HBM several times (PMC: 2.9 MB fetched per frame for 0.43 MB of pixels).  Here XCD k owns the zone rows k, k + 8,
    // k + 16: the grid is 8 x ceil(zrows / 8) x zcols workgroups, those beyond the last zone row leave at once.
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int cr = xcd + 8 * (j / zcols), cc = j % zcols, t = threadIdx.x;
    if (cr >= zrows) return;                             // (whole workgroup, before any barrier)
    const int zone = cr * zcols + cc;
    const int y = cr * ZONE + t / ZONE, x = cc * ZONE + t % ZONE;
    const bool live = t < ZONE * ZONE && y < h && x < w;
    const bool counted = live && y != skip_row && x != skip_col;
    for (int f = t; f < nframes; f += ZTHREADS) zcount[f] = 0;
    __syncthreads();
    const int npx = h * w, px = y * w + x;
    Mix m;
    if (live) mix_load(m, gw_, gv_, mean_, nmodes_, npx, px);
    bool background = true;
    // Frames in groups of GRP, software-pipelined: the byte loads of group g + 1 are in flight while the updates of
    // group g run back to back.  The pixel loads do not depend on the mixture state -- only the update is a chain.
    constexpr int GRP = 4;
    uint8_t cur[GRP][3], nxt[GRP][3];
    const uint8_t* base = gobans + (size_t)(live ? px : 0) * 3;
    const size_t fstride = (size_t)npx * 3;
#pragma unroll
    for (int j = 0; j < GRP; j++) {
        const uint8_t* q = base + (size_t)(j < nframes ? j : nframes - 1) * fstride;
#pragma unroll
        for (int c = 0; c < 3; c++) cur[j][c] = q[c];
    }
    for (int f0 = 0; f0 < nframes; f0 += GRP) {
#pragma unroll
        for (int j = 0; j < GRP; j++) {
            const int f = f0 + GRP + j < nframes ? f0 + GRP + j : nframes - 1;
            const uint8_t* q = base + (size_t)f * fstride;
#pragma unroll
            for (int c = 0; c < 3; c++) nxt[j][c] = q[c];
        }
#pragma unroll
        for (int j = 0; j < GRP; j++) {
            const int f = f0 + j;
            if (f < nframes) {                              // uniform across the workgroup
                const float2 r = rates[f];
                if (live) background = mix_update(m, (float)cur[j][0], (float)cur[j][1], (float)cur[j][2], r.x, r.y);
                const unsigned long long fgmask = __ballot(counted && !background);
                if ((t & 63) == 0 && fgmask) atomicAdd(&zcount[f], __popcll(fgmask));
            }
        }
#pragma unroll
        for (int j = 0; j < GRP; j++)
#pragma unroll
            for (int c = 0; c < 3; c++) cur[j][c] = nxt[j][c];
    }
    if (live) {
        mix_store(m, gw_, gv_, mean_, nmodes_, npx, px);
        if (last_fg) last_fg[px] = background ? 0 : 255;
    }
    __syncthreads();
    for (int f = t; f < nframes; f += ZTHREADS) fgcount[(size_t)f * nzones + zone] = zcount[f];
}

// box sums of a foreground mask over the 361 zones (the per-frame finder's form of the same counts)
__global__ __launch_bounds__(ZTHREADS) void zone_count_kernel(const uint8_t* __restrict__ mask, int nframes, int side,
                                                              int32_t* __restrict__ fgcount)
{
    __shared__ int acc;
    const int cells = (side + ZONE - 1) / ZONE;
    const int f = blockIdx.y, cr = blockIdx.x / cells, cc = blockIdx.x % cells, t = threadIdx.x;
    const int y = cr * ZONE + t / ZONE, x = cc * ZONE + t % ZONE;
    const bool counted = t < ZONE * ZONE && y < side && x < side && !(cr == cells - 1 && y == side - 1) &&
                         !(cc == cells - 1 && x == side - 1);
    if (t == 0) acc = 0;
    __syncthreads();
    const bool on = counted && mask[(size_t)f * side * side + (size_t)y * side + x] != 0;
    const unsigned long long b = __ballot(on);
    if ((t & 63) == 0 && b) atomicAdd(&acc, __popcll(b));
    __syncthreads();
    if (t == 0) fgcount[(size_t)f * (cells * cells) + blockIdx.x] = acc;
}

}  // namespace

// learning rate -> (alphaT, prune) exactly as one apply() call derives them; advances the model's frame count
static void mog2_rate(Mog2State& st, double learning_rate, float* alphaT, float* prune)
{
    const int history = 500;
    ++st.nframes;
    const int lim = 2 * st.nframes < history ? 2 * st.nframes : history;
    const double lr = (learning_rate >= 0 && st.nframes > 1) ? learning_rate : 1. / lim;
    *alphaT = (float)lr;
    *prune = (float)(-lr * 0.05f);
}

int k_mog2_run(ck_ctx* ctx, Mog2State& st, const uint8_t* d_gobans, int n, const double* learning_rates,
               int32_t* d_fgcount, uint8_t* d_last_fg, int skip_row, int skip_col)
{
    TimeScope ts(ctx, "mog2");
    if (n > 8192) return ck_fail(ctx, CK_ERR_CAPACITY, "mog2 run: %d frames in one run (max 8192)", n);
    for (int f = 0; f < n; f++)                                     // validate BEFORE the model's frame count moves
        if (learning_rates[f] >= 1) return ck_fail(ctx, CK_ERR_ARG, "mog2 run: learning rate >= 1 (model reset) inside a run");
    // the rates travel through the model's OWN pinned + device buffers (not the context's shared scratch)
    const size_t rbytes = (size_t)n * 2 * sizeof(float);
    if (st.rates_cap < rbytes) {
        if (st.rates_host) CK_HIP(ctx, hipStreamSynchronize(ctx->stream));      // a queued run may still read the old ones
        if (st.rates_host) (void)hipHostFree(st.rates_host);
        if (st.rates_dev) (void)hipFree(st.rates_dev);
        st.rates_host = nullptr; st.rates_dev = nullptr; st.rates_cap = 0;
        const size_t cap = rbytes < 4096 ? 4096 : rbytes * 2;
        CK_HIP(ctx, hipHostMalloc((void**)&st.rates_host, cap, hipHostMallocDefault));
        CK_HIP(ctx, hipMalloc((void**)&st.rates_dev, cap));
        st.rates_cap = cap;
    } else if (st.rates_busy) {
        CK_HIP(ctx, hipEventSynchronize(st.rates_done));                        // previous upload has left the host buffer
    }
    if (!st.rates_done) CK_HIP(ctx, hipEventCreateWithFlags(&st.rates_done, hipEventDisableTiming));
    for (int f = 0; f < n; f++) mog2_rate(st, learning_rates[f], &st.rates_host[2 * (size_t)f], &st.rates_host[2 * (size_t)f + 1]);
    CK_HIP(ctx, hipMemcpyAsync(st.rates_dev, st.rates_host, rbytes, hipMemcpyHostToDevice, ctx->stream));
    CK_HIP(ctx, hipEventRecord(st.rates_done, ctx->stream));
    st.rates_busy = true;
    const int zrows = (st.h + ZONE - 1) / ZONE, zcols = (st.w + ZONE - 1) / ZONE;
    const int blocks = 8 * ((zrows + 7) / 8) * zcols;               // XCD k <- zone rows k, k + 8, ... (see the kernel)
    hipLaunchKernelGGL(mog2_run_kernel, dim3(blocks), dim3(ZTHREADS), (size_t)n * sizeof(int), ctx->stream,
                       d_gobans, n, st.h, st.w, skip_row, skip_col, (float*)st.weight.p, (float*)st.variance.p, (float*)st.mean.p,
                       (uint8_t*)st.nmodes.p, (const float2*)st.rates_dev, d_fgcount, d_last_fg);
    CK_HIP(ctx, hipGetLastError());
    return CK_OK;
}

int k_zone_counts(ck_ctx* ctx, const uint8_t* d_mask, int n, int side, int32_t* d_fgcount)
{
    const int cells = (side + ZONE - 1) / ZONE;
    hipLaunchKernelGGL(zone_count_kernel, dim3(cells * cells, n), dim3(ZTHREADS), 0, ctx->stream, d_mask, n, side, d_fgcount);
    CK_HIP(ctx, hipGetLastError());
    return CK_OK;
}

int k_mog2_apply(ck_ctx* ctx, Mog2State& st, const uint8_t* d_img, double learning_rate, uint8_t* d_fg)
{
    TimeScope ts(ctx, "mog2");
    const int npx = st.h * st.w;
    if (learning_rate >= 1) {
        CK_HIP(ctx, hipMemsetAsync(st.nmodes.p, 0, (size_t)npx, ctx->stream));
        st.nframes = 0;
    }
    float alphaT, prune;
    mog2_rate(st, learning_rate, &alphaT, &prune);
    hipLaunchKernelGGL(mog2_kernel, dim3((npx + 255) / 256), dim3(256), 0, ctx->stream, d_img, npx,
                       (float*)st.weight.p, (float*)st.variance.p, (float*)st.mean.p, (uint8_t*)st.nmodes.p,
                       alphaT, prune, d_fg);
    CK_HIP(ctx, hipGetLastError());
    return CK_OK;
}
