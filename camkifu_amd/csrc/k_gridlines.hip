// k_gridlines.hip -- StonesFinder.find_intersections for a batch of goban images
// (reference: src/camkifu/stone/stonesfinder.py:516-552; update_grid :888-947 runs on the host, ck_stonegeom.cpp).
//
//   gray = cvtColor(img, BGR2GRAY); level = Otsu(gray); canny = Canny(gray, level / 2, level)
//   for each of the 361 intersection zones:  HoughLinesP(zone, 1, pi / 180, 3/4 side, minLineLength 2/3 side, maxLineGap 0)
//
// A  one kernel turns the interleaved image into three equal grey planes (K2's Canny kernel takes planes) and the
//    grey histogram (LDS bins); the Otsu level is the library's double-precision scan on the host, one round trip
//    for the whole batch; Canny is K2's kernel with per-image thresholds.
// B  the progressive probabilistic Hough transform is serial in its points (each accepted line removes pixels the
//    next draw may have hit) but the 361 x n zones are independent and tiny: ONE WAVE PER ZONE.  The 180 x numrho
//    accumulator of a zone (signed 8-bit counters) lives in LDS, the lanes share the 180 angles of every vote and
//    un-vote, the random draw (cv::RNG, seeded the same for every call of the library function), the fixed-point walk
//    along the winning line and the bookkeeping are wave-uniform.
#include <math.h>

#include "ck_common.h"
#include "ck_stonegeom.h"

namespace {

constexpr int GS = 19;
constexpr int ZMAX = 40;                 // largest zone side the LDS layout is sized for
constexpr int NANG = 180;

__global__ __launch_bounds__(256) void gray_planes_hist_kernel(const uint8_t* __restrict__ bgr, int h, int w, int pitch,
                                                               uint8_t* __restrict__ planes, int* __restrict__ hist)
{
    __shared__ int bins[256];
    const int f = blockIdx.y;
    bins[threadIdx.x] = 0;
    __syncthreads();
    const int npx = h * w;
    const uint8_t* src = bgr + (size_t)f * npx * 3;
    uint8_t* dst = planes + (size_t)f * 3 * h * pitch;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npx; p += gridDim.x * 256) {
        const int y = p / w, x = p - y * w;
        const int g = (1868 * src[3 * (size_t)p] + 9617 * src[3 * (size_t)p + 1] + 4899 * src[3 * (size_t)p + 2] + (1 << 13)) >> 14;
        const size_t o = (size_t)y * pitch + x;
        dst[o] = (uint8_t)g;
        dst[o + (size_t)h * pitch] = (uint8_t)g;
        dst[o + (size_t)2 * h * pitch] = (uint8_t)g;
        atomicAdd(&bins[g], 1);
    }
    __syncthreads();
    if (bins[threadIdx.x]) atomicAdd(&hist[f * 256 + threadIdx.x], bins[threadIdx.x]);
}

__device__ __forceinline__ int wave_max_first(int val, int n)
{
    // largest val, smallest n among equals (the serial scan keeps the first angle that reaches the maximum): one key
    int key = (val << 8) | (255 - n);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) key = max(key, __shfl_xor(key, d));
    return (key & ~255) | (255 - (key & 255));
}

// B: HoughLinesP of one zone per wave.  lines: up to CK_ZONE_LINES (x0, y0, x1, y1) int16 per zone, in the order found.
// A counter stays within +- the number of zone pixels in one line's band (< 64 for ZMAX = 40; it goes NEGATIVE where a
// kept line takes back the votes of pixels that had not been drawn yet, as in the library): signed 8-bit counters, so
// several zones share a CU's LDS.  The walk along the winning line is closed-form (step t of direction k is start + t * delta): every
// lane tests one step, a ballot finds where the line ends -- no chain of dependent LDS reads.
__global__ __launch_bounds__(64) void hough_zones_kernel(const uint8_t* __restrict__ edges, int side, const int32_t* __restrict__ rects,
                                                         const float* __restrict__ trig /* cos[180], sin[180] */, int numrho_max, int zpx /* largest zone, pixels */,
                                                         int16_t* __restrict__ lines, int32_t* __restrict__ nlines, int* __restrict__ overflow)
{
    extern __shared__ int8_t acc[];                            // NANG x numrho counters, then the mask and the point list
    const int z = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
    const int x0 = rects[4 * z], y0 = rects[4 * z + 1], x1 = rects[4 * z + 2], y1 = rects[4 * z + 3];
    const int height = x1 - x0, width = y1 - y0;              // the reference's x runs along rows
    const int numrho = (width + height) * 2 + 1, half = (numrho - 1) / 2;
    uint8_t* mask = reinterpret_cast<uint8_t*>(acc) + ((NANG * numrho_max + 3) & ~3);
    uint16_t* nzloc = reinterpret_cast<uint16_t*>(mask + zpx);
    const int min_side = min(height, width);
    const int threshold = (int)(min_side * 3 / 4.0), min_len = (int)(min_side * 2 / 3.0);     // int(min_side * 3 / 4), int(min_side * 2 / 3)
    {
        uint32_t* a4 = reinterpret_cast<uint32_t*>(acc);
        for (int i = lane; i < (NANG * numrho + 3) / 4; i += 64) a4[i] = 0;
    }
    // stage 1: the non-zero points in raster order
    const uint8_t* img = edges + ((size_t)f * side + x0) * side + y0;
    int count = 0;
    for (int base = 0; base < height * width; base += 64) {
        const int p = base + lane;
        bool on = false;
        if (p < height * width) {
            const int i = p / width, j = p - i * width;
            on = img[(size_t)i * side + j] != 0;
            mask[p] = on ? 1 : 0;
        }
        const unsigned long long b = __builtin_amdgcn_ballot_w64(on);
        if (on) nzloc[count + __builtin_popcountll(b & ((1ull << lane) - 1ull))] = (uint16_t)p;
        count += __builtin_popcountll(b);
    }
    __syncthreads();
    float ct[3], st[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int n = lane + 64 * k;
        ct[k] = n < NANG ? trig[n] : 0.f;
        st[k] = n < NANG ? trig[NANG + n] : 0.f;
    }
    unsigned long long state = 0xFFFFFFFFFFFFFFFFull;          // RNG rng((uint64)-1)
    int found = 0;
    for (; count > 0; count--) {
        state = (unsigned long long)(unsigned)state * 4164903690u + (state >> 32);
        const int idx = (int)((unsigned)state % (unsigned)count);
        const int p = nzloc[idx], last = nzloc[count - 1];
        const bool live = mask[p] != 0;
        __syncthreads();
        if (lane == 0) nzloc[idx] = (uint16_t)last;            // "remove" the point by overriding it with the last one
        if (!live) continue;
        const int i = p / width, j = p - i * width;
        int best = threshold - 1, best_n = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int n = lane + 64 * k;
            if (n < NANG) {
                const int r = __float2int_rn((float)j * ct[k] + (float)i * st[k]) + half;
                const int val = ++acc[n * numrho + r];
                if (best < val) { best = val; best_n = n; }
            }
        }
        if (!__builtin_amdgcn_ballot_w64(best >= threshold)) continue;      // too weak a candidate (the usual case): next point
        const int max_n = wave_max_first(best, best_n) & 255;
        // the winning line's direction, from the lane that holds its angle
        const int src = max_n & 63, kk = max_n >> 6;
        const float cs_n = __shfl(kk == 0 ? ct[0] : (kk == 1 ? ct[1] : ct[2]), src);
        const float sn_n = __shfl(kk == 0 ? st[0] : (kk == 1 ? st[1] : st[2]), src);
        const float a = -sn_n, b = cs_n;
        int xs = j, ys = i, dx0, dy0;
        const bool xflag = fabsf(a) > fabsf(b);
        if (xflag) {
            dx0 = a > 0 ? 1 : -1;
            dy0 = __float2int_rn(b * 65536.f / fabsf(a));
            ys = (ys << 16) + (1 << 15);
        } else {
            dy0 = b > 0 ? 1 : -1;
            dx0 = __float2int_rn(a * 65536.f / fabsf(b));
            xs = (xs << 16) + (1 << 15);
        }
        // walk both ways over the remaining points: lane t looks at step t; maxLineGap = 0, so the first empty pixel
        // (or the zone border) ends the line.  Step 0 is the point itself.
        int qk[2], endt[2], ex[2], ey[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int x = xs + lane * (k ? -dx0 : dx0), y = ys + lane * (k ? -dy0 : dy0);
            const int j1 = xflag ? x : x >> 16, i1 = xflag ? y >> 16 : y;
            const bool inb = j1 >= 0 && j1 < width && i1 >= 0 && i1 < height;
            qk[k] = i1 * width + j1;
            const bool ok = inb && mask[inb ? qk[k] : 0] != 0;
            const unsigned long long stop = ~__builtin_amdgcn_ballot_w64(ok);      // a zone side is < 64: some lane always stops
            endt[k] = __builtin_ctzll(stop) - 1;
            ex[k] = __shfl(j1, endt[k]);
            ey[k] = __shfl(i1, endt[k]);
        }
        const bool good = abs(ex[1] - ex[0]) >= min_len || abs(ey[1] - ey[0]) >= min_len;
        if (good) {
            // take the votes of the line's points back (the point itself once)
            for (int k = 0; k < 2; k++)
                for (int t = k; t <= endt[k]; t++) {
                    const int q = __shfl(qk[k], t);
                    const int i1 = q / width, j1 = q - i1 * width;
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const int n = lane + 64 * c;
                        if (n < NANG) acc[n * numrho + __float2int_rn((float)j1 * ct[c] + (float)i1 * st[c]) + half]--;
                    }
                }
        }
        __syncthreads();
        if (lane <= endt[0]) mask[qk[0]] = 0;
        if (lane <= endt[1]) mask[qk[1]] = 0;
        __syncthreads();
        if (good) {
            if (found < CK_ZONE_LINES) {
                if (lane == 0) {
                    int16_t* o = lines + (((size_t)f * gridDim.x + z) * CK_ZONE_LINES + found) * 4;
                    o[0] = (int16_t)ex[0]; o[1] = (int16_t)ey[0]; o[2] = (int16_t)ex[1]; o[3] = (int16_t)ey[1];
                }
            } else if (lane == 0) *overflow = 1;
            found++;
        }
    }
    if (lane == 0) nlines[(size_t)f * gridDim.x + z] = found;
}

}  // namespace

int k_grid_lines(ck_ctx* ctx, const uint8_t* d_goban, int n, int side, const int32_t* rects, const int16_t** lines_out,
                 const int32_t** nlines_out, uint8_t* edges_out)
{
    const int nz = GS * GS;
    int zmax = 0;
    for (int z = 0; z < nz; z++) {
        const int32_t* q = rects + 4 * z;
        if (q[0] < 0 || q[1] < 0 || q[2] > side || q[3] > side || q[2] <= q[0] || q[3] <= q[1])
            return ck_fail(ctx, CK_ERR_ARG, "zone %d: rectangle (%d, %d, %d, %d) outside the %d image", z, q[0], q[1], q[2], q[3], side);
        if (std::min(q[2] - q[0], q[3] - q[1]) < 4)
            return ck_fail(ctx, CK_ERR_ARG, "zone %d: %d x %d pixels, at least 4 x 4", z, q[2] - q[0], q[3] - q[1]);
        zmax = std::max(zmax, std::max(q[2] - q[0], q[3] - q[1]));
    }
    if (zmax > ZMAX) return ck_fail(ctx, CK_ERR_ARG, "intersection zone of %d pixels: at most %d", zmax, ZMAX);
    const size_t fpx = (size_t)side * side, npx = fpx * n;
    const int pitch = ck_pitch(side);
    CK_TRY(ck_ensure(ctx, ctx->planes, (size_t)n * 3 * side * pitch));
    CK_TRY(ck_ensure(ctx, ctx->misc, (size_t)n * 256 * 4 + 4096));
    CK_TRY(ck_ensure(ctx, ctx->map, npx));
    CK_TRY(ck_ensure(ctx, ctx->labels, npx * 4));
    CK_TRY(ck_ensure(ctx, ctx->edges, npx));
    int* d_hist = (int*)ctx->misc.p;
    {
        TimeScope ts(ctx, "grid_gray");
        CK_HIP(ctx, hipMemsetAsync(d_hist, 0, (size_t)n * 256 * 4, ctx->stream));
        hipLaunchKernelGGL(gray_planes_hist_kernel, dim3(64, n), dim3(256), 0, ctx->stream, d_goban, side, side, pitch,
                           (uint8_t*)ctx->planes.p, d_hist);
        CK_HIP(ctx, hipGetLastError());
    }
    std::vector<int> hist((size_t)n * 256), thr((size_t)n * 2);
    CK_HIP(ctx, hipMemcpyAsync(hist.data(), d_hist, hist.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    CK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int f = 0; f < n; f++) {
        const double level = ck_otsu_level(&hist[(size_t)f * 256], fpx);
        thr[2 * f] = (int)std::floor(level / 2);
        thr[2 * f + 1] = (int)std::floor(level);
    }
    // small uploads share one buffer: thresholds, zone rectangles, trig table
    std::vector<float> trig(2 * NANG);
    {
        const float theta = (float)(3.1415926535897932384626433832795 / 180);
        for (int k = 0; k < NANG; k++) {
            trig[k] = (float)cos((double)k * theta);
            trig[NANG + k] = (float)sin((double)k * theta);
        }
    }
    const size_t small = (size_t)n * 8 + (size_t)nz * 16 + trig.size() * 4 + 64;
    CK_TRY(ck_ensure(ctx, ctx->mats, small));
    int* d_thr = (int*)ctx->mats.p;
    int32_t* d_rects = d_thr + (size_t)n * 2;
    float* d_trig = (float*)(d_rects + (size_t)nz * 4);
    int* d_over = (int*)(d_trig + trig.size());
    CK_HIP(ctx, hipMemcpyAsync(d_thr, thr.data(), thr.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    CK_HIP(ctx, hipMemcpyAsync(d_rects, rects, (size_t)nz * 16, hipMemcpyHostToDevice, ctx->stream));
    CK_HIP(ctx, hipMemcpyAsync(d_trig, trig.data(), trig.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    CK_HIP(ctx, hipMemsetAsync(d_over, 0, 4, ctx->stream));
    CK_HIP(ctx, hipStreamSynchronize(ctx->stream));           // thr / trig are locals: the copies must be done before they go
    CK_TRY(k_canny_planar(ctx, (const uint8_t*)ctx->planes.p, n, side, side, pitch, 0, 0, (uint8_t*)ctx->map.p,
                          (int32_t*)ctx->labels.p, (uint8_t*)ctx->edges.p, nullptr, nullptr, d_thr));
    const size_t line_bytes = (size_t)n * nz * CK_ZONE_LINES * 4 * sizeof(int16_t), cnt_bytes = (size_t)n * nz * 4;
    CK_TRY(ck_ensure(ctx, ctx->pts, line_bytes + cnt_bytes + 64));
    int16_t* d_lines = (int16_t*)ctx->pts.p;
    int32_t* d_nlines = (int32_t*)((char*)ctx->pts.p + line_bytes);
    {
        TimeScope ts(ctx, "grid_hough");
        const int numrho_max = 4 * zmax + 1;
        const int zpx = (zmax * zmax + 3) & ~3;
        const size_t lds = (((size_t)NANG * numrho_max + 3) & ~(size_t)3) + (size_t)zpx * 3;
        CK_HIP(ctx, hipFuncSetAttribute((const void*)hough_zones_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(hough_zones_kernel, dim3(nz, n), dim3(64), lds, ctx->stream, (const uint8_t*)ctx->edges.p, side,
                           (const int32_t*)d_rects, (const float*)d_trig, numrho_max, zpx, d_lines, d_nlines, d_over);
        CK_HIP(ctx, hipGetLastError());
    }
    int over = 0;
    CK_TRY(ck_ensure_pinned(ctx, line_bytes + cnt_bytes + 64));       // the tables are mostly empty: a pinned landing area, read in place
    int16_t* lines = (int16_t*)ctx->host_pinned;
    int32_t* nlines = (int32_t*)((char*)ctx->host_pinned + line_bytes);
    *lines_out = lines;
    *nlines_out = nlines;
    CK_HIP(ctx, hipMemcpyAsync(lines, d_lines, line_bytes + cnt_bytes, hipMemcpyDeviceToHost, ctx->stream));
    CK_HIP(ctx, hipMemcpyAsync(&over, d_over, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (edges_out) CK_HIP(ctx, hipMemcpyAsync(edges_out, ctx->edges.p, npx, hipMemcpyDeviceToHost, ctx->stream));
    CK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (over) return ck_fail(ctx, CK_ERR_CAPACITY, "a zone gave more than %d lines", CK_ZONE_LINES);
    return CK_OK;
}
