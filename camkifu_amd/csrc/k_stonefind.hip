// k_stonefind.hip -- SfContours.find_stones for a batch of goban images
// (reference: src/camkifu/stone/sf_contours.py:48-111, with analyse_fg :207-249, extract_contours_fg :251-300,
//  _filter_contours :186-205, _find_centers :302-330, find_color :128-184).
//
// What runs where.  Dense pixel work on the GPU, batched over the n goban images of the call:
//   A  the opening of the foreground mask the reference asks for -- cv2.morphologyEx(fg, MORPH_OPEN, (5, 5),
//      iterations=3): the binding reads the tuple as a 2x1 element, three iterations fold into one 4x1 element anchored
//      at its last row, so both passes look at rows y-3 .. y (rows outside the view are ignored) -- fused with the
//      crop to the analysed sub-image and written as three equal planes for K2's Canny kernel (25 / 75);
//   B  get_canny on the cropped image (ck_goban_canny_dev: medians 13 + 7, Otsu, Canny);
//   C  the external contours of both edge maps in ONE labelling pass over 2n maps (k_contour_survey), with the
//      border follower that counts CHAIN_APPROX_SIMPLE vertices;
//   E  the hull mask (row spans painted into a byte image) and, per intersection zone, the number of mask pixels and
//      the channel sums under and outside the mask -- one wave per zone.
// Per-contour geometry on the host between C and E (D): hull, float rotating calipers, the filters of the reference in
// their order, the hull raster and fill ratio, the chamfer distance of the few foreground candidates
// (ck_stonegeom.cpp); and after E the zone means and find_color, whose raster order is part of the result.
#include <math.h>

#include <algorithm>
#include <chrono>
#include <thread>

#include "ck_common.h"
#include "ck_stonegeom.h"

namespace {

constexpr int GS = 19;

// A: crop + opening.  One thread per pixel of the sub-image; at most 7 rows of the mask are read.
__global__ __launch_bounds__(256) void fg_open_kernel(const uint8_t* __restrict__ fg, int side, int x0, int y0, int hs, int ws,
                                                      uint8_t* __restrict__ planes, int pitch, uint8_t* __restrict__ crop)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), f = blockIdx.z;
    if (x >= ws || y >= hs) return;
    const uint8_t* src = fg + ((size_t)f * side + x0) * side + y0 + x;      // (x0, y0) = first row / first column of the view
    int v[7];
#pragma unroll
    for (int k = 0; k < 7; k++) v[k] = y - k >= 0 ? src[(size_t)(y - k) * side] : 255;    // 255: ignored by the erosion
    int out = 0;
#pragma unroll
    for (int a = 0; a < 4; a++) {
        if (y - a < 0) break;                                                // rows above the view: ignored by the dilation
        out = max(out, min(min(v[a], v[a + 1]), min(v[a + 2], v[a + 3])));
    }
    uint8_t* dst = planes + ((size_t)f * 3 * hs + y) * pitch + x;
    dst[0] = (uint8_t)out;
    dst[(size_t)hs * pitch] = (uint8_t)out;
    dst[(size_t)2 * hs * pitch] = (uint8_t)out;
    crop[((size_t)f * hs + y) * ws + x] = (uint8_t)v[0];
}

__global__ __launch_bounds__(256) void crop_bgr_kernel(const uint8_t* __restrict__ img, int side, int x0, int y0, int hs, int ws,
                                                       uint8_t* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, f = blockIdx.z;
    if (i >= ws * 3) return;
    out[((size_t)f * hs + y) * ws * 3 + i] = img[(((size_t)f * side + x0 + y) * side + y0) * 3 + i];
}

// E1: paint row spans (frame, y, xa, xb inclusive); 16 lanes per span
__global__ __launch_bounds__(256) void paint_spans_kernel(const int32_t* __restrict__ spans, int nspans, int hs, int ws,
                                                          uint8_t* __restrict__ mask)
{
    const int s = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
    if (s >= nspans) return;
    const int f = spans[4 * (size_t)s], y = spans[4 * (size_t)s + 1], xa = spans[4 * (size_t)s + 2], xb = spans[4 * (size_t)s + 3];
    uint8_t* row = mask + ((size_t)f * hs + y) * ws;
    for (int x = xa + l; x <= xb; x += 16) row[x] = 1;
}

// E2: one wave per (zone, frame): mask pixels, channel sums under the mask, channel sums of the whole zone
__global__ __launch_bounds__(64) void zone_sums_kernel(const uint8_t* __restrict__ img /* n*hs*ws*3 */, const uint8_t* __restrict__ mask,
                                                       int hs, int ws, const int32_t* __restrict__ rects /* nz * 4: a0, b0, a1, b1 in the view */,
                                                       int32_t* __restrict__ sums /* n * nz * 7 */)
{
    const int z = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
    const int a0 = rects[4 * z], b0 = rects[4 * z + 1], a1 = rects[4 * z + 2], b1 = rects[4 * z + 3];
    const int zw = b1 - b0, area = (a1 - a0) * zw;
    int acc[7] = { 0, 0, 0, 0, 0, 0, 0 };
    for (int p = lane; p < area; p += 64) {
        const int y = a0 + p / zw, x = b0 + p % zw;
        const size_t o = ((size_t)f * hs + y) * ws + x;
        const int m = mask[o];
        const uint8_t* px = img + o * 3;
        acc[0] += m;
#pragma unroll
        for (int k = 0; k < 3; k++) { acc[1 + k] += m ? px[k] : 0; acc[4 + k] += px[k]; }
    }
#pragma unroll
    for (int k = 0; k < 7; k++)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc[k] += __shfl_xor(acc[k], d);
    if (lane == 0) {
        int32_t* o = sums + ((size_t)f * gridDim.x + z) * 7;
#pragma unroll
        for (int k = 0; k < 7; k++) o[k] = acc[k];
    }
}

template <typename F>
void frames_parallel(int n, F fn) { ck_parallel_for(n, 16, fn); }

struct Box { float w, h, angle; };

// minAreaRect of a contour from its hull (the hull of a hull is itself: the big sort runs once per contour)
Box box_of(const std::vector<int32_t>& hull)
{
    float wha[3];
    ck_min_area_rect_box(hull.data(), (int)(hull.size() / 2), wha);
    return { wha[0], wha[1], wha[2] };
}

// D, foreground side: extract_contours_fg's filters, then analyse_fg's "closed enough" test.  Returns -1 where the
// reference would raise (ZeroDivisionError in _find_centers).
int fg_contours(const std::vector<CkContour>& found, const uint8_t* sub_fg, int hs, int ws, double radius,
                std::vector<const CkContour*>& kept)
{
    std::vector<const CkContour*> cands;
    std::vector<uint8_t> bits;
    for (const CkContour& c : found) {
        if (c.nvert < 10) continue;                                   // too few points to describe a stone
        const std::vector<int32_t> hull = ck_hull_points(c.pts.data(), (int)(c.pts.size() / 2));
        const Box b = box_of(hull);
        const double lo = std::min(b.w, b.h), hi = std::max(b.w, b.h);
        if (lo < 3.0 / 2 * radius) continue;                          // one side too small
        if (5 * radius < hi) continue;                                // two stones at most
        const double angle = (double)b.angle * (3.14159265358979323846 / 180.0);          // math.radians
        if (2.5 * radius < hi && std::max(std::fabs(std::cos(angle)), std::fabs(std::sin(angle))) < 0.97) continue;
        int bx, by, bw, bh;
        ck_raster_polygon(hull.data(), (int)(hull.size() / 2), &bx, &by, &bw, &bh, bits);
        long long sum = 0;
        for (int y = 0; y < bh; y++)
            for (int x = 0; x < bw; x++)
                if (bits[(size_t)y * bw + x]) sum += sub_fg[(size_t)(by + y) * ws + bx + x];
        const double ratio = (double)sum / bh / bw / 255;
        if (ratio < 0.3) continue;                                    // interior too black
        cands.push_back(&c);
    }
    if (cands.empty()) return 0;
    std::vector<uint8_t> ghost((size_t)hs * ws, 0), negative;
    std::vector<int32_t> dist;
    for (const CkContour* c : cands) {
        int x0 = 1 << 30, x1 = -1, y0 = 1 << 30, y1 = -1;
        for (size_t i = 0; i + 1 < c->pts.size(); i += 2) {
            const int x = c->pts[i], y = c->pts[i + 1];
            ghost[(size_t)y * ws + x] = 255;                          // outlines accumulate from one contour to the next
            x0 = std::min(x0, x); x1 = std::max(x1, x); y0 = std::min(y0, y); y1 = std::max(y1, y);
        }
        const int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
        negative.resize((size_t)bw * bh);
        for (int y = 0; y < bh; y++)
            for (int x = 0; x < bw; x++) negative[(size_t)y * bw + x] = (uint8_t)(255 - ghost[(size_t)(y0 + y) * ws + x0 + x]);
        ck_chamfer5(negative.data(), bh, bw, dist);
        const int got = ck_has_stone_center(dist.data(), bh, bw, radius);
        if (got < 0) return -1;
        if (got) kept.push_back(c);
    }
    return 0;
}

}  // namespace

int k_contour_stones(ck_ctx* ctx, const uint8_t* d_goban, const uint8_t* d_fg, int n, int side, const int32_t* rects,
                     int rs, int re, int cs, int ce, uint8_t* stones, int16_t* zones_out, uint8_t* mask_out)
{
    static const bool prof = getenv("CK_PROFILE_HOST") != nullptr;   // debugging aid: host-side lap times on stderr
    auto t_start = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!prof) return;
        (void)hipStreamSynchronize(ctx->stream);
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[contour_stones] %-18s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_start).count());
        t_start = now;
    };
    const int R = re - rs, C = ce - cs, nz = R * C;
    const int x0 = rects[((size_t)rs * GS + cs) * 4], y0 = rects[((size_t)rs * GS + cs) * 4 + 1];
    const int x1 = rects[((size_t)(re - 1) * GS + ce - 1) * 4 + 2], y1 = rects[((size_t)(re - 1) * GS + ce - 1) * 4 + 3];
    const int hs = x1 - x0, ws = y1 - y0;                 // the reference's x runs along rows
    if (x0 < 0 || y0 < 0 || x1 > side || y1 > side || hs < 16 || ws < 16)
        return ck_fail(ctx, CK_ERR_ARG, "zone rectangles give a %d x %d view at (%d, %d) of a %d image", hs, ws, x0, y0, side);
    std::vector<int32_t> zr((size_t)nz * 4);
    for (int r = 0; r < R; r++)
        for (int c = 0; c < C; c++) {
            const int32_t* q = rects + ((size_t)(r + rs) * GS + c + cs) * 4;
            int32_t* o = zr.data() + ((size_t)r * C + c) * 4;
            o[0] = q[0] - x0; o[1] = q[1] - y0; o[2] = q[2] - x0; o[3] = q[3] - y0;
            if (o[0] < 0 || o[1] < 0 || o[2] > hs || o[3] > ws || o[2] <= o[0] || o[3] <= o[1])
                return ck_fail(ctx, CK_ERR_ARG, "zone (%d, %d) does not lie inside the analysed view", r + rs, c + cs);
        }
    const double radius = (double)side / GS / 2;          // StonesFinder.stone_radius (stonesfinder.py:578-584)
    const size_t spx = (size_t)hs * ws, npx = (size_t)n * spx;
    const int pitch = ck_pitch(ws);

    // ---- A, B: the two edge maps of every image, side by side in one buffer -------------------------------------
    CK_TRY(ck_ensure(ctx, ctx->edges, 2 * npx + 64));
    CK_TRY(ck_ensure(ctx, ctx->goban, npx * 3 + npx + 2 * npx + 64));     // cropped BGR, cropped fg, mask (+ slack)
    CK_TRY(ck_ensure(ctx, ctx->planes, (size_t)n * 3 * hs * pitch));
    CK_TRY(ck_ensure(ctx, ctx->map, npx));
    CK_TRY(ck_ensure(ctx, ctx->labels, 2 * npx * 4));
    uint8_t* d_edges = (uint8_t*)ctx->edges.p;
    uint8_t* d_sub = (uint8_t*)ctx->goban.p;
    uint8_t* d_subfg = d_sub + npx * 3;
    uint8_t* d_mask = d_subfg + ((npx + 15) & ~(size_t)15);
    {
        TimeScope ts(ctx, "stonefind_open");
        hipLaunchKernelGGL(fg_open_kernel, dim3((ws + 63) / 64, (hs + 3) / 4, n), dim3(256), 0, ctx->stream, d_fg, side, x0, y0, hs, ws,
                           (uint8_t*)ctx->planes.p, pitch, d_subfg);
        hipLaunchKernelGGL(crop_bgr_kernel, dim3((ws * 3 + 255) / 256, hs, n), dim3(256), 0, ctx->stream, d_goban, side, x0, y0, hs, ws, d_sub);
        CK_HIP(ctx, hipGetLastError());
    }
    CK_TRY(ck_ensure_pinned(ctx, npx));
    const uint8_t* h_fg = (const uint8_t*)ctx->host_pinned;
    CK_HIP(ctx, hipMemcpyAsync(ctx->host_pinned, d_subfg, npx, hipMemcpyDeviceToHost, ctx->stream));
    CK_TRY(k_canny_planar(ctx, (const uint8_t*)ctx->planes.p, n, hs, ws, pitch, 25, 75, (uint8_t*)ctx->map.p,
                          (int32_t*)ctx->labels.p, d_edges, nullptr));
    lap("open + fg canny");
    CK_TRY(ck_goban_canny_dev(ctx, d_sub, n, hs, ws, d_edges + npx, nullptr));
    lap("get_canny");

    // ---- C: external contours of the 2n maps -------------------------------------------------------------------------
    std::vector<std::vector<CkContour>> found;
    CK_TRY(k_contour_survey(ctx, d_edges, 2 * n, hs, ws, found));
    lap("contour survey");

    // ---- D: which hulls make the mask -----------------------------------------------------------------------------
    std::vector<std::vector<int32_t>> spans((size_t)n);
    std::vector<int> bad((size_t)n, 0);
    frames_parallel(n, [&](int f) {
        std::vector<const CkContour*> kept;
        if (fg_contours(found[f], h_fg + (size_t)f * spx, hs, ws, radius, kept) < 0) { bad[f] = 1; return; }
        std::vector<std::vector<int32_t>> hulls;
        for (const CkContour* c : kept) hulls.push_back(ck_hull_points(c->pts.data(), (int)(c->pts.size() / 2)));
        for (const CkContour& c : found[(size_t)n + f]) {                    // _filter_contours
            if (c.nvert < 10) continue;
            std::vector<int32_t> hull = ck_hull_points(c.pts.data(), (int)(c.pts.size() / 2));
            const Box b = box_of(hull);
            if (10 * radius < std::max(b.w, b.h)) continue;
            hulls.push_back(std::move(hull));
        }
        std::vector<uint8_t> bits;
        auto& sp = spans[f];
        for (const std::vector<int32_t>& hull : hulls) {
            int bx, by, bw, bh;
            ck_raster_polygon(hull.data(), (int)(hull.size() / 2), &bx, &by, &bw, &bh, bits);
            for (int y = 0; y < bh; y++)
                for (int x = 0; x < bw;) {
                    if (!bits[(size_t)y * bw + x]) { x++; continue; }
                    int e = x;
                    while (e + 1 < bw && bits[(size_t)y * bw + e + 1]) e++;
                    sp.push_back(f); sp.push_back(by + y); sp.push_back(bx + x); sp.push_back(bx + e);
                    x = e + 1;
                }
        }
    });
    lap("host geometry");
    for (int f = 0; f < n; f++)
        if (bad[f]) return ck_fail(ctx, CK_ERR_STATE, "image %d: a foreground contour thinner than a stone radius reached _find_centers "
                                                      "(the reference divides by zero there)", f);
    std::vector<int32_t> all;
    for (int f = 0; f < n; f++) all.insert(all.end(), spans[f].begin(), spans[f].end());
    const int nspans = (int)(all.size() / 4);

    // ---- E: mask and zone sums -----------------------------------------------------------------------------------
    CK_TRY(ck_ensure(ctx, ctx->pts, all.size() * 4 + (size_t)nz * 16 + (size_t)n * nz * 28 + 256));
    int32_t* d_spans = (int32_t*)ctx->pts.p;
    int32_t* d_rects = d_spans + all.size();
    int32_t* d_sums = d_rects + (size_t)nz * 4;
    std::vector<int32_t> sums((size_t)n * nz * 7);
    {
        TimeScope ts(ctx, "stonefind_zones");
        CK_HIP(ctx, hipMemsetAsync(d_mask, 0, npx, ctx->stream));
        if (nspans) {
            CK_HIP(ctx, hipMemcpyAsync(d_spans, all.data(), all.size() * 4, hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(paint_spans_kernel, dim3((nspans + 15) / 16), dim3(256), 0, ctx->stream, (const int32_t*)d_spans, nspans, hs, ws, d_mask);
        }
        CK_HIP(ctx, hipMemcpyAsync(d_rects, zr.data(), zr.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(zone_sums_kernel, dim3(nz, n), dim3(64), 0, ctx->stream, (const uint8_t*)d_sub, (const uint8_t*)d_mask, hs, ws,
                           (const int32_t*)d_rects, d_sums);
        CK_HIP(ctx, hipGetLastError());
    }
    CK_HIP(ctx, hipMemcpyAsync(sums.data(), d_sums, sums.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (mask_out) CK_HIP(ctx, hipMemcpyAsync(mask_out, d_mask, npx, hipMemcpyDeviceToHost, ctx->stream));
    CK_HIP(ctx, hipStreamSynchronize(ctx->stream));

    lap("mask + zone sums");
    // ---- zone means (int16, truncated as numpy stores a float into an int16 slot) and colours, in raster order -------
    std::vector<int16_t> zones((size_t)nz * 4);
    for (int f = 0; f < n; f++) {
        for (int z = 0; z < nz; z++) {
            const int32_t* s = sums.data() + ((size_t)f * nz + z) * 7;
            const int32_t* q = zr.data() + (size_t)z * 4;
            const int area = (q[2] - q[0]) * (q[3] - q[1]), visible = s[0];
            int16_t* o = zones.data() + (size_t)z * 4;
            if (0.4 * area < visible) {                   // a zone is masked if more than 60% of its pixels are
                o[0] = 1;
                for (int k = 0; k < 3; k++) o[1 + k] = (int16_t)((double)s[1 + k] / (double)visible);
            } else {
                o[0] = 0;
                for (int k = 0; k < 3; k++) o[1 + k] = (int16_t)((double)(s[4 + k] - s[1 + k]) / (double)(area - visible));
            }
        }
        uint8_t* st = stones + (size_t)f * GS * GS;
        memset(st, 0, GS * GS);
        ck_find_colors(zones.data(), R, C, st + (size_t)rs * GS + cs, GS);
        if (zones_out) memcpy(zones_out + (size_t)f * nz * 4, zones.data(), zones.size() * 2);
    }
    lap("colours");
    return CK_OK;
}
