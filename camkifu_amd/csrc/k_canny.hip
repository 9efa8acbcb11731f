// k_canny.hip -- K2: cv2.Canny(median, 25, 75) on the 3-channel median image
// (reference: src/camkifu/board/bf_auto.py:73; aperture 3, L1 gradient).
//
//   nms kernel   : LDS-staged (TW+4)x(TH+4)x3 tile -> Sobel dx/dy per channel (replicate
//                  border) -> channel with the largest |dx|+|dy| -> non-maximum suppression
//                  with the fixed-point tan(22.5) sector test -> map {0 cand, 1 no, 2 strong}.
//                  HBM-bound: reads 3 B/px, writes 1 B/px (+4 B/px label init for candidates).
//   hysteresis   : the serial stack flood of the CPU algorithm becomes an 8-connected
//                  union-find over candidate pixels; a component is an edge iff it holds a
//                  strong pixel.  Same result set, no iteration-until-stable loop.
#include "ck_common.h"
#include "ck_uf.h"

namespace {

constexpr int TW = 64, TH = 16;
constexpr int LW = TW + 4, LH = TH + 4;     // pixel tile with 2-px halo
constexpr int MW = TW + 2, MH = TH + 2;     // magnitude tile with 1-px halo

__global__ __launch_bounds__(256) void canny_nms_kernel(const uint8_t* __restrict__ planes, int h, int w, int pitch,
                                                        int low, int high, uint8_t* __restrict__ map,
                                                        int32_t* __restrict__ labels, int32_t* __restrict__ cand,
                                                        int* __restrict__ cand_count)
{
    __shared__ uint8_t px[3][LH][LW + 4];
    __shared__ int32_t mg[MH][MW + 1];       // mag | sector << 16
    const int f = blockIdx.z;
    const int ox = blockIdx.x * TW, oy = blockIdx.y * TH;
    const int tid = threadIdx.x;
    const uint8_t* base = planes + (size_t)f * 3 * h * pitch;

    for (int i = tid; i < 3 * LH * LW; i += 256) {
        const int c = i / (LH * LW), r = (i / LW) % LH, col = i % LW;
        int y = oy - 2 + r, x = ox - 2 + col;
        y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
        x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
        px[c][r][col] = base[((size_t)c * h + y) * pitch + x];
    }
    __syncthreads();

    const int TG22 = 13573;   // (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5)
    for (int i = tid; i < MH * MW; i += 256) {
        const int r = i / MW, col = i % MW;
        const int y = oy - 1 + r, x = ox - 1 + col;
        int32_t packed = 0;                       // outside the image the magnitude is 0
        if (y >= 0 && y < h && x >= 0 && x < w) {
            // Sobel taps use replicate border relative to the IMAGE, so re-clamp here
            const int r0 = (y - 1 < 0 ? 0 : y - 1) - (oy - 2), r1 = y - (oy - 2), r2 = (y + 1 > h - 1 ? h - 1 : y + 1) - (oy - 2);
            const int c0 = (x - 1 < 0 ? 0 : x - 1) - (ox - 2), c1 = x - (ox - 2), c2 = (x + 1 > w - 1 ? w - 1 : x + 1) - (ox - 2);
            int best = -1, bdx = 0, bdy = 0;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int a00 = px[c][r0][c0], a01 = px[c][r0][c1], a02 = px[c][r0][c2];
                const int a10 = px[c][r1][c0], a12 = px[c][r1][c2];
                const int a20 = px[c][r2][c0], a21 = px[c][r2][c1], a22 = px[c][r2][c2];
                const int dx = (a02 + 2 * a12 + a22) - (a00 + 2 * a10 + a20);
                const int dy = (a20 + 2 * a21 + a22) - (a00 + 2 * a01 + a02);
                const int m = abs(dx) + abs(dy);
                if (m > best) { best = m; bdx = dx; bdy = dy; }
            }
            const int ax = abs(bdx), ay = abs(bdy) << 15;
            const int tg22x = ax * TG22;
            int sector;
            if (ay < tg22x) sector = 0;
            else if (ay > tg22x + (ax << 16)) sector = 1;
            else sector = ((bdx ^ bdy) < 0) ? 3 : 2;
            packed = best | (sector << 16);
        }
        mg[r][col] = packed;
    }
    __syncthreads();

    int* cnt = cand_count + f;
    int32_t* clist = cand + (size_t)f * h * w;
    for (int i = tid; i < TH * TW; i += 256) {          // TH*TW is a multiple of 256: wave-uniform trip count
        const int r = i / TW, col = i % TW;
        const int y = oy + r, x = ox + col;
        const bool inside = y < h && x < w;
        bool keep = false;
        int m = 0;
        if (inside) {
            const int32_t pk = mg[r + 1][col + 1];
            m = pk & 0xFFFF;
            const int sector = pk >> 16;
            if (m > low) {
                if (sector == 0) keep = m > (mg[r + 1][col] & 0xFFFF) && m >= (mg[r + 1][col + 2] & 0xFFFF);
                else if (sector == 1) keep = m > (mg[r][col + 1] & 0xFFFF) && m >= (mg[r + 2][col + 1] & 0xFFFF);
                else {
                    const int s = sector == 3 ? -1 : 1;
                    keep = m > (mg[r][col + 1 - s] & 0xFFFF) && m > (mg[r + 2][col + 1 + s] & 0xFFFF);
                }
            }
            const size_t idx = ((size_t)f * h + y) * w + x;
            map[idx] = keep ? (m > high ? 2 : 0) : 1;
            if (keep) labels[idx] = y * w + x;
        }
        const int slot = wave_append(cnt, keep);
        if (keep) clist[slot] = y * w + x;
    }
}

// The three hysteresis kernels walk the per-frame candidate list (a few % of the pixels)
// with a fixed grid and a grid-stride loop; the count is read from device memory.
constexpr int LIST_BLOCKS = 64;

// link every candidate with its already-scanned 8-neighbours (W, N, and NW / NE only when
// N is not itself a candidate -- otherwise the link is implied)
__global__ __launch_bounds__(256) void canny_link_kernel(const uint8_t* __restrict__ map, int h, int w,
                                                         int32_t* __restrict__ labels, const int32_t* __restrict__ cand,
                                                         const int* __restrict__ cand_count)
{
    const int f = blockIdx.y;
    const int n = cand_count[f];
    const uint8_t* m = map + (size_t)f * h * w;
    int32_t* L = labels + (size_t)f * h * w;
    const int32_t* C = cand + (size_t)f * h * w;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += LIST_BLOCKS * 256) {
        const int p = C[i];
        const int y = p / w, x = p - y * w;
        if (x > 0 && m[p - 1] != 1) uf_union(L, p, p - 1);
        if (y > 0) {
            if (m[p - w] != 1) uf_union(L, p, p - w);
            else {
                if (x > 0 && m[p - w - 1] != 1) uf_union(L, p, p - w - 1);
                if (x < w - 1 && m[p - w + 1] != 1) uf_union(L, p, p - w + 1);
            }
        }
    }
}

__global__ __launch_bounds__(256) void canny_flatten_mark_kernel(const uint8_t* __restrict__ map, int h, int w,
                                                                 int32_t* __restrict__ labels, uint8_t* __restrict__ edges,
                                                                 const int32_t* __restrict__ cand, const int* __restrict__ cand_count)
{
    const int f = blockIdx.y;
    const int n = cand_count[f];
    const size_t off = (size_t)f * h * w;
    const int32_t* C = cand + off;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += LIST_BLOCKS * 256) {
        const int p = C[i];
        const int root = uf_find(labels + off, p);
        labels[off + p] = root;
        if (map[off + p] == 2) edges[off + root] = 255;
    }
}

__global__ __launch_bounds__(256) void canny_final_kernel(int h, int w, const int32_t* __restrict__ labels,
                                                          uint8_t* __restrict__ edges, const int32_t* __restrict__ cand,
                                                          const int* __restrict__ cand_count)
{
    const int f = blockIdx.y;
    const int n = cand_count[f];
    const size_t off = (size_t)f * h * w;
    const int32_t* C = cand + off;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += LIST_BLOCKS * 256) {
        const int p = C[i];
        const int root = labels[off + p];
        if (root != p) edges[off + p] = edges[off + root];
    }
}

}  // namespace

int k_canny_planar(ck_ctx* ctx, const uint8_t* d_planes, int n, int h, int w, int pitch, int low, int high,
                   uint8_t* d_map, int32_t* d_labels, uint8_t* d_edges, uint8_t* d_map_out)
{
    if (low > high) { int t = low; low = high; high = t; }
    const size_t npx = (size_t)n * h * w;
    CK_TRY(ck_ensure(ctx, ctx->labels2, npx * 4));                 // candidate lists (one slab per frame)
    CK_TRY(ck_ensure(ctx, ctx->misc, (size_t)n * 64 + 4096));
    int32_t* d_cand = (int32_t*)ctx->labels2.p;
    int* d_count = (int*)ctx->misc.p;
    {
        TimeScope ts(ctx, "canny_nms");
        CK_HIP(ctx, hipMemsetAsync(d_count, 0, (size_t)n * 4, ctx->stream));
        dim3 grid((w + TW - 1) / TW, (h + TH - 1) / TH, n);
        hipLaunchKernelGGL(canny_nms_kernel, grid, dim3(256), 0, ctx->stream, d_planes, h, w, pitch, low, high, d_map,
                           d_labels, d_cand, d_count);
        CK_HIP(ctx, hipGetLastError());
    }
    if (d_map_out) CK_HIP(ctx, hipMemcpyAsync(d_map_out, d_map, npx, hipMemcpyDeviceToDevice, ctx->stream));
    {
        TimeScope ts(ctx, "canny_hyst");
        CK_HIP(ctx, hipMemsetAsync(d_edges, 0, npx, ctx->stream));
        dim3 grid(LIST_BLOCKS, n);
        hipLaunchKernelGGL(canny_link_kernel, grid, dim3(256), 0, ctx->stream, (const uint8_t*)d_map, h, w, d_labels,
                           (const int32_t*)d_cand, (const int*)d_count);
        hipLaunchKernelGGL(canny_flatten_mark_kernel, grid, dim3(256), 0, ctx->stream, (const uint8_t*)d_map, h, w,
                           d_labels, d_edges, (const int32_t*)d_cand, (const int*)d_count);
        hipLaunchKernelGGL(canny_final_kernel, grid, dim3(256), 0, ctx->stream, h, w, (const int32_t*)d_labels, d_edges,
                           (const int32_t*)d_cand, (const int*)d_count);
        CK_HIP(ctx, hipGetLastError());
    }
    return CK_OK;
}
