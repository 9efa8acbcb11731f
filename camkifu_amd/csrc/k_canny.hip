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
constexpr int LH = TH + 4;                   // pixel tile rows (2-px halo)
constexpr int LWD = TW / 4 + 2;              // pixel tile dwords per row (4-px halo each side)
constexpr int MW = TW + 2, MH = TH + 2;     // magnitude tile with 1-px halo

__global__ __launch_bounds__(256) void canny_nms_kernel(const uint8_t* __restrict__ planes, int h, int w, int pitch,
                                                        int low, int high, uint8_t* __restrict__ map,
                                                        int32_t* __restrict__ labels, int32_t* __restrict__ cand,
                                                        int* __restrict__ cand_count)
{
    // pixel tile: columns ox-4 .. ox+67 (18 aligned dwords per row), rows oy-2 .. oy+17
    __shared__ uint32_t pxw[3][LH][LWD];
    __shared__ int32_t mg[MH][MW + 1];       // mag | sector << 16
    __shared__ int32_t cbuf[TH * TW];        // candidates of this tile
    __shared__ int ccount, cbase;
    const int f = blockIdx.z;
    const int ox = blockIdx.x * TW, oy = blockIdx.y * TH;
    const int tid = threadIdx.x;
    const uint8_t* base = planes + (size_t)f * 3 * h * pitch;
    if (tid == 0) ccount = 0;

    for (int i = tid; i < 3 * LH * LWD; i += 256) {
        const int c = i / (LH * LWD), r = (i / LWD) % LH, cd = i % LWD;
        int y = oy - 2 + r;
        y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
        const int x = ox - 4 + 4 * cd;                 // pitch is a multiple of 64 and ox of 64: aligned
        uint32_t v = 0;
        if (x >= 0 && x < pitch) v = *reinterpret_cast<const uint32_t*>(base + ((size_t)c * h + y) * pitch + x);
        pxw[c][r][cd] = v;
    }
    __syncthreads();
    const uint8_t (*px)[LH][LWD * 4] = reinterpret_cast<const uint8_t (*)[LH][LWD * 4]>(pxw);

    const int TG22 = 13573;   // (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5)
    // Gradient tile (MH x MW, 1-px halo around the output tile).  One thread = one column x 6 rows, walked
    // top to bottom with the separable Sobel kept in registers: per pixel row the horizontal difference
    // r - l and the horizontal smooth l + 2c + r of each channel; dx = d[-1] + 2 d[0] + d[+1],
    // dy = s[+1] - s[-1].  Rows outside the image are already replicated in the pixel tile (clamped row
    // loads); columns are replicated by clamping the two side taps.
    if (tid < 3 * MW) {
        const int col = tid % MW, seg = tid / MW;
        const int x = ox - 1 + col;
        const bool xin = x >= 0 && x < w;
        const int cl = (x - 1 < 0 ? 0 : x - 1) - (ox - 4), cc = x - (ox - 4), cr = (x + 1 > w - 1 ? w - 1 : x + 1) - (ox - 4);
        constexpr int RSEG = MH / 3;                   // 6 gradient rows per thread
        int hd[3][3], hs[3][3];                        // [row slot][channel]
#pragma unroll
        for (int k = 0; k < RSEG + 2; k++) {
            const int pr = seg * RSEG + k;             // pixel-tile row
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int l = xin ? px[c][pr][cl] : 0, m = xin ? px[c][pr][cc] : 0, r = xin ? px[c][pr][cr] : 0;
                hd[k % 3][c] = r - l;
                hs[k % 3][c] = l + 2 * m + r;
            }
            if (k >= 2) {
                const int gr = seg * RSEG + k - 2;     // gradient-tile row
                const int y = oy - 1 + gr;
                int best = -1, bdx = 0, bdy = 0;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const int dx = hd[(k - 2) % 3][c] + 2 * hd[(k - 1) % 3][c] + hd[k % 3][c];
                    const int dy = hs[k % 3][c] - hs[(k - 2) % 3][c];
                    const int m = abs(dx) + abs(dy);
                    if (m > best) { best = m; bdx = dx; bdy = dy; }
                }
                int32_t packed = 0;                    // outside the image the magnitude is 0
                if (xin && y >= 0 && y < h) {
                    const int ax = abs(bdx), ay = abs(bdy) << 15;
                    const int tg22x = ax * TG22;
                    int sector;
                    if (ay < tg22x) sector = 0;
                    else if (ay > tg22x + (ax << 16)) sector = 1;
                    else sector = ((bdx ^ bdy) < 0) ? 3 : 2;
                    packed = best | (sector << 16);
                }
                mg[gr][col] = packed;
            }
        }
    }
    __syncthreads();

    // NMS: one thread = 4 consecutive pixels of one row (one dword of the map)
    {
        const int r = tid >> 4, col0 = (tid & 15) * 4;
        const int y = oy + r;
        uint32_t mapw = 0x01010101u;
        int keepmask = 0;
        if (y < h) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int col = col0 + k, x = ox + col;
                if (x >= w) continue;
                const int32_t pk = mg[r + 1][col + 1];
                const int m = pk & 0xFFFF, sector = pk >> 16;
                bool keep = false;
                if (m > low) {
                    if (sector == 0) keep = m > (mg[r + 1][col] & 0xFFFF) && m >= (mg[r + 1][col + 2] & 0xFFFF);
                    else if (sector == 1) keep = m > (mg[r][col + 1] & 0xFFFF) && m >= (mg[r + 2][col + 1] & 0xFFFF);
                    else {
                        const int s = sector == 3 ? -1 : 1;
                        keep = m > (mg[r][col + 1 - s] & 0xFFFF) && m > (mg[r + 2][col + 1 + s] & 0xFFFF);
                    }
                }
                if (keep) {
                    const uint32_t v = m > high ? 2u : 0u;
                    mapw = (mapw & ~(0xFFu << (8 * k))) | (v << (8 * k));
                    keepmask |= 1 << k;
                }
            }
            const int x0 = ox + col0;
            const size_t idx = ((size_t)f * h + y) * w + x0;
            if (x0 + 3 < w && ((idx & 3) == 0)) *reinterpret_cast<uint32_t*>(map + idx) = mapw;
            else {
#pragma unroll
                for (int k = 0; k < 4; k++) if (x0 + k < w) map[idx + k] = (uint8_t)(mapw >> (8 * k));
            }
        }
        const int nk = __builtin_popcount(keepmask);
        int slot = nk ? atomicAdd(&ccount, nk) : 0;
        if (nk) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (keepmask & (1 << k)) {
                    const int p = y * w + ox + col0 + k;
                    labels[(size_t)f * h * w + p] = p;
                    cbuf[slot++] = p;
                }
        }
    }
    __syncthreads();
    if (tid == 0) cbase = ccount ? atomicAdd(cand_count + f, ccount) : 0;
    __syncthreads();
    int32_t* clist = cand + (size_t)f * h * w + cbase;
    for (int i = tid; i < ccount; i += 256) clist[i] = cbuf[i];
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

// border_flag (nullable): frames with an edge pixel on the image frame are flagged -- K3 clears the frame
// before labelling, so only the others may reuse these component roots as they are
__global__ __launch_bounds__(256) void canny_final_kernel(int h, int w, const int32_t* __restrict__ labels,
                                                          uint8_t* __restrict__ edges, const int32_t* __restrict__ cand,
                                                          const int* __restrict__ cand_count, int* __restrict__ border_flag)
{
    const int f = blockIdx.y;
    const int n = cand_count[f];
    const size_t off = (size_t)f * h * w;
    const int32_t* C = cand + off;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += LIST_BLOCKS * 256) {
        const int p = C[i];
        const int root = labels[off + p];
        const uint8_t e = edges[off + root];           // root entries were written by the previous kernel
        if (root != p) edges[off + p] = e;
        if (border_flag && e) {
            const int y = p / w, x = p - y * w;
            if (x == 0 || y == 0 || x == w - 1 || y == h - 1) border_flag[f] = 1;
        }
    }
}

}  // namespace

int k_canny_planar(ck_ctx* ctx, const uint8_t* d_planes, int n, int h, int w, int pitch, int low, int high,
                   uint8_t* d_map, int32_t* d_labels, uint8_t* d_edges, uint8_t* d_map_out, int* d_border_flag)
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
        if (d_border_flag) CK_HIP(ctx, hipMemsetAsync(d_border_flag, 0, (size_t)n * 4, ctx->stream));
        dim3 grid(LIST_BLOCKS, n);
        hipLaunchKernelGGL(canny_link_kernel, grid, dim3(256), 0, ctx->stream, (const uint8_t*)d_map, h, w, d_labels,
                           (const int32_t*)d_cand, (const int*)d_count);
        hipLaunchKernelGGL(canny_flatten_mark_kernel, grid, dim3(256), 0, ctx->stream, (const uint8_t*)d_map, h, w,
                           d_labels, d_edges, (const int32_t*)d_cand, (const int*)d_count);
        hipLaunchKernelGGL(canny_final_kernel, grid, dim3(256), 0, ctx->stream, h, w, (const int32_t*)d_labels, d_edges,
                           (const int32_t*)d_cand, (const int*)d_count, d_border_flag);
        CK_HIP(ctx, hipGetLastError());
    }
    return CK_OK;
}
