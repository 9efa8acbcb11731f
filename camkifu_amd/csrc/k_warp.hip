// k_warp.hip -- K8: cv2.warpPerspective(frame, M, (380, 380))  INTER_LINEAR, BORDER_CONSTANT 0
// (reference: src/camkifu/stone/stonesfinder.py:140).
//
// One thread per four destination pixels of a row; source taps are a gather inside the board
// quad, served by L2.  Coordinates follow the library bit for bit:
// homography in float64 evaluated per 64x16 destination block (X0 + M0*x1)*(32/W), rounded
// half-to-even to 1/32 px, 15-bit integer bilinear weights, (sum + 2^14) >> 15.
// fp contraction is off so the double arithmetic matches a scalar CPU evaluation.
#include "ck_common.h"

#pragma clang fp contract(off)

namespace {

// source coordinates and weights of one destination pixel, exactly as the library computes them
struct WarpTap { int sx, sy, w00, w01, w10, w11; };
__device__ __forceinline__ WarpTap warp_tap(const double* __restrict__ M, int dsize, int dx, int dy)
{
#pragma clang fp contract(off)
    // the library walks the destination in blocks of bw0 x bh0 and restarts the linear
    // terms at each block origin: reproduce the same association of the additions
    int bh0 = 16 < dsize ? 16 : dsize;
    int bw0 = 1024 / bh0 < dsize ? 1024 / bh0 : dsize;
    const int bx = (dx / bw0) * bw0, x1 = dx - bx;
    const double X0 = M[0] * bx + M[1] * dy + M[2];
    const double Y0 = M[3] * bx + M[4] * dy + M[5];
    const double W0 = M[6] * bx + M[7] * dy + M[8];
    double W = W0 + M[6] * x1;
    W = W != 0.0 ? 32.0 / W : 0.0;
    double fX = (X0 + M[0] * x1) * W;
    double fY = (Y0 + M[3] * x1) * W;
    fX = fX < -2147483648.0 ? -2147483648.0 : (fX > 2147483647.0 ? 2147483647.0 : fX);
    fY = fY < -2147483648.0 ? -2147483648.0 : (fY > 2147483647.0 ? 2147483647.0 : fY);
    const int X = (int)rint(fX), Y = (int)rint(fY);
    WarpTap t;
    t.sx = X >> 5; t.sy = Y >> 5;
    t.sx = t.sx < -32768 ? -32768 : (t.sx > 32767 ? 32767 : t.sx);
    t.sy = t.sy < -32768 ? -32768 : (t.sy > 32767 ? 32767 : t.sy);
    const int fx = X & 31, fy = Y & 31;
    t.w00 = (32 - fy) * (32 - fx) * 32; t.w01 = (32 - fy) * fx * 32;
    t.w10 = fy * (32 - fx) * 32; t.w11 = fy * fx * 32;
    return t;
}

// One thread = FOUR destination pixels of a row (round 4; one pixel per thread before): the 12 result bytes leave as
// three aligned dwords, and a pixel whose four taps -- and one spare source pixel -- lie inside the frame reads each source
// row as ONE unaligned 8-byte load (the two taps of a row are 6 contiguous bytes of the interleaved frame) instead of
// six single-byte loads.  15 memory instructions per pixel became 2.75; the arithmetic is unchanged.
__global__ __launch_bounds__(256) void warp_kernel(const uint8_t* __restrict__ src, int h, int w,
                                                   const double* __restrict__ minv, int m_count, int dsize,
                                                   uint8_t* __restrict__ dst)
{
#pragma clang fp contract(off)
    const int dx0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int f = blockIdx.z;
    if (dx0 >= dsize || dy >= dsize) return;
    const double* M = minv + (m_count == 1 ? 0 : (size_t)f * 9);
    const uint8_t* s = src + (size_t)f * h * w * 3;
    struct __attribute__((packed, aligned(1))) u64u { uint32_t a, b; };
    uint32_t out[3] = {0u, 0u, 0u};                   // 12 bytes: B G R of four pixels
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int dx = dx0 + k;
        if (dx >= dsize) break;
        const WarpTap t = warp_tap(M, dsize, dx, dy);
        int v[3];
        if (t.sx >= 0 && t.sx + 2 < w && t.sy >= 0 && t.sy + 1 < h) {
            const uint8_t* p = s + ((size_t)t.sy * w + t.sx) * 3;
            const u64u r0 = *reinterpret_cast<const u64u*>(p), r1 = *reinterpret_cast<const u64u*>(p + (size_t)w * 3);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int v00 = (r0.a >> (8 * c)) & 0xFF;
                const int v01 = c == 0 ? (int)(r0.a >> 24) : (int)((r0.b >> (8 * (c - 1))) & 0xFF);
                const int v10 = (r1.a >> (8 * c)) & 0xFF;
                const int v11 = c == 0 ? (int)(r1.a >> 24) : (int)((r1.b >> (8 * (c - 1))) & 0xFF);
                v[c] = (v00 * t.w00 + v01 * t.w01 + v10 * t.w10 + v11 * t.w11 + (1 << 14)) >> 15;
            }
        } else {
            const bool x0in = (unsigned)t.sx < (unsigned)w, x1in = (unsigned)(t.sx + 1) < (unsigned)w;
            const bool y0in = (unsigned)t.sy < (unsigned)h, y1in = (unsigned)(t.sy + 1) < (unsigned)h;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int v00 = (x0in && y0in) ? s[((size_t)t.sy * w + t.sx) * 3 + c] : 0;
                const int v01 = (x1in && y0in) ? s[((size_t)t.sy * w + t.sx + 1) * 3 + c] : 0;
                const int v10 = (x0in && y1in) ? s[((size_t)(t.sy + 1) * w + t.sx) * 3 + c] : 0;
                const int v11 = (x1in && y1in) ? s[((size_t)(t.sy + 1) * w + t.sx + 1) * 3 + c] : 0;
                v[c] = (v00 * t.w00 + v01 * t.w01 + v10 * t.w10 + v11 * t.w11 + (1 << 14)) >> 15;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const uint32_t b = (uint32_t)(v[c] < 0 ? 0 : (v[c] > 255 ? 255 : v[c]));
            const int pos = 3 * k + c;                    // byte position among the 12
            out[pos >> 2] |= b << (8 * (pos & 3));
        }
    }
    uint8_t* d = dst + (((size_t)f * dsize + dy) * dsize + dx0) * 3;
    if (dx0 + 3 < dsize && ((reinterpret_cast<uintptr_t>(d) & 3) == 0)) {
        uint32_t* d4 = reinterpret_cast<uint32_t*>(d);
        d4[0] = out[0]; d4[1] = out[1]; d4[2] = out[2];
    } else {
        for (int k = 0; k < 4 && dx0 + k < dsize; k++)
#pragma unroll
            for (int c = 0; c < 3; c++) { const int pos = 3 * k + c; d[pos] = (uint8_t)(out[pos >> 2] >> (8 * (pos & 3))); }
    }
}

}  // namespace

int k_warp(ck_ctx* ctx, const uint8_t* d_bgr, int n, int h, int w, const double* d_minv, int m_count,
           int dsize, uint8_t* d_out)
{
    TimeScope ts(ctx, "warp");
    dim3 grid((dsize + 255) / 256, (dsize + 3) / 4, n);      // a thread owns four pixels of a row: 256 per wave
    hipLaunchKernelGGL(warp_kernel, grid, dim3(256), 0, ctx->stream, d_bgr, h, w, d_minv, m_count, dsize, d_out);
    CK_HIP(ctx, hipGetLastError());
    return CK_OK;
}
