// k_warp.hip -- K8: cv2.warpPerspective(frame, M, (380, 380))  INTER_LINEAR, BORDER_CONSTANT 0
// (reference: src/camkifu/stone/stonesfinder.py:140).
//
// One thread per destination pixel (coalesced 3-byte writes along x); source taps are a
// gather inside the board quad, served by L2.  Coordinates follow the library bit for bit:
// homography in float64 evaluated per 64x16 destination block (X0 + M0*x1)*(32/W), rounded
// half-to-even to 1/32 px, 15-bit integer bilinear weights, (sum + 2^14) >> 15.
// fp contraction is off so the double arithmetic matches a scalar CPU evaluation.
#include "ck_common.h"

#pragma clang fp contract(off)

namespace {

__global__ __launch_bounds__(256) void warp_kernel(const uint8_t* __restrict__ src, int h, int w,
                                                   const double* __restrict__ minv, int m_count, int dsize,
                                                   uint8_t* __restrict__ dst)
{
#pragma clang fp contract(off)
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int f = blockIdx.z;
    if (dx >= dsize || dy >= dsize) return;
    const double* M = minv + (m_count == 1 ? 0 : (size_t)f * 9);
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
    int sx = X >> 5, sy = Y >> 5;
    sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);
    sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
    const int fx = X & 31, fy = Y & 31;
    const int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32;
    const int w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
    const uint8_t* s = src + (size_t)f * h * w * 3;
    const bool x0in = (unsigned)sx < (unsigned)w, x1in = (unsigned)(sx + 1) < (unsigned)w;
    const bool y0in = (unsigned)sy < (unsigned)h, y1in = (unsigned)(sy + 1) < (unsigned)h;
    uint8_t* d = dst + (((size_t)f * dsize + dy) * dsize + dx) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int v00 = (x0in && y0in) ? s[((size_t)sy * w + sx) * 3 + c] : 0;
        const int v01 = (x1in && y0in) ? s[((size_t)sy * w + sx + 1) * 3 + c] : 0;
        const int v10 = (x0in && y1in) ? s[((size_t)(sy + 1) * w + sx) * 3 + c] : 0;
        const int v11 = (x1in && y1in) ? s[((size_t)(sy + 1) * w + sx + 1) * 3 + c] : 0;
        int v = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
        d[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

}  // namespace

int k_warp(ck_ctx* ctx, const uint8_t* d_bgr, int n, int h, int w, const double* d_minv, int m_count,
           int dsize, uint8_t* d_out)
{
    TimeScope ts(ctx, "warp");
    dim3 grid((dsize + 63) / 64, (dsize + 3) / 4, n);
    hipLaunchKernelGGL(warp_kernel, grid, dim3(256), 0, ctx->stream, d_bgr, h, w, d_minv, m_count, dsize, d_out);
    CK_HIP(ctx, hipGetLastError());
    return CK_OK;
}
