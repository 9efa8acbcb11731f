// k_color.hip -- frame source: planar YUV 4:2:0 (I420) -> interleaved BGR in HBM.
// A video file hands the pipeline I420 frames (1.5 B/px); uploading those and converting on the
// device halves the PCIe traffic of the fast-file path, and every later stage reads the same
// HxWx3 BGR buffer cv2.VideoCapture.read() gives the reference's finders (core/vmanager.py:506-509).
// BT.601 studio range, 20-bit fixed point (cv2.cvtColor COLOR_YUV2BGR_I420 constants).
//
// HBM-bound: reads 1.5 B/px, writes 3 B/px.  One thread = 4x2 pixels (two Y dwords, one U and
// one V ushort, two 12-byte stores); a wave covers 256 x 2 pixels with contiguous 768-byte rows.
#include "ck_common.h"

namespace {

constexpr int CY = 1220542, CUB = 2116026, CUG = -409993, CVG = -852492, CVR = 1673527, SHIFT = 20;

__device__ __forceinline__ uint32_t sat8(int v) { return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// converts PX horizontally adjacent pixels of rows y and y+1 starting at even column x
template <int PX>
__device__ __forceinline__ void convert_block(const uint8_t* __restrict__ Y, const uint8_t* __restrict__ U,
                                              const uint8_t* __restrict__ V, int w, int x, int y,
                                              uint8_t* __restrict__ bgr)
{
    uint8_t yv[2][PX], uv[PX / 2], vv[PX / 2];
    if constexpr (PX == 4) {
        const uint32_t a = *reinterpret_cast<const uint32_t*>(Y + (size_t)y * w + x);
        const uint32_t b = *reinterpret_cast<const uint32_t*>(Y + (size_t)(y + 1) * w + x);
        const uint16_t u2 = *reinterpret_cast<const uint16_t*>(U + (size_t)(y / 2) * (w / 2) + x / 2);
        const uint16_t v2 = *reinterpret_cast<const uint16_t*>(V + (size_t)(y / 2) * (w / 2) + x / 2);
#pragma unroll
        for (int k = 0; k < 4; k++) { yv[0][k] = (uint8_t)(a >> (8 * k)); yv[1][k] = (uint8_t)(b >> (8 * k)); }
        uv[0] = (uint8_t)u2; uv[1] = (uint8_t)(u2 >> 8); vv[0] = (uint8_t)v2; vv[1] = (uint8_t)(v2 >> 8);
    } else {
#pragma unroll
        for (int k = 0; k < PX; k++) { yv[0][k] = Y[(size_t)y * w + x + k]; yv[1][k] = Y[(size_t)(y + 1) * w + x + k]; }
#pragma unroll
        for (int k = 0; k < PX / 2; k++) {
            uv[k] = U[(size_t)(y / 2) * (w / 2) + x / 2 + k];
            vv[k] = V[(size_t)(y / 2) * (w / 2) + x / 2 + k];
        }
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
        uint8_t o[PX * 3];
#pragma unroll
        for (int k = 0; k < PX; k++) {
            const int u = (int)uv[k / 2] - 128, v = (int)vv[k / 2] - 128;
            const int ruv = (1 << (SHIFT - 1)) + CVR * v;
            const int guv = (1 << (SHIFT - 1)) + CVG * v + CUG * u;
            const int buv = (1 << (SHIFT - 1)) + CUB * u;
            int yy = (int)yv[r][k] - 16;
            yy = (yy < 0 ? 0 : yy) * CY;
            o[3 * k] = (uint8_t)sat8((yy + buv) >> SHIFT);
            o[3 * k + 1] = (uint8_t)sat8((yy + guv) >> SHIFT);
            o[3 * k + 2] = (uint8_t)sat8((yy + ruv) >> SHIFT);
        }
        uint8_t* d = bgr + ((size_t)(y + r) * w + x) * 3;
        if constexpr (PX == 4) {
            // x % 4 == 0 and w % 4 == 0: the 12 bytes are dword aligned
            uint32_t* d4 = reinterpret_cast<uint32_t*>(d);
            d4[0] = (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16) | ((uint32_t)o[3] << 24);
            d4[1] = (uint32_t)o[4] | ((uint32_t)o[5] << 8) | ((uint32_t)o[6] << 16) | ((uint32_t)o[7] << 24);
            d4[2] = (uint32_t)o[8] | ((uint32_t)o[9] << 8) | ((uint32_t)o[10] << 16) | ((uint32_t)o[11] << 24);
        } else {
#pragma unroll
            for (int k = 0; k < PX * 3; k++) d[k] = o[k];
        }
    }
}

template <int PX>
__global__ __launch_bounds__(256) void i420_to_bgr_kernel(const uint8_t* __restrict__ src, int h, int w,
                                                          uint8_t* __restrict__ dst)
{
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * PX;
    const int y = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 2;
    const int f = blockIdx.z;
    if (x >= w || y >= h) return;
    const uint8_t* Y = src + (size_t)f * (h * w * 3 / 2);
    const uint8_t* U = Y + (size_t)h * w;
    const uint8_t* V = U + (size_t)(h / 2) * (w / 2);
    convert_block<PX>(Y, U, V, w, x, y, dst + (size_t)f * h * w * 3);
}

// cvtColor(COLOR_BGR2GRAY) of a planar BGR frame (8-bit fixed point: (1868 B + 9617 G + 4899 R + 2^13) >> 14) folded
// straight into a 256-bin histogram per frame (LDS atomics, one flush per workgroup): the input of the Otsu level of
// SfContours.get_canny (stone/sf_contours.py:338-339).
__global__ __launch_bounds__(256) void gray_hist_kernel(const uint8_t* __restrict__ planes, int h, int w, int pitch,
                                                        int* __restrict__ hist)
{
    __shared__ int lh[256];
    const int f = blockIdx.y;
    lh[threadIdx.x] = 0;
    __syncthreads();
    const uint8_t* B = planes + (size_t)f * 3 * h * pitch;
    const uint8_t* G = B + (size_t)h * pitch;
    const uint8_t* R = G + (size_t)h * pitch;
    const int npx = h * w;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npx; i += gridDim.x * 256) {
        const int y = i / w, x = i - y * w;
        const int o = y * pitch + x;
        const int g = (1868 * B[o] + 9617 * G[o] + 4899 * R[o] + (1 << 13)) >> 14;
        atomicAdd(&lh[g], 1);
    }
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(&hist[f * 256 + threadIdx.x], lh[threadIdx.x]);
}

}  // namespace

int k_i420_to_bgr(ck_ctx* ctx, const uint8_t* d_i420, int n, int h, int w, uint8_t* d_bgr)
{
    TimeScope ts(ctx, "i420_to_bgr");
    const bool wide = (w % 4) == 0 && (((uintptr_t)d_i420 | (uintptr_t)d_bgr) & 3) == 0 && ((h * w * 3 / 2) % 4) == 0;
    if (wide) {
        dim3 grid((w / 4 + 63) / 64, (h / 2 + 3) / 4, n);
        hipLaunchKernelGGL(i420_to_bgr_kernel<4>, grid, dim3(256), 0, ctx->stream, d_i420, h, w, d_bgr);
    } else {
        dim3 grid((w / 2 + 63) / 64, (h / 2 + 3) / 4, n);
        hipLaunchKernelGGL(i420_to_bgr_kernel<2>, grid, dim3(256), 0, ctx->stream, d_i420, h, w, d_bgr);
    }
    CK_HIP(ctx, hipGetLastError());
    return CK_OK;
}

int k_gray_hist(ck_ctx* ctx, const uint8_t* d_planes, int n, int h, int w, int pitch, int* d_hist)
{
    CK_HIP(ctx, hipMemsetAsync(d_hist, 0, (size_t)n * 256 * 4, ctx->stream));
    int blocks = (h * w + 256 * 16 - 1) / (256 * 16);
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(gray_hist_kernel, dim3(blocks, n), dim3(256), 0, ctx->stream, d_planes, h, w, pitch, d_hist);
    CK_HIP(ctx, hipGetLastError());
    return CK_OK;
}
