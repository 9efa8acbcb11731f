/*
 * ora_color.c -- CPU oracle (TEST INFRASTRUCTURE ONLY) for the frame-source colour conversion:
 * planar YUV 4:2:0 (I420, as stored in a .y4m file) -> interleaved BGR, the layout every consumer
 * of the reference's capture gets from cv2.VideoCapture.read() (core/vmanager.py:506-509, 584).
 *
 * The reference decodes through OpenCV's FFmpeg backend, which is not available here; the
 * conversion below is the fixed-point ITU-R BT.601 (studio range) formula of
 * cv2.cvtColor(..., COLOR_YUV2BGR_I420) as recalled from OpenCV 3.1 (20-bit coefficients).
 * "parity unpinned": the reference holds no decoded-frame fixture.
 */
#include "ck_oracle.h"

static uint8_t sat8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

void ora_i420_to_bgr(const uint8_t* i420, int h, int w, uint8_t* bgr)
{
    const int CY = 1220542, CUB = 2116026, CUG = -409993, CVG = -852492, CVR = 1673527, SHIFT = 20;
    const uint8_t* Y = i420;
    const uint8_t* U = Y + (long)h * w;
    const uint8_t* V = U + (long)(h / 2) * (w / 2);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int u = (int)U[(y / 2) * (w / 2) + x / 2] - 128, v = (int)V[(y / 2) * (w / 2) + x / 2] - 128;
            const int ruv = (1 << (SHIFT - 1)) + CVR * v;
            const int guv = (1 << (SHIFT - 1)) + CVG * v + CUG * u;
            const int buv = (1 << (SHIFT - 1)) + CUB * u;
            int yy = (int)Y[(long)y * w + x] - 16;
            yy = (yy < 0 ? 0 : yy) * CY;
            uint8_t* o = bgr + ((long)y * w + x) * 3;
            o[0] = sat8((yy + buv) >> SHIFT);
            o[1] = sat8((yy + guv) >> SHIFT);
            o[2] = sat8((yy + ruv) >> SHIFT);
        }
}
