/*
 * ora_cnn.c -- ORACLE (test infrastructure only; see ck_oracle.h).
 *
 * K10  NNManager._subregion / getrect / _get_rect_nn / _get_x: 10x10 patches of 40x40x3
 *      with origins {0,40,..,320,340}   (/root/reference/src/camkifu/stone/nn_manager.py:92-126,
 *      216-218, 256-275)
 * K11  NNManager.create_net: Conv5x5x32 relu, Conv5x5x32 relu, MaxPool2, Conv3x3x90 relu,
 *      Conv3x3x90 relu, MaxPool2, Flatten, Dense160 relu, Dense81 softmax; 'valid', stride 1,
 *      channels-last   (nn_manager.py:277-298).  Fed raw uint8 0..255 (nn_cache.py:47-51).
 * K12  NNCache.predict_4_stones / predict_all_stones + NNManager.compute_stones
 *      (/root/reference/src/camkifu/stone/nn_cache.py:25-41, nn_manager.py:246-254)
 *
 * Keras-1 on Theano runs conv2d as a TRUE convolution (kernel flipped in both spatial
 * axes); kernels are stored [kh][kw][cin][cout].  Accumulation here is a float32 fmaf
 * chain in (kh, kw, cin) order, bias added last.  "parity unpinned" for the numerics (no
 * weights or activations ship with the reference); the codec is pinned by
 * test/camkifu/stone/test_tmanager.py:18-27.
 */
#include "ck_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* valid true-convolution + bias + relu, channels-last.  in: H x W x Cin float */
static void conv_relu(const float* in, int H, int W, int Cin,
                      const float* k, const float* b, int KH, int KW, int Cout, float* out)
{
    int OH = H - KH + 1, OW = W - KW + 1;
    for (int oy = 0; oy < OH; oy++)
        for (int ox = 0; ox < OW; ox++)
            for (int o = 0; o < Cout; o++) {
                float acc = 0.f;
                for (int i = 0; i < KH; i++)
                    for (int j = 0; j < KW; j++) {
                        const float* ip = in + ((size_t)(oy + i) * W + (ox + j)) * Cin;
                        const float* kp = k + (((size_t)(KH - 1 - i) * KW + (KW - 1 - j)) * Cin) * Cout + o;
                        for (int c = 0; c < Cin; c++) acc = fmaf(ip[c], kp[(size_t)c * Cout], acc);
                    }
                acc += b[o];
                out[((size_t)oy * OW + ox) * Cout + o] = acc > 0.f ? acc : 0.f;
            }
}

static void maxpool2(const float* in, int H, int W, int C, float* out)
{
    int OH = H / 2, OW = W / 2;
    for (int y = 0; y < OH; y++)
        for (int x = 0; x < OW; x++)
            for (int c = 0; c < C; c++) {
                float a = in[((size_t)(2 * y) * W + 2 * x) * C + c];
                float b = in[((size_t)(2 * y) * W + 2 * x + 1) * C + c];
                float d = in[((size_t)(2 * y + 1) * W + 2 * x) * C + c];
                float e = in[((size_t)(2 * y + 1) * W + 2 * x + 1) * C + c];
                float m = a > b ? a : b; m = m > d ? m : d; m = m > e ? m : e;
                out[((size_t)y * OW + x) * C + c] = m;
            }
}

static void dense(const float* in, int nin, const float* w, const float* b, int nout, float* out)
{
    for (int o = 0; o < nout; o++) {
        float acc = 0.f;
        for (int i = 0; i < nin; i++) acc = fmaf(in[i], w[(size_t)i * nout + o], acc);
        out[o] = acc + b[o];
    }
}

void ora_cnn_forward(const ora_cnn_weights* Wt, const uint8_t* patches, int n,
                     float* y_out, float* logits_out)
{
    ora_cnn_forward_maps(Wt, patches, n, y_out, logits_out, 0, 0);
}

/* the same forward pass; optionally also the outputs of the two MaxPooling2D layers (nn_manager.py:286, 292):
 * pool2_out n x 16 x 16 x 32, pool4_out n x 6 x 6 x 90, channels-last */
void ora_cnn_forward_maps(const ora_cnn_weights* Wt, const uint8_t* patches, int n,
                          float* y_out, float* logits_out, float* pool2_out, float* pool4_out)
{
    /* patches are independent: OpenMP over patches (the CPU baseline uses every core) */
#pragma omp parallel
    {
    float* x0 = (float*)malloc(sizeof(float) * 40 * 40 * 3);
    float* a1 = (float*)malloc(sizeof(float) * 36 * 36 * 32);
    float* a2 = (float*)malloc(sizeof(float) * 32 * 32 * 32);
    float* p2 = (float*)malloc(sizeof(float) * 16 * 16 * 32);
    float* a3 = (float*)malloc(sizeof(float) * 14 * 14 * 90);
    float* a4 = (float*)malloc(sizeof(float) * 12 * 12 * 90);
    float* p4 = (float*)malloc(sizeof(float) * 6 * 6 * 90);
    float h1[160], lg[81];
#pragma omp for schedule(dynamic)
    for (int p = 0; p < n; p++) {
        const uint8_t* src = patches + (size_t)p * 4800;
        for (int i = 0; i < 4800; i++) x0[i] = (float)src[i];
        conv_relu(x0, 40, 40, 3, Wt->c1w, Wt->c1b, 5, 5, 32, a1);
        conv_relu(a1, 36, 36, 32, Wt->c2w, Wt->c2b, 5, 5, 32, a2);
        maxpool2(a2, 32, 32, 32, p2);
        conv_relu(p2, 16, 16, 32, Wt->c3w, Wt->c3b, 3, 3, 90, a3);
        conv_relu(a3, 14, 14, 90, Wt->c4w, Wt->c4b, 3, 3, 90, a4);
        maxpool2(a4, 12, 12, 90, p4);
        if (pool2_out) memcpy(pool2_out + (size_t)p * 8192, p2, sizeof(float) * 8192);
        if (pool4_out) memcpy(pool4_out + (size_t)p * 3240, p4, sizeof(float) * 3240);
        dense(p4, 3240, Wt->d1w, Wt->d1b, 160, h1);
        for (int i = 0; i < 160; i++) h1[i] = h1[i] > 0.f ? h1[i] : 0.f;
        dense(h1, 160, Wt->d2w, Wt->d2b, 81, lg);
        if (logits_out) memcpy(logits_out + (size_t)p * 81, lg, sizeof lg);
        float mx = lg[0];
        for (int i = 1; i < 81; i++) mx = lg[i] > mx ? lg[i] : mx;
        float e[81], sum = 0.f;
        for (int i = 0; i < 81; i++) { e[i] = expf(lg[i] - mx); sum += e[i]; }
        for (int i = 0; i < 81; i++) y_out[(size_t)p * 81 + i] = e[i] / sum;
    }
    free(p4); free(a4); free(a3); free(p2); free(a2); free(a1); free(x0);
    }
}

/* NNManager._subregion + _get_rect_nn, gsize=19, split=10, step=2, 380x380 image:
 * region index i -> first pixel row 20*rs where rs = 2i, except i=9 -> rs=17 (340). */
static int region_origin(int i) { int rs = 2 * i; if (19 - rs < 2) rs = 17; return 20 * rs; }

void ora_cnn_region_patches(const uint8_t* goban, uint8_t* patches)
{
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 10; j++) {
            int x0 = region_origin(i), y0 = region_origin(j);   /* x = row, y = col in the reference's naming */
            uint8_t* dst = patches + (size_t)(i * 10 + j) * 4800;
            for (int r = 0; r < 40; r++)
                memcpy(dst + (size_t)r * 120, goban + ((size_t)(x0 + r) * 380 + y0) * 3, 120);
        }
}

void ora_cnn_predict_regions(const ora_cnn_weights* Wt, const uint8_t* goban,
                             float* y_out, float* logits_out)
{
    uint8_t* patches = (uint8_t*)malloc(100 * 4800);
    ora_cnn_region_patches(goban, patches);
    ora_cnn_forward(Wt, patches, 100, y_out, logits_out);
    free(patches);
}

void ora_decode_all(const float* y, uint8_t* labels, double* conf)
{
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 10; j++) {
            const float* yy = y + (size_t)(i * 10 + j) * 81;
            int label = 0;
            for (int k = 1; k < 81; k++) if (yy[k] > yy[label]) label = k;   /* np.argmax: first max */
            /* confidence = max(y) / sum(y): python sum() over float32 scalars promotes to
             * float64 and adds in index order */
            double s = 0.;
            for (int k = 0; k < 81; k++) s += (double)yy[k];
            double cf = (double)yy[label] / s;
            int rs = 2 * i, cs = 2 * j;
            if (19 - rs < 2) rs = 17;
            if (19 - cs < 2) cs = 17;
            int kk = label;
            int digit[4];
            for (int d = 3; d >= 0; d--) {          /* compute_stones */
                int p3 = 1; for (int t = 0; t < d; t++) p3 *= 3;
                digit[d] = kk / p3; kk %= p3;
            }
            for (int d = 0; d < 4; d++) {
                int r = rs + d / 2, c = cs + d % 2;
                labels[r * 19 + c] = (uint8_t)digit[d];
                conf[r * 19 + c] = cf;
            }
        }
}
