// The block-scaled MFMA with e4m3 operands (v_mfma_scale_f32_16x16x128_f8f6f4) as the split-precision classifier would use it
// for its cross terms (tools/sim_split_q8.py): operand lane map and scale semantics checked with exact data, cycles per
// instruction, and v_cvt_scalef32_pk_fp8_f16's rounding against a host model of OCP e4m3.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_scale_f8.hip -o tools/micro/mfma_scale_f8 && tools/micro/mfma_scale_f8
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// e4m3 byte of a small non-negative integer or of v * 2^-k chosen by the host
static uint8_t e4m3_of(float v)
{
    if (v == 0.f) return 0;
    const uint8_t s = v < 0 ? 0x80 : 0;
    v = fabsf(v);
    int e;
    float m = frexpf(v, &e);            // v = m 2^e, m in [0.5, 1)
    e -= 1; m *= 2.f;                   // v = m 2^e, m in [1, 2)
    if (e < -6) {                       // subnormal: step 2^-9
        const int q = (int)lrintf(v * 512.f);
        return s | (uint8_t)q;
    }
    int q = (int)lrintf((m - 1.f) * 8.f);
    if (q == 8) { q = 0; e += 1; }
    if (e > 8 || (e == 8 && q > 6)) { e = 8; q = 6; }     // 448
    return s | (uint8_t)(((e + 7) << 3) | q);
}
static float e4m3_val(uint8_t b)
{
    const int e = (b >> 3) & 15, q = b & 7;
    const float v = e == 0 ? q / 512.f : (1.f + q / 8.f) * exp2f((float)(e - 7));
    return (b & 0x80) ? -v : v;
}

__global__ void one(const v8i* a, const v8i* b, v4f* c, const int* sa, const int* sb)
{
    v4f acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
    c[threadIdx.x] = acc;
}

template <int WHICH>
__global__ __launch_bounds__(256) void spin(float* out, int iters)
{
    v8i a, b;
    for (int i = 0; i < 8; i++) { a[i] = 0x38383838 + threadIdx.x; b[i] = 0x38383838 + blockIdx.x; }
    h8 ah, bh;
    for (int i = 0; i < 8; i++) { ah[i] = (_Float16)(1.f + threadIdx.x); bh[i] = (_Float16)(1.f + blockIdx.x); }
    v4f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const int one_ = 127;
    for (int i = 0; i < iters; i++) {
#define SC(C) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+a"(C) : "v"(a), "v"(b), "v"(one_))
#define HF(C) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(C) : "v"(ah), "v"(bh))
        if constexpr (WHICH == 0) { SC(c0); SC(c1); SC(c2); SC(c3); }
        else if constexpr (WHICH == 1) { HF(c0); HF(c1); HF(c2); HF(c3); }
        else { HF(c0); HF(c0); SC(c1); HF(c2); HF(c2); SC(c3); }     // the classifier's mix: two taps' main terms per scaled instruction
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

__global__ void cvt(const h2* in, s2* out, int n, float scale)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { s2 old = {0, 0}; out[i] = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(old, in[i], scale, false); }
}

int main()
{
    // ---- 1. lane map and scales: A[m][k], B[k][n] with k = 32 (lane >> 4) + j in byte j of the lane's 32 bytes; block scale 2^(s - 127)
    uint8_t A[16][128], B[128][16];
    srand(7);
    for (int m = 0; m < 16; m++) for (int k = 0; k < 128; k++) A[m][k] = e4m3_of((float)(rand() % 9) - 4.f);
    for (int k = 0; k < 128; k++) for (int n = 0; n < 16; n++) B[k][n] = e4m3_of((float)(rand() % 7) - 3.f + 0.5f * (n & 1));
    uint8_t ha[64][32], hb[64][32];
    int hsa[64], hsb[64];
    for (int l = 0; l < 64; l++) {
        for (int j = 0; j < 32; j++) { ha[l][j] = A[l & 15][32 * (l >> 4) + j]; hb[l][j] = B[32 * (l >> 4) + j][l & 15]; }
        hsa[l] = 127 - 3 * (l >> 4);            // A's block kq of every row scaled by 2^(-3 kq)
        hsb[l] = 127 + ((l >> 4) & 1);          // B's blocks 1 and 3 of every column by 2
    }
    void *da, *db, *dc, *dsa, *dsb;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dc, 64 * 16); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
    int bad = 0;
    for (int pass = 0; pass < 2; pass++) {
        // pass 0: every scale 127 (= 1.0), pass 1: the scales above
        if (pass == 0) { int ones[64]; for (int l = 0; l < 64; l++) ones[l] = 127; hipMemcpy(dsa, ones, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, ones, 256, hipMemcpyHostToDevice); }
        else { hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice); }
        hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, (const v8i*)da, (const v8i*)db, (v4f*)dc, (const int*)dsa, (const int*)dsb);
        float hc[64][4];
        hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
        // hypotheses for the k of byte j of lane group q: 0: 32 q + j;  1: 16 q + j (j < 16), 64 + 16 q + j - 16 (j >= 16)
        for (int hyp = 0; hyp < 2; hyp++) {
            int wrong = 0;
            for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
                const int n = l & 15, m = 4 * (l >> 4) + r;
                double want = 0;
                for (int q = 0; q < 4; q++) for (int j = 0; j < 32; j++) {
                    // the data were laid out under hypothesis 0: lane group q, byte j holds A[m][32 q + j] and B[32 q + j][n]
                    const int k = 32 * q + j;
                    // under hypothesis 1 the hardware pairs byte j of group q of A with byte j of group q of B all the same: the sum
                    // over (q, j) is identical unless scales differ per block -- so the hypotheses only differ in which scale applies
                    const int blk = hyp == 0 ? q : (j < 16 ? (q >> 1) : 2 + (q >> 1));
                    const double sa_ = pass ? exp2(-3.0 * blk) : 1.0, sb_ = pass ? exp2((double)(blk & 1)) : 1.0;
                    want += (double)e4m3_val(A[m][k]) * sa_ * e4m3_val(B[k][n]) * sb_;
                }
                if (fabs(want - hc[l][r]) > 1e-6 * fmax(1.0, fabs(want))) { if (wrong < 3 && hyp == 0) printf("  pass %d: D[%d][%d] = %g, want %g\n", pass, m, n, hc[l][r], want); wrong++; }
            }
            printf("pass %d (%s), hypothesis %d: %d of 256 wrong\n", pass, pass ? "block scales" : "scales 1.0", hyp, wrong);
            if (hyp == 0) bad += wrong;
        }
    }
    // ---- 2. cycles
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const double ghz = prop.clockRate * 1e-6;
    const int cus = prop.multiProcessorCount, iters = 20000;
    float* dout;
    hipMalloc(&dout, cus * 4 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"scale_f32_16x16x128 e4m3", "f32_16x16x32_f16", "2 x f16 + 1 x scaled e4m3 (x2)"};
    for (int which = 0; which < 3; which++) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(spin<0>, dim3(cus), dim3(256), 0, 0, dout, iters);
            else if (which == 1) hipLaunchKernelGGL(spin<1>, dim3(cus), dim3(256), 0, 0, dout, iters);
            else hipLaunchKernelGGL(spin<2>, dim3(cus), dim3(256), 0, 0, dout, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        const double cyc = best * 1e-3 * ghz * 1e9 / iters;
        printf("%s: %.2f ms, %.1f cycles per loop iteration per SIMD at %.2f GHz nominal (%s)\n", names[which], best, cyc, ghz,
               which == 2 ? "4 f16 + 2 scaled" : "4 instructions");
    }

    // ---- 3. the conversion: every finite non-negative half in steps, against the host's round-to-nearest-even e4m3
    const int n = 1 << 15;
    uint16_t* hin = (uint16_t*)malloc(n * 4);
    for (int i = 0; i < n; i++) { hin[2 * i] = (uint16_t)i; hin[2 * i + 1] = (uint16_t)(0x8000 | i); }     // all positive halves up to 0x7FFF (incl. inf / nan at the end)
    void *din, *dq;
    hipMalloc(&din, n * 4); hipMalloc(&dq, n * 4);
    hipMemcpy(din, hin, n * 4, hipMemcpyHostToDevice);
    for (int pass = 0; pass < 2; pass++) {
        const float scale = pass ? 0.25f : 1.0f;
        hipLaunchKernelGGL(cvt, dim3(n / 256), dim3(256), 0, 0, (const h2*)din, (s2*)dq, n, scale);
        uint16_t* hq = (uint16_t*)malloc(n * 4);
        hipMemcpy(hq, dq, n * 4, hipMemcpyDeviceToHost);
        int diff = 0, shown = 0;
        for (int i = 0; i < 0x7C00; i++) {
            _Float16 h; uint16_t u = (uint16_t)i; memcpy(&h, &u, 2);
            const float v = (float)h / scale;
            const uint8_t want = e4m3_of(v), got = (uint8_t)(hq[2 * i] & 0xFF);
            if (want != got) { diff++; if (shown++ < 6) printf("  half 0x%04x = %g (scale %g): got 0x%02x = %g, model 0x%02x = %g\n", i, (float)h, scale, got, e4m3_val(got), want, e4m3_val(want)); }
        }
        printf("v_cvt_scalef32_pk_fp8_f16, scale %g: %d of %d positive halves differ from e4m3(round-to-nearest-even(h / scale)), second byte of the pair is the negated value: %s\n",
               scale, diff, 0x7C00, ((hq[2 * 1000] >> 8) & 0xFF) == (0x80 | (hq[2 * 1000] & 0xFF)) ? "yes" : "no");
        free(hq);
    }
    return bad != 0;
}
