// Do MFMA waves and VALU waves resident on the same SIMD overlap?  One workgroup = 8 waves:
// waves 0..3 issue back-to-back f32 MFMAs, waves 4..7 issue plain integer VALU work.
//   mode 0: only the MFMA waves work, mode 1: only the VALU waves, mode 2: both.
// If T(2) ~ max(T(0), T(1)) the matrix pipe and the vector ALU co-execute; if T(2) ~ T(0)+T(1) they do not.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/coexec.hip -o tools/micro/coexec && tools/micro/coexec
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void coexec(float* out, int mode, int iters_m, int iters_v)
{
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (mode == 1) return;
        float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
        f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters_m; i++) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        }
        out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        if (mode == 0) return;
        unsigned x0 = threadIdx.x, x1 = blockIdx.x, x2 = 7, x3 = 11;
        for (int i = 0; i < iters_v; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                x0 = __builtin_amdgcn_lerp(x0, x1, 0x01010101u) + 3u;
                x1 = (x1 ^ x2) + (x3 >> 7);
                x2 = __builtin_amdgcn_perm(x2, x3, 0x03020100u) & 0x7f7f7f7fu;
                x3 = x3 + x0;
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = (float)(x0 ^ x1 ^ x2 ^ x3);
    }
}

int main()
{
    float* d;
    hipMalloc(&d, 8192 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 2048;
    const int iters_m = 4000;       // 16000 MFMAs x 32 cycles = 512k cycles per MFMA wave
    for (int iters_v = 1000; iters_v <= 4000; iters_v *= 2) {
        float t[3];
        for (int mode = 0; mode < 3; mode++) {
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(coexec, dim3(blocks), dim3(512), 0, 0, d, mode, iters_m, iters_v);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&t[mode], e0, e1);
            }
        }
        printf("iters_v %d: mfma-only %.2f ms, valu-only %.2f ms, both %.2f ms (sum %.2f, max %.2f)\n", iters_v, t[0], t[1], t[2],
               t[0] + t[1], t[0] > t[1] ? t[0] : t[1]);
    }
    return 0;
}
