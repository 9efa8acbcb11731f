// Sustained f32 MFMA rate of this GPU: back-to-back v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 on
// independent accumulators, no memory traffic.  Calibrates the "mfma" roofline used by bench.py.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak && tools/micro/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WHICH>
__global__ __launch_bounds__(256) void spin(float* out, int iters)
{
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    if constexpr (WHICH == 0) {
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters; i++) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
        }
        out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters; i++) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        }
        out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    }
}

int main()
{
    float* d;
    hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 2048, iters = 20000;
    for (int which = 0; which < 2; which++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(spin<0>, dim3(blocks), dim3(256), 0, 0, d, iters);
            else hipLaunchKernelGGL(spin<1>, dim3(blocks), dim3(256), 0, 0, d, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)blocks * 4 * iters * 4 * (which == 0 ? 4096.0 : 2048.0);
            printf("%s rep %d: %.2f ms  %.1f TFLOP/s\n", which == 0 ? "32x32x2f32" : "16x16x4f32", rep, ms, flops / ms * 1e-9);
        }
    }
    return 0;
}
