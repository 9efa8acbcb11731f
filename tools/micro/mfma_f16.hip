// Sustained rate of v_mfma_f32_16x16x32_f16 (the split-precision convolutions' instruction) on this GPU:
//   0: 8 independent accumulators, back to back        1: 6 accumulators, each hit three times in a row (the kernels' order)
//   2: as 1 with two ds_read_b128 per three MFMAs (the A fragments), 4 waves per SIMD
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f16.hip -o tools/micro/mfma_f16 && tools/micro/mfma_f16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int WHICH>
__global__ __launch_bounds__(256) void spin(float* out, int iters)
{
    __shared__ uint4 buf[2048];
    h8 a, b;
    for (int k = 0; k < 8; k++) { a[k] = (_Float16)(threadIdx.x * 1e-3f + k); b[k] = (_Float16)(blockIdx.x * 1e-3f - k); }
    for (int i = threadIdx.x; i < 2048; i += 256) buf[i] = make_uint4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    f32x4 c[8];
    for (int k = 0; k < 8; k++) c[k] = f32x4{0, 0, 0, 0};
    if constexpr (WHICH == 0) {
        for (int i = 0; i < iters; i++)
#pragma unroll
            for (int k = 0; k < 8; k++) c[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[k], 0, 0, 0);
    } else if constexpr (WHICH == 1) {
        for (int i = 0; i < iters; i++)
#pragma unroll
            for (int k = 0; k < 6; k++) {
                c[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[k], 0, 0, 0);
                c[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c[k], 0, 0, 0);
                c[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c[k], 0, 0, 0);
            }
    } else {
        int idx = threadIdx.x;
        for (int i = 0; i < iters; i++)
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const h8 x = __builtin_bit_cast(h8, buf[(idx + 64 * k) & 2047]);
                const h8 y = __builtin_bit_cast(h8, buf[(idx + 64 * k + 1024) & 2047]);
                idx += 7;
                c[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(y, b, c[k], 0, 0, 0);
                c[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, a, c[k], 0, 0, 0);
                c[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, b, c[k], 0, 0, 0);
            }
    }
    float s = 0;
    for (int k = 0; k < 8; k++) s += c[k][0] + c[k][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    float* d;
    hipMalloc(&d, 8192 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4 * 4, iters = 4000;      // 16 waves per CU = 4 per SIMD
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    for (int which = 0; which < 3; which++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(spin<0>, dim3(blocks), dim3(256), 0, 0, d, iters);
            else if (which == 1) hipLaunchKernelGGL(spin<1>, dim3(blocks), dim3(256), 0, 0, d, iters);
            else hipLaunchKernelGGL(spin<2>, dim3(blocks), dim3(256), 0, 0, d, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double n = (double)blocks * 4 * iters * (which == 0 ? 8 : 18);        // wave-level MFMAs
            printf("variant %d rep %d: %.2f ms  %.1f TFLOP/s  %.2f ns per MFMA per SIMD (%.1f cycles at the nominal %d MHz)\n", which, rep, ms,
                   n * 16384.0 / ms * 1e-9, ms * 1e6 / (n / 1024.0), ms * 1e6 / (n / 1024.0) * clk * 1e-6, clk / 1000);
        }
    }
    return 0;
}
