// Issue cost of the i8 MFMA shapes on this GPU (cycles per instruction per SIMD at the measured clock), alone and
// with VALU work in the other waves of the SIMD.  Calibrates the matrix-core median (k_median.hip).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_i8.hip -o tools/micro/mfma_i8 && tools/micro/mfma_i8
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int WHICH>
__global__ __launch_bounds__(512) void spin(int* out, int iters, int mode)
{
    // mode 0: every workgroup issues MFMAs; 1: every workgroup issues VALU ops; 2: 512-thread workgroups, waves 0-3 VALU and waves 4-7 MFMA (wave w runs on SIMD w % 4)
    // (both kinds then share every SIMD)
    int acc = 0;
    if (mode == 1 || (mode == 2 && threadIdx.x < 256)) {                      // waves that only do VALU work (v_med3_i32 chains)
        int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                a0 = __builtin_amdgcn_perm(a0, a1, 0x05040100);
                a1 = __builtin_amdgcn_perm(a1, a2, 0x05040100);
                a2 = __builtin_amdgcn_perm(a2, a3, 0x05040100);
                a3 = __builtin_amdgcn_perm(a3, a0, 0x05040100);
            }
        }
        acc = a0 + a1 + a2 + a3;
    } else {
        v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {(int)blockIdx.x, 5, 6, 7};
        long a8 = threadIdx.x * 0x0101010101010101l, b8 = blockIdx.x * 0x0101010101010101l;
        if constexpr (WHICH == 0) {
            v4i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
            for (int i = 0; i < iters; i++) {
                c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
            }
            acc = c0[0] + c1[1] + c2[2] + c3[3];
        } else if constexpr (WHICH == 1) {
            v4i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
            for (int i = 0; i < iters; i++) {
                c0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, b8, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, b8, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, b8, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, b8, c3, 0, 0, 0);
            }
            acc = c0[0] + c1[1] + c2[2] + c3[3];
        } else if constexpr (WHICH == 2) {
            v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
            for (int i = 0; i < iters; i++) {
                c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
            }
            acc = c0[0] + c1[1] + c2[2] + c3[3];
        } else {
            v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
            for (int i = 0; i < iters; i++) {
                c0 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a8, b8, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a8, b8, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a8, b8, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a8, b8, c3, 0, 0, 0);
            }
            acc = c0[0] + c1[1] + c2[2] + c3[3];
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}

int main()
{
    int* d;
    hipMalloc(&d, 8192 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const double ghz = prop.clockRate * 1e-6;
    const int cus = prop.multiProcessorCount;
    const int iters = 20000;
    const char* names[4] = {"i32_16x16x64_i8", "i32_16x16x32_i8", "i32_32x32x32_i8", "i32_32x32x16_i8"};
    const double macs[4] = {16. * 16 * 64, 16. * 16 * 32, 32. * 32 * 32, 32. * 32 * 16};
    printf("%s  %d CUs  %.2f GHz\n", prop.name, cus, ghz);
    for (int which = 0; which < 4; which++)
        for (int mode = 0; mode < 3; mode++) {
            // 4 workgroups of 4 waves per CU and kind: 4 waves of that kind on every SIMD
            const int blocks = cus * 4;
            const int threads = mode == 2 ? 512 : 256;
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (which == 0) hipLaunchKernelGGL(spin<0>, dim3(blocks), dim3(threads), 0, 0, d, iters, mode);
                else if (which == 1) hipLaunchKernelGGL(spin<1>, dim3(blocks), dim3(threads), 0, 0, d, iters, mode);
                else if (which == 2) hipLaunchKernelGGL(spin<2>, dim3(blocks), dim3(threads), 0, 0, d, iters, mode);
                else hipLaunchKernelGGL(spin<3>, dim3(blocks), dim3(threads), 0, 0, d, iters, mode);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            const double cyc = best * 1e-3 * ghz * 1e9;
            const double n_mfma = 4.0 * iters * 4, n_valu = 4.0 * iters * 16;     // per SIMD
            if (mode == 0)
                printf("%s alone: %.2f ms  %.1f cycles per MFMA per SIMD  %.0f TOP/s\n", names[which], best, cyc / n_mfma,
                       2 * macs[which] * n_mfma * cus * 4 / best * 1e-9);
            else if (mode == 1)
                printf("  VALU alone: %.2f ms  %.2f cycles per op per SIMD\n", best, cyc / n_valu);
            else
                printf("  both on every SIMD: %.2f ms (sum of the two alone = no overlap, max = full overlap)\n", best);
        }
    return 0;
}
