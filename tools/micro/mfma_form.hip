// Does an i8 MFMA overlap with VALU work of the other waves on its SIMD, and does it matter whether its accumulator
// lives in VGPRs or AGPRs?  512-thread workgroups: waves 0-3 run VALU chains, waves 4-7 MFMA chains (wave w -> SIMD w % 4).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_form.hip -o tools/micro/mfma_form && tools/micro/mfma_form
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int FORM>
__global__ __launch_bounds__(512) void spin(int* out, int iters, int mode)
{
    int acc = 0;
    if (mode == 1 || (mode == 2 && threadIdx.x < 256)) {
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
        v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int i = 0; i < iters; i++) {
            if constexpr (FORM == 0) {      // accumulators in VGPRs
                asm volatile("v_mfma_i32_16x16x64_i8 %0, %4, %5, %0\n v_mfma_i32_16x16x64_i8 %1, %4, %5, %1\n"
                             "v_mfma_i32_16x16x64_i8 %2, %4, %5, %2\n v_mfma_i32_16x16x64_i8 %3, %4, %5, %3\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
            } else if constexpr (FORM == 1) {   // accumulators in AGPRs
                asm volatile("v_mfma_i32_16x16x64_i8 %0, %4, %5, %0\n v_mfma_i32_16x16x64_i8 %1, %4, %5, %1\n"
                             "v_mfma_i32_16x16x64_i8 %2, %4, %5, %2\n v_mfma_i32_16x16x64_i8 %3, %4, %5, %3\n"
                             : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "v"(a), "v"(b));
            } else if constexpr (FORM == 2) {   // everything in AGPRs
                asm volatile("v_mfma_i32_16x16x64_i8 %0, %4, %5, %0\n v_mfma_i32_16x16x64_i8 %1, %4, %5, %1\n"
                             "v_mfma_i32_16x16x64_i8 %2, %4, %5, %2\n v_mfma_i32_16x16x64_i8 %3, %4, %5, %3\n"
                             : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "a"(a), "a"(b));
            } else {                            // VGPR destination, constant 0 as C (no accumulator read)
                asm volatile("v_mfma_i32_16x16x64_i8 %0, %4, %5, 0\n v_mfma_i32_16x16x64_i8 %1, %4, %5, 0\n"
                             "v_mfma_i32_16x16x64_i8 %2, %4, %5, 0\n v_mfma_i32_16x16x64_i8 %3, %4, %5, 0\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
            }
        }
        acc = c0[0] + c1[1] + c2[2] + c3[3];
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
    const char* names[4] = {"acc in VGPRs", "acc in AGPRs", "A, B and acc in AGPRs", "VGPR dst, C = 0"};
    for (int form = 0; form < 4; form++)
        for (int mode = 0; mode < 3; mode++) {
            const int blocks = cus * 4, threads = mode == 2 ? 512 : 256;
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (form == 0) hipLaunchKernelGGL(spin<0>, dim3(blocks), dim3(threads), 0, 0, d, iters, mode);
                else if (form == 1) hipLaunchKernelGGL(spin<1>, dim3(blocks), dim3(threads), 0, 0, d, iters, mode);
                else if (form == 2) hipLaunchKernelGGL(spin<2>, dim3(blocks), dim3(threads), 0, 0, d, iters, mode);
                else hipLaunchKernelGGL(spin<3>, dim3(blocks), dim3(threads), 0, 0, d, iters, mode);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            const double cyc = best * 1e-3 * ghz * 1e9;
            if (mode == 0) printf("%s: MFMA alone %.2f ms (%.1f cycles each)", names[form], best, cyc / (4.0 * iters * 4));
            else if (mode == 1) printf("   VALU alone %.2f ms", best);
            else printf("   both %.2f ms\n", best);
        }
    return 0;
}
