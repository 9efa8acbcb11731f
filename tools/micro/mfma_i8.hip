// Issue cost of the i8 MFMA shapes on this GPU (cycles per instruction per SIMD at the measured clock), alone and
// with VALU work in the other waves of the SIMD.  Calibrates the matrix-core median (k_median.hip).
// Mode 3 (round 5): what the VALU waves KEEP of their rate while the matrix pipe of their SIMD is busy
// with each shape -- the VALU waves run until the last MFMA wave has finished and count their operations.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_i8.hip -o tools/micro/mfma_i8 && tools/micro/mfma_i8
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ unsigned int g_done;            // mode 3: MFMA waves that have finished
__device__ unsigned long long g_valu_ops;  // mode 3: VALU wave-instructions issued while MFMA waves were running
__device__ unsigned long long g_mfma_ticks;// mode 3: sum over MFMA waves of their loop's duration (100 MHz ticks)

template <int WHICH>
__global__ __launch_bounds__(512) void spin(int* out, int iters, int mode)
{
    if (mode == 3 && threadIdx.x < 256) {      // VALU waves: the same v_perm chains, until every MFMA wave is done
        int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
        const unsigned int n_mfma_waves = gridDim.x * 4;
        unsigned long long n = 0;
        // (bounded: ~2 s of polling at most, should part of the grid not be resident -- then the MFMA waves it waits for never start)
        while (__hip_atomic_load(&g_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_mfma_waves && n < 256ull * 4000000ull) {
            for (int i = 0; i < 16; i++) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    a0 = __builtin_amdgcn_perm(a0, a1, 0x05040100);
                    a1 = __builtin_amdgcn_perm(a1, a2, 0x05040100);
                    a2 = __builtin_amdgcn_perm(a2, a3, 0x05040100);
                    a3 = __builtin_amdgcn_perm(a3, a0, 0x05040100);
                }
            }
            n += 16 * 16;
        }
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_valu_ops, n);
        out[blockIdx.x * 512 + threadIdx.x] = a0 + a1 + a2 + a3;
        return;
    }
    const unsigned long long t_begin = wall_clock64();
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
    if (mode == 3 && (threadIdx.x & 63) == 0) {
        atomicAdd(&g_mfma_ticks, wall_clock64() - t_begin);
        __hip_atomic_fetch_add(&g_done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
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
        for (int mode = 0; mode < 4; mode++) {
            // 2 workgroups per CU: 2 waves of a kind on every SIMD, and every workgroup resident at once whatever the shape's
            // register count (round 4 ran 4 per CU: with the 32x32 shapes' 64 accumulator registers the fourth waited its turn)
            const int blocks = cus * 2;
            const int threads = mode >= 2 ? 512 : 256;
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                unsigned int zero = 0;
                unsigned long long zero64 = 0;
                hipMemcpyToSymbol(HIP_SYMBOL(g_done), &zero, sizeof zero);
                hipMemcpyToSymbol(HIP_SYMBOL(g_valu_ops), &zero64, sizeof zero64);
                hipMemcpyToSymbol(HIP_SYMBOL(g_mfma_ticks), &zero64, sizeof zero64);
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
            const double n_mfma = 2.0 * iters * 4, n_valu = 2.0 * iters * 16;     // per SIMD
            if (mode == 0)
                printf("%s alone: %.2f ms  %.1f cycles per MFMA per SIMD  %.0f TOP/s\n", names[which], best, cyc / n_mfma,
                       2 * macs[which] * n_mfma * cus * 4 / best * 1e-9);
            else if (mode == 1)
                printf("  VALU alone: %.2f ms  %.2f cycles per op per SIMD\n", best, cyc / n_valu);
            else if (mode == 2)
                printf("  both on every SIMD: %.2f ms (sum of the two alone = no overlap, max = full overlap)\n", best);
            else {
                unsigned long long ops = 0, ticks = 0;
                hipMemcpyFromSymbol(&ops, HIP_SYMBOL(g_valu_ops), sizeof ops);
                hipMemcpyFromSymbol(&ticks, HIP_SYMBOL(g_mfma_ticks), sizeof ticks);
                const double mfma_s = (double)ticks * 1e-8 / (blocks * 4.0);            // mean duration of an MFMA wave's loop
                const double simds = cus * 4.0;
                // per SIMD: one MFMA wave (4 x iters MFMAs) and one VALU wave of each of the 2 workgroups of the CU
                printf("  VALU beside MFMA: MFMA waves %.2f ms each = %.1f cycles per MFMA per SIMD; the VALU waves issued %.3g ops in %.2f ms = "
                       "%.2f cycles per op per SIMD\n", mfma_s * 1e3, mfma_s * ghz * 1e9 / n_mfma, (double)ops, best,
                       best * 1e-3 * ghz * 1e9 / ((double)ops / simds));
                if ((double)ops / (blocks * 4.0) >= 256.0 * 4000000.0) printf("  (!) the VALU waves ran into their bound: part of the grid was not resident\n");
            }
        }
    return 0;
}
