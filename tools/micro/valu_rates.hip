// Issue cost of the VALU instructions the median kernel is made of, on gfx950: each kernel runs ITERS x 32
// independent copies of ONE instruction per wave (8 chains x 4 unrolled), 4 waves per SIMD resident.
// Prints cycles per wave-instruction per SIMD (1.0 = one wave64 instruction issued every cycle).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rates.hip -o tools/micro/valu_rates && tools/micro/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHAINS 8
template <int OP>
__global__ __launch_bounds__(256) void spin(uint32_t* out, int iters, uint32_t k1, uint32_t k2)
{
    uint32_t x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = threadIdx.x * 2654435761u + c * 97u + blockIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                uint32_t v = x[c];
                // the trailing xor keeps the compiler from folding the four unrolled copies into one
                if (OP == 0) v = (v + k1) ^ k2;                                            // v_add_u32 + v_xor
                else if (OP == 1) v = (v * k1) ^ k2;                                       // v_mul_lo_u32 + v_xor
                else if (OP == 2) v = __builtin_amdgcn_lerp(v, k1, k2);                    // v_lerp_u8
                else if (OP == 3) v = __builtin_amdgcn_perm(v, k1, k2);                    // v_perm_b32
                else if (OP == 4) v = (v + k1 + k2) ^ k1;                                  // v_add3_u32 + v_xor
                else if (OP == 5) v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true) + k1;   // dpp mov (+ add)
                else if (OP == 6) v = __builtin_amdgcn_alignbyte(v, k1, 2);                // v_alignbyte_b32
                else if (OP == 7) v = ((v << 8) + v) ^ k2;                                 // v_lshl_add_u32 + v_xor
                else if (OP == 8) v = (uint32_t)(((uint64_t)v * k1 + k2)) ^ k1;            // v_mad_u64_u32 + v_xor
                else if (OP == 9) v = ((~v & k1) | k2) + k1;                               // bitop3 + add
                else if (OP == 10) v = ((v >> 7) & k1) + k2;                               // shift, and, add
                x[c] = v;
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s ^= x[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
static void run(const char* name, uint32_t* d, int ninstr_per_iter)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4, iters = 4000;       // 4 blocks of 4 waves per CU -> 4 waves per SIMD
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(spin<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 0x01010101u + rep, 0x0F0F0F0Fu);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    // per SIMD: 4 waves x iters x 32 op-groups
    const double wave_instr = 4.0 * iters * 32.0 * ninstr_per_iter;
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-28s %7.3f ms  %6.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms, cycles / wave_instr);
}

int main()
{
    uint32_t* d;
    hipMalloc(&d, 1024 * 256 * 4);
    run<0>("v_add_u32 + v_xor", d, 2);
    run<1>("v_mul_lo_u32 + v_xor", d, 2);
    run<2>("v_lerp_u8", d, 1);
    run<3>("v_perm_b32", d, 1);
    run<4>("v_add3_u32 + v_xor", d, 2);
    run<5>("v_mov_dpp + v_add (2 instr)", d, 2);
    run<6>("v_alignbyte_b32", d, 1);
    run<7>("v_lshl_add_u32 + v_xor", d, 2);
    run<8>("v_mad_u64_u32 + v_xor", d, 2);
    run<9>("and-not-or (bitop3) + add", d, 2);
    run<10>("shift + and + add (3 instr)", d, 3);
    return 0;
}
