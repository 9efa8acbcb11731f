// which bits of HW_REG_HW_ID / HW_REG_XCC_ID tell the CUs of gfx950 apart?  One wave per workgroup, 256 * 8 workgroups that
// each hold enough LDS for ONE workgroup per CU, so that every resident workgroup sits on its own CU: the distinct
// values of a candidate key over one resident round must be 256.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <set>
__global__ void k(unsigned* out)
{
    extern __shared__ int pad[];
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hwid; out[2 * blockIdx.x + 1] = xcc; pad[0] = 1; }
    // stay resident for a while so that one round of workgroups really is one per CU
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 20000) { }
}
int main()
{
    const int N = 256;
    unsigned *d, h[2 * N];
    (void)hipMalloc(&d, sizeof h);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL(k, dim3(N), dim3(64), 100 * 1024, 0, d);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int lo = 0; lo < 32; lo += 4) {
        std::set<unsigned> keys;
        for (int i = 0; i < N; i++) keys.insert(((h[2 * i + 1] & 0xF) << 8) | ((h[2 * i] >> lo) & 0xFF));
        printf("xcc[3:0] + hw_id[%d:%d]: %zu distinct of %d workgroups\n", lo + 7, lo, keys.size(), N);
    }
    std::set<unsigned> x; for (int i = 0; i < N; i++) x.insert(h[2 * i + 1] & 0xF);
    printf("xcc ids: %zu distinct; sample hw_id 0x%08x 0x%08x 0x%08x xcc 0x%x 0x%x\n", x.size(), h[0], h[2], h[4], h[1], h[3]);
    return 0;
}
