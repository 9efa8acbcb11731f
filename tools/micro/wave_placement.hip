// Where the waves of a workgroup land: per workgroup size (3, 4, 6, 8 waves) and LDS request, the share of waves on each of a CU's
// four SIMDs, and how many workgroups are resident per CU.  (A 3- or 6-wave workgroup does not fill the SIMDs evenly; whether
// the NEXT workgroup of the CU starts where the last one ended decides whether that matters.)
//   hipcc -O3 --offload-arch=gfx950 tools/micro/wave_placement.hip -o /tmp/wave_placement && /tmp/wave_placement
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <vector>

__global__ void probe(unsigned* out, int spin, int vgpr_hog)
{
    extern __shared__ unsigned char lds[];
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = wall_clock64();
    float a = threadIdx.x;
    for (int i = 0; i < spin; i++) a = a * 1.0001f + 0.5f;          // ~spin x 8 cycles: long enough for the CU to fill up
    lds[threadIdx.x] = (unsigned char)a;
    const unsigned long long t1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        out[4 * w] = (xcc & 0xF) << 8 | ((hwid >> 8) & 0xFF);         // CU key
        out[4 * w + 1] = (hwid >> 4) & 3;                              // SIMD
        out[4 * w + 2] = (unsigned)t0;
        out[4 * w + 3] = (unsigned)t1;
    }
}

int main()
{
    const int cases[][2] = {{192, 37632}, {384, 75264}, {256, 75264}, {512, 75264}, {256, 80448}, {256, 81920}, {256, 54144}, {256, 53760}, {256, 53248}, {256, 52224}, {256, 51200},
                            {256, 40960}, {256, 40448}, {256, 39936}, {256, 38912}};
    unsigned* d;
    hipMalloc(&d, 1 << 24);
    for (auto& c : cases) {
        const int threads = c[0], lds = c[1], waves = threads / 64, blocks = 256 * 40;
        hipMemset(d, 0, 1 << 24);
        hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), lds, 0, d, 4000, 0);
        hipDeviceSynchronize();
        std::vector<unsigned> h((size_t)4 * blocks * waves);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        double simd[4] = {0, 0, 0, 0};
        std::map<unsigned, std::vector<std::pair<unsigned, int>>> ev;
        for (int w = 0; w < blocks * waves; w++) {
            simd[h[4 * w + 1]] += 1;
            if (w % waves == 0) { ev[h[4 * w]].push_back({h[4 * w + 2], +1}); ev[h[4 * w]].push_back({h[4 * w + 3], -1}); }
        }
        double at[8] = {0}, span = 0;
        for (auto& cu : ev) {
            std::sort(cu.second.begin(), cu.second.end());
            int live = 0;
            for (size_t k = 0; k + 1 < cu.second.size(); k++) {
                live += cu.second[k].second;
                at[live < 7 ? live : 7] += (double)(cu.second[k + 1].first - cu.second[k].first);
            }
            span += (double)(cu.second.back().first - cu.second.front().first);
        }
        double mean = 0;
        for (int k = 0; k < 8; k++) mean += k * at[k] / span;
        const double tot = simd[0] + simd[1] + simd[2] + simd[3];
        printf("%d waves, %6d B of LDS: waves per SIMD %.3f %.3f %.3f %.3f of the total; %.2f workgroups resident per CU (%zu CUs)\n", waves, lds,
               simd[0] / tot, simd[1] / tot, simd[2] / tot, simd[3] / tot, mean, ev.size());
    }
    return 0;
}
