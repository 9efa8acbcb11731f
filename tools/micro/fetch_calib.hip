// What does rocprofv3's FETCH_SIZE report for a streaming read of KNOWN size, per load width?  (measurement:
// the median and NMS kernels' FETCH_SIZE read 3.04 MB per 1080p frame for 6.2 MB inputs; MI355X_MICROARCH.md documents the
// halving for 16-byte-per-lane streams only.)  Each kernel streams its own 512 MiB region once, with the load shape named:
//   b8x3   one byte per lane at a 3-byte stride (the median kernel's old column-per-lane byte loads of interleaved BGR)
//   b32    one dword per lane, coalesced (the NMS kernel's old staging)
//   b64 / b128   8 / 16 bytes per lane, coalesced
// Run:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- tools/micro/fetch_calib
// then  python3 tools/pmc_kernels.py out read_b    (counter in KiB; expected 524288 KiB per kernel if it counts every byte)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
constexpr size_t REGION = 512ull << 20;

__global__ void read_b8x3(const uint8_t* p, size_t n, unsigned* sink)
{
    // a wave reads 192 consecutive bytes as 3 byte-loads per lane (offsets 3 lane + c): every byte once
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) / 64, lane = threadIdx.x & 63;
    const size_t nwaves = (size_t)gridDim.x * blockDim.x / 64;
    unsigned acc = 0;
    for (size_t base = wave * 192; base + 192 <= n; base += nwaves * 192)
        acc += p[base + 3 * lane] + p[base + 3 * lane + 1] + p[base + 3 * lane + 2];
    if (acc == 0xFFFFFFFFu) *sink = acc;
}
template <typename T>
__global__ void read_wide(const T* p, size_t n, unsigned* sink)
{
    const size_t i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (size_t i = i0; i < n; i += step) {
        const T v = p[i];
        const unsigned* u = reinterpret_cast<const unsigned*>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; k++) acc += u[k];
    }
    if (acc == 0xFFFFFFFFu) *sink = acc;
}
int main()
{
    uint8_t* buf;
    unsigned* sink;
    if (hipMalloc(&buf, 4 * REGION) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, 4 * REGION);
    (void)hipDeviceSynchronize();
    const dim3 grid(256 * 8), block(256);
    hipLaunchKernelGGL(read_b8x3, grid, block, 0, 0, buf, REGION, sink);
    hipLaunchKernelGGL(read_wide<uint32_t>, grid, block, 0, 0, (const uint32_t*)(buf + REGION), REGION / 4, sink);
    hipLaunchKernelGGL(read_wide<uint2>, grid, block, 0, 0, (const uint2*)(buf + 2 * REGION), REGION / 8, sink);
    hipLaunchKernelGGL(read_wide<uint4>, grid, block, 0, 0, (const uint4*)(buf + 3 * REGION), REGION / 16, sink);
    const hipError_t e = hipDeviceSynchronize();
    printf("fetch_calib: %s; each kernel streamed %zu KiB\n", hipGetErrorString(e), REGION >> 10);
    return e != hipSuccess;
}
