// check of the DPP row reduction used by the median kernel's tile classification: min / max / sum of one value per lane
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(const unsigned* in, unsigned* out)
{
    unsigned lo = in[threadIdx.x], hi = lo, sum = lo;
#define CK_DPP(V, CTRL) (unsigned)__builtin_amdgcn_update_dpp((int)(V), (int)(V), CTRL, 0xF, 0xF, false)
#define CK_ROW_REDUCE(V, OP)                                                     \
    { unsigned t_; t_ = CK_DPP(V, 0xB1); V = OP(V, t_); t_ = CK_DPP(V, 0x4E); V = OP(V, t_);   \
      t_ = CK_DPP(V, 0x141); V = OP(V, t_); t_ = CK_DPP(V, 0x140); V = OP(V, t_); }
#define CK_MIN(a, b) ((a) < (b) ? (a) : (b))
#define CK_MAX(a, b) ((a) > (b) ? (a) : (b))
#define CK_ADD(a, b) ((a) + (b))
    CK_ROW_REDUCE(lo, CK_MIN)
    CK_ROW_REDUCE(hi, CK_MAX)
    CK_ROW_REDUCE(sum, CK_ADD)
    unsigned wlo = 255u, whi = 0u, wsum = 0u;
    for (int rr = 0; rr < 4; rr++) {
        const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)lo, 16 * rr), b = (unsigned)__builtin_amdgcn_readlane((int)hi, 16 * rr);
        wlo = CK_MIN(wlo, a); whi = CK_MAX(whi, b);
        wsum += (unsigned)__builtin_amdgcn_readlane((int)sum, 16 * rr);
    }
    if (threadIdx.x == 0) { out[0] = wlo; out[1] = whi; out[2] = wsum; }
    out[4 + threadIdx.x] = lo; out[68 + threadIdx.x] = sum;
}
int main()
{
    unsigned h[64], *d, *o, r[132];
    unsigned lo = 255, hi = 0, sum = 0;
    srand(3);
    for (int i = 0; i < 64; i++) { h[i] = 60 + rand() % 40; lo = h[i] < lo ? h[i] : lo; hi = h[i] > hi ? h[i] : hi; sum += h[i]; }
    (void)hipMalloc(&d, sizeof h); (void)hipMalloc(&o, sizeof r);
    (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o);
    (void)hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    printf("min %u (want %u)  max %u (want %u)  sum %u (want %u)\n", r[0], lo, r[1], hi, r[2], sum);
    printf("row mins per lane: "); for (int i = 0; i < 64; i += 5) printf("%u ", r[4 + i]); printf("\n");
    return !(r[0] == lo && r[1] == hi && r[2] == sum);
}
