// k_records.hip -- per-frame result records (ck_frame_record, include/camkifu_amd.h) written IN PLACE in HBM.
// The two halves of a record come from two contexts (board path, stones path) on two streams at once; they touch
// disjoint bytes: [0, 536) and [536, 1436).  A record is 1440 bytes, so every half starts 8-byte aligned.
#include <stddef.h>

#include "ck_common.h"

static_assert(sizeof(ck_frame_record) == 1440, "ck_frame_record has implicit padding");
static_assert(offsetof(ck_frame_record, region_conf) == 536 && offsetof(ck_frame_record, region_label) == 1336, "record layout");

#define CK_REC_BOARD_BYTES 536

// one wave per record: 134 dwords of the packed part -> the head of the record
__global__ void __launch_bounds__(256) records_put_board_kernel(const uint32_t* __restrict__ parts, int n, uint32_t* __restrict__ rec)
{
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (f >= n) return;
    const uint32_t* src = parts + (size_t)f * (CK_REC_BOARD_BYTES / 4);
    uint32_t* dst = rec + (size_t)f * (sizeof(ck_frame_record) / 4);
    for (int i = lane; i < CK_REC_BOARD_BYTES / 4; i += 64) dst[i] = src[i];
}

// one wave per record: 100 doubles and 100 bytes (as 25 dwords: both ends are 4-byte aligned)
__global__ void __launch_bounds__(256) records_put_regions_kernel(const uint32_t* __restrict__ rlabel, const double* __restrict__ rconf,
                                                                  int n, uint8_t* __restrict__ rec)
{
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (f >= n) return;
    uint8_t* r = rec + (size_t)f * sizeof(ck_frame_record);
    double* conf = (double*)(r + offsetof(ck_frame_record, region_conf));
    uint32_t* lab = (uint32_t*)(r + offsetof(ck_frame_record, region_label));
    for (int i = lane; i < 100; i += 64) conf[i] = rconf[(size_t)f * 100 + i];
    if (lane < 25) lab[lane] = rlabel[(size_t)f * 25 + lane];
}

int k_records_put_board(ck_ctx* ctx, const uint8_t* d_parts, int n, ck_frame_record* d_rec)
{
    if (((uintptr_t)d_rec & 7) != 0) return ck_fail(ctx, CK_ERR_ARG, "records must be 8-byte aligned");
    hipLaunchKernelGGL(records_put_board_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, (const uint32_t*)d_parts, n, (uint32_t*)d_rec);
    CK_HIP(ctx, hipGetLastError());
    return CK_OK;
}

int k_records_put_regions(ck_ctx* ctx, const uint8_t* d_rlabel, const double* d_rconf, int n, ck_frame_record* d_rec)
{
    if (((uintptr_t)d_rec & 7) != 0) return ck_fail(ctx, CK_ERR_ARG, "records must be 8-byte aligned");
    hipLaunchKernelGGL(records_put_regions_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, (const uint32_t*)d_rlabel, d_rconf, n, (uint8_t*)d_rec);
    CK_HIP(ctx, hipGetLastError());
    return CK_OK;
}
