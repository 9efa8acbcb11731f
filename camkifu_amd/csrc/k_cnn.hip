// k_cnn.hip -- K10..K12: the stone classifier of SfNeural.
//   K10 patch extraction  NNManager._get_x            (reference: src/camkifu/stone/nn_manager.py:216-218, 256-275)
//   K11 the Keras net     NNManager.create_net        (nn_manager.py:277-298)
//   K12 decode            NNCache.predict_all_stones  (src/camkifu/stone/nn_cache.py:25-41, nn_manager.py:246-254)
//
// The four convolutions and the first dense layer are dense contractions and run on the
// matrix cores as implicit GEMMs (M = output pixels, N = output channels, K = kh*kw*cin):
//   fp32 mode: v_mfma_f32_16x16x4_f32 -- exact f32; K is accumulated in (kh, kw, cin) order like
//              the scalar CPU chain.  16x16 tiles because (a) 196- and 144-pixel, 90-channel layers
//              pad far less than in 32x32 tiles and (b) the 16x16x4 form sustains 155 TFLOP/s on
//              this part against 138 for 32x32x2 (tools/micro/mfma_peak.hip).  f32 MFMA runs on the
//              vector ALUs (same 157.3 TFLOP/s peak): it does not overlap VALU work of other waves
//              (tools/micro/coexec.hip), so MFMA-bound and VALU-bound kernels simply add up.
//   bf16 mode: k_cnn_bf16.hip (conv1 + conv2 and conv3 + conv4 fused, one bf16 MFMA per product); here only its dense tail.
// A (activations) is staged in LDS with pixel / row strides chosen so that one ds_read_b32 of a
// half-wave hits 32 distinct banks; B (weights) streams from L2 as coalesced dwordx4 fragments and
// is reused by the R register-blocked pixel tiles of a wave.  Wave counts per workgroup are
// multiples of 4 so every SIMD carries the same number of MFMA chains.  Tried and dropped: persistent
// workgroups with the next unit's rows prefetched into registers (the two workgroups of a CU then run
// in lock step and lose the staging / compute overlap that independent workgroups have: conv2 +5 %),
// explicit software pipelining of the LDS reads (no change with 4 waves per SIMD), two patches per
// workgroup for the odd tile counts of conv3 / conv4 (+4 %).
// Everything else (160->81 dense, softmax, base-3 decode) is byte/float VALU work.
#include <math.h>

#include <algorithm>

#include <type_traits>

#include "ck_common.h"

#ifndef C3_WM
#define C3_WM 2
#endif
#ifndef C4_WM
#define C4_WM 2
#endif
#ifndef H2_STAGE_UNROLL
#define H2_STAGE_UNROLL 8
#endif
#ifndef H2C2_SWZ
#define H2C2_SWZ 1      // conv2: swizzled LDS tile, 24 pixel tiles per block (3 per wave); 0 = padded tile, 16 per block
#endif
#ifndef H2_FC1
#define H2_FC1 1         // first dense layer in split precision too
#endif
#ifndef H2_FUSE34
#define H2_FUSE34 1      // conv3 and conv4 in one workgroup, conv3's output stays in LDS
#endif
#ifndef H2_FUSE1
#define H2_FUSE1 1       // conv1 computed inside conv2 (needs H2C2_SWZ)
#endif
#ifndef H2C2S_PF
#define H2C2S_PF 1      // 3 x 2 accumulator tiles leave room for a 2-slot weight ring only (PF 2: 141 VGPRs, 3 waves per SIMD, 17.3 us)
#endif
#ifndef H2C2S_SB
#define H2C2S_SB true
#endif
#ifndef H2C3_PF
#define H2C3_PF 2
#endif
#ifndef H2C3_SB
#define H2C3_SB false
#endif
#ifndef H2C2_PF
#define H2C2_PF 2
#endif
#ifndef H2C34_PF
#define H2C34_PF 2
#endif
#ifndef H2C2_TB
#define H2C2_TB 16
#endif
#ifndef H2C2_WM
#define H2C2_WM 8
#endif
constexpr float H2_WSCALE = 256.f;      // split-precision mode: weights are stored x 2^8 (their lo halves stay normal fp16)
#ifndef C1_R
#define C1_R 3
#endif
#ifndef C1_GRID
#define C1_GRID (256 * 3)      // persistent workgroups: three per CU
#endif
#ifndef C2_TB
#define C2_TB 16
#endif
#ifndef C2_WM
#define C2_WM 4
#endif
#ifndef C2_RN
#define C2_RN 1
#endif

#pragma clang fp contract(off)

#ifndef H2C2S_TB
#define H2C2S_TB 24      // pooling tiles (4x4 output pixels) per conv2 workgroup: 24 = 3 tile rows x 3 workgroups per patch (14.9 us
                         // per frame); 32 = 4 x 2 (fewer weight loads per MFMA, but one workgroup per CU: 16.4 us); 16 = 2 x 4: 16.6 us
#endif
#ifndef H2_DBG_SKIP
#define H2_DBG_SKIP 0     // profiling aid (results are then WRONG): 1 = leave out the k-loops, 2 = leave out the fused conv1 tiles
#endif
#ifndef H2_DBG_TIME
#define H2_DBG_TIME 0     // profiling aid: per-workgroup phase times of the fused conv1 + conv2 kernel (100 MHz wall clock),
#endif                    // summed into g_h2_prof and printed by the host after the launch
#if H2_DBG_TIME
__device__ unsigned long long g_h2_prof[8];
__device__ unsigned long long g_h34_prof[8];
// residency log of the fused conv1 + conv2 kernel: per wave (CU key, start, end) in 10 ns ticks -> the host prints how many
// workgroups a CU really holds over time and how long a slot stays empty between two of them
constexpr int H2_LOG_CAP = 1 << 18;
__device__ unsigned long long g_h2_log[3 * H2_LOG_CAP];
#define H34_STAMP(K) do { if (NXT_W > 0 && threadIdx.x == 0) { const unsigned long long now__ = wall_clock64(); atomicAdd(&g_h34_prof[K], now__ - t_prev__); t_prev__ = now__; } } while (0)
#define H2_STAMP(K) do { if (FUSE1 && threadIdx.x == 0) { const unsigned long long now__ = wall_clock64(); atomicAdd(&g_h2_prof[K], now__ - t_prev__); t_prev__ = now__; } } while (0)
#else
#define H2_STAMP(K) do { } while (0)
#define H34_STAMP(K) do { } while (0)
#endif
#ifndef H2_REBALANCE
#define H2_REBALANCE 1    // conv2: the tiles of a patch's short last block dealt out evenly over its waves
#endif
#ifndef H2_PRIO
#define H2_PRIO 1         // wave priority 3 outside the k-loop (staging, fused conv1, epilogue), 0 inside: see conv_h2_body
#endif
#ifndef H2C34_RN
#define H2C34_RN 3        // channel tiles per wave in the fused conv3 + conv4 kernel: 3 -> 4 waves per workgroup, 2 -> 6 waves
#endif
#ifndef H2C34_MINW
#define H2C34_MINW 2      // minimum waves per SIMD asked of the compiler for that kernel
#endif
#ifndef H2_C1_U
#define H2_C1_U 1         // tiles of the fused conv1 a wave keeps in flight at once (2 and 3 measured the same: the phase is short of issue slots, not of independent work)
#endif
#ifndef H2C2S_WM
#define H2C2S_WM 8       // waves per conv2 workgroup along the pixel tiles: R = H2C2S_TB / H2C2S_WM tiles per wave
#endif
#ifndef CK_H2_OPMAJOR
#define CK_H2_OPMAJOR 0     // developer knob: product-major order of the split-precision MFMAs inside a k-step (same sums bit for bit;
                            // measured: 14.8 vs 14.9 us per frame when held to 128 VGPRs, 16.6 at 130 -- the chain order is not the limit)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__host__ __device__ constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

// float -> bf16, round to nearest even (activations are finite)
__host__ __device__ __forceinline__ uint16_t f2bf(float f)
{
    union { float f; uint32_t u; } x;
    x.f = f;
    return (uint16_t)((x.u + 0x7FFFu + ((x.u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t h)
{
    union { float f; uint32_t u; } x;
    x.u = (uint32_t)h << 16;
    return x.f;
}

// region index -> first pixel row/col of its 40x40 patch (nn_manager.py:92-126, 256-275)
__device__ __forceinline__ int region_origin(int i) { return i == 9 ? 340 : 40 * i; }

// ------------------------------------------------------------------------------------------
// conv2..conv4: implicit-GEMM valid convolution + bias + relu (+ fused 2x2 max-pool) on
// 16-pixel x 16-channel tiles.  blockIdx.x = patch, blockIdx.y = group of TB pixel tiles;
// wave (wm, wn) owns R = TB / WAVES_M pixel tiles x RN channel tiles.
//   in : [patch][H][W][CIN] f32
//   wc : [channel tile][group of 4 k-steps][lane][4] f32 (pack_mfma16: the B fragment lane (k-slot, column)
//        needs for k-step 4g+e is element e; K = (kh, kw, cin padded to 4), zero padded)
//   out: POOL ? [patch][OH/2*OW/2][COUT] : [patch][OH*OW][COUT]
// With POOL a tile is a 4x4 block of output pixels = four 2x2 pooling windows (tile row 4q+e =
// window q, corner e): the four accumulator registers of a lane are exactly one window, so
// max-pooling is in-lane.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// smallest stride >= n with stride = rem (mod 32)
__host__ __device__ constexpr int lds_stride(int n, int rem) { return n + ((rem - n % 32) + 32) % 32; }

template <int H, int W, int CIN, int KH, int KW, int COUT, int TB, int YB, int WAVES_M, int RN, bool POOL>
__global__ __launch_bounds__(64 * WAVES_M * (cdiv(COUT, 16) / RN)) void conv_mfma16_f32_kernel(
    const float* __restrict__ in, const float* __restrict__ wc, const float* __restrict__ bias,
    float* __restrict__ out)
{
#pragma clang fp contract(off)
    constexpr int OH = H - KH + 1, OW = W - KW + 1, M = OH * OW;
    constexpr int NT = cdiv(COUT, 16), WAVES_N = NT / RN;
    constexpr int CINP = cdiv(CIN, 4) * 4;
    // LDS layout [row][col][channel] with pixel stride CS = 2 (mod 32) dwords and row stride RS chosen so
    // that the 16 pixels x 2 k-slots a half-wave reads in one ds_read_b32 fall on 32 distinct banks:
    //   plain tiles (16 consecutive output pixels, wrapping rows):  bank = 2*m + k      -> RS = 2*OW (mod 32)
    //   pooling tiles (4 x 4 output pixels):                        bank = 2*col + 8*row + k -> RS = 8 (mod 32)
    constexpr int CS = lds_stride(CINP, 2);
    constexpr int RS = lds_stride(W * CS, POOL ? 8 : (2 * OW) % 32);
    constexpr int KS = KH * KW * (CINP / 4), SG = cdiv(KS, 4);
    constexpr int NTHREADS = 64 * WAVES_M * WAVES_N;
    constexpr int R = cdiv(TB, WAVES_M);               // pixel tiles per wave (the last wave row may own R-1)
    constexpr int RT = POOL ? (OH / 4) * (OW / 4) : cdiv(M, 16);       // tiles per patch
    static_assert(NT % RN == 0, "channel tiles split evenly over the waves");
    static_assert(TB * YB >= RT, "the grid owns every pixel tile of the patch");
    static_assert(R * WAVES_M - TB <= 1, "uneven split handled for one missing tile only");
    static_assert(!POOL || (OW % 4 == 0 && OH % 4 == 0 && TB % (OW / 4) == 0), "pooling tiles are 4x4 output pixels, whole tile rows per block");
    // input rows one workgroup can touch
    constexpr int ROWS_RAW = POOL ? 4 * (TB / (OW / 4)) + KH - 1 : (TB * 16 + OW - 2) / OW + 1 + KH - 1;
    constexpr int ROWS = ROWS_RAW < H ? ROWS_RAW : H;
    __shared__ float lds[ROWS * RS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WAVES_N, wm = wave / WAVES_N;
    const int l15 = lane & 15, kq = lane >> 4;
    const int patch = blockIdx.x;
    const int tile_blk = blockIdx.y * TB;
    const int oy_min = POOL ? 4 * (tile_blk / (OW / 4)) : (tile_blk * 16) / OW;
    int row_cnt = H - oy_min;
    if (row_cnt > ROWS) row_cnt = ROWS;

    {
        const float* g = in + ((size_t)patch * H + oy_min) * W * CIN;
        if constexpr (CIN % 4 == 0) {
            const float4* g4 = reinterpret_cast<const float4*>(g);
#pragma unroll 4
            for (int i = tid; i < row_cnt * W * (CIN / 4); i += NTHREADS) {
                const int pxl = i / (CIN / 4), c = i % (CIN / 4);
                const float4 v = g4[i];
                float* d = &lds[(pxl / W) * RS + (pxl % W) * CS + 4 * c];     // 8-byte aligned (CS, RS even)
                *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
                *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
            }
        } else if constexpr (CIN % 2 == 0) {
            const float2* g2 = reinterpret_cast<const float2*>(g);
#pragma unroll 4
            for (int i = tid; i < row_cnt * W * (CIN / 2); i += NTHREADS) {
                const int pxl = i / (CIN / 2), c = i % (CIN / 2);
                const float2 v = g2[i];
                *reinterpret_cast<float2*>(&lds[(pxl / W) * RS + (pxl % W) * CS + 2 * c]) = v;
            }
        } else {
            for (int i = tid; i < row_cnt * W * CIN; i += NTHREADS) {
                const int pxl = i / CIN;
                lds[(pxl / W) * RS + (pxl % W) * CS + i % CIN] = g[i];
            }
        }
        if constexpr (CINP > CIN)
            for (int i = tid; i < row_cnt * W * (CINP - CIN); i += NTHREADS) {
                const int pxl = i / (CINP - CIN);
                lds[(pxl / W) * RS + (pxl % W) * CS + CIN + i % (CINP - CIN)] = 0.f;
            }
    }
    __syncthreads();

    const int tile0 = tile_blk + wm * R;
    int abase[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        int t = tile0 + r;
        if (t > RT - 1) t = RT - 1;
        int oy, ox;
        if constexpr (POOL) {
            // tile t = 4x4 output pixels = 2x2 pooling windows; tile row 4q+e = window q, corner e
            const int ty = t / (OW / 4), tx = t % (OW / 4), q = l15 >> 2, sub = l15 & 3;
            oy = 4 * ty + 2 * (q >> 1) + (sub >> 1);
            ox = 4 * tx + 2 * (q & 1) + (sub & 1);
        } else {
            int m = t * 16 + l15;
            if (m > M - 1) m = M - 1;
            oy = m / OW; ox = m % OW;
        }
        abase[r] = (oy - oy_min) * RS + ox * CS + kq;
    }
    f32x4 acc[R][RN];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int n = 0; n < RN; n++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[r][n][e] = 0.f;

    // real tiles of this wave (wave-uniform): all R, or R-1 for the last wave row of an uneven split
    int nv = TB - wm * R;
    nv = nv > R ? R : nv;
    const float4* wq = reinterpret_cast<const float4*>(wc) + (size_t)(wn * RN) * SG * 64 + lane;
    auto k_loop = [&](auto nv_tag) {
        constexpr int NV = decltype(nv_tag)::value;
        // one dwordx4 load per lane brings the B fragments of four consecutive k-steps (fully coalesced,
        // 1 KB per wave); the loads run PFG groups ahead of the MFMAs through a register ring with
        // compile-time indices
        constexpr int PFG = 3;
        float4 bq[PFG][RN];
#pragma unroll
        for (int u = 0; u < PFG; u++)
#pragma unroll
            for (int n = 0; n < RN; n++) bq[u][n] = u < SG ? wq[((size_t)n * SG + u) * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < KH; i++) {
#pragma unroll
            for (int j = 0; j < KW; j++) {
#pragma unroll
                for (int cc = 0; cc < CINP / 4; cc++) {
                    const int step = (i * KW + j) * (CINP / 4) + cc;
                    const int g = step / 4, e = step % 4;
                    float b[RN];
#pragma unroll
                    for (int n = 0; n < RN; n++) {
                        const float4 q = bq[g % PFG][n];
                        b[n] = e == 0 ? q.x : e == 1 ? q.y : e == 2 ? q.z : q.w;
                        if ((e == 3 || step == KS - 1) && g + PFG < SG) bq[g % PFG][n] = wq[((size_t)n * SG + g + PFG) * 64];
                    }
#pragma unroll
                    for (int r = 0; r < NV; r++) {
                        const float a = lds[abase[r] + i * RS + j * CS + 4 * cc];
#pragma unroll
                        for (int n = 0; n < RN; n++)
                            acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[n], acc[r][n], 0, 0, 0);
                    }
                }
            }
        }
    };
    if (nv == R) k_loop(std::integral_constant<int, R>{});
    else if constexpr (R * WAVES_M > TB) k_loop(std::integral_constant<int, R - 1>{});

#pragma unroll
    for (int n = 0; n < RN; n++) {
        const int co = (wn * RN + n) * 16 + l15;
        const float bv = co < COUT ? bias[co] : 0.f;
        if constexpr (POOL) {
            float* o = out + (size_t)patch * (M / 4) * COUT;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int t = tile0 + r;
                float mx = acc[r][n][0] > acc[r][n][1] ? acc[r][n][0] : acc[r][n][1];
                const float m2 = acc[r][n][2] > acc[r][n][3] ? acc[r][n][2] : acc[r][n][3];
                mx = mx > m2 ? mx : m2;
                mx = mx + bv;                     // max commutes with the (monotone) bias add and relu
                mx = mx > 0.f ? mx : 0.f;
                const int py = 2 * (t / (OW / 4)) + (kq >> 1), px = 2 * (t % (OW / 4)) + (kq & 1);
                if (r < nv && t < RT && co < COUT) o[(size_t)(py * (OW / 2) + px) * COUT + co] = mx;
            }
        } else {
            float* o = out + (size_t)patch * M * COUT;
#pragma unroll
            for (int r = 0; r < R; r++) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int m = (tile0 + r) * 16 + 4 * kq + e;
                    float v = acc[r][n][e] + bv;
                    v = v > 0.f ? v : 0.f;
                    if (r < nv && m < M && co < COUT) o[(size_t)m * COUT + co] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// conv1: 5x5x3 -> 32 on the raw 40x40 u8 patch (K10 gather fused: the patch is read straight
// from the goban image), + bias + relu.  K = 75 is walked densely in (kh, kw, cin) order --
// a kernel row is 15 consecutive floats of the staged [row][col*3 + c] image -- in 19 k-steps of
// v_mfma_f32_16x16x4_f32 (k = 75 is a zero weight).  The operands are swapped relative to
// conv_mfma16_f32_kernel (A = weights, B = pixels), so a lane's four accumulators are four
// consecutive channels of one pixel and the result leaves as 16-byte stores.
//   work unit = 12 output rows of one patch = 27 pixel tiles, WAVES waves x R tiles
//   wf : [2 channel tiles][19 steps][64 lanes] f32 (pack_conv1)
//   out: [patch][36*36][32] f32
template <int R>
__global__ __launch_bounds__(64 * (27 / R)) void conv1_mfma16_kernel(
    const uint8_t* __restrict__ goban, const float* __restrict__ wf, const float* __restrict__ bias,
    float* __restrict__ out, int nunits)
{
#pragma clang fp contract(off)
    constexpr int WAVES = 27 / R, NTHREADS = 64 * WAVES;
    constexpr int OW = 36, ROWS = 16, RS = 140;           // RS = 12 (mod 32): a tile wrapping to the next row keeps its bank walk
    static_assert(27 % R == 0, "27 tiles per work unit");
    constexpr int NPT = cdiv(ROWS * 30, NTHREADS);       // dwords of the next unit each thread keeps in flight
    __shared__ float lds[ROWS * RS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;

    // work unit = (patch, third of its output rows); persistent workgroups walk the units with
    // the 16 x 120 input bytes of the NEXT unit already in flight (patch rows are 4-byte aligned in
    // the goban image) while the current one is on the matrix cores
    uint32_t raw[NPT];
    auto fetch = [&](int unit) {
#pragma unroll
        for (int q = 0; q < NPT; q++) {
            const int d = tid + q * NTHREADS;
            raw[q] = 0u;
            if (unit < nunits && d < ROWS * 30) {
                const int patch = unit / 3, by = unit % 3;
                const int frame = patch / 100, reg = patch % 100;
                const int py0 = region_origin(reg / 10) + 12 * by, px0 = region_origin(reg % 10);
                const uint8_t* src = goban + ((size_t)frame * 380 + py0 + d / 30) * 380 * 3 + (size_t)px0 * 3;
                raw[q] = reinterpret_cast<const uint32_t*>(src)[d % 30];
            }
        }
    };
    int unit = blockIdx.x;
    fetch(unit);

    // this lane's weight fragments for all 19 steps and both channel tiles
    float wq[19][2];
#pragma unroll
    for (int s = 0; s < 19; s++)
#pragma unroll
        for (int n = 0; n < 2; n++) wq[s][n] = wf[(n * 19 + s) * 64 + lane];
    float4 bv[2];
#pragma unroll
    for (int n = 0; n < 2; n++) bv[n] = *reinterpret_cast<const float4*>(bias + n * 16 + 4 * kq);
    int abase[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int m = (wave * R + r) * 16 + l15;          // pixel inside the unit's 12 rows
        abase[r] = (m / OW) * RS + (m % OW) * 3;
    }

    for (; unit < nunits; unit += gridDim.x) {
#pragma unroll
        for (int q = 0; q < NPT; q++) {
            const int d = tid + q * NTHREADS;
            if (d < ROWS * 30) {
                float* o = &lds[(d / 30) * RS + 4 * (d % 30)];
                o[0] = (float)(raw[q] & 0xFFu); o[1] = (float)((raw[q] >> 8) & 0xFFu);
                o[2] = (float)((raw[q] >> 16) & 0xFFu); o[3] = (float)(raw[q] >> 24);
            }
        }
        __syncthreads();
        fetch(unit + gridDim.x);

        f32x4 acc[R][2];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc[r][n][e] = 0.f;
#pragma unroll
        for (int s = 0; s < 19; s++) {
            int k = 4 * s + kq;
            k = k > 74 ? 74 : k;                          // the padding slot re-reads the last tap (its weight is zero)
            const int i0 = (4 * s) / 15;
            const int i = k >= 15 * (i0 + 1) ? i0 + 1 : i0;
            const int koff = i * RS + (k - 15 * i);
#pragma unroll
            for (int r = 0; r < R; r++) {
                const float a = lds[abase[r] + koff];
#pragma unroll
                for (int n = 0; n < 2; n++)
                    acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[s][n], a, acc[r][n], 0, 0, 0);
            }
        }
        __syncthreads();                                  // every wave is done reading before the next unit lands
        const int patch = unit / 3, by = unit % 3;
#pragma unroll
        for (int n = 0; n < 2; n++) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const size_t m = (size_t)patch * 1296 + (size_t)by * 432 + (wave * R + r) * 16 + l15;
                float4 v;
                v.x = acc[r][n][0] + bv[n].x; v.y = acc[r][n][1] + bv[n].y; v.z = acc[r][n][2] + bv[n].z; v.w = acc[r][n][3] + bv[n].w;
                v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                *reinterpret_cast<float4*>(out + m * 32 + n * 16 + 4 * kq) = v;
            }
        }
    }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__host__ __device__ constexpr int lds_stride_b(int n, int rem, int mod) { return n + ((rem - n % mod) + mod) % mod; }

// ------------------------------------------------------------------------------------------
// Split-precision mode (CK_CNN_F16X2): f32-accurate convolutions on the fp16 matrix pipe.
// Every f32 operand x is carried as two halves, hi = fp16(x) and lo = fp16(x - hi) (22 mantissa bits
// together); a product is the three MFMAs lo*hi + hi*lo + hi*hi on v_mfma_f32_16x16x32_f16 with f32
// accumulation (the lo*lo term is below 2^-22 of the product).  Three fp16 MFMAs of 16 cycles replace
// eight f32 MFMAs of 32 cycles per 16x16x32 block: the ceiling moves from 155 to ~830 TFLOP/s-equivalent.
// Weights are pre-scaled by 2^8 so that their lo halves stay normal fp16 numbers (undone on the accumulator,
// exact).  Results agree with the f32 kernels to ~1e-6 (tests), not bit for bit: the k-ordered f32 chain
// is what CK_CNN_FP32 keeps.
//   in : [patch][H][W][CIN] f32       wt : [16-channel tile][k-step][hi|lo][lane][8] fp16 (pack_mfma16_h2)
//   out: as conv_mfma16_f32_kernel
// LDS: per pixel [hi CINP][lo CINP][8 pad] halves (pixel stride = 16 mod 128 bytes), row stride chosen as in
// the bf16 kernel; one k-step = 32 channels of one kernel tap.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// swizzled conv2 tile: chunk c of the pixel at column x of tile row y sits in slot c ^ h2_swz(x, y)
__device__ __forceinline__ int h2_swz(int x, int y) { return ((x & 6) ^ (x & 1) ^ (y << 2)) & 7; }

__device__ __forceinline__ void split_h2(float x, _Float16& hi, _Float16& lo)
{
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// Two values at once, in four instructions instead of ten (round 4): hi = the fp16 TOWARDS ZERO of x (one
// v_cvt_pkrtz_f16_f32 for both), residual x - hi by v_fma_mix_f32 (the fp16 operand is widened inside the instruction),
// lo = fp16 of the residuals (one v_cvt_pkrtz again).  Rounding hi down instead of to nearest leaves a residual of up to one
// fp16 ulp instead of half of one -- lo still holds it to 2^-10 of that, 2^-20 of x: the sum hi + lo is as good as before.
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_h2x2(float a, float b, h2v& hi, h2v& lo)
{
    const uint32_t hu = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));
    float ra, rb;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hu), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hu), "v"(b));
    hi = __builtin_bit_cast(h2v, hu);
    lo = __builtin_bit_cast(h2v, __builtin_amdgcn_cvt_pkrtz(ra, rb));
}

// Padding of the non-swizzled split-precision tiles, chosen against the REAL ds_read_b128 lane groups
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32: MI355X_MICROARCH.md, LDS) with a model of every fragment read of
// the layer (tools/lds_conflict_model.py): conv3 (plain tiles of 16 raster pixels, 32 channels) is conflict-free with
// 48 halves of pixel padding and rows = 32 halves mod 64, conv4 (pooling tiles, 96 channels) with no pixel padding and
// rows = 16 mod 64; the round-2 values (8 / 48 and 8 / 32) cost 7.7 and 8.0 LDS cycles per read instead of 4.
template <int CINP, bool POOL>
constexpr int h2_pspad() { return POOL ? (CINP == 96 ? 0 : 8) : (CINP == 32 ? 48 : 8); }
template <int CINP, bool POOL, int OW>
constexpr int h2_rsrem() { return POOL ? (CINP == 96 ? 16 : 32) : (CINP == 32 ? 32 : (8 * OW) % 64); }

constexpr int H2_IN1_HALVES = 21 * 128;     // the fused conv1's staged pixel rows (conv2: 16 + 4 rows + one zero row, 128 halves each)
// halves of LDS one block's input tile takes (same formulas as in the body)
template <int H, int W, int CIN, int KH, int KW, int TB, bool POOL, bool SWZ>
constexpr int h2_tile_halves()
{
    constexpr int OW = W - KW + 1, M = (H - KH + 1) * OW, CINP = cdiv(CIN, 32) * 32;
    constexpr int PS = SWZ ? 2 * CINP : 2 * CINP + h2_pspad<CINP, POOL>();
    constexpr int RS = SWZ ? W * PS : lds_stride_b(W * PS, h2_rsrem<CINP, POOL, OW>(), 64);
    constexpr int ROWS_RAW = POOL ? 4 * (TB / (OW / 4)) + KH - 1 : (TB * 16 + OW - 2) / OW + 1 + KH - 1;
    (void)M;
    return (ROWS_RAW < H ? ROWS_RAW : H) * RS;
}

// The layer itself, as a device function over a caller-owned LDS tile so that two layers can share one workgroup:
//   IN_LDS  : the input tile is already in `lds` (written by the previous layer's call), nothing is staged;
//   NXT_W>0 : the output (relu, not pooled) is not written to `out` but split into hi/lo halves straight into `lds`
//             in the NEXT layer's tile layout (NXT_W pixels per row, NXT_PS halves per pixel, NXT_RS per row,
//             NXT_CINP channels per plane, channels COUT..NXT_CINP-1 zeroed) -- after a barrier, because that tile
//             overlays this layer's input.
template <int H, int W, int CIN, int KH, int KW, int COUT, int TB, int YB, int WAVES_M, int RN, bool POOL, int PF, bool SB, bool SWZ = false,
          bool FUSE1 = false, bool IN_LDS = false, int NXT_W = 0, int NXT_PS = 0, int NXT_RS = 0, int NXT_CINP = 0>
__device__ __forceinline__ void conv_h2_body(
    _Float16* __restrict__ lds, const int patch, const int blk_y,
    const float* __restrict__ in, const uint16_t* __restrict__ wt, const float* __restrict__ bias,
    float* __restrict__ out, float wscale_inv, int* __restrict__ overflow,
    const uint8_t* __restrict__ goban1 = nullptr, const uint16_t* __restrict__ wf1 = nullptr, const float* __restrict__ bias1 = nullptr)
{
#pragma clang fp contract(off)
    constexpr int OH = H - KH + 1, OW = W - KW + 1, M = OH * OW;
    constexpr int NT = cdiv(COUT, 16), WAVES_N = NT / RN;
    constexpr int CINP = cdiv(CIN, 32) * 32;
    // SWZ: no padding -- a pixel is exactly its 8 chunks of 16 bytes (hi 0..3, lo 4..7), rows are whole multiples of
    // 256 bytes, and chunk c of the pixel at column x of tile row y sits in slot c ^ h2_swz(x, y) =
    // c ^ (2 (x >> 1) ^ (x & 1) ^ 4 (y & 1)).  ds_read_b128 is served in four groups of 16 lanes that are NOT contiguous
    // ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32: MI355X_MICROARCH.md, LDS): a group holds the two diagonal 2x2
    // windows of the pooling tile with k-quarter kq and the two off-diagonal ones with kq ^ 1.  Found by exhaustive
    // search over linear swizzles against a model of every fragment read of the layer (tools/lds_conflict_model.py):
    // every group lands on 16 distinct 16-byte slots of the 64 banks, for every tap -- and the 8-byte stores of the
    // fused conv1 (16 consecutive pixels of a row per lane group) spread over the 32 store banks two deep, as before.
    // (The round-2 swizzle, (x >> 1) & 7 with rows 32 bytes apart, is conflict-free for CONTIGUOUS groups of 16
    // lanes; with the real groups the model gives 9.6 LDS cycles per read instead of 4.)
    static_assert(!SWZ || (CINP == 32 && POOL), "swizzled layout: 32 channels, pooling tiles");
    constexpr int PS = SWZ ? 2 * CINP : 2 * CINP + h2_pspad<CINP, POOL>();              // halves per pixel
    constexpr int RS = SWZ ? W * PS : lds_stride_b(W * PS, h2_rsrem<CINP, POOL, OW>(), 64);     // halves per row
    constexpr int KS = KH * KW * (CINP / 32);
    constexpr int NTHREADS = 64 * WAVES_M * WAVES_N;
    constexpr int R = cdiv(TB, WAVES_M);
    constexpr int RT = POOL ? (OH / 4) * (OW / 4) : cdiv(M, 16);
    static_assert(NT % RN == 0 && TB * YB >= RT && R * WAVES_M - TB <= 1, "tile split");
    static_assert(!POOL || (OW % 4 == 0 && OH % 4 == 0 && TB % (OW / 4) == 0), "pooling tiles are 4x4 output pixels");
    static_assert(CIN % 2 == 0, "channel pairs");
    constexpr int ROWS_RAW = POOL ? 4 * (TB / (OW / 4)) + KH - 1 : (TB * 16 + OW - 2) / OW + 1 + KH - 1;
    constexpr int ROWS = ROWS_RAW < H ? ROWS_RAW : H;
    static_assert(ROWS * RS == h2_tile_halves<H, W, CIN, KH, KW, TB, POOL, SWZ>(), "tile size helper out of step");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WAVES_N, wm = wave / WAVES_N;
    const int l15 = lane & 15, kq = lane >> 4;
#if H2_DBG_TIME
    unsigned long long t_prev__ = wall_clock64();
    const unsigned long long t_born__ = t_prev__;
#endif
    // Two workgroups share a CU, and while one is in its k-loop (MFMA after MFMA) the other is usually staging its
    // tile or running the fused conv1: short phases of loads, LDS traffic and vector arithmetic that need few issue
    // slots but, at equal priority, wait behind the other workgroup's matrix instructions for every one of them
    // (measured per workgroup: 7.8 us of staging + conv1 next to 8.4 us of k-loop, for 6 % of the flops).  They run
    // at wave priority 3, the k-loop at 0: the issue arbiter serves the short phase first, the matrix pipe stays fed
    // by the other workgroup.
    if constexpr (H2_PRIO != 0) __builtin_amdgcn_s_setprio(3);
    const int tile_blk = blk_y * TB;
    const int oy_min = POOL ? 4 * (tile_blk / (OW / 4)) : (tile_blk * 16) / OW;
    int row_cnt = H - oy_min;
    if (row_cnt > ROWS) row_cnt = ROWS;
    if constexpr (FUSE1) {
        // conv1 (5x5x3 -> 32, relu) of the 40x40 u8 patch computed HERE for the rows this block needs: its output goes
        // straight into the swizzled tile as hi/lo halves and never exists in HBM.  Same arithmetic as conv1_h2_kernel
        // (K = 6 kernel rows x 16 slots in three k-steps, weights as the A operand, two MFMAs per product), so the
        // values are bit-identical to the unfused pair of kernels; a block recomputes the 4 halo rows it shares with
        // its neighbour (+33 % of a layer that is 6 % of the network).
        static_assert(SWZ && W == 36 && CIN == 32 && KH == 5, "conv1 fusion is wired for conv2");
        constexpr int IRS = 128;                              // halves per staged input row: 120 used, 8 zero
        constexpr int IROWS = ROWS + 4;
        static_assert((IROWS + 1) * IRS == H2_IN1_HALVES, "staged pixel rows: helper out of step");
        __shared__ __attribute__((aligned(16))) _Float16 in1[(IROWS + 1) * IRS];       // + one zero row under kernel row 5
        // the pixels of block by_ of patch p_ as halves into `dst`
        auto stage_pixels = [&](int p_, int by_, _Float16* dst) {
            const int oy_s = 4 * ((by_ * TB) / (OW / 4));
            int rows_s = H - oy_s;
            if (rows_s > ROWS) rows_s = ROWS;
            const int frame = p_ / 100, reg = p_ % 100;
            const int py0 = region_origin(reg / 10) + oy_s, px0 = region_origin(reg % 10);
            const int rows_in = rows_s + 4;
            const uint8_t* src = goban1 + ((size_t)frame * 380 + py0) * 380 * 3 + (size_t)px0 * 3;
            for (int d = tid; d < IROWS * 32 + IRS / 4; d += NTHREADS) {
                const int r = d >> 5, cd = d & 31;            // 30 dwords of pixels + 2 of padding per row; then the zero row
                uint32_t raw = 0u;
                if (r < rows_in && cd < 30) raw = reinterpret_cast<const uint32_t*>(src + (size_t)r * 380 * 3)[cd];
                typedef _Float16 h4v __attribute__((ext_vector_type(4)));
                *reinterpret_cast<h4v*>(&dst[4 * d]) = h4v{(_Float16)(float)(raw & 0xFFu), (_Float16)(float)((raw >> 8) & 0xFFu),
                                                           (_Float16)(float)((raw >> 16) & 0xFFu), (_Float16)(float)(raw >> 24)};
            }
        };
        stage_pixels(patch, blk_y, in1);
        h8 w1[3][2][2];
#pragma unroll
        for (int sx = 0; sx < 3; sx++)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int pl = 0; pl < 2; pl++)
                    w1[sx][n][pl] = __builtin_bit_cast(h8, reinterpret_cast<const uint4*>(wf1)[((n * 3 + sx) * 2 + pl) * 64 + lane]);
        float4 bv1[2];
#pragma unroll
        for (int n = 0; n < 2; n++) bv1[n] = *reinterpret_cast<const float4*>(bias1 + n * 16 + 4 * kq);
        __syncthreads();
        H2_STAMP(0);                                          // pixels staged, conv1 weights loaded
        const int ntile1 = row_cnt * W / 16;                 // 36 or 27 tiles of 16 conv1 pixels
        constexpr int NW = NTHREADS / 64, R1 = cdiv(ROWS * W / 16, NW);
        float big = 0.f;
        // U tiles of a wave in flight at once, their operations interleaved in program order (the compiler keeps it):
        // one tile is a chain of dependent steps -- five LDS dwords per k-step, funnel shifts, 12 MFMAs in two chains,
        // bias / relu / split, stores -- that takes ~1 us next to the other workgroup's k-loop, and a wave has 4 or 5.
        auto conv1_tiles = [&](auto u_tag, int t_first) {
            constexpr int U = decltype(u_tag)::value;
            int a0[U], my[U], mx[U];
            uint32_t ash[U];
            f32x4 c1[U][2];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int m = (t_first + NW * u) * 16 + l15;
                my[u] = m / W; mx[u] = m % W;
                a0[u] = (my[u] + (kq >> 1)) * IRS + mx[u] * 3 + 8 * (kq & 1);
                ash[u] = (uint32_t)(a0[u] & 1) << 4;
#pragma unroll
                for (int n = 0; n < 2; n++) {                    // the accumulators start at bias x weight scale (a power of two: exact)
                    c1[u][n][0] = bv1[n].x * H2_WSCALE; c1[u][n][1] = bv1[n].y * H2_WSCALE;
                    c1[u][n][2] = bv1[n].z * H2_WSCALE; c1[u][n][3] = bv1[n].w * H2_WSCALE;
                }
            }
#pragma unroll
            for (int sx = 0; sx < 3; sx++) {
                h8 a[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    // 8 halves from an odd or even half offset: five aligned dwords, funnel-shifted by 0 or 16 bits
                    const uint32_t* aq = reinterpret_cast<const uint32_t*>(&in1[(a0[u] & ~1) + 2 * sx * IRS]);
                    const uint32_t d0 = aq[0], d1 = aq[1], d2 = aq[2], d3 = aq[3], d4 = aq[4];
                    const uint4 au = make_uint4(__builtin_amdgcn_alignbit(d1, d0, ash[u]), __builtin_amdgcn_alignbit(d2, d1, ash[u]),
                                                __builtin_amdgcn_alignbit(d3, d2, ash[u]), __builtin_amdgcn_alignbit(d4, d3, ash[u]));
                    a[u] = __builtin_bit_cast(h8, au);
                }
#pragma unroll
                for (int n = 0; n < 2; n++) {
#pragma unroll
                    for (int u = 0; u < U; u++) c1[u][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[sx][n][1], a[u], c1[u][n], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < U; u++) c1[u][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[sx][n][0], a[u], c1[u][n], 0, 0, 0);
                }
            }
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int u = 0; u < U; u++) {
                    // lane (pixel l15, kq) holds channels 16 n + 4 kq .. + 3: half a chunk of the pixel
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        v[e] = c1[u][n][e] * wscale_inv;
                        v[e] = v[e] > 0.f ? v[e] : 0.f;
                    }
                    big = fmaxf(big, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
                    typedef _Float16 h4v __attribute__((ext_vector_type(4)));
                    h2v h01, l01, h23, l23;
                    split_h2x2(v[0], v[1], h01, l01);
                    split_h2x2(v[2], v[3], h23, l23);
                    const h4v hi = {h01[0], h01[1], h23[0], h23[1]}, lo = {l01[0], l01[1], l23[0], l23[1]};
                    const int sw = h2_swz(mx[u], my[u]), c = 2 * n + (kq >> 1);
                    _Float16* px = &lds[my[u] * RS + mx[u] * PS + 4 * (kq & 1)];
                    *reinterpret_cast<h4v*>(px + ((c ^ sw) << 3)) = hi;
                    *reinterpret_cast<h4v*>(px + ((c ^ sw ^ 4) << 3)) = lo;
                }
        };
        {
            constexpr int U = H2_C1_U;
            int r = 0;
            if (H2_DBG_SKIP != 2) {
                for (; r + U <= R1 && wave + NW * (r + U - 1) < ntile1; r += U) conv1_tiles(std::integral_constant<int, U>{}, wave + NW * r);
                for (; r < R1 && wave + NW * r < ntile1; r++) conv1_tiles(std::integral_constant<int, 1>{}, wave + NW * r);
            }
        }
        if (overflow && !(big <= 65000.f)) *overflow = 1;
        H2_STAMP(4);                                          // wave 0's conv1 tiles
    } else if constexpr (!IN_LDS) {
        const float2* g = reinterpret_cast<const float2*>(in + ((size_t)patch * H + oy_min) * W * CIN);
        float big = 0.f;
        // H2_STAGE_UNROLL loads in flight per thread, then their splits (written out: the inline asm of the split keeps
        // the compiler from unrolling a loop with a run-time trip count itself)
        const int n_pairs = row_cnt * W * (CIN / 2);
        for (int i0 = tid; i0 < n_pairs; i0 += H2_STAGE_UNROLL * NTHREADS) {
            float2 vv[H2_STAGE_UNROLL];
#pragma unroll
            for (int u = 0; u < H2_STAGE_UNROLL; u++) {
                const int i = i0 + u * NTHREADS;
                vv[u] = i < n_pairs ? g[i] : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < H2_STAGE_UNROLL; u++) {
                const int i = i0 + u * NTHREADS;
                if (i >= n_pairs) break;
                const int pxl = i / (CIN / 2), c = i % (CIN / 2);
                const float2 v = vv[u];
                // an activation outside the fp16 range would become inf (and the relu would then hide the NaN): the
                // largest magnitude is tracked and checked once, the host recomputes a flagged batch with the f32 kernels
                big = fmaxf(big, fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)));
                h2v hh, ll;
                split_h2x2(v.x, v.y, hh, ll);
                if constexpr (SWZ) {
                    const int x = pxl % W, sw = h2_swz(x, pxl / W);
                    _Float16* d = &lds[(pxl / W) * RS + x * PS + (2 * c & 7)];
                    *reinterpret_cast<h2v*>(d + (((c >> 2) ^ sw) << 3)) = hh;
                    *reinterpret_cast<h2v*>(d + (((c >> 2) ^ sw ^ 4) << 3)) = ll;
                } else {
                    _Float16* d = &lds[(pxl / W) * RS + (pxl % W) * PS + 2 * c];
                    *reinterpret_cast<h2v*>(d) = hh;
                    *reinterpret_cast<h2v*>(d + CINP) = ll;
                }
            }
        }
        if (overflow && !(big <= 65000.f)) *overflow = 1;       // also true for NaN
        if constexpr (CINP > CIN)
            for (int i = tid; i < row_cnt * W * (CINP - CIN); i += NTHREADS) {
                const int pxl = i / (CINP - CIN), c = CIN + i % (CINP - CIN);
                _Float16* d = &lds[(pxl / W) * RS + (pxl % W) * PS + c];
                d[0] = (_Float16)0.f;
                d[CINP] = (_Float16)0.f;
            }
    }
    if constexpr (!IN_LDS) __syncthreads();
    H2_STAMP(1);                                              // conv1 tiles done (all waves)
    H34_STAMP(4);                                             // conv3: input staged

    // The last block of a patch may hold fewer tiles than TB (conv2: 64 pooling tiles = 24 + 24 + 16).  With the fixed
    // R tiles per wave five of its eight waves would do three tiles and the rest next to nothing, and the block would take as
    // long as a full one for two thirds of the work: its tiles are dealt out evenly instead (H2_REBALANCE).
    int r_blk = R;
    if constexpr (H2_REBALANCE && FUSE1) {
        const int left = RT - tile_blk;
        if (left < TB) r_blk = (left + WAVES_M - 1) / WAVES_M;
    }
    const int tile0 = tile_blk + wm * r_blk;
    int abase[R], axr[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        int t = tile0 + r;
        if (t > RT - 1) t = RT - 1;
        int oy, ox;
        if constexpr (POOL) {
            const int ty = t / (OW / 4), tx = t % (OW / 4), q = l15 >> 2, sub = l15 & 3;
            oy = 4 * ty + 2 * (q >> 1) + (sub >> 1);
            ox = 4 * tx + 2 * (q & 1) + (sub & 1);
        } else {
            int m = t * 16 + l15;
            if (m > M - 1) m = M - 1;
            oy = m / OW; ox = m % OW;
        }
        abase[r] = SWZ ? (oy - oy_min) * RS + ox * PS : (oy - oy_min) * RS + ox * PS + 8 * kq;
        axr[r] = SWZ ? (ox | ((oy - oy_min) & 1) << 18) : ox;      // swizzled layout: column, and 4 x the parity of the tile row
    }
    // The accumulators start at bias x weight scale (a power of two: exact), so the epilogue has no global load to wait
    // for -- the bias fetch hides behind the staging instead of sitting between the last MFMA and the stores.
    f32x4 acc[R][RN];
#pragma unroll
    for (int n = 0; n < RN; n++) {
        const int co0 = (wn * RN + n) * 16 + l15;
        const float b0 = (co0 < COUT ? bias[co0] : 0.f) * H2_WSCALE;
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[r][n][e] = b0;
    }

    int nv = TB - wm * R;
    nv = nv > R ? R : nv;
    if constexpr (H2_REBALANCE && FUSE1) {
        const int left = RT - tile_blk;
        if (left < TB) {
            nv = left - wm * r_blk;
            nv = nv > r_blk ? r_blk : (nv < 0 ? 0 : nv);
        }
    }
    // B fragments: [tile][step][plane][lane] uint4
    const uint4* wq = reinterpret_cast<const uint4*>(wt) + (size_t)(wn * RN) * KS * 128 + lane;
    auto k_loop = [&](auto nv_tag) {
        constexpr int NV = decltype(nv_tag)::value;
        // Weight fragments run PF k-steps ahead of their use through a ring of PF + 1 register slots.  A step's
        // MFMAs are short here (16 cycles each), so the distance has to be real: the loads for step s + PF are
        // issued at the top of step s and a scheduling barrier keeps the compiler from sinking them next to
        // their first use (which is what it does otherwise: vmcnt(0) right behind the load).  SB = false leaves the
        // schedule to the compiler: with 7 tiles x 3 channel tiles per wave (conv3) the barriers make it keep every
        // A fragment of a step live and the occupancy collapses.
        constexpr int NS = PF + 1;
        uint4 bq[NS][RN][2];
        if constexpr (SWZ) {
            // column tap outermost: the swizzle of a pixel depends on its column x = ox + j and on the parity of its row.
            // The two fragment addresses of a pixel tile are computed once per tap column; a row of the other parity
            // flips bit 2 of the slot, which is exactly what tells the hi plane from the lo plane -- so for odd taps i the
            // two addresses simply change roles, and the KH rows are immediate offsets.  The weight fragments stay in
            // (i, j) order in memory; seq = j * KH + i is the order they are used in.
            auto step_of = [](int seq) { return (seq % KH) * KW + seq / KH; };
#pragma unroll
            for (int u = 0; u < PF; u++)
#pragma unroll
                for (int n = 0; n < RN; n++)
#pragma unroll
                    for (int pl = 0; pl < 2; pl++) bq[u][n][pl] = wq[((size_t)n * KS + step_of(u)) * 128 + pl * 64];
#pragma unroll
            for (int j = 0; j < KW; j++) {
                // hi-plane address for even taps i; the lo plane -- and the hi plane for odd i -- is 32 halves away (slot ^ 4:
                // pixels and rows are multiples of 64 halves, so the slot bits of an address can be flipped with an xor)
                int a0j[NV > 0 ? NV : 1], a1j[NV > 0 ? NV : 1];
#pragma unroll
                for (int r = 0; r < NV; r++) {
                    const int x = (axr[r] & 0xFFFF) + j;
                    const int ch = kq ^ h2_swz(x, 0) ^ (axr[r] >> 16);
                    a0j[r] = abase[r] + j * PS + (ch << 3);
                    a1j[r] = a0j[r] ^ 32;
                }
#pragma unroll
                for (int i = 0; i < KH; i++) {
                    const int seq = j * KH + i;
                    if (seq + PF < KS) {
#pragma unroll
                        for (int n = 0; n < RN; n++) {
                            bq[(seq + PF) % NS][n][0] = wq[((size_t)n * KS + step_of(seq + PF)) * 128];
                            bq[(seq + PF) % NS][n][1] = wq[((size_t)n * KS + step_of(seq + PF)) * 128 + 64];
                        }
                    }
                    if constexpr (SB) __builtin_amdgcn_sched_barrier(0);
                    h8 bh[RN], bl[RN];
#pragma unroll
                    for (int n = 0; n < RN; n++) {
                        bh[n] = __builtin_bit_cast(h8, bq[seq % NS][n][0]);
                        bl[n] = __builtin_bit_cast(h8, bq[seq % NS][n][1]);
                    }
#if CK_H2_OPMAJOR
                    // product-major order: the three products of a step go round all NV x RN accumulators before any
                    // accumulator is touched again (each still sees al*bh, ah*bl, ah*bh in that order: same sums, bit for
                    // bit), and every A fragment of the step is requested before the first MFMA
                    h8 ahv[NV > 0 ? NV : 1], alv[NV > 0 ? NV : 1];
#pragma unroll
                    for (int r = 0; r < NV; r++) {
                        ahv[r] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(&lds[((i & 1) ? a1j[r] : a0j[r]) + i * RS]));
                        alv[r] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(&lds[((i & 1) ? a0j[r] : a1j[r]) + i * RS]));
                    }
#pragma unroll
                    for (int r = 0; r < NV; r++)
#pragma unroll
                        for (int n = 0; n < RN; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alv[r], bh[n], acc[r][n], 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < NV; r++)
#pragma unroll
                        for (int n = 0; n < RN; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahv[r], bl[n], acc[r][n], 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < NV; r++)
#pragma unroll
                        for (int n = 0; n < RN; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahv[r], bh[n], acc[r][n], 0, 0, 0);
#else
#pragma unroll
                    for (int r = 0; r < NV; r++) {
                        const h8 ah = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(&lds[((i & 1) ? a1j[r] : a0j[r]) + i * RS]));
                        const h8 al = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(&lds[((i & 1) ? a0j[r] : a1j[r]) + i * RS]));
#pragma unroll
                        for (int n = 0; n < RN; n++) {
                            acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[n], acc[r][n], 0, 0, 0);
                            acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[n], acc[r][n], 0, 0, 0);
                            acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[n], acc[r][n], 0, 0, 0);
                        }
                    }
#endif
                    if constexpr (SB) __builtin_amdgcn_sched_barrier(0);
                }
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < PF; u++)
#pragma unroll
            for (int n = 0; n < RN; n++)
#pragma unroll
                for (int pl = 0; pl < 2; pl++)
                    bq[u][n][pl] = u < KS ? wq[((size_t)n * KS + u) * 128 + pl * 64] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < KH; i++) {
#pragma unroll
            for (int j = 0; j < KW; j++) {
#pragma unroll
                for (int cc = 0; cc < CINP / 32; cc++) {
                    const int step = (i * KW + j) * (CINP / 32) + cc;
                    if (step + PF < KS) {
#pragma unroll
                        for (int n = 0; n < RN; n++) {
                            bq[(step + PF) % NS][n][0] = wq[((size_t)n * KS + step + PF) * 128];
                            bq[(step + PF) % NS][n][1] = wq[((size_t)n * KS + step + PF) * 128 + 64];
                        }
                    }
                    if constexpr (SB) __builtin_amdgcn_sched_barrier(0);
                    h8 bh[RN], bl[RN];
#pragma unroll
                    for (int n = 0; n < RN; n++) {
                        bh[n] = __builtin_bit_cast(h8, bq[step % NS][n][0]);
                        bl[n] = __builtin_bit_cast(h8, bq[step % NS][n][1]);
                    }
#pragma unroll
                    for (int r = 0; r < NV; r++) {
                        const _Float16* ap = &lds[abase[r] + i * RS + j * PS + 32 * cc];
                        const h8 ah = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(ap));
                        const h8 al = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(ap + CINP));
#pragma unroll
                        for (int n = 0; n < RN; n++) {
                            acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[n], acc[r][n], 0, 0, 0);
                            acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[n], acc[r][n], 0, 0, 0);
                            acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[n], acc[r][n], 0, 0, 0);
                        }
                    }
                    if constexpr (SB) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };
    const bool idle = tile0 >= RT;             // the last block of a patch may hold fewer tiles than waves x R
    if constexpr (H2_PRIO != 0) __builtin_amdgcn_s_setprio(0);
    if (!idle && H2_DBG_SKIP != 1) {
        if (nv == R) k_loop(std::integral_constant<int, R>{});
        else if constexpr (R * WAVES_M > TB || (H2_REBALANCE && FUSE1)) {
            if (nv == R - 1 || !(H2_REBALANCE && FUSE1)) k_loop(std::integral_constant<int, R - 1>{});
            else if constexpr (R >= 3) { if (nv == R - 2) k_loop(std::integral_constant<int, R - 2>{}); }
        }
    }

    float nxt_big = 0.f;
    if constexpr (H2_PRIO != 0) __builtin_amdgcn_s_setprio(3);
    if constexpr (NXT_W > 0) __syncthreads();          // every wave is done with the input tile the output overlays
    H2_STAMP(2);                                       // wave 0's k-loop
    H34_STAMP(5);                                      // conv3: k-loop + barrier
    if (idle) return;

#pragma unroll
    for (int n = 0; n < RN; n++) {
        const int co = (wn * RN + n) * 16 + l15;
        if constexpr (POOL) {
            float* o = out + (size_t)patch * (M / 4) * COUT;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int t = tile0 + r;
                float mx = acc[r][n][0] > acc[r][n][1] ? acc[r][n][0] : acc[r][n][1];
                const float m2 = acc[r][n][2] > acc[r][n][3] ? acc[r][n][2] : acc[r][n][3];
                mx = mx > m2 ? mx : m2;
                mx = mx * wscale_inv;                 // the weight scale is a power of two: exact (the bias is in the sum already)
                mx = mx > 0.f ? mx : 0.f;
                const int py = 2 * (t / (OW / 4)) + (kq >> 1), px = 2 * (t % (OW / 4)) + (kq & 1);
                if (r < nv && t < RT && co < COUT) o[(size_t)(py * (OW / 2) + px) * COUT + co] = mx;
            }
        } else if constexpr (NXT_W > 0) {
            static_assert(NT * 16 == NXT_CINP && M == NXT_W * NXT_W, "the next layer's tile holds exactly this layer's output");
            // (round 4: this relayout was 6.3 of the conv3 phase's 20 us per workgroup -- a division and a remainder by
            // NXT_W per stored VALUE; the four pixels a lane holds are consecutive, so one division per tile does)
#pragma unroll
            for (int r = 0; r < R; r++) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    v[e] = acc[r][n][e] * wscale_inv;
                    v[e] = (v[e] > 0.f && co < COUT) ? v[e] : 0.f;
                }
                nxt_big = fmaxf(nxt_big, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
                h2v hi2[2], lo2[2];
                split_h2x2(v[0], v[1], hi2[0], lo2[0]);
                split_h2x2(v[2], v[3], hi2[1], lo2[1]);
                const int m0 = (tile0 + r) * 16 + 4 * kq;
                int y = m0 / NXT_W, x = m0 - y * NXT_W;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (r < nv && m0 + e < M) {
                        _Float16* d = &lds[y * NXT_RS + x * NXT_PS + co];
                        d[0] = hi2[e >> 1][e & 1];
                        d[NXT_CINP] = lo2[e >> 1][e & 1];
                    }
                    if (++x == NXT_W) { x = 0; y++; }
                }
            }
        } else {
            float* o = out + (size_t)patch * M * COUT;
#pragma unroll
            for (int r = 0; r < R; r++) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int m = (tile0 + r) * 16 + 4 * kq + e;
                    float v = acc[r][n][e] * wscale_inv;
                    v = v > 0.f ? v : 0.f;
                    if (r < nv && m < M && co < COUT) o[(size_t)m * COUT + co] = v;
                }
            }
        }
    }
    if constexpr (NXT_W > 0) {
        if (overflow && !(nxt_big <= 65000.f)) *overflow = 1;
    }
    H2_STAMP(3);                                       // epilogue
    H34_STAMP(6);                                      // conv3: outputs split into conv4's tile
#if H2_DBG_TIME
    if (FUSE1 && threadIdx.x == 0) atomicAdd(&g_h2_prof[7], 1ull);
    if (FUSE1 && lane == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const unsigned wgid = patch * YB + blk_y;
        const unsigned slot = wgid * (NTHREADS / 64) + wave;
        if (slot < (unsigned)H2_LOG_CAP) {
            g_h2_log[3 * slot] = ((unsigned long long)wgid << 16) | ((xcc & 0xFu) << 8) | ((hwid >> 8) & 0xFFu);
            g_h2_log[3 * slot + 1] = t_born__;
            g_h2_log[3 * slot + 2] = wall_clock64();
        }
    }
#endif
}

template <int H, int W, int CIN, int KH, int KW, int COUT, int TB, int YB, int WAVES_M, int RN, bool POOL, int PF, bool SB, bool SWZ = false,
          bool FUSE1 = false>
// (two workgroups of 8 waves must fit a CU: 4 waves per SIMD = 128 VGPRs -- the second launch bound is HIP's minimum waves per SIMD; said explicitly, because one register more halves the occupancy)
__global__ __launch_bounds__(64 * WAVES_M * (cdiv(COUT, 16) / RN), (64 * WAVES_M * (cdiv(COUT, 16) / RN) >= 512 ? 4 : 1)) void conv_mfma16_h2_kernel(
    const float* __restrict__ in, const uint16_t* __restrict__ wt, const float* __restrict__ bias,
    float* __restrict__ out, float wscale_inv, int* __restrict__ overflow,
    const uint8_t* __restrict__ goban1 = nullptr, const uint16_t* __restrict__ wf1 = nullptr, const float* __restrict__ bias1 = nullptr)
{
    __shared__ __attribute__((aligned(16))) _Float16 lds[h2_tile_halves<H, W, CIN, KH, KW, TB, POOL, SWZ>()];
    conv_h2_body<H, W, CIN, KH, KW, COUT, TB, YB, WAVES_M, RN, POOL, PF, SB, SWZ, FUSE1>(lds, blockIdx.x, blockIdx.y, in, wt, bias, out,
                                                                                        wscale_inv, overflow, goban1, wf1, bias1);
}

// conv3 (3x3x32 -> 90, relu) and conv4 (3x3x90 -> 90, relu, 2x2 max-pool) of one patch in one workgroup: conv3's
// 14x14x90 output is written as hi/lo halves into conv4's LDS tile (which overlays conv3's own input tile once its
// k-loop is done) instead of going to HBM and back.  Same arithmetic in the same order as the two kernels apart.
__global__ __launch_bounds__(64 * 2 * (6 / H2C34_RN), H2C34_MINW) void conv34_h2_kernel(
    const float* __restrict__ in, const uint16_t* __restrict__ wt3, const float* __restrict__ bias3,
    const uint16_t* __restrict__ wt4, const float* __restrict__ bias4, float* __restrict__ out, float wscale_inv,
    int* __restrict__ overflow)
{
    constexpr int T3 = h2_tile_halves<16, 16, 32, 3, 3, 13, false, false>(), T4 = h2_tile_halves<14, 14, 90, 3, 3, 9, true, false>();
    constexpr int PS4 = 2 * 96 + h2_pspad<96, true>(), RS4 = lds_stride_b(14 * PS4, h2_rsrem<96, true, 12>(), 64);
    __shared__ __attribute__((aligned(16))) _Float16 lds[T3 > T4 ? T3 : T4];
#if H2_DBG_TIME
    const unsigned long long t0__ = wall_clock64();
#endif
    conv_h2_body<16, 16, 32, 3, 3, 90, 13, 1, 2, H2C34_RN, false, H2C3_PF, H2C3_SB, false, false, false, 14, PS4, RS4, 96>(
        lds, blockIdx.x, 0, in, wt3, bias3, nullptr, wscale_inv, overflow);
#if H2_DBG_TIME
    const unsigned long long t1__ = wall_clock64();
#endif
    __syncthreads();
#if H2_DBG_TIME
    const unsigned long long t2__ = wall_clock64();
#endif
    conv_h2_body<14, 14, 90, 3, 3, 90, 9, 1, 2, H2C34_RN, true, H2C34_PF, true, false, false, true>(
        lds, blockIdx.x, 0, nullptr, wt4, bias4, out, wscale_inv, overflow);
#if H2_DBG_TIME
    if (threadIdx.x == 0) {
        atomicAdd(&g_h34_prof[0], t1__ - t0__); atomicAdd(&g_h34_prof[1], t2__ - t1__); atomicAdd(&g_h34_prof[2], wall_clock64() - t2__);
        atomicAdd(&g_h34_prof[3], 1ull);
    }
#endif
}

// conv1 in split-precision mode.  The u8 pixels are exact in fp16, so only the weights are split and a product
// is two MFMAs (x*lo + x*hi).  K = 75 is laid out as 6 kernel rows x 16 slots (slot = kw*3 + cin, slot 15 and
// row 5 are zero weights): three k-steps of 32; a lane's fragment is 8 consecutive halves of one input row of
// the staged fp16 tile (unaligned, read as 8 ds_read_u16 -- there are only nine fragments per wave and unit).
// Same persistent unit walk, operand swap and 16-byte stores as conv1_mfma16_kernel.
//   wf : [2 channel tiles][3 steps][hi|lo][64 lanes][8] fp16 (pack_conv1_h2), weights x 2^8
template <int R>
__global__ __launch_bounds__(64 * (27 / R)) void conv1_h2_kernel(
    const uint8_t* __restrict__ goban, const uint16_t* __restrict__ wf, const float* __restrict__ bias,
    float* __restrict__ out, int nunits, float wscale_inv)
{
#pragma clang fp contract(off)
    constexpr int WAVES = 27 / R, NTHREADS = 64 * WAVES;
    constexpr int OW = 36, ROWS = 16, RS = 128;           // halves per staged row: 120 used, 8 zero
    constexpr int NPT = cdiv(ROWS * 30, NTHREADS);
    __shared__ _Float16 lds[(ROWS + 1) * RS];            // one extra all-zero row for kernel row 5
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;

    uint32_t raw[NPT];
    auto fetch = [&](int unit) {
#pragma unroll
        for (int q = 0; q < NPT; q++) {
            const int d = tid + q * NTHREADS;
            raw[q] = 0u;
            if (unit < nunits && d < ROWS * 30) {
                const int patch = unit / 3, by = unit % 3;
                const int frame = patch / 100, reg = patch % 100;
                const int py0 = region_origin(reg / 10) + 12 * by, px0 = region_origin(reg % 10);
                const uint8_t* src = goban + ((size_t)frame * 380 + py0 + d / 30) * 380 * 3 + (size_t)px0 * 3;
                raw[q] = reinterpret_cast<const uint32_t*>(src)[d % 30];
            }
        }
    };
    int unit = blockIdx.x;
    fetch(unit);
    for (int i = tid; i < ROWS * 8 + RS; i += NTHREADS) {                      // zero the padding columns and the extra row
        if (i < ROWS * 8) lds[(i / 8) * RS + 120 + i % 8] = (_Float16)0.f;
        else lds[ROWS * RS + i - ROWS * 8] = (_Float16)0.f;
    }
    h8 wq[3][2][2];
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int pl = 0; pl < 2; pl++)
                wq[s][n][pl] = __builtin_bit_cast(h8, reinterpret_cast<const uint4*>(wf)[((n * 3 + s) * 2 + pl) * 64 + lane]);
    float4 bv[2];
#pragma unroll
    for (int n = 0; n < 2; n++) bv[n] = *reinterpret_cast<const float4*>(bias + n * 16 + 4 * kq);
    int abase[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int m = (wave * R + r) * 16 + l15;
        abase[r] = (m / OW + (kq >> 1)) * RS + (m % OW) * 3 + 8 * (kq & 1);
    }

    for (; unit < nunits; unit += gridDim.x) {
#pragma unroll
        for (int q = 0; q < NPT; q++) {
            const int d = tid + q * NTHREADS;
            if (d < ROWS * 30) {
                _Float16* o = &lds[(d / 30) * RS + 4 * (d % 30)];
                o[0] = (_Float16)(float)(raw[q] & 0xFFu); o[1] = (_Float16)(float)((raw[q] >> 8) & 0xFFu);
                o[2] = (_Float16)(float)((raw[q] >> 16) & 0xFFu); o[3] = (_Float16)(float)(raw[q] >> 24);
            }
        }
        __syncthreads();
        fetch(unit + gridDim.x);

        f32x4 acc[R][2];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc[r][n][e] = 0.f;
#pragma unroll
        for (int s = 0; s < 3; s++) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const _Float16* ap = &lds[abase[r] + 2 * s * RS];
                h8 a;
#pragma unroll
                for (int e = 0; e < 8; e++) a[e] = ap[e];
#pragma unroll
                for (int n = 0; n < 2; n++) {
                    acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[s][n][1], a, acc[r][n], 0, 0, 0);
                    acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[s][n][0], a, acc[r][n], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        const int patch = unit / 3, by = unit % 3;
#pragma unroll
        for (int n = 0; n < 2; n++) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const size_t m = (size_t)patch * 1296 + (size_t)by * 432 + (wave * R + r) * 16 + l15;
                float4 v;
                v.x = acc[r][n][0] * wscale_inv + bv[n].x; v.y = acc[r][n][1] * wscale_inv + bv[n].y;
                v.z = acc[r][n][2] * wscale_inv + bv[n].z; v.w = acc[r][n][3] * wscale_inv + bv[n].w;
                v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                *reinterpret_cast<float4*>(out + m * 32 + n * 16 + 4 * kq) = v;
            }
        }
    }
}

// dense 3240 -> 160 + relu as an MFMA GEMM over patches (v_mfma_f32_16x16x4_f32, K ascending).
// Workgroup = 64 patches x all 160 outputs: wave w owns patches 16w..16w+15 and the ten 16-output
// tiles.  The activations are staged through LDS in chunks of 120 values per patch (coalesced
// float4 loads, pixel stride 130 dwords = 2 mod 32: conflict-free fragment reads); the weights
// stream from L2 as packed dwordx2 fragments ([tile][pair of k-steps][lane][2], pack_fc1) through
// a 3-deep register ring.  Operands are swapped (A = weights, B = activations) so a lane ends up
// with four consecutive outputs of one patch: 16-byte stores.
__global__ __launch_bounds__(256) void fc1_mfma16_kernel(const float* __restrict__ x, const float* __restrict__ wq_,
                                                         const float* __restrict__ bias, float* __restrict__ out, int npatch)
{
#pragma clang fp contract(off)
    constexpr int KIN = 3240, NOUT = 160, NT = 10, KC = 120, NCHUNK = KIN / KC, GC = KC / 8, CSR = 130;
    constexpr int SG2 = KIN / 8;                   // pairs of k-steps
    constexpr int PF = 3;
    static_assert(GC % PF == 0, "ring slots stay aligned across chunks");
    __shared__ float lds[64 * CSR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int p0 = blockIdx.x * 64;
    const float2* wq = reinterpret_cast<const float2*>(wq_) + lane;
    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
        for (int e = 0; e < 4; e++) acc[n][e] = 0.f;
    float2 bq[PF][NT];
#pragma unroll
    for (int u = 0; u < PF; u++)
#pragma unroll
        for (int n = 0; n < NT; n++) bq[u][n] = wq[((size_t)n * SG2 + u) * 64];
    const int abase = (wave * 16 + l15) * CSR + kq;
    for (int ch = 0; ch < NCHUNK; ch++) {
        __syncthreads();                           // previous chunk fully consumed
        for (int i = tid; i < 64 * (KC / 4); i += 256) {
            const int row = i / (KC / 4), c4 = i % (KC / 4);
            int p = p0 + row;
            p = p > npatch - 1 ? npatch - 1 : p;
            const float4 v = *reinterpret_cast<const float4*>(x + (size_t)p * KIN + ch * KC + 4 * c4);
            float* d = &lds[row * CSR + 4 * c4];
            *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
            *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
        }
        __syncthreads();
#pragma unroll
        for (int gl = 0; gl < GC; gl++) {
            const int g = ch * GC + gl;            // global pair index
            float2 b[NT];
#pragma unroll
            for (int n = 0; n < NT; n++) {
                b[n] = bq[gl % PF][n];
                if (g + PF < SG2) bq[gl % PF][n] = wq[((size_t)n * SG2 + g + PF) * 64];
            }
            const float a0 = lds[abase + 8 * gl], a1 = lds[abase + 8 * gl + 4];
#pragma unroll
            for (int n = 0; n < NT; n++) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[n].x, a0, acc[n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NT; n++) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[n].y, a1, acc[n], 0, 0, 0);
        }
    }
    const int p = p0 + wave * 16 + l15;
    if (p < npatch) {
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const float4 bv = *reinterpret_cast<const float4*>(bias + n * 16 + 4 * kq);
            float4 v;
            v.x = acc[n][0] + bv.x; v.y = acc[n][1] + bv.y; v.z = acc[n][2] + bv.z; v.w = acc[n][3] + bv.w;
            v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
            *reinterpret_cast<float4*>(out + (size_t)p * NOUT + n * 16 + 4 * kq) = v;
        }
    }
}

// The same dense layer in split precision (CK_CNN_F16X2): v_mfma_f32_16x16x32_f16, three MFMAs per product.
// The f32 kernel above streams the whole 2 MB weight matrix through every wave (16 patches) and is bound by the
// L2 for it; here a workgroup of 64 patches splits the OUTPUT tiles over its waves (tiles w, w + 4, w + 8), so each
// weight fragment is loaded once per workgroup, and every wave reads all 64 patches' activations from the LDS chunk
// (split into hi/lo halves once, when the chunk is staged; the next chunk's loads are in flight meanwhile).
//   wt : [10 output tiles][104 k-steps][hi|lo][64 lanes][8] fp16, weights x 2^8, K padded 3240 -> 3328 with zeros
__global__ __launch_bounds__(256) void fc1_h2_kernel(const float* __restrict__ x, const uint16_t* __restrict__ wt,
                                                     const float* __restrict__ bias, float* __restrict__ out, int npatch,
                                                     float wscale_inv, int* __restrict__ overflow)
{
#pragma clang fp contract(off)
    constexpr int KIN = 3240, NOUT = 160, KC = 128, NCH = 26, KS = 104, RSH = 2 * KC + 8;  // 264 halves per patch row
    constexpr int SPC = KC / 32;                    // k-steps per chunk: even, so the 2-slot weight ring stays in phase
    constexpr int NLD = 64 * (KC / 4) / 256;        // float4 loads per thread and chunk
    __shared__ __attribute__((aligned(16))) _Float16 lds[64 * RSH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int p0 = blockIdx.x * 64;
    const int ntw = wave < 2 ? 3 : 2;               // output tiles wave, wave + 4, wave + 8 (< 10)
    const uint4* wq = reinterpret_cast<const uint4*>(wt) + lane;
    f32x4 acc[3][4];
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[t][j][e] = 0.f;

    // Few workgroups (200 per 128-frame call), each a chain of 26 chunks: what the kernel waits for is latency, so everything
    // runs well ahead (round 5) -- the activations TWO chunks (two register sets), the weight fragments a whole chunk (a ring
    // of SPC slots: the fragments of k-step s + SPC are requested into the slot step s has just used).  One step ahead, as
    // first built, a call took 115 us: 104 k-steps x an L2 round trip.
    float4 raw[2][NLD];
    auto fetch = [&](int ch, float4 (&dst)[NLD]) {
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int i = tid + 256 * q, row = i / (KC / 4), c4 = i % (KC / 4);
            int p = p0 + row;
            p = p > npatch - 1 ? npatch - 1 : p;
            const int k = ch * KC + 4 * c4;
            dst[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ch < NCH && k < KIN) dst[q] = *reinterpret_cast<const float4*>(x + (size_t)p * KIN + k);
        }
    };
    uint4 wf[SPC][3][2];                             // [ring slot][tile][hi|lo]
    auto wload = [&](int slot, int s) {
#pragma unroll
        for (int t = 0; t < 3; t++)
            if (t < ntw) {
                wf[slot][t][0] = wq[(((size_t)(wave + 4 * t) * KS + s) * 2) * 64];
                wf[slot][t][1] = wq[(((size_t)(wave + 4 * t) * KS + s) * 2 + 1) * 64];
            }
    };
    fetch(0, raw[0]);
    fetch(1, raw[1]);
#pragma unroll
    for (int ks = 0; ks < SPC; ks++) wload(ks, ks);
    float big = 0.f;
    auto chunk = [&](int ch, float4 (&mine)[NLD]) {
        __syncthreads();                             // previous chunk fully consumed
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int i = tid + 256 * q, row = i / (KC / 4), c4 = i % (KC / 4);
            const float v[4] = {mine[q].x, mine[q].y, mine[q].z, mine[q].w};
            typedef _Float16 h4v __attribute__((ext_vector_type(4)));
            big = fmaxf(big, fmaxf(fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])), fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3]))));
            h2v h01, l01, h23, l23;
            split_h2x2(v[0], v[1], h01, l01);
            split_h2x2(v[2], v[3], h23, l23);
            const h4v hi = {h01[0], h01[1], h23[0], h23[1]}, lo = {l01[0], l01[1], l23[0], l23[1]};
            *reinterpret_cast<h4v*>(&lds[row * RSH + 4 * c4]) = hi;
            *reinterpret_cast<h4v*>(&lds[row * RSH + KC + 4 * c4]) = lo;
        }
        __syncthreads();
        fetch(ch + 2, mine);
#pragma unroll
        for (int ks = 0; ks < SPC; ks++) {
            const int s = ch * SPC + ks;
            h8 xh[4], xl[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const _Float16* ap = &lds[(16 * j + l15) * RSH + 32 * ks + 8 * kq];
                xh[j] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(ap));
                xl[j] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(ap + KC));
            }
#pragma unroll
            for (int t = 0; t < 3; t++)
                if (t < ntw) {
                    const h8 wh = __builtin_bit_cast(h8, wf[ks][t][0]), wl = __builtin_bit_cast(h8, wf[ks][t][1]);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[j], acc[t][j], 0, 0, 0);
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[j], acc[t][j], 0, 0, 0);
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[j], acc[t][j], 0, 0, 0);
                    }
                }
            if (s + SPC < KS) wload(ks, s + SPC);
        }
    };
    static_assert(NCH % 2 == 0, "26 chunks: pairs");
    for (int ch = 0; ch < NCH; ch += 2) {
        chunk(ch, raw[0]);
        chunk(ch + 1, raw[1]);
    }
    if (overflow && !(big <= 65000.f)) *overflow = 1;
#pragma unroll
    for (int t = 0; t < 3; t++)
        if (t < ntw) {
            const int o0 = (wave + 4 * t) * 16 + 4 * kq;
            const float4 bv = *reinterpret_cast<const float4*>(bias + o0);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int p = p0 + 16 * j + l15;
                float4 v;
                v.x = acc[t][j][0] * wscale_inv + bv.x; v.y = acc[t][j][1] * wscale_inv + bv.y;
                v.z = acc[t][j][2] * wscale_inv + bv.z; v.w = acc[t][j][3] * wscale_inv + bv.w;
                v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                if (p < npatch) *reinterpret_cast<float4*>(out + (size_t)p * NOUT + o0) = v;
            }
        }
}

// dense 160 -> 81 and softmax (nn_manager.py:295) on v_mfma_f32_16x16x4_f32: one wave = 16 patches x all 81 outputs (six
// 16-output tiles, the last one 1 real column), 40 k-steps of 4.  An f32 MFMA is bit for bit a k-ordered fmaf chain, i.e.
// the very sums of the round-1 kernel, which gave a wave ONE patch and a lane one output and loaded every weight per lane
// from L1: 0.37 us per frame for 2.6 MFLOP (a form with the weights as scalar operands of v_fmac was tried first: LDS reads
// and scalar loads share one counter, every k waited for the scalar cache -- slower than round 1's).  A lane ends up with
// four patches' logits of one output per tile; the softmax of a patch runs over the 16 lanes of a row group (DPP
// butterflies) and the six tiles.
__global__ __launch_bounds__(256) void fc2_softmax_kernel(const float* __restrict__ h1, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int npatch)
{
#pragma clang fp contract(off)
    constexpr int NO = 81, NT = 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, kq = lane >> 4;
    const int p0 = (blockIdx.x * 4 + wave) * 16;
    if (p0 >= npatch) return;
    int pa = p0 + l15;                                     // A operand: patch l15 of the group, k = 4 s + kq
    pa = pa < npatch ? pa : npatch - 1;
    const float* ha = h1 + (size_t)pa * 160 + kq;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int s = 0; s < 40; s++) {
        const float a = ha[4 * s];
        const float* wr = w + (size_t)(4 * s + kq) * NO + l15;          // B operand: w[k][16 t + l15]
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const float bw = 16 * t + l15 < NO ? wr[16 * t] : 0.f;
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw, acc[t], 0, 0, 0);
        }
    }
    // lane (l15, kq): logits of patches p0 + 4 kq + e (e = 0 .. 3) for the outputs 16 t + l15
    float v[NT][4];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int o = 16 * t + l15;
        const float bo = o < NO ? bias[o] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; e++) v[t][e] = o < NO ? acc[t][e] + bo : -INFINITY;
    }
#define CK_ROW16(V, OP)                                                                                              \
    { float t_;                                                                                                      \
      t_ = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V), 0xB1, 0xF, 0xF, false)); V = OP(V, t_);  /* quad_perm [1,0,3,2] */ \
      t_ = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V), 0x4E, 0xF, 0xF, false)); V = OP(V, t_);  /* quad_perm [2,3,0,1] */ \
      t_ = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V), 0x141, 0xF, 0xF, false)); V = OP(V, t_); /* row_half_mirror */     \
      t_ = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V), 0x140, 0xF, 0xF, false)); V = OP(V, t_); } /* row_mirror */
#define CK_FMAX(a, b) fmaxf(a, b)
#define CK_FADD(a, b) ((a) + (b))
#pragma unroll
    for (int e = 0; e < 4; e++) {
        float mx = v[0][e];
#pragma unroll
        for (int t = 1; t < NT; t++) mx = fmaxf(mx, v[t][e]);
        CK_ROW16(mx, CK_FMAX)
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            v[t][e] = expf(v[t][e] - mx);                 // exp(-inf) = 0 for the padding columns
            sum += v[t][e];
        }
        CK_ROW16(sum, CK_FADD)
        const int p = p0 + 4 * kq + e;
        if (p < npatch) {
#pragma unroll
            for (int t = 0; t < NT; t++)
                if (16 * t + l15 < NO) y[(size_t)p * NO + 16 * t + l15] = v[t][e] / sum;
        }
    }
#undef CK_ROW16
#undef CK_FMAX
#undef CK_FADD
}

}  // namespace

// K12 decode, one workgroup per frame.  NNCache.predict_all_stones writes the 100 regions in
// (i, j) raster order, so where region 9 overlaps region 8 (intersection row / column 17) the
// later region wins: cell (r, c) takes its value from region (I(r), I(c)) with I(17) = I(18) = 9.
namespace {
__global__ __launch_bounds__(128) void decode_kernel(const float* __restrict__ y, uint8_t* __restrict__ labels,
                                                     double* __restrict__ conf, int nframes, int* __restrict__ nonfinite,
                                                     uint8_t* __restrict__ region_label, double* __restrict__ region_conf)
{
    __shared__ int lab[100];
    __shared__ double cf[100];
    const int f = blockIdx.x, t = threadIdx.x;
    if (f >= nframes) return;
    if (t < 100) {
        const float* yy = y + ((size_t)f * 100 + t) * 81;
        int label = 0;
        double s = 0.0;
        for (int k = 0; k < 81; k++) {
            if (yy[k] > yy[label]) label = k;       // first maximum, like np.argmax
            s += (double)yy[k];                      // python sum(): float64, index order
        }
        lab[t] = label;
        cf[t] = (double)yy[label] / s;
        // the regions themselves (NNCache.predict_4_stones / predict_stone read them, nn_cache.py:16-31): where
        // region 8 and region 9 overlap the 19x19 grid below only keeps region 9's answer
        if (region_label) region_label[(size_t)f * 100 + t] = (uint8_t)label;
        if (region_conf) region_conf[(size_t)f * 100 + t] = cf[t];
        if (nonfinite && !(s - s == 0.0)) *nonfinite = 1;      // inf / NaN somewhere in this softmax
    }
    __syncthreads();
    for (int cell = t; cell < 361; cell += 128) {
        const int r = cell / 19, c = cell % 19;
        const int i = r >= 17 ? 9 : r / 2, j = c >= 17 ? 9 : c / 2;
        const int rs = i == 9 ? 17 : 2 * i, cs = j == 9 ? 17 : 2 * j;
        const int d = (r - rs) * 2 + (c - cs);       // base-3 digit index (compute_stones)
        int kk = lab[i * 10 + j];
        for (int q = 0; q < d; q++) kk /= 3;
        labels[(size_t)f * 361 + cell] = (uint8_t)(kk % 3);
        conf[(size_t)f * 361 + cell] = cf[i * 10 + j];
    }
}
}  // namespace

// B operand pack of conv_mfma16_f32_kernel: [channel tile of 16][group of 4 k-steps][lane = kslot*16 + col][4]
// with K = (kh, kw, cin padded to a multiple of 4) and k = 4*step + kslot; flip applied, padding zero
static void pack_mfma16(const float* k, int KH, int KW, int CIN, int COUT, std::vector<float>& dst)
{
    const int CINP = (CIN + 3) / 4 * 4, KS = KH * KW * CINP / 4, SG = (KS + 3) / 4, NT = (COUT + 15) / 16;
    dst.assign((size_t)NT * SG * 64 * 4, 0.f);
    for (int nt = 0; nt < NT; nt++)
        for (int step = 0; step < KS; step++)
            for (int lane = 0; lane < 64; lane++) {
                const int kidx = 4 * step + lane / 16, ij = kidx / CINP, c = kidx % CINP, o = nt * 16 + lane % 16;
                if (c >= CIN || o >= COUT) continue;
                const int i = ij / KW, j = ij % KW;
                dst[(((size_t)nt * SG + step / 4) * 64 + lane) * 4 + step % 4] =
                    k[(((size_t)(KH - 1 - i) * KW + (KW - 1 - j)) * CIN + c) * COUT + o];
            }
}

// A operand pack of conv1_mfma16_kernel: [channel tile][step][lane = kslot*16 + channel], K = 75 dense
// in (kh, kw, cin) order, k = 4*step + kslot, flip applied, k = 75 zero
static void pack_conv1(const float* k, std::vector<float>& dst)
{
    dst.assign((size_t)2 * 19 * 64, 0.f);
    for (int n = 0; n < 2; n++)
        for (int s = 0; s < 19; s++)
            for (int lane = 0; lane < 64; lane++) {
                const int kk = 4 * s + lane / 16, o = n * 16 + lane % 16;
                if (kk >= 75) continue;
                const int i = kk / 15, j = (kk % 15) / 3, c = kk % 3;
                dst[((size_t)n * 19 + s) * 64 + lane] = k[(((size_t)(4 - i) * 5 + (4 - j)) * 3 + c) * 32 + o];
            }
}

// A operand pack of fc1_mfma16_kernel: [output tile of 16][pair of k-steps][lane = kslot*16 + output][2],
// k = 4*step + kslot ascending
static void pack_fc1(const float* w, std::vector<float>& dst)
{
    const int KIN = 3240, NOUT = 160, SG2 = KIN / 8;
    dst.assign((size_t)(NOUT / 16) * SG2 * 64 * 2, 0.f);
    for (int nt = 0; nt < NOUT / 16; nt++)
        for (int step = 0; step < KIN / 4; step++)
            for (int lane = 0; lane < 64; lane++)
                dst[(((size_t)nt * SG2 + step / 2) * 64 + lane) * 2 + step % 2] =
                    w[(size_t)(4 * step + lane / 16) * NOUT + nt * 16 + lane % 16];
}

int k_cnn_pack_weights(ck_ctx* ctx, const float* const w[12], int space)
{
    static const size_t counts[12] = { 5 * 5 * 3 * 32, 32, 5 * 5 * 32 * 32, 32, 3 * 3 * 32 * 90, 90,
                                       3 * 3 * 90 * 90, 90, 3240 * 160, 160, 160 * 81, 81 };
    std::vector<std::vector<float>> host(12);
    for (int i = 0; i < 12; i++) {
        host[i].resize(counts[i]);
        if (space == CK_DEVICE) CK_HIP(ctx, hipMemcpy(host[i].data(), w[i], counts[i] * 4, hipMemcpyDeviceToHost));
        else if (space == CK_HOST) memcpy(host[i].data(), w[i], counts[i] * 4);
        else return ck_fail(ctx, CK_ERR_ARG, "bad memory space %d", space);
    }
    auto up = [&](DevBuf& b, const std::vector<float>& v) -> int {
        CK_TRY(ck_ensure(ctx, b, v.size() * 4));
        CK_HIP(ctx, hipMemcpy(b.p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
        return CK_OK;
    };
    std::vector<float> t;
    pack_conv1(host[0].data(), t);              CK_TRY(up(ctx->cnn.c1w, t));
    pack_mfma16(host[2].data(), 5, 5, 32, 32, t); CK_TRY(up(ctx->cnn.c2w, t));
    pack_mfma16(host[4].data(), 3, 3, 32, 90, t); CK_TRY(up(ctx->cnn.c3w, t));
    pack_mfma16(host[6].data(), 3, 3, 90, 90, t); CK_TRY(up(ctx->cnn.c4w, t));
    CK_TRY(up(ctx->cnn.c1b, host[1])); CK_TRY(up(ctx->cnn.c2b, host[3]));
    CK_TRY(up(ctx->cnn.c3b, host[5])); CK_TRY(up(ctx->cnn.c4b, host[7]));
    pack_fc1(host[8].data(), t);               CK_TRY(up(ctx->cnn.d1w, t));
    CK_TRY(up(ctx->cnn.d1b, host[9]));
    CK_TRY(up(ctx->cnn.d2w, host[10])); CK_TRY(up(ctx->cnn.d2b, host[11]));
    // bf16 packs in MFMA fragment order: [16-channel tile][k-step][lane = kslot*16 + channel][8 consecutive cin],
    // k-step = 32 input channels of one kernel tap; flip applied, padding channels zero
    auto pack_bf = [&](const float* k, int KH, int KW, int CIN, int CINP, int COUT, int COUTS, DevBuf& dst) -> int {
        const int KS = KH * KW * (CINP / 32), NT = COUTS / 16;
        std::vector<uint16_t> v((size_t)NT * KS * 64 * 8, 0);
        for (int nt = 0; nt < NT; nt++)
            for (int i = 0; i < KH; i++)
                for (int j = 0; j < KW; j++)
                    for (int cc = 0; cc < CINP / 32; cc++)
                        for (int lane = 0; lane < 64; lane++)
                            for (int e = 0; e < 8; e++) {
                                const int c = 32 * cc + 8 * (lane / 16) + e, o = nt * 16 + lane % 16;
                                if (c >= CIN || o >= COUT) continue;
                                const int step = (i * KW + j) * (CINP / 32) + cc;
                                v[(((size_t)nt * KS + step) * 64 + lane) * 8 + e] =
                                    f2bf(k[(((size_t)(KH - 1 - i) * KW + (KW - 1 - j)) * CIN + c) * COUT + o]);
                            }
        CK_TRY(ck_ensure(ctx, dst, v.size() * 2));
        CK_HIP(ctx, hipMemcpy(dst.p, v.data(), v.size() * 2, hipMemcpyHostToDevice));
        return CK_OK;
    };
    CK_TRY(k_cnn_bf16_pack_conv1(ctx, host[0].data(), ctx->cnn.c1w_f16));
    CK_TRY(k_cnn_bf16_pack_fc1(ctx, host[8].data(), ctx->cnn.d1w_bfp));
    CK_TRY(pack_bf(host[2].data(), 5, 5, 32, 32, 32, 32, ctx->cnn.c2w_bf));
    CK_TRY(pack_bf(host[4].data(), 3, 3, 32, 32, 90, 96, ctx->cnn.c3w_bf));
    CK_TRY(pack_bf(host[6].data(), 3, 3, 90, 96, 90, 96, ctx->cnn.c4w_bf));
    // split-precision packs: weights x 2^8 as hi / lo fp16 planes in MFMA fragment order
    // [16-channel tile][k-step][plane][lane = kslot*16 + channel][8 consecutive cin], flip applied
    auto pack_h2 = [&](const float* k, int KH, int KW, int CIN, int COUT, DevBuf& dst) -> int {
        const int CINP = (CIN + 31) / 32 * 32, KS = KH * KW * (CINP / 32), NT = (COUT + 15) / 16;
        std::vector<uint16_t> v((size_t)NT * KS * 2 * 64 * 8, 0);
        for (int nt = 0; nt < NT; nt++)
            for (int i = 0; i < KH; i++)
                for (int j = 0; j < KW; j++)
                    for (int cc = 0; cc < CINP / 32; cc++)
                        for (int lane = 0; lane < 64; lane++)
                            for (int e = 0; e < 8; e++) {
                                const int c = 32 * cc + 8 * (lane / 16) + e, o = nt * 16 + lane % 16;
                                if (c >= CIN || o >= COUT) continue;
                                const int step = (i * KW + j) * (CINP / 32) + cc;
                                const float wv = k[(((size_t)(KH - 1 - i) * KW + (KW - 1 - j)) * CIN + c) * COUT + o] * H2_WSCALE;
                                const _Float16 hi = (_Float16)wv;
                                const _Float16 lo = (_Float16)(wv - (float)hi);
                                const size_t base = (((size_t)nt * KS + step) * 2) * 64 * 8 + (size_t)lane * 8 + e;
                                memcpy(&v[base], &hi, 2);
                                memcpy(&v[base + 64 * 8], &lo, 2);
                            }
        CK_TRY(ck_ensure(ctx, dst, v.size() * 2));
        CK_HIP(ctx, hipMemcpy(dst.p, v.data(), v.size() * 2, hipMemcpyHostToDevice));
        return CK_OK;
    };
    {   // conv1: [channel tile][step][plane][lane][8]; k = (kernel row 2*step + kslot/2, slot 8*(kslot%2) + e), slot = kw*3 + cin
        std::vector<uint16_t> v((size_t)2 * 3 * 2 * 64 * 8, 0);
        for (int nt = 0; nt < 2; nt++)
            for (int st = 0; st < 3; st++)
                for (int lane = 0; lane < 64; lane++)
                    for (int e = 0; e < 8; e++) {
                        const int kqq = lane / 16, i = 2 * st + (kqq >> 1), slot = 8 * (kqq & 1) + e, o = nt * 16 + lane % 16;
                        if (i > 4 || slot > 14) continue;
                        const int j = slot / 3, c = slot % 3;
                        const float wv = host[0][(((size_t)(4 - i) * 5 + (4 - j)) * 3 + c) * 32 + o] * H2_WSCALE;
                        const _Float16 hi = (_Float16)wv;
                        const _Float16 lo = (_Float16)(wv - (float)hi);
                        const size_t base = (((size_t)nt * 3 + st) * 2) * 64 * 8 + (size_t)lane * 8 + e;
                        memcpy(&v[base], &hi, 2);
                        memcpy(&v[base + 64 * 8], &lo, 2);
                    }
        CK_TRY(ck_ensure(ctx, ctx->cnn.c1w_h2, v.size() * 2));
        CK_HIP(ctx, hipMemcpy(ctx->cnn.c1w_h2.p, v.data(), v.size() * 2, hipMemcpyHostToDevice));
    }
    CK_TRY(pack_h2(host[2].data(), 5, 5, 32, 32, ctx->cnn.c2w_h2));
    CK_TRY(pack_h2(host[4].data(), 3, 3, 32, 90, ctx->cnn.c3w_h2));
    CK_TRY(pack_h2(host[6].data(), 3, 3, 90, 90, ctx->cnn.c4w_h2));
    CK_TRY(k_cnn_q8_pack(ctx, host[0].data(), host[2].data(), host[4].data(), host[6].data()));
    {   // dense1 for fc1_h2_kernel: [output tile][k-step][plane][lane = kslot*16 + output][8 consecutive k], weights x 2^8
        std::vector<uint16_t> v((size_t)10 * 104 * 2 * 64 * 8, 0);
        for (int t = 0; t < 10; t++)
            for (int st = 0; st < 104; st++)
                for (int lane = 0; lane < 64; lane++)
                    for (int e = 0; e < 8; e++) {
                        const int k = 32 * st + 8 * (lane / 16) + e, o = 16 * t + lane % 16;
                        if (k >= 3240) continue;
                        const float wv = host[8][(size_t)k * 160 + o] * H2_WSCALE;
                        const _Float16 hi = (_Float16)wv;
                        const _Float16 lo = (_Float16)(wv - (float)hi);
                        const size_t base = (((size_t)t * 104 + st) * 2) * 64 * 8 + (size_t)lane * 8 + e;
                        memcpy(&v[base], &hi, 2);
                        memcpy(&v[base + 64 * 8], &lo, 2);
                    }
        CK_TRY(ck_ensure(ctx, ctx->cnn.d1w_h2, v.size() * 2));
        CK_HIP(ctx, hipMemcpy(ctx->cnn.d1w_h2.p, v.data(), v.size() * 2, hipMemcpyHostToDevice));
    }
    ctx->cnn.set = true;
    return CK_OK;
}

// Tuning knob (developer only): extra dynamic LDS requested by the two big classifier kernels.  Enough of it leaves ONE
// workgroup per CU instead of two, i.e. room (registers, wave slots) for waves of the board path running on other streams.
static int lds_pad_conv2() { static const int v = getenv("CK_CONV2_LDS_PAD") ? atoi(getenv("CK_CONV2_LDS_PAD")) : 0; return v; }
static int lds_pad_conv34() { static const int v = getenv("CK_CONV34_LDS_PAD") ? atoi(getenv("CK_CONV34_LDS_PAD")) : 0; return v; }

int k_cnn_predict(ck_ctx* ctx, const uint8_t* d_goban, int nframes, float* d_y, uint8_t* d_labels, double* d_conf,
                  int* d_nonfinite, uint8_t* d_rlabel, double* d_rconf)
{
    // convolutions run in chunks of frames so the activation scratch stays bounded; the dense
    // tail runs once over the whole batch (its 32-patch MFMA tiles need many waves in flight)
    const int CHUNK = 128;
    const size_t a1_sz = (size_t)CHUNK * 100 * 36 * 36 * 32 * 4;      // conv1 out (36*36*32), later conv3 out (14*14*90)
    const size_t p2_sz = (size_t)CHUNK * 100 * 12 * 12 * 90 * 4;      // conv2+pool out (16*16*32), later conv4 out (12*12*90)
    CK_TRY(ck_ensure(ctx, ctx->act0, a1_sz));
    CK_TRY(ck_ensure(ctx, ctx->act1, p2_sz));
    CK_TRY(ck_ensure(ctx, ctx->act2, (size_t)nframes * 100 * (3240 + 160) * 4));
    float* a1 = (float*)ctx->act0.p;
    float* p2 = (float*)ctx->act1.p;
    float* p4_all = (float*)ctx->act2.p;
    float* h1 = p4_all + (size_t)nframes * 100 * 3240;
    const CnnWeights& W = ctx->cnn;
    if (ctx->cnn_mode == CK_CNN_BF16) {
        // k_cnn_bf16.hip: conv1 + conv2 and conv3 + conv4 fused, one bf16 MFMA per product (conv1 on the fp16 pipe: its u8
        // operand is exact there); the first dense layer on v_mfma_f32_32x32x16_bf16
        uint16_t* b2 = (uint16_t*)ctx->act1.p;           // pooled conv2 output of a chunk: 16*16*32 per patch
        uint16_t* q4_all = (uint16_t*)ctx->act2.p;       // pooled conv4 output, all frames: 36*96 per patch
        float* hb = (float*)(q4_all + (((size_t)nframes * 100 * 3456 + 7) & ~(size_t)7));
        for (int f0 = 0; f0 < nframes; f0 += CHUNK) {
            const int nf = nframes - f0 < CHUNK ? nframes - f0 : CHUNK;
            CK_TRY(k_cnn_bf16_convs(ctx, d_goban + (size_t)f0 * 380 * 380 * 3, nf * 100, b2, q4_all + (size_t)f0 * 100 * 3456));
        }
        TimeScope ts(ctx, "cnn_tail");
        const int np = nframes * 100;
        CK_TRY(k_cnn_bf16_fc1(ctx, q4_all, np, hb));
        hipLaunchKernelGGL(fc2_softmax_kernel, dim3((np + 63) / 64), dim3(256), 0, ctx->stream, (const float*)hb,
                           (const float*)W.d2w.p, (const float*)W.d2b.p, d_y, np);
        hipLaunchKernelGGL(decode_kernel, dim3(nframes), dim3(128), 0, ctx->stream, (const float*)d_y, d_labels, d_conf, nframes, d_nonfinite, d_rlabel, d_rconf);
        CK_HIP(ctx, hipGetLastError());
        return CK_OK;
    }
    const bool h2 = ck_cnn_split(ctx->cnn_mode);
    const bool q8 = ctx->cnn_mode == CK_CNN_F16Q8 && W.q8_ok;      // (weights outside the e4m3 range: the three-MFMA kernels)
    for (int f0 = 0; f0 < nframes; f0 += CHUNK) {
        const int nf = nframes - f0 < CHUNK ? nframes - f0 : CHUNK;
        const int np = nf * 100;
        const uint8_t* gob = d_goban + (size_t)f0 * 380 * 380 * 3;
        float* p4 = p4_all + (size_t)f0 * 100 * 3240;
        {
            TimeScope ts(ctx, "cnn_conv1");
            if (q8 || (h2 && H2C2_SWZ && H2_FUSE1)) {
                // conv1 is computed inside conv2's staging (below)
            } else if (h2)
                hipLaunchKernelGGL((conv1_h2_kernel<C1_R>), dim3(std::min(np * 3, C1_GRID)), dim3(64 * (27 / C1_R)), 0, ctx->stream, gob,
                                   (const uint16_t*)W.c1w_h2.p, (const float*)W.c1b.p, a1, np * 3, 1.f / H2_WSCALE);
            else
            hipLaunchKernelGGL((conv1_mfma16_kernel<C1_R>), dim3(std::min(np * 3, C1_GRID)), dim3(64 * (27 / C1_R)), 0,
                               ctx->stream, gob, (const float*)W.c1w.p, (const float*)W.c1b.p, a1, np * 3);
        }
        {
            TimeScope ts(ctx, "cnn_conv2");
            // 32 rows: 4 groups of (4 waves x 2 rows), pooled output 16x16x32
            // 8x8 pooling tiles of 4x4 pixels: workgroups of whole tile rows, pooled output 16x16x32
            if (q8) {
                CK_TRY(k_cnn_q8_conv12(ctx, gob, np, p2, d_nonfinite));
            } else if (h2) {
#if H2C2_SWZ
                hipLaunchKernelGGL((conv_mfma16_h2_kernel<36, 36, 32, 5, 5, 32, H2C2S_TB, 64 / H2C2S_TB + (64 % H2C2S_TB != 0), H2C2S_WM, 2, true, H2C2S_PF, H2C2S_SB, true, H2_FUSE1 != 0>), dim3(np, 64 / H2C2S_TB + (64 % H2C2S_TB != 0)), dim3(64 * H2C2S_WM), (size_t)lds_pad_conv2(), ctx->stream,
                                   (const float*)a1, (const uint16_t*)W.c2w_h2.p, (const float*)W.c2b.p, p2, 1.f / H2_WSCALE, d_nonfinite,
                                   gob, (const uint16_t*)W.c1w_h2.p, (const float*)W.c1b.p);
#if H2_DBG_TIME
                {
                    unsigned long long hp[8];
                    (void)hipStreamSynchronize(ctx->stream);
                    (void)hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_h2_prof), sizeof hp);
                    if (hp[7]) fprintf(stderr, "[conv2 phases, us per workgroup over %llu workgroups] stage %.2f  conv1 tiles (wave 0) %.2f  wait for the others %.2f  k-loop %.2f  epilogue %.2f\n", hp[7],
                                       hp[0] * 0.01 / hp[7], hp[4] * 0.01 / hp[7], hp[1] * 0.01 / hp[7], hp[2] * 0.01 / hp[7], hp[3] * 0.01 / hp[7]);
                    memset(hp, 0, sizeof hp);
                    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_h2_prof), hp, sizeof hp);
                    // residency per CU from the per-wave log
                    const int nw = std::min(np * 3 * 8, H2_LOG_CAP);
                    std::vector<unsigned long long> lg((size_t)3 * nw);
                    (void)hipMemcpyFromSymbol(lg.data(), HIP_SYMBOL(g_h2_log), lg.size() * 8);
                    struct Wg { unsigned long long a, b, wsum; int n; };
                    std::map<unsigned, std::map<unsigned, Wg>> cus;          // CU key -> workgroup id -> its interval
                    unsigned long long t_lo = ~0ull, t_hi = 0;
                    for (int i = 0; i < nw; i++) {
                        const unsigned long long key = lg[3 * i], a = lg[3 * i + 1], b = lg[3 * i + 2];
                        if (!b) continue;
                        Wg& g = cus[(unsigned)(key & 0xFFFF)][(unsigned)(key >> 16)];
                        if (!g.n) { g.a = a; g.b = b; } else { g.a = std::min(g.a, a); g.b = std::max(g.b, b); }
                        g.wsum += b - a; g.n++;
                        t_lo = std::min(t_lo, a); t_hi = std::max(t_hi, b);
                    }
                    double life = 0, wlife = 0, gap = 0, resid = 0; size_t nwg = 0, ngap = 0, nwave = 0;
                    std::vector<double> gaps;
                    for (auto& cu : cus) {
                        std::vector<std::pair<unsigned long long, int>> ev;
                        double busy = 0;
                        for (auto& w : cu.second) {
                            life += (double)(w.second.b - w.second.a); wlife += (double)w.second.wsum; nwave += w.second.n; nwg++;
                            busy += (double)(w.second.b - w.second.a);
                            ev.push_back({w.second.a, +1}); ev.push_back({w.second.b, -1});
                        }
                        resid += busy;
                        std::sort(ev.begin(), ev.end());
                        // a slot falls empty at every end; the next start on this CU fills it
                        std::vector<unsigned long long> ends;
                        for (auto& e : ev) {
                            if (e.second < 0) ends.push_back(e.first);
                            else if (!ends.empty()) { gaps.push_back((double)(e.first - ends.front())); ends.erase(ends.begin()); }
                        }
                    }
                    std::sort(gaps.begin(), gaps.end());
                    for (double g : gaps) gap += g;
                    ngap = gaps.size();
                    fprintf(stderr, "[conv2 residency] %zu CUs, %zu workgroups: workgroup lifetime %.2f us (its waves %.2f), %.2f workgroups resident per CU over the kernel's %.0f us; "
                                    "slot empty between two workgroups: mean %.2f us, median %.2f, p90 %.2f (%zu refills)\n",
                            cus.size(), nwg, 0.01 * life / std::max<size_t>(nwg, 1), 0.01 * wlife / std::max<size_t>(nwave, 1),
                            resid / ((double)(t_hi - t_lo) * std::max<size_t>(cus.size(), 1)), 0.01 * (double)(t_hi - t_lo),
                            0.01 * gap / std::max<size_t>(ngap, 1), ngap ? 0.01 * gaps[ngap / 2] : 0.0, ngap ? 0.01 * gaps[ngap * 9 / 10] : 0.0, ngap);
                    std::fill(lg.begin(), lg.end(), 0ull);
                    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_h2_log), lg.data(), lg.size() * 8);
                }
#endif
#else
                hipLaunchKernelGGL((conv_mfma16_h2_kernel<36, 36, 32, 5, 5, 32, H2C2_TB, 64 / H2C2_TB, H2C2_WM, 2, true, H2C2_PF, true>), dim3(np, 64 / H2C2_TB), dim3(64 * H2C2_WM), 0, ctx->stream,
                                   (const float*)a1, (const uint16_t*)W.c2w_h2.p, (const float*)W.c2b.p, p2, 1.f / H2_WSCALE, d_nonfinite);
#endif
            }
            else
            hipLaunchKernelGGL((conv_mfma16_f32_kernel<36, 36, 32, 5, 5, 32, C2_TB, 64 / C2_TB, C2_WM, C2_RN, true>), dim3(np, 64 / C2_TB),
                               dim3(64 * C2_WM * (2 / C2_RN)), 0, ctx->stream, (const float*)a1, (const float*)W.c2w.p,
                               (const float*)W.c2b.p, p2);
        }
        float* a3 = a1;
        if (q8) {
            TimeScope ts(ctx, "cnn_conv4");
            CK_TRY(k_cnn_q8_conv34(ctx, p2, np, p4, d_nonfinite));
        } else if (h2 && H2_FUSE34) {
            TimeScope ts(ctx, "cnn_conv4");
            // conv3 + conv4 of a patch in one workgroup; pooled 6x6x90 written directly
            hipLaunchKernelGGL(conv34_h2_kernel, dim3(np), dim3(64 * 2 * (6 / H2C34_RN)), (size_t)lds_pad_conv34(), ctx->stream, (const float*)p2, (const uint16_t*)W.c3w_h2.p,
                               (const float*)W.c3b.p, (const uint16_t*)W.c4w_h2.p, (const float*)W.c4b.p, p4, 1.f / H2_WSCALE, d_nonfinite);
#if H2_DBG_TIME
            {
                unsigned long long hp[8];
                (void)hipStreamSynchronize(ctx->stream);
                (void)hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_h34_prof), sizeof hp);
                if (hp[3]) fprintf(stderr, "[conv3+4 phases, us per workgroup over %llu workgroups] stage + conv3 + relayout (wave 0) %.2f (stage %.2f, k-loop %.2f, relayout %.2f)  barrier %.2f  conv4 + epilogue %.2f\n",
                                   hp[3], hp[0] * 0.01 / hp[3], hp[4] * 0.01 / hp[3], hp[5] * 0.01 / hp[3], hp[6] * 0.01 / hp[3], hp[1] * 0.01 / hp[3], hp[2] * 0.01 / hp[3]);
                memset(hp, 0, sizeof hp);
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_h34_prof), hp, sizeof hp);
            }
#endif
        } else {
            {
                TimeScope ts(ctx, "cnn_conv3");
                // 13 pixel tiles x 6 channel tiles of 16
                if (h2)
                    hipLaunchKernelGGL((conv_mfma16_h2_kernel<16, 16, 32, 3, 3, 90, 13, 1, 2, 3, false, H2C3_PF, H2C3_SB>), dim3(np), dim3(256), 0, ctx->stream,
                                       (const float*)p2, (const uint16_t*)W.c3w_h2.p, (const float*)W.c3b.p, a3, 1.f / H2_WSCALE, d_nonfinite);
                else
                hipLaunchKernelGGL((conv_mfma16_f32_kernel<16, 16, 32, 3, 3, 90, 13, 1, C3_WM, 1, false>), dim3(np), dim3(384 * C3_WM), 0,
                                   ctx->stream, (const float*)p2, (const float*)W.c3w.p, (const float*)W.c3b.p, a3);
            }

            {
                TimeScope ts(ctx, "cnn_conv4");
                // 9 tiles of four pooling windows x 6 channel tiles; pooled 6x6x90 written directly
                if (h2)
                    hipLaunchKernelGGL((conv_mfma16_h2_kernel<14, 14, 90, 3, 3, 90, 9, 1, 2, 3, true, H2C34_PF, true>), dim3(np), dim3(256), 0, ctx->stream,
                                       (const float*)a3, (const uint16_t*)W.c4w_h2.p, (const float*)W.c4b.p, p4, 1.f / H2_WSCALE, d_nonfinite);
                else
                hipLaunchKernelGGL((conv_mfma16_f32_kernel<14, 14, 90, 3, 3, 90, 9, 1, C4_WM, 1, true>), dim3(np), dim3(384 * C4_WM), 0,
                                   ctx->stream, (const float*)a3, (const float*)W.c4w.p, (const float*)W.c4b.p, p4);
            }
        }
        CK_HIP(ctx, hipGetLastError());
    }
    {
        TimeScope ts(ctx, "cnn_tail");
        const int np = nframes * 100;
        if (h2 && H2_FC1)
            hipLaunchKernelGGL(fc1_h2_kernel, dim3((np + 63) / 64), dim3(256), 0, ctx->stream, (const float*)p4_all,
                               (const uint16_t*)W.d1w_h2.p, (const float*)W.d1b.p, h1, np, 1.f / H2_WSCALE, d_nonfinite);
        else
        hipLaunchKernelGGL(fc1_mfma16_kernel, dim3((np + 63) / 64), dim3(256), 0, ctx->stream,
                           (const float*)p4_all, (const float*)W.d1w.p, (const float*)W.d1b.p, h1, np);
        hipLaunchKernelGGL(fc2_softmax_kernel, dim3((np + 63) / 64), dim3(256), 0, ctx->stream, (const float*)h1,
                           (const float*)W.d2w.p, (const float*)W.d2b.p, d_y, np);
        hipLaunchKernelGGL(decode_kernel, dim3(nframes), dim3(128), 0, ctx->stream, (const float*)d_y, d_labels, d_conf, nframes, d_nonfinite, d_rlabel, d_rconf);
        CK_HIP(ctx, hipGetLastError());
    }
    return CK_OK;
}
