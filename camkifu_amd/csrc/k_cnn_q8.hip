// k_cnn_q8.hip -- the convolutions of the stone classifier in CK_CNN_F16Q8 mode (reference: NNManager.create_net,
// src/camkifu/stone/nn_manager.py:277-298; patches :216-218, 256-275): split precision with the cross terms in 8 bits.
//
// The split-precision mode (k_cnn.hip) writes an f32 product a * w as a_hi * w_hi + a_hi * w_lo + a_lo * w_hi on the fp16
// MFMA: three instructions of 16 cycles per 32 k.  The two cross terms are 2^-11 of the main term, so their operands need
// only a few bits -- here they are rounded to OCP e4m3 and go through the block-scaled MFMA
// (v_mfma_scale_f32_16x16x128_f8f6f4, tools/micro/mfma_scale_f8.hip): ONE instruction covers both cross terms of TWO taps
// (four blocks of 32 k: [tap A: a_hi | a_lo, tap B: a_hi | a_lo] against [w_lo | w_hi, w_lo | w_hi]) with the 2^-11 in
// the blocks' E8M0 scales, into the SAME f32 accumulator as the main term.  Per tap 16 + 16 cycles of matrix pipe instead
// of 48; the maps stay within 5e-5 of their scale (tools/sim_split_q8.py; bar 1e-4).
//
// Same blocking as k_cnn_bf16.hip (a pixel tile is 16 pixels of one row, a wave owns a column strip of output rows, an
// input row's fragment serves the vertical taps of several output rows, the 2x2 max-pool stays in the lane).  The k-loop
// is a sequence of SWEEPS over the strip's input rows: one per tap column for the main term (fp16 fragments from the hi
// plane), one per PAIR of tap columns for the cross terms (e4m3 fragments of the two columns from the q plane, 8 registers).
//   conv12_q8_kernel  conv1 (5x5x3 -> 32, relu) + conv2 (5x5x32 -> 32, relu, 2x2 max-pool) of a THIRD of a patch per workgroup
//   conv34_q8_kernel  conv3 (3x3x32 -> 90, relu) + conv4 (3x3x90 -> 90, relu, 2x2 max-pool) of one patch per workgroup
// Activations in LDS: per pixel and 32 channels a hi plane (32 halves, round-to-nearest fp16 of the f32 value) and a q
// plane (64 bytes: e4m3(hi / 4) of the 32 channels, then e4m3((value - hi) * 2^11 / 4)).  Both planes have the geometry of
// the bf16 kernels' tiles (64 bytes per pixel and 32 channels), hence their swizzles.  Weights: x 2^8 as in k_cnn.hip; the
// main term reads the hi planes of the c?w_h2 packs, the cross terms their own e4m3 packs (k_cnn_q8_pack).
// A value beyond the e4m3 range (|x| >= 1792 with the block scale 4; the conversion then gives NaN) raises the same flag
// as an fp16 overflow in k_cnn.hip and the batch is recomputed by the f32 chain (ck_api.hip: cnn_finish).
#include <algorithm>
#include <map>
#include <type_traits>
#include <utility>

#include "ck_common.h"

#ifndef Q8_C2_D
#define Q8_C2_D 4            // conv2: depth of the fragment ring (reads in flight + the one in use)
#endif
#ifndef Q8_C3_D
#define Q8_C3_D 3            // conv3: depth of the fragment ring
#endif
#ifndef Q8_C3_PF
#define Q8_C3_PF 2           // conv3: sweeps the weight fragments run ahead (a main sweep is 336 pipe cycles: one ahead does not cover an L2 round trip)
#endif
#ifndef Q8_C4_PF
#define Q8_C4_PF 3           // conv4: the same (a main sweep of five tiles is 240 pipe cycles)
#endif
#ifndef Q8_C4_D
#define Q8_C4_D 6            // conv4: depth of the fragment ring
#endif
#ifndef Q8_PRIO
#define Q8_PRIO 1            // wave priority 3 outside the k-loops (staging, conv1, relayout, epilogue): these short phases of loads, LDS traffic
#endif                       // and vector arithmetic otherwise wait behind the other workgroup's matrix instructions for every issue slot
#ifndef Q8_DBG_TIME
#define Q8_DBG_TIME 0        // profiling aid: phase times per workgroup (thread 0, 100 MHz wall clock) summed into g_q8_prof
#endif

namespace {

#if Q8_DBG_TIME
__device__ unsigned long long g_q8_prof[16];
constexpr int Q8_LOG_CAP = 1 << 16;
__device__ unsigned long long g_q8_log[2][3 * Q8_LOG_CAP];           // per kernel and workgroup: CU key, first and last stamp of thread 0
__device__ __forceinline__ void q8_log(int which, unsigned long long t_born)
{
    if (threadIdx.x == 0 && blockIdx.x < Q8_LOG_CAP) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_q8_log[which][3 * blockIdx.x] = 1ull | ((unsigned long long)((xcc & 0xFu) << 8 | ((hwid >> 8) & 0xFFu)) << 8);
        g_q8_log[which][3 * blockIdx.x + 1] = t_born;
        g_q8_log[which][3 * blockIdx.x + 2] = wall_clock64();
    }
}
#define Q8_STAMP(K) do { if (threadIdx.x == 0) { const unsigned long long now__ = wall_clock64(); atomicAdd(&g_q8_prof[K], now__ - t_prev__); t_prev__ = now__; } } while (0)
#define Q8_STAMP_BEGIN unsigned long long t_prev__ = wall_clock64(); const unsigned long long t_born__ = t_prev__
#else
#define Q8_STAMP(K) do { } while (0)
#define Q8_STAMP_BEGIN do { } while (0)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr float Q8_WSCALE = 256.f;        // = H2_WSCALE of k_cnn.hip: the c?w_h2 packs hold w x 2^8
constexpr int Q8_SA = 2, Q8_SW = 2;       // block scales 2^2 of the e4m3 operands (activations, weights): range +-1792
constexpr float Q8_LIMIT = 1700.f;        // an activation above this raises the overflow flag

__device__ __forceinline__ int region_origin(int i) { return i == 9 ? 340 : 40 * i; }
__device__ __forceinline__ int swz32(int x) { return (x >> 1) & 3; }

// four consecutive channels of a pixel (f32, relu applied) -> hi (4 halves), e4m3(hi / 4) (4 bytes), e4m3((v - hi) * 2^11 / 4)
struct Split4 { uint2 hi; uint32_t qh, ql; };
__device__ __forceinline__ Split4 split4(float v0, float v1, float v2, float v3)
{
    const h2 p01 = __builtin_convertvector(f32x2{v0, v1}, h2), p23 = __builtin_convertvector(f32x2{v2, v3}, h2);
    const uint32_t u01 = __builtin_bit_cast(uint32_t, p01), u23 = __builtin_bit_cast(uint32_t, p23);
    float r0, r1, r2, r3;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(u01), "v"(v0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(u01), "v"(v1));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(u23), "v"(v2));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(u23), "v"(v3));
    constexpr float SH = (float)(1 << Q8_SA), SL = (float)(1 << Q8_SA) / 2048.f;
    s16x2 qh = {0, 0}, ql = {0, 0};
    qh = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(qh, p01, SH, false);
    qh = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(qh, p23, SH, true);
    ql = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(ql, r0, r1, SL, false);
    ql = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(ql, r2, r3, SL, true);
    Split4 s;
    s.hi = make_uint2(u01, u23);
    s.qh = __builtin_bit_cast(uint32_t, qh);
    s.ql = __builtin_bit_cast(uint32_t, ql);
    return s;
}

// E8M0 scale bytes of the four 32-k blocks of the scaled MFMA: lane group g supplies block g's.  Activations: blocks 0 / 2
// are hi (2^SA), 1 / 3 lo (2^(SA - 11)); weights: blocks 0 / 2 are w_lo (2^(SW - 11)), 1 / 3 w_hi (2^SW).
__device__ __forceinline__ int scale_act(int kq) { return (kq & 1) ? 127 + Q8_SA - 11 : 127 + Q8_SA; }
__device__ __forceinline__ int scale_wgt(int kq) { return (kq & 1) ? 127 + Q8_SW : 127 + Q8_SW - 11; }

// straight-line expansion of a loop body over 0 .. N - 1 with the index a compile-time constant (a `#pragma unroll` nest of a
// few hundred steps runs into the unroller's budget, silently, and the register arrays indexed by it go to scratch)
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

__device__ __forceinline__ i32x8 pair8(const uint4& a, const uint4& b)
{
    return i32x8{(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)b.x, (int)b.y, (int)b.z, (int)b.w};
}

// ------------------------------------------------------------------------------------------------------------------
// conv1 + conv2 of a third of a patch: T = 12 (thirds 0, 1) or 8 (third 2) output rows x 32 columns of conv2.
//   grid  : 3 * npatch workgroups of 256 threads; item = (patch, third)
//   goban : [frame][380][380][3] u8 (the K10 patch gather is fused: region_origin)
//   w1    : conv1 weights x 2^8 as fp16 A fragments [plane hi / lo][channel tile 2][k-step 4][lane][8], k as in
//           conv12_bf16_kernel (fragment f = 4 s + kq = kernel row f / 3, tap pair f % 3; e = (tap, channel)); the u8 pixels
//           are exact halves: two MFMAs per product
//   w2m   : conv2 main term: the c2w_h2 pack [channel tile 2][tap 25][plane][lane][8] (plane 0 = fp16(w x 2^8))
//   w2x   : conv2 cross terms [channel tile 2][kernel row 5 x column pair 3, then the 3 vertical pairs of column 4][lane][32 bytes]: e4m3, bytes 0..15 the first
//           column of the pair, 16..31 the second (zero for the pair (4, -)); lane group 0 / 1: w_lo of input channels
//           0..15 / 16..31, group 2 / 3: w_hi of the same
//   out   : [patch][16 * 16 pooled pixels][32] f32 (as conv_mfma16_h2_kernel writes it)
// LDS: pixels as halves [20 rows][42 px][B, G, R, 0]; conv1's output [16 rows][36 px] as a hi plane (32 halves per pixel)
// and a q plane (64 bytes per pixel), both swizzled like the bf16 tile.  80 448 B: two workgroups per CU.
constexpr int C12_PIX_RS = 42 * 4, C12_PIX_ROWS = 20;          // halves
constexpr int C12_ROWS = 16, C12_RS = 36 * 32;                 // halves per row of the hi plane = half the bytes of a q-plane row

template <int T>
__device__ __forceinline__ void conv12_q8_body(
    uint16_t* __restrict__ thi, uint8_t* __restrict__ tq, uint16_t* __restrict__ pix, const int patch, const int r0,
    const uint8_t* __restrict__ goban, const uint16_t* __restrict__ w1, const float* __restrict__ b1,
    const uint16_t* __restrict__ w2m, const uint8_t* __restrict__ w2x, const float* __restrict__ b2,
    float* __restrict__ out, int* __restrict__ overflow)
{
#pragma clang fp contract(off)
    constexpr int ROWS = T + 4, PROWS = T + 8, NTILE = ROWS * 36 / 16;
    static_assert(ROWS * 36 % 16 == 0 && ROWS <= C12_ROWS && PROWS <= C12_PIX_ROWS, "tile geometry");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int frame = patch / 100, reg = patch % 100;
    const int px0 = region_origin(reg % 10);
    Q8_STAMP_BEGIN;
    if (Q8_PRIO) __builtin_amdgcn_s_setprio(3);
    // the PROWS x 40 pixels of the item: a thread fetches 12 bytes (4 pixels) and turns them into 32 bytes of halves (byte b
    // becomes the half 0x6400 | b = 1024 + b by one v_perm_b32 per two values, minus 1024 by one packed subtraction)
    if (tid < PROWS * 10) {
        const int r = tid / 10, g = tid % 10;
        const uint32_t* sp = reinterpret_cast<const uint32_t*>(goban + ((size_t)frame * 380 + region_origin(reg / 10) + r0 + r) * 1140 + (size_t)px0 * 3) + 3 * g;
        const uint32_t d0 = sp[0], d1 = sp[1], d2 = sp[2];
        const uint32_t p1 = __builtin_amdgcn_alignbit(d1, d0, 24), p2 = __builtin_amdgcn_alignbit(d2, d1, 16);
        const h2 k1024 = {(_Float16)1024.f, (_Float16)1024.f};
        auto bg = [&](uint32_t p, uint32_t sel) {
            return __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, __builtin_amdgcn_perm(p, 0x64646464u, sel)) - k1024);
        };
        uint4 lo, hi;
        lo.x = bg(d0, 0x00050004u); lo.y = bg(d0, 0x000C0006u);
        lo.z = bg(p1, 0x00050004u); lo.w = bg(p1, 0x000C0006u);
        hi.x = bg(p2, 0x00050004u); hi.y = bg(p2, 0x000C0006u);
        hi.z = bg(d2, 0x00060005u); hi.w = bg(d2, 0x000C0007u);
        uint4* d = reinterpret_cast<uint4*>(&pix[r * C12_PIX_RS + 16 * g]);
        d[0] = lo; d[1] = hi;
    } else if (tid >= 256 - 2 * 16) {     // columns 40 and 41 (under the zero weights of the sixth tap): finite values
        for (int r = tid - (256 - 32); r < PROWS; r += 32) *reinterpret_cast<uint4*>(&pix[r * C12_PIX_RS + 160]) = make_uint4(0, 0, 0, 0);
    }

    // conv2's first sweep of weight fragments is requested here, a whole conv1 ahead of its use (at the k-loop's start it cost an
    // exposed L2 round trip per workgroup)
    const int n = wave & 1, s = wave >> 1;
    const uint4* wm = reinterpret_cast<const uint4*>(w2m) + (size_t)n * 25 * 2 * 64 + lane;
    uint4 wb[2][5][2];
#pragma unroll
    for (int i = 0; i < 5; i++) wb[0][i][0] = wm[(size_t)((i * 5 + 0) * 2) * 64];
    // conv1's weights (A operand), both planes, and its bias x 2^8 while the pixels land
    h8 wa[2][4][2];
#pragma unroll
    for (int pl = 0; pl < 2; pl++)
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int n = 0; n < 2; n++) wa[pl][s][n] = __builtin_bit_cast(h8, reinterpret_cast<const uint4*>(w1)[((pl * 2 + n) * 4 + s) * 64 + lane]);
    float4 bv1[2];
#pragma unroll
    for (int n = 0; n < 2; n++) {
        bv1[n] = *reinterpret_cast<const float4*>(b1 + n * 16 + 4 * kq);
        bv1[n].x *= Q8_WSCALE; bv1[n].y *= Q8_WSCALE; bv1[n].z *= Q8_WSCALE; bv1[n].w *= Q8_WSCALE;
    }
    int foff[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int f = 4 * s + kq > 14 ? 14 : 4 * s + kq;
        foff[s] = (f / 3) * C12_PIX_RS + 8 * (f % 3);
    }
    __syncthreads();
    Q8_STAMP(0);                                           // pixels staged, conv1's weights here

    // ---- conv1: ROWS x 36 pixels = NTILE tiles of 16 raster pixels.  D = W x P: a lane ends up with four consecutive
    // channels of one pixel per channel tile -> relu, x 2^-8, split, one 8-byte store (hi) and two 4-byte stores (q).
    float big = 0.f;
    {
        auto frags = [&](int t, uint4 (&pf)[4]) {
            const int m = 16 * t + l15, my = m / 36, mx = m - 36 * my;
            const uint16_t* pp = &pix[my * C12_PIX_RS + 4 * mx];
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const uint2 f0 = *reinterpret_cast<const uint2*>(pp + foff[s]), f1 = *reinterpret_cast<const uint2*>(pp + foff[s] + 4);
                pf[s] = make_uint4(f0.x, f0.y, f1.x, f1.y);
            }
        };
        // a tile in two halves: its 16 MFMAs, and -- one tile later, next to the NEXT tile's MFMAs, which do not depend on them --
        // the relu, split and stores of its results (in one piece every tile was a chain: LDS -> 16 MFMAs -> 50 vector
        // instructions -> stores, the matrix pipe idle under the second half)
        auto tile_mfma = [&](const uint4 (&pf)[4], f32x4 (&c1)[2]) {
#pragma unroll
            for (int n = 0; n < 2; n++) { c1[n][0] = bv1[n].x; c1[n][1] = bv1[n].y; c1[n][2] = bv1[n].z; c1[n][3] = bv1[n].w; }
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int pl = 0; pl < 2; pl++)
#pragma unroll
                    for (int n = 0; n < 2; n++) c1[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[pl][s][n], __builtin_bit_cast(h8, pf[s]), c1[n], 0, 0, 0);
        };
        auto tile_out = [&](int t, const f32x4 (&c1)[2]) {
            const int m = 16 * t + l15, my = m / 36, mx = m - 36 * my;
            const int sw = swz32(mx);
            uint16_t* hp = &thi[my * C12_RS + 32 * mx + 4 * (kq & 1)];
            uint8_t* qp = &tq[my * (2 * C12_RS) + 64 * mx + 4 * kq];
#pragma unroll
            for (int n = 0; n < 2; n++) {
                const float v0 = fmaxf(c1[n][0], 0.f) * (1.f / Q8_WSCALE), v1 = fmaxf(c1[n][1], 0.f) * (1.f / Q8_WSCALE);
                const float v2 = fmaxf(c1[n][2], 0.f) * (1.f / Q8_WSCALE), v3 = fmaxf(c1[n][3], 0.f) * (1.f / Q8_WSCALE);
                big = fmaxf(big, fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)));
                const Split4 sp = split4(v0, v1, v2, v3);
                *reinterpret_cast<uint2*>(hp + (((2 * n + (kq >> 1)) ^ sw) << 3)) = sp.hi;
                *reinterpret_cast<uint32_t*>(qp + ((n ^ sw) << 4)) = sp.qh;             // hi bytes of channels 16 n + 4 kq ..: chunk n
                *reinterpret_cast<uint32_t*>(qp + (((2 + n) ^ sw) << 4)) = sp.ql;       // lo bytes: chunk 2 + n
            }
        };
        constexpr int NI = (NTILE + 3) / 4;
        uint4 pf[2][4];
        f32x4 c1[2][2];
        frags(wave, pf[0]);
#pragma unroll
        for (int i = 0; i <= NI; i++) {
            const int t = wave + 4 * i;
            if (i + 1 < NI && t + 4 < NTILE) frags(t + 4, pf[(i + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (i < NI && t < NTILE) tile_mfma(pf[i & 1], c1[i & 1]);
            if (i > 0 && t - 4 < NTILE) tile_out(t - 4, c1[(i - 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (overflow && !(big <= Q8_LIMIT)) *overflow = 1;
    Q8_STAMP(1);                                           // wave 0's conv1 tiles
    __syncthreads();
    Q8_STAMP(2);                                           // ... the other waves'
    if (Q8_PRIO) __builtin_amdgcn_s_setprio(0);

    // ---- conv2: wave = (channel tile n, 16-column strip s), T output rows.  Eight sweeps over the strip's ROWS input rows:
    // M0 M1 X01 M2 M3 X23 M4 X4 (M j: main term of tap column j; X: cross terms of a pair of columns).  The five weight
    // fragments of a sweep (taps (0..4, j)) are in registers, the next sweep's in flight.
    f32x4 acc[T];
    {
        const float bb = b2[16 * n + l15] * Q8_WSCALE;
#pragma unroll
        for (int r = 0; r < T; r++) { acc[r][0] = bb; acc[r][1] = bb; acc[r][2] = bb; acc[r][3] = bb; }
    }
    const int sa = scale_act(kq), sb = scale_wgt(kq);
    // sweep k: kind (0 main, 1 cross terms of two columns, 2 cross terms of the last column with its taps paired VERTICALLY: the
    // fragments of rows y and y + 1 against taps (0, 1), (2, 3), (4, -) -- three instructions per row instead of five), first
    // column, pair index
    constexpr int NSW = 8;
    constexpr int SW_KIND[NSW] = {0, 0, 1, 0, 0, 1, 0, 2};
    constexpr int SW_COL[NSW] = {0, 1, 0, 2, 3, 2, 4, 4};
    constexpr int SW_PAIR[NSW] = {0, 0, 0, 0, 0, 1, 0, 2};
    const uint4* wx = reinterpret_cast<const uint4*>(w2x) + ((size_t)n * 18 * 64 + lane) * 2;
    auto wload = [&](int k, uint4 (&dst)[5][2]) {
#pragma unroll
        for (int i = 0; i < 5; i++) {
            if (SW_KIND[k] == 0) dst[i][0] = wm[(size_t)((i * 5 + SW_COL[k]) * 2) * 64];
            else if (SW_KIND[k] == 1) { dst[i][0] = wx[(size_t)((i * 3 + SW_PAIR[k]) * 64) * 2]; dst[i][1] = wx[(size_t)((i * 3 + SW_PAIR[k]) * 64) * 2 + 1]; }
            else if (i < 3) { dst[i][0] = wx[(size_t)((15 + i) * 64) * 2]; dst[i][1] = wx[(size_t)((15 + i) * 64) * 2 + 1]; }
        }
    };
    static_assert(SW_KIND[0] == 0 && SW_COL[0] == 0, "the first sweep's weights are loaded before conv1");
    // byte address of the lane's 16-byte chunk in a plane row (both planes: 64 bytes per pixel, chunk kq ^ swz32(x)) per column
    int ab[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        int x = 16 * s + l15 + j;
        x = x > 35 ? 35 : x;                                // (column 5 exists only as the empty half of the pair (4, -))
        ab[j] = 64 * x + ((kq ^ swz32(x)) << 4);
    }
    constexpr int D = Q8_C2_D, NT = NSW * ROWS;
    const uint8_t* th8 = reinterpret_cast<const uint8_t*>(thi);
    uint4 ar[D][2];
    auto a_read = [&](int t, uint4 (&dst)[2]) {
        const int k = t / ROWS, y = t % ROWS;
        if (SW_KIND[k] == 0) dst[0] = *reinterpret_cast<const uint4*>(th8 + y * (2 * C12_RS) + ab[SW_COL[k]]);
        else if (SW_KIND[k] == 1) {
            dst[0] = *reinterpret_cast<const uint4*>(tq + y * (2 * C12_RS) + ab[SW_COL[k]]);
            dst[1] = *reinterpret_cast<const uint4*>(tq + y * (2 * C12_RS) + ab[SW_COL[k] + 1]);
        } else {
            const int y1 = y + 1 < ROWS ? y + 1 : y;          // (below the last row: the empty half of (4, -), any finite bytes)
            dst[0] = *reinterpret_cast<const uint4*>(tq + y * (2 * C12_RS) + ab[SW_COL[k]]);
            dst[1] = *reinterpret_cast<const uint4*>(tq + y1 * (2 * C12_RS) + ab[SW_COL[k]]);
        }
    };
#pragma unroll
    for (int t = 0; t < D - 1; t++) a_read(t, ar[t]);
#pragma unroll
    for (int k = 0; k < NSW; k++) {
#pragma unroll
        for (int y = 0; y < ROWS; y++) {
            const int t = k * ROWS + y;
            if (t + D - 1 < NT) a_read(t + D - 1, ar[(t + D - 1) % D]);
            if (y == 0 && k + 1 < NSW) wload(k + 1, wb[(k + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const int r = SW_KIND[k] == 2 ? y - 2 * i : y - i;      // (kind 2: instruction i covers the taps 2 i and 2 i + 1)
                if (r >= 0 && r < T && !(SW_KIND[k] == 2 && i > 2)) {
                    if (SW_KIND[k] == 0)
                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, ar[t % D][0]), __builtin_bit_cast(h8, wb[k & 1][i][0]), acc[r], 0, 0, 0);
                    else
                        acc[r] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(pair8(ar[t % D][0], ar[t % D][1]), pair8(wb[k & 1][i][0], wb[k & 1][i][1]),
                                                                                  acc[r], 0, 0, 0, sa, 0, sb);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    Q8_STAMP(3);                                           // wave 0's k-loop
    if (Q8_PRIO) __builtin_amdgcn_s_setprio(3);

    // ---- 2x2 max-pool in the lane (bias in the sums already; max commutes with the relu), x 2^-8, f32 out
    float* o = out + (size_t)patch * 256 * 32 + 16 * n + l15;
    float big2 = 0.f;
#pragma unroll
    for (int r2 = 0; r2 < T / 2; r2++)
#pragma unroll
        for (int p = 0; p < 2; p++) {
            float v = fmaxf(fmaxf(acc[2 * r2][2 * p], acc[2 * r2][2 * p + 1]), fmaxf(acc[2 * r2 + 1][2 * p], acc[2 * r2 + 1][2 * p + 1]));
            v = fmaxf(v, 0.f) * (1.f / Q8_WSCALE);
            if (!(v <= Q8_LIMIT)) big2 = 1e30f;            // (also true for NaN: an e4m3 operand out of range)
            const int py = r0 / 2 + r2, px = 8 * s + 2 * kq + p;
            o[(py * 16 + px) * 32] = v;
        }
    if (overflow && big2 != 0.f) *overflow = 1;
    Q8_STAMP(4);                                           // epilogue issued
#if Q8_DBG_TIME
    if (threadIdx.x == 0) atomicAdd(&g_q8_prof[7], 1ull);
    q8_log(0, t_born__);
#endif
}

__global__ __launch_bounds__(256, 2) void conv12_q8_kernel(
    const uint8_t* __restrict__ goban, const uint16_t* __restrict__ w1, const float* __restrict__ b1,
    const uint16_t* __restrict__ w2m, const uint8_t* __restrict__ w2x, const float* __restrict__ b2,
    float* __restrict__ out, int* __restrict__ overflow)
{
    __shared__ __attribute__((aligned(16))) uint16_t thi[C12_ROWS * C12_RS];
    __shared__ __attribute__((aligned(16))) uint8_t tq[C12_ROWS * 2 * C12_RS];
    __shared__ __attribute__((aligned(16))) uint16_t pix[C12_PIX_ROWS * C12_PIX_RS];
    const int patch = blockIdx.x / 3, third = blockIdx.x % 3;
    if (third < 2) conv12_q8_body<12>(thi, tq, pix, patch, 12 * third, goban, w1, b1, w2m, w2x, b2, out, overflow);
    else conv12_q8_body<8>(thi, tq, pix, patch, 24, goban, w1, b1, w2m, w2x, b2, out, overflow);
}

// ------------------------------------------------------------------------------------------------------------------
// conv3 + conv4 of one patch per workgroup of four waves.
//   in   : [patch][16 * 16][32] f32 (conv12's pooled output)      out : [patch][6 * 6][90] f32
//   w3m  : the c3w_h2 pack [channel tile 6][tap 9][plane][lane][8]          w3x : e4m3 [channel tile 6][kernel row 3 x pair 2, then the 2 vertical pairs of column 2][lane][32 B]
//   w4m  : the c4w_h2 pack [channel tile 6][step 27 = tap * 3 + cc][plane][lane][8]      w4x : e4m3 [channel tile 6][pair 14 of steps][lane][32 B]
// conv3 (D = W x P on row tiles as in conv34_bf16_kernel, in twelve units (channel tile, seven output rows), three per wave):
// sweeps M0 M1 X01 M2 X2 over the nine input rows of a unit; a lane ends up with four consecutive channels of a pixel -> split,
// into conv4's tile.  conv4 (wave = (five / four pooling tiles, three channel tiles), D = P x W, pool in the lane): per pair of
// k-steps two main sweeps and one cross sweep over the wave's tiles.
// LDS (bytes): conv4's tile [14][14][96 channels] in two planes of 192 bytes per pixel (hi at 0, q at 37 632), chunk (cc, kq) of
// the pixel in row y at slot 4 cc + (kq ^ 2 (y & 1)).  conv3's input (a hi plane [256 px][64] and a q plane [256 px][64],
// swizzled by swz32) lies where rows 0 .. 6 of the two planes will be (16 384 < 7 x 2 688): the first pass's rows 7 .. 13 are
// stored at once, rows 0 .. 6 after a barrier.  75 264 B: two workgroups per CU.
constexpr int C34_PS = 192, C34_RS = 14 * 192, C34_PLANE = 14 * C34_RS;      // bytes

// conv4's sweeps: k = 3 u + {0, 1, 2} = M(2u), M(2u + 1), X(u); the last pair has one step: M(26), X(13)
constexpr int C4_NSW = 27 + 14;
constexpr int c4_kind(int k) { return k % 3 == 2 || k == C4_NSW - 1 ? 1 : 0; }
constexpr int c4_step(int k) { return 2 * (k / 3) + (k % 3 == 1 ? 1 : 0); }

__device__ __forceinline__ int c34_row_base(int y, int plane) { return plane * C34_PLANE + y * C34_RS; }

// Four waves per workgroup, one per SIMD (256 registers each): conv3 in twelve units (channel tile, pass), three per wave --
// (w, rows 7-13), one of the four units of tiles 4 and 5, (w, rows 0-6) --, conv4 wave = (five / four pooling tiles, three channel
// tiles).  (The first form had six waves -- conv3 wave = channel tile, conv4 wave = (row of three tiles, three channel tiles) --
// and two SIMDs of a CU carried two of them between the barriers: 6.66 against 6.1 us per frame,
// tools/variants/conv34_q8_six_waves.hip.txt.)
__device__ __forceinline__ void conv34_q8_body(
    uint8_t* __restrict__ lds, const float* __restrict__ in, const uint16_t* __restrict__ w3m, const uint8_t* __restrict__ w3x, const float* __restrict__ b3,
    const uint16_t* __restrict__ w4m, const uint8_t* __restrict__ w4x, const float* __restrict__ b4,
    float* __restrict__ out, int* __restrict__ overflow)
{
#pragma clang fp contract(off)
    constexpr int NW = 4, NTHR = 64 * NW;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, kq = lane >> 4;
    const int patch = blockIdx.x;
    Q8_STAMP_BEGIN;
    const int s_act = scale_act(kq), s_wgt = scale_wgt(kq);
    if (Q8_PRIO) __builtin_amdgcn_s_setprio(3);

    {   // stage: four consecutive channels of a pixel per step -> hi plane at 0, q plane at C34_PLANE.  All loads of a thread
        // first (left as a loop the compiler keeps one in flight: 3.9 us per workgroup, measured)
        constexpr int NLD = (2048 + NTHR - 1) / NTHR;
        const float4* g = reinterpret_cast<const float4*>(in + (size_t)patch * 256 * 32);
        float4 v[NLD];
#pragma unroll
        for (int k = 0; k < NLD; k++) v[k] = tid + NTHR * k < 2048 ? g[tid + NTHR * k] : make_float4(0.f, 0.f, 0.f, 0.f);
        __builtin_amdgcn_sched_barrier(0);
        float big = 0.f;
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int c = tid + NTHR * k;
            big = fmaxf(big, fmaxf(fmaxf(v[k].x, v[k].y), fmaxf(v[k].z, v[k].w)));
            const Split4 sp = split4(v[k].x, v[k].y, v[k].z, v[k].w);
            const int px = c >> 3, c4 = c & 7, sw = swz32(px & 15);
            if (c < 2048) {
                *reinterpret_cast<uint2*>(&lds[64 * px + (((c4 >> 1) ^ sw) << 4) + 8 * (c4 & 1)]) = sp.hi;
                *reinterpret_cast<uint32_t*>(&lds[C34_PLANE + 64 * px + (((c4 >> 2) ^ sw) << 4) + 4 * (c4 & 3)]) = sp.qh;
                *reinterpret_cast<uint32_t*>(&lds[C34_PLANE + 64 * px + (((2 + (c4 >> 2)) ^ sw) << 4) + 4 * (c4 & 3)]) = sp.ql;
            }
        }
        if (overflow && !(big <= Q8_LIMIT)) *overflow = 1;
    }
    __syncthreads();
    Q8_STAMP(8);                                           // input staged
    if (Q8_PRIO) __builtin_amdgcn_s_setprio(0);

    // ---- conv3
    auto store_px = [&](int n, int oy, const Split4& sp) {
        const int c8 = 2 * n + (kq >> 1);                   // 16-byte chunk of the hi plane: 0 .. 11
        const int cq = 4 * (n >> 1) + (n & 1);              // chunk of the q plane holding the hi bytes (lo bytes: + 2)
        const int f = (oy & 1) << 1;
        if (l15 < 14) {
            uint8_t* p = &lds[c34_row_base(oy, 0) + l15 * C34_PS];
            *reinterpret_cast<uint2*>(p + (((c8 & ~3) | ((c8 & 3) ^ f)) << 4) + 8 * (kq & 1)) = sp.hi;
            uint8_t* q = &lds[c34_row_base(oy, 1) + l15 * C34_PS + 4 * kq];
            *reinterpret_cast<uint32_t*>(q + (((cq & ~3) | ((cq & 3) ^ f)) << 4)) = sp.qh;
            *reinterpret_cast<uint32_t*>(q + ((((cq + 2) & ~3) | (((cq + 2) & 3) ^ f)) << 4)) = sp.ql;
        }
    };
    float big3 = 0.f;
    int ab[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int x = l15 + (j < 3 ? j : 2);                // (column 3 exists only as the empty half of the pair (2, -): any finite bytes)
        ab[j] = 64 * x + ((kq ^ swz32(x)) << 4);
    }
    // The wave's units (channel tile n, output rows 7 pass .. 7 pass + 6), rows 7 .. 13 first: they lie behind conv3's input in
    // both planes and are stored at once; rows 0 .. 6 wait, split, for the barrier.  ONE sequence of sweeps over all units: the
    // weight fragments run WPF sweeps ahead and the fragment reads D - 1 steps ahead ACROSS the unit boundaries (unit by
    // unit, every unit began by waiting an L2 round trip for its first weights with nothing left to overlap it).
    constexpr int NU = 3;
    const int un[NU] = {wave, 4 + (wave >> 1), wave};      // the middle unit: tile 4 or 5, either pass
    const int up[NU] = {1, 1 - (wave & 1), 0};
    Split4 c3[NU][7];
    {
        constexpr int NSW = 5, ROWS = 9, UT = NSW * ROWS, NT = NU * UT, D = Q8_C3_D;
        constexpr int SW_KIND[NSW] = {0, 0, 1, 0, 2};       // (2: the last column's cross terms with the taps paired vertically, as in conv2)
        constexpr int SW_COL[NSW] = {0, 1, 0, 2, 2};
        constexpr int SW_PAIR[NSW] = {0, 0, 0, 0, 1};
        constexpr int WPF = Q8_C3_PF;
        uint4 wb[WPF + 1][3][2];
        auto wload = [&](int K, uint4 (&dst)[3][2]) {       // K: sweep of the whole sequence
            const int u = K / NSW, k = K % NSW;
            const uint4* wm = reinterpret_cast<const uint4*>(w3m) + (size_t)un[u] * 9 * 2 * 64 + lane;
            const uint4* wx = reinterpret_cast<const uint4*>(w3x) + ((size_t)un[u] * 8 * 64 + lane) * 2;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                if (SW_KIND[k] == 0) dst[i][0] = wm[(size_t)((i * 3 + SW_COL[k]) * 2) * 64];
                else if (SW_KIND[k] == 1) { dst[i][0] = wx[(size_t)((i * 2 + SW_PAIR[k]) * 64) * 2]; dst[i][1] = wx[(size_t)((i * 2 + SW_PAIR[k]) * 64) * 2 + 1]; }
                else if (i < 2) { dst[i][0] = wx[(size_t)((6 + i) * 64) * 2]; dst[i][1] = wx[(size_t)((6 + i) * 64) * 2 + 1]; }
            }
        };
        uint4 ar[D][2];
        auto p_read = [&](int t, uint4 (&dst)[2]) {
            const int u = t / UT, k = (t % UT) / ROWS, y = t % ROWS;
            const int rb = 7 * 1024 * up[u];                // byte offset of the unit's first input row in either plane
            if (SW_KIND[k] == 0) dst[0] = *reinterpret_cast<const uint4*>(&lds[rb + 1024 * y + ab[SW_COL[k]]]);
            else if (SW_KIND[k] == 1) {
                dst[0] = *reinterpret_cast<const uint4*>(&lds[C34_PLANE + rb + 1024 * y + ab[SW_COL[k]]]);
                dst[1] = *reinterpret_cast<const uint4*>(&lds[C34_PLANE + rb + 1024 * y + ab[SW_COL[k] + 1]]);
            } else {
                const int y1 = y + 1 < ROWS ? y + 1 : y;
                dst[0] = *reinterpret_cast<const uint4*>(&lds[C34_PLANE + rb + 1024 * y + ab[SW_COL[k]]]);
                dst[1] = *reinterpret_cast<const uint4*>(&lds[C34_PLANE + rb + 1024 * y1 + ab[SW_COL[k]]]);
            }
        };
#pragma unroll
        for (int K = 0; K < WPF; K++) wload(K, wb[K]);
#pragma unroll
        for (int t = 0; t < D - 1; t++) p_read(t, ar[t]);
        f32x4 acc[7];
#pragma unroll
        for (int u = 0; u < NU; u++) {
            {
                const int c0 = 16 * un[u] + 4 * kq;         // 90 real channels: 88 .. 91 straddles the end
                const float bx = c0 < 90 ? b3[c0] * Q8_WSCALE : 0.f, by = c0 + 1 < 90 ? b3[c0 + 1] * Q8_WSCALE : 0.f;
                const float bz = c0 + 2 < 90 ? b3[c0 + 2] * Q8_WSCALE : 0.f, bw = c0 + 3 < 90 ? b3[c0 + 3] * Q8_WSCALE : 0.f;
#pragma unroll
                for (int r = 0; r < 7; r++) { acc[r][0] = bx; acc[r][1] = by; acc[r][2] = bz; acc[r][3] = bw; }
            }
#pragma unroll
            for (int k = 0; k < NSW; k++) {
#pragma unroll
                for (int y = 0; y < ROWS; y++) {
                    const int K = u * NSW + k, t = K * ROWS + y;
                    if (t + D - 1 < NT) p_read(t + D - 1, ar[(t + D - 1) % D]);
                    if (y == 0 && K + WPF < NU * NSW) wload(K + WPF, wb[(K + WPF) % (WPF + 1)]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        const int r = SW_KIND[k] == 2 ? y - 2 * i : y - i;
                        if (r >= 0 && r < 7 && !(SW_KIND[k] == 2 && i > 1)) {
                            if (SW_KIND[k] == 0)
                                acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, wb[K % (WPF + 1)][i][0]), __builtin_bit_cast(h8, ar[t % D][0]), acc[r], 0, 0, 0);
                            else
                                acc[r] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(pair8(wb[K % (WPF + 1)][i][0], wb[K % (WPF + 1)][i][1]), pair8(ar[t % D][0], ar[t % D][1]),
                                                                                          acc[r], 0, 0, 0, s_wgt, 0, s_act);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int r = 0; r < 7; r++) {
                const float v0 = fmaxf(acc[r][0], 0.f) * (1.f / Q8_WSCALE), v1 = fmaxf(acc[r][1], 0.f) * (1.f / Q8_WSCALE);
                const float v2 = fmaxf(acc[r][2], 0.f) * (1.f / Q8_WSCALE), v3 = fmaxf(acc[r][3], 0.f) * (1.f / Q8_WSCALE);
                if (l15 < 14 && !(fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)) <= Q8_LIMIT)) big3 = 1e30f;
                c3[u][r] = split4(v0, v1, v2, v3);
            }
            if (u + 1 < NU) {                               // (the last unit is a pass 0 by construction)
                if (up[u]) {
#pragma unroll
                    for (int r = 0; r < 7; r++) store_px(un[u], 7 + r, c3[u][r]);
                }
            }
        }
    }
    // ---- conv4: wave = (group of MT pooling tiles from T0 on, channel tiles 3 nh .. 3 nh + 2): a fragment read feeds three
    // MFMAs (with two channel tiles per wave the LDS, at 1 650 reads of 1 KB per patch, took as long as the matrix pipe)
    constexpr int MT = 5, NN = 3;
    // (the group of five tiles goes to waves 0, 1 of even patches and to waves 2, 3 of odd ones, so that the two workgroups of a CU
    // do not both put their longer waves on the same SIMDs)
    const int nh = wave & 1, T0 = MT * ((wave >> 1) ^ (patch & 1));
    const bool last_tile = T0 + MT - 1 < 9;                 // (tile 9 of the second group does not exist; uniform over the wave)
    f32x4 acc[MT][NN];
#pragma unroll
    for (int n = 0; n < NN; n++) {
        const int co = 16 * (NN * nh + n) + l15;
        const float bb = co < 90 ? b4[co] * Q8_WSCALE : 0.f;
#pragma unroll
        for (int t = 0; t < MT; t++) { acc[t][n][0] = bb; acc[t][n][1] = bb; acc[t][n][2] = bb; acc[t][n][3] = bb; }
    }
    const int q = l15 >> 2, sub = l15 & 3, dy = 2 * (q >> 1) + (sub >> 1), dx = 2 * (q & 1) + (sub & 1);
    // the lane's pixel inside pooling tile T0 + tl, for taps in rows of either parity (the parity flips bit 1 of the slot);
    // everything else of a fragment's address (tap, channel block, plane) is an immediate offset of the read
    int ta[MT][2];
#pragma unroll
    for (int tl = 0; tl < MT; tl++) {
        const int t = T0 + tl, tt = t > 8 ? 8 : t;          // (tile 9: clamped, never computed)
        const int base = (4 * (tt / 3) + dy) * C34_RS + (4 * (tt % 3) + dx) * C34_PS;
        ta[tl][0] = base + ((kq ^ ((dy & 1) << 1)) << 4);
        ta[tl][1] = base + ((kq ^ (((dy + 1) & 1) << 1)) << 4);
    }
    const uint4* wm4 = reinterpret_cast<const uint4*>(w4m) + (size_t)(NN * nh) * 27 * 2 * 64 + lane;
    const uint4* wx4 = reinterpret_cast<const uint4*>(w4x) + ((size_t)(NN * nh) * 14 * 64 + lane) * 2;
    // sweeps: per pair u of k-steps: M(2u), M(2u + 1), X(u); the last pair has one step
    constexpr int PF = Q8_C4_PF, D4 = Q8_C4_D, NT4 = C4_NSW * MT;
    uint4 wb4[PF + 1][NN][2], ar4[D4][2];
    auto wload4 = [&](auto kc, uint4 (&dst)[NN][2]) {
        constexpr int k = decltype(kc)::value;
#pragma unroll
        for (int n = 0; n < NN; n++) {
            if constexpr (c4_kind(k) == 0) dst[n][0] = wm4[((size_t)n * 27 + c4_step(k)) * 2 * 64];
            else { dst[n][0] = wx4[((size_t)n * 14 + k / 3) * 64 * 2]; dst[n][1] = wx4[((size_t)n * 14 + k / 3) * 64 * 2 + 1]; }
        }
    };
    auto unit_addr = [&](int step, int tl, int plane) {
        const int tap = step / 3, cc = step % 3, i = tap / 3, j = tap % 3;
        return ta[tl][i & 1] + (i * C34_RS + j * C34_PS + 64 * cc + plane * C34_PLANE);
    };
    auto a_read4 = [&](auto uc, uint4 (&dst)[2]) {
        constexpr int u = decltype(uc)::value, k = u / MT, tl = u % MT;
        if constexpr (c4_kind(k) == 0) dst[0] = *reinterpret_cast<const uint4*>(&lds[unit_addr(c4_step(k), tl, 0)]);
        else {
            constexpr int s0 = 2 * (k / 3), s1 = s0 + 1 < 27 ? s0 + 1 : s0;       // (the empty half of the last pair: any finite bytes)
            dst[0] = *reinterpret_cast<const uint4*>(&lds[unit_addr(s0, tl, 1)]);
            dst[1] = *reinterpret_cast<const uint4*>(&lds[unit_addr(s1, tl, 1)]);
        }
    };
    static_for<PF>([&](auto kc) { wload4(kc, wb4[decltype(kc)::value]); });     // (requested before the barriers of the relayout: an L2 round trip ahead)
    Q8_STAMP(9);                                           // wave 0's conv3
    if (Q8_PRIO) __builtin_amdgcn_s_setprio(3);
    __syncthreads();                                       // every wave is done with conv3's input, which rows 0 .. 6 overlay
#pragma unroll
    for (int u = 0; u < NU; u++)
        if (!up[u]) {
#pragma unroll
            for (int oy = 0; oy < 7; oy++) store_px(un[u], oy, c3[u][oy]);
        }
    if (overflow && big3 != 0.f) *overflow = 1;
    __syncthreads();
    Q8_STAMP(10);                                          // barrier, rows 0 .. 6 stored, barrier
    if (Q8_PRIO) __builtin_amdgcn_s_setprio(0);

    static_for<D4 - 1>([&](auto uc) { a_read4(uc, ar4[decltype(uc)::value]); });
    static_for<NT4>([&](auto uc) {
        constexpr int u = decltype(uc)::value, k = u / MT, tl = u % MT;
        if constexpr (u + D4 - 1 < NT4) a_read4(std::integral_constant<int, u + D4 - 1>{}, ar4[(u + D4 - 1) % D4]);
        if constexpr (tl == 0 && k + PF < C4_NSW) wload4(std::integral_constant<int, k + PF>{}, wb4[(k + PF) % (PF + 1)]);
        __builtin_amdgcn_sched_barrier(0);
        if (tl + 1 < MT || last_tile) {
#pragma unroll
            for (int n = 0; n < NN; n++) {
                if constexpr (c4_kind(k) == 0)
                    acc[tl][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, ar4[u % D4][0]), __builtin_bit_cast(h8, wb4[k % (PF + 1)][n][0]), acc[tl][n], 0, 0, 0);
                else
                    acc[tl][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(pair8(ar4[u % D4][0], ar4[u % D4][1]), pair8(wb4[k % (PF + 1)][n][0], wb4[k % (PF + 1)][n][1]),
                                                                                   acc[tl][n], 0, 0, 0, s_act, 0, s_wgt);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    Q8_STAMP(11);                                          // wave 0's conv4 loop
    if (Q8_PRIO) __builtin_amdgcn_s_setprio(3);
    float* o = out + (size_t)patch * 3240;
    float big4 = 0.f;
#pragma unroll
    for (int n = 0; n < NN; n++) {
        const int co = 16 * (NN * nh + n) + l15;
#pragma unroll
        for (int tl = 0; tl < MT; tl++) {
            const int t = T0 + tl;
            float v = fmaxf(fmaxf(acc[tl][n][0], acc[tl][n][1]), fmaxf(acc[tl][n][2], acc[tl][n][3]));
            v = fmaxf(v, 0.f) * (1.f / Q8_WSCALE);
            const int py = 2 * (t / 3) + (kq >> 1), px = 2 * (t % 3) + (kq & 1);
            if (t < 9 && co < 90) {
                if (!(v <= 65000.f)) big4 = 1e30f;
                o[(py * 6 + px) * 90 + co] = v;
            }
        }
    }
    if (overflow && big4 != 0.f) *overflow = 1;
    Q8_STAMP(12);
#if Q8_DBG_TIME
    if (threadIdx.x == 0) atomicAdd(&g_q8_prof[15], 1ull);
    q8_log(1, t_born__);
#endif
}

__global__ __launch_bounds__(256, 2) void conv34_q8_kernel(
    const float* __restrict__ in, const uint16_t* __restrict__ w3m, const uint8_t* __restrict__ w3x, const float* __restrict__ b3,
    const uint16_t* __restrict__ w4m, const uint8_t* __restrict__ w4x, const float* __restrict__ b4,
    float* __restrict__ out, int* __restrict__ overflow)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * C34_PLANE];
    conv34_q8_body(lds, in, w3m, w3x, b3, w4m, w4x, b4, out, overflow);
}

}  // namespace

// ---- host side: packs ------------------------------------------------------------------------------------------------
// OCP e4m3 (no infinities, 0x7F = NaN, largest 448), round to nearest even; the caller keeps |v| <= 448
static uint8_t e4m3_of(float v)
{
    if (v == 0.f || v != v) return 0;
    const uint8_t sgn = v < 0 ? 0x80 : 0;
    v = fabsf(v);
    if (v > 448.f) v = 448.f;
    int e;
    float m = frexpf(v, &e);
    e -= 1; m *= 2.f;                               // v = m 2^e, m in [1, 2)
    if (e < -6) return sgn | (uint8_t)lrintf(v * 512.f);        // subnormals: steps of 2^-9 (8 = the smallest normal's code)
    int q = (int)lrintf((m - 1.f) * 8.f);
    if (q == 8) { q = 0; e += 1; }
    if (e > 8 || (e == 8 && q > 6)) { e = 8; q = 6; }
    return sgn | (uint8_t)(((e + 7) << 3) | q);
}

// cross-term weights of a convolution: [channel tile][unit pair][lane][32 bytes].  A unit is 32 input channels of one tap;
// `pairs` lists the two units (tap, channel block) of every scaled MFMA, -1 for an empty half.  Lane = group * 16 + output
// channel; group 0 / 1: w_lo of input channels 0..15 / 16..31 of the unit, group 2 / 3: w_hi.  Weights x 2^8, flip applied.
struct Unit { int i, j, cc; };
static int pack_cross(ck_ctx* ctx, const float* k, int KH, int KW, int CIN, int COUT, int NTILE, const std::vector<std::pair<Unit, Unit>>& pairs,
                      DevBuf& dst, float* wmax)
{
    std::vector<uint8_t> v((size_t)NTILE * pairs.size() * 64 * 32, 0);
    const float SW = (float)(1 << Q8_SW);
    for (int nt = 0; nt < NTILE; nt++)
        for (size_t u = 0; u < pairs.size(); u++)
            for (int lane = 0; lane < 64; lane++)
                for (int half = 0; half < 2; half++) {
                    const Unit& un = half ? pairs[u].second : pairs[u].first;
                    if (un.i < 0) continue;
                    for (int e = 0; e < 16; e++) {
                        const int g = lane / 16, c = 32 * un.cc + 16 * (g & 1) + e, o = nt * 16 + lane % 16;
                        if (c >= CIN || o >= COUT) continue;
                        const float wv = k[(((size_t)(KH - 1 - un.i) * KW + (KW - 1 - un.j)) * CIN + c) * COUT + o] * Q8_WSCALE;
                        const _Float16 hi = (_Float16)wv;
                        const float lo = (float)(_Float16)(wv - (float)hi);
                        if (wmax && fabsf((float)hi) > *wmax) *wmax = fabsf((float)hi);
                        v[(((size_t)nt * pairs.size() + u) * 64 + lane) * 32 + 16 * half + e] = g < 2 ? e4m3_of(lo * 2048.f / SW) : e4m3_of((float)hi / SW);
                    }
                }
    CK_TRY(ck_ensure(ctx, dst, v.size()));
    CK_HIP(ctx, hipMemcpy(dst.p, v.data(), v.size(), hipMemcpyHostToDevice));
    return CK_OK;
}

// host[0], [2], [4], [6]: the Keras kernels [kh][kw][cin][cout] of conv1 .. conv4
int k_cnn_q8_pack(ck_ctx* ctx, const float* k1, const float* k2, const float* k3, const float* k4)
{
    CnnWeights& W = ctx->cnn;
    {   // conv1: both planes of w x 2^8 in conv12_bf16_kernel's fragment order
        std::vector<uint16_t> v((size_t)2 * 2 * 4 * 64 * 8, 0);
        for (int nt = 0; nt < 2; nt++)
            for (int s = 0; s < 4; s++)
                for (int lane = 0; lane < 64; lane++)
                    for (int e = 0; e < 8; e++) {
                        const int f = 4 * s + lane / 16, i = f / 3, j = 2 * (f % 3) + (e >> 2), c = e & 3, o = nt * 16 + lane % 16;
                        if (f > 14 || j > 4 || c > 2) continue;
                        const float wv = k1[(((size_t)(4 - i) * 5 + (4 - j)) * 3 + c) * 32 + o] * Q8_WSCALE;
                        const _Float16 hi = (_Float16)wv, lo = (_Float16)(wv - (float)hi);
                        memcpy(&v[((((size_t)0 * 2 + nt) * 4 + s) * 64 + lane) * 8 + e], &hi, 2);
                        memcpy(&v[((((size_t)1 * 2 + nt) * 4 + s) * 64 + lane) * 8 + e], &lo, 2);
                    }
        CK_TRY(ck_ensure(ctx, W.c1w_q8, v.size() * 2));
        CK_HIP(ctx, hipMemcpy(W.c1w_q8.p, v.data(), v.size() * 2, hipMemcpyHostToDevice));
    }
    float wmax = 0.f;
    {   // conv2: per kernel row the column pairs (0, 1), (2, 3), (4, -)
        std::vector<std::pair<Unit, Unit>> pr;
        for (int i = 0; i < 5; i++) {
            pr.push_back({Unit{i, 0, 0}, Unit{i, 1, 0}});
            pr.push_back({Unit{i, 2, 0}, Unit{i, 3, 0}});
            pr.push_back({Unit{i, 4, 0}, Unit{-1, 0, 0}});       // (not read by the kernel -- the last column's taps are paired vertically,
                                                                 //  below -- but the kernel's fragment index is 3 i + pair: the slot stays)
        }
        pr.push_back({Unit{0, 4, 0}, Unit{1, 4, 0}});
        pr.push_back({Unit{2, 4, 0}, Unit{3, 4, 0}});
        pr.push_back({Unit{4, 4, 0}, Unit{-1, 0, 0}});
        CK_TRY(pack_cross(ctx, k2, 5, 5, 32, 32, 2, pr, W.c2x_q8, &wmax));
    }
    {   // conv3: per kernel row the column pairs (0, 1), (2, -)
        std::vector<std::pair<Unit, Unit>> pr;
        for (int i = 0; i < 3; i++) {
            pr.push_back({Unit{i, 0, 0}, Unit{i, 1, 0}});
            pr.push_back({Unit{i, 2, 0}, Unit{-1, 0, 0}});       // (not read: the vertical pairs below; kept for the fragment index 2 i + pair)
        }
        pr.push_back({Unit{0, 2, 0}, Unit{1, 2, 0}});
        pr.push_back({Unit{2, 2, 0}, Unit{-1, 0, 0}});
        CK_TRY(pack_cross(ctx, k3, 3, 3, 32, 90, 6, pr, W.c3x_q8, &wmax));
    }
    {   // conv4: pairs of consecutive k-steps (step = tap * 3 + channel block), the 27th alone
        std::vector<std::pair<Unit, Unit>> pr;
        for (int u = 0; u < 14; u++) {
            const int s0 = 2 * u, s1 = 2 * u + 1;
            pr.push_back({Unit{s0 / 9, (s0 / 3) % 3, s0 % 3}, s1 < 27 ? Unit{s1 / 9, (s1 / 3) % 3, s1 % 3} : Unit{-1, 0, 0}});
        }
        CK_TRY(pack_cross(ctx, k4, 3, 3, 90, 90, 6, pr, W.c4x_q8, &wmax));
    }
    // weights beyond the e4m3 range of their block scale: the mode is not available with them (k_cnn_predict then runs the
    // three-MFMA kernels instead)
    W.q8_ok = wmax <= 448.f * (float)(1 << Q8_SW);
    return CK_OK;
}

#if Q8_DBG_TIME
// workgroups resident per CU over the kernel's duration, from the per-workgroup log
static void q8_residency(const char* name, int which, int nwg)
{
    const int n = std::min(nwg, (int)Q8_LOG_CAP);
    std::vector<unsigned long long> lg((size_t)3 * Q8_LOG_CAP);
    (void)hipMemcpyFromSymbol(lg.data(), HIP_SYMBOL(g_q8_log), lg.size() * 8, (size_t)which * lg.size() * 8);
    std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> ev;      // CU key -> (time, +1 / -1)
    unsigned long long lo = ~0ull, hi = 0;
    double life = 0;
    int cnt = 0;
    for (int i = 0; i < n; i++) {
        if (!lg[3 * i]) continue;
        auto& e = ev[(unsigned)(lg[3 * i] >> 8)];
        e.push_back({lg[3 * i + 1], +1}); e.push_back({lg[3 * i + 2], -1});
        life += (double)(lg[3 * i + 2] - lg[3 * i + 1]);
        lo = std::min(lo, lg[3 * i + 1]); hi = std::max(hi, lg[3 * i + 2]);
        cnt++;
    }
    // time a CU spends with 0, 1, 2, 3+ workgroups of this kernel (between the first start and the last end on that CU)
    double at[4] = {0, 0, 0, 0}, span = 0;
    for (auto& cu : ev) {
        std::sort(cu.second.begin(), cu.second.end());
        int live = 0;
        for (size_t k = 0; k + 1 < cu.second.size(); k++) {
            live += cu.second[k].second;
            at[std::min(live, 3)] += (double)(cu.second[k + 1].first - cu.second[k].first);
        }
        span += (double)(cu.second.back().first - cu.second.front().first);
    }
    if (cnt) fprintf(stderr, "[q8 residency] %s: %d workgroups on %zu CUs, lifetime %.2f us, %.2f resident per CU over %.0f us; share of a CU's time with 0 / 1 / 2 / 3+ resident: %.2f %.2f %.2f %.2f\n",
                     name, cnt, ev.size(), 0.01 * life / cnt, life / ((double)(hi - lo) * ev.size()), 0.01 * (double)(hi - lo),
                     at[0] / span, at[1] / span, at[2] / span, at[3] / span);
    std::fill(lg.begin(), lg.end(), 0ull);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_q8_log), lg.data(), lg.size() * 8, (size_t)which * lg.size() * 8);
}
#endif

// conv3 + conv4 of np patches: pooled conv2 output in, pooled conv4 output (p4, f32 [patch][36][90]) out
int k_cnn_q8_conv34(ck_ctx* ctx, const float* p2, int np, float* p4, int* overflow)
{
    const CnnWeights& W = ctx->cnn;
    hipLaunchKernelGGL(conv34_q8_kernel, dim3(np), dim3(256), 0, ctx->stream, p2, (const uint16_t*)W.c3w_h2.p, (const uint8_t*)W.c3x_q8.p,
                       (const float*)W.c3b.p, (const uint16_t*)W.c4w_h2.p, (const uint8_t*)W.c4x_q8.p, (const float*)W.c4b.p, p4, overflow);
    CK_HIP(ctx, hipGetLastError());
#if Q8_DBG_TIME
    {
        unsigned long long hp[16];
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_q8_prof), sizeof hp);
        if (hp[15])
            fprintf(stderr, "[q8 phases, us per workgroup] conv34 (%llu): stage %.2f  conv3 %.2f  relayout %.2f  conv4 %.2f  epilogue %.2f\n",
                    hp[15], hp[8] * 0.01 / hp[15], hp[9] * 0.01 / hp[15], hp[10] * 0.01 / hp[15], hp[11] * 0.01 / hp[15], hp[12] * 0.01 / hp[15]);
        memset(hp, 0, sizeof hp);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_q8_prof), hp, sizeof hp);
        q8_residency("conv34", 1, np);
    }
#endif
    return CK_OK;
}

// conv1 + conv2 of np patches: goban images in, pooled conv2 output (p2, f32 [patch][256][32]) out
int k_cnn_q8_conv12(ck_ctx* ctx, const uint8_t* gob, int np, float* p2, int* overflow)
{
    const CnnWeights& W = ctx->cnn;
    hipLaunchKernelGGL(conv12_q8_kernel, dim3(3 * np), dim3(256), 0, ctx->stream, gob, (const uint16_t*)W.c1w_q8.p, (const float*)W.c1b.p,
                       (const uint16_t*)W.c2w_h2.p, (const uint8_t*)W.c2x_q8.p, (const float*)W.c2b.p, p2, overflow);
    CK_HIP(ctx, hipGetLastError());
#if Q8_DBG_TIME
    {
        unsigned long long hp[16];
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_q8_prof), sizeof hp);
        if (hp[7])
            fprintf(stderr, "[q8 phases, us per workgroup] conv12 (%llu): stage %.2f  conv1 %.2f  wait %.2f  k-loop %.2f  epilogue %.2f\n",
                    hp[7], hp[0] * 0.01 / hp[7], hp[1] * 0.01 / hp[7], hp[2] * 0.01 / hp[7], hp[3] * 0.01 / hp[7], hp[4] * 0.01 / hp[7]);
        memset(hp, 0, sizeof hp);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_q8_prof), hp, sizeof hp);
        q8_residency("conv12", 0, 3 * np);
    }
#endif
    return CK_OK;
}
