// k_cnn_bf16.hip -- the convolutions of the stone classifier in CK_CNN_BF16 mode (BASELINE config 5: "stone-CNN in bf16
// on MFMA"; reference: NNManager.create_net, src/camkifu/stone/nn_manager.py:277-298, patches :216-218, 256-275).
//
// Two kernels, one MFMA per product (v_mfma_f32_16x16x32_bf16, f32 accumulate), no activation of conv1 or conv3 in HBM:
//   conv12_bf16_kernel  conv1 (5x5x3 -> 32, relu) + conv2 (5x5x32 -> 32, relu, 2x2 max-pool) of HALF a patch per workgroup
//   conv34_bf16_kernel  conv3 (3x3x32 -> 90, relu) + conv4 (3x3x90 -> 90, relu, 2x2 max-pool) of one patch per workgroup
// What shapes both: with N = 32 output channels an A fragment read from LDS feeds only two MFMAs, so a classic
// (pixel tiles) x (channel tiles) register block reads one 1 KB fragment per two or three MFMAs and the LDS, not the
// matrix pipe, sets the pace.  Here a pixel tile is 16 consecutive pixels of ONE row and a wave owns a column strip of R
// output rows: the fragment of input row y serves the taps i = 0 .. KH-1 of output rows y - i, i.e. up to KH x (channel
// tiles) MFMAs per LDS read, and the weight fragments of a tap column stay in registers for the whole strip.  The 2x2
// max-pool stays inside a lane: the four accumulator registers of a lane are four consecutive pixels of a row (two
// horizontal pairs), the rows of a pair are two accumulators of the same wave.
#include <algorithm>

#include "ck_common.h"

#ifndef BF_C12_MINW
#define BF_C12_MINW 3        // workgroups (of four waves) per CU the compiler is asked to leave room for: 3 x 53 760 B of LDS
#endif
#ifndef BF_C2_D
#define BF_C2_D 5            // conv2: depth of the fragment ring (reads in flight + the one in use)
#endif
#ifndef BF_DBG_TIME
#define BF_DBG_TIME 0        // profiling aid: phase times per workgroup (wave 0, 100 MHz wall clock) summed into g_bf_prof,
#endif                       // printed by the host after every launch pair
#ifndef BF_C3_D
#define BF_C3_D 4            // conv3: depth of the fragment ring
#endif
#ifndef BF_C4_D
#define BF_C4_D 6            // conv4: depth of the fragment ring (two MFMAs per fragment)
#endif
#ifndef BF_C4_PF
#define BF_C4_PF 2           // conv4: k-steps the weight fragments run ahead
#endif
#ifndef BF_C34_MINW
#define BF_C34_MINW 3        // 4 workgroups of three waves per CU
#endif

namespace {

#if BF_DBG_TIME
__device__ unsigned long long g_bf_prof[16];
#define BF_STAMP(K) do { if (threadIdx.x == 0) { const unsigned long long now__ = wall_clock64(); atomicAdd(&g_bf_prof[K], now__ - t_prev__); t_prev__ = now__; } } while (0)
#define BF_STAMP_BEGIN unsigned long long t_prev__ = wall_clock64()
#else
#define BF_STAMP(K) do { } while (0)
#define BF_STAMP_BEGIN do { } while (0)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// region index -> first pixel row / column of its 40x40 patch (nn_manager.py:92-126, 256-275)
__device__ __forceinline__ int region_origin(int i) { return i == 9 ? 340 : 40 * i; }

// two f32 -> packed bf16 (round to nearest even, v_cvt_pk_bf16_f32) with the relu applied on the packed halves
// (a negative bf16 is a negative int16: v_pk_max_i16 against 0)
__device__ __forceinline__ uint32_t pk_bf16_relu(float a, float b)
{
    const bf16x2 v = __builtin_convertvector(f32x2{a, b}, bf16x2);
    const s16x2 s = __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), s16x2{0, 0});
    return __builtin_bit_cast(uint32_t, s);
}
__device__ __forceinline__ uint16_t bf16_relu(float a)
{
    return (uint16_t)(pk_bf16_relu(a, 0.f) & 0xFFFFu);
}

// 16-byte chunk c (8 channels) of the pixel at column x of a 32-channel bf16 tile row sits in slot c ^ swz32(x): found by
// exhaustive search over linear swizzles against the real ds_read_b128 lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31}, + 32: MI355X_MICROARCH.md, LDS) -- a fragment (16 consecutive pixels x 4 chunks) is read in 4 LDS cycles at
// every column offset, and the 8-byte stores of conv1 (16 consecutive pixels per lane group) go two deep.
__device__ __forceinline__ int swz32(int x) { return (x >> 1) & 3; }

// ------------------------------------------------------------------------------------------------------------------
// conv1 + conv2 of half a patch: 16 output rows x 32 columns of conv2 = 8 x 16 pooled pixels x 32 channels.
//   grid  : 2 * npatch workgroups of 256 threads; item = (patch, half)
//   goban : [frame][380][380][3] u8 (the K10 patch gather is fused: region_origin)
//   w1    : conv1 weights as fp16 A fragments [channel tile 2][k-step 4][lane][8]: k = 32 s + 8 kq + e <-> fragment
//           index f = 4 s + kq = kernel row i * 3 + tap pair p (f = 15: zero), e = (tap 2 p + (e >> 2), channel e & 3);
//           the u8 pixels are exact halves, so conv1 runs on v_mfma_f32_16x16x32_f16 with 11-bit weights (flip applied)
//   w2    : conv2 weights bf16 [channel tile 2][tap 25][lane][8] (pack_bf of k_cnn.hip: lane = kslot * 16 + channel)
//   out   : [patch][16 * 16 pooled pixels][32] bf16
// LDS: the staged pixels as halves [24 rows][40 px][B, G, R, 0] (a fragment = two neighbouring taps = 16 aligned bytes)
// and conv1's output tile [20 rows][36 px][32 ch] bf16, swizzled (swz32).  53 760 B = 42 allocation units of 1 280 B: three
// workgroups per CU.  (With two padding columns per pixel row, 54 144 B, only TWO were resident -- tools/micro/wave_placement.hip:
// LDS is handed out in units of 1 280 B and 3 x 43 of them exceed the CU's 128.  The sixth tap of a kernel row, which the
// padding fed, has zero weights and needs finite halves only: the last pixel tile column re-reads its fifth.)
#ifndef BF_C12_PIXPAD
#define BF_C12_PIXPAD 0      // developer A/B: 1 = the two padding columns back (54 144 B, two workgroups per CU)
#endif
constexpr int C12_PIX_RS = (40 + 2 * BF_C12_PIXPAD) * 4, C12_PIX_ROWS = 24;          // halves
constexpr int C12_TILE_RS = 36 * 32, C12_TILE_ROWS = 20;       // halves

__global__ __launch_bounds__(256, BF_C12_MINW) void conv12_bf16_kernel(
    const uint8_t* __restrict__ goban, const uint16_t* __restrict__ w1, const float* __restrict__ b1,
    const uint16_t* __restrict__ w2, const float* __restrict__ b2, uint16_t* __restrict__ out)
{
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) uint16_t tile[C12_TILE_ROWS * C12_TILE_RS];
    __shared__ __attribute__((aligned(16))) uint16_t pix[C12_PIX_ROWS * C12_PIX_RS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    // (round 5, measured level and dropped: a workgroup that takes both halves of a patch with the second half's pixels
    // requested before the first half's k-loop -- 5.2 against 4.9-5.1 us per frame: three independent workgroups per CU hide the
    // staging latency already)
    const int patch = blockIdx.x >> 1;
    const int frame = patch / 100, reg = patch % 100;
    const int px0 = region_origin(reg % 10);
    BF_STAMP_BEGIN;
    // the 24 x 40 pixels a half needs: a thread fetches 12 bytes (4 pixels) ...
    uint32_t d0 = 0, d1 = 0, d2 = 0;
    auto load_raw = [&](int half_) {
        if (tid < 240) {
            const int r = tid / 10, g = tid % 10;
            const uint32_t* sp = reinterpret_cast<const uint32_t*>(goban + ((size_t)frame * 380 + region_origin(reg / 10) + 16 * half_ + r) * 1140 + (size_t)px0 * 3) + 3 * g;
            d0 = sp[0]; d1 = sp[1]; d2 = sp[2];
        }
    };
    // ... and turns them into 32 bytes of halves.  Byte b becomes the half 0x6400 | b = 1024 + b (one v_perm_b32 per two values
    // against a constant), minus 1024 by one packed subtraction.
    auto store_pix = [&]() {
        if (tid < 240) {
            const int r = tid / 10, g = tid % 10;
            const uint32_t p1 = __builtin_amdgcn_alignbit(d1, d0, 24), p2 = __builtin_amdgcn_alignbit(d2, d1, 16);
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const h2 k1024 = {(_Float16)1024.f, (_Float16)1024.f};
            auto bg = [&](uint32_t p, uint32_t sel) {       // two bytes of p as halves
                return __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, __builtin_amdgcn_perm(p, 0x64646464u, sel)) - k1024);
            };
            uint4 lo, hi;
            lo.x = bg(d0, 0x00050004u); lo.y = bg(d0, 0x000C0006u);       // (B, G), (R, 0): selector 0x0C is the constant byte 0
            lo.z = bg(p1, 0x00050004u); lo.w = bg(p1, 0x000C0006u);
            hi.x = bg(p2, 0x00050004u); hi.y = bg(p2, 0x000C0006u);
            hi.z = bg(d2, 0x00060005u); hi.w = bg(d2, 0x000C0007u);       // the fourth pixel is bytes 1 .. 3 of d2
            uint4* d = reinterpret_cast<uint4*>(&pix[r * C12_PIX_RS + 16 * g]);
            d[0] = lo; d[1] = hi;
        }
    };
#if BF_C12_PIXPAD
    if (tid >= 240)
        for (int r = tid - 240; r < C12_PIX_ROWS; r += 16) *reinterpret_cast<uint4*>(&pix[r * C12_PIX_RS + 160]) = make_uint4(0, 0, 0, 0);
#endif
    const int half = blockIdx.x & 1;
    load_raw(half);
    store_pix();

    // conv1's weights (A operand) and bias while the pixels land
    h8 wa[4][2];
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int n = 0; n < 2; n++) wa[s][n] = __builtin_bit_cast(h8, reinterpret_cast<const uint4*>(w1)[(n * 4 + s) * 64 + lane]);
    float4 bv1[2];
#pragma unroll
    for (int n = 0; n < 2; n++) bv1[n] = *reinterpret_cast<const float4*>(b1 + n * 16 + 4 * kq);
    // fragment f = 4 s + kq of a pixel: kernel row f / 3, taps 2 (f % 3) and 2 (f % 3) + 1; f = 15 has zero weights and
    // re-reads fragment 14
    int foff[4];
    bool last_pair[4];                                     // the fragment holds taps 4 and 5 (the sixth: zero weights)
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int f = 4 * s + kq > 14 ? 14 : 4 * s + kq;
        foff[s] = (f / 3) * C12_PIX_RS + 8 * (f % 3);
        last_pair[s] = f % 3 == 2;
    }
    __syncthreads();
    BF_STAMP(0);                                           // pixels staged, conv1's weights here

    // ---- conv1: 20 rows x 36 pixels = 45 tiles of 16 raster pixels.  D = W x P: a lane ends up with four consecutive
    // channels of one pixel per channel tile -> relu, two packed conversions, one 8-byte store into the swizzled tile.
    // A wave's tiles are t = wave + 4 i, i = 0 .. 10 (and 44 for wave 0).  The pixel fragments of tile i + 1 are requested
    // before the MFMAs of tile i (a tile alone is one dependent chain: LDS -> 8 MFMAs -> conversion -> store).
    {
        auto frags = [&](int t, uint4 (&pf)[4]) {
            const int m = 16 * t + l15, my = m / 36, mx = m - 36 * my;
            const uint16_t* pp = &pix[my * C12_PIX_RS + 4 * mx];
#pragma unroll
            for (int s = 0; s < 4; s++) {
                // (pixel column 40 does not exist: under the sixth tap's zero weights the last column reads its fifth tap's pixel again)
                const uint2 f0 = *reinterpret_cast<const uint2*>(pp + foff[s]), f1 = *reinterpret_cast<const uint2*>(pp + foff[s] + (last_pair[s] && mx == 35 ? 0 : 4));
                pf[s] = make_uint4(f0.x, f0.y, f1.x, f1.y);
            }
        };
        auto tile_of = [&](int t, const uint4 (&pf)[4]) {
            const int m = 16 * t + l15, my = m / 36, mx = m - 36 * my;
            f32x4 c1[2];
#pragma unroll
            for (int n = 0; n < 2; n++) { c1[n][0] = bv1[n].x; c1[n][1] = bv1[n].y; c1[n][2] = bv1[n].z; c1[n][3] = bv1[n].w; }
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int n = 0; n < 2; n++) c1[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[s][n], __builtin_bit_cast(h8, pf[s]), c1[n], 0, 0, 0);
            uint16_t* tp = &tile[my * C12_TILE_RS + 32 * mx + 4 * (kq & 1)];
            const int sw = swz32(mx);
#pragma unroll
            for (int n = 0; n < 2; n++) {
                const uint2 v = make_uint2(pk_bf16_relu(c1[n][0], c1[n][1]), pk_bf16_relu(c1[n][2], c1[n][3]));
                *reinterpret_cast<uint2*>(tp + (((2 * n + (kq >> 1)) ^ sw) << 3)) = v;
            }
        };
        uint4 pf[2][4];
        frags(wave, pf[0]);
#pragma unroll
        for (int i = 0; i < 11; i++) {
            const int t = wave + 4 * i;
            if (i + 1 < 11 || wave == 0) frags(t + 4, pf[(i + 1) & 1]);      // (tile 44 exists for wave 0 only)
            __builtin_amdgcn_sched_barrier(0);
            tile_of(t, pf[i & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (wave == 0) tile_of(44, pf[1]);
    }
    BF_STAMP(1);                                           // wave 0's conv1 tiles
    __syncthreads();
    BF_STAMP(2);                                           // ... the other waves'

    // ---- conv2: wave = (channel tile n, 8 output rows, both 16-column strips).  Per tap column j the five weight
    // fragments (taps (0..4, j)) are held in registers (the next column's are in flight); input row y of a strip is read
    // once and feeds output rows y - 4 .. y.
    const int n = wave & 1, rh = wave >> 1;
    f32x4 acc[8][2];
    {
        const float bb = b2[16 * n + l15];
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int s = 0; s < 2; s++) { acc[r][s][0] = bb; acc[r][s][1] = bb; acc[r][s][2] = bb; acc[r][s][3] = bb; }
    }
    const uint4* wq = reinterpret_cast<const uint4*>(w2) + (size_t)n * 25 * 64 + lane;
    uint4 bq[2][5];
#pragma unroll
    for (int i = 0; i < 5; i++) bq[0][i] = wq[(i * 5) * 64];
    // The 120 fragment reads (5 tap columns x 12 input rows x 2 strips) run BF_C2_D - 1 reads ahead of their MFMAs through a
    // register ring, the next column's weights are requested when a column starts; scheduling barriers keep the compiler
    // from sinking either next to its first use (which is what it does otherwise: 2 reads, lgkmcnt(1), 5 MFMAs, ...).
    int ab[5][2];
#pragma unroll
    for (int j = 0; j < 5; j++)
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int x = 16 * s + l15 + j;
            ab[j][s] = (8 * rh) * C12_TILE_RS + 32 * x + ((kq ^ swz32(x)) << 3);
        }
    constexpr int D = BF_C2_D, NT = 5 * 24;
    uint4 ar[D];
    auto a_read = [&](int t) { return *reinterpret_cast<const uint4*>(&tile[ab[t / 24][t % 2] + ((t % 24) / 2) * C12_TILE_RS]); };
#pragma unroll
    for (int t = 0; t < D - 1; t++) ar[t] = a_read(t);
#pragma unroll
    for (int j = 0; j < 5; j++) {
#pragma unroll
        for (int y = 0; y < 12; y++) {
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int t = 24 * j + 2 * y + s;
                if (t + D - 1 < NT) ar[(t + D - 1) % D] = a_read(t + D - 1);
                if (t % 24 == 0 && j + 1 < 5) {
#pragma unroll
                    for (int i = 0; i < 5; i++) bq[(j + 1) & 1][i] = wq[(i * 5 + j + 1) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 a = __builtin_bit_cast(bf16x8, ar[t % D]);
#pragma unroll
                for (int i = 0; i < 5; i++) {
                    const int r = y - i;
                    if (r >= 0 && r < 8)
                        acc[r][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, bq[j & 1][i]), acc[r][s], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    BF_STAMP(3);                                           // wave 0's k-loop
    // ---- 2x2 max-pool in the lane (the bias is in the sums already; max commutes with the relu), bf16, out
    uint16_t* o = out + (size_t)patch * 256 * 32 + 16 * n + l15;
#pragma unroll
    for (int r2 = 0; r2 < 4; r2++)
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const float v = fmaxf(fmaxf(acc[2 * r2][s][2 * p], acc[2 * r2][s][2 * p + 1]), fmaxf(acc[2 * r2 + 1][s][2 * p], acc[2 * r2 + 1][s][2 * p + 1]));
                const int py = 8 * half + 4 * rh + r2, px = 8 * s + 2 * kq + p;
                o[(py * 16 + px) * 32] = bf16_relu(v);
            }
    BF_STAMP(4);                                           // epilogue issued
#if BF_DBG_TIME
    if (threadIdx.x == 0) atomicAdd(&g_bf_prof[7], 1ull);
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// conv3 + conv4 of one patch per workgroup of three waves; wave w owns the channel tiles 2 w and 2 w + 1 of both layers.
//   in  : [patch][16 * 16][32] bf16 (conv12's output)      out : [patch][6 * 6][96] bf16 (channels 90 .. 95 zero)
//   w3  : bf16 [channel tile 6][tap 9][lane][8]             w4  : bf16 [channel tile 6][k-step 27 = tap * 3 + cc][lane][8]
// conv3 runs as D = W x P on ROW tiles (16 columns of one output row, 14 real: the two others read the next row's first
// pixels and are never stored), so that an input row's fragment feeds the three taps of a column and a lane ends up with
// four consecutive channels of a pixel -- the 8-byte unit of conv4's tile.  Its 14 rows take two passes of 7 (the
// accumulators of 14 rows x 2 channel tiles would not leave room for the weights); the first pass waits packed to bf16.
// conv4 runs as D = P x W on the nine 4 x 4 pooling tiles of its 12 x 12 output (exact cover, pool in the lane).
// LDS: conv3's input [16][16][32] swizzled like conv2's tile (16 KB), overlaid after a barrier by conv4's tile
// [14][14][96] with chunk (cc, kq) of the pixel in row y at slot 4 cc + (kq ^ 2 (y & 1)) -- conflict-free for the
// pooling-tile fragment reads under the real lane groups, no padding: 37 632 B, four workgroups per CU.
constexpr int C34_T4_PS = 96, C34_T4_RS = 14 * 96;              // halves

__global__ __launch_bounds__(192, BF_C34_MINW) void conv34_bf16_kernel(
    const uint16_t* __restrict__ in, const uint16_t* __restrict__ w3, const float* __restrict__ b3,
    const uint16_t* __restrict__ w4, const float* __restrict__ b4, uint16_t* __restrict__ out)
{
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) uint16_t lds[14 * C34_T4_RS];
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int patch = blockIdx.x;
    BF_STAMP_BEGIN;

    {
        const uint4* g = reinterpret_cast<const uint4*>(in + (size_t)patch * 256 * 32);
        for (int c = tid; c < 1024; c += 192) {
            const int px = c >> 2, ch = c & 3, x = px & 15;
            *reinterpret_cast<uint4*>(&lds[32 * px + ((ch ^ swz32(x)) << 3)]) = g[c];
        }
    }
    __syncthreads();
    BF_STAMP(8);                                           // input staged

    // ---- conv3: rows 7 .. 13 first -- their place in conv4's tile lies behind conv3's input (7 x 2 688 B > 16 KB), so they
    // are stored at once; rows 0 .. 6 wait packed in registers for the barrier behind which the input may be overwritten
    auto store_row = [&](int oy, int n, uint32_t v01, uint32_t v23) {
        const int c8 = 2 * (2 * wn + n) + (kq >> 1);         // 16-byte chunk of the pixel: 0 .. 11
        const int slot = (c8 & ~3) | ((c8 & 3) ^ ((oy & 1) << 1));
        if (l15 < 14) *reinterpret_cast<uint2*>(&lds[oy * C34_T4_RS + l15 * C34_T4_PS + 8 * slot + 4 * (kq & 1)]) = make_uint2(v01, v23);
    };
    uint32_t c3[7][2][2];                                  // rows 0 .. 6 as packed bf16: [row][channel tile][pair of channels]
    {
        float4 bv[2];
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const int c0 = 16 * (2 * wn + n) + 4 * kq;      // 90 real channels: 88 .. 91 straddles the end
            bv[n].x = c0 < 90 ? b3[c0] : 0.f; bv[n].y = c0 + 1 < 90 ? b3[c0 + 1] : 0.f;
            bv[n].z = c0 + 2 < 90 ? b3[c0 + 2] : 0.f; bv[n].w = c0 + 3 < 90 ? b3[c0 + 3] : 0.f;
        }
#pragma unroll
        for (int pass = 1; pass >= 0; pass--) {
            // (an opaque copy of the pointer per pass: the weight fragments of the two passes are the same loads, and kept
            // live across both they cost 72 registers)
            const uint4* wq = reinterpret_cast<const uint4*>(w3) + (size_t)(2 * wn) * 9 * 64 + lane;
            asm volatile("" : "+v"(wq));
            f32x4 acc[7][2];
#pragma unroll
            for (int r = 0; r < 7; r++)
#pragma unroll
                for (int n = 0; n < 2; n++) { acc[r][n][0] = bv[n].x; acc[r][n][1] = bv[n].y; acc[r][n][2] = bv[n].z; acc[r][n][3] = bv[n].w; }
            // fragment reads BF_C3_D - 1 ahead of their MFMAs, the weights of the next tap column requested when a column
            // starts (as in conv12_bf16_kernel)
            constexpr int D = BF_C3_D, NT = 27;
            int ab[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int x = l15 + j;
                ab[j] = 32 * x + ((kq ^ swz32(x)) << 3) + 7 * pass * 512;
            }
            auto p_read = [&](int t) { return *reinterpret_cast<const uint4*>(&lds[ab[t / 9] + (t % 9) * 512]); };
            uint4 ar[D], wa[2][3][2];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int n = 0; n < 2; n++) wa[0][i][n] = wq[((size_t)n * 9 + i * 3) * 64];
#pragma unroll
            for (int t = 0; t < D - 1; t++) ar[t] = p_read(t);
#pragma unroll
            for (int j = 0; j < 3; j++) {
#pragma unroll
                for (int y = 0; y < 9; y++) {
                    const int t = 9 * j + y;
                    if (t + D - 1 < NT) ar[(t + D - 1) % D] = p_read(t + D - 1);
                    if (y == 0 && j + 1 < 3) {
#pragma unroll
                        for (int i = 0; i < 3; i++)
#pragma unroll
                            for (int n = 0; n < 2; n++) wa[(j + 1) & 1][i][n] = wq[((size_t)n * 9 + i * 3 + j + 1) * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const bf16x8 p = __builtin_bit_cast(bf16x8, ar[t % D]);
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        const int r = y - i;
                        if (r >= 0 && r < 7) {
#pragma unroll
                            for (int n = 0; n < 2; n++)
                                acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa[j & 1][i][n]), p, acc[r][n], 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int r = 0; r < 7; r++)
#pragma unroll
                for (int n = 0; n < 2; n++) {
                    const uint32_t v01 = pk_bf16_relu(acc[r][n][0], acc[r][n][1]), v23 = pk_bf16_relu(acc[r][n][2], acc[r][n][3]);
                    if (pass == 1) store_row(7 + r, n, v01, v23);
                    else { c3[r][n][0] = v01; c3[r][n][1] = v23; }
                }
        }
    }
    BF_STAMP(9);                                           // wave 0's conv3
    __syncthreads();                                       // every wave is done with conv3's input, which rows 0 .. 6 overlay
#pragma unroll
    for (int oy = 0; oy < 7; oy++)
#pragma unroll
        for (int n = 0; n < 2; n++) store_row(oy, n, c3[oy][n][0], c3[oy][n][1]);
    __syncthreads();
    BF_STAMP(10);                                          // barrier, rows 0 .. 6 stored, barrier

    // ---- conv4: pooling tile t = (ty, tx): lane row l15 = window q, corner `sub`
    f32x4 acc[9][2];
#pragma unroll
    for (int n = 0; n < 2; n++) {
        const int co = 16 * (2 * wn + n) + l15;
        const float bb = co < 90 ? b4[co] : 0.f;
#pragma unroll
        for (int t = 0; t < 9; t++) { acc[t][n][0] = bb; acc[t][n][1] = bb; acc[t][n][2] = bb; acc[t][n][3] = bb; }
    }
    const int q = l15 >> 2, sub = l15 & 3, dy = 2 * (q >> 1) + (sub >> 1), dx = 2 * (q & 1) + (sub & 1);
    // fragment address of the lane's pixel for taps in even rows / odd rows (the row parity flips bit 1 of the slot)
    const int a_even = dy * C34_T4_RS + dx * C34_T4_PS + ((kq ^ ((dy & 1) << 1)) << 3);
    const int a_odd = dy * C34_T4_RS + dx * C34_T4_PS + ((kq ^ (((dy + 1) & 1) << 1)) << 3);
    const uint4* wq4 = reinterpret_cast<const uint4*>(w4) + (size_t)(2 * wn) * 27 * 64 + lane;
    constexpr int PF = BF_C4_PF, D = BF_C4_D, NT = 27 * 9;
    uint4 bq[PF + 1][2], ar[D];
    auto a_read = [&](int u) {
        const int step = u / 9, t = u % 9, i = step / 9, j = (step / 3) % 3, cc = step % 3, ty = t / 3, tx = t % 3;
        return *reinterpret_cast<const uint4*>(&lds[((i & 1) ? a_odd : a_even) + (4 * ty + i) * C34_T4_RS + (4 * tx + j) * C34_T4_PS + 32 * cc]);
    };
#pragma unroll
    for (int u = 0; u < PF; u++)
#pragma unroll
        for (int n = 0; n < 2; n++) bq[u][n] = wq4[((size_t)n * 27 + u) * 64];
#pragma unroll
    for (int u = 0; u < D - 1; u++) ar[u] = a_read(u);
#pragma unroll
    for (int step = 0; step < 27; step++) {
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int u = 9 * step + t;
            if (u + D - 1 < NT) ar[(u + D - 1) % D] = a_read(u + D - 1);
            if (t == 0 && step + PF < 27) {
#pragma unroll
                for (int n = 0; n < 2; n++) bq[(step + PF) % (PF + 1)][n] = wq4[((size_t)n * 27 + step + PF) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 a = __builtin_bit_cast(bf16x8, ar[u % D]);
#pragma unroll
            for (int n = 0; n < 2; n++)
                acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, bq[step % (PF + 1)][n]), acc[t][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    BF_STAMP(11);                                          // wave 0's conv4 loop
    uint16_t* o = out + (size_t)patch * 36 * 96;
#pragma unroll
    for (int n = 0; n < 2; n++) {
        const int co = 16 * (2 * wn + n) + l15;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const float v = fmaxf(fmaxf(acc[t][n][0], acc[t][n][1]), fmaxf(acc[t][n][2], acc[t][n][3]));
            const int py = 2 * (t / 3) + (kq >> 1), px = 2 * (t % 3) + (kq & 1);
            o[(py * 6 + px) * 96 + co] = bf16_relu(v);       // channels 90 .. 95: zero weights, zero bias
        }
    }
    BF_STAMP(12);
#if BF_DBG_TIME
    if (threadIdx.x == 0) atomicAdd(&g_bf_prof[15], 1ull);
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// dense 3240 -> 160 + relu (nn_manager.py:293) on the pooled conv4 maps, bf16 operands.  One workgroup = 64 patches x all
// 160 outputs, so the activations leave HBM once (the round-1 kernel gave a wave 32 patches x 32 outputs and read every
// activation five times: 0.84 us per frame, all of it that traffic).  D = W x X: wave w owns the output tiles w, w + 4,
// w + 8 (< 10) and all four patch tiles; the activations of a chunk of 128 k go through LDS (copied as they are: the maps
// are bf16 already), the weight fragments stream from L2 in fragment order, a chunk ahead.
//   x  : [patch][36 px][96 ch] bf16 (conv34's output; channels 90 .. 95 are zero)
//   wt : bf16 [output tile 10][k-step 108][lane][8] with k = px * 96 + ch (pack below)      out : [patch][160] f32
__global__ __launch_bounds__(256) void fc1_bf16_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ wt,
                                                       const float* __restrict__ bias, float* __restrict__ out, int npatch)
{
#pragma clang fp contract(off)
    constexpr int KIN = 3456, NOUT = 160, KC = 128, NCH = KIN / KC, KS = KIN / 32, RSH = KC + 8;   // 272-byte LDS rows
    constexpr int SPC = KC / 32;                     // k-steps per chunk = slots of the weight ring (wf[SPC][3], slot = k-step within the chunk:
                                                     // wload(ks, s + SPC) refills a slot with the same k-step of the next chunk)
    constexpr int NLD = 64 * (KC / 8) / 256;         // 16-byte loads per thread and chunk
    __shared__ __attribute__((aligned(16))) uint16_t lds[64 * RSH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int p0 = blockIdx.x * 64;
    const int ntw = wave < 2 ? 3 : 2;
    const uint4* wq = reinterpret_cast<const uint4*>(wt) + lane;
    f32x4 acc[3][4];
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Few workgroups (200 per 128-frame call), each a chain of 27 chunks: what it waits for is latency, so everything runs
    // well ahead -- the activations TWO chunks (two register sets), the weight fragments a whole chunk (a ring of SPC slots:
    // the fragment of k-step s + SPC is requested into the slot step s has just used).  One step ahead, as first built, the
    // kernel took 63 us per call: 108 k-steps x an L2 round trip.
    uint4 raw[2][NLD];
    auto fetch = [&](int ch, uint4 (&dst)[NLD]) {
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int i = tid + 256 * q, row = i / (KC / 8), c8 = i % (KC / 8);
            int p = p0 + row;
            p = p > npatch - 1 ? npatch - 1 : p;
            dst[q] = ch < NCH ? *reinterpret_cast<const uint4*>(x + (size_t)p * KIN + ch * KC + 8 * c8) : make_uint4(0, 0, 0, 0);
        }
    };
    uint4 wf[SPC][3];
    auto wload = [&](int slot, int s) {
#pragma unroll
        for (int t = 0; t < 3; t++)
            if (t < ntw) wf[slot][t] = wq[((size_t)(wave + 4 * t) * KS + s) * 64];
    };
    fetch(0, raw[0]);
    fetch(1, raw[1]);
#pragma unroll
    for (int ks = 0; ks < SPC; ks++) wload(ks, ks);
    auto chunk = [&](int ch, uint4 (&mine)[NLD]) {
        __syncthreads();                              // previous chunk fully consumed
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int i = tid + 256 * q, row = i / (KC / 8), c8 = i % (KC / 8);
            *reinterpret_cast<uint4*>(&lds[row * RSH + 8 * c8]) = mine[q];
        }
        __syncthreads();
        fetch(ch + 2, mine);
#pragma unroll
        for (int ks = 0; ks < SPC; ks++) {
            const int s = ch * SPC + ks;
            bf16x8 xb[4];
#pragma unroll
            for (int j = 0; j < 4; j++) xb[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&lds[(16 * j + l15) * RSH + 32 * ks + 8 * kq]));
#pragma unroll
            for (int t = 0; t < 3; t++)
                if (t < ntw) {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ks][t]), xb[j], acc[t][j], 0, 0, 0);
                }
            if (s + SPC < KS) wload(ks, s + SPC);
        }
    };
    static_assert(NCH % 2 == 1, "27 chunks: pairs, then one");
    for (int ch = 0; ch + 1 < NCH; ch += 2) {
        chunk(ch, raw[0]);
        chunk(ch + 1, raw[1]);
    }
    chunk(NCH - 1, raw[0]);
#pragma unroll
    for (int t = 0; t < 3; t++)
        if (t < ntw) {
            const int o0 = (wave + 4 * t) * 16 + 4 * kq;
            const float4 bv = *reinterpret_cast<const float4*>(bias + o0);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int p = p0 + 16 * j + l15;
                float4 v = make_float4(acc[t][j][0] + bv.x, acc[t][j][1] + bv.y, acc[t][j][2] + bv.z, acc[t][j][3] + bv.w);
                v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                if (p < npatch) *reinterpret_cast<float4*>(out + (size_t)p * NOUT + o0) = v;
            }
        }
}

}  // namespace

// conv1's weights for conv12_bf16_kernel: fp16 A fragments [channel tile][k-step][lane = kslot * 16 + channel][8] (layout at
// the kernel).  `k1` is the Keras kernel [kh][kw][cin][cout]; the flip of the Theano convolution is applied here.
int k_cnn_bf16_pack_conv1(ck_ctx* ctx, const float* k1, DevBuf& dst)
{
    std::vector<uint16_t> v((size_t)2 * 4 * 64 * 8, 0);
    for (int nt = 0; nt < 2; nt++)
        for (int s = 0; s < 4; s++)
            for (int lane = 0; lane < 64; lane++)
                for (int e = 0; e < 8; e++) {
                    const int f = 4 * s + lane / 16, i = f / 3, j = 2 * (f % 3) + (e >> 2), c = e & 3, o = nt * 16 + lane % 16;
                    if (f > 14 || j > 4 || c > 2) continue;
                    const _Float16 h = (_Float16)k1[(((size_t)(4 - i) * 5 + (4 - j)) * 3 + c) * 32 + o];
                    memcpy(&v[(((size_t)nt * 4 + s) * 64 + lane) * 8 + e], &h, 2);
                }
    CK_TRY(ck_ensure(ctx, dst, v.size() * 2));
    CK_HIP(ctx, hipMemcpy(dst.p, v.data(), v.size() * 2, hipMemcpyHostToDevice));
    return CK_OK;
}

// dense-1 weights for fc1_bf16_kernel: bf16 A fragments [output tile][k-step][lane = kslot * 16 + output][8], k = px * 96 + ch over the
// padded maps; `w` is the Keras matrix [3240 = px * 90 + ch][160]
int k_cnn_bf16_pack_fc1(ck_ctx* ctx, const float* w, DevBuf& dst)
{
    std::vector<uint16_t> v((size_t)10 * 108 * 64 * 8, 0);
    for (int t = 0; t < 10; t++)
        for (int st = 0; st < 108; st++)
            for (int lane = 0; lane < 64; lane++)
                for (int e = 0; e < 8; e++) {
                    const int k = 32 * st + 8 * (lane / 16) + e, px = k / 96, c = k % 96, o = 16 * t + lane % 16;
                    if (c >= 90) continue;
                    const __bf16 b = (__bf16)w[(size_t)(px * 90 + c) * 160 + o];
                    memcpy(&v[(((size_t)t * 108 + st) * 64 + lane) * 8 + e], &b, 2);
                }
    CK_TRY(ck_ensure(ctx, dst, v.size() * 2));
    CK_HIP(ctx, hipMemcpy(dst.p, v.data(), v.size() * 2, hipMemcpyHostToDevice));
    return CK_OK;
}

int k_cnn_bf16_fc1(ck_ctx* ctx, const uint16_t* q4, int np, float* h1)
{
    hipLaunchKernelGGL(fc1_bf16_kernel, dim3((np + 63) / 64), dim3(256), 0, ctx->stream, q4, (const uint16_t*)ctx->cnn.d1w_bfp.p,
                       (const float*)ctx->cnn.d1b.p, h1, np);
    CK_HIP(ctx, hipGetLastError());
    return CK_OK;
}

// the four convolutions of `np` patches: goban images in, pooled conv2 output (p2) and pooled conv4 output (q4) out
int k_cnn_bf16_convs(ck_ctx* ctx, const uint8_t* gob, int np, uint16_t* p2, uint16_t* q4)
{
    const CnnWeights& W = ctx->cnn;
    {
        TimeScope ts(ctx, "cnn_conv2");
        hipLaunchKernelGGL(conv12_bf16_kernel, dim3(2 * np), dim3(256), 0, ctx->stream, gob, (const uint16_t*)W.c1w_f16.p,
                           (const float*)W.c1b.p, (const uint16_t*)W.c2w_bf.p, (const float*)W.c2b.p, p2);
    }
    {
        TimeScope ts(ctx, "cnn_conv4");
        hipLaunchKernelGGL(conv34_bf16_kernel, dim3(np), dim3(192), 0, ctx->stream, (const uint16_t*)p2, (const uint16_t*)W.c3w_bf.p,
                           (const float*)W.c3b.p, (const uint16_t*)W.c4w_bf.p, (const float*)W.c4b.p, q4);
    }
    CK_HIP(ctx, hipGetLastError());
#if BF_DBG_TIME
    {
        unsigned long long hp[16];
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_bf_prof), sizeof hp);
        if (hp[7] && hp[15])
            fprintf(stderr, "[bf16 phases, us per workgroup] conv12 (%llu): stage %.2f  conv1 %.2f  wait %.2f  k-loop %.2f  epilogue %.2f | "
                            "conv34 (%llu): stage %.2f  conv3 %.2f  relayout %.2f  conv4 %.2f  epilogue %.2f\n",
                    hp[7], hp[0] * 0.01 / hp[7], hp[1] * 0.01 / hp[7], hp[2] * 0.01 / hp[7], hp[3] * 0.01 / hp[7], hp[4] * 0.01 / hp[7],
                    hp[15], hp[8] * 0.01 / hp[15], hp[9] * 0.01 / hp[15], hp[10] * 0.01 / hp[15], hp[11] * 0.01 / hp[15], hp[12] * 0.01 / hp[15]);
        memset(hp, 0, sizeof hp);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bf_prof), hp, sizeof hp);
    }
#endif
    return CK_OK;
}
