// k_median.hip -- K1: exact 15x15 median of an 8-bit BGR frame, replicate border.
// Replaces cv2.medianBlur(frame, 15)  (reference: src/camkifu/board/bf_auto.py:72).
//
// Algorithm (threshold decomposition + per-tile radix descent, all in registers):
//   median(p) = min { t : #{window(p) <= t} >= 113 }.
//   For one threshold t the count is a 15x15 box sum of the indicator (x <= t), which is
//   separable and shared by every pixel of a tile.  Each pixel resolves its median bit by
//   bit (MSB first); at bit b a pixel whose decided prefix is q asks threshold
//   t = q + 2^b - 1.  A wave evaluates the box sum only for the DISTINCT prefixes q present
//   in its tile (a 256-bit wave-uniform set), so a tile costs (#distinct medians-ish)
//   box filters instead of 255 -- medians of a 15x15 window vary slowly.
//
// The box sums run on the i8 matrix cores (median_mfma_kernel below).  The first form of this kernel did them with SWAR byte
// adds and DPP row shifts (38 us per 1080p frame, VALU-issue bound: ~20.6 box filters per 48x64 tile x 376 instructions); it
// lives in tools/variants/median15_swar.hip.txt with the tile shapes and shortcuts that were costed against it.
#include <type_traits>

#include "ck_common.h"

namespace {

// ------------------------------------------------------------------------------------------
// Matrix-core variant (the one launched).  Same radix descent, but the 15x15 box sum of the indicator
// (x <= t) is two chained i8 MFMAs instead of SWAR adds and DPP shifts, which moves most of the
// per-threshold VALU work to the matrix pipe (that pipe is otherwise idle in the board path):
//   pass 1  C1[x][y_out] = sum_{y_in} Ind[y_in][x] * band[y_in][y_out]      A = the pixels themselves: a lane
//           holds one column x and 16 consecutive rows as bytes, so the indicator (-128 per byte, SWAR, 2 ops per
//           dword) IS the A operand; B = a constant band of -4.  C1 = 512 * (vertical count) <= 7680: as four
//           bytes that is (0, 2 * count, 0, 0).
//   pass 2  R[y_out][x_out] = sum_{x_in} C1[y_out][x_in] * band[x_in][x_out] + 256 * 113.  The four i32 results a
//           lane holds after pass 1 (4 adjacent columns of one output row) are used AS THEY ARE as the 16 k-bytes
//           of the A operand: 12 of the 16 bytes are zero and the constant B (band of -128) has its weights at the
//           other four, so nothing is repacked; a 16-column output tile needs the two column tiles its 30-column
//           window spans = two accumulating MFMAs.  R = 256 * (113 - S):  >= 256 if the median is above t, <= 0
//           otherwise.
//   update  med = med3(med, R, q + 2^b): one VALU op per pixel.  A pixel whose prefix is not q is left alone by the
//           arithmetic itself: prefix > q means S < 113 and med > q + 2^b (median of the three = med); prefix < q means
//           S >= 113, R <= 0 < med < q + 2^b.
// The prefixes alive at the next level are published through 256 flag bytes in LDS (ds_write_b8 with the
// median as the address: no VALU work) once per level.
// Tile: 64x64 input pixels (62 used) -> 48x48 medians per wave; 12 + 18 MFMAs and 68 VALU ops per threshold
// (32 indicator, 36 update) against 376 VALU ops per 48x64 tile in the SWAR kernel above; ~19.6 thresholds per
// tile on 1080p board scenes.
// What bounds it (rocprofv3 --pmc, tools/pmc_median.sh; micro-benchmarks tools/micro/mfma_i8.hip, mfma_form.hip):
// one v_mfma_i32_16x16x64_i8 holds the matrix pipe for 16-18 cycles and keeps the SIMD from issuing VALU ops
// for ~8.5 of them, whichever register file its operands live in; VALU ops of other waves fill the rest.  Per
// threshold that is 84 x 4.4 (VALU, with the per-tile overhead) + 30 x 8.5 = ~625 cycles against 30 x 17 = 510
// of matrix-pipe time: the counters show the matrix pipe 56 % busy, the vector ALUs 63 %, both at once 25 % of
// the time.  Tried on top, all within noise of each other (19.6 - 20.8 us on one box):
//   * merging the two column-tile quads of a window into one operand (byte 1 | byte 3, one v_lshl_or per dword)
//     for 0 / 6 / 7 / 9 of the 9 output tiles: trades 4 VALU ops for 1 MFMA, the same cost either way;
//   * packing all four column tiles into one 64-column operand (3 v_perm per quad, 9 pass-2 MFMAs): 21.6 us;
//   * explicit MFMA / VALU interleaving of one wave's instruction stream (sched_group_barrier): no change,
//     the 4 waves of a SIMD already fill each other's gaps.
typedef int v4i __attribute__((ext_vector_type(4)));

#ifndef MED_DBG
#define MED_DBG 0          // 1: count tiles / scanned tiles / scans given up / (threshold, block) box counts of both kinds
#endif
#if MED_DBG
__device__ unsigned long long g_med_dbg[8];
#define MED_COUNT(K, V) do { if (lane == 0) atomicAdd(&g_med_dbg[K], (unsigned long long)(V)); } while (0)
#else
#define MED_COUNT(K, V) do { } while (0)
#endif

constexpr int MT = 48;      // medians per tile edge

struct BandTable {
    uint32_t v[5 * 64 * 4];  // [vertical band for output-row tile 0..2 | horizontal band, same / next column tile][lane][4]
};
constexpr BandTable make_band_table(int K)
{
    BandTable T{};
    for (int lane = 0; lane < 64; lane++) {
        const int n = lane & 15, g = lane >> 4;
        for (int t = 0; t < 3; t++)
            for (int d = 0; d < 4; d++) {
                uint32_t wv = 0;
                for (int e = 0; e < 4; e++) {
                    const int y_in = 16 * g + 4 * d + e;          // pass 1: k position (g, 4d+e) <-> input row
                    const int dv = y_in - (16 * t + n);
                    if (dv >= 0 && dv <= K - 1) wv |= 0xFCu << (8 * e);  // -4
                }
                T.v[(t * 64 + lane) * 4 + d] = wv;
            }
        for (int k = 0; k < 2; k++)
            for (int e = 0; e < 4; e++) {
                // pass 2: dword e of the A operand is the raw i32 of column 4g + e of column tile u + k; its byte 1
                // carries the value.  Output column n of tile u.
                const int dh = 16 * k + 4 * g + e - n;
                T.v[((3 + k) * 64 + lane) * 4 + e] = (dh >= 0 && dh <= K - 1) ? (0x80u << 8) : 0u;  // -128
            }
    }
    return T;
}
template <int K>
__device__ const BandTable g_band = make_band_table(K);

__device__ __forceinline__ int imed3(int a, int b, int c)
{
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    const int t = hi < c ? hi : c;
    return lo > t ? lo : t;                 // v_med3_i32
}

// waves_per_eu(3): a register budget below 256 makes the compiler pick the MFMA forms that write VGPRs; with the
// default budget the results land in AGPRs and every one of them costs a v_accvgpr_read.
// K = window edge (odd, 3..17: a 64-pixel input tile must cover 48 + K - 1); rank (K*K+1)/2, halo K/2.
template <int K>
__attribute__((amdgpu_waves_per_eu(3))) __global__ __launch_bounds__(64) void median_mfma_kernel(
    const uint8_t* __restrict__ in, int h, int w, uint8_t* __restrict__ out, int pitch, uint8_t* __restrict__ range)
{
    static_assert(K % 2 == 1 && K >= 3 && MT + K - 1 <= 64, "window size");
    static_assert(MT == CK_RANGE_TILE, "the value bounds are per tile of this kernel");
    constexpr int HK = K / 2, RANK = (K * K + 1) / 2;
    __shared__ uint32_t flags[3][64];
    const int lane = threadIdx.x;
    const int n = lane & 15, g = lane >> 4;
    // XCD-aware tile order (as in the NMS kernel): workgroups go round-robin over the 8 XCDs by linear id; XCD k takes the
    // k-th contiguous eighth of the (x, y, plane) sequence, so horizontal neighbours share an L2 -- their halo columns are
    // fetched once and the two halves of a 64-byte output line (tile rows are 48 bytes) meet in one L2 before write-back.
    int bxi = blockIdx.x, byi = blockIdx.y, bzi = blockIdx.z;
    {
        const unsigned G = gridDim.x * gridDim.y * gridDim.z;
        if ((G & 7u) == 0) {
            const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            const unsigned t = (lin & 7u) * (G >> 3) + (lin >> 3);
            // channel innermost: the three planes of a tile position read the same interleaved BGR lines
            const unsigned pos = t / 3u;
            bxi = (int)(pos % gridDim.x);
            byi = (int)((pos / gridDim.x) % gridDim.y);
            bzi = (int)(3u * (pos / (gridDim.x * gridDim.y)) + t % 3u);
        }
    }
    const int ox = bxi * MT, oy = byi * MT;
    const int f = bzi / 3, c = bzi % 3;

    // ---- load, TRANSPOSED (round 4): row tile i, lane (n, g) <- row oy - K/2 + 16 i + n, columns ox - K/2 + 16 g + 0..15
    // of channel c, as 16 bytes.  The box count is symmetric in x and y, so the same two MFMA passes run on the
    // transposed tile (pass 1 sums along a row, pass 2 down the columns) and a lane ends up with FOUR HORIZONTALLY
    // ADJACENT medians of one row: med[t][u][e] <-> row 16 u + n, column 16 t + 4 g + e.  What that buys is the memory
    // side: a lane's 16 pixels are 48 contiguous bytes of the interleaved frame -- 13 aligned dwords, fetched as 16-byte
    // loads and picked apart with v_perm -- where the column-per-lane form needed 64 single-byte loads per lane, each
    // of which holds the texture-address unit for 16 cycles: ~1 000 cycles per tile, 31 tiles per CU and frame = the
    // whole 15 us the kernel took, whatever its arithmetic did (round 4: neither 20 % fewer box counts nor 24 more per
    // tile moved the time).  The finished medians leave as one dword per lane and block: no LDS transpose either.
    v4i nx[4];
    if (ox >= HK && ox - HK + 65 < w && oy >= HK && oy - HK + 63 < h) {
        // interior tile (two spare pixels behind the region: the 52 bytes fetched per lane end up to 4 bytes behind the 48
        // that are used, and those must still be the frame's).  The loads start SH bytes before the lane's first pixel:
        // with rows of whole dwords (w % 4 == 0) and the usual aligned frame buffer that is an aligned dword for every
        // lane, row and tile (3 (ox - HK + 16 g) = -3 HK mod 4: ox is a multiple of 48) -- any other pointer or width
        // just loads unaligned, the byte positions are relative to the load address either way.  The channel's bytes sit
        // at fixed offsets SH + c + 3 k: compile-time v_perm selectors.
        constexpr int SH = (4 - (3 * HK) % 4) % 4;
        struct __attribute__((packed, aligned(4))) q16 { uint32_t a, b, c, d; };
        const uint8_t* base = in + (size_t)f * h * w * 3 + ((size_t)(oy - HK) * w + (ox - HK)) * 3 - SH;
        uint32_t voff = (uint32_t)((n * w + 16 * g) * 3);
        asm volatile("" : "+v"(voff));          // keep it one per-lane offset: scalar base + voff + immediate per load
        uint32_t raw[4][13];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint8_t* p = base + (size_t)(uint32_t)(16 * i * w * 3) + voff;
#pragma unroll
            for (int v = 0; v < 3; v++) {
                const q16 d = *reinterpret_cast<const q16*>(p + 16 * v);
                raw[i][4 * v] = d.a; raw[i][4 * v + 1] = d.b; raw[i][4 * v + 2] = d.c; raw[i][4 * v + 3] = d.d;
            }
            raw[i][12] = *reinterpret_cast<const uint32_t*>(p + 48);
        }
        auto pick = [&](auto c_tag) {
            constexpr int C = decltype(c_tag)::value;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const int p0 = SH + C + 12 * d;                     // byte position of the dword's first pixel (folds: d unrolled)
                    const int q0 = p0 >> 2, q2 = (p0 + 6) >> 2;
                    const uint32_t sel_lo = (uint32_t)(p0 - 4 * q0) | ((uint32_t)(p0 + 3 - 4 * q0) << 8) | 0x0c0c0000u;
                    const uint32_t sel_hi = (uint32_t)(p0 + 6 - 4 * q2) | ((uint32_t)(p0 + 9 - 4 * q2) << 8) | 0x0c0c0000u;
                    const uint32_t lo16 = __builtin_amdgcn_perm(raw[i][q0 + 1], raw[i][q0], sel_lo);
                    const uint32_t hi16 = __builtin_amdgcn_perm(raw[i][q2 + 1 > 12 ? 12 : q2 + 1], raw[i][q2], sel_hi);
                    nx[i][d] = (int)~__builtin_amdgcn_perm(hi16, lo16, 0x05040100u);
                }
        };
        if (c == 0) pick(std::integral_constant<int, 0>{});
        else if (c == 1) pick(std::integral_constant<int, 1>{});
        else pick(std::integral_constant<int, 2>{});
    } else {
        // rim tiles (replicate border): byte by byte
        const uint8_t* src = in + (size_t)f * h * w * 3 + c;
        int xo[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            int x = ox - HK + 16 * g + j;
            x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
            xo[j] = x * 3;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int y = oy - HK + 16 * i + n;
            y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
            const uint8_t* row = src + (size_t)y * w * 3;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const uint32_t v = (uint32_t)row[xo[4 * d]] | ((uint32_t)row[xo[4 * d + 1]] << 8) |
                                   ((uint32_t)row[xo[4 * d + 2]] << 16) | ((uint32_t)row[xo[4 * d + 3]] << 24);
                nx[i][d] = (int)~v;
            }
        }
    }
    const v4i* bt = reinterpret_cast<const v4i*>(g_band<K>.v);
    v4i bv[3], bh[2];
#pragma unroll
    for (int t = 0; t < 3; t++) bv[t] = bt[t * 64 + lane];
#pragma unroll
    for (int k = 0; k < 2; k++) bh[k] = bt[(3 + k) * 64 + lane];
    const v4i zero = {0, 0, 0, 0};
    const v4i c113 = {256 * RANK, 256 * RANK, 256 * RANK, 256 * RANK};      // 113 for the 15x15 window

    int med[3][3][4];
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int u = 0; u < 3; u++)
#pragma unroll
            for (int e = 0; e < 4; e++) med[t][u][e] = 0;
#pragma unroll
    for (int t = 0; t < 3; t++) flags[t][lane] = 0;

    // One threshold t on the row blocks in `here`: indicator, vertical + horizontal box count, update.  R = 256 (RANK -
    // count) per pixel: > 0 where the median is above t.  WANT & 1: is some R of a row block > 0 (bit t of `pos`);
    // WANT & 2: is some R <= 0 (`nonpos`) -- what the linear scan below needs to know when a row block is finished.
    auto evaluate = [&](auto want_tag, int thr, const bool (&here)[3], unsigned& pos, unsigned& nonpos) {
        constexpr int WANT = decltype(want_tag)::value;
        const uint32_t T = (uint32_t)(thr + 1) * 0x01010101u;       // t + 1 in every byte
        const int C = thr + 1;
        v4i ind[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int d = 0; d < 4; d++)
                ind[i][d] = (int)(__builtin_amdgcn_lerp((uint32_t)nx[i][d], T, 0u) & 0x80808080u);   // -128 where x <= t
#pragma unroll
        for (int t = 0; t < 3; t++) {
            if (!here[t]) continue;                                  // wave-uniform
            v4i c1v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) c1v[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ind[i], bv[t], zero, 0, 0, 0);
            v4i r[3];
#pragma unroll
            for (int u = 0; u < 3; u++) r[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(c1v[u], bh[0], c113, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 3; u++) r[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(c1v[u + 1], bh[1], r[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 3; u++)
#pragma unroll
                for (int e = 0; e < 4; e++) med[t][u][e] = imed3(med[t][u][e], r[u][e], C);
            if constexpr ((WANT & 1) != 0) {
                int m = r[0][0];
#pragma unroll
                for (int u = 0; u < 3; u++)
#pragma unroll
                    for (int e = 0; e < 4; e++) m = m > r[u][e] ? m : r[u][e];
                if (__builtin_amdgcn_ballot_w64(m > 0) != 0) pos |= 1u << t;
            }
            if constexpr ((WANT & 2) != 0) {
                int m = r[0][0];
#pragma unroll
                for (int u = 0; u < 3; u++)
#pragma unroll
                    for (int e = 0; e < 4; e++) m = m < r[u][e] ? m : r[u][e];
                if (__builtin_amdgcn_ballot_w64(m <= 0) != 0) nonpos |= 1u << t;
            }
        }
    };

    // ---- flat tiles: a linear scan around a guessed level instead of the radix descent (round 4) ------------------
    // The radix tree pays one box count per LEVEL even where a whole row block holds a single median (8 for 1 value),
    // and a contiguous run of medians that straddles a power of two pays for the alignment as well: 10.7 box counts per
    // row block on the flat 72 % of a board frame's tiles, which hold 4.3 distinct medians.  Where the tile's input is
    // flat (256 sampled pixels span <= 28 levels: sensor noise around one colour, no stone, no edge), the medians hug the
    // sample mean g, so the thresholds g, g + 1, ... are evaluated until no pixel of a row block has its median above
    // the threshold, then g - 1, g - 2, ... until none has it at or below: every t from (lowest median - 1) to the highest
    // median is seen, which pins every pixel exactly (a pixel of median m needs m - 1 and m) -- span + 1 box counts per
    // row block, the lower bound for a contiguous run.  The update is the same med3: with R > 0 it is max(med, t + 1),
    // with R <= 0 min(med, t + 1), and both leave a finished pixel where it is in either scan direction.  A tile that
    // turns out not to be flat (a scan longer than SCAN_CAP steps) starts over with the radix descent; nothing is
    // assumed about the guess, it only has to be cheap.  -20 % box counts per 1080p board frame (simulation against the
    // oracle's medians: profiles/r04_median_scan.txt), other content classes take the radix path as before.
#ifndef MED_SCAN_RANGE
#define MED_SCAN_RANGE 28
#endif
#ifndef MED_SCAN_CAP
#define MED_SCAN_CAP 12
#endif
    constexpr int SCAN_RANGE = MED_SCAN_RANGE, SCAN_CAP = MED_SCAN_CAP;
    bool done = false;
    int lo_bound = 0, hi_bound = 255;       // lo_bound <= every median of the tile <= hi_bound (-> range)
    MED_COUNT(0, 1);
    // The scan: thresholds g0, g0 + 1, ... until no row block has a median above, then g0 - 1, g0 - 2, ... until none has one
    // at or below.  It works on whatever the radix descent has decided so far (med = a pixel's prefix <= its median <= ...):
    // every update is max(med, t + 1) where the median is above t and min(med, t + 1) where it is not, and a prefix is never
    // above its median.  cap: steps per direction before giving up (< 0: none).  -> false when it gave up.
    auto linear_scan = [&](int g0, int cap) {
        unsigned up = 7u, down = 0u;                                 // row blocks still scanning in each direction
        int steps = 0;
        int thr = g0;
        for (; up && thr <= 254; thr++, steps++) {
            if (steps == cap) return false;
            const bool here[3] = {(up & 1u) != 0, (up & 2u) != 0, (up & 4u) != 0};
            MED_COUNT(4, __builtin_popcount(up));
            unsigned pos = 0, nonpos = 0;
            if (thr == g0) evaluate(std::integral_constant<int, 3>{}, thr, here, pos, nonpos);
            else evaluate(std::integral_constant<int, 1>{}, thr, here, pos, nonpos);
            if (thr == g0) down = nonpos;                            // some median <= g: g - 1 has to be looked at
            up = pos;
        }
        // the upward scan ended at thr - 1 with no median above it (or ran into 255)
        const int top = up ? 255 : thr - 1;
        steps = 0;
        for (thr = g0 - 1; down && thr >= 0; thr--, steps++) {
            if (steps == cap) return false;
            const bool here[3] = {(down & 1u) != 0, (down & 2u) != 0, (down & 4u) != 0};
            MED_COUNT(4, __builtin_popcount(down));
            unsigned pos = 0, nonpos = 0;
            evaluate(std::integral_constant<int, 2>{}, thr, here, pos, nonpos);
            down = nonpos;
        }
        // the downward scan ended at thr + 1 with no median at or below it (never started: none <= g0 = thr + 1)
        hi_bound = top;
        lo_bound = down ? 0 : thr + 2;
        return true;
    };
    int g_mean;                             // the level 256 sampled pixels of the tile average to (the scans' first guess)
    {
        // 4 samples per lane: rows 16 g + {1, 6, 9, 14} of columns n, 16 + n, 32 + n, 48 + n (inverted bytes)
        const uint32_t s0 = ((uint32_t)nx[0][0] >> 8) & 0xFFu, s1 = ((uint32_t)nx[1][1] >> 16) & 0xFFu;
        const uint32_t s2 = ((uint32_t)nx[2][2] >> 8) & 0xFFu, s3 = ((uint32_t)nx[3][3] >> 16) & 0xFFu;
        uint32_t lo = s0 < s1 ? s0 : s1, hi = s0 < s1 ? s1 : s0;
        lo = lo < s2 ? lo : s2; lo = lo < s3 ? lo : s3;
        hi = hi > s2 ? hi : s2; hi = hi > s3 ? hi : s3;
        uint32_t sum = s0 + s1 + s2 + s3;
        // row of 16 lanes: butterflies by DPP; the four rows meet on the scalar unit
        // (each DPP read goes through a temporary: `V = V < dpp(V) ? V : dpp(V)` with __builtin_amdgcn_mov_dpp written
        // twice compiles to a minimum that is 0 -- tools/micro/dpp_reduce.hip)
#define CK_DPP(V, CTRL) (uint32_t)__builtin_amdgcn_update_dpp((int)(V), (int)(V), CTRL, 0xF, 0xF, false)
#define CK_ROW_REDUCE(V, OP)                                                                                         \
        { uint32_t t_;                                                                                               \
          t_ = CK_DPP(V, 0xB1); V = OP(V, t_);     /* quad_perm [1,0,3,2] */                                         \
          t_ = CK_DPP(V, 0x4E); V = OP(V, t_);     /* quad_perm [2,3,0,1] */                                         \
          t_ = CK_DPP(V, 0x141); V = OP(V, t_);    /* row_half_mirror */                                             \
          t_ = CK_DPP(V, 0x140); V = OP(V, t_); }  /* row_mirror */
#define CK_MIN(a, b) ((a) < (b) ? (a) : (b))
#define CK_MAX(a, b) ((a) > (b) ? (a) : (b))
#define CK_ADD(a, b) ((a) + (b))
        CK_ROW_REDUCE(lo, CK_MIN)
        CK_ROW_REDUCE(hi, CK_MAX)
        CK_ROW_REDUCE(sum, CK_ADD)
#undef CK_ROW_REDUCE
#undef CK_DPP
        uint32_t wlo = 255u, whi = 0u, wsum = 0u;
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)lo, 16 * rr), bq = (uint32_t)__builtin_amdgcn_readlane((int)hi, 16 * rr);
            wlo = CK_MIN(wlo, a); whi = CK_MAX(whi, bq);
            wsum += (uint32_t)__builtin_amdgcn_readlane((int)sum, 16 * rr);
        }
#undef CK_MIN
#undef CK_MAX
#undef CK_ADD
        g_mean = 255 - (int)((wsum + 128u) >> 8);                    // the samples are inverted pixels: mean of 256, rounded
        g_mean = g_mean < 0 ? 0 : (g_mean > 254 ? 254 : g_mean);
        if ((int)(whi - wlo) <= SCAN_RANGE) {
            MED_COUNT(1, 1);
            if (linear_scan(g_mean, SCAN_CAP)) done = true;
            else {
                MED_COUNT(2, 1);
                lo_bound = 0; hi_bound = 255;
#pragma unroll
                for (int t = 0; t < 3; t++)
#pragma unroll
                    for (int u = 0; u < 3; u++)
#pragma unroll
                        for (int e = 0; e < 4; e++) med[t][u][e] = 0;
            }
        }
    }

    // The prefixes alive at a level are kept PER ROW BLOCK of 16 output rows (bit l of ct[t][k] <-> prefix 4 l + k is held
    // by some pixel of rows 16 t .. 16 t + 15; all wave-uniform, i.e. scalar registers): a threshold is evaluated only in
    // the row blocks that hold its prefix -- a block without it would be left unchanged by the arithmetic anyway -- which
    // drops 18 % of the (threshold, row block) box filters on board scenes: their 10 MFMAs and 12 updates each.
    unsigned long long ct[3][4];
#pragma unroll
    for (int t = 0; t < 3; t++) { ct[t][0] = 1ull; ct[t][1] = ct[t][2] = ct[t][3] = 0ull; }
    for (int b = done ? -1 : 7; b >= 0; b--) {
        const int half = 1 << b;
        unsigned long long c0 = ct[0][0] | ct[1][0] | ct[2][0], c1 = ct[0][1] | ct[1][1] | ct[2][1];
        unsigned long long c2 = ct[0][2] | ct[1][2] | ct[2][2], c3 = ct[0][3] | ct[1][3] | ct[2][3];
        for (;;) {
            int q;
            bool here[3];
#define CK_TAKE(K, CK)                                                                                   \
            { const int l = __builtin_ctzll(CK); q = 4 * l + K; CK &= CK - 1;                            \
              here[0] = (ct[0][K] >> l) & 1; here[1] = (ct[1][K] >> l) & 1; here[2] = (ct[2][K] >> l) & 1; }
            if (c0) CK_TAKE(0, c0)
            else if (c1) CK_TAKE(1, c1)
            else if (c2) CK_TAKE(2, c2)
            else if (c3) CK_TAKE(3, c3)
            else break;
#undef CK_TAKE
            unsigned pos = 0, nonpos = 0;
            MED_COUNT(3, (int)here[0] + (int)here[1] + (int)here[2]);
            evaluate(std::integral_constant<int, 0>{}, q + half - 1, here, pos, nonpos);
        }
        if (b == 0) break;
        // the distinct prefixes of each row block -> its set for the next level
#pragma unroll
        for (int t = 0; t < 3; t++) {
            uint8_t* fb = reinterpret_cast<uint8_t*>(flags[t]);
#pragma unroll
            for (int u = 0; u < 3; u++)
#pragma unroll
                for (int e = 0; e < 4; e++) fb[med[t][u][e]] = 1;
        }
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const uint32_t fw = flags[t][lane];
            flags[t][lane] = 0;
            ct[t][0] = __builtin_amdgcn_ballot_w64((fw & 0xFFu) != 0);
            ct[t][1] = __builtin_amdgcn_ballot_w64((fw & 0xFF00u) != 0);
            ct[t][2] = __builtin_amdgcn_ballot_w64((fw & 0xFF0000u) != 0);
            ct[t][3] = __builtin_amdgcn_ballot_w64((fw & 0xFF000000u) != 0);
        }
        // ---- hybrid (round 6, MEASURED AND LEFT OFF: MED_HYBRID_RUN 0): after the top levels the surviving prefixes are buckets
        // of 2^b values; where the whole tile's buckets form ONE short run, finish with a linear scan of that range instead of
        // the rest of the tree.  It fires where the INPUT is not flat but the medians are (textured and noisy content), and buys
        // little: a dense span of S medians costs the tree S + 8 box counts per row block (S / 2^(b+1) + 1 prefixes per level)
        // and the scan S + 2 plus the distance of its first guess from the range, which after four levels is only known to 16.
        // Same box (docs/lab_notes.md, profiles/r06_k1_variants.txt): noise of sigma 10 -17 %, the 1/f table texture -2.7 %,
        // the bench scene +3 % (a quarter of its tiles take it and mostly guess badly: their input mean is a stone's, not
        // the wood's).  Exact either way (tools/k1_content.py compares every class with a sort-based median).
#ifndef MED_HYBRID_LEVEL
#define MED_HYBRID_LEVEL 4
#endif
#ifndef MED_HYBRID_RUN
#define MED_HYBRID_RUN 0
#endif
        if constexpr (MED_HYBRID_RUN > 0) if (b == MED_HYBRID_LEVEL) {
            // after level b the prefixes are multiples of 2^b: all of them in byte 0 of their flag word when b >= 2
            static_assert(MED_HYBRID_LEVEL >= 2 && MED_HYBRID_LEVEL <= 6, "hybrid level");
            const unsigned long long U = ct[0][0] | ct[1][0] | ct[2][0];
            const int first = __builtin_ctzll(U) << 2, last = (63 - __builtin_clzll(U)) << 2;      // lowest / highest prefix alive
            const int buckets = ((last - first) >> b) + 1;
            if (buckets <= MED_HYBRID_RUN && __builtin_popcountll(U) == buckets) {
                MED_COUNT(5, 1);
                int g0 = g_mean < first ? first : g_mean;
                const int top_level = last + (1 << b) - 1;
                g0 = g0 > top_level ? top_level : g0;
                g0 = g0 > 254 ? 254 : g0;
                (void)linear_scan(g0, -1);
                break;
            }
        }
    }
    if (range && lane == 0) {
        uint8_t* rp = range + ((size_t)((f * 3 + c) * gridDim.y + byi) * gridDim.x + bxi) * 2;
        *reinterpret_cast<uint16_t*>(rp) = (uint16_t)(lo_bound | (hi_bound << 8));
    }

    // ---- store: planar; lane (n, g) holds columns 16 t + 4 g + e of row 16 u + n: one aligned dword per block
    uint8_t* plane = out + ((size_t)(f * 3 + c) * h) * pitch;
#pragma unroll
    for (int u = 0; u < 3; u++) {
        const int y = oy + 16 * u + n;
        if (y >= h) continue;
        uint8_t* row = plane + (size_t)y * pitch;
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int x = ox + 16 * t + 4 * g;
            const uint32_t word = (uint32_t)med[t][u][0] | ((uint32_t)med[t][u][1] << 8) | ((uint32_t)med[t][u][2] << 16) |
                                  ((uint32_t)med[t][u][3] << 24);
            if (x + 3 < pitch) *reinterpret_cast<uint32_t*>(row + x) = word;       // (columns w .. pitch - 1 are padding)
            else {
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (x + e < w) row[x + e] = (uint8_t)med[t][u][e];
            }
        }
    }
}

__global__ void planar_to_interleaved_kernel(const uint8_t* __restrict__ planes, int h, int w, int pitch,
                                             uint8_t* __restrict__ out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int f = blockIdx.z;
    if (x >= w) return;
    const uint8_t* p = planes + (size_t)f * 3 * h * pitch + (size_t)y * pitch + x;
    uint8_t* o = out + (((size_t)f * h + y) * w + x) * 3;
    o[0] = p[0];
    o[1] = p[(size_t)h * pitch];
    o[2] = p[(size_t)2 * h * pitch];
}

__global__ void interleaved_to_planar_kernel(const uint8_t* __restrict__ in, int h, int w, int pitch,
                                             uint8_t* __restrict__ planes)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int f = blockIdx.z;
    if (x >= w) return;
    const uint8_t* i = in + (((size_t)f * h + y) * w + x) * 3;
    uint8_t* p = planes + (size_t)f * 3 * h * pitch + (size_t)y * pitch + x;
    p[0] = i[0];
    p[(size_t)h * pitch] = i[1];
    p[(size_t)2 * h * pitch] = i[2];
}

}  // namespace

int k_median_planar(ck_ctx* ctx, const uint8_t* d_bgr, int n, int h, int w, int ksize, uint8_t* d_planes, int pitch, uint8_t* d_range)
{
    TimeScope ts(ctx, "median");
#if !CK_TILE_RANGE
    d_range = nullptr;
#endif
    dim3 grid((w + MT - 1) / MT, (h + MT - 1) / MT, n * 3);
#define CK_MEDIAN_CASE(KS) case KS: hipLaunchKernelGGL(median_mfma_kernel<KS>, grid, dim3(64), 0, ctx->stream, d_bgr, h, w, d_planes, pitch, d_range); break;
    switch (ksize) {
        CK_MEDIAN_CASE(3) CK_MEDIAN_CASE(5) CK_MEDIAN_CASE(7) CK_MEDIAN_CASE(9) CK_MEDIAN_CASE(11) CK_MEDIAN_CASE(13)
        CK_MEDIAN_CASE(15) CK_MEDIAN_CASE(17)
    default: return ck_fail(ctx, CK_ERR_ARG, "median window %d: odd sizes 3..17 only", ksize);
    }
#undef CK_MEDIAN_CASE
#if MED_DBG
    {
        unsigned long long c[8];
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(g_med_dbg), sizeof c);
        fprintf(stderr, "[median dbg] tiles %llu  scanned %llu  given up %llu  hybrid %llu  box counts per tile: radix %.2f  scan %.2f\n", c[0], c[1], c[2], c[5],
                (double)c[3] / (double)(c[0] ? c[0] : 1), (double)c[4] / (double)(c[0] ? c[0] : 1));
        memset(c, 0, sizeof c);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_med_dbg), c, sizeof c);
    }
#endif
    CK_HIP(ctx, hipGetLastError());
    return CK_OK;
}

int k_median15_planar(ck_ctx* ctx, const uint8_t* d_bgr, int n, int h, int w, uint8_t* d_planes, int pitch, uint8_t* d_range)
{
    return k_median_planar(ctx, d_bgr, n, h, w, 15, d_planes, pitch, d_range);
}

int k_planar_to_interleaved(ck_ctx* ctx, const uint8_t* d_planes, int n, int h, int w, int pitch, uint8_t* d_out)
{
    TimeScope ts(ctx, "repack");
    dim3 grid((w + 255) / 256, h, n);
    hipLaunchKernelGGL(planar_to_interleaved_kernel, grid, dim3(256), 0, ctx->stream, d_planes, h, w, pitch, d_out);
    CK_HIP(ctx, hipGetLastError());
    return CK_OK;
}

int k_interleaved_to_planar(ck_ctx* ctx, const uint8_t* d_in, int n, int h, int w, int pitch, uint8_t* d_planes)
{
    TimeScope ts(ctx, "repack");
    dim3 grid((w + 255) / 256, h, n);
    hipLaunchKernelGGL(interleaved_to_planar_kernel, grid, dim3(256), 0, ctx->stream, d_in, h, w, pitch, d_planes);
    CK_HIP(ctx, hipGetLastError());
    return CK_OK;
}
