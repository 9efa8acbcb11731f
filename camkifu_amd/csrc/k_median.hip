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
// Two kernels evaluate the box sums.  median15_mfma_kernel (further down, the one launched) does them on the
// i8 matrix cores: 20 us per 1080p frame.  median15_kernel (first below, kept as the all-VALU reference of
// the same descent; build with -DMED_MFMA=0) does them with SWAR adds and DPP shifts: 38 us.
//
// median15_kernel mapping: one wave = one 48x64 output tile of one channel.  64 lanes = 16 (x) x 4 (y);
// a lane owns 4 adjacent pixels packed in one dword (SWAR, byte lanes) and 16 output rows,
// so the vertical pass is in-register and the horizontal pass crosses at most 4 lanes of
// the same 16-lane DPP row.  Lanes 12..15 of each row only supply halo.
//
// HBM traffic: each input byte is read ~(64*78)/(48*64) = 1.6x (L2 absorbs the halo),
// output written once, planar [n][3][h][pitch] so the next stage reads dwords.
//
// The SWAR kernel is VALU-issue bound (every instruction below costs one 4-cycle wave64 slot, v_mad_u64_u32
// two: tools/micro/valu_rates.hip), so its time is (#box filters per tile) x (376 instructions).
// Costed against the distinct-prefix counts of real median images (1080p board scenes, ~20.6 box
// filters per 48x64 tile, lower bound ~12.7 from the distinct medians):
//   * tile shapes 48x16 .. 48x128 and full-wave rows 240x8 .. 240x32: 48x48 / 48x64 are the minimum
//     (smaller tiles pay the 14-row / 14-column halo, larger ones hold more distinct medians);
//   * 8 pixels per lane (112-wide tiles, better lane use, fewer DPP shifts): the tile then holds
//     ~23-25 prefixes and the registers drop the occupancy -- a wash;
//   * skipping thresholds outside the tile's input value range, or a sampled-range + linear scan:
//     no gain on board scenes (the input range of a tile is wide; flat tiles are already cheap);
//   * a second copy of the threshold body for levels with a single prefix (the per-row prefix test
//     drops out, -3 of 23 ops per row on ~35 % of the thresholds): 138 VGPRs -> 3 waves/SIMD, or 9
//     spills when held to 4; measured 40.1 / 38.8 us against 38.3 us for the single body.
#include "ck_common.h"

#ifndef MED_MFMA
#define MED_MFMA 1
#endif
namespace {

constexpr int HOUT = 16;              // output rows per lane
constexpr int HIN = HOUT + 14;        // input rows per lane
constexpr int TILE_W = 48;            // valid output columns per wave
constexpr int TILE_H = 4 * HOUT;      // output rows per wave

template <int K>
__device__ __forceinline__ uint32_t lane_right(uint32_t v)
{
    // value held by the lane K places to the right inside the 16-lane DPP row (row_shl:K);
    // lanes shifted in from beyond the row read 0.  A VALU move, no LDS crossbar traffic.
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + K, 0xf, 0xf, true);
}

__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t nbytes)
{
    return __builtin_amdgcn_alignbyte(hi, lo, nbytes);
}

// 15-wide horizontal box sum of packed byte counters: result byte t of lane lx is
// sum_{k=0..14} V[4*lx + t + k].  Per lane: PS = inclusive byte prefix sums of its word,
// G = the word total in every byte; the window is (suffix of this word) + (two full words)
// + (a 3+t pixel prefix spanning the words 3 and 4 lanes to the right): 4 row shifts.
__device__ __forceinline__ uint32_t hsum15(uint32_t w)
{
    const uint32_t a = w + (w << 8);
    const uint32_t PS = a + (a << 16);
    const uint32_t G = __builtin_amdgcn_perm(PS, PS, 0x03030303u);
    const uint32_t SS = G - (PS << 8);                                  // suffix sums
    const uint32_t PS3 = lane_right<3>(PS), PS4 = lane_right<4>(PS);
    const uint32_t tail = alignbyte(PS4, PS3, 2) + __builtin_amdgcn_perm(PS3, PS3, 0x03030C0Cu);
    return SS + lane_right<1>(G) + lane_right<2>(G) + tail;
}

struct Set256 {
    unsigned long long w[4];
    __device__ __forceinline__ void clear() { w[0] = w[1] = w[2] = w[3] = 0; }
    __device__ __forceinline__ void set(int v)
    {
        unsigned long long bit = 1ull << (v & 63);
        switch (v >> 6) {
        case 0: w[0] |= bit; break;
        case 1: w[1] |= bit; break;
        case 2: w[2] |= bit; break;
        default: w[3] |= bit; break;
        }
    }
};

__global__ __launch_bounds__(64) void median15_kernel(const uint8_t* __restrict__ in, int h, int w,
                                                      uint8_t* __restrict__ out, int pitch)
{
    const int lane = threadIdx.x;
    const int lx = lane & 15, ly = lane >> 4;
    const int ox = blockIdx.x * TILE_W, oy = blockIdx.y * TILE_H;
    const int f = blockIdx.z / 3, c = blockIdx.z % 3;
    const uint8_t* src = in + (size_t)f * h * w * 3 + c;

    // ---- load the lane's 4 x HIN pixels (replicate border = clamped coordinates) --------
    uint32_t nx[HIN];
    int xo[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int x = ox - 7 + 4 * lx + k;
        x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
        xo[k] = x * 3;
    }
    const int ybase = oy + HOUT * ly - 7;
#pragma unroll
    for (int r = 0; r < HIN; r++) {
        int y = ybase + r;
        y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
        const uint8_t* row = src + (size_t)y * w * 3;
        uint32_t v = (uint32_t)row[xo[0]] | ((uint32_t)row[xo[1]] << 8) |
                     ((uint32_t)row[xo[2]] << 16) | ((uint32_t)row[xo[3]] << 24);
        nx[r] = ~v;
    }

    uint32_t lo[HOUT];
#pragma unroll
    for (int j = 0; j < HOUT; j++) lo[j] = 0;
    const uint32_t vm80 = (lx < 12) ? 0x80808080u : 0u;   // lanes 12..15 have no full window

    Set256 cur, nxt;
    cur.clear();
    cur.set(0);
    for (int b = 7; b >= 0; b--) {
        nxt.clear();
        const int half = 1 << b;
        for (;;) {
            int q;
            if (cur.w[0]) { q = __builtin_ctzll(cur.w[0]); cur.w[0] &= cur.w[0] - 1; }
            else if (cur.w[1]) { q = 64 + __builtin_ctzll(cur.w[1]); cur.w[1] &= cur.w[1] - 1; }
            else if (cur.w[2]) { q = 128 + __builtin_ctzll(cur.w[2]); cur.w[2] &= cur.w[2] - 1; }
            else if (cur.w[3]) { q = 192 + __builtin_ctzll(cur.w[3]); cur.w[3] &= cur.w[3] - 1; }
            else break;
            {
                const uint32_t T = (uint32_t)(q + half) * 0x01010101u;   // t + 1 in every byte
                const uint32_t Q = (uint32_t)q * 0x01010101u;
                uint32_t B[HIN];
#pragma unroll
                for (int r = 0; r < HIN; r++)
                    B[r] = (__builtin_amdgcn_lerp(nx[r], T, 0u) >> 7) & 0x01010101u;   // x <= t
                uint32_t V = 0;
#pragma unroll
                for (int r = 0; r < 15; r++) V += B[r];
                uint32_t any_hi = 0, any_lo = 0;
#pragma unroll
                for (int j = 0; j < HOUT; j++) {
                    if (j > 0) V = V + B[j + 14] - B[j - 1];
                    const uint32_t S = hsum15(V);
                    const uint32_t nD = S + 0x0F0F0F0Fu;                          // bit7 set <=> S >= 113
                    const uint32_t nz = __builtin_amdgcn_lerp(lo[j] ^ Q, 0xFFFFFFFFu, 0u);   // bit7 set <=> prefix != q
                    const uint32_t eq7 = (~nz & vm80);      // ~nz & vm80
                    const uint32_t upd7 = (~nD & eq7);      // eq & (S < 113): median above t
                    lo[j] += upd7 >> (7 - b);
                    any_hi |= upd7;
                    any_lo |= eq7 & nD;
                }
                if (__builtin_amdgcn_ballot_w64(any_hi != 0)) nxt.set(q + half);
                if (__builtin_amdgcn_ballot_w64(any_lo != 0)) nxt.set(q);
            }
        }
        cur = nxt;
    }

    // ---- store: planar, one dword per lane-row ------------------------------------------
    if (lx < 12) {
        const int x = ox + 4 * lx;
        if (x < w) {
            uint8_t* dst = out + ((size_t)(f * 3 + c) * h) * pitch + x;
#pragma unroll
            for (int j = 0; j < HOUT; j++) {
                const int y = oy + HOUT * ly + j;
                if (y < h) *reinterpret_cast<uint32_t*>(dst + (size_t)y * pitch) = lo[j];
            }
        }
    }
}

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
    const uint8_t* __restrict__ in, int h, int w, uint8_t* __restrict__ out, int pitch)
{
    static_assert(K % 2 == 1 && K >= 3 && MT + K - 1 <= 64, "window size");
    constexpr int HK = K / 2, RANK = (K * K + 1) / 2;
    __shared__ uint32_t flags[3][64];
    __shared__ uint32_t otile[MT * MT / 4];                    // the finished tile, for dword stores
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
    const uint8_t* src = in + (size_t)f * h * w * 3 + c;

    // ---- load: column tile i, lane (n, g) <- column ox - K/2 + 16 i + n, rows oy - K/2 + 16 g + 0..15 (replicate border)
    v4i nx[4];
    if (ox >= HK && ox - HK + 63 < w && oy >= HK && oy - HK + 63 < h) {
        // interior tile: one per-lane offset, everything else is wave-uniform (scalar base + immediate)
        const uint8_t* base = src + ((size_t)(oy - HK) * w + (ox - HK)) * 3;
        uint32_t voff = (uint32_t)((16 * g * w + n) * 3);
        asm volatile("" : "+v"(voff));          // keep it one per-lane offset: scalar base + voff + immediate per load
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint8_t* r0 = base + (size_t)(uint32_t)((4 * d) * w * 3);
            const uint8_t* r1 = base + (size_t)(uint32_t)((4 * d + 1) * w * 3);
            const uint8_t* r2 = base + (size_t)(uint32_t)((4 * d + 2) * w * 3);
            const uint8_t* r3 = base + (size_t)(uint32_t)((4 * d + 3) * w * 3);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t v = (uint32_t)(r0 + 48 * i)[voff] | ((uint32_t)(r1 + 48 * i)[voff] << 8) |
                                   ((uint32_t)(r2 + 48 * i)[voff] << 16) | ((uint32_t)(r3 + 48 * i)[voff] << 24);
                nx[i][d] = (int)~v;
            }
        }
    } else {
        int yo[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            int y = oy - HK + 16 * g + j;
            y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
            yo[j] = y * w * 3;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int x = ox - HK + 16 * i + n;
            x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
            const uint8_t* col = src + x * 3;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const uint32_t v = (uint32_t)col[yo[4 * d]] | ((uint32_t)col[yo[4 * d + 1]] << 8) |
                                   ((uint32_t)col[yo[4 * d + 2]] << 16) | ((uint32_t)col[yo[4 * d + 3]] << 24);
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

    // The prefixes alive at a level are kept PER ROW BLOCK of 16 output rows (bit l of ct[t][k] <-> prefix 4 l + k is held
    // by some pixel of rows 16 t .. 16 t + 15; all wave-uniform, i.e. scalar registers): a threshold is evaluated only in
    // the row blocks that hold its prefix -- a block without it would be left unchanged by the arithmetic anyway -- which
    // drops 18 % of the (threshold, row block) box filters on board scenes: their 10 MFMAs and 12 updates each.
    unsigned long long ct[3][4];
#pragma unroll
    for (int t = 0; t < 3; t++) { ct[t][0] = 1ull; ct[t][1] = ct[t][2] = ct[t][3] = 0ull; }
    for (int b = 7; b >= 0; b--) {
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
            const uint32_t T = (uint32_t)(q + half) * 0x01010101u;      // t + 1 in every byte
            const int C = q + half;
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
            }
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
    }

    // ---- store: planar; lane (n, g) holds rows 16 t + 4 g + e of column 16 u + n
    uint8_t* plane = out + ((size_t)(f * 3 + c) * h) * pitch;
    if (ox + MT <= w && oy + MT <= h) {
        // whole tile inside the image: through LDS, so that HBM sees dwords (12 per tile row) instead of byte stores --
        // the byte form wrote 8.1 MB per 1080p frame for 6.2 MB of output (PMC WRITE_SIZE, partial-line writes)
        uint8_t* tb = reinterpret_cast<uint8_t*>(otile);
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int u = 0; u < 3; u++) tb[(16 * t + 4 * g + e) * MT + 16 * u + n] = (uint8_t)med[t][u][e];
        // one wave = the whole workgroup: its own LDS writes are visible to it after the counter wait
        __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        if (lane < 60) {                                        // 5 tile rows of 12 dwords per round, 10 rounds (48 rows)
            const int r0 = lane / 12, c4 = lane % 12;
            uint8_t* dst = plane + (size_t)(oy + r0) * pitch + ox + 4 * c4;
#pragma unroll
            for (int it = 0; it < 10; it++) {
                if (5 * it + r0 < MT)
                    *reinterpret_cast<uint32_t*>(dst + (size_t)(5 * it) * pitch) = otile[(5 * it + r0) * (MT / 4) + c4];
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int x = ox + 16 * u + n;
            if (x < w) {
#pragma unroll
                for (int t = 0; t < 3; t++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int y = oy + 16 * t + 4 * g + e;
                        if (y < h) plane[(size_t)y * pitch + x] = (uint8_t)med[t][u][e];
                    }
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

int k_median_planar(ck_ctx* ctx, const uint8_t* d_bgr, int n, int h, int w, int ksize, uint8_t* d_planes, int pitch)
{
    TimeScope ts(ctx, "median");
#if MED_MFMA
    dim3 grid((w + MT - 1) / MT, (h + MT - 1) / MT, n * 3);
#define CK_MEDIAN_CASE(KS) case KS: hipLaunchKernelGGL(median_mfma_kernel<KS>, grid, dim3(64), 0, ctx->stream, d_bgr, h, w, d_planes, pitch); break;
    switch (ksize) {
        CK_MEDIAN_CASE(3) CK_MEDIAN_CASE(5) CK_MEDIAN_CASE(7) CK_MEDIAN_CASE(9) CK_MEDIAN_CASE(11) CK_MEDIAN_CASE(13)
        CK_MEDIAN_CASE(15) CK_MEDIAN_CASE(17)
    default: return ck_fail(ctx, CK_ERR_ARG, "median window %d: odd sizes 3..17 only", ksize);
    }
#undef CK_MEDIAN_CASE
#else
    if (ksize != 15) return ck_fail(ctx, CK_ERR_ARG, "the SWAR median kernel is 15x15 only");
    dim3 grid((w + TILE_W - 1) / TILE_W, (h + TILE_H - 1) / TILE_H, n * 3);
    hipLaunchKernelGGL(median15_kernel, grid, dim3(64), 0, ctx->stream, d_bgr, h, w, d_planes, pitch);
#endif
    CK_HIP(ctx, hipGetLastError());
    return CK_OK;
}

int k_median15_planar(ck_ctx* ctx, const uint8_t* d_bgr, int n, int h, int w, uint8_t* d_planes, int pitch)
{
    return k_median_planar(ctx, d_bgr, n, h, w, 15, d_planes, pitch);
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
