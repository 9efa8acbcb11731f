// k_canny.hip -- K2: cv2.Canny(median, 25, 75) on the 3-channel median image
// (reference: src/camkifu/board/bf_auto.py:73; aperture 3, L1 gradient).
//
//   nms kernel   : LDS-staged 3-channel tile with halo -> Sobel dx/dy per channel (replicate
//                  border; on the fp16 matrix pipe as two band-matrix products, or in packed 16-bit
//                  vector math) -> channel with the largest |dx|+|dy| -> non-maximum suppression
//                  with the fixed-point tan(22.5) sector test -> map {0 cand, 1 no, 2 strong}; the
//                  NMS kernel also clears the edge image.  Nominally HBM-bound (reads 3 B/px,
//                  writes 1 B/px + 4 B/px label init for candidates), in practice instruction issue.
//   hysteresis   : the serial stack flood of the CPU algorithm becomes an 8-connected
//                  union-find over candidate pixels; a component is an edge iff it holds a
//                  strong pixel.  Same result set, no iteration-until-stable loop.
#include "ck_common.h"
#include "ck_uf.h"

// (The gradient phase on the fp16 matrix pipe -- canny_nms_mfma_kernel, built and measured in round 4: 8.1 us per 1080p frame
// against this file's 5.5 -- is archived in tools/variants/canny_nms_mfma.hip.txt.)

#ifndef NMS_LOCAL_UF
#define NMS_LOCAL_UF 1     // tile-local union-find of the candidates inside the NMS kernel
#endif

namespace {

constexpr int TW = 64;


// ------------------------------------------------------------------------------------------
// Packed kernel.  The
// kernel is VALU-bound, so what counts is instructions per pixel (round 2: ~105 lane-ops per pixel, 60 of them in the
// gradient phase; this version: see DESIGN.md 4):
//   * gradient phase in packed 16-bit math (v_pk_*): a thread owns 4 adjacent columns as two u16 pairs and walks PK
//     gradient rows down its PK + 2 pixel rows (the horizontal pass is amortised over PK rows); one v_perm per tap pair
//     widens the bytes, `2 a + b` is one v_pk_mad.  What it keeps per pixel is ONE 16-bit key = magnitude * 4 + (3 -
//     channel): the maximum over the three keys is the largest |dx| + |dy| and, on ties, the first channel -- no dx / dy
//     selects, no dx / dy arrays in LDS.
//   * NMS phase: a quad of pixels with no key above the low threshold (most of a median-filtered frame) costs one
//     packed max, two compares and the map store.  A candidate re-derives dx, dy of its channel from the 8 taps still in
//     the pixel tile, then runs the two fixed-point tangent tests.
// Tile: 64 x (15 PK - 2) output pixels per 256-thread workgroup (17 column quads x 15 row segments of PK gradient rows).
#ifndef NMS_PK
#define NMS_PK 2
#endif
#ifndef NMS_STOP
#define NMS_STOP 0         // profiling aid (tools/pmc_serial.sh): leave the kernel after phase n -- results are then WRONG
#endif
constexpr int RANGE_TILE = CK_RANGE_TILE;    // edge of the tiles the value bounds of k_median_planar refer to
constexpr int PK = NMS_PK;                   // gradient rows per thread
constexpr int PGR = 15 * PK;                 // gradient rows (1-px halo)
constexpr int PTH = PGR - 2;                 // output rows per tile
constexpr int PLH = PTH + 4;                 // pixel rows (2-px halo)
constexpr int PGW = 68;                      // gradient columns held: x = ox - 2 .. ox + 65
constexpr int PLWD = 20;                     // pixel tile dwords per row: x = ox - 8 .. ox + 71, staged with 16-byte loads
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u16x2 widen(uint32_t hi, uint32_t lo, uint32_t sel)
{
    return __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(hi, lo, sel));
}
// 2 a + c on both halves in one instruction (the compiler turns the multiplication into a shift and adds separately)
__device__ __forceinline__ u16x2 mad2(u16x2 a, u16x2 c)
{
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, 2, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(__builtin_bit_cast(uint32_t, a)), "v"(__builtin_bit_cast(uint32_t, c)));
    return __builtin_bit_cast(u16x2, r);
}
__device__ __forceinline__ s16x2 mad2(s16x2 a, s16x2 c)
{
    uint32_t r;
    asm("v_pk_mad_i16 %0, %1, 2, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(__builtin_bit_cast(uint32_t, a)), "v"(__builtin_bit_cast(uint32_t, c)));
    return __builtin_bit_cast(s16x2, r);
}
// 4 m + TAG on both halves
template <int TAG>
__device__ __forceinline__ u16x2 key_of(u16x2 m)
{
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, 4, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(__builtin_bit_cast(uint32_t, m)), "n"(TAG));
    return __builtin_bit_cast(u16x2, r);
}

__global__ __launch_bounds__(256) void canny_nms_packed_kernel(const uint8_t* __restrict__ planes, int h, int w, int pitch,
                                                               int low, int high, uint8_t* __restrict__ map,
                                                               int32_t* __restrict__ labels, int32_t* __restrict__ cand,
                                                               int* __restrict__ cand_count, uint8_t* __restrict__ edges_zero,
    const int* __restrict__ thr /* nullable: per-frame (low, high) pairs */,
    const uint8_t* __restrict__ trange /* nullable: (low bound, high bound) of the pixels of every RANGE_TILE^2 tile */, int trw, int trh)
{
    // XCD-aware tile order.  Workgroups go round-robin over the 8 XCDs by their linear id, so with the plain (x, y, frame)
    // mapping the eight horizontal neighbours of a tile row sit on eight different L2s and every halo line (a tile row
    // touches three 64-B lines per plane for 64 useful bytes) is fetched from HBM by each of them.  Here XCD k owns the
    // k-th contiguous eighth of the tile sequence instead: neighbours share an L2.  (Identity when the grid is no
    // multiple of 8.)
    int bxi = blockIdx.x, byi = blockIdx.y, f = blockIdx.z;
    {
        const unsigned G = gridDim.x * gridDim.y * gridDim.z;
        if ((G & 7u) == 0) {
            const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            const unsigned t = (lin & 7u) * (G >> 3) + (lin >> 3);
            bxi = (int)(t % gridDim.x);
            byi = (int)((t / gridDim.x) % gridDim.y);
            f = (int)(t / (gridDim.x * gridDim.y));
        }
    }
    if (thr) { low = thr[2 * f]; high = thr[2 * f + 1]; }
    // pixel tile: columns ox-8 .. ox+71 (20 dwords per row: five 16-byte loads; the gradient needs ox-2 .. ox+65), rows
    // oy-2 .. oy+PTH+1, border replicated.  The candidate buffer of the last phase reuses its space.
    constexpr int LWD = PLWD;
    constexpr int PXW = 3 * PLH * LWD, CBUF = PTH * TW;
    __shared__ __attribute__((aligned(16))) uint32_t smem[PXW > CBUF ? PXW : CBUF];
    uint32_t (*pxw)[PLH][LWD] = reinterpret_cast<uint32_t (*)[PLH][LWD]>(smem);
    int32_t* cbuf = reinterpret_cast<int32_t*>(smem);
    __shared__ __attribute__((aligned(8))) uint16_t mag[PGR][PGW];          // keys: magnitude * 4 + (3 - channel)
    __shared__ __attribute__((aligned(16))) int lab[PTH * TW];              // tile-local union-find parents
    __shared__ int ccount, cbase;
    const int ox = bxi * TW, oy = byi * PTH;
    const int tid = threadIdx.x;
    const uint8_t* base = planes + (size_t)f * 3 * h * pitch;
    if (tid == 0) ccount = 0;

    // Flat tile (round 4): the producer of the planes (the median kernel) leaves, per tile of its own and channel, bounds
    // lo <= every pixel <= hi.  With all nine pixels of a Sobel window within R levels, |dx| + |dy| = max(|dx + dy|, |dx - dy|)
    // and dx + dy = 2 (i - a) + 2 (f - d) + 2 (h - b), dx - dy = 2 (c - g) + 2 (f - d) - 2 (h - b) for the window a b c /
    // d e f / g h i: at most 6 R.  So where the tiles under this tile's pixel region (2-px halo, replicated at the frame's
    // rim: no new values) span R levels with 6 R <= low, no magnitude exceeds `low`: no candidate, map = 1, nothing to
    // stage or to compute.  On a board frame that is half of the tiles (paper, table, wood between the lines).
    if (trange) {
        // max hi - min lo over the (at most 3 x 2) tiles of a channel is within bounds iff hi_i - lo_j is for every pair (i, j):
        // one pair per lane (3 channels x 6 x 6), its two bytes loaded at once -- the test costs ONE memory latency instead of
        // a chain of up to six (the tile waits for it before it stages anything, and a flat tile is nothing but this wait)
        static_assert(TW + 4 <= 2 * RANGE_TILE + 1 && PTH + 4 <= RANGE_TILE + 1, "a tile's pixel region spans at most 3 x 2 range tiles");
        bool flat = low >= 0;
        if (tid < 108 && flat) {
            const int xa = (ox - 2 < 0 ? 0 : ox - 2) / RANGE_TILE, xb = (ox + TW + 1 > w - 1 ? w - 1 : ox + TW + 1) / RANGE_TILE;
            const int ya = (oy - 2 < 0 ? 0 : oy - 2) / RANGE_TILE, yb = (oy + PTH + 1 > h - 1 ? h - 1 : oy + PTH + 1) / RANGE_TILE;
            const int c = tid / 36, pair = tid % 36, ti = pair / 6, tj = pair % 6;
            const int yi = ya + ti / 3, xi = xa + ti % 3, yj = ya + tj / 3, xj = xa + tj % 3;
            if (yi <= yb && xi <= xb && yj <= yb && xj <= xb) {
                const uint8_t* rc = trange + (size_t)(f * 3 + c) * trh * trw * 2;
                const int hi = rc[((size_t)yi * trw + xi) * 2 + 1], lo = rc[((size_t)yj * trw + xj) * 2];
                flat = 6 * (hi - lo) <= low;
            }
        }
        if (__syncthreads_and(flat)) {
            for (int q = tid; q < PTH * 16; q += 256) {
                const int y = oy + (q >> 4), x0 = ox + (q & 15) * 4;
                if (y >= h) break;
                const size_t idx = ((size_t)f * h + y) * w + x0;
                if (x0 + 3 < w && ((idx & 3) == 0)) {
                    *reinterpret_cast<uint32_t*>(map + idx) = 0x01010101u;
                    *reinterpret_cast<uint32_t*>(edges_zero + idx) = 0u;
                } else {
                    for (int k = 0; k < 4; k++)
                        if (x0 + k < w) { map[idx + k] = 1; edges_zero[idx + k] = 0; }
                }
            }
            return;
        }
    }

    const uint32_t plane = (uint32_t)h * pitch;
    if (ox >= 8 && ox + TW + 8 <= w && oy >= 2 && oy + PTH + 2 <= h) {
        // interior tile (all but the frame's rim): no clamping, 16 bytes per load (8-byte aligned: ox is a multiple of 64),
        // one load per plane and thread: 32 rows x 5 quads
        struct __attribute__((packed, aligned(8))) q16 { uint32_t a, b, c, d; };
        const uint8_t* org = base + (uint32_t)((oy - 2) * pitch) + (ox - 8);
        for (int i = tid; i < PLH * (LWD / 4); i += 256) {
            const int r = i / (LWD / 4), v = i - r * (LWD / 4);
            const uint8_t* q = org + (uint32_t)(r * pitch) + 16 * v;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const q16 d = *reinterpret_cast<const q16*>(q + c * plane);
                *reinterpret_cast<uint4*>(&pxw[c][r][4 * v]) = make_uint4(d.a, d.b, d.c, d.d);
            }
        }
    } else {
        for (int i = tid; i < PLH * LWD; i += 256) {
            const int r = i / LWD, cd = i % LWD;
            int y = oy - 2 + r;
            y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
            const int x = ox - 8 + 4 * cd;                 // pitch is a multiple of 64 and ox of 64: aligned
            const uint8_t* row = base + (uint32_t)(y * pitch);
            if (x >= 0 && x + 3 < w) {
#pragma unroll
                for (int c = 0; c < 3; c++) pxw[c][r][cd] = *reinterpret_cast<const uint32_t*>(row + c * plane + x);
            } else {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    uint32_t v = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        int xc = x + k;
                        xc = xc < 0 ? 0 : (xc > w - 1 ? w - 1 : xc);
                        v |= (uint32_t)row[c * plane + xc] << (8 * k);
                    }
                    pxw[c][r][cd] = v;
                }
            }
        }
    }
    __syncthreads();
    if (NMS_STOP == 1) { map[((size_t)f * h + (oy < h ? oy : 0)) * w + (tid % w)] = (uint8_t)smem[tid]; return; }

    if (tid < 17 * 15) {
        const int cj = tid % 17, seg = tid / 17;
        // this thread's gradient columns: array index 4 cj + i  <->  x = ox - 2 + 4 cj + i
        const int x0 = ox - 2 + 4 * cj;
        u16x2 xmP, xmQ;                                // 0xFFFF where the column lies inside the image
        xmP[0] = (x0 >= 0 && x0 < w) ? 0xFFFF : 0;
        xmP[1] = (x0 + 1 >= 0 && x0 + 1 < w) ? 0xFFFF : 0;
        xmQ[0] = (x0 + 2 >= 0 && x0 + 2 < w) ? 0xFFFF : 0;
        xmQ[1] = (x0 + 3 >= 0 && x0 + 3 < w) ? 0xFFFF : 0;
        s16x2 hdP[3][3], hdQ[3][3];                    // [row slot][channel]
        u16x2 hsP[3][3], hsQ[3][3];
#pragma unroll
        for (int k = 0; k < PK + 2; k++) {
            const int pr = seg * PK + k;               // pixel-tile row
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const uint32_t A = pxw[c][pr][cj + 1], B = pxw[c][pr][cj + 2];
                const u16x2 p12 = widen(B, A, 0x0c020c01u), p23 = widen(B, A, 0x0c030c02u), p34 = widen(B, A, 0x0c040c03u),
                            p45 = widen(B, A, 0x0c050c04u), p56 = widen(B, A, 0x0c060c05u);
                hdP[k % 3][c] = __builtin_bit_cast(s16x2, (u16x2)(p34 - p12));
                hsP[k % 3][c] = mad2(p23, p12) + p34;
                hdQ[k % 3][c] = __builtin_bit_cast(s16x2, (u16x2)(p56 - p34));
                hsQ[k % 3][c] = mad2(p45, p34) + p56;
            }
            if (k >= 2) {
                const int gr = seg * PK + k - 2;       // gradient row  <->  y = oy - 1 + gr
                const int y = oy - 1 + gr;
                const unsigned short ym = (y >= 0 && y < h) ? 0xFFFF : 0;
                auto key3 = [&](const s16x2 (&hd)[3][3], const u16x2 (&hs)[3][3]) {
                    u16x2 best;
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const s16x2 dx = mad2(hd[(k - 1) % 3][c], hd[(k - 2) % 3][c]) + hd[k % 3][c];
                        const s16x2 dy = __builtin_bit_cast(s16x2, (u16x2)(hs[k % 3][c] - hs[(k - 2) % 3][c]));
                        const u16x2 m = __builtin_bit_cast(u16x2, (s16x2)(__builtin_elementwise_max(dx, -dx) + __builtin_elementwise_max(dy, -dy)));
                        // the largest key = the largest magnitude, and among equals the first channel
                        if (c == 0) best = key_of<3>(m);
                        else if (c == 1) best = __builtin_elementwise_max(best, key_of<2>(m));
                        else best = __builtin_elementwise_max(best, key_of<1>(m));
                    }
                    return best;
                };
                const u16x2 bP = key3(hdP, hsP) & xmP & ym, bQ = key3(hdQ, hsQ) & xmQ & ym;   // outside the image: 0
                *reinterpret_cast<uint2*>(&mag[gr][4 * cj]) = make_uint2(__builtin_bit_cast(uint32_t, bP), __builtin_bit_cast(uint32_t, bQ));
            }
        }
    }
    __syncthreads();
    if (NMS_STOP == 2) { map[((size_t)f * h + (oy < h ? oy : 0)) * w + (tid % w)] = (uint8_t)mag[tid % PGR][tid % PGW]; return; }   // (keeps the phases alive)

    // NMS: one thread = 4 consecutive pixels of one row (one dword of the map)
    const int TG22 = 13573;   // (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5)
    const uint16_t* magf = &mag[0][0];
    const uint8_t* pxb = reinterpret_cast<const uint8_t*>(smem);
    const unsigned lowkey = (unsigned)(low < 0 ? 0 : (low > 8191 ? 8191 : low)) * 4u + 3u;     // m > low  <=>  key > 4 low + 3
    constexpr int NQ = (PTH * 16 + 255) / 256;
    int km[NQ];
#pragma unroll
    for (int it = 0; it < NQ; it++) {
        const int q = tid + 256 * it;
        km[it] = 0;
        if (q >= PTH * 16) continue;
        const int r = q >> 4, col0 = (q & 15) * 4;
        const int y = oy + r;
        uint32_t mapw = 0x01010101u;
        int keepmask = 0;
        if (y < h) {
            const int i0 = (r + 1) * PGW + col0 + 2;   // output pixel (r, col)  <->  gradient (r + 1, array column col + 2)
            const uint32_t m01 = *reinterpret_cast<const uint32_t*>(magf + i0), m23 = *reinterpret_cast<const uint32_t*>(magf + i0 + 2);
            const uint32_t top = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, m01), __builtin_bit_cast(u16x2, m23)));
            const bool any = (low < 0) || (top & 0xFFFFu) > lowkey || (top >> 16) > lowkey;      // keys beyond the image are 0
            if (__builtin_amdgcn_ballot_w64(any) != 0 && any) {
                const unsigned kk[4] = {m01 & 0xFFFFu, m01 >> 16, m23 & 0xFFFFu, m23 >> 16};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int m = (int)(kk[k] >> 2);
                    if (m > low && ox + col0 + k < w) {
                        const int i = i0 + k;
                        // dx, dy of the chosen channel from its 8 taps: gradient (gr, idx) = pixel-tile rows gr .. gr + 2,
                        // bytes idx + 1 .. idx + 3 of the row
                        const int c = 3 - (int)(kk[k] & 3u);
                        const uint8_t* t0 = pxb + ((c * PLH + r + 1) * LWD) * 4 + col0 + k + 7;
                        const uint8_t* t1 = t0 + LWD * 4;
                        const uint8_t* t2 = t1 + LWD * 4;
                        const int a00 = t0[0], a01 = t0[1], a02 = t0[2], a10 = t1[0], a12 = t1[2], a20 = t2[0], a21 = t2[1], a22 = t2[2];
                        const int bdx = (a02 + 2 * a12 + a22) - (a00 + 2 * a10 + a20);
                        const int bdy = (a20 + 2 * a21 + a22) - (a00 + 2 * a01 + a02);
                        const int ax = abs(bdx), ay = abs(bdy) << 15;
                        const int tg22x = ax * TG22;
                        int o, m2 = m;
                        if (ay < tg22x) { o = 1; m2 = m + 1; }                             // sector 0: left, right (>=)
                        else if (ay > tg22x + (ax << 16)) { o = PGW; m2 = m + 1; }         // sector 1: up, down (>=)
                        else o = ((bdx ^ bdy) < 0) ? PGW - 1 : PGW + 1;                    // diagonals
                        if (m > (int)(magf[i - o] >> 2) && m2 > (int)(magf[i + o] >> 2)) {
                            const uint32_t v = m > high ? 2u : 0u;
                            mapw = (mapw & ~(0xFFu << (8 * k))) | (v << (8 * k));
                            keepmask |= 1 << k;
                        }
                    }
                }
            }
            const int x0 = ox + col0;
            const size_t idx = ((size_t)f * h + y) * w + x0;
            // the edge image of the hysteresis pass starts out all zero: cleared here, next to the map store,
            // instead of by a memset of its own
            if (x0 + 3 < w && ((idx & 3) == 0)) {
                *reinterpret_cast<uint32_t*>(map + idx) = mapw;
                *reinterpret_cast<uint32_t*>(edges_zero + idx) = 0u;
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (x0 + k < w) { map[idx + k] = (uint8_t)(mapw >> (8 * k)); edges_zero[idx + k] = 0; }
            }
        }
        km[it] = keepmask;
    }
    if (NMS_STOP == 3) return;
#if NMS_LOCAL_UF
    // Tile-local part of the hysteresis union-find, in LDS: every candidate is linked with its W / N (or NW, NE)
    // neighbours inside this tile, so the global pass (canny_link_kernel) only has to visit the candidates on the
    // tile's left, right and top edges.  Hooking is by smaller index and local order = raster order, so a local root
    // is the first pixel of its local component, as the global structure wants.
#pragma unroll
    for (int it = 0; it < NQ; it++) {
        const int q = tid + 256 * it;
        if (q >= PTH * 16) continue;
        // local index = r * 64 + col = 4 q + k; a pixel whose west neighbour inside the quad is kept starts out
        // pointing at the head of that little run (one union less)
        int4 v = make_int4(-1, -1, -1, -1);
        if (__builtin_amdgcn_ballot_w64(km[it] != 0) != 0) {      // most waves hold no kept pixel at all
            v.x = (km[it] & 1) ? 4 * q : -1;
            v.y = (km[it] & 2) ? ((km[it] & 1) ? v.x : 4 * q + 1) : -1;
            v.z = (km[it] & 4) ? ((km[it] & 2) ? v.y : 4 * q + 2) : -1;
            v.w = (km[it] & 8) ? ((km[it] & 4) ? v.z : 4 * q + 3) : -1;
        }
        *reinterpret_cast<int4*>(lab + 4 * q) = v;
    }
    {
        int mine = 0;
#pragma unroll
        for (int it = 0; it < NQ; it++) mine |= km[it];
        // parents in place; the pixel tile is dead from here on (cbuf).  A tile without a single kept pixel (the flat
        // parts of a median-filtered frame) has nothing to link and nothing to list: done.
        if (!__syncthreads_or(mine)) return;
    }
    auto lfind = [&](int a) {
        int p = __hip_atomic_load(lab + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (p != a) { a = p; p = __hip_atomic_load(lab + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        return a;
    };
    auto lunion = [&](int a, int b) {
        for (;;) {
            a = lfind(a); b = lfind(b);
            if (a == b) return;
            if (a < b) { const int t = a; a = b; b = t; }
            const int old = atomicMin(lab + a, b);
            if (old == a) return;
            a = old;
        }
    };
#pragma unroll
    for (int it = 0; it < NQ; it++) {
        const int q = tid + 256 * it;
        if (q >= PTH * 16 || !km[it]) continue;
        const int r = q >> 4, col0 = (q & 15) * 4;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (!(km[it] & (1 << k))) continue;
            const int i = 4 * q + k, col = col0 + k;
            if (k == 0 && col > 0 && lab[i - 1] >= 0) lunion(i, i - 1);       // k > 0: linked at initialisation
            if (r > 0) {
                if (lab[i - TW] >= 0) lunion(i, i - TW);
                else {
                    if (col > 0 && lab[i - TW - 1] >= 0) lunion(i, i - TW - 1);
                    if (col < TW - 1 && lab[i - TW + 1] >= 0) lunion(i, i - TW + 1);
                }
            }
        }
    }
    __syncthreads();
    if (NMS_STOP == 4) return;
#else
    {
        int mine = 0;
#pragma unroll
        for (int it = 0; it < NQ; it++) mine |= km[it];
        if (!__syncthreads_or(mine)) return;           // the pixel tile is dead from here on (cbuf)
    }
#endif
#pragma unroll
    for (int it = 0; it < NQ; it++) {
        const int q = tid + 256 * it;
        const int nk = __builtin_popcount(km[it]);
        if (q >= PTH * 16 || !nk) continue;
        const int r = q >> 4, col0 = (q & 15) * 4;
        const int y = oy + r;
        int slot = atomicAdd(&ccount, nk);
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (km[it] & (1 << k)) {
                const int p = y * w + ox + col0 + k;
#if NMS_LOCAL_UF
                const int root = lfind(4 * q + k);
                labels[(size_t)f * h * w + p] = (oy + (root >> 6)) * w + ox + (root & 63);
#else
                labels[(size_t)f * h * w + p] = p;
#endif
                cbuf[slot++] = p;
            }
    }
    __syncthreads();
    // (round 5: one list segment and counter per TILE ROW instead of this returning atomic on one word per frame -- ~1 000 per
    // 1080p frame, the pattern that had cost prep_runs 1 us -- measured level here, 4.42 -> 4.44 us, and the three list kernels
    // behind it lost their even split over the workgroups, 1.15 -> 1.26 us: the kernel is bound by vector issue and the
    // atomic sits at its very end, behind nothing that waits for it.  Dropped.)
    if (tid == 0) cbase = ccount ? atomicAdd(cand_count + f, ccount) : 0;
    __syncthreads();
    int32_t* clist = cand + (size_t)f * h * w + cbase;
    for (int i = tid; i < ccount; i += 256) clist[i] = cbuf[i];
}


// geometry of the NMS kernel that is launched: its tile-local union-find links everything inside a tile, the global
// pass (canny_link_kernel) only visits the candidates on a tile's left / right / top edge
constexpr int NTW = TW, NTH = PTH;

// The three hysteresis kernels walk the per-frame candidate list (a few % of the pixels)
// with a fixed grid and a grid-stride loop; the count is read from device memory.
#ifndef CANNY_LIST_BLOCKS
#define CANNY_LIST_BLOCKS 64
#endif
constexpr int LIST_BLOCKS = CANNY_LIST_BLOCKS;
// link every candidate with its already-scanned 8-neighbours (W, N, and NW / NE only when
// N is not itself a candidate -- otherwise the link is implied)
__global__ __launch_bounds__(256) void canny_link_kernel(const uint8_t* __restrict__ map, int h, int w,
                                                         int32_t* __restrict__ labels, const int32_t* __restrict__ cand,
                                                         const int* __restrict__ cand_count)
{
    int f, bx;
    list_frame_block(LIST_BLOCKS, f, bx);
    const int n = cand_count[f];
    const uint8_t* m = map + (size_t)f * h * w;
    int32_t* L = labels + (size_t)f * h * w;
    const int32_t* C = cand + (size_t)f * h * w;
    for (int i = bx * 256 + threadIdx.x; i < n; i += LIST_BLOCKS * 256) {
        const int p = C[i];
        const int y = p / w, x = p - y * w;
#if NMS_LOCAL_UF
        // links inside a tile of the NMS kernel were made there, in LDS: only a tile's left / right / top edge has
        // a W, N, NW or NE neighbour in another tile
        { const int c = x % NTW; if (c != 0 && c != NTW - 1 && y % NTH != 0) continue; }
#endif
        if (x > 0 && m[p - 1] != 1) uf_union(L, p, p - 1);
        if (y > 0) {
            if (m[p - w] != 1) uf_union(L, p, p - w);
            else {
                if (x > 0 && m[p - w - 1] != 1) uf_union(L, p, p - w - 1);
                if (x < w - 1 && m[p - w + 1] != 1) uf_union(L, p, p - w + 1);
            }
        }
    }
}

__global__ __launch_bounds__(256) void canny_flatten_mark_kernel(const uint8_t* __restrict__ map, int h, int w,
                                                                 int32_t* __restrict__ labels, uint8_t* __restrict__ edges,
                                                                 const int32_t* __restrict__ cand, const int* __restrict__ cand_count)
{
    int f, bx;
    list_frame_block(LIST_BLOCKS, f, bx);
    const int n = cand_count[f];
    const size_t off = (size_t)f * h * w;
    const int32_t* C = cand + off;
    for (int i = bx * 256 + threadIdx.x; i < n; i += LIST_BLOCKS * 256) {
        const int p = C[i];
        const int root = uf_find(labels + off, p);
        labels[off + p] = root;
        if (map[off + p] == 2) edges[off + root] = 255;
    }
}

// border_flag (nullable): frames with an edge pixel on the image frame are flagged -- K3 clears the frame
// before labelling, so only the others may reuse these component roots as they are
__global__ __launch_bounds__(256) void canny_final_kernel(int h, int w, const int32_t* __restrict__ labels,
                                                          uint8_t* __restrict__ edges, const int32_t* __restrict__ cand,
                                                          const int* __restrict__ cand_count, int* __restrict__ border_flag)
{
    int f, bx;
    list_frame_block(LIST_BLOCKS, f, bx);
    const int n = cand_count[f];
    const size_t off = (size_t)f * h * w;
    const int32_t* C = cand + off;
    for (int i = bx * 256 + threadIdx.x; i < n; i += LIST_BLOCKS * 256) {
        const int p = C[i];
        const int root = labels[off + p];
        const uint8_t e = edges[off + root];           // root entries were written by the previous kernel
        if (root != p) edges[off + p] = e;
        if (border_flag && e) {
            const int y = p / w, x = p - y * w;
            if (x == 0 || y == 0 || x == w - 1 || y == h - 1) border_flag[f] = 1;
        }
    }
}

}  // namespace

int k_canny_planar(ck_ctx* ctx, const uint8_t* d_planes, int n, int h, int w, int pitch, int low, int high,
                   uint8_t* d_map, int32_t* d_labels, uint8_t* d_edges, uint8_t* d_map_out, int* d_border_flag,
                   const int* d_thr, const uint8_t* d_range)
{
    if (low > high) { int t = low; low = high; high = t; }
#if !CK_TILE_RANGE
    d_range = nullptr;
#endif
    const size_t npx = (size_t)n * h * w;
    CK_TRY(ck_ensure(ctx, ctx->labels2, npx * 4));                 // candidate lists (one slab per frame)
    CK_TRY(ck_ensure(ctx, ctx->misc, (size_t)n * 64 + 4096));
    int32_t* d_cand = (int32_t*)ctx->labels2.p;
    int* d_count = (int*)ctx->misc.p;
    {
        TimeScope ts(ctx, "canny_nms");
        CK_HIP(ctx, hipMemsetAsync(d_count, 0, (size_t)n * 4, ctx->stream));
        dim3 grid((w + TW - 1) / TW, (h + PTH - 1) / PTH, n);
        hipLaunchKernelGGL(canny_nms_packed_kernel, grid, dim3(256), 0, ctx->stream, d_planes, h, w, pitch, low, high, d_map,
                           d_labels, d_cand, d_count, d_edges, d_thr, d_range, (w + RANGE_TILE - 1) / RANGE_TILE,
                           (h + RANGE_TILE - 1) / RANGE_TILE);
        CK_HIP(ctx, hipGetLastError());
    }
    if (d_map_out) CK_HIP(ctx, hipMemcpyAsync(d_map_out, d_map, npx, hipMemcpyDeviceToDevice, ctx->stream));
    {
        TimeScope ts(ctx, "canny_hyst");
        if (d_border_flag) CK_HIP(ctx, hipMemsetAsync(d_border_flag, 0, (size_t)n * 4, ctx->stream));
        const dim3 grid = list_grid(LIST_BLOCKS, n);
        hipLaunchKernelGGL(canny_link_kernel, grid, dim3(256), 0, ctx->stream, (const uint8_t*)d_map, h, w, d_labels,
                           (const int32_t*)d_cand, (const int*)d_count);
        hipLaunchKernelGGL(canny_flatten_mark_kernel, grid, dim3(256), 0, ctx->stream, (const uint8_t*)d_map, h, w,
                           d_labels, d_edges, (const int32_t*)d_cand, (const int*)d_count);
        hipLaunchKernelGGL(canny_final_kernel, grid, dim3(256), 0, ctx->stream, h, w, (const int32_t*)d_labels, d_edges,
                           (const int32_t*)d_cand, (const int*)d_count, d_border_flag);
        CK_HIP(ctx, hipGetLastError());
    }
    return CK_OK;
}
