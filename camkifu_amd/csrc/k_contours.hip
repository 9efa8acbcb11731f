// k_contours.hip -- K3..K6 of BoardFinderAuto._detect / find_lines
// (reference: src/camkifu/board/bf_auto.py:75-84, 105-133; core/imgutil.py:291-315, 409-434):
//   K3 cv2.findContours(canny, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)
//   K4 minAreaRect of every contour, bisect.insort by area, keep the 3 biggest, area gate
//   K5 cv2.drawContours(ghost, contours, pos, 255, 1) for those 3
//   K6 cv2.HoughLines(ghost, 1, pi/180, thresh)
//
// GPU formulation.  The serial border following of the CPU library is replaced by its
// closed form (proved equal on the oracle side, tests/test_oracle_properties.py):
//   * clear the 1-px image frame;
//   * S0 = the 4-connected background component that contains the frame;
//   * RETR_EXTERNAL contours = outer borders of the 8-connected edge components whose
//     first pixel (raster order) has its west neighbour in S0;
//   * the outer border of such a component = its pixels with a 4-neighbour in S0, which is
//     also exactly what drawContours(thickness=1) paints from the compressed vertex list.
// Both labelings share ONE int32 parent image: edge pixels are union-find nodes
// (8-connectivity), background pixels point at the first pixel of their horizontal zero-run
// and only those run heads are nodes (4-connectivity: runs are united where they overlap
// vertically), so the dense background costs one union per run, not per pixel.
// The three biggest contours need float rotating calipers; that scalar, branchy search runs
// on the host for the few components whose bounding box could still make the top three.
// Hough voting keeps a slab of theta rows in LDS per workgroup (LDS atomics);
// peaks are found on the slab itself (16-bit counters): no accumulator in HBM (the two-kernel form through a global
// accumulator is archived in tools/variants/hough_unfused.hip.txt).
#include <math.h>

#include <algorithm>
#include <chrono>
#include <thread>

#include "ck_common.h"

#include "ck_uf.h"

#pragma clang fp contract(off)

namespace {

constexpr int MAXC_LIMIT = 1 << 17;     // top-level components per frame
constexpr int PEAK_CAP = 1 << 14;       // Hough peaks per frame
constexpr int NUMANGLE = 180;

struct FrameTab {        // device-side per-frame counters
    int n_edges;
    int n_border;
    int n_roots;
    int n_hough_pts;
    int n_peaks;
    int overflow;
    int runs_overflow;      // run-table form: more edge pixels than run nodes were provided for (-> dense form, on the host's say)
    int pad1;
};

#ifndef CCL_LIST_BLOCKS
#define CCL_LIST_BLOCKS 32      // 16 .. 128 measured in round 4 (ccl per frame: 16: 4.36, 24: 4.42, 32: 4.35, 48: 4.5, 128: 4.98 us)
#endif
constexpr int LIST_BLOCKS = CCL_LIST_BLOCKS;          // grid-stride blocks per frame for the list kernels
#ifndef CK_HOUGH_THREADS
#define CK_HOUGH_THREADS 1024    // the theta slab fills a CU's LDS (one workgroup per CU): all the latency hiding comes from its own waves
#endif
constexpr int HOUGH_THREADS = CK_HOUGH_THREADS;
#ifndef CK_HOUGH_SMALL_N
#define CK_HOUGH_SMALL_N 32      // calls of at most this many frames use the small Hough slabs (k_board_lines)
#endif
constexpr int HOUGH_SMALL_N = CK_HOUGH_SMALL_N;

// ---- A. frame clearing + parent initialisation + edge list -----------------------------
// one wave per image row; walks the row in 64-pixel segments carrying the position of the
// last edge pixel seen so far.  Dense pass: reads 1 B/px, writes 1 + 4 B/px.
__global__ __launch_bounds__(256) void prep_rows_kernel(const uint8_t* __restrict__ edges, int h, int w,
                                                        uint8_t* __restrict__ ez, int32_t* __restrict__ L,
                                                        FrameTab* __restrict__ tab, int32_t* __restrict__ elist)
{
    const int lane = threadIdx.x & 63;
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int f = blockIdx.y;
    if (y >= h) return;                                  // whole wave leaves together
    const size_t off = ((size_t)f * h + y) * w;
    const bool row_inner = y > 0 && y < h - 1;
    int32_t* E = elist + (size_t)f * h * w;
    int last_edge = -1;                       // wave-uniform
    for (int x0 = 0; x0 < w; x0 += 64) {
        const int x = x0 + lane;
        bool e = false;
        if (x < w) e = row_inner && x > 0 && x < w - 1 && edges[off + x] != 0;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(e);
        if (x < w) {
            ez[off + x] = e ? 1 : 0;
            int parent;
            if (e) parent = y * w + x;
            else {
                const unsigned long long below = mask & ((1ull << lane) - 1ull);
                const int le = below ? x0 + 63 - __builtin_clzll(below) : last_edge;
                parent = y * w + le + 1;          // head of this zero-run
            }
            L[off + x] = parent;
        }
        if (mask) {
            const int slot = wave_append(&tab[f].n_edges, e);
            if (e) E[slot] = y * w + x;
            last_edge = x0 + 63 - __builtin_clzll(mask);
        }
    }
}

// Same as above for w % 4 == 0: a lane owns 4 consecutive pixels (one dword of the edge map),
// a wave covers 256 pixels per step; 4 ballots give every edge pixel its list slot.
__global__ __launch_bounds__(256) void prep_rows4_kernel(const uint8_t* __restrict__ edges, int h, int w,
                                                         uint8_t* __restrict__ ez, int32_t* __restrict__ L,
                                                         FrameTab* __restrict__ tab, int32_t* __restrict__ elist)
{
    const int lane = threadIdx.x & 63;
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int f = blockIdx.y;
    if (y >= h) return;                                  // whole wave leaves together
    const size_t off = ((size_t)f * h + y) * w;
    const bool row_inner = y > 0 && y < h - 1;
    int32_t* E = elist + (size_t)f * h * w;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int last_edge = -1;                       // wave-uniform: last edge column of earlier steps
    for (int x0 = 0; x0 < w; x0 += 256) {
        const int x = x0 + 4 * lane;
        uint32_t v = 0;
        if (x < w && row_inner) v = *reinterpret_cast<const uint32_t*>(edges + off + x);
        int nib = ((v & 0xFFu) ? 1 : 0) | ((v & 0xFF00u) ? 2 : 0) | ((v & 0xFF0000u) ? 4 : 0) | ((v & 0xFF000000u) ? 8 : 0);
        if (x == 0) nib &= ~1;                           // cleared frame: first and last column
        if (x + 3 == w - 1) nib &= ~8;
        const unsigned long long has = __builtin_amdgcn_ballot_w64(nib != 0);
        // column of the last edge pixel in lanes before this one (or in earlier steps)
        const int my_last = nib ? x + 31 - __builtin_clz((unsigned)nib) : -1;
        const unsigned long long below = has & lt;
        const int src = below ? 63 - __builtin_clzll(below) : 0;
        int prev_last = __shfl(my_last, src);
        if (!below) prev_last = last_edge;
        if (x < w) {
            *reinterpret_cast<uint32_t*>(ez + off + x) =
                (nib & 1 ? 1u : 0u) | (nib & 2 ? 0x100u : 0u) | (nib & 4 ? 0x10000u : 0u) | (nib & 8 ? 0x1000000u : 0u);
            int4 par;
            int le = prev_last;
            par.x = (nib & 1) ? y * w + x : y * w + le + 1;
            if (nib & 1) le = x;
            par.y = (nib & 2) ? y * w + x + 1 : y * w + le + 1;
            if (nib & 2) le = x + 1;
            par.z = (nib & 4) ? y * w + x + 2 : y * w + le + 1;
            if (nib & 4) le = x + 2;
            par.w = (nib & 8) ? y * w + x + 3 : y * w + le + 1;
            *reinterpret_cast<int4*>(L + off + x) = par;
        }
        if (has) {
            const unsigned long long b0 = __builtin_amdgcn_ballot_w64(nib & 1), b1 = __builtin_amdgcn_ballot_w64(nib & 2);
            const unsigned long long b2 = __builtin_amdgcn_ballot_w64(nib & 4), b3 = __builtin_amdgcn_ballot_w64(nib & 8);
            const int total = __builtin_popcountll(b0) + __builtin_popcountll(b1) + __builtin_popcountll(b2) + __builtin_popcountll(b3);
            const int leader = __builtin_ctzll(has);
            int base = 0;
            if (lane == leader) base = atomicAdd(&tab[f].n_edges, total);
            base = __builtin_amdgcn_readlane(base, leader);
            int slot = base + __builtin_popcountll(b0 & lt) + __builtin_popcountll(b1 & lt) +
                       __builtin_popcountll(b2 & lt) + __builtin_popcountll(b3 & lt);
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (nib & (1 << k)) E[slot++] = y * w + x + k;
            const int hl = 63 - __builtin_clzll(has);
            last_edge = __shfl(my_last, hl);
        }
    }
}

// Rows of up to 256*NS pixels, fully unrolled: all NS dword loads of the row are in flight at
// once and the row takes ONE atomic on the frame's edge counter (the loop version above pays a
// load round trip and an atomic per 256 pixels).
// (Tried: writing a background pixel's parent only where later kernels read it -- next to an edge pixel, at x = 0 --
// which needs the edge bits of the rows above and below: bit-exact, but the two extra row loads and the scattered
// 4-byte stores cost more than the dense 16-byte stores they replace: ccl 7.95 against 7.65 us.)
template <int NS>
__global__ __launch_bounds__(256) void prep_rows4u_kernel(const uint8_t* __restrict__ edges, int h, int w,
                                                          uint8_t* __restrict__ ez, int32_t* __restrict__ L,
                                                          FrameTab* __restrict__ tab, int32_t* __restrict__ elist,
                                                          const int* __restrict__ canny_border_flag)
{
    const int lane = threadIdx.x & 63;
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int f = blockIdx.y;
    if (y >= h) return;                                  // whole wave leaves together
    // L still holds Canny's hysteresis labels: every edge pixel already points at the first pixel of its
    // 8-connected component.  Unless an edge touched the image frame (cleared below, which may split a
    // component) those parents are exactly what the edge unions would rebuild, so they are kept.
    const bool keep_edge_parents = canny_border_flag != nullptr && canny_border_flag[f] == 0;
    const size_t off = ((size_t)f * h + y) * w;
    const bool row_inner = y > 0 && y < h - 1;
    int32_t* E = elist + (size_t)f * h * w;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t v[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int x = 256 * s + 4 * lane;
        v[s] = 0;
        if (x < w && row_inner) v[s] = *reinterpret_cast<const uint32_t*>(edges + off + x);
    }
    int nib[NS], before[NS];          // before: edge pixels of this row in earlier lanes / steps
    unsigned long long has[NS];
    int total = 0;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int x = 256 * s + 4 * lane;
        int nb = ((v[s] & 0xFFu) ? 1 : 0) | ((v[s] & 0xFF00u) ? 2 : 0) | ((v[s] & 0xFF0000u) ? 4 : 0) | ((v[s] & 0xFF000000u) ? 8 : 0);
        if (x == 0) nb &= ~1;                            // cleared frame: first and last column
        if (x + 3 == w - 1) nb &= ~8;
        nib[s] = nb;
        has[s] = __builtin_amdgcn_ballot_w64(nb != 0);
        before[s] = total;
        if (has[s]) {                                    // wave-uniform
            const unsigned long long b0 = __builtin_amdgcn_ballot_w64(nb & 1), b1 = __builtin_amdgcn_ballot_w64(nb & 2);
            const unsigned long long b2 = __builtin_amdgcn_ballot_w64(nb & 4), b3 = __builtin_amdgcn_ballot_w64(nb & 8);
            before[s] += __builtin_popcountll(b0 & lt) + __builtin_popcountll(b1 & lt) +
                         __builtin_popcountll(b2 & lt) + __builtin_popcountll(b3 & lt);
            total += __builtin_popcountll(b0) + __builtin_popcountll(b1) + __builtin_popcountll(b2) + __builtin_popcountll(b3);
        }
    }
    int base = 0;
    if (total) {
        if (lane == 0) base = atomicAdd(&tab[f].n_edges, total);
        base = __builtin_amdgcn_readfirstlane(base);
    }
    int last_edge = -1;                                  // wave-uniform: last edge column of earlier steps
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int x = 256 * s + 4 * lane;
        const int nb = nib[s];
        const int my_last = nb ? x + 31 - __builtin_clz((unsigned)nb) : -1;
        const unsigned long long below = has[s] & lt;
        const int src = below ? 63 - __builtin_clzll(below) : 0;
        int prev_last = __shfl(my_last, src);
        if (!below) prev_last = last_edge;
        if (x < w) {
            *reinterpret_cast<uint32_t*>(ez + off + x) =
                (nb & 1 ? 1u : 0u) | (nb & 2 ? 0x100u : 0u) | (nb & 4 ? 0x10000u : 0u) | (nb & 8 ? 0x1000000u : 0u);
            int4 par;
            int le = prev_last;
            par.x = (nb & 1) ? y * w + x : y * w + le + 1;
            if (nb & 1) le = x;
            par.y = (nb & 2) ? y * w + x + 1 : y * w + le + 1;
            if (nb & 2) le = x + 1;
            par.z = (nb & 4) ? y * w + x + 2 : y * w + le + 1;
            if (nb & 4) le = x + 2;
            par.w = (nb & 8) ? y * w + x + 3 : y * w + le + 1;
            if (keep_edge_parents && nb) {
                const int4 old = *reinterpret_cast<const int4*>(L + off + x);
                if (nb & 1) par.x = old.x;
                if (nb & 2) par.y = old.y;
                if (nb & 4) par.z = old.z;
                if (nb & 8) par.w = old.w;
            }
            *reinterpret_cast<int4*>(L + off + x) = par;
            int slot = base + before[s];
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (nb & (1 << k)) E[slot++] = y * w + x + k;
        }
        if (has[s]) last_edge = __shfl(my_last, 63 - __builtin_clzll(has[s]));
    }
}

// ---- B. unions: edge pixels (8-connectivity) and background runs (4-connectivity) -------
// Work items: every edge pixel, plus one item per image row for the run that starts at x = 0.
// A stretch of columns where this row and the row above are both background starts either at
// x = 0 or right after an edge pixel of one of the two rows: one union per stretch.
__global__ __launch_bounds__(256) void link_list_kernel(const uint8_t* __restrict__ ez, int h, int w,
                                                        int32_t* __restrict__ labels, const FrameTab* __restrict__ tab,
                                                        const int32_t* __restrict__ elist,
                                                        const int* __restrict__ canny_border_flag)
{
    int f, bx;
    list_frame_block(LIST_BLOCKS, f, bx);
    const bool edges_linked = canny_border_flag != nullptr && canny_border_flag[f] == 0;   // parents kept from Canny
    const int ne = tab[f].n_edges;
    const uint8_t* e = ez + (size_t)f * h * w;
    int32_t* L = labels + (size_t)f * h * w;
    const int32_t* E = elist + (size_t)f * h * w;
    for (int i = bx * 256 + threadIdx.x; i < ne + h - 1; i += LIST_BLOCKS * 256) {
        if (i < ne) {
            const int p = E[i];                        // 1 <= x <= w-2, 1 <= y <= h-2
            if (!edges_linked) {
                if (e[p - 1]) uf_union(L, p, p - 1);
                if (e[p - w]) uf_union(L, p, p - w);
                else {
                    if (e[p - w - 1]) uf_union(L, p, p - w - 1);
                    if (e[p - w + 1]) uf_union(L, p, p - w + 1);
                }
            }
            const int q1 = p + 1, q2 = p + w + 1;      // background stretches opening right of p
            if (!e[q1] && !e[q1 - w]) uf_union(L, q1, q1 - w);
            // (if the pixel below p is an edge pixel too -- a vertical stroke -- that pixel's own q1 is this very
            // stretch: leave it to it)
            if (!e[q2] && !e[q2 - w] && !e[p + w]) uf_union(L, q2, q2 - w);
        } else {
            const int q = (i - ne + 1) * w;            // x = 0 of rows 1 .. h-1: always background
            uf_union(L, q, q - w);
        }
    }
}

// ---- C. flatten the nodes that later lookups go through --------------------------------
__global__ __launch_bounds__(256) void flatten_list_kernel(const uint8_t* __restrict__ ez, int h, int w,
                                                           int32_t* __restrict__ labels, const FrameTab* __restrict__ tab,
                                                           const int32_t* __restrict__ elist,
                                                           const int* __restrict__ canny_border_flag)
{
    int f, bx;
    list_frame_block(LIST_BLOCKS, f, bx);
    const bool edges_flat = canny_border_flag != nullptr && canny_border_flag[f] == 0;   // Canny left every edge pixel at its root
    const int ne = tab[f].n_edges;
    const size_t off = (size_t)f * h * w;
    const uint8_t* e = ez + off;
    int32_t* L = labels + off;
    const int32_t* E = elist + off;
    for (int i = bx * 256 + threadIdx.x; i < ne + h; i += LIST_BLOCKS * 256) {
        if (i < ne) {
            const int p = E[i];
            if (!edges_flat) L[p] = uf_find(L, p);
            if (!e[p + 1]) L[p + 1] = uf_find(L, p + 1);       // head of the run right of p
        } else {
            const int q = (i - ne) * w;
            L[q] = uf_find(L, q);
        }
    }
}

__device__ __forceinline__ bool in_s0(const int32_t* L, int q, int root0)
{
    // q is a background pixel: L[q] is its run head (or already a root); heads are flattened
    return L[L[q]] == root0;
}

// ---- D. top-level roots -------------------------------------------------------------------
__global__ __launch_bounds__(256) void roots_list_kernel(int h, int w, const int32_t* __restrict__ labels,
                                                         int32_t* __restrict__ compid, FrameTab* __restrict__ tab, int maxc,
                                                         int32_t* __restrict__ roots, int32_t* __restrict__ aabb,
                                                         const int32_t* __restrict__ elist)
{
    int f, bx;
    list_frame_block(LIST_BLOCKS, f, bx);
    const int ne = tab[f].n_edges;
    const size_t off = (size_t)f * h * w;
    const int32_t* L = labels + off;
    const int32_t* E = elist + off;
    const int root0 = L[0];
    for (int i = bx * 256 + threadIdx.x; i < ne; i += LIST_BLOCKS * 256) {
        const int p = E[i];
        if (L[p] != p) continue;
        if (!in_s0(L, p - 1, root0)) continue;      // west neighbour of a first pixel is background
        const int slot = atomicAdd(&tab[f].n_roots, 1);
        if (slot >= maxc) { tab[f].overflow = 1; compid[off + p] = -1; continue; }
        compid[off + p] = slot;
        roots[(size_t)f * maxc + slot] = p;
        int32_t* bb = aabb + ((size_t)f * maxc + slot) * 4;
        bb[0] = 0x7fffffff; bb[1] = -1; bb[2] = 0x7fffffff; bb[3] = -1;
    }
}

// ---- E. outer-border list + bounding boxes ----------------------------------------------
__global__ __launch_bounds__(256) void border_list_kernel(const uint8_t* __restrict__ ez, int h, int w,
                                                          const int32_t* __restrict__ labels, const int32_t* __restrict__ compid,
                                                          int maxc, FrameTab* __restrict__ tab, int32_t* __restrict__ aabb,
                                                          const int32_t* __restrict__ elist, int32_t* __restrict__ blist)
{
    int f, bx;
    list_frame_block(LIST_BLOCKS, f, bx);
    const int ne = tab[f].n_edges;
    const size_t off = (size_t)f * h * w;
    const uint8_t* e = ez + off;
    const int32_t* L = labels + off;
    const int32_t* E = elist + off;
    int32_t* B = blist + off;
    const int root0 = L[0];
    const int trips = (ne + LIST_BLOCKS * 256 - 1) / (LIST_BLOCKS * 256);      // uniform trip count
    for (int t = 0; t < trips; t++) {
        const int i = t * LIST_BLOCKS * 256 + bx * 256 + threadIdx.x;
        bool b = false;
        int p = 0, cslot = -1;
        if (i < ne) {
            p = E[i];
            // the four neighbours at once (three rounds of independent loads instead of up to twelve dependent ones behind
            // short-circuit tests): edge byte, run head, head's root.  An edge neighbour looks up L[0], which is root0's own
            // node and never counts because of the edge test.
            const int q[4] = { p - 1, p + 1, p - w, p + w };
            int ev[4], head[4], top[4];
#pragma unroll
            for (int k = 0; k < 4; k++) ev[k] = e[q[k]];
#pragma unroll
            for (int k = 0; k < 4; k++) head[k] = L[ev[k] ? 0 : q[k]];
#pragma unroll
            for (int k = 0; k < 4; k++) top[k] = L[head[k]];
#pragma unroll
            for (int k = 0; k < 4; k++) b |= !ev[k] && top[k] == root0;
            if (b) {
                cslot = compid[off + L[p]];
                if ((unsigned)cslot >= (unsigned)maxc) b = false;
            }
        }
        // bounding boxes: a wave's border pixels belong to one component or to a few; reduce per component inside the wave
        // and let one lane update the box -- and only where the box actually grows (a relaxed read first: a stale value
        // merely costs a redundant atomic, min / max are monotone), so the big contours stop hammering four addresses
        {
            unsigned long long todo = __builtin_amdgcn_ballot_w64(b);
            const int y = p / w, x = p - y * w;
            const int lane = threadIdx.x & 63;
            while (todo) {
                const int lead = __builtin_ctzll(todo);
                const int s_lead = __builtin_amdgcn_readlane(cslot, lead);
                const bool mine = b && cslot == s_lead;
                todo &= ~__builtin_amdgcn_ballot_w64(mine);
                int mnx = mine ? x : 0x7fffffff, mxx = mine ? x : -1, mny = mine ? y : 0x7fffffff, mxy = mine ? y : -1;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    mnx = min(mnx, __shfl_xor(mnx, d)); mxx = max(mxx, __shfl_xor(mxx, d));
                    mny = min(mny, __shfl_xor(mny, d)); mxy = max(mxy, __shfl_xor(mxy, d));
                }
                if (lane == lead) {
                    int32_t* bb = aabb + ((size_t)f * maxc + s_lead) * 4;
                    if (mnx < uf_load(bb, 0)) atomicMin(bb + 0, mnx);
                    if (mxx > uf_load(bb, 1)) atomicMax(bb + 1, mxx);
                    if (mny < uf_load(bb, 2)) atomicMin(bb + 2, mny);
                    if (mxy > uf_load(bb, 3)) atomicMax(bb + 3, mxy);
                }
            }
        }
        const int slot = wave_append(&tab[f].n_border, b);
        if (b) B[slot] = p;
    }
}

// ==========================================================================================================
// RUN-TABLE form of steps A..E (the board path; k_contour_survey keeps the dense form above).
// The dense form writes an int32 parent for EVERY pixel (8.3 MB per 1080p frame) only so that a background pixel
// can name the horizontal zero-run it lies in.  An edge map is a few percent dense, so here the runs are named
// through the edge list instead:
//   * bits  [h][w64]  the cleared-frame edge map, one bit per pixel (259 KB per 1080p frame);
//   * rank  [h][w64]  edge pixels of the row in the words before this one (exclusive prefix, u16);
//   * rowbase[h]      where the row's segment of the edge list starts (rows are appended with one atomic each, so
//                     the segments are in no particular order -- nothing needs raster order of the LIST);
//   * the run that starts at x = 0 of row y is node y; the run right of the edge pixel at list position i is node
//     h + i.  A background pixel (y, x) lies in the run of the last edge pixel of its row before x:
//     r = rank[y][x / 64] + popcount(bits below x) -> node r ? h + rowbase[y] + r - 1 : y.  Three small independent
//     loads instead of a dependent one into a 8 MB image;
//   * rp[h + n_edges] union-find parents of the run nodes (4-connectivity).  The edge components (8-connectivity)
//     stay in the dense label image, which only ever holds them at edge pixels: Canny's labels when no edge touched
//     the image frame, rebuilt here otherwise.
// prep then moves 2 MB in + 2.4 MB out per 1080p frame instead of 2 + 10.3.
#ifndef CCL_RUNS_EZ
#define CCL_RUNS_EZ 0           // 1: prep_runs also writes the cleared-frame edge bytes (round 3; nothing in the run-table form reads them)
#endif
#ifndef CCL_PRELINK
#define CCL_PRELINK 1
#endif
struct RunTab {
    unsigned long long* bits;
    uint16_t* rank;
    int32_t* rowbase;
    int32_t* rp;
    int w64;
    int cap_e;                   // edge pixels per frame that have a run node (rp holds h + cap_e parents per frame)
    size_t rp_stride;            // ints per frame in rp
};

__device__ __forceinline__ bool rt_edge(const unsigned long long* __restrict__ bf, int w64, int y, int x)
{
    return (bf[(size_t)y * w64 + (x >> 6)] >> (x & 63)) & 1ull;
}
// node of the background pixel (y, x)
__device__ __forceinline__ int rt_run(const unsigned long long* __restrict__ bf, const uint16_t* __restrict__ rk,
                                      const int32_t* __restrict__ rb, int h, int w64, int y, int x)
{
    const size_t wi = (size_t)y * w64 + (x >> 6);
    const int r = (int)rk[wi] + __builtin_popcountll(bf[wi] & ((1ull << (x & 63)) - 1ull));
    return r ? h + rb[y] + r - 1 : y;
}

// OR of a 64-bit value over the 16 lanes of a DPP row (every lane of the row gets the result)
__device__ __forceinline__ unsigned long long row16_or(unsigned long long v)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
#define CK_DPP_OR(CTRL)                                                                   \
    lo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, CTRL, 0xf, 0xf, true);       \
    hi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, CTRL, 0xf, 0xf, true);
    CK_DPP_OR(0xB1)      // quad_perm [1, 0, 3, 2]
    CK_DPP_OR(0x4E)      // quad_perm [2, 3, 0, 1]
    CK_DPP_OR(0x141)     // row_half_mirror
    CK_DPP_OR(0x140)     // row_mirror
#undef CK_DPP_OR
    return ((unsigned long long)hi << 32) | lo;
}

// A'. one wave per image row: cleared-frame edge bytes (later kernels read them), bit words + rank prefix, the row's
// segment of the edge list, the run nodes' parents.  Rows of up to 256 * NS pixels, all loads of a row in flight at once.
template <int NS>
__global__ __launch_bounds__(256) void prep_runs_kernel(const uint8_t* __restrict__ edges, int h, int w,
                                                        uint8_t* __restrict__ ez, int32_t* __restrict__ L,
                                                        FrameTab* __restrict__ tab, int32_t* __restrict__ elist, RunTab rt,
                                                        const int* __restrict__ canny_border_flag)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int y = blockIdx.x * 4 + wv;
    const int f = blockIdx.y;
    const bool row_live = y < h;                         // (rows beyond the image still meet the workgroup's barriers)
    __shared__ int wtotal[4], wbase;
    // L holds Canny's hysteresis labels at the edge pixels: kept unless an edge touched the image frame (cleared
    // below, which may split a component)
    const bool keep_edge_parents = canny_border_flag != nullptr && canny_border_flag[f] == 0;
    const size_t off = ((size_t)f * h + y) * w;
    const bool row_inner = y > 0 && y < h - 1;
    int32_t* E = elist + (size_t)f * h * w;
    int32_t* rp = rt.rp + (size_t)f * rt.rp_stride;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t v[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int x = 256 * s + 4 * lane;
        v[s] = 0;
        if (x < w && row_inner && row_live) v[s] = *reinterpret_cast<const uint32_t*>(edges + off + x);
    }
    int nib[NS], before[NS];          // before: edge pixels of this row in earlier lanes / steps
    int total = 0;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int x = 256 * s + 4 * lane;
        int nb = ((v[s] & 0xFFu) ? 1 : 0) | ((v[s] & 0xFF00u) ? 2 : 0) | ((v[s] & 0xFF0000u) ? 4 : 0) | ((v[s] & 0xFF000000u) ? 8 : 0);
        if (x == 0) nb &= ~1;                            // cleared frame: first and last column
        if (x + 3 == w - 1) nb &= ~8;
        nib[s] = nb;
        before[s] = total;
        if (__builtin_amdgcn_ballot_w64(nb != 0)) {      // wave-uniform
            const unsigned long long b0 = __builtin_amdgcn_ballot_w64(nb & 1), b1 = __builtin_amdgcn_ballot_w64(nb & 2);
            const unsigned long long b2 = __builtin_amdgcn_ballot_w64(nb & 4), b3 = __builtin_amdgcn_ballot_w64(nb & 8);
            before[s] += __builtin_popcountll(b0 & lt) + __builtin_popcountll(b1 & lt) +
                         __builtin_popcountll(b2 & lt) + __builtin_popcountll(b3 & lt);
            total += __builtin_popcountll(b0) + __builtin_popcountll(b1) + __builtin_popcountll(b2) + __builtin_popcountll(b3);
        }
    }
    // one returning atomic per WORKGROUP (four rows) instead of one per row: the rows' counts meet in LDS, thread 0 takes
    // the segment for all four, each row starts behind the rows before it (1 080 atomics per frame on ONE word were the
    // kernel's longest dependency)
    if (lane == 0) wtotal[wv] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int all = wtotal[0] + wtotal[1] + wtotal[2] + wtotal[3];
        wbase = all ? atomicAdd(&tab[f].n_edges, all) : 0;
    }
    __syncthreads();
    if (!row_live) return;
    int base = wbase;
    for (int kk = 0; kk < wv; kk++) base += wtotal[kk];
    base = __builtin_amdgcn_readfirstlane(base);
    // the run nodes are provided for cap_e edge pixels per frame (a quarter of the pixels: Canny output is a few percent
    // dense); a frame beyond that says so and is redone in the dense form by the host -- nothing below indexes rp then
    const bool nodes_fit = base + total <= rt.cap_e;
    if (!nodes_fit && lane == 0) tab[f].runs_overflow = 1;
    if (lane == 0) {
        rt.rowbase[(size_t)f * h + y] = base;
        // the run that starts at x = 0.  Column 0 of the cleared frame is background in every row: these runs are one
        // component by construction, so they start out pointing at node 0 (row 0, the root of the outer background)
        // instead of being chained row to row by h - 1 unions
        rp[y] = CCL_PRELINK ? 0 : y;
    }
    unsigned long long* bw = rt.bits + ((size_t)f * h + y) * rt.w64;
    uint16_t* rw = rt.rank + ((size_t)f * h + y) * rt.w64;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int x = 256 * s + 4 * lane;
        const int nb = nib[s];
        const unsigned long long word = row16_or((unsigned long long)nb << (4 * (lane & 15)));
        if (x < w) {
            if (ez)                                      // (the run-table form itself never reads the cleared bytes: round 4)
                *reinterpret_cast<uint32_t*>(ez + off + x) =
                    (nb & 1 ? 1u : 0u) | (nb & 2 ? 0x100u : 0u) | (nb & 4 ? 0x10000u : 0u) | (nb & 8 ? 0x1000000u : 0u);
            if ((lane & 15) == 0) {
                bw[4 * s + (lane >> 4)] = word;
                rw[4 * s + (lane >> 4)] = (uint16_t)before[s];
            }
            if (nb) {
                int slot = base + before[s];
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (nb & (1 << k)) {
                        const int p = y * w + x + k;
                        E[slot] = p;
                        // the stretch right of the row's LAST edge pixel reaches column w - 1, background in every row and
                        // joined to row 0 through it: born into the outer background as well
                        if (nodes_fit) rp[h + slot] = (CCL_PRELINK && slot == base + total - 1) ? 0 : h + slot;
                        if (!keep_edge_parents) L[(size_t)f * h * w + p] = p;
                        slot++;
                    }
            }
        }
    }
}

// B'. unions: the background stretches that open right of an edge pixel (this row and the row below) and -- only when
// Canny's labels could not be kept -- the edge pixels themselves.  (The runs at x = 0 and those that reach x = w - 1 are
// born linked to node 0 in A': round 4.  Chained by unions, 1 080 of them on one growing path per frame, they were the
// long dependency of this kernel.)
__global__ __launch_bounds__(256) void link_runs_kernel(int h, int w, int32_t* __restrict__ labels, const FrameTab* __restrict__ tab,
                                                        const int32_t* __restrict__ elist, RunTab rt,
                                                        const int* __restrict__ canny_border_flag)
{
    int f, bx;
    list_frame_block(LIST_BLOCKS, f, bx);
    if (tab[f].runs_overflow) return;                  // this call is redone in the dense form (k_board_lines)
    const bool edges_linked = canny_border_flag != nullptr && canny_border_flag[f] == 0;   // parents kept from Canny
    const int ne = tab[f].n_edges, w64 = rt.w64;
    int32_t* L = labels + (size_t)f * h * w;
    const int32_t* E = elist + (size_t)f * h * w;
    const unsigned long long* bf = rt.bits + (size_t)f * h * w64;
    const uint16_t* rk = rt.rank + (size_t)f * h * w64;
    const int32_t* rb = rt.rowbase + (size_t)f * h;
    int32_t* rp = rt.rp + (size_t)f * rt.rp_stride;
    for (int i = bx * 256 + threadIdx.x; i < ne + (CCL_PRELINK ? 0 : h - 1); i += LIST_BLOCKS * 256) {
        if (i < ne) {
            const int p = E[i];                        // 1 <= x <= w-2, 1 <= y <= h-2
            const int y = p / w, x = p - y * w;
            const bool e_n = rt_edge(bf, w64, y - 1, x), e_ne = rt_edge(bf, w64, y - 1, x + 1);
            const bool e_e = rt_edge(bf, w64, y, x + 1), e_s = rt_edge(bf, w64, y + 1, x), e_se = rt_edge(bf, w64, y + 1, x + 1);
            if (!edges_linked) {
                if (rt_edge(bf, w64, y, x - 1)) uf_union(L, p, p - 1);
                if (e_n) uf_union(L, p, p - w);
                else {
                    if (rt_edge(bf, w64, y - 1, x - 1)) uf_union(L, p, p - w - 1);
                    if (e_ne) uf_union(L, p, p - w + 1);
                }
            }
            // the stretch opening right of p: (y, x + 1) and the pixel above it both background
            if (!e_e && !e_ne) uf_union(rp, h + i, rt_run(bf, rk, rb, h, w64, y - 1, x + 1));
            // the one opening right of p in the row below (if the pixel below p is an edge pixel too -- a vertical
            // stroke -- that pixel's own stretch is this very one: leave it to it)
            if (!e_se && !e_e && !e_s) uf_union(rp, rt_run(bf, rk, rb, h, w64, y + 1, x + 1), h + i);
        } else {
            const int yy = i - ne + 1;                 // x = 0 of rows 1 .. h-1: always background
            uf_union(rp, yy, yy - 1);
        }
    }
}

// C' + D'. every run node at its root (one lookup from here on), every edge pixel at its root where Canny did not
// leave it there
__global__ __launch_bounds__(256) void flatten_runs_kernel(int h, int w, int32_t* __restrict__ labels, const FrameTab* __restrict__ tab,
                                                           const int32_t* __restrict__ elist, RunTab rt,
                                                           const int* __restrict__ canny_border_flag)
{
    int f, bx;
    list_frame_block(LIST_BLOCKS, f, bx);
    if (tab[f].runs_overflow) return;                  // this call is redone in the dense form (k_board_lines)
    const bool edges_flat = canny_border_flag != nullptr && canny_border_flag[f] == 0;
    const int ne = tab[f].n_edges;
    int32_t* L = labels + (size_t)f * h * w;
    const int32_t* E = elist + (size_t)f * h * w;
    int32_t* rp = rt.rp + (size_t)f * rt.rp_stride;
    for (int i = bx * 256 + threadIdx.x; i < ne + h; i += LIST_BLOCKS * 256) {
        rp[i] = uf_find(rp, i);                          // (a plain store of an ancestor: safe next to concurrent finds)
        if (!edges_flat && i < ne) { const int p = E[i]; L[p] = uf_find(L, p); }
    }
}

// D' + E'. top-level roots, then the outer-border list + bounding boxes (two kernels: the border pass needs every
// root's slot).  A background neighbour is in S0 iff its run's root is the root of node 0 (row 0 is all background).
__global__ __launch_bounds__(256) void roots_runs_kernel(int h, int w, const int32_t* __restrict__ labels,
                                                         int32_t* __restrict__ compid, FrameTab* __restrict__ tab, int maxc,
                                                         int32_t* __restrict__ roots, int32_t* __restrict__ aabb,
                                                         const int32_t* __restrict__ elist, RunTab rt)
{
    int f, bx;
    list_frame_block(LIST_BLOCKS, f, bx);
    if (tab[f].runs_overflow) return;                  // this call is redone in the dense form (k_board_lines)
    const int ne = tab[f].n_edges;
    const size_t off = (size_t)f * h * w;
    const int32_t* L = labels + off;
    const int32_t* E = elist + off;
    const int32_t* rb = rt.rowbase + (size_t)f * h;
    const int32_t* rp = rt.rp + (size_t)f * rt.rp_stride;
    const int root0 = rp[0];
    for (int i = bx * 256 + threadIdx.x; i < ne; i += LIST_BLOCKS * 256) {
        const int p = E[i];
        if (L[p] != p) continue;
        const int y = p / w;
        const int west = (i == rb[y]) ? y : h + i - 1;  // the west neighbour of a component's first pixel is background
        if (rp[west] != root0) continue;
        const int slot = atomicAdd(&tab[f].n_roots, 1);
        if (slot >= maxc) { tab[f].overflow = 1; compid[off + p] = -1; continue; }
        compid[off + p] = slot;
        roots[(size_t)f * maxc + slot] = p;
        int32_t* bb = aabb + ((size_t)f * maxc + slot) * 4;
        bb[0] = 0x7fffffff; bb[1] = -1; bb[2] = 0x7fffffff; bb[3] = -1;
    }
}

__global__ __launch_bounds__(256) void border_runs_kernel(int h, int w, const int32_t* __restrict__ labels,
                                                          const int32_t* __restrict__ compid, int maxc, FrameTab* __restrict__ tab,
                                                          int32_t* __restrict__ aabb, const int32_t* __restrict__ elist,
                                                          int32_t* __restrict__ blist, RunTab rt)
{
    int f, bx;
    list_frame_block(LIST_BLOCKS, f, bx);
    if (tab[f].runs_overflow) return;                  // this call is redone in the dense form (k_board_lines)
    const int ne = tab[f].n_edges, w64 = rt.w64;
    const size_t off = (size_t)f * h * w;
    const int32_t* L = labels + off;
    const int32_t* E = elist + off;
    int32_t* B = blist + off;
    const unsigned long long* bf = rt.bits + (size_t)f * h * w64;
    const uint16_t* rk = rt.rank + (size_t)f * h * w64;
    const int32_t* rb = rt.rowbase + (size_t)f * h;
    const int32_t* rp = rt.rp + (size_t)f * rt.rp_stride;
    const int root0 = rp[0];
    const int trips = (ne + LIST_BLOCKS * 256 - 1) / (LIST_BLOCKS * 256);      // uniform trip count
    for (int t = 0; t < trips; t++) {
        const int i = t * LIST_BLOCKS * 256 + bx * 256 + threadIdx.x;
        bool b = false;
        int p = 0, cslot = -1;
        if (i < ne) {
            p = E[i];
            const int y = p / w, x = p - y * w;
            // the four neighbours: edge bit, then the node of a background one (independent loads), then its root
            const size_t wi = (size_t)y * w64 + (x >> 6);
            const unsigned long long here = bf[wi];
            const bool e_w = (x & 63) ? (here >> ((x & 63) - 1)) & 1ull : rt_edge(bf, w64, y, x - 1);
            const bool e_e = ((x & 63) != 63) ? (here >> ((x & 63) + 1)) & 1ull : rt_edge(bf, w64, y, x + 1);
            const bool e_n = rt_edge(bf, w64, y - 1, x), e_s = rt_edge(bf, w64, y + 1, x);
            const int n_w = (i == rb[y]) ? y : h + i - 1;                     // the run that ends at p - 1 (if p - 1 is background)
            const int n_e = h + i;
            const int n_n = e_n ? 0 : rt_run(bf, rk, rb, h, w64, y - 1, x);
            const int n_s = e_s ? 0 : rt_run(bf, rk, rb, h, w64, y + 1, x);
            const int t_w = rp[e_w ? 0 : n_w], t_e = rp[e_e ? 0 : n_e], t_n = rp[n_n], t_s = rp[n_s];
            b = (!e_w && t_w == root0) || (!e_e && t_e == root0) || (!e_n && t_n == root0) || (!e_s && t_s == root0);
            if (b) {
                cslot = compid[off + L[p]];
                if ((unsigned)cslot >= (unsigned)maxc) b = false;
            }
        }
        // bounding boxes: reduce per component inside the wave, one lane updates the box and only where it grows
        {
            unsigned long long todo = __builtin_amdgcn_ballot_w64(b);
            const int y = p / w, x = p - y * w;
            const int lane = threadIdx.x & 63;
            while (todo) {
                const int lead = __builtin_ctzll(todo);
                const int s_lead = __builtin_amdgcn_readlane(cslot, lead);
                const bool mine = b && cslot == s_lead;
                todo &= ~__builtin_amdgcn_ballot_w64(mine);
                int mnx = mine ? x : 0x7fffffff, mxx = mine ? x : -1, mny = mine ? y : 0x7fffffff, mxy = mine ? y : -1;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    mnx = min(mnx, __shfl_xor(mnx, d)); mxx = max(mxx, __shfl_xor(mxx, d));
                    mny = min(mny, __shfl_xor(mny, d)); mxy = max(mxy, __shfl_xor(mxy, d));
                }
                if (lane == lead) {
                    int32_t* bb = aabb + ((size_t)f * maxc + s_lead) * 4;
                    if (mnx < uf_load(bb, 0)) atomicMin(bb + 0, mnx);
                    if (mxx > uf_load(bb, 1)) atomicMax(bb + 1, mxx);
                    if (mny < uf_load(bb, 2)) atomicMin(bb + 2, mny);
                    if (mxy > uf_load(bb, 3)) atomicMax(bb + 3, mxy);
                }
            }
        }
        const int slot = wave_append(&tab[f].n_border, b);
        if (b) B[slot] = p;
    }
}

// pack the first `k` entries of every frame's root / bounding-box table into dense arrays so the
// host needs one contiguous copy instead of a strided one
__global__ void pack_tables_kernel(const int32_t* __restrict__ roots, const int32_t* __restrict__ aabb, int maxc, int k, int n,
                                   int32_t* __restrict__ out /* n*k roots, then n*k*4 boxes */)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * k) return;
    const int f = i / k, s = i % k;
    out[i] = roots[(size_t)f * maxc + s];
    const int32_t* bb = aabb + ((size_t)f * maxc + s) * 4;
    int32_t* ob = out + (size_t)n * k + (size_t)i * 4;
    ob[0] = bb[0]; ob[1] = bb[1]; ob[2] = bb[2]; ob[3] = bb[3];
}

__global__ void pack_peaks_kernel(const int32_t* __restrict__ peaks, int k, int n, int32_t* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * k) return;
    const int f = i / k, s = i % k;
    out[2 * (size_t)i] = peaks[((size_t)f * PEAK_CAP + s) * 2];
    out[2 * (size_t)i + 1] = peaks[((size_t)f * PEAK_CAP + s) * 2 + 1];
}

// a border pixel in the middle of a straight run cannot be a hull vertex
__device__ __forceinline__ bool mid_of_run(const uint8_t* e, int p, int w)
{
    return (e[p - 1] && e[p + 1]) || (e[p - w] && e[p + w]) ||
           (e[p - w - 1] && e[p + w + 1]) || (e[p - w + 1] && e[p + w - 1]);
}

// the same test on the edge image as it came in (frame not cleared): a neighbour on the image frame counts as background
__device__ __forceinline__ bool mid_of_run_raw(const uint8_t* e, int p, int w, int h)
{
    const int y = p / w, x = p - y * w;
    auto at = [&](int dy, int dx) {
        const int yy = y + dy, xx = x + dx;
        return xx > 0 && xx < w - 1 && yy > 0 && yy < h - 1 && e[yy * w + xx] != 0;
    };
    return (at(0, -1) && at(0, 1)) || (at(-1, 0) && at(1, 0)) || (at(-1, -1) && at(1, 1)) || (at(-1, 1) && at(1, -1));
}

// ---- F. gather hull-candidate points of the components the host asked for ---------------
__global__ __launch_bounds__(256) void gather_points_kernel(const uint8_t* __restrict__ ez, int h, int w,
                                                            const int32_t* __restrict__ labels, const int32_t* __restrict__ compid,
                                                            int maxc, const uint8_t* __restrict__ want, int wpitch, const FrameTab* __restrict__ tab,
                                                            const int32_t* __restrict__ blist, int32_t* __restrict__ counter, int cap,
                                                            int32_t* __restrict__ pts /* x|y<<16, slot ; then frame */,
                                                            int keep_mid /* 1: every border pixel, not only hull candidates */,
                                                            int raw = 0 /* 1: `ez` is the edge image as it came in (frame not cleared) */)
{
    int f, bx;
    list_frame_block(LIST_BLOCKS, f, bx);
    const int nb = tab[f].n_border;
    const size_t off = (size_t)f * h * w;
    const int32_t* B = blist + off;
    const int trips = (nb + LIST_BLOCKS * 256 - 1) / (LIST_BLOCKS * 256);       // uniform trip count
    for (int t = 0; t < trips; t++) {
        const int i = t * LIST_BLOCKS * 256 + bx * 256 + threadIdx.x;
        bool take = false;
        int p = 0, slot = 0;
        if (i < nb) {
            p = B[i];
            slot = compid[off + labels[off + p]];
            take = want[(size_t)f * wpitch + slot] && (keep_mid || !(raw ? mid_of_run_raw(ez + off, p, w, h) : mid_of_run(ez + off, p, w)));
        }
        const int k = wave_append(counter, take);           // one atomic per wave
        if (take && k < cap) {
            const int y = p / w, x = p - y * w;
            pts[2 * (size_t)k] = x | (y << 16);
            pts[2 * (size_t)k + 1] = slot;
            pts[2 * (size_t)cap + k] = f;
        }
    }
}

// The same gather into one SEGMENT PER FRAME (round 4): frame f appends to seg[f * cap ...] through its own counter fcnt[f],
// so the waves of different frames no longer queue on ONE counter word (40 000 returning atomics per 128-frame call:
// 150 us of the kernel's 150).  compact_points_kernel then makes the dense list the host copies.
__global__ __launch_bounds__(256) void gather_segments_kernel(const uint8_t* __restrict__ ez, int h, int w,
                                                              const int32_t* __restrict__ labels, const int32_t* __restrict__ compid,
                                                              int maxc, const uint8_t* __restrict__ want, int wpitch, const FrameTab* __restrict__ tab,
                                                              const int32_t* __restrict__ blist, int32_t* __restrict__ fcnt, int cap,
                                                              int32_t* __restrict__ seg /* per frame: cap x (x|y<<16, slot) */,
                                                              int raw /* 1: `ez` is the edge image as it came in (frame not cleared) */)
{
    int f, bx;
    list_frame_block(LIST_BLOCKS, f, bx);
    const int nb = tab[f].n_border;
    const size_t off = (size_t)f * h * w;
    const int32_t* B = blist + off;
    int32_t* out = seg + (size_t)f * cap * 2;
    const int trips = (nb + LIST_BLOCKS * 256 - 1) / (LIST_BLOCKS * 256);       // uniform trip count
    for (int t = 0; t < trips; t++) {
        const int i = t * LIST_BLOCKS * 256 + bx * 256 + threadIdx.x;
        bool take = false;
        int p = 0, slot = 0;
        if (i < nb) {
            p = B[i];
            slot = compid[off + labels[off + p]];
            take = want[(size_t)f * wpitch + slot] && !(raw ? mid_of_run_raw(ez + off, p, w, h) : mid_of_run(ez + off, p, w));
        }
        const int k = wave_append(fcnt + f, take);
        if (take && k < cap) {
            const int y = p / w, x = p - y * w;
            out[2 * (size_t)k] = x | (y << 16);
            out[2 * (size_t)k + 1] = slot;
        }
    }
}

// segments -> the dense arrays gather_points_kernel writes: pts[2 k], pts[2 k + 1] = (x | y << 16, slot), pts[2 gcap + k] = frame;
// total[0] = points, total[1] = 1 if some frame had more points than its segment holds (the host then falls back)
__global__ __launch_bounds__(256) void compact_points_kernel(const int32_t* __restrict__ fcnt, int n, int cap, const int32_t* __restrict__ seg,
                                                             int gcap, int32_t* __restrict__ pts, int32_t* __restrict__ total)
{
    const int f = blockIdx.x;
    int base = 0, over = 0;
    for (int g = 0; g < n; g++) {                          // n <= a few hundred: every block adds up the counts before it
        const int c = fcnt[g];
        over |= c > cap;
        if (g < f) base += c < cap ? c : cap;
    }
    const int mine = fcnt[f] < cap ? fcnt[f] : cap;
    if (f == n - 1 && threadIdx.x == 0) { total[0] = base + mine; total[1] = over; }
    const int32_t* in = seg + (size_t)f * cap * 2;
    for (int i = threadIdx.x; i < mine; i += 256) {
        const int k = base + i;
        if (k < gcap) {
            pts[2 * (size_t)k] = in[2 * (size_t)i];
            pts[2 * (size_t)k + 1] = in[2 * (size_t)i + 1];
            pts[2 * (size_t)gcap + k] = f;
        }
    }
}

// every outer-border pixel of every map with the slot of its contour, at a position known in advance (survey only)
__global__ __launch_bounds__(256) void survey_points_kernel(int h, int w, const int32_t* __restrict__ labels, const int32_t* __restrict__ compid,
                                                            const FrameTab* __restrict__ tab, const int32_t* __restrict__ blist,
                                                            const int32_t* __restrict__ base, int32_t* __restrict__ pts /* x | y << 16, slot */)
{
    int f, bx;
    list_frame_block(LIST_BLOCKS, f, bx);
    const int nb = tab[f].n_border;
    const size_t off = (size_t)f * h * w;
    const int32_t* B = blist + off;
    int32_t* out = pts + 2 * (size_t)base[f];
    for (int i = bx * 256 + threadIdx.x; i < nb; i += LIST_BLOCKS * 256) {
        const int p = B[i];
        const int y = p / w, x = p - y * w;
        out[2 * (size_t)i] = x | (y << 16);
        out[2 * (size_t)i + 1] = compid[off + labels[off + p]];
    }
}

// ---- F2. vertex count of every external contour as CHAIN_APPROX_SIMPLE would store it -------
// (SfContours filters on `cont.shape[0]`, stone/sf_contours.py:197, 266.)  One lane per contour runs the library's
// border follower from the pixel the raster scan would start it at -- the first pixel of the component, which is the
// union-find root -- reading the edge bytes only: the marks the serial algorithm writes into its image steer where
// LATER borders start, never the path of the one being followed.  A vertex is stored where the step direction changes.
// step of direction d: 0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE (y grows downwards); two bits per direction in a register
// constant -- a table in memory would put a load with a per-lane index into every step of the follower
__device__ __forceinline__ int trace_dx(int d) { return (int)((0x901Au >> (2 * d)) & 3u) - 1; }     // 1, 1, 0, -1, -1, -1, 0, 1
__device__ __forceinline__ int trace_dy(int d) { return (int)((0xA901u >> (2 * d)) & 3u) - 1; }     // 0, -1, -1, -1, 0, 1, 1, 1

__global__ __launch_bounds__(64) void trace_count_kernel(const uint8_t* __restrict__ ez, int h, int w, FrameTab* __restrict__ tab,
                                                         const int32_t* __restrict__ roots, int maxc, int32_t* __restrict__ nvert)
{
    const int f = blockIdx.y, s = blockIdx.x * blockDim.x + threadIdx.x;
    const int nr = min(tab[f].n_roots, maxc);
    if (s >= nr) return;
    const uint8_t* e = ez + (size_t)f * h * w;
    const int p0 = roots[(size_t)f * maxc + s];
    int dir = 4, first = -1;
    do {                                             // clockwise from west: the first neighbour on the border
        dir = (dir - 1) & 7;
        const int q = p0 + trace_dy(dir) * w + trace_dx(dir);
        if (e[q]) { first = q; break; }
    } while (dir != 4);
    int count = 1;                                   // an isolated pixel is a contour of one point
    if (first >= 0) {
        count = 0;
        int cur = p0, prev_dir = dir ^ 4;
        const long long cap = 8ll * h * w;           // every (pixel, direction) pair at most once: far above any real border
        long long step = 0;
        for (; step < cap; step++) {
            int nxt;
            for (;;) {                               // counter-clockwise from the pixel we came from
                dir = (dir + 1) & 7;
                nxt = cur + trace_dy(dir) * w + trace_dx(dir);
                if (e[nxt]) break;
            }
            if (dir != prev_dir) { count++; prev_dir = dir; }
            if (nxt == p0 && cur == first) break;
            cur = nxt;
            dir = (dir + 4) & 7;
        }
        if (step >= cap) tab[f].overflow = 1;
    }
    nvert[(size_t)f * maxc + s] = count;
}

// The same follower with the edge map of its workgroup bit-packed in LDS (one workgroup per map: 379 x 379 pixels are
// 18 KB of bits).  A step reads the eight neighbour bits at once (independent LDS reads, one latency) and finds the next
// border pixel with a rotate + count-trailing-zeros instead of up to seven dependent loads from L2: the kernel is bound by
// its longest contour, so latency per step is what counts.
__global__ __launch_bounds__(256) void trace_count_lds_kernel(const uint8_t* __restrict__ ez, int h, int w, FrameTab* __restrict__ tab,
                                                              const int32_t* __restrict__ roots, int maxc, int32_t* __restrict__ nvert)
{
    extern __shared__ uint32_t bits[];                         // h rows of W dwords
    const int f = blockIdx.x;
    const int W = (w + 31) >> 5;
    const uint8_t* e = ez + (size_t)f * h * w;
    for (int i = threadIdx.x; i < h * W + 1; i += 256) bits[i] = 0;      // (+ one word the last window may touch)
    __syncthreads();
    // pack: the map as one flat byte array, eight coalesced loads in flight per thread; edge pixels are few, each sets its
    // bit with an LDS atomic
    const int npx = h * w;
    for (int base = 0; base < npx; base += 256 * 8) {
        uint8_t v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { const int i = base + k * 256 + threadIdx.x; v[k] = i < npx ? e[i] : 0; }
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (v[k]) {
                const int i = base + k * 256 + threadIdx.x, y = i / w, x = i - y * w;
                atomicOr(&bits[y * W + (x >> 5)], 1u << (x & 31));
            }
    }
    __syncthreads();
    const int nr = min(tab[f].n_roots, maxc);
    auto around = [&](int x, int y) {                          // bit d = neighbour in direction d
        // three rows, each a 3-bit window x-1 .. x+1 cut out of two consecutive words (the second word only matters when
        // the window straddles it; past the last word of a row it is never looked at)
        const int xl = x - 1, sh = xl & 31;
        const uint32_t* r = bits + (y - 1) * W + (xl >> 5);
        const uint32_t t0 = (uint32_t)(((unsigned long long)r[0] | ((unsigned long long)r[1] << 32)) >> sh) & 7u;
        const uint32_t t1 = (uint32_t)(((unsigned long long)r[W] | ((unsigned long long)r[W + 1] << 32)) >> sh) & 7u;
        const uint32_t t2 = (uint32_t)(((unsigned long long)r[2 * W] | ((unsigned long long)r[2 * W + 1] << 32)) >> sh) & 7u;
        return (t1 >> 2) | ((t0 >> 2) << 1) | (((t0 >> 1) & 1u) << 2) | ((t0 & 1u) << 3) | ((t1 & 1u) << 4) | ((t2 & 1u) << 5) |
               (((t2 >> 1) & 1u) << 6) | ((t2 >> 2) << 7);
    };
    for (int s = threadIdx.x; s < nr; s += 256) {
        const int p0 = roots[(size_t)f * maxc + s];
        const int y0 = p0 / w, x0 = p0 - y0 * w;
        uint32_t m = around(x0, y0);
        int count = 1;                                         // an isolated pixel is a contour of one point
        if (m) {
            int dir = 4;
            do { dir = (dir - 1) & 7; } while (!((m >> dir) & 1u));         // clockwise from west: the first neighbour on the border
            const int fx = x0 + trace_dx(dir), fy = y0 + trace_dy(dir);
            int x = x0, y = y0, prev_dir = dir ^ 4;
            count = 0;
            const long long cap = 8ll * h * w;
            long long step = 0;
            for (; step < cap; step++) {
                const uint32_t rot = ((m | (m << 8)) >> (dir + 1)) & 0xFFu;   // counter-clockwise from the pixel we came from
                dir = (dir + 1 + __builtin_ctz(rot)) & 7;
                const int nx = x + trace_dx(dir), ny = y + trace_dy(dir);
                if (dir != prev_dir) { count++; prev_dir = dir; }
                if (nx == x0 && ny == y0 && x == fx && y == fy) break;
                x = nx; y = ny;
                dir = (dir + 4) & 7;
                m = around(x, y);
            }
            if (step >= cap) tab[f].overflow = 1;
        }
        nvert[(size_t)f * maxc + s] = count;
    }
}

// ---- G. ghost image + Hough point list ------------------------------------------------------
__global__ __launch_bounds__(256) void ghost_list_kernel(int h, int w, const int32_t* __restrict__ labels,
                                                         const int32_t* __restrict__ compid, const int32_t* __restrict__ sel,
                                                         FrameTab* __restrict__ tab, const int32_t* __restrict__ blist, int pcap,
                                                         uint32_t* __restrict__ hpts, uint8_t* __restrict__ ghost)
{
    int f, bx;
    list_frame_block(LIST_BLOCKS, f, bx);
    if (!sel[f * 4 + 3]) return;
    const int nb = tab[f].n_border;
    const size_t off = (size_t)f * h * w;
    const int32_t* B = blist + off;
    const int s0 = sel[f * 4], s1 = sel[f * 4 + 1], s2 = sel[f * 4 + 2];
    for (int i = bx * 256 + threadIdx.x; i < nb; i += LIST_BLOCKS * 256) {
        const int p = B[i];
        const int slot = compid[off + labels[off + p]];
        if (slot != s0 && slot != s1 && slot != s2) continue;
        if (ghost) ghost[off + p] = 255;
        const int k = atomicAdd(&tab[f].n_hough_pts, 1);
        const int y = p / w, x = p - y * w;
        if (k < pcap) hpts[(size_t)f * pcap + k] = (uint32_t)x | ((uint32_t)y << 16);
        else tab[f].overflow = 1;
    }
}

// ---- H + I fused: the votes of rb theta rows (plus one halo row either side) as 16-bit counters packed two per
// LDS dword, peaks found on the slab itself: the accumulator never exists in HBM.  A cell's count is at most the
// number of ghost pixels on one discrete line (< w + h), so the halves cannot carry into each other.
__global__ __launch_bounds__(HOUGH_THREADS) void hough_vote_peaks_kernel(const uint32_t* __restrict__ hpts, FrameTab* __restrict__ tab,
                                                               int pcap, const float* __restrict__ trig /* cos[180], sin[180] */,
                                                               int numrho, int rb, int threshold,
                                                               int32_t* __restrict__ peaks /* f*PEAK_CAP*2 */)
{
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) uint32_t slab16[];
    const int f = blockIdx.y;
    const int nthreads = blockDim.x;                      // HOUGH_THREADS, or fewer for the small-batch launch
    const int n0 = blockIdx.x * rb;                       // first inner row
    const int stride = numrho + 2;                        // as the global accumulator had: a zero guard cell either side
    const int rowdw = (stride + 1) >> 1;                  // dwords per LDS row
    const int nrows = rb + 2;                             // LDS row j <-> theta row n0 - 1 + j
    for (int i = threadIdx.x; i < nrows * rowdw; i += nthreads) slab16[i] = 0u;
    __syncthreads();
    int npts = tab[f].n_hough_pts;
    if (npts > pcap) npts = pcap;
    const uint32_t* P = hpts + (size_t)f * pcap;
    const int half = (numrho - 1) / 2;
    const int j_lo = n0 == 0 ? 1 : 0;
    int j_hi = NUMANGLE - (n0 - 1);                       // exclusive: rows past theta 179 stay zero
    if (j_hi > nrows) j_hi = nrows;
    for (int i = threadIdx.x; i < npts; i += nthreads) {
        const uint32_t pk = P[i];
        const float xf = (float)(pk & 0xFFFF), yf = (float)(pk >> 16);
        for (int j = j_lo; j < j_hi; j++) {
            const int n = n0 - 1 + j;
            const float a = xf * trig[n];
            const float b = yf * trig[NUMANGLE + n];
            const float s = a + b;
            const int c = (int)rintf(s) + half + 1;       // cell index inside the row
            atomicAdd(&slab16[j * rowdw + (c >> 1)], 1u << (16 * (c & 1)));
        }
    }
    __syncthreads();
    const uint16_t* cnt = reinterpret_cast<const uint16_t*>(slab16);
    const int rowhw = 2 * rowdw;
    for (int k = 0; k < rb; k++) {
        const int n = n0 + k;
        if (n >= NUMANGLE) break;
        const uint16_t* row = cnt + (k + 1) * rowhw;
        // two cells per dword: nearly every cell is below the threshold, and then one load and two compares settle both
        const uint32_t* roww = slab16 + (k + 1) * rowdw;
        for (int d = threadIdx.x; d < rowdw; d += nthreads) {
            const uint32_t two = roww[d];
            if ((int)(two & 0xFFFFu) <= threshold && (int)(two >> 16) <= threshold) continue;
#pragma unroll
            for (int hlf = 0; hlf < 2; hlf++) {
                const int c = 2 * d + hlf, r = c - 1;           // cell c of the row <-> rho index r (cell 0 and the last are guards)
                if (r < 0 || r >= numrho) continue;
                const int v = hlf ? (int)(two >> 16) : (int)(two & 0xFFFFu);
                if (v <= threshold) continue;
                if (v > row[r] && v >= row[r + 2] && v > row[r + 1 - rowhw] && v >= row[r + 1 + rowhw]) {
                    const int i = atomicAdd(&tab[f].n_peaks, 1);
                    if (i < PEAK_CAP) {
                        peaks[((size_t)f * PEAK_CAP + i) * 2] = (n + 1) * stride + r + 1;
                        peaks[((size_t)f * PEAK_CAP + i) * 2 + 1] = v;
                    } else tab[f].overflow = 1;
                }
            }
        }
    }
}

template <typename F>
void parallel_for(int n, F fn) { ck_parallel_for(n, 16, fn); }

struct Comp {
    int root;       // first pixel (raster index) = discovery order key
    int slot;
    double ub;      // bounding-box area: upper bound of the minAreaRect area
    double area;    // exact area once known
    bool known;
};

}  // namespace

int k_board_lines(ck_ctx* ctx, const uint8_t* d_edges, int n, int h, int w, int hough_thresh,
                  float* lines, int cap, ck_board_result* res, uint8_t* d_ghost_out, const int* d_canny_border_flag)
{
    static const bool prof = getenv("CK_PROFILE_HOST") != nullptr;   // debugging aid: host-side lap times on stderr
    auto t_start = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!prof) return;
        (void)hipStreamSynchronize(ctx->stream);
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[board_lines] %-18s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_start).count());
        t_start = now;
    };
    const size_t fpx = (size_t)h * w, npx = fpx * n;
    int maxc = (int)(fpx / 4 + 1);
    if (maxc > MAXC_LIMIT) maxc = MAXC_LIMIT;
    const int pcap = (int)(fpx < (1u << 16) ? fpx : (fpx / 8 > (1u << 16) ? fpx / 8 : (1u << 16)));
    const int numrho = 2 * (w + h) + 1;
    const int stride = numrho + 2;
    if (w > 65535 || h > 65535) return ck_fail(ctx, CK_ERR_ARG, "image side > 65535");

    CK_TRY(ck_ensure(ctx, ctx->ghost, npx));                       // ez
    CK_TRY(ck_ensure(ctx, ctx->labels, npx * 4));
    CK_TRY(ck_ensure(ctx, ctx->labels2, npx * 4));                 // compid (only read at roots)
    CK_TRY(ck_ensure(ctx, ctx->lists, npx * 8));                   // edge list + border list
    const size_t tab_bytes = sizeof(FrameTab) * (size_t)n;
    const size_t sel_bytes = sizeof(int32_t) * 4 * (size_t)n;
    CK_TRY(ck_ensure(ctx, ctx->misc, tab_bytes + sel_bytes + 4 * NUMANGLE * 2 + 64));
    CK_TRY(ck_ensure(ctx, ctx->comp, (size_t)n * maxc * (4 + 16 + 1)));
    uint8_t* ez = (uint8_t*)ctx->ghost.p;
    int32_t* L = (int32_t*)ctx->labels.p;
    int32_t* compid = (int32_t*)ctx->labels2.p;
    int32_t* elist = (int32_t*)ctx->lists.p;
    int32_t* blist = elist + npx;
    FrameTab* d_tab = (FrameTab*)ctx->misc.p;
    int32_t* d_sel = (int32_t*)((char*)ctx->misc.p + tab_bytes);
    float* d_trig = (float*)((char*)ctx->misc.p + tab_bytes + sel_bytes);
    int32_t* d_roots = (int32_t*)ctx->comp.p;
    int32_t* d_aabb = d_roots + (size_t)n * maxc;
    uint8_t* d_want = (uint8_t*)(d_aabb + (size_t)n * maxc * 4);

    const dim3 lgrid = list_grid(LIST_BLOCKS, n), lblock(256);
    // the dword path needs 4-byte aligned rows: w % 4 == 0 and an aligned base pointer
    const bool dwords = (w & 3) == 0 && ((uintptr_t)d_edges & 3) == 0;
    bool run_table = dwords && w <= 4096;
    auto launch_ccl = [&](bool runs) -> int {
        TimeScope ts(ctx, "ccl");
        CK_HIP(ctx, hipMemsetAsync(d_tab, 0, tab_bytes, ctx->stream));
        if (runs) {
            // run-table form: no dense parent image.  Reuse of Canny's component roots (ck_board_detect only) needs this
            // form's row kernel, which preserves them.
            const int* kflag = d_canny_border_flag;
            RunTab rt;
            rt.w64 = (w + 63) / 64;
            // run nodes for a quarter of the pixels (at least 64 K): a frame with more edge pixels than that sends the
            // call to the dense form below
            const size_t cap_e = std::min(fpx, std::max(fpx / 4, (size_t)1 << 16));
            rt.cap_e = (int)cap_e;
            rt.rp_stride = cap_e + (size_t)h;
            const size_t bits_b = (size_t)n * h * rt.w64 * 8, rank_b = (size_t)n * h * rt.w64 * 2, rb_b = (size_t)n * h * 4;
            const size_t rank_o = bits_b, rb_o = (rank_o + rank_b + 15) & ~(size_t)15, rp_o = (rb_o + rb_b + 15) & ~(size_t)15;
            CK_TRY(ck_ensure(ctx, ctx->runs, rp_o + (size_t)n * rt.rp_stride * 4));
            rt.bits = (unsigned long long*)ctx->runs.p;
            rt.rank = (uint16_t*)((char*)ctx->runs.p + rank_o);
            rt.rowbase = (int32_t*)((char*)ctx->runs.p + rb_o);
            rt.rp = (int32_t*)((char*)ctx->runs.p + rp_o);
            if (w <= 1024)
                hipLaunchKernelGGL(prep_runs_kernel<4>, dim3((h + 3) / 4, n), dim3(256), 0, ctx->stream, d_edges, h, w, CCL_RUNS_EZ ? ez : (uint8_t*)nullptr, L, d_tab, elist, rt, kflag);
            else if (w <= 2048)
                hipLaunchKernelGGL(prep_runs_kernel<8>, dim3((h + 3) / 4, n), dim3(256), 0, ctx->stream, d_edges, h, w, CCL_RUNS_EZ ? ez : (uint8_t*)nullptr, L, d_tab, elist, rt, kflag);
            else
                hipLaunchKernelGGL(prep_runs_kernel<16>, dim3((h + 3) / 4, n), dim3(256), 0, ctx->stream, d_edges, h, w, CCL_RUNS_EZ ? ez : (uint8_t*)nullptr, L, d_tab, elist, rt, kflag);
            hipLaunchKernelGGL(link_runs_kernel, lgrid, lblock, 0, ctx->stream, h, w, L, (const FrameTab*)d_tab, (const int32_t*)elist, rt, kflag);
            hipLaunchKernelGGL(flatten_runs_kernel, lgrid, lblock, 0, ctx->stream, h, w, L, (const FrameTab*)d_tab, (const int32_t*)elist, rt, kflag);
            hipLaunchKernelGGL(roots_runs_kernel, lgrid, lblock, 0, ctx->stream, h, w, (const int32_t*)L, compid, d_tab, maxc,
                               d_roots, d_aabb, (const int32_t*)elist, rt);
            hipLaunchKernelGGL(border_runs_kernel, lgrid, lblock, 0, ctx->stream, h, w, (const int32_t*)L, (const int32_t*)compid, maxc,
                               d_tab, d_aabb, (const int32_t*)elist, blist, rt);
        } else {
            // dense form: every label rebuilt from the edge map (Canny's are not reused)
            if (dwords)
                hipLaunchKernelGGL(prep_rows4_kernel, dim3((h + 3) / 4, n), dim3(256), 0, ctx->stream, d_edges, h, w, ez, L, d_tab, elist);
            else
                hipLaunchKernelGGL(prep_rows_kernel, dim3((h + 3) / 4, n), dim3(256), 0, ctx->stream, d_edges, h, w, ez, L, d_tab, elist);
            hipLaunchKernelGGL(link_list_kernel, lgrid, lblock, 0, ctx->stream, (const uint8_t*)ez, h, w, L,
                               (const FrameTab*)d_tab, (const int32_t*)elist, (const int*)nullptr);
            hipLaunchKernelGGL(flatten_list_kernel, lgrid, lblock, 0, ctx->stream, (const uint8_t*)ez, h, w, L,
                               (const FrameTab*)d_tab, (const int32_t*)elist, (const int*)nullptr);
            hipLaunchKernelGGL(roots_list_kernel, lgrid, lblock, 0, ctx->stream, h, w, (const int32_t*)L, compid, d_tab, maxc,
                               d_roots, d_aabb, (const int32_t*)elist);
            hipLaunchKernelGGL(border_list_kernel, lgrid, lblock, 0, ctx->stream, (const uint8_t*)ez, h, w, (const int32_t*)L,
                               (const int32_t*)compid, maxc, d_tab, d_aabb, (const int32_t*)elist, blist);
        }
        CK_HIP(ctx, hipGetLastError());
        return CK_OK;
    };
    CK_TRY(launch_ccl(run_table));

    lap("ccl kernels");
    // ---- host: component tables (one strided copy each) -------------------------------------
    std::vector<FrameTab> tab((size_t)n);
    CK_HIP(ctx, hipMemcpyAsync(tab.data(), d_tab, tab_bytes, hipMemcpyDeviceToHost, ctx->stream));
    CK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (run_table) {
        bool redo = false;
        for (int f = 0; f < n; f++) redo = redo || tab[f].runs_overflow;
        if (redo) {                                       // an edge map denser than the run nodes provided for: dense form
            run_table = false;
            CK_TRY(launch_ccl(false));
            CK_HIP(ctx, hipMemcpyAsync(tab.data(), d_tab, tab_bytes, hipMemcpyDeviceToHost, ctx->stream));
            CK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    int nc_max = 0;
    for (int f = 0; f < n; f++) {
        if (tab[f].overflow) return ck_fail(ctx, CK_ERR_CAPACITY, "frame %d: more than %d external contours", f, maxc);
        nc_max = std::max(nc_max, tab[f].n_roots);
    }
    std::vector<std::vector<Comp>> comps((size_t)n);
    {
        std::vector<int32_t> hroots((size_t)n * nc_max), haabb((size_t)n * nc_max * 4);
        if (nc_max) {
            const size_t cnt = (size_t)n * nc_max;
            CK_TRY(ck_ensure(ctx, ctx->pts, cnt * 20 + 64));
            int32_t* d_pack = (int32_t*)ctx->pts.p;
            hipLaunchKernelGGL(pack_tables_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const int32_t*)d_roots, (const int32_t*)d_aabb, maxc, nc_max, n, d_pack);
            CK_HIP(ctx, hipMemcpyAsync(hroots.data(), d_pack, cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
            CK_HIP(ctx, hipMemcpyAsync(haabb.data(), d_pack + cnt, cnt * 16, hipMemcpyDeviceToHost, ctx->stream));
            CK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        for (int f = 0; f < n; f++) {
            const int nc = tab[f].n_roots;
            res[f].status = nc == 0 ? CK_BOARD_NO_CONTOUR : CK_BOARD_LINES;
            res[f].n_contours = nc; res[f].n_lines = 0; res[f].reserved = 0; res[f].biggest_area = 0;
            auto& cv = comps[f];
            cv.resize((size_t)nc);
            const int32_t* hr = hroots.data() + (size_t)f * nc_max;
            const int32_t* hb = haabb.data() + (size_t)f * nc_max * 4;
            for (int s = 0; s < nc; s++) {
                const double dx = (double)hb[4 * s + 1] - hb[4 * s], dy = (double)hb[4 * s + 3] - hb[4 * s + 2];
                cv[s] = { hr[s], s, dx * dy, 0.0, dx * dy == 0.0 };
            }
        }
    }

    lap("tables d2h");
    // ---- exact areas for the components that can still reach the top three -------------------
    const int gcap = (int)(npx < (1u << 22) ? npx : (1u << 22));     // points per gather round
    CK_TRY(ck_ensure(ctx, ctx->pts, (size_t)gcap * 12 + 64));
    int32_t* d_pts = (int32_t*)ctx->pts.p;
    int32_t* d_counter = d_pts + (size_t)gcap * 3;
    std::vector<uint8_t> want((size_t)n * std::max(nc_max, 1));
    auto third_best = [](const std::vector<Comp>& cv) {
        double b[3] = { -1, -1, -1 };
        for (const Comp& c : cv) if (c.known) {
            double a = c.area;
            for (int i = 0; i < 3; i++) if (a > b[i]) std::swap(a, b[i]);
        }
        return b[2];          // -1 while fewer than three areas are known
    };
    for (int round = 0; round < 3 && nc_max > 0; round++) {
        bool any = false;
        std::fill(want.begin(), want.end(), 0);
        for (int f = 0; f < n; f++) {
            auto& cv = comps[f];
            if (cv.empty()) continue;
            uint8_t* wf = want.data() + (size_t)f * nc_max;
            if (round == 0) {
                std::vector<int> idx;
                for (int s = 0; s < (int)cv.size(); s++) if (!cv[s].known) idx.push_back(s);
                const int k = std::min<int>(16, (int)idx.size());
                std::partial_sort(idx.begin(), idx.begin() + k, idx.end(), [&](int a, int b) { return cv[a].ub > cv[b].ub; });
                for (int i = 0; i < k; i++) { wf[idx[i]] = 1; any = true; }
            } else {
                const double third = third_best(cv);
                for (auto& c : cv)
                    if (!c.known && c.ub * (1.0 + 1e-5) >= third) { wf[c.slot] = 1; any = true; }
            }
        }
        if (!any) break;
        if (round == 2) return ck_fail(ctx, CK_ERR_STATE, "contour selection did not converge");
        {
            TimeScope ts(ctx, "contour_gather");
            CK_HIP(ctx, hipMemcpyAsync(d_want, want.data(), (size_t)n * nc_max, hipMemcpyHostToDevice, ctx->stream));
            // one segment per frame + compaction; a frame with more points than its segment holds (never seen: a
            // segment is gcap / n >= 32 768 points at 1080p) sends the round through the single-counter kernel instead
            const int segcap = gcap / n;
            CK_TRY(ck_ensure(ctx, ctx->accum, (size_t)n * segcap * 8 + (size_t)n * 4 + 64));
            int32_t* d_seg = (int32_t*)ctx->accum.p;
            int32_t* d_fcnt = d_seg + (size_t)n * segcap * 2;
            CK_HIP(ctx, hipMemsetAsync(d_fcnt, 0, (size_t)n * 4, ctx->stream));
            CK_HIP(ctx, hipMemsetAsync(d_counter, 0, 8, ctx->stream));
            // (run-table form: the cleared-frame bytes were not written; the hull-candidate test reads the edge image as it came in)
            const bool raw_edges = run_table && !CCL_RUNS_EZ;
            const uint8_t* e_img = raw_edges ? d_edges : (const uint8_t*)ez;
            hipLaunchKernelGGL(gather_segments_kernel, lgrid, lblock, 0, ctx->stream, e_img, h, w,
                               (const int32_t*)L, (const int32_t*)compid, maxc, (const uint8_t*)d_want, nc_max,
                               (const FrameTab*)d_tab, (const int32_t*)blist, d_fcnt, segcap, d_seg, (int)raw_edges);
            hipLaunchKernelGGL(compact_points_kernel, dim3(n), dim3(256), 0, ctx->stream, (const int32_t*)d_fcnt, n, segcap,
                               (const int32_t*)d_seg, gcap, d_pts, d_counter);
            CK_HIP(ctx, hipGetLastError());
        }
        int npts = 0;
        {
            int two[2] = {0, 0};
            CK_HIP(ctx, hipMemcpyAsync(two, d_counter, 8, hipMemcpyDeviceToHost, ctx->stream));
            CK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            npts = two[0];
            if (two[1]) {                                   // a segment overflowed: the whole round again, one counter for all
                TimeScope ts(ctx, "contour_gather");
                CK_HIP(ctx, hipMemsetAsync(d_counter, 0, 4, ctx->stream));
                const bool raw_edges = run_table && !CCL_RUNS_EZ;
                hipLaunchKernelGGL(gather_points_kernel, lgrid, lblock, 0, ctx->stream, raw_edges ? d_edges : (const uint8_t*)ez, h, w,
                                   (const int32_t*)L, (const int32_t*)compid, maxc, (const uint8_t*)d_want, nc_max,
                                   (const FrameTab*)d_tab, (const int32_t*)blist, d_counter, gcap, d_pts, 0, (int)raw_edges);
                CK_HIP(ctx, hipGetLastError());
                CK_HIP(ctx, hipMemcpyAsync(&npts, d_counter, 4, hipMemcpyDeviceToHost, ctx->stream));
                CK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            }
        }
        if (npts > gcap) return ck_fail(ctx, CK_ERR_CAPACITY, "too many contour points (%d > %d)", npts, gcap);
        std::vector<int32_t> hp((size_t)npts * 2), hf((size_t)npts);
        if (npts) {
            CK_HIP(ctx, hipMemcpyAsync(hp.data(), d_pts, (size_t)npts * 8, hipMemcpyDeviceToHost, ctx->stream));
            CK_HIP(ctx, hipMemcpyAsync(hf.data(), d_pts + (size_t)gcap * 2, (size_t)npts * 4, hipMemcpyDeviceToHost, ctx->stream));
            CK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        lap("  gather kernel+d2h");
        // bucket by (frame, slot)
        std::vector<std::vector<std::vector<int32_t>>> bucket((size_t)n);
        for (int f = 0; f < n; f++) bucket[f].resize(comps[f].size());
        for (int i = 0; i < npts; i++) {
            auto& b = bucket[hf[i]][hp[2 * (size_t)i + 1]];
            b.push_back(hp[2 * (size_t)i] & 0xFFFF);
            b.push_back(hp[2 * (size_t)i] >> 16);
        }
        lap("  bucket");
        parallel_for(n, [&](int f) {
            auto& cv = comps[f];
            for (size_t s = 0; s < cv.size(); s++) {
                if (!want[(size_t)f * nc_max + s]) continue;
                float wh[2];
                ck_min_area_rect(bucket[f][s].data(), (int)(bucket[f][s].size() / 2), wh);
                cv[s].area = (double)wh[0] * (double)wh[1];
                cv[s].known = true;
            }
        });
        lap("  hull+calipers");
    }

    lap("gather+calipers");
    // ---- selection: bisect.insort order = (area ascending, discovery order descending) ------
    std::vector<int32_t> sel((size_t)n * 4, -1);
    bool any_go = false;
    for (int f = 0; f < n; f++) {
        auto& cv = comps[f];
        sel[(size_t)f * 4 + 3] = 0;
        if (cv.empty()) continue;
        std::vector<const Comp*> known;
        for (const Comp& c : cv) if (c.known) known.push_back(&c);
        std::sort(known.begin(), known.end(), [](const Comp* a, const Comp* b) {
            if (a->area != b->area) return a->area > b->area;
            return a->root < b->root;                    // raster-earlier contour ranks higher
        });
        res[f].biggest_area = known[0]->area;
        const double frame_area = (double)h * (double)w;
        if (!(frame_area / 3 < known[0]->area)) { res[f].status = CK_BOARD_TOO_SMALL; continue; }
        for (int i = 0; i < 3 && i < (int)known.size(); i++) sel[(size_t)f * 4 + i] = known[i]->slot;
        sel[(size_t)f * 4 + 3] = 1;
        any_go = true;
    }
    if (d_ghost_out) CK_HIP(ctx, hipMemsetAsync(d_ghost_out, 0, npx, ctx->stream));
    if (!any_go) return CK_OK;

    // ---- ghost, Hough --------------------------------------------------------------------------
    std::vector<float> trig(2 * NUMANGLE);
    {
        const float theta = (float)(3.1415926535897932384626433832795 / 180);
        float ang = 0.f;
        for (int k = 0; k < NUMANGLE; ang += theta, k++) {
            trig[NUMANGLE + k] = (float)(sin((double)ang) * 1.f);
            trig[k] = (float)(cos((double)ang) * 1.f);
        }
    }
    CK_TRY(ck_ensure(ctx, ctx->peaks, (size_t)n * PEAK_CAP * 8 + (size_t)n * pcap * 4));
    int32_t* d_peaks = (int32_t*)ctx->peaks.p;
    uint32_t* d_hpts = (uint32_t*)(d_peaks + (size_t)n * PEAK_CAP * 2);
    {
        TimeScope ts(ctx, "ghost");
        CK_HIP(ctx, hipMemcpyAsync(d_sel, sel.data(), sel_bytes, hipMemcpyHostToDevice, ctx->stream));
        CK_HIP(ctx, hipMemcpyAsync(d_trig, trig.data(), trig.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(ghost_list_kernel, lgrid, lblock, 0, ctx->stream, h, w, (const int32_t*)L, (const int32_t*)compid,
                           (const int32_t*)d_sel, d_tab, (const int32_t*)blist, pcap, d_hpts, d_ghost_out);
        CK_HIP(ctx, hipGetLastError());
    }
    {
        TimeScope ts(ctx, "hough_vote");
        const size_t row_bytes = (size_t)((stride + 1) / 2) * 4;           // 16-bit counters, two per dword
        int rb = (int)((144 * 1024) / row_bytes) - 2;                      // inner rows per workgroup (+ 2 halo rows)
        if (rb > 10) rb = 10;
        if (rb < 1) return ck_fail(ctx, CK_ERR_ARG, "image too large for the Hough LDS slab");
        int threads = HOUGH_THREADS;
        // A call of a few frames (the hold-off-aware fold's windows, a live finder's single frame) is a latency matter, and
        // next to the classifier -- two workgroups of 79 KB on every CU -- a workgroup that wants a whole CU's LDS waits until
        // that kernel's grid drains: 1.5 ms of a 16-frame call's 3.0 (tools/board_call_latency.py).  Small batches take
        // slabs that fit the hole ONE retiring classifier workgroup leaves (<= 72 KB, 512 threads): same peaks (rows with
        // their halo, sorted on the host), more workgroups re-reading the point list.
        if (n <= HOUGH_SMALL_N) {
            const int rs = (int)((72 * 1024) / row_bytes) - 2;
            if (rs >= 1) { rb = rs < rb ? rs : rb; threads = HOUGH_THREADS < 512 ? HOUGH_THREADS : 512; }
        }
        CK_HIP(ctx, hipFuncSetAttribute((const void*)hough_vote_peaks_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)(144 * 1024)));
        hipLaunchKernelGGL(hough_vote_peaks_kernel, dim3((NUMANGLE + rb - 1) / rb, n), dim3(threads), (rb + 2) * row_bytes, ctx->stream,
                           (const uint32_t*)d_hpts, d_tab, pcap, (const float*)d_trig, numrho, rb, hough_thresh, d_peaks);
        CK_HIP(ctx, hipGetLastError());
    }
    lap("ghost+hough");
    CK_HIP(ctx, hipMemcpyAsync(tab.data(), d_tab, tab_bytes, hipMemcpyDeviceToHost, ctx->stream));
    CK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int np_max = 0;
    for (int f = 0; f < n; f++) {
        if (tab[f].overflow) return ck_fail(ctx, CK_ERR_CAPACITY, "frame %d: Hough point/peak capacity exceeded", f);
        if (res[f].status == CK_BOARD_LINES) np_max = std::max(np_max, tab[f].n_peaks);
    }
    if (!np_max) return CK_OK;
    std::vector<int32_t> pk((size_t)n * np_max * 2);
    {
        const size_t cnt = (size_t)n * np_max;
        CK_TRY(ck_ensure(ctx, ctx->pts, cnt * 8 + 64));
        hipLaunchKernelGGL(pack_peaks_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const int32_t*)d_peaks, np_max, n, (int32_t*)ctx->pts.p);
        CK_HIP(ctx, hipMemcpyAsync(pk.data(), ctx->pts.p, cnt * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    CK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const float theta = (float)(3.1415926535897932384626433832795 / 180);
    const double scale = 1. / (numrho + 2);
    std::vector<int> order;
    for (int f = 0; f < n; f++) {
        if (res[f].status != CK_BOARD_LINES) continue;
        const int np = tab[f].n_peaks;
        res[f].n_lines = np;
        if (!np) continue;
        const int32_t* pf = pk.data() + (size_t)f * np_max * 2;
        order.resize((size_t)np);
        for (int i = 0; i < np; i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](int a, int b) {
            if (pf[2 * a + 1] != pf[2 * b + 1]) return pf[2 * a + 1] > pf[2 * b + 1];
            return pf[2 * a] < pf[2 * b];
        });
        for (int i = 0; i < np && i < cap; i++) {
            const int idx = pf[2 * order[i]];
            const int nn = (int)floor(idx * scale) - 1;
            const int rr = idx - (nn + 1) * (numrho + 2) - 1;
            lines[((size_t)f * cap + i) * 2] = (rr - (numrho - 1) * 0.5f) * 1.f;
            lines[((size_t)f * cap + i) * 2 + 1] = 0.f + nn * theta;
        }
    }
    lap("peaks d2h+sort");
    return CK_OK;
}


// ---- external contours of a batch of (small) edge maps, handed to the host -------------------------------------
// The survey SfContours.find_stones needs of each edge map (stone/sf_contours.py:78-83, 264-270): every RETR_EXTERNAL
// contour with the length of its CHAIN_APPROX_SIMPLE vertex list and its outer-border pixels (the set drawContours
// paints with thickness 1, and a superset of the hull vertices), in the order cv2 hands contours back: last found
// first.  Same labelling kernels as the board path above; the follower (F2) only counts.
int k_contour_survey(ck_ctx* ctx, const uint8_t* d_edges, int n, int h, int w, std::vector<std::vector<CkContour>>& out)
{
    const size_t fpx = (size_t)h * w, npx = fpx * n;
    int maxc = (int)(fpx / 4 + 1);
    if (maxc > MAXC_LIMIT) maxc = MAXC_LIMIT;
    if (w > 65535 || h > 65535) return ck_fail(ctx, CK_ERR_ARG, "image side > 65535");
    if (h < 3 || w < 3) return ck_fail(ctx, CK_ERR_ARG, "edge map smaller than 3x3");
    CK_TRY(ck_ensure(ctx, ctx->ghost, npx));
    CK_TRY(ck_ensure(ctx, ctx->labels, npx * 4));
    CK_TRY(ck_ensure(ctx, ctx->labels2, npx * 4));
    CK_TRY(ck_ensure(ctx, ctx->lists, npx * 8));
    const size_t tab_bytes = sizeof(FrameTab) * (size_t)n;
    CK_TRY(ck_ensure(ctx, ctx->misc, tab_bytes + 64));
    CK_TRY(ck_ensure(ctx, ctx->comp, (size_t)n * maxc * (4 + 16 + 1 + 4)));
    uint8_t* ez = (uint8_t*)ctx->ghost.p;
    int32_t* L = (int32_t*)ctx->labels.p;
    int32_t* compid = (int32_t*)ctx->labels2.p;
    int32_t* elist = (int32_t*)ctx->lists.p;
    int32_t* blist = elist + npx;
    FrameTab* d_tab = (FrameTab*)ctx->misc.p;
    int32_t* d_roots = (int32_t*)ctx->comp.p;
    int32_t* d_aabb = d_roots + (size_t)n * maxc;
    int32_t* d_nvert = d_aabb + (size_t)n * maxc * 4;
    const dim3 lgrid = list_grid(LIST_BLOCKS, n), lblock(256);
    {
        TimeScope ts(ctx, "survey_ccl");
        CK_HIP(ctx, hipMemsetAsync(d_tab, 0, tab_bytes, ctx->stream));
        const bool dwords = (w & 3) == 0 && ((uintptr_t)d_edges & 3) == 0;
        if (dwords)
            hipLaunchKernelGGL(prep_rows4_kernel, dim3((h + 3) / 4, n), dim3(256), 0, ctx->stream, d_edges, h, w, ez, L, d_tab, elist);
        else
            hipLaunchKernelGGL(prep_rows_kernel, dim3((h + 3) / 4, n), dim3(256), 0, ctx->stream, d_edges, h, w, ez, L, d_tab, elist);
        hipLaunchKernelGGL(link_list_kernel, lgrid, lblock, 0, ctx->stream, (const uint8_t*)ez, h, w, L,
                           (const FrameTab*)d_tab, (const int32_t*)elist, (const int*)nullptr);
        hipLaunchKernelGGL(flatten_list_kernel, lgrid, lblock, 0, ctx->stream, (const uint8_t*)ez, h, w, L,
                           (const FrameTab*)d_tab, (const int32_t*)elist, (const int*)nullptr);
        hipLaunchKernelGGL(roots_list_kernel, lgrid, lblock, 0, ctx->stream, h, w, (const int32_t*)L, compid, d_tab, maxc,
                           d_roots, d_aabb, (const int32_t*)elist);
        hipLaunchKernelGGL(border_list_kernel, lgrid, lblock, 0, ctx->stream, (const uint8_t*)ez, h, w, (const int32_t*)L,
                           (const int32_t*)compid, maxc, d_tab, d_aabb, (const int32_t*)elist, blist);
        CK_HIP(ctx, hipGetLastError());
    }
    std::vector<FrameTab> tab((size_t)n);
    CK_HIP(ctx, hipMemcpyAsync(tab.data(), d_tab, tab_bytes, hipMemcpyDeviceToHost, ctx->stream));
    CK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int nc_max = 0;
    size_t nb_total = 0;
    for (int f = 0; f < n; f++) {
        if (tab[f].overflow) return ck_fail(ctx, CK_ERR_CAPACITY, "edge map %d: more than %d external contours", f, maxc);
        nc_max = std::max(nc_max, tab[f].n_roots);
        nb_total += (size_t)tab[f].n_border;
    }
    out.assign((size_t)n, {});
    if (!nc_max) return CK_OK;
    {
        TimeScope ts(ctx, "survey_trace");
        const size_t bit_bytes = (size_t)h * ((w + 31) / 32) * 4 + 4;
        if (bit_bytes <= 64 * 1024)                            // goban-sized maps: follow the borders in LDS
            hipLaunchKernelGGL(trace_count_lds_kernel, dim3(n), dim3(256), bit_bytes, ctx->stream, (const uint8_t*)ez, h, w, d_tab,
                               (const int32_t*)d_roots, maxc, d_nvert);
        else
            hipLaunchKernelGGL(trace_count_kernel, dim3((nc_max + 63) / 64, n), dim3(64), 0, ctx->stream, (const uint8_t*)ez, h, w, d_tab,
                               (const int32_t*)d_roots, maxc, d_nvert);
        CK_HIP(ctx, hipGetLastError());
    }
    // roots and vertex counts (strided tables -> dense), then every outer-border pixel with its contour slot: the border
    // list of a map is already dense, so pixel i of map f lands at base[f] + i -- no counter, and the host gets the maps
    // as contiguous segments it can bucket in parallel
    const size_t cnt = (size_t)n * nc_max;
    std::vector<int32_t> hroots(cnt), hnvert(cnt);
    CK_HIP(ctx, hipMemcpy2DAsync(hroots.data(), (size_t)nc_max * 4, d_roots, (size_t)maxc * 4, (size_t)nc_max * 4, (size_t)n,
                                 hipMemcpyDeviceToHost, ctx->stream));
    CK_HIP(ctx, hipMemcpy2DAsync(hnvert.data(), (size_t)nc_max * 4, d_nvert, (size_t)maxc * 4, (size_t)nc_max * 4, (size_t)n,
                                 hipMemcpyDeviceToHost, ctx->stream));
    std::vector<int32_t> base((size_t)n + 1, 0);
    for (int f = 0; f < n; f++) base[f + 1] = base[f] + tab[f].n_border;
    const size_t npts = nb_total;
    CK_TRY(ck_ensure(ctx, ctx->pts, npts * 8 + (size_t)(n + 1) * 4 + 64));
    int32_t* d_pts = (int32_t*)ctx->pts.p;
    int32_t* d_base = d_pts + npts * 2;
    CK_TRY(ck_ensure_pinned(ctx, npts * 8 + 64, 1));
    int32_t* hp = (int32_t*)ctx->host_pinned2;
    {
        TimeScope ts(ctx, "survey_gather");
        CK_HIP(ctx, hipMemcpyAsync(d_base, base.data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(survey_points_kernel, lgrid, lblock, 0, ctx->stream, h, w, (const int32_t*)L, (const int32_t*)compid,
                           (const FrameTab*)d_tab, (const int32_t*)blist, (const int32_t*)d_base, d_pts);
        CK_HIP(ctx, hipGetLastError());
    }
    if (npts) CK_HIP(ctx, hipMemcpyAsync(hp, d_pts, npts * 8, hipMemcpyDeviceToHost, ctx->stream));
    CK_HIP(ctx, hipMemcpyAsync(tab.data(), d_tab, tab_bytes, hipMemcpyDeviceToHost, ctx->stream));
    CK_HIP(ctx, hipStreamSynchronize(ctx->stream));           // (base is a local: its upload is done too)
    for (int f = 0; f < n; f++)
        if (tab[f].overflow) return ck_fail(ctx, CK_ERR_STATE, "edge map %d: the border follower did not close", f);
    // per map: contours in cv2 order (root descending = reverse discovery), border pixels bucketed by counting
    parallel_for(n, [&](int f) {
        const int nc = tab[f].n_roots;
        if (!nc) return;
        std::vector<int> order((size_t)nc), pos((size_t)nc), fill((size_t)nc, 0);
        for (int s = 0; s < nc; s++) order[s] = s;
        const int32_t* hr = hroots.data() + (size_t)f * nc_max;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return hr[a] > hr[b]; });
        auto& cv = out[f];
        cv.resize((size_t)nc);
        for (int k = 0; k < nc; k++) {
            const int s = order[k];
            pos[s] = k;
            cv[k].root = hr[s];
            cv[k].nvert = hnvert[(size_t)f * nc_max + s];
        }
        const int32_t* seg = hp + 2 * (size_t)base[f];
        const int nb = base[f + 1] - base[f];
        for (int i = 0; i < nb; i++) fill[seg[2 * i + 1]]++;
        for (int s = 0; s < nc; s++) { cv[pos[s]].pts.resize(2 * (size_t)fill[s]); fill[s] = 0; }
        for (int i = 0; i < nb; i++) {
            const int s = seg[2 * i + 1];
            int32_t* dst = cv[pos[s]].pts.data() + 2 * (size_t)fill[s]++;
            dst[0] = seg[2 * i] & 0xFFFF;
            dst[1] = seg[2 * i] >> 16;
        }
    });
    return CK_OK;
}
