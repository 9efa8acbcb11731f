// ck_uf.h -- lock-free union-find on a label image (parents are linear pixel indices inside
// one frame; a root satisfies L[r] == r; hooking always points the larger root at the
// smaller one with an agent-scope atomicMin, so the structure is a forest at every instant
// and the result does not depend on dispatch order or XCD placement).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ int uf_load(const int* L, int a)
{
    // relaxed agent-scope load: served from L2/memory, never from a stale per-CU L1 line
    return __hip_atomic_load(L + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int uf_find(const int* L, int a)
{
    int p = uf_load(L, a);
    while (p != a) { a = p; p = uf_load(L, a); }
    return a;
}

// find with path compression: the start node is re-pointed at the root.  Only non-root nodes
// are ever written with a plain store, and always with an ancestor of smaller index, so a
// concurrent atomicMin on the same node still sees a valid (older or newer) ancestor.
__device__ __forceinline__ int uf_find_compress(int* L, int a)
{
    const int start = a;
    int p = uf_load(L, a);
    const int first = p;
    while (p != a) { a = p; p = uf_load(L, a); }
    if (first != a && start != a)
        __hip_atomic_store(L + start, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return a;
}

__device__ __forceinline__ void uf_union(int* L, int a, int b)
{
    for (;;) {
        a = uf_find_compress(L, a);
        b = uf_find_compress(L, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }          // a > b: hook a under b
        const int old = atomicMin(L + a, b);
        if (old == a) return;                             // a was still a root: done
        a = old;                                          // somebody re-parented a meanwhile
    }
}

// Wave-aggregated append: every lane with `flag` gets a unique slot of a global list whose
// element count lives at *counter.  One atomic per wave.  Returns the slot (or -1).
__device__ __forceinline__ int wave_append(int* counter, bool flag)
{
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(flag);
    if (mask == 0) return -1;
    const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int leader = __builtin_ctzll(mask);
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __builtin_popcountll(mask));
    base = __builtin_amdgcn_readlane(base, leader);
    if (!flag) return -1;
    return base + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
}

// List kernels: workgroup -> (frame, block of the frame's grid-stride loop).  Launched either as a 2-D grid (block, frame) or,
// with CK_LIST_XCD, as a 1-D grid of nblk * frames workgroups in which all blocks of a frame get ids congruent modulo 8:
// workgroup ids go round-robin over the 8 XCDs, so a frame's union-find traffic stays in ONE L2 (frames % 8 == 0 only).
#ifndef CK_LIST_XCD
#define CK_LIST_XCD 1
#endif
__device__ __forceinline__ void list_frame_block(int nblk, int& f, int& bx)
{
    if (gridDim.y > 1) { f = blockIdx.y; bx = blockIdx.x; return; }
    const int L = blockIdx.x, n = gridDim.x / nblk;
    if ((n & 7) == 0) { const int slot = L >> 3; f = (slot / nblk) * 8 + (L & 7); bx = slot % nblk; }
    else { f = L / nblk; bx = L % nblk; }
}
static inline dim3 list_grid(int nblk, int n) { return CK_LIST_XCD ? dim3((unsigned)(nblk * n)) : dim3(nblk, n); }
