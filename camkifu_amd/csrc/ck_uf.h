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

__device__ __forceinline__ void uf_union(int* L, int a, int b)
{
    for (;;) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }          // a > b: hook a under b
        const int old = atomicMin(L + a, b);
        if (old == a) return;                             // a was still a root: done
        a = old;                                          // somebody re-parented a meanwhile
    }
}
