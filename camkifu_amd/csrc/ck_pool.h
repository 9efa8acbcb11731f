// ck_pool.h -- host loops over independent items on the library's worker threads (no HIP in here: the sanitizer harness
// tools/sanitize/pool_stress.cpp includes it on its own).
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>

// Host loop over independent items on the library's worker threads (ck_pool.cpp: a process-wide pool, started on first
// use).  The caller works too and claims items like a helper, so a pool that is busy with another context's loop only
// means fewer helpers, never a wait for them to START: a helper that arrives late finds nothing to claim and touches
// nothing but the (shared, heap) job record.  An exception inside an item is carried back and rethrown on the
// caller's thread, where the entry point's bracket turns it into an error code.
// (Threads used to be created per loop: ~25 us each, 0.4 ms of a 16-frame board call's 1.3 -- tools/board_call_latency.py.)
int ck_pool_size();
void ck_pool_submit(std::function<void()> task);
struct CkLoopJob {
    int n = 0;
    std::atomic<int> next{0}, active{0};
    std::atomic<bool> failed{false};
    std::mutex m;
    std::condition_variable cv;
};
template <typename F>
void ck_parallel_for(int n, int max_threads, F fn)
{
    int nt = ck_pool_size() + 1;
    if (nt > max_threads) nt = max_threads;
    if (nt > n) nt = n;
    if (nt <= 1) { for (int i = 0; i < n; i++) fn(i); return; }
    auto job = std::make_shared<CkLoopJob>();
    job->n = n;
    F* body = &fn;                                   // only dereferenced for a claimed item, i.e. while this frame is alive
    auto claim = [job, body]() {
        CkLoopJob& j = *job;
        j.active.fetch_add(1);
        for (int i; !j.failed.load(std::memory_order_relaxed) && (i = j.next.fetch_add(1)) < j.n;) {
            try {
                (*body)(i);
            } catch (...) {
                j.failed.store(true);
            }
        }
        if (j.active.fetch_sub(1) == 1) {
            std::lock_guard<std::mutex> lock(j.m);
            j.cv.notify_all();
        }
    };
    try {
        for (int t = 1; t < nt; t++) ck_pool_submit(claim);
    } catch (...) {                                  // (no memory for the queue entry: fewer helpers)
    }
    claim();
    {
        // every item has been claimed by now; wait for the helpers that are still inside one
        std::unique_lock<std::mutex> lock(job->m);
        job->cv.wait(lock, [&] { return job->active.load() == 0; });
    }
    if (job->failed.load()) throw std::runtime_error("a host worker failed (out of memory?)");
}

