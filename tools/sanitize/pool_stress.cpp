// Stress of ck_parallel_for (ck_pool.h) on the worker pool (ck_pool.cpp) for ThreadSanitizer and ASan / UBSan: concurrent
// loops from six caller threads, 1 .. 97 items, results checked, and an item that throws (the exception must come back on
// the caller's thread).  tools/sanitize/run.sh builds and runs it both ways.
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <vector>
#include <stdio.h>
#include <unistd.h>
#include "../../camkifu_amd/csrc/ck_pool.h"
int main()
{
    std::atomic<long> grand{0};
    std::vector<std::thread> callers;
    for (int c = 0; c < 6; c++)
        callers.emplace_back([&, c] {
            for (int rep = 0; rep < 400; rep++) {
                const int n = 1 + (rep * 7 + c) % 97;
                std::vector<int> out((size_t)n, 0);
                ck_parallel_for(n, 16, [&](int i) { out[(size_t)i] = i * i + c; });
                long s = 0;
                for (int i = 0; i < n; i++) s += out[(size_t)i] - c;
                long want = 0;
                for (int i = 0; i < n; i++) want += (long)i * i;
                if (s != want) { printf("MISMATCH caller %d rep %d\n", c, rep); _exit(1); }
                grand += s;
                if (rep % 50 == 49) {
                    bool threw = false;
                    try { ck_parallel_for(n, 16, [&](int i) { if (i == n / 2) throw std::runtime_error("x"); }); } catch (const std::exception&) { threw = true; }
                    if (!threw) { printf("no exception\n"); _exit(1); }
                }
            }
        });
    for (auto& t : callers) t.join();
    printf("ok %ld (pool %d)\n", grand.load(), ck_pool_size());
    return 0;
}
