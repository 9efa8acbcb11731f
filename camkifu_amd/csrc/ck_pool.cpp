// ck_pool.cpp -- the library's host worker threads (ck_parallel_for in ck_common.h): a process-wide pool, created on
// first use and never torn down (its threads sleep on a condition variable; at process exit they simply end with it,
// which avoids every static-destruction order question with contexts that are still being closed).
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <stdlib.h>

namespace {

struct Pool {
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> queue;
    int size = 0;

    Pool()
    {
        // CK_HOST_THREADS overrides; otherwise up to 16 (a one-GPU share of a host), never more than the machine has
        unsigned hw = std::thread::hardware_concurrency();
        int want = (int)(hw ? hw : 4);
        if (want > 16) want = 16;
        if (const char* e = getenv("CK_HOST_THREADS")) {
            const int v = atoi(e);
            if (v >= 1 && v <= 256) want = v;
        }
        for (int t = 0; t < want - 1; t++) {            // the caller of a loop is the remaining worker
            try {
                std::thread([this] { run(); }).detach();
                size++;
            } catch (...) {
                break;                                   // thread limit: a smaller pool
            }
        }
    }

    void run()
    {
        for (;;) {
            std::function<void()> task;
            {
                std::unique_lock<std::mutex> lock(m);
                cv.wait(lock, [this] { return !queue.empty(); });
                task = std::move(queue.front());
                queue.pop_front();
            }
            task();                                      // (ck_parallel_for's claim: catches what the items throw)
        }
    }
};

Pool& pool()
{
    static Pool* p = new Pool();                         // leaked on purpose, see the top of the file
    return *p;
}

}  // namespace

int ck_pool_size() { return pool().size; }

void ck_pool_submit(std::function<void()> task)
{
    Pool& p = pool();
    {
        std::lock_guard<std::mutex> lock(p.m);
        p.queue.push_back(std::move(task));
    }
    p.cv.notify_one();
}
