// Exercises rusterix_amd/csrc/rxr_parallel.h (the host worker pool) under ThreadSanitizer: jobs of many sizes, from two caller
// threads at once (the multi-device context uploads on one host thread per member), every item exactly once, and a forked child
// that starts its own workers.  Built and run by tests/test_parallel_pool.py; prints "ok" on success.
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

#define RXR_PARALLEL_MIN_WEIGHT 16
#include "../rusterix_amd/csrc/rxr_parallel.h"

static bool sweep(unsigned salt) {
    for (size_t n : {size_t(0), size_t(1), size_t(2), size_t(7), size_t(64), size_t(1000), size_t(4097)}) {
        std::vector<std::atomic<unsigned>> hits(n);
        for (auto &h : hits) h.store(0);
        std::vector<unsigned long long> out(n, 0);
        rxr_parallel::run(n, n * 8, [&](size_t i) {
            hits[i].fetch_add(1);
            unsigned long long acc = salt;
            for (unsigned k = 0; k < 50 + (i % 17) * 40; ++k) acc = acc * 6364136223846793005ull + i + k;  // uneven items
            out[i] = acc;
        });
        for (size_t i = 0; i < n; ++i) {
            if (hits[i].load() != 1) return false;
            unsigned long long acc = salt;
            for (unsigned k = 0; k < 50 + (i % 17) * 40; ++k) acc = acc * 6364136223846793005ull + i + k;
            if (out[i] != acc) return false;
        }
    }
    return true;
}

// run_with: the workers take the items while the CALLER runs its own function, which consumes the items' results in order as they
// complete (rxr_upload_frame ships a group of batches as soon as its copies have landed)
static bool sweep_with_main(unsigned salt) {
    for (size_t n : {size_t(2), size_t(9), size_t(300), size_t(2048)}) {
        std::vector<std::atomic<unsigned>> done(n);
        for (auto &d : done) d.store(0);
        std::vector<unsigned long long> out(n, 0), seen(n, 0);
        bool main_ran = false;
        const bool pooled = rxr_parallel::run_with(n, n * 64, [&](size_t i) {
            unsigned long long acc = salt;
            for (unsigned k = 0; k < 30 + (i % 13) * 50; ++k) acc = acc * 6364136223846793005ull + i + k;
            out[i] = acc;
            done[i].store(1, std::memory_order_release);
        }, [&] {
            main_ran = true;
            for (size_t i = 0; i < n; ++i) {
                while (done[i].load(std::memory_order_acquire) == 0) std::this_thread::yield();
                seen[i] = out[i];
            }
        });
        if (!pooled) return false;  // (the caller made sure the pool has more than one thread and the job is heavy enough)
        if (!main_ran) return false;
        for (size_t i = 0; i < n; ++i) {
            unsigned long long acc = salt;
            for (unsigned k = 0; k < 30 + (i % 13) * 50; ++k) acc = acc * 6364136223846793005ull + i + k;
            if (seen[i] != acc) return false;
        }
    }
    // too small for the pool: nothing runs, the caller is told to do both itself
    bool touched = false;
    if (rxr_parallel::run_with(4, 1, [&](size_t) { touched = true; }, [&] { touched = true; }) || touched) return false;
    return true;
}

int main(int argc, char **argv) {
    const bool with_fork = !(argc > 1 && std::string(argv[1]) == "nofork");  // (ThreadSanitizer refuses new threads after a multi-threaded fork)
    if (rxr_parallel::threads() < 2) {
        std::puts("ok (single thread)");
        return 0;
    }
    std::atomic<bool> good{true};
    for (int round = 0; round < 20; ++round) {
        std::thread other([&] { if (!sweep(1000u + round)) good = false; });
        if (!sweep((unsigned)round)) good = false;
        if (!sweep_with_main((unsigned)round + 500u)) good = false;
        other.join();
    }
    if (!good) {
        std::puts("FAILED: an item ran zero or several times, or a result is wrong");
        return 1;
    }
    if (!with_fork) {
        std::puts("ok");
        return 0;
    }
    // a forked child has no workers: it must build its own and still get every item
    pid_t pid = fork();
    if (pid == 0) _exit(sweep(77u) ? 0 : 3);
    int status = 0;
    waitpid(pid, &status, 0);
    if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) {
        std::printf("FAILED: forked child status %d\n", status);
        return 1;
    }
    if (!sweep(99u)) return 1;  // and the parent's pool is still alive
    std::puts("ok");
    return 0;
}
