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
