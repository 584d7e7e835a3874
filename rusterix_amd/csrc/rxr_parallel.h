// A small persistent worker pool for the HOST side of the path (header only; each shared library that includes it gets its own).
//
// The reference projects a scene's batches with rayon (`par_iter_mut` over chunks and batch lists, src/scene.rs:162-211) and walks
// the screen tiles with `par_iter` (src/rasterizer.rs:274).  The tiles went to the GPU; what stays on the host per frame -- one
// `clip_and_project` per batch (host mirror, rusterix_host.cpp) and the flattening of the projected batches into the pinned staging
// blob (rxr_upload_frame, rxr_api.hip) -- is independent per batch and runs through this pool.
//
//   rxr_parallel::run(n_items, weight, [&](size_t i) { ... });
//
// Items are handed out one at a time through an atomic cursor (batches differ in size).  `weight` is the caller's estimate of the
// total work in "elements" (vertices + triangles, or bytes / 64): below RXR_PARALLEL_MIN_WEIGHT the loop runs inline on the caller's
// thread, so that small frames (the bench frame has 54 triangles) never pay a wake-up.
// Threads: RXR_HOST_THREADS if set (1 = never create a thread), else min(CPUs this process may run on, 64) -- rayon's default is
// every logical CPU -- and never more than one per RXR_PARALLEL_MIN_WEIGHT / 4 elements of the job, so that a mid-sized frame wakes
// a few workers and the 1 M-triangle grid all of them (measured on the 256-thread host of the GPU box, profiles/r03/c5_e2e_*: that
// scene's Scene::project takes 10.8 / 5.5 / 3.9 / 2.5 ms on 8 / 16 / 32 / 64 threads).  Workers are created on first use, sleep on a
// condition variable between jobs and are joined when the library is unloaded.  A forked child starts without workers
// (pthread_atfork) and creates its own on first use.
//
//   rxr_parallel::run_with(n_items, weight, fn, main_fn)   the same, but the CALLING thread runs main_fn() instead of taking items:
//   it returns when main_fn has returned and every item is done (rxr_upload_frame: the workers copy batches into pinned memory while
//   the caller issues the host->device copies of the ranges that are complete).
#pragma once
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <new>
#include <thread>
#include <type_traits>
#include <vector>

#ifndef RXR_PARALLEL_MIN_WEIGHT
#define RXR_PARALLEL_MIN_WEIGHT 65536
#endif

namespace rxr_parallel {

class Pool {
  public:
    static Pool &get() {
        static Pool p;
        return p;
    }
    unsigned threads() {
        std::lock_guard<std::mutex> lk(mu_);
        return wanted_locked();
    }
    // fn(i) for i in [0, n), each exactly once; returns when all are done.  Not re-entrant: one job at a time per pool (callers hold
    // the library's own lock: g_mu in the host mirror, the context in rxr_upload_frame).
    template <class F> void run(size_t n, size_t weight, F &&fn) {
        run_impl(n, weight, static_cast<F &&>(fn), (void (*)(void *)) nullptr, nullptr);
    }
    // the calling thread runs main_fn() while the workers take the items; false (nothing done) when the job is too small for the
    // pool or the pool has one thread: the caller then does both itself, one after the other
    template <class F, class M> bool run_with(size_t n, size_t weight, F &&fn, M &&main_fn) {
        using Mn = typename std::remove_reference<M>::type;
        struct MainThunk {
            static void call(void *self) { (*(Mn *)self)(); }
        };
        return run_impl(n, weight, static_cast<F &&>(fn), &MainThunk::call, (void *)&main_fn);
    }

  private:
    template <class F> bool run_impl(size_t n, size_t weight, F &&fn, void (*main_call)(void *), void *main_arg) {
        if (n == 0 && !main_call) return true;
        unsigned want;
        {
            std::lock_guard<std::mutex> lk(mu_);
            want = wanted_locked();
        }
        // a thread per RXR_PARALLEL_MIN_WEIGHT / 4 elements at most: waking 63 workers for a job of 70 000 elements costs more than it saves
        const size_t by_weight = weight / ((size_t)RXR_PARALLEL_MIN_WEIGHT / 4u) + 1u;
        if (by_weight < want) want = (unsigned)by_weight;
        if (n <= 1 || want <= 1 || weight < (size_t)RXR_PARALLEL_MIN_WEIGHT) {
            if (main_call) return false;
            for (size_t i = 0; i < n; ++i) fn(i);
            return true;
        }
        using Fn = typename std::remove_reference<F>::type;  // (callers pass temporaries and named lambdas alike)
        struct Thunk {
            Fn *f;
            static void call(void *self, size_t i) { (*((Thunk *)self)->f)(i); }
        } thunk{&fn};
        std::unique_lock<std::mutex> job_lock(job_mu_);  // one job at a time
        {
            std::unique_lock<std::mutex> lk(mu_);
            start_workers_locked(want - 1);
            call_ = &Thunk::call;
            arg_ = &thunk;
            n_ = n;
            next_.store(0, std::memory_order_relaxed);
            active_ = want - 1;   // the first `active_` workers take part in this job, the others go back to sleep at once
            busy_ = (unsigned)workers_.size();
            ++generation_;
        }
        cv_work_.notify_all();
        if (main_call) main_call(main_arg);
        else drain();
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return busy_ == 0; });
        call_ = nullptr;
        return true;
    }

  public:

  private:
    Pool() { pthread_atfork(nullptr, nullptr, &Pool::in_child); }
    ~Pool() { stop(); }
    Pool(const Pool &) = delete;

    static unsigned allowed_cpus() {
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0) {
            int c = CPU_COUNT(&set);
            if (c > 0) return (unsigned)c;
        }
        unsigned h = std::thread::hardware_concurrency();
        return h ? h : 1u;
    }
    unsigned wanted_locked() {
        if (!wanted_) {
            const char *e = getenv("RXR_HOST_THREADS");
            long v = e ? atol(e) : 0;
            wanted_ = v > 0 ? (unsigned)(v > 256 ? 256 : v) : (allowed_cpus() < 64u ? allowed_cpus() : 64u);
        }
        return wanted_;
    }
    void start_workers_locked(unsigned n) {
        while (workers_.size() < n) {
            const unsigned index = (unsigned)workers_.size();
            workers_.emplace_back([this, index, seen = generation_]() mutable { worker(index, seen); });
        }
    }
    void drain() {
        for (;;) {
            size_t i = next_.fetch_add(1, std::memory_order_relaxed);
            if (i >= n_) break;
            call_(arg_, i);
        }
    }
    void worker(unsigned index, unsigned long long seen) {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_work_.wait(lk, [&] { return stop_ || generation_ != seen; });
            if (stop_) return;
            seen = generation_;
            if (index < active_) {
                lk.unlock();
                drain();
                lk.lock();
            }
            if (--busy_ == 0) cv_done_.notify_all();
        }
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_work_.notify_all();
        for (std::thread &t : workers_)
            if (t.joinable()) t.join();
        workers_.clear();
    }
    // the child of a fork has none of the parent's threads: forget them (their std::thread objects are abandoned, not joined) and
    // re-initialise the synchronisation objects in place -- a mutex the parent held at fork time would stay locked for ever
    static void in_child() {
        Pool &p = get();
        new (&p.workers_) std::vector<std::thread>();
        new (&p.mu_) std::mutex();
        new (&p.job_mu_) std::mutex();
        new (&p.cv_work_) std::condition_variable();
        new (&p.cv_done_) std::condition_variable();
        p.busy_ = 0;
        p.call_ = nullptr;
    }

    std::mutex mu_, job_mu_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<std::thread> workers_;
    void (*call_)(void *, size_t) = nullptr;
    void *arg_ = nullptr;
    size_t n_ = 0;
    std::atomic<size_t> next_{0};
    unsigned busy_ = 0, wanted_ = 0, active_ = 0;
    unsigned long long generation_ = 0;
    bool stop_ = false;
};

template <class F> inline void run(size_t n, size_t weight, F &&fn) { Pool::get().run(n, weight, static_cast<F &&>(fn)); }
template <class F, class M> inline bool run_with(size_t n, size_t weight, F &&fn, M &&main_fn) {
    return Pool::get().run_with(n, weight, static_cast<F &&>(fn), static_cast<M &&>(main_fn));
}
inline unsigned threads() { return Pool::get().threads(); }

}  // namespace rxr_parallel
