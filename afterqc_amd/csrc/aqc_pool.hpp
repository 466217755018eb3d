// aqc_pool.hpp — the I/O-side thread pool of the whole-input pipe (host code only): pread pieces, newline counts, inflate /
// deflate of independent blocks, the speculative sections of the parallel gunzip.
//   parallel_for   blocking, the caller works too; its jobs go to the FRONT lane (short, somebody waits for them)
//   submit         fire and forget; `background` jobs (long speculative work: the gunzip sections) have a lane of their own,
//                  served only while the front lane is empty: a front job is short and somebody WAITS for it (the gunzip
//                  consumer for the translation of the sections it just committed); a section is work for later.  (The lanes
//                  used to be served in turn: every other visit of a worker then cost the waiting consumer a 7 ms section —
//                  up to a section per worker and chunk, which is what held `.gz -> .gz` at 33 Mreads/s whoever decoded.)
//   help_front     the caller runs one queued front job itself instead of sleeping until a worker gets to it — a job somebody
//                  SUBMITTED, never the helper closure of another thread's parallel_for (that one loops until its whole batch
//                  is done: the gunzip consumer would sit in a writer's deflate batch while its own data is ready)
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <pthread.h>
#include <time.h>

#include <thread>
#include <vector>

namespace aqc_host {

class Pool {
public:
    explicit Pool(int n) {
        for (int i = 0; i < n; ++i) th_.emplace_back([this] { loop(); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    int size() const { return (int)th_.size(); }
    // CPU seconds the pool's threads have used so far (diagnostics: AQC_PIPE_DEBUG)
    double cpu_seconds() {
        double total = 0;
        for (auto& t : th_) {
            clockid_t cid;
            timespec ts;
            if (pthread_getcpuclockid(t.native_handle(), &cid) == 0 && clock_gettime(cid, &ts) == 0) total += (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
        }
        return total;
    }

    void submit(std::function<void()> job, bool background = false) {
        if (th_.empty()) { job(); return; }
        {
            std::lock_guard<std::mutex> g(mu_);
            if (background) bg_.push_back(std::move(job));
            else q_.push_back(Job{std::move(job), false});
        }
        cv_.notify_one();
    }

    // one front job, if there is one, on the calling thread
    bool help_front() {
        std::function<void()> job;
        {
            std::lock_guard<std::mutex> g(mu_);
            auto it = q_.begin();
            while (it != q_.end() && it->batch_helper) ++it;
            if (it == q_.end()) return false;
            job = std::move(it->fn);
            q_.erase(it);
        }
        job();
        return true;
    }

    // run fn(i) for i in [0, n) on the pool and wait for all of them
    void parallel_for(size_t n, const std::function<void(size_t)>& fn) {
        if (n == 0) return;
        if (n == 1 || th_.empty()) {
            for (size_t i = 0; i < n; ++i) fn(i);
            return;
        }
        struct Batch {
            std::atomic<size_t> next{0}, done{0};
            size_t n;
            const std::function<void(size_t)>* fn;
            std::mutex mu;
            std::condition_variable cv;
        };
        auto b = std::make_shared<Batch>();
        b->n = n;
        b->fn = &fn;
        const size_t helpers = std::min(n - 1, th_.size());
        {
            std::lock_guard<std::mutex> g(mu_);
            for (size_t k = 0; k < helpers; ++k)
                q_.push_back(Job{[b] {
                    for (;;) {
                        const size_t i = b->next.fetch_add(1);
                        if (i >= b->n) break;
                        (*b->fn)(i);
                        if (b->done.fetch_add(1) + 1 == b->n) {
                            std::lock_guard<std::mutex> g2(b->mu);
                            b->cv.notify_all();
                        }
                    }
                }, true});
        }
        cv_.notify_all();
        // the caller works too
        for (;;) {
            const size_t i = b->next.fetch_add(1);
            if (i >= n) break;
            fn(i);
            b->done.fetch_add(1);
        }
        std::unique_lock<std::mutex> lk(b->mu);
        b->cv.wait(lk, [&] { return b->done.load() >= n; });
    }

private:
    void loop() {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || !q_.empty() || !bg_.empty(); });
                const bool take_bg = !bg_.empty() && q_.empty();
                if (take_bg) { job = std::move(bg_.front()); bg_.pop_front(); }
                else if (!q_.empty()) { job = std::move(q_.front().fn); q_.pop_front(); }
                else if (stop_) return;
                else continue;
            }
            job();
        }
    }
    std::vector<std::thread> th_;
    struct Job {
        std::function<void()> fn;
        bool batch_helper;      // a parallel_for helper: runs until its batch is done (workers only, see help_front)
    };
    std::deque<Job> q_;
    std::deque<std::function<void()>> bg_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false;
};

}  // namespace aqc_host
