// trg_workers.h -- the persistent host threads of a device group (trg_group.cpp): one per device, started once, handed a job per
// call through a condition variable.  A C2 band on 8 GPUs is 0.26-0.38 ms of device time: spawning and joining eight std::threads
// per frame, as round 2 did, costs the same order.  Header-only and free of HIP so that the hand-off is unit-tested on the CPU
// (tests/helpers/workers_test.cpp).
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace trg {

class DeviceWorkers {
public:
    // n workers; on_start(rank) runs once on each worker's own thread before its first job (hipSetDevice for rank's device).
    // A group of one has no thread at all: its jobs run on the caller's.
    explicit DeviceWorkers(int n, std::function<void(int)> on_start = nullptr) : n_(n), rc_((size_t)(n > 0 ? n : 0), 0) {
        if (n_ <= 1) { if (n_ == 1 && on_start) on_start(0); return; }
        threads_.reserve((size_t)n_);
        for (int r = 0; r < n_; ++r)
            threads_.emplace_back([this, r, on_start] {
                if (on_start) on_start(r);
                unsigned long long seen = 0;
                for (;;) {
                    std::function<int(int)> job;
                    {
                        std::unique_lock<std::mutex> lk(m_);
                        start_.wait(lk, [&] { return quit_ || gen_ != seen; });
                        if (quit_) return;
                        seen = gen_;
                        job = job_;
                    }
                    const int rc = job(r);
                    {
                        std::lock_guard<std::mutex> lk(m_);
                        rc_[(size_t)r] = rc;
                        if (--pending_ == 0) done_.notify_all();
                    }
                }
            });
    }
    ~DeviceWorkers() {
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_ = true;
        }
        start_.notify_all();
        for (std::thread &t : threads_) t.join();
    }
    DeviceWorkers(const DeviceWorkers &) = delete;
    DeviceWorkers &operator=(const DeviceWorkers &) = delete;

    int size() const { return n_; }

    // f(rank) on every worker at the same time; returns when all are done.  The result of rank r is rc(r) until the next run.
    // One caller at a time (the group's API is single-threaded per group, like a context's).
    void run(const std::function<int(int)> &f) {
        if (n_ <= 0) return;
        if (n_ == 1) { rc_[0] = f(0); return; }
        std::unique_lock<std::mutex> lk(m_);
        job_ = f;
        pending_ = n_;
        ++gen_;
        start_.notify_all();
        done_.wait(lk, [&] { return pending_ == 0; });
        job_ = nullptr;
    }
    int rc(int rank) const { return rc_[(size_t)rank]; }
    // the first rank whose last job did not return 0, or -1
    int first_failure() const {
        for (int r = 0; r < n_; ++r)
            if (rc_[(size_t)r] != 0) return r;
        return -1;
    }

private:
    int n_;
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable start_, done_;
    std::function<int(int)> job_;
    unsigned long long gen_ = 0;
    int pending_ = 0;
    bool quit_ = false;
    std::vector<int> rc_;
};

}  // namespace trg
