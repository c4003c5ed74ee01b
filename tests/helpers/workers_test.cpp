// CPU unit test of the device group's persistent host threads (toyraygun_amd/csrc/trg_workers.h): every job reaches every worker
// exactly once, on the SAME thread each time; run() returns only when all are done; failures come back per rank; a group of one
// runs on the caller's thread; thousands of hand-offs; destruction with idle workers.  Built with g++ -pthread (and -fsanitize=thread).
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

#include "../../toyraygun_amd/csrc/trg_workers.h"

#define CHECK(c) do { if (!(c)) { printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main() {
    for (int n : { 1, 2, 3, 8 }) {
        std::vector<std::thread::id> started((size_t)n);
        std::atomic<int> starts{ 0 };
        trg::DeviceWorkers w(n, [&](int r) { started[(size_t)r] = std::this_thread::get_id(); ++starts; });
        CHECK(w.size() == n);
        std::vector<long long> sum((size_t)n, 0);   // slot r is only ever touched by worker r: no lock needed if the hand-off is right
        std::vector<std::thread::id> seen((size_t)n);
        const int rounds = 3000;
        for (int k = 0; k < rounds; ++k) {
            std::atomic<int> in_flight{ 0 };
            w.run([&](int r) {
                ++in_flight;
                if (k == 0) seen[(size_t)r] = std::this_thread::get_id();
                else if (seen[(size_t)r] != std::this_thread::get_id()) return -99;   // a job of rank r on another thread
                sum[(size_t)r] += k + r;
                return (k % 7 == 3 && r == n - 1) ? -5 : 0;
            });
            CHECK(in_flight.load() == n);                       // run() returned after every worker had run the job
            for (int r = 0; r < n; ++r) CHECK(w.rc(r) == ((k % 7 == 3 && r == n - 1) ? -5 : 0));
            CHECK(w.first_failure() == ((k % 7 == 3) ? n - 1 : -1));
        }
        CHECK(starts.load() == n);                              // on_start ran once per worker
        for (int r = 0; r < n; ++r) {
            CHECK(sum[(size_t)r] == (long long)rounds * (rounds - 1) / 2 + (long long)rounds * r);
            CHECK(seen[(size_t)r] == started[(size_t)r]);        // jobs run on the thread that ran on_start (the one bound to the device)
            if (n == 1) CHECK(seen[0] == std::this_thread::get_id());   // a group of one has no thread of its own
            else CHECK(seen[(size_t)r] != std::this_thread::get_id());
        }
    }
    {   // destroyed without ever running a job
        trg::DeviceWorkers idle(4);
    }
    printf("workers ok\n");
    return 0;
}
