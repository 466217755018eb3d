// pool_selftest.cpp — aqc_pool.hpp's lane policy (test infrastructure; built and run by tests/test_gz_codec.py):
//   * front jobs overtake queued background jobs: with every worker busy and both lanes full, all front jobs are started
//     before any further background job is;
//   * help_front() runs a queued front job on the calling thread and reports an empty lane;
//   * parallel_for still completes with the pool saturated by background jobs;
//   * help_front() never runs the helper closure of another thread's parallel_for.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <mutex>
#include <thread>
#include <vector>

#include "../../afterqc_amd/csrc/aqc_pool.hpp"

int main() {
    using namespace std::chrono;
    int failures = 0;
    {
        aqc_host::Pool pool(4);
        std::atomic<int> gate{0}, order{0}, bg_started{0}, front_done{0};
        std::vector<int> front_rank(32, -1), bg_rank(16, -1);
        // four background jobs occupy the workers until the gate opens
        for (int i = 0; i < 4; ++i)
            pool.submit([&] { bg_started++; while (!gate.load()) std::this_thread::sleep_for(microseconds(50)); }, true);
        while (bg_started.load() < 4) std::this_thread::sleep_for(microseconds(50));
        // now both lanes fill up while nobody can take anything
        for (int i = 0; i < 16; ++i) pool.submit([&, i] { bg_rank[i] = order++; std::this_thread::sleep_for(microseconds(200)); }, true);
        for (int i = 0; i < 32; ++i) pool.submit([&, i] { front_rank[i] = order++; front_done++; }, false);
        gate = 1;
        const auto t0 = steady_clock::now();
        while (order.load() < 48 && steady_clock::now() - t0 < seconds(20)) std::this_thread::sleep_for(microseconds(100));
        int worst_front = -1, first_bg = 1 << 30;
        for (int r : front_rank) worst_front = r > worst_front ? r : worst_front;
        for (int r : bg_rank) first_bg = r < first_bg ? r : first_bg;
        const bool ok = order.load() == 48 && worst_front >= 0 && worst_front < first_bg;
        printf("front lane first: last front job ranked %d, first queued background job %d  %s\n", worst_front, first_bg, ok ? "ok" : "FAIL");
        if (!ok) ++failures;
    }
    {
        aqc_host::Pool pool(2);
        std::atomic<int> gate{0}, started{0}, ran{0};
        for (int i = 0; i < 2; ++i) pool.submit([&] { started++; while (!gate.load()) std::this_thread::sleep_for(microseconds(50)); }, true);
        while (started.load() < 2) std::this_thread::sleep_for(microseconds(50));
        const std::thread::id me = std::this_thread::get_id();
        std::atomic<int> on_caller{0};
        for (int i = 0; i < 5; ++i) pool.submit([&] { ran++; if (std::this_thread::get_id() == me) on_caller++; }, false);
        int helped = 0;
        while (pool.help_front()) ++helped;
        const bool ok = helped == 5 && ran.load() == 5 && on_caller.load() == 5 && !pool.help_front();
        printf("help_front: %d jobs run by the caller while the workers were busy  %s\n", helped, ok ? "ok" : "FAIL");
        if (!ok) ++failures;
        // parallel_for with the workers still stuck: the caller does the work
        std::atomic<long> sum{0};
        pool.parallel_for(100, [&](size_t i) { sum += (long)i; });
        const bool ok2 = sum.load() == 4950;
        printf("parallel_for under saturation: sum %ld  %s\n", sum.load(), ok2 ? "ok" : "FAIL");
        if (!ok2) ++failures;
        gate = 1;
    }
    {
        // help_front never takes the helper closure of somebody else's parallel_for (it would sit in that batch until it is done):
        // with the workers stuck, a thread in a long parallel_for has its helpers queued; the main thread's help_front must find only
        // the job that was SUBMITTED — and report an empty lane afterwards although the helpers are still there.
        aqc_host::Pool pool(2);
        std::atomic<int> gate{0}, started{0}, batch_gate{0}, in_batch{0}, submitted_ran{0};
        for (int i = 0; i < 2; ++i) pool.submit([&] { started++; while (!gate.load()) std::this_thread::sleep_for(microseconds(50)); }, true);
        while (started.load() < 2) std::this_thread::sleep_for(microseconds(50));
        const std::thread::id me = std::this_thread::get_id();
        std::atomic<int> batch_on_me{0};
        std::thread other([&] {
            pool.parallel_for(8, [&](size_t) {
                in_batch++;
                if (std::this_thread::get_id() == me) batch_on_me++;
                while (!batch_gate.load()) std::this_thread::sleep_for(microseconds(50));
            });
        });
        while (in_batch.load() < 1) std::this_thread::sleep_for(microseconds(50));       // (its helpers are queued, it works on index 0)
        pool.submit([&] { submitted_ran++; }, false);
        int helped = 0;
        while (pool.help_front()) ++helped;
        const bool ok = helped == 1 && submitted_ran.load() == 1 && batch_on_me.load() == 0;
        printf("help_front leaves parallel_for helpers alone: helped %d, batch indices run by the helper %d  %s\n", helped, batch_on_me.load(), ok ? "ok" : "FAIL");
        if (!ok) ++failures;
        batch_gate = 1;
        gate = 1;
        other.join();
    }
    if (failures) { printf("%d pool checks FAILED\n", failures); return 1; }
    printf("all pool checks passed\n");
    return 0;
}
