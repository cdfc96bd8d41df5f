// CPU-side stress harness of the lock-step combiner (qilaplace.jl_amd/csrc/qil_lockstep_core.h -- the code libqilhip.so runs,
// bound here to a stub launch function instead of HIP).  Built by tests/test_lockstep_stress.py with g++ -fsanitize=thread.
//
//   G launcher threads, each serving a group of up to 16 chains (rings), as qil_run_batch_on does;
//   one producer thread per chain: random bursts of launch requests of a few kernel classes under random (monotone) progress
//   keys, read-back requests it then PARKS on (the stub "device" completes a read-back when its launch is issued, optionally
//   later from a device thread), ring drains, and -- for some chains -- a launch that FAILS in the middle.
//
// Checked: every chain's requests are issued exactly once and in the chain's own order (also around failures), a combined
// launch never mixes kernel classes and never exceeds 16 operands, a chain whose launch failed sees the status at its next
// commit / park, every thread terminates.  ThreadSanitizer reports any data race in the protocol itself.
//
//   usage: lockstep_stress <chains> <groups> <steps per chain> <seed> [device-delay-us]
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "qil_lockstep_core.h"

struct StubStream {
    int group = 0;
};
struct StubReq {
    const void* kern = nullptr;
    struct {
        unsigned x = 1, y = 1, z = 1;
    } block;
    size_t lds = 0;
    int (*launch_group)(StubReq* const* reqs, int n, StubStream s) = nullptr;
    uint64_t progress = 0;
    unsigned seq = 0;
    // payload of the stub
    int chain = 0;
    long serial = 0;
    bool fail = false;
    unsigned long long* ticket_word = nullptr;     // a read-back: completing it stores `ticket` here
    unsigned long long ticket = 0;
};

static std::vector<std::atomic<long>> g_issued;    // per chain: serial of the last request issued
static std::atomic<long> g_errors{0}, g_launches{0}, g_requests{0};
static int g_delay_us = 0;
static char g_classes[5];

static int stub_launch(StubReq* const* reqs, int n, StubStream) {
    if (n < 1 || n > QIL_LS_MAXB) ++g_errors;
    bool fail = false;
    for (int i = 0; i < n; ++i) {
        if (reqs[i]->kern != reqs[0]->kern || reqs[i]->block.x != reqs[0]->block.x) ++g_errors;     // one class per launch
        for (int j = 0; j < i; ++j)
            if (reqs[j]->chain == reqs[i]->chain) ++g_errors;                                        // one head per chain
        const long prev = g_issued[(size_t)reqs[i]->chain].exchange(reqs[i]->serial, std::memory_order_relaxed);
        if (prev + 1 != reqs[i]->serial) ++g_errors;                                                // in order, exactly once
        fail = fail || reqs[i]->fail;
    }
    if (g_delay_us) std::this_thread::sleep_for(std::chrono::microseconds(g_delay_us));
    for (int i = 0; i < n; ++i)
        if (reqs[i]->ticket_word) __atomic_store_n(reqs[i]->ticket_word, reqs[i]->ticket, __ATOMIC_RELEASE);
    ++g_launches;
    g_requests += n;
    return fail ? 6 : 0;
}

int main(int argc, char** argv) {
    const int chains = argc > 1 ? atoi(argv[1]) : 64, groups = argc > 2 ? atoi(argv[2]) : 4;
    const int steps = argc > 3 ? atoi(argv[3]) : 400;
    const unsigned seed = argc > 4 ? (unsigned)atoi(argv[4]) : 1u;
    g_delay_us = argc > 5 ? atoi(argv[5]) : 0;
    if (chains > groups * QIL_LS_MAXB) {
        fprintf(stderr, "at most %d chains for %d groups\n", groups * QIL_LS_MAXB, groups);
        return 2;
    }
    g_issued = std::vector<std::atomic<long>>((size_t)chains);
    for (auto& a : g_issued) a.store(0);
    using Q = qil_chainq_t<StubReq>;
    using LS = qil_lockstep_t<StubReq, StubStream>;
    std::vector<Q> rings((size_t)chains);
    std::vector<LS> ls((size_t)groups);
    std::vector<Q*> qof((size_t)chains);
    {   // slot k belongs to group k % groups, as in qil_run_batch_on: group g owns a contiguous part of `rings`
        int off = 0;
        for (int g = 0; g < groups; ++g) {
            ls[(size_t)g].q = rings.data() + off;
            ls[(size_t)g].nslots = (chains - g + groups - 1) / groups;
            ls[(size_t)g].stream.group = g;
            for (int k = g, slot = 0; k < chains; k += groups, ++slot) qof[(size_t)k] = ls[(size_t)g].q + slot;
            off += ls[(size_t)g].nslots;
        }
    }
    std::vector<unsigned long long> words((size_t)chains, 0);
    std::vector<long> produced((size_t)chains, 0);
    std::vector<int> failed_seen((size_t)chains, 0), failed_planned((size_t)chains, 0);
    auto producer = [&](int c) {
        std::mt19937 rng(seed * 7919u + (unsigned)c);
        Q& q = *qof[(size_t)c];
        uint64_t key = 0;
        long serial = 0;
        unsigned long long ticket = 0;
        const bool will_fail = rng() % 5 == 0;
        const int fail_at = will_fail ? (int)(rng() % (unsigned)steps) : -1;
        failed_planned[(size_t)c] = will_fail;
        bool stop = false;
        auto push = [&](int cls, bool readback, bool fail) -> int {
            StubReq* r = qil_ls_begin(q, key, nullptr);
            r->kern = &g_classes[cls];
            r->block.x = 64u << (cls & 1);
            r->lds = (size_t)(rng() % 4) * 1024;
            r->launch_group = &stub_launch;
            r->chain = c;
            r->serial = ++serial;
            r->fail = fail;
            r->ticket_word = readback ? &words[(size_t)c] : nullptr;
            r->ticket = readback ? ++ticket : 0;
            return qil_ls_commit(q);
        };
        for (int s = 0; s < steps && !stop; ++s) {
            if (rng() % 3 == 0) {                                  // next site / phase
                key += 1 + rng() % 3;
                qil_ls_set_key(q, key);
            }
            const int burst = 1 + (int)(rng() % 20);
            for (int b = 0; b < burst && !stop; ++b)
                if (push((int)(rng() % 5), false, s == fail_at && b == 0) != 0) stop = true;
            if (!stop && rng() % 2 == 0) {                         // a data-dependent decision: read-back, then sleep on it
                if (push(4, true, false) != 0) stop = true;
                const int st = stop ? 6 : qil_ls_park(q, &words[(size_t)c], ticket, 99, 30.0);
                if (st == 99) ++g_errors;                          // lost wake-up / lost request
                if (st != 0) stop = true;
            }
            if (!stop && rng() % 16 == 0) qil_ls_drain(q);         // qil_stream(ctx): stream order for something else
            if (rng() % 64 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 300));   // host work
        }
        failed_seen[(size_t)c] = stop;
        produced[(size_t)c] = serial;
        qil_ls_drain(q);                                           // as the batch runner does before it marks the slot dead
        q.live.store(0, std::memory_order_release);
    };
    std::vector<std::thread> prod, launch;
    for (int g = 0; g < groups; ++g) launch.emplace_back([&, g] { qil_ls_run(&ls[(size_t)g], false); });
    for (int c = 0; c < chains; ++c) prod.emplace_back(producer, c);
    for (auto& t : prod) t.join();
    for (auto& t : launch) t.join();
    long errors = g_errors.load();
    for (int c = 0; c < chains; ++c) {
        if (g_issued[(size_t)c].load() != produced[(size_t)c]) {
            fprintf(stderr, "chain %d: produced %ld requests, %ld issued\n", c, produced[(size_t)c], g_issued[(size_t)c].load());
            ++errors;
        }
        // a launch that fails carries the status to EVERY chain of that combined launch (the whole table launch failed), so a
        // planned failure of a group neighbour may stop this chain too; stopping is an error only if no chain of the group failed
        bool group_fails = false;
        for (int k = c % groups; k < chains; k += groups) group_fails = group_fails || failed_planned[(size_t)k];
        if (failed_seen[(size_t)c] && !group_fails) {
            fprintf(stderr, "chain %d stopped although no launch of its group failed\n", c);
            ++errors;
        }
        if (failed_planned[(size_t)c] && !failed_seen[(size_t)c] && produced[(size_t)c] > 0) {
            // the failing request was queued (fail_at < steps) unless the chain had stopped earlier: it must have seen the status
            fprintf(stderr, "chain %d never saw the status of its failed launch\n", c);
            ++errors;
        }
    }
    long hist_total = 0;
    for (int g = 0; g < groups; ++g) hist_total += ls[(size_t)g].requests;
    if (hist_total != g_requests.load()) ++errors;
    printf("lockstep_stress: %d chains in %d groups, %ld requests in %ld launches (%.2f per launch), %ld errors\n", chains, groups,
           g_requests.load(), g_launches.load(), (double)g_requests.load() / (double)std::max(1L, g_launches.load()), errors);
    return errors ? 1 : 0;
}
