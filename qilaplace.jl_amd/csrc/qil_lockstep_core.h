// HIP-free core of the lock-step combiner (see qil_launch.h for what it is for): the per-chain request rings, the launcher
// loop, the park / wake protocol of chain threads that wait for a read-back.  Templates over the request type and the stream
// handle, so that the SAME code is compiled into libqilhip.so (qil_context.hip instantiates it with qil_launch_req / hipStream_t)
// and into the CPU-side stress harness tests/lockstep_stress.cpp (a stub launch function, g++ -fsanitize=thread: 64 producer
// threads, random progress keys, read-back waits, failing launches).
//
// Request type requirements: fields `kern` (identity of the kernel class), `block` (.x .y .z), `lds`, `progress`, `seq`, and
// `launch_group(Req* const* reqs, int n, Stream s) -> int` (0 = ok).
#pragma once

#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <ctime>
#include <thread>

constexpr unsigned QIL_RING = 256;       // launch requests a chain may be ahead of the launcher
constexpr int QIL_LS_MAXB = 16;          // operands per combined launch (= QIL_MAXB of qil_launch.h)

template <class Req>
struct qil_chainq_t {
    Req ring[QIL_RING];
    alignas(64) std::atomic<unsigned> head{0};     // next request to issue (launcher)
    alignas(64) std::atomic<unsigned> tail{0};     // next free entry (the chain's thread)
    std::atomic<uint64_t> key{0};                  // where the chain is working (mirrors the context's progress key)
    std::atomic<unsigned> seq{0};                  // launches it has queued since the key last changed
    std::atomic<int> live{1};
    std::atomic<int> status{0};                    // first failed launch of this chain
    // a chain thread that waits for a read-back SLEEPS here (futex) and the group's launcher, which polls anyway, watches the
    // ticket word for it: the GPU boxes give a process a CPU quota (16 CPUs), and 32 chain threads spinning on their tickets
    // exhaust it -- every thread is then throttled for the rest of the scheduler period (measured: three 45-55 ms stalls of
    // all four queues per 32-chain batch)
    std::atomic<uint32_t> parked{0};
    std::atomic<const unsigned long long*> wait_word{nullptr};
    std::atomic<unsigned long long> wait_ticket{0};
};

template <class Req, class Stream>
struct qil_lockstep_t {
    int nslots = 0;
    qil_chainq_t<Req>* q = nullptr;
    Stream stream{};                               // the one stream all slots share
    long long requests = 0, launches = 0, timeouts = 0;
    double launch_us = 0, total_us = 0;            // QIL_BATCH_DEBUG: time inside the launch calls / of the launcher loop
    long long group_hist[QIL_LS_MAXB + 1] = {};
};

static inline void qil_futex_wait_for(std::atomic<uint32_t>* a, uint32_t expected, long timeout_ns) {
    timespec ts{0, timeout_ns};
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(a), FUTEX_WAIT_PRIVATE, expected, &ts, nullptr, 0);
}
static inline void qil_futex_wake_one(std::atomic<uint32_t>* a) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(a), FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0);
}
static inline void qil_spin_pause(int& spins) {
    if (++spins < (1 << 14))
        __builtin_ia32_pause();
    else
        std::this_thread::yield();
}

// chain side: sleep until *word >= ticket (the launcher wakes the thread; the futex timeout only bounds a lost wake-up).
// Returns 0 when the ticket has arrived, the chain's failed-launch status if a combined launch of this chain failed meanwhile and
// everything it queued has been consumed (its read-back kernel may never run), `timeout_code` after `timeout_s` seconds -- the
// caller must not wait for ever on a device that has faulted.
template <class Req>
int qil_ls_park(qil_chainq_t<Req>& q, const unsigned long long* word, unsigned long long ticket, int timeout_code, double timeout_s = 60.0) {
    const auto t0 = std::chrono::steady_clock::now();
    while (__atomic_load_n(word, __ATOMIC_ACQUIRE) < ticket) {
        const int st = q.status.load(std::memory_order_acquire);
        if (st != 0 && q.head.load(std::memory_order_acquire) == q.tail.load(std::memory_order_relaxed)) return st;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return timeout_code;
        q.wait_word.store(word, std::memory_order_relaxed);
        q.wait_ticket.store(ticket, std::memory_order_relaxed);
        q.parked.store(1, std::memory_order_release);
        if (__atomic_load_n(word, __ATOMIC_ACQUIRE) >= ticket) {
            q.parked.store(0, std::memory_order_relaxed);
            break;
        }
        qil_futex_wait_for(&q.parked, 1, 20 * 1000 * 1000);
        q.parked.store(0, std::memory_order_relaxed);
    }
    return 0;
}
// launcher side: wake the chains whose tickets have arrived
template <class Req, class Stream>
void qil_ls_wake_arrived(qil_lockstep_t<Req, Stream>* ls) {
    for (int s = 0; s < ls->nslots; ++s) {
        qil_chainq_t<Req>& q = ls->q[s];
        if (q.parked.load(std::memory_order_acquire) == 1) {
            const unsigned long long* w = q.wait_word.load(std::memory_order_relaxed);
            if (w && __atomic_load_n(w, __ATOMIC_ACQUIRE) >= q.wait_ticket.load(std::memory_order_relaxed)) {
                uint32_t one = 1;
                if (q.parked.compare_exchange_strong(one, 2, std::memory_order_acq_rel)) qil_futex_wake_one(&q.parked);
            }
        }
    }
}

// chain side: the next ring entry to fill (waits while the ring is full; *ring_wait_us accumulates that wait when not null)
template <class Req>
Req* qil_ls_begin(qil_chainq_t<Req>& q, uint64_t progress_key, double* ring_wait_us) {
    const unsigned t = q.tail.load(std::memory_order_relaxed);
    int spins = 0;
    if (t - q.head.load(std::memory_order_acquire) >= QIL_RING) {                       // ring full
        const auto t0 = std::chrono::steady_clock::now();
        while (t - q.head.load(std::memory_order_acquire) >= QIL_RING) qil_spin_pause(spins);
        if (ring_wait_us) *ring_wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    Req* r = &q.ring[t % QIL_RING];
    r->progress = progress_key;
    r->seq = q.seq.load(std::memory_order_relaxed);
    return r;
}
template <class Req>
int qil_ls_commit(qil_chainq_t<Req>& q) {                    // publishes the entry; returns the chain's sticky launch status
    q.seq.store(q.seq.load(std::memory_order_relaxed) + 1, std::memory_order_relaxed);
    q.tail.store(q.tail.load(std::memory_order_relaxed) + 1, std::memory_order_release);
    return q.status.load(std::memory_order_relaxed);
}
template <class Req>
void qil_ls_drain(qil_chainq_t<Req>& q) {                    // until the launcher has issued everything this chain queued
    const unsigned t = q.tail.load(std::memory_order_relaxed);
    int spins = 0;
    while (q.head.load(std::memory_order_acquire) != t) qil_spin_pause(spins);
}
template <class Req>
void qil_ls_set_key(qil_chainq_t<Req>& q, uint64_t key) {    // the chain moves on to another site / phase
    q.seq.store(0, std::memory_order_relaxed);
    q.key.store(key, std::memory_order_release);
}

// the launcher: until every chain has left and every ring is empty.  A launch goes out when every chain that is no further
// along than the head to be issued has queued its own next step (so that they can share the launch) -- or after `patience` of
// waiting for such a chain (it may be deep in host work or waiting for the device).
template <class Req, class Stream>
void qil_ls_run(qil_lockstep_t<Req, Stream>* ls, bool timing) {
    int spins = 0;
    const auto t_begin = std::chrono::steady_clock::now();
    // (measured, 8 chains chi 256: 50 us -> 7441 launches / 161 ms, 200 us -> 7039 / 160 ms, 1 ms -> 6979 / 155 ms, 5 ms -> 6976 / 146 ms)
    const auto patience = std::chrono::microseconds(2000);
    bool waiting = false;
    std::chrono::steady_clock::time_point wait_since;
    for (;;) {
        qil_ls_wake_arrived(ls);
        // order = (progress key, position inside the key's segment): chains running the same program queue the same kernel
        // at the same position, so serving the smallest position first re-aligns chains that are one step apart
        uint64_t headkey = ~0ull, idlekey = ~0ull;
        unsigned headseq = ~0u, idleseq = ~0u;
        bool any = false;
        unsigned heads[QIL_LS_MAXB], tails[QIL_LS_MAXB];
        int lead = -1;
        for (int s = 0; s < ls->nslots; ++s) {
            qil_chainq_t<Req>& q = ls->q[s];
            const int live = q.live.load(std::memory_order_acquire);
            tails[s] = q.tail.load(std::memory_order_acquire);
            heads[s] = q.head.load(std::memory_order_relaxed);
            if (heads[s] != tails[s]) {
                any = true;
                const Req& r = q.ring[heads[s] % QIL_RING];
                if (r.progress < headkey || (r.progress == headkey && r.seq < headseq)) {
                    headkey = r.progress;
                    headseq = r.seq;
                    lead = s;
                }
            } else if (live) {
                any = true;
                const uint64_t k = q.key.load(std::memory_order_acquire);
                const unsigned sq = q.seq.load(std::memory_order_relaxed);
                if (k < idlekey || (k == idlekey && sq < idleseq)) {
                    idlekey = k;
                    idleseq = sq;
                }
            }
        }
        if (!any) break;
        if (lead < 0) {                                         // nothing queued anywhere
            qil_spin_pause(spins);
            continue;
        }
        if (idlekey < headkey || (idlekey == headkey && idleseq <= headseq)) {   // a chain that is not ahead has not queued this step yet
            const auto now = std::chrono::steady_clock::now();
            if (!waiting) {
                waiting = true;
                wait_since = now;
            }
            if (now - wait_since < patience) {
                __builtin_ia32_pause();
                continue;
            }
            ++ls->timeouts;
        }
        waiting = false;
        spins = 0;
        // The furthest-behind head decides the kernel class of this launch, and EVERY ring head of that class rides it -- also
        // the heads of chains that are further along.  (Until r04 only heads at exactly the same (key, position) were combined:
        // chains of different shapes -- the 64 (operator, state) pairs of a damping sweep: different bond dimensions, different
        // split-K decisions and sweep counts -- are almost never at the same position, and 88 % of their launches carried ONE
        // request: 240 k launches per batch of 64 pairs, the four streams launch-rate-bound at 4 us each.  A head that is issued
        // early keeps its chain's own order; chains that are ahead advance only while their next kernel is of the class the
        // laggard needs, which is what re-aligns them.  Requests of one class may differ in dynamic LDS: the launch takes the
        // largest, qil_launch_group.)
        Req* lr = &ls->q[lead].ring[heads[lead] % QIL_RING];
        Req* grp[QIL_LS_MAXB];
        int gs[QIL_LS_MAXB], n = 0;
        for (int s = 0; s < ls->nslots && n < QIL_LS_MAXB; ++s) {
            if (heads[s] == tails[s]) continue;
            Req* r = &ls->q[s].ring[heads[s] % QIL_RING];
            if (r->kern != lr->kern || r->block.x != lr->block.x || r->block.y != lr->block.y || r->block.z != lr->block.z) continue;
            gs[n] = s;
            grp[n++] = r;
        }
        const auto tl0 = timing ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        const int st = lr->launch_group(grp, n, ls->stream);
        if (timing) ls->launch_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tl0).count();
        ++ls->launches;
        ls->requests += n;
        ++ls->group_hist[n];
        for (int k = 0; k < n; ++k) {
            qil_chainq_t<Req>& q = ls->q[gs[k]];
            if (st != 0) {
                int ok = 0;
                q.status.compare_exchange_strong(ok, st);
            }
            q.head.store(q.head.load(std::memory_order_relaxed) + 1, std::memory_order_release);
        }
    }
    ls->total_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count();
}
