// Context, error handling, caching device pool, containers (T1-T3 of SURVEY.md section 8a).
#include <cstdarg>

#include "qil_internal.h"
#include "qil_launch.h"
#include "qil_lockstep_core.h"

#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <memory>
#include <chrono>
#include <system_error>
#include <thread>

// ---------------------------------------------------------------- errors
static thread_local char g_err[1024] = "";
static thread_local unsigned g_fail_count = 0;
unsigned qil_fail_count() { return g_fail_count; }

void qil_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int qil_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    ++g_fail_count;
    return code;
}

extern "C" const char* qil_last_error(void) { return g_err; }
extern "C" const char* qil_version(void) { return "qilhip 0.1.0 (gfx950)"; }

extern "C" int qil_device_count(int* out) {
    QIL_REQUIRE(out, QIL_EINVAL_ARG, "qil_device_count: null out");
    QIL_HIP(hipGetDeviceCount(out));
    return QIL_OK;
}

// ---------------------------------------------------------------- context
int qil_ctx_activate(qil_context* ctx) {
    QIL_REQUIRE(ctx, QIL_EINVAL_ARG, "null context");
    QIL_HIP(hipSetDevice(ctx->device));
    return QIL_OK;
}

extern "C" int qil_context_create(int device, void* stream, qil_context** out) {
    QIL_REQUIRE(out, QIL_EINVAL_ARG, "qil_context_create: null out");
    int count = 0;
    QIL_HIP(hipGetDeviceCount(&count));
    QIL_REQUIRE(device >= 0 && device < count, QIL_EINVAL_ARG,
                "qil_context_create: device %d out of range (%d visible)", device, count);
    QIL_HIP(hipSetDevice(device));
    qil_context* ctx = new qil_context();
    ctx->device = device;
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete ctx;
            return qil_fail(QIL_EHIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
        }
        ctx->owns_stream = true;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
    hipEventCreate(&ctx->t0);
    hipEventCreate(&ctx->t1);
    *out = ctx;
    return QIL_OK;
}

extern "C" int qil_context_synchronize(qil_context* ctx) {
    QIL_TRY(qil_ctx_activate(ctx));
    QIL_HIP(qil_stream_sync(ctx));
    return QIL_OK;
}

extern "C" int qil_context_trim(qil_context* ctx) {
    QIL_TRY(qil_ctx_activate(ctx));
    QIL_HIP(qil_stream_sync(ctx));
    for (auto& kv : ctx->free_blocks) hipFree(kv.second);
    ctx->free_blocks.clear();
    ctx->bytes_cached = 0;
    for (qil_context* w : ctx->workers) QIL_TRY(qil_context_trim(w));
    return QIL_OK;
}

extern "C" int qil_context_destroy(qil_context* ctx) {
    if (!ctx) return QIL_OK;
    for (qil_context* w : ctx->workers) qil_context_destroy(w);
    ctx->workers.clear();
    hipSetDevice(ctx->device);
    qil_stream_sync(ctx);
    for (qil_comm* cm : ctx->comms) qil_comm_orphan(cm);  // communicators first: their handles stay valid and empty
    ctx->comms.clear();
    for (qil_chain* c : ctx->chains) {                    // handles the caller has not destroyed yet: orphan them
        c->ctx = nullptr;
        for (void*& p : c->site) p = nullptr;
    }
    ctx->chains.clear();
    for (auto& kv : ctx->free_blocks) hipFree(kv.second);
    for (auto& kv : ctx->live_blocks) hipFree(kv.first);  // ... their blocks go with the pool
    if (ctx->pinned) hipHostFree(ctx->pinned);
    if (ctx->dev_scratch) hipFree(ctx->dev_scratch);
    if (ctx->desc_host) hipHostFree(ctx->desc_host);
    if (ctx->desc_dev) hipFree(ctx->desc_dev);
    for (int i = 0; i < qil_context::kDescSlots; ++i)
        if (ctx->desc_event[i]) hipEventDestroy(ctx->desc_event[i]);
    for (auto& pr : ctx->prof_events) {
        hipEventDestroy(pr.first);
        hipEventDestroy(pr.second);
    }
    for (auto e : ctx->event_pool) hipEventDestroy(e);
    if (ctx->rb_host) hipHostFree(ctx->rb_host);
    if (ctx->st_host) hipHostFree(ctx->st_host);
    if (ctx->st_dev) hipFree(ctx->st_dev);
    if (ctx->sync_event) hipEventDestroy(ctx->sync_event);
    if (ctx->t0) hipEventDestroy(ctx->t0);
    if (ctx->t1) hipEventDestroy(ctx->t1);
    if (ctx->owns_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return QIL_OK;
}

extern "C" int qil_context_mem_info(qil_context* ctx, int64_t* in_use, int64_t* cached, int64_t* dfree,
                                    int64_t* dtotal) {
    QIL_TRY(qil_ctx_activate(ctx));
    size_t f = 0, t = 0;
    QIL_HIP(hipMemGetInfo(&f, &t));
    size_t used = ctx->bytes_in_use, held = ctx->bytes_cached + ctx->lend_cached;
    for (const qil_context* w : ctx->workers) {             // the batch workers' pools belong to this context
        used += w->bytes_in_use;
        held += w->bytes_cached;
    }
    if (in_use) *in_use = (int64_t)used;
    if (cached) *cached = (int64_t)held;
    if (dfree) *dfree = (int64_t)f;
    if (dtotal) *dtotal = (int64_t)t;
    return QIL_OK;
}

extern "C" int qil_context_unowned_bytes(qil_context* ctx, int64_t* out) {
    QIL_REQUIRE(ctx && out, QIL_EINVAL_ARG, "null argument");
    int64_t tot = 0;
    for (const auto& kv : ctx->live_blocks)
        if (!kv.second.owned) tot += (int64_t)kv.second.bytes;
    *out = tot;
    return QIL_OK;
}

extern "C" int qil_context_fail_alloc_after(qil_context* ctx, int64_t n) {
    QIL_REQUIRE(ctx, QIL_EINVAL_ARG, "null context");
    ctx->fail_alloc_countdown = n;
    return QIL_OK;
}

int qil_ctx_alloc(qil_context* ctx, size_t bytes, void** out) {
    if (ctx->fail_alloc_countdown >= 0 && ctx->fail_alloc_countdown-- == 0)
        return qil_fail(QIL_ENOMEM, "injected allocation failure (qil_context_fail_alloc_after)");
    if (bytes == 0) bytes = 16;
    bytes = (bytes + 255) & ~(size_t)255;
    struct alloc_timer {                                         // QIL_BATCH_DEBUG accounting
        qil_context* c;
        std::chrono::steady_clock::time_point t0;
        explicit alloc_timer(qil_context* cc) : c(cc) {
            if (c->dbg_times) t0 = std::chrono::steady_clock::now();
        }
        ~alloc_timer() {
            if (c->dbg_times) {
                c->dbg_alloc_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                ++c->dbg_alloc_n;
            }
        }
    } timer(ctx);
    auto it = ctx->free_blocks.find(bytes);
    if (it != ctx->free_blocks.end()) {
        *out = it->second;
        ctx->free_blocks.erase(it);
        ctx->bytes_cached -= bytes;
    } else if ([&]() {
                   qil_context* lender = ctx->parent ? ctx->parent : ctx;
                   if (!lender->lending) return false;
                   std::lock_guard<std::mutex> lock(lender->pool_mutex);
                   auto pit = lender->lend_blocks.find(bytes);
                   if (pit == lender->lend_blocks.end()) return false;
                   *out = pit->second;
                   lender->lend_blocks.erase(pit);
                   lender->lend_cached -= bytes;
                   return true;
               }()) {
        // taken from the blocks the home context lends for the duration of a batch
    } else {
        ++ctx->dbg_alloc_miss;
        hipError_t e = hipMalloc(out, bytes);
        if (e == hipErrorOutOfMemory && !ctx->free_blocks.empty()) {
            // give cached blocks back and retry once
            (void)hipGetLastError();
            qil_stream_sync(ctx);
            for (auto& kv : ctx->free_blocks) hipFree(kv.second);
            ctx->free_blocks.clear();
            ctx->bytes_cached = 0;
            e = hipMalloc(out, bytes);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return qil_fail(e == hipErrorOutOfMemory ? QIL_ENOMEM : QIL_EHIP,
                            "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        }
    }
    ctx->live_blocks[*out] = qil_context::live_block{bytes, ++ctx->alloc_serial, false};
    ctx->bytes_in_use += bytes;
    return QIL_OK;
}

// Blocks are recycled in stream order (single stream per context), so a block freed here can be
// handed out again immediately: work that still reads it was enqueued earlier on the same stream.
int qil_ctx_free(qil_context* ctx, void* p) {
    if (!p) return QIL_OK;
    auto it = ctx->live_blocks.find(p);
    if (it == ctx->live_blocks.end()) {
        // not counted as a call failure: cleanup code may free a block the error path already reclaimed
        qil_set_error("qil_ctx_free: unknown block");
        return QIL_EINVAL_ARG;
    }
    size_t bytes = it->second.bytes;
    ctx->live_blocks.erase(it);
    ctx->bytes_in_use -= bytes;
    ctx->free_blocks.emplace(bytes, p);
    ctx->bytes_cached += bytes;
    // The cache is keyed by exact size: a long run over data-dependent shapes (truncated bonds) would pile up small
    // blocks of ever new sizes.  Past kMaxCachedBlocks entries the SMALL ones (cheap to allocate again) go back to
    // the driver; the large recurring ones (apply outputs) stay, they are what the cache is for.
    constexpr size_t kMaxCachedBlocks = 8192, kSmall = 1u << 20;
    if (ctx->free_blocks.size() > kMaxCachedBlocks) {
        (void)qil_stream_sync(ctx);
        auto end = ctx->free_blocks.upper_bound(kSmall);
        for (auto f = ctx->free_blocks.begin(); f != end; ++f) {
            (void)hipFree(f->second);
            ctx->bytes_cached -= f->first;
        }
        ctx->free_blocks.erase(ctx->free_blocks.begin(), end);
    }
    return QIL_OK;
}

static void mark_owned(qil_context* ctx, void* p) {
    auto it = ctx->live_blocks.find(p);
    if (it != ctx->live_blocks.end()) it->second.owned = true;
}

qil_call_scope::qil_call_scope(qil_context* c)
    : ctx(c), serial0(c ? c->alloc_serial : 0), fails0(qil_fail_count()) {
    // per-call heuristics state starts afresh: what one call learns about its operands (qr_impl's Cholesky QR refusals) must not
    // leak into the next one -- an item of a batch then takes exactly the route it takes alone (bit-identical results)
    if (c) {
        c->cholqr_skip = 0;
        // ... and so does the R^-1 block CholeskyQR2 parks for a certificate (cholqr2 / certify_no_truncation): it is state of
        // ONE call.  A pointer that survived the previous call is released only if the pool still knows THAT allocation
        // (same address AND same allocation serial, not owned by a handle: the address alone may have been handed out again,
        // ADVICE r04), and forgotten either way, so no call can free or certify against somebody else's block.
        if (c->rinv) {
            auto it = c->live_blocks.find(c->rinv);
            if (it != c->live_blocks.end() && it->second.serial == c->rinv_serial && !it->second.owned) qil_ctx_free(c, c->rinv);
        }
        c->rinv = nullptr;
        c->rinv_for = nullptr;
        c->want_rinv = false;
    }
}

qil_call_scope::~qil_call_scope() {
    if (!ctx || qil_fail_count() == fails0) return;
    // the call failed: nothing it allocated for itself may survive it.  Work that still touches these blocks
    // was enqueued on the context's stream, and the pool recycles in stream order, so this is safe without
    // waiting.
    for (auto it = ctx->live_blocks.begin(); it != ctx->live_blocks.end();) {
        if (it->second.serial > serial0 && !it->second.owned) {
            ctx->bytes_in_use -= it->second.bytes;
            ctx->free_blocks.emplace(it->second.bytes, it->first);
            ctx->bytes_cached += it->second.bytes;
            it = ctx->live_blocks.erase(it);
        } else {
            ++it;
        }
    }
    // the parked R^-1 block (if any) has just gone back to the pool with the rest: forget the pointer, do not free it again
    ctx->rinv = nullptr;
    ctx->rinv_for = nullptr;
    ctx->want_rinv = false;
}

int qil_ctx_pinned(qil_context* ctx, size_t bytes, void** out) {
    if (bytes > ctx->pinned_bytes) {
        QIL_HIP(qil_stream_sync(ctx));
        if (ctx->pinned) hipHostFree(ctx->pinned);
        ctx->pinned = nullptr;
        size_t nb = bytes < (1u << 16) ? (1u << 16) : bytes * 2;
        QIL_HIP(hipHostMalloc(&ctx->pinned, nb, hipHostMallocDefault));
        ctx->pinned_bytes = nb;
    }
    *out = ctx->pinned;
    return QIL_OK;
}

int qil_ctx_dev_scratch(qil_context* ctx, size_t bytes, void** out) {
    if (bytes > ctx->dev_scratch_bytes) {
        QIL_HIP(qil_stream_sync(ctx));
        if (ctx->dev_scratch) hipFree(ctx->dev_scratch);
        ctx->dev_scratch = nullptr;
        size_t nb = bytes < (1u << 16) ? (1u << 16) : bytes * 2;
        QIL_HIP(hipMalloc(&ctx->dev_scratch, nb));
        ctx->dev_scratch_bytes = nb;
    }
    *out = ctx->dev_scratch;
    return QIL_OK;
}

int qil_ctx_desc_acquire(qil_context* ctx, size_t bytes, void** host, void** dev, int* slot) {
    QIL_REQUIRE(bytes <= qil_context::kDescSlotBytes, QIL_EINVAL_ARG,
                "descriptor table of %zu bytes exceeds the slot size", bytes);
    if (!ctx->desc_host) {
        const size_t tot = qil_context::kDescSlotBytes * qil_context::kDescSlots;
        QIL_HIP(hipHostMalloc(&ctx->desc_host, tot, hipHostMallocDefault));
        QIL_HIP(hipMalloc(&ctx->desc_dev, tot));
        for (int i = 0; i < qil_context::kDescSlots; ++i)
            QIL_HIP(hipEventCreateWithFlags(&ctx->desc_event[i], hipEventDisableTiming));
    }
    const int k = ctx->desc_next;
    ctx->desc_next = (k + 1) % qil_context::kDescSlots;
    if (ctx->desc_used[k]) QIL_HIP(hipEventSynchronize(ctx->desc_event[k]));
    *host = static_cast<char*>(ctx->desc_host) + (size_t)k * qil_context::kDescSlotBytes;
    *dev = static_cast<char*>(ctx->desc_dev) + (size_t)k * qil_context::kDescSlotBytes;
    *slot = k;
    return QIL_OK;
}

int qil_ctx_desc_commit(qil_context* ctx, int slot) {
    QIL_HIP(hipEventRecord(ctx->desc_event[slot], qil_stream(ctx)));
    ctx->desc_used[slot] = true;
    return QIL_OK;
}


// ---------------------------------------------------------------- timers / profile
extern "C" int qil_timer_start(qil_context* ctx) {
    QIL_TRY(qil_ctx_activate(ctx));
    QIL_HIP(hipEventRecord(ctx->t0, qil_stream(ctx)));
    return QIL_OK;
}

extern "C" int qil_timer_stop(qil_context* ctx, double* ms) {
    QIL_TRY(qil_ctx_activate(ctx));
    QIL_REQUIRE(ms, QIL_EINVAL_ARG, "qil_timer_stop: null out");
    QIL_HIP(hipEventRecord(ctx->t1, qil_stream(ctx)));
    QIL_HIP(hipEventSynchronize(ctx->t1));
    float f = 0.f;
    QIL_HIP(hipEventElapsedTime(&f, ctx->t0, ctx->t1));
    *ms = (double)f;
    return QIL_OK;
}

// ---- the box's own store-only ceiling: what a kernel that does nothing but write 16 B per lane reaches on
// THIS GPU, so that a roofline fraction measured on one box of the pool can be compared with one measured on another (the
// apply's figure moved 0.78 ... 0.85 of the 8 TB/s spec between boxes in r05).  Four writers, the best one is reported:
// hipMemsetAsync, one contiguous 256 KiB span per workgroup with plain / non-temporal stores, a grid-stride fill.
namespace {
typedef double qil_fill_d2 __attribute__((ext_vector_type(2)));
template <bool NT>
__global__ __launch_bounds__(256) void hbm_fill_span(qil_fill_d2* __restrict__ p, long long span, double v) {
    const qil_fill_d2 val{v, v + 1.0};
    qil_fill_d2* q = p + blockIdx.x * span;
    for (long long i = threadIdx.x; i < span; i += 256) {
        if (NT) __builtin_nontemporal_store(val, q + i);
        else q[i] = val;
    }
}
__global__ __launch_bounds__(256) void hbm_fill_stride(qil_fill_d2* __restrict__ p, long long n, double v) {
    const qil_fill_d2 val{v, v + 1.0};
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = val;
}
}  // namespace

extern "C" int qil_hbm_store_peak(qil_context* ctx, int64_t bytes, int reps, double* best_gbs, int* best_kind) {
    QIL_REQUIRE(ctx && best_gbs, QIL_EINVAL_ARG, "qil_hbm_store_peak: null argument");
    QIL_REQUIRE(bytes >= (1 << 26) && reps >= 1 && reps <= 100, QIL_EINVAL_ARG, "qil_hbm_store_peak: bytes >= 64 MiB, 1 <= reps <= 100");
    QIL_TRY(qil_ctx_activate(ctx));
    qil_call_scope call_scope(ctx);
    const long long span = 256 * 1024 / 16;                           // 16-B elements per workgroup span
    const long long n = bytes / 16 / span * span;
    void* buf = nullptr;
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)n * 16, &buf));
    qil_fill_d2* d = static_cast<qil_fill_d2*>(buf);
    hipStream_t s = qil_stream(ctx);
    auto launch = [&](int kind) {
        switch (kind) {
            case 0: (void)hipMemsetAsync(d, 0, (size_t)n * 16, s); break;
            case 1: hipLaunchKernelGGL(hbm_fill_span<false>, dim3((unsigned)(n / span)), dim3(256), 0, s, d, span, 1.0); break;
            case 2: hipLaunchKernelGGL(hbm_fill_span<true>, dim3((unsigned)(n / span)), dim3(256), 0, s, d, span, 1.0); break;
            default: hipLaunchKernelGGL(hbm_fill_stride, dim3(32768), dim3(256), 0, s, d, n, 1.0); break;
        }
    };
    double best = 0.0;
    int which = 0;
    for (int kind = 0; kind < 4; ++kind) {
        launch(kind);                                                 // untimed first touch
        QIL_HIP(hipEventRecord(ctx->t0, s));
        for (int r = 0; r < reps; ++r) launch(kind);
        QIL_HIP(hipEventRecord(ctx->t1, s));
        QIL_HIP(hipEventSynchronize(ctx->t1));
        QIL_HIP(hipGetLastError());
        float ms = 0.f;
        QIL_HIP(hipEventElapsedTime(&ms, ctx->t0, ctx->t1));
        const double gbs = ms > 0 ? (double)n * 16 * reps / (ms * 1e-3) / 1e9 : 0.0;
        if (gbs > best) {
            best = gbs;
            which = kind;
        }
    }
    qil_ctx_free(ctx, buf);
    *best_gbs = best;
    if (best_kind) *best_kind = which;
    return QIL_OK;
}

extern "C" int qil_profile_enable(qil_context* ctx, int on) {
    QIL_REQUIRE(ctx, QIL_EINVAL_ARG, "null context");
    ctx->profile = on != 0;
    return QIL_OK;
}

static int get_event(qil_context* ctx, hipEvent_t* e) {
    if (!ctx->event_pool.empty()) {
        *e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return QIL_OK;
    }
    QIL_HIP(hipEventCreate(e));
    return QIL_OK;
}

int qil_ctx_event(qil_context* ctx, hipEvent_t* e) { return get_event(ctx, e); }
void qil_ctx_event_release(qil_context* ctx, hipEvent_t e) {
    if (e) ctx->event_pool.push_back(e);
}
int qil_ctx_prof_begin(qil_context* ctx) {
    if (!ctx->profile) return QIL_OK;
    hipEvent_t a, b;
    QIL_TRY(get_event(ctx, &a));
    QIL_TRY(get_event(ctx, &b));
    ctx->prof_events.emplace_back(a, b);
    QIL_HIP(hipEventRecord(a, qil_stream(ctx)));
    return QIL_OK;
}

int qil_ctx_prof_end(qil_context* ctx) {
    if (!ctx->profile) return QIL_OK;
    QIL_HIP(hipEventRecord(ctx->prof_events.back().second, qil_stream(ctx)));
    return QIL_OK;
}

extern "C" int qil_profile_read(qil_context* ctx, int64_t* n_launches, double* total_ms, int reset) {
    QIL_TRY(qil_ctx_activate(ctx));
    QIL_HIP(qil_stream_sync(ctx));
    double tot = 0;
    for (auto& pr : ctx->prof_events) {
        float f = 0.f;
        QIL_HIP(hipEventElapsedTime(&f, pr.first, pr.second));
        tot += f;
    }
    if (n_launches) *n_launches = (int64_t)ctx->prof_events.size();
    if (total_ms) *total_ms = tot;
    if (reset) {
        for (auto& pr : ctx->prof_events) {
            ctx->event_pool.push_back(pr.first);
            ctx->event_pool.push_back(pr.second);
        }
        ctx->prof_events.clear();
    }
    return QIL_OK;
}

// ---------------------------------------------------------------- containers
int qil_chain_alloc(qil_context* ctx, qil_chain* c, int64_t n, int dtype, int paired, int phys_rank,
                    const int64_t* bond_dims, const int64_t* site_ids) {
    QIL_REQUIRE(ctx, QIL_EINVAL_ARG, "null context");
    QIL_REQUIRE(n >= 1, QIL_EINVAL_LENGTH, "a tensor chain needs at least one site (got %lld)", (long long)n);
    QIL_REQUIRE(dtype == QIL_F64 || dtype == QIL_C64, QIL_EINVAL_ARG, "unknown dtype %d", dtype);
    QIL_REQUIRE(!paired || n % 2 == 0, QIL_EINVAL_LENGTH,
                "paired chains need an even number of tensors (got %lld)", (long long)n);
    QIL_REQUIRE(n == 1 || bond_dims, QIL_EINVAL_ARG, "null bond_dims");
    qil_chain_bind(c, ctx);
    c->dtype = dtype;
    c->paired = paired ? 1 : 0;
    c->phys_rank = phys_rank;
    c->dims.assign((size_t)n + 1, 1);
    for (int64_t i = 0; i + 1 < n; ++i) {
        QIL_REQUIRE(bond_dims[i] >= 1, QIL_EINVAL_ARG, "bond %lld has dimension %lld", (long long)(i + 1),
                    (long long)bond_dims[i]);
        c->dims[(size_t)i + 1] = bond_dims[i];
    }
    c->site_ids.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) c->site_ids[(size_t)i] = site_ids ? site_ids[i] : i + 1;
    c->site.assign((size_t)n, nullptr);
    QIL_TRY(qil_ctx_activate(ctx));
    for (int64_t i = 0; i < n; ++i) {
        int s = qil_ctx_alloc(ctx, c->site_bytes(i), &c->site[(size_t)i]);
        if (s != QIL_OK) {
            qil_chain_release(c);
            return s;
        }
        mark_owned(ctx, c->site[(size_t)i]);
    }
    return QIL_OK;
}

// ---------------------------------------------------------------- batches of independent chains
static void chain_move(qil_chain* c, qil_context* to);

void qil_ctx_transfer(qil_context* from, qil_context* to, void* p) {
    if (from == to || !p) return;
    auto it = from->live_blocks.find(p);
    if (it == from->live_blocks.end()) return;
    qil_context::live_block b = it->second;
    from->live_blocks.erase(it);
    from->bytes_in_use -= b.bytes;
    b.serial = ++to->alloc_serial;
    to->live_blocks.emplace(p, b);
    to->bytes_in_use += b.bytes;
}

void qil_chain_rebind(qil_chain* c, qil_context* to) { chain_move(c, to); }

static void chain_move(qil_chain* c, qil_context* to) {
    qil_context* from = c->ctx;
    for (void* p : c->site)
        if (p) qil_ctx_transfer(from, to, p);
    from->chains.erase(c);
    c->ctx = to;
    to->chains.insert(c);
}

// ---------------------------------------------------------------- lock-step batches (see qil_launch.h)
// The rings, the launcher loop and the park / wake protocol are the HIP-free templates of qil_lockstep_core.h (the same code
// runs under ThreadSanitizer in tests/lockstep_stress.cpp); here they are bound to qil_launch_req and the HIP stream.
static_assert(QIL_LS_MAXB == QIL_MAXB, "qil_lockstep_core.h and qil_launch.h disagree on the operands per combined launch");
using qil_chainq = qil_chainq_t<qil_launch_req>;
struct qil_lockstep : qil_lockstep_t<qil_launch_req, hipStream_t> {};

void qil_progress_step(qil_context* ctx, bool new_pass, long long step) {
    if (!ctx) return;
    uint64_t pass = ctx->progress_key >> 40;
    if (new_pass) ++pass;
    ctx->progress_key = (pass << 40) | ((uint64_t)(step & 0xfffff) << 16);
    if (ctx->lockstep) qil_ls_set_key(ctx->lockstep->q[ctx->ls_slot], ctx->progress_key);
}
void qil_progress_phase(qil_context* ctx, int phase) {
    if (!ctx) return;
    const uint64_t nk = (ctx->progress_key & ~0xffffull) | (uint64_t)(phase & 0xffff);
    if (nk == ctx->progress_key) return;
    ctx->progress_key = nk;
    if (ctx->lockstep) qil_ls_set_key(ctx->lockstep->q[ctx->ls_slot], nk);
}

// chain side: sleep until *word >= ticket.  QIL_OK, the chain's failed-launch status, or QIL_EHIP after 60 s.
int qil_lockstep_park(qil_context* ctx, const unsigned long long* word, unsigned long long ticket) {
    return qil_ls_park(ctx->lockstep->q[ctx->ls_slot], word, ticket, QIL_EHIP);
}

qil_launch_req* qil_lockstep_begin(qil_lockstep* ls, qil_context* ctx) {
    return qil_ls_begin(ls->q[ctx->ls_slot], ctx->progress_key, ctx->dbg_times ? &ctx->dbg_ring_us : nullptr);
}
int qil_lockstep_commit(qil_lockstep* ls, qil_context* ctx) {
    const int st = qil_ls_commit(ls->q[ctx->ls_slot]);
    return st == QIL_OK ? QIL_OK : qil_fail(st, "a combined launch of this chain failed");
}
void qil_lockstep_drain(qil_context* ctx) {
    if (ctx->lockstep) qil_ls_drain(ctx->lockstep->q[ctx->ls_slot]);
}

hipError_t qil_stream_sync(qil_context* ctx) {
    if (!ctx->lockstep) return hipStreamSynchronize(ctx->stream);
    qil_lockstep_drain(ctx);
    if (!ctx->sync_event) {
        const hipError_t e = hipEventCreateWithFlags(&ctx->sync_event, hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    hipError_t e = hipEventRecord(ctx->sync_event, ctx->stream);
    if (e != hipSuccess) return e;
    return qil_event_sync(ctx, ctx->sync_event);
}

// (Letting the launcher go on without a chain that is parked in such a wait was measured and dropped: the chains fall out of
// step -- 8 / 16 / 32 chains 100 / 152 / 340 ms against 94 / 133 / 302 ms.)
hipError_t qil_event_sync(qil_context* ctx, hipEvent_t ev) {
    if (ctx->lockstep) qil_lockstep_drain(ctx);
    return hipEventSynchronize(ev);
}

static void lockstep_run(qil_lockstep* ls, bool timing) { qil_ls_run(ls, timing); }

// CPUs this process may keep busy: the cgroup quota (cpu.max, v2; cfs_quota_us / cfs_period_us, v1) or the affinity mask, whichever
// is smaller, divided by the ranks that share the node (LOCAL_WORLD_SIZE of torch.distributed.run / bench.py's spawner);
// QIL_CPU_BUDGET overrides.  The launchers of a lock-step batch POLL, so their number must stay below it: the GPU boxes give a
// process 16 CPUs of quota, and 8 ranks x (4 launchers + a chain thread awake) exceed it -- every thread of the cgroup is then
// throttled for the rest of the scheduler period.
int qil_cpu_budget() {
    static const int budget = []() {
        if (const char* e = getenv("QIL_CPU_BUDGET")) return std::max(1, atoi(e));
        double cpus = (double)std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = std::min(cpus, (double)std::max(1, CPU_COUNT(&set)));
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64] = {0};
            long long period = 0;
            if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) cpus = std::min(cpus, (double)atoll(q) / (double)period);
            fclose(f);
        } else {
            long long quota = -1, period = 0;
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                if (fscanf(g, "%lld", &quota) != 1) quota = -1;
                fclose(g);
            }
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(g, "%lld", &period) != 1) period = 0;
                fclose(g);
            }
            if (quota > 0 && period > 0) cpus = std::min(cpus, (double)quota / (double)period);
        }
        int ranks = 1;
        if (const char* e = getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1, atoi(e));
        return std::max(1, (int)(cpus / ranks));
    }();
    return budget;
}
extern "C" int qil_host_cpu_budget(int* out) {
    QIL_REQUIRE(out, QIL_EINVAL_ARG, "null out");
    *out = qil_cpu_budget();
    return QIL_OK;
}

int qil_run_batch_on(qil_context* home, int64_t nb, const std::function<void(int64_t, qil_context*)>& place,
                     const std::function<int(int64_t, qil_context*)>& fn) {
    if (nb <= 0) return QIL_OK;
    QIL_REQUIRE(home, QIL_EINVAL_ARG, "batch: null context");
    // One chain per hardware queue: the runtime multiplexes streams onto 4 queues, and streams that share one serialise
    // (measured with rocprofv3 --kernel-trace: a fifth stream lands on an occupied queue and its chain runs at half speed).
    // The calling thread drives the home stream itself, so nw chains use nw streams.
    constexpr int max_workers = 8;                              // (8 streams on the 4 queues: 2.09x single for 8 chains, 4 streams: 2.22x)
    // a context that is already one slot of a running batch (the home context during its own batch, or a worker) runs
    // nested batches inline: the slots are taken, and the batch mutex is held by the outer call
    // Lock-step groups (qil_launch.h): up to QIL_MAXB chains share ONE stream and a launcher that issues the same step of all of
    // them as one table launch; up to 4 such groups run side by side on streams of their own (one per hardware queue).  One
    // group is a serial stream -- measured, compress! chi 256 -> 128: 8 chains 146-160 ms as one group against 136 ms on 8
    // streams, 4 chains 109 against 75 ms -- so small batches keep the stream-per-chain form and lock-step takes over where
    // the 4 hardware queues are the limit: from 5 chains on.
    const bool lockstep = nb >= 5;
    constexpr int kMaxGroups = 4;
    // all four queues, the chains dealt over them; slot k belongs to group k % ng -- but never more polling launchers than the
    // process's CPU budget leaves room for next to one chain thread that is awake (qil_cpu_budget), and never more than QIL_MAXB
    // chains per group (a table launch carries at most that many operands; the launcher's scan arrays are that long)
    const int ng_cap = lockstep ? std::max(1, std::min(kMaxGroups, qil_cpu_budget() - 1)) : 0;
    const int nw = (home->lending || home->parent) ? 1
                   : (int)std::min<int64_t>(nb, lockstep ? QIL_MAXB * ng_cap : std::max(1, max_workers));
    const int ng = lockstep ? std::min(ng_cap, nw) : 0;
    if (nw <= 1) {
        int first = QIL_OK;
        std::string msg;
        for (int64_t j = 0; j < nb; ++j) {
            const int s = fn(j, home);
            if (s != QIL_OK && first == QIL_OK) {
                first = s;
                msg = qil_last_error();
                msg += " (item " + std::to_string(j) + " of the batch)";
            }
        }
        return first == QIL_OK ? QIL_OK : qil_fail(first, "%s", msg.c_str());
    }
    QIL_TRY(qil_ctx_activate(home));
    std::lock_guard<std::mutex> lock(home->batch_mutex);
    while ((int)home->workers.size() < nw - 1) {
        qil_context* w = nullptr;
        QIL_TRY(qil_context_create(home->device, nullptr, &w));
        w->parent = home;
        home->workers.push_back(w);
    }
    auto slot_ctx = [&](int k) { return k == 0 ? home : home->workers[(size_t)k - 1]; };
    hipEvent_t ready = nullptr;                               // everything enqueued on the home stream so far
    QIL_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    hipError_t he = hipEventRecord(ready, home->stream);
    if (he != hipSuccess) {
        (void)hipEventDestroy(ready);
        return qil_fail(QIL_EHIP, "hipEventRecord failed: %s", hipGetErrorString(he));
    }
    if (place)
        for (int64_t j = 0; j < nb; ++j)
            if (j % nw) place(j, slot_ctx((int)(j % nw)));
    std::vector<qil_lockstep> lsg((size_t)std::max(ng, 1));
    std::unique_ptr<qil_chainq[]> rings;
    std::vector<hipStream_t> own_stream((size_t)nw, nullptr);
    if (lockstep) {
        rings.reset(new qil_chainq[(size_t)nw]);
        for (int k = 0; k < nw; ++k) own_stream[(size_t)k] = slot_ctx(k)->stream;
        int off = 0;
        for (int g = 0; g < ng; ++g) {
            qil_lockstep& ls = lsg[(size_t)g];
            ls.q = rings.get() + off;
            ls.nslots = (nw - g + ng - 1) / ng;                  // slots g, g + ng, g + 2 ng, ...
            off += ls.nslots;
            ls.stream = own_stream[(size_t)g];                   // group 0: the home stream; group g: its first slot's own stream
            if (g && hipStreamWaitEvent(ls.stream, ready, 0) != hipSuccess) {
                (void)hipEventDestroy(ready);
                for (int k = 1; k < nw; ++k) {                    // place() has moved chains to the workers: hand them back
                    const std::vector<qil_chain*> held(slot_ctx(k)->chains.begin(), slot_ctx(k)->chains.end());
                    for (qil_chain* c : held) chain_move(c, home);
                }
                return qil_fail(QIL_EHIP, "hipStreamWaitEvent failed");
            }
        }
        for (int k = 0; k < nw; ++k) {
            qil_context* w = slot_ctx(k);
            qil_lockstep& ls = lsg[(size_t)(k % ng)];
            w->stream = ls.stream;
            w->ls_slot = k / ng;
            w->progress_key = 0;
            w->lockstep = &ls;
        }
    }
    home->lend_blocks.swap(home->free_blocks);               // lend the cache (free_blocks is now empty)
    home->lend_cached = home->bytes_cached;
    home->bytes_cached = 0;
    home->lending = true;
    std::vector<int> status((size_t)nb, QIL_OK);
    std::vector<std::string> message((size_t)nb);
    const bool batch_debug = getenv("QIL_BATCH_DEBUG") != nullptr;
    const auto t_batch = std::chrono::steady_clock::now();
    auto drive = [&](int k) {
        qil_context* w = slot_ctx(k);
        const bool in_step = lockstep && w->lockstep;            // (a slot whose thread could not be started runs alone afterwards)
        w->dbg_times = batch_debug;
        int s0 = QIL_OK;
        if ((k || lockstep) && (hipSetDevice(w->device) != hipSuccess || (!in_step && w != home && hipStreamWaitEvent(w->stream, ready, 0) != hipSuccess))) s0 = QIL_EHIP;
        for (int64_t j = k; j < nb; j += nw) {
            const auto tj0 = std::chrono::steady_clock::now();
            const int s = s0 != QIL_OK ? s0 : fn(j, w);
            if (batch_debug) {
                fprintf(stderr, "[batch] slot %d item %lld: start %.2f ms, took %.2f ms; %lld read-backs waited %.2f ms, ring full %.2f ms, "
                        "%lld allocations (%lld beyond the caches) %.2f ms\n", k, (long long)j,
                        std::chrono::duration<double, std::milli>(tj0 - t_batch).count(),
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tj0).count(), w->dbg_rb_n,
                        w->dbg_rb_us / 1e3, w->dbg_ring_us / 1e3, w->dbg_alloc_n, w->dbg_alloc_miss, w->dbg_alloc_us / 1e3);
                w->dbg_rb_n = w->dbg_alloc_n = w->dbg_alloc_miss = 0;
                w->dbg_rb_us = w->dbg_ring_us = w->dbg_alloc_us = 0;
            }
            if (s != QIL_OK) {
                status[(size_t)j] = s;
                message[(size_t)j] = s0 != QIL_OK ? "worker stream setup failed" : qil_last_error();
            }
        }
        if (in_step) {
            qil_lockstep_drain(w);
            w->lockstep->q[w->ls_slot].live.store(0, std::memory_order_release);    // the launcher no longer waits for this slot
        } else {
            (void)hipStreamSynchronize(w->stream);
        }
    };
    std::vector<std::thread> threads;
    threads.reserve((size_t)nw - 1);
    std::vector<int> inline_slots;                            // slots whose thread could not be started run here afterwards
    auto demote = [&](int k) {                                // slot k leaves its lock-step group: its own stream, plain launches
        qil_context* w = slot_ctx(k);
        w->lockstep->q[w->ls_slot].live.store(0, std::memory_order_release);
        w->lockstep = nullptr;
        w->stream = own_stream[(size_t)k];
    };
    // lock-step: one launcher per group, this thread serves group 0.  The launchers start BEFORE any chain thread:
    // a group whose launcher thread cannot be started (thread exhaustion) is dissolved here, while nothing has been queued --
    // its chains run on their own streams with plain launches instead of filling rings nobody serves until group 0 is done
    // (spurious QIL_EHIP after qil_ls_park's 60 s).
    std::vector<std::thread> launchers;
    if (lockstep)
        for (int g = 1; g < ng; ++g) {
            try {
                launchers.emplace_back([&, g]() {
                    (void)hipSetDevice(home->device);
                    lockstep_run(&lsg[(size_t)g], batch_debug);
                });
            } catch (const std::system_error&) {
                for (int k = g; k < nw; k += ng) demote(k);
            }
        }
    // every slot, the home one included, runs on a thread of its own in the lock-step form
    for (int k = lockstep ? 0 : 1; k < nw; ++k) {
        try {
            threads.emplace_back(drive, k);
        } catch (const std::system_error&) {
            inline_slots.push_back(k);
            if (lockstep && slot_ctx(k)->lockstep) demote(k);
        }
    }
    if (lockstep) {
        lockstep_run(&lsg[0], batch_debug);
        for (auto& t : launchers) t.join();
    } else {
        drive(0);
    }
    for (int k : inline_slots) drive(k);
    for (auto& t : threads) t.join();
    if (lockstep) {
        for (int g = 0; g < ng; ++g) (void)hipStreamSynchronize(lsg[(size_t)g].stream);
        for (int k = 0; k < nw; ++k) {
            qil_context* w = slot_ctx(k);
            w->stream = own_stream[(size_t)k];
            w->lockstep = nullptr;
        }
        if (batch_debug)
            for (int g = 0; g < ng; ++g) {
                const qil_lockstep& ls = lsg[(size_t)g];
                fprintf(stderr, "[batch] lock-step group %d (%d chains): %lld requests in %lld launches (%lld after waiting for a chain); "
                        "launches by group size:", g, ls.nslots, ls.requests, ls.launches, ls.timeouts);
                for (int c = 1; c <= QIL_MAXB; ++c)
                    if (ls.group_hist[c]) fprintf(stderr, " %d:%lld", c, ls.group_hist[c]);
                fprintf(stderr, "; launcher %.1f ms, %.1f ms of it inside launch calls\n", ls.total_us / 1e3, ls.launch_us / 1e3);
            }
    }
    home->lending = false;
    // every stream of the batch is idle: the chains the workers hold (moved there or created there), what is left of the
    // lent blocks and the workers' caches go (back) to the home context
    for (int k = 1; k < nw; ++k) {
        qil_context* w = slot_ctx(k);
        const std::vector<qil_chain*> held(w->chains.begin(), w->chains.end());
        for (qil_chain* c : held) chain_move(c, home);
    }
    for (auto& kv : home->lend_blocks) home->free_blocks.emplace(kv.first, kv.second);
    home->bytes_cached += home->lend_cached;
    home->lend_blocks.clear();
    home->lend_cached = 0;
    // (the workers KEEP their caches between batches: a slot runs the same chain shapes every time, and a worker that starts
    // with an empty cache takes every block from the lender under its mutex -- 32 chain threads: 1 600 allocations of 32 us
    // each per chain, on the critical path of every lock-step step; qil_context_mem_info / qil_context_trim cover the workers)
    (void)hipEventDestroy(ready);
    for (int64_t j = 0; j < nb; ++j)
        if (status[(size_t)j] != QIL_OK)
            return qil_fail(status[(size_t)j], "%s (item %lld of the batch)", message[(size_t)j].c_str(), (long long)j);
    return QIL_OK;
}

int qil_run_batch(qil_chain* const* items, int64_t nb, const std::function<int(qil_chain*)>& fn) {
    if (nb <= 0) return QIL_OK;
    QIL_REQUIRE(items, QIL_EINVAL_ARG, "batch: null item array");
    std::set<const qil_chain*> seen;
    for (int64_t j = 0; j < nb; ++j) {
        QIL_REQUIRE(items[j] && items[j]->ctx, QIL_EINVAL_ARG, "batch: item %lld is null or has no context", (long long)j);
        QIL_REQUIRE(items[j]->ctx == items[0]->ctx, QIL_EINVAL_ARG, "batch: item %lld lives in another context", (long long)j);
        QIL_REQUIRE(seen.insert(items[j]).second, QIL_EINVAL_ARG, "batch: item %lld appears twice", (long long)j);
    }
    return qil_run_batch_on(
        items[0]->ctx, nb, [&](int64_t j, qil_context* slot) { chain_move(items[j], slot); },
        [&](int64_t j, qil_context*) { return fn(items[j]); });
}

void qil_chain_bind(qil_chain* c, qil_context* ctx) {
    c->ctx = ctx;
    if (ctx) ctx->chains.insert(c);
}

int qil_chain_release(qil_chain* c) {
    if (!c || !c->ctx) return QIL_OK;
    for (void*& p : c->site) {
        if (p) qil_ctx_free(c->ctx, p);
        p = nullptr;
    }
    return QIL_OK;
}

int qil_chain_set_site(qil_chain* c, int64_t i, void* p, int64_t dl, int64_t dr) {
    if (c->site[(size_t)i]) QIL_TRY(qil_ctx_free(c->ctx, c->site[(size_t)i]));
    c->site[(size_t)i] = p;
    mark_owned(c->ctx, p);
    c->dims[(size_t)i] = dl;
    c->dims[(size_t)i + 1] = dr;
    return QIL_OK;
}

void qil_chain_adopt(qil_chain* c, int64_t i, void* p) {
    c->site[(size_t)i] = p;
    mark_owned(c->ctx, p);
}

template <class H>
static int chain_create(qil_context* ctx, int64_t n, int dtype, int paired, int phys_rank,
                        const int64_t* bond_dims, const int64_t* site_ids, const void* const* site_ptrs,
                        H** out) {
    QIL_REQUIRE(out, QIL_EINVAL_ARG, "null out");
    H* h = new H();
    int s = qil_chain_alloc(ctx, h, n, dtype, paired, phys_rank, bond_dims, site_ids);
    if (s != QIL_OK) {
        delete h;
        return s;
    }
    if (site_ptrs) {
        for (int64_t i = 0; i < n; ++i) {
            if (!site_ptrs[i]) {
                qil_chain_release(h);
                delete h;
                return qil_fail(QIL_EINVAL_ARG, "site %lld: null host pointer", (long long)(i + 1));
            }
            hipError_t e = hipMemcpyAsync(h->site[(size_t)i], site_ptrs[i], h->site_bytes(i),
                                          hipMemcpyHostToDevice, qil_stream(ctx));
            if (e != hipSuccess) {
                qil_chain_release(h);
                delete h;
                return qil_fail(QIL_EHIP, "upload of site %lld failed: %s", (long long)(i + 1),
                                hipGetErrorString(e));
            }
        }
        // host buffers may be pageable and are not retained: finish the copies before returning
        hipError_t e = qil_stream_sync(ctx);
        if (e != hipSuccess) {
            qil_chain_release(h);
            delete h;
            return qil_fail(QIL_EHIP, "upload failed: %s", hipGetErrorString(e));
        }
    }
    *out = h;
    return QIL_OK;
}

extern "C" int qil_mps_create(qil_context* ctx, int64_t n, int dtype, int paired, const int64_t* bond_dims,
                              const int64_t* site_ids, const void* const* site_ptrs, double amplitude,
                              qil_mps** out) {
    QIL_REQUIRE(site_ptrs, QIL_EINVAL_ARG, "qil_mps_create: null site_ptrs");
    QIL_TRY(chain_create<qil_mps>(ctx, n, dtype, paired, 1, bond_dims, site_ids, site_ptrs, out));
    (*out)->amplitude = amplitude;
    return QIL_OK;
}

extern "C" int qil_mps_alloc(qil_context* ctx, int64_t n, int dtype, int paired, const int64_t* bond_dims,
                             const int64_t* site_ids, double amplitude, qil_mps** out) {
    QIL_TRY(chain_create<qil_mps>(ctx, n, dtype, paired, 1, bond_dims, site_ids, nullptr, out));
    (*out)->amplitude = amplitude;
    return QIL_OK;
}

extern "C" int qil_mpo_create(qil_context* ctx, int64_t n, int dtype, int paired, const int64_t* bond_dims,
                              const int64_t* site_ids, const void* const* site_ptrs, qil_mpo** out) {
    QIL_REQUIRE(site_ptrs, QIL_EINVAL_ARG, "qil_mpo_create: null site_ptrs");
    return chain_create<qil_mpo>(ctx, n, dtype, paired, 2, bond_dims, site_ids, site_ptrs, out);
}

extern "C" int qil_mpo_alloc(qil_context* ctx, int64_t n, int dtype, int paired, const int64_t* bond_dims,
                             const int64_t* site_ids, qil_mpo** out) {
    return chain_create<qil_mpo>(ctx, n, dtype, paired, 2, bond_dims, site_ids, nullptr, out);
}

static int chain_destroy(qil_chain* c) {
    if (!c) return QIL_OK;
    if (c->ctx) hipSetDevice(c->ctx->device);
    qil_chain_release(c);
    return QIL_OK;
}

extern "C" int qil_mps_destroy(qil_mps* psi) {
    chain_destroy(psi);
    delete psi;
    return QIL_OK;
}
extern "C" int qil_mpo_destroy(qil_mpo* W) {
    chain_destroy(W);
    delete W;
    return QIL_OK;
}

int qil_mps_clone_to(qil_context* ctx, const qil_mps* psi, qil_mps** out) {
    QIL_TRY(chain_create<qil_mps>(ctx, psi->n(), psi->dtype, psi->paired, 1, psi->dims.data() + 1,
                                  psi->site_ids.data(), nullptr, out));
    (*out)->amplitude = psi->amplitude;
    for (int64_t i = 0; i < psi->n(); ++i)
        QIL_HIP(hipMemcpyAsync((*out)->site[(size_t)i], psi->site[(size_t)i], psi->site_bytes(i),
                               hipMemcpyDeviceToDevice, qil_stream(ctx)));
    return QIL_OK;
}

extern "C" int qil_mps_clone(const qil_mps* psi, qil_mps** out) {
    QIL_REQUIRE(psi && out, QIL_EINVAL_ARG, "qil_mps_clone: null argument");
    return qil_mps_clone_to(psi->ctx, psi, out);
}

#define CHAIN_GETTERS(PFX, TYPE)                                                                        \
    extern "C" int PFX##_nsites(const TYPE* c, int64_t* n) {                                            \
        QIL_REQUIRE(c && n, QIL_EINVAL_ARG, #PFX "_nsites: null argument");                             \
        *n = c->n();                                                                                    \
        return QIL_OK;                                                                                  \
    }                                                                                                   \
    extern "C" int PFX##_dtype(const TYPE* c, int* d) {                                                 \
        QIL_REQUIRE(c && d, QIL_EINVAL_ARG, #PFX "_dtype: null argument");                              \
        *d = c->dtype;                                                                                  \
        return QIL_OK;                                                                                  \
    }                                                                                                   \
    extern "C" int PFX##_is_paired(const TYPE* c, int* p) {                                             \
        QIL_REQUIRE(c && p, QIL_EINVAL_ARG, #PFX "_is_paired: null argument");                          \
        *p = c->paired;                                                                                 \
        return QIL_OK;                                                                                  \
    }                                                                                                   \
    extern "C" int PFX##_bond_dims(const TYPE* c, int64_t* b) {                                         \
        QIL_REQUIRE(c && (b || c->n() == 1), QIL_EINVAL_ARG, #PFX "_bond_dims: null argument");         \
        for (int64_t i = 0; i + 1 < c->n(); ++i) b[i] = c->dims[(size_t)i + 1];                         \
        return QIL_OK;                                                                                  \
    }                                                                                                   \
    extern "C" int PFX##_site_ids(const TYPE* c, int64_t* s) {                                          \
        QIL_REQUIRE(c && s, QIL_EINVAL_ARG, #PFX "_site_ids: null argument");                           \
        for (int64_t i = 0; i < c->n(); ++i) s[i] = c->site_ids[(size_t)i];                             \
        return QIL_OK;                                                                                  \
    }                                                                                                   \
    extern "C" int PFX##_site_nbytes(const TYPE* c, int64_t i, int64_t* nb) {                           \
        QIL_REQUIRE(c && nb, QIL_EINVAL_ARG, #PFX "_site_nbytes: null argument");                       \
        QIL_REQUIRE(i >= 0 && i < c->n(), QIL_EINVAL_ARG, "site index %lld out of range", (long long)i); \
        *nb = (int64_t)c->site_bytes(i);                                                                \
        return QIL_OK;                                                                                  \
    }                                                                                                   \
    extern "C" int PFX##_download_site(const TYPE* c, int64_t i, void* dst) {                           \
        QIL_REQUIRE(c && dst, QIL_EINVAL_ARG, #PFX "_download_site: null argument");                    \
        QIL_REQUIRE(i >= 0 && i < c->n(), QIL_EINVAL_ARG, "site index %lld out of range", (long long)i); \
        QIL_TRY(qil_ctx_activate(c->ctx));                                                              \
        QIL_HIP(hipMemcpyAsync(dst, c->site[(size_t)i], c->site_bytes(i), hipMemcpyDeviceToHost,        \
                               qil_stream(c->ctx)));                                                        \
        QIL_HIP(qil_stream_sync(c->ctx));                                                  \
        return QIL_OK;                                                                                  \
    }                                                                                                   \
    extern "C" int PFX##_site_device_ptr(const TYPE* c, int64_t i, void** p) {                          \
        QIL_REQUIRE(c && p, QIL_EINVAL_ARG, #PFX "_site_device_ptr: null argument");                    \
        QIL_REQUIRE(i >= 0 && i < c->n(), QIL_EINVAL_ARG, "site index %lld out of range", (long long)i); \
        *p = c->site[(size_t)i];                                                                        \
        return QIL_OK;                                                                                  \
    }

CHAIN_GETTERS(qil_mps, qil_mps)
CHAIN_GETTERS(qil_mpo, qil_mpo)

extern "C" int qil_mps_upload_site(qil_mps* c, int64_t i, const void* src) {
    QIL_REQUIRE(c && src, QIL_EINVAL_ARG, "qil_mps_upload_site: null argument");
    QIL_REQUIRE(i >= 0 && i < c->n(), QIL_EINVAL_ARG, "site index %lld out of range", (long long)i);
    QIL_TRY(qil_ctx_activate(c->ctx));
    QIL_HIP(hipMemcpyAsync(c->site[(size_t)i], src, c->site_bytes(i), hipMemcpyHostToDevice, qil_stream(c->ctx)));
    QIL_HIP(qil_stream_sync(c->ctx));
    return QIL_OK;
}

extern "C" int qil_mps_amplitude(const qil_mps* psi, double* a) {
    QIL_REQUIRE(psi && a, QIL_EINVAL_ARG, "qil_mps_amplitude: null argument");
    *a = psi->amplitude;
    return QIL_OK;
}
extern "C" int qil_mps_set_amplitude(qil_mps* psi, double a) {
    QIL_REQUIRE(psi, QIL_EINVAL_ARG, "qil_mps_set_amplitude: null argument");
    psi->amplitude = a;
    return QIL_OK;
}

static int chain_fill_random(qil_chain* c, uint64_t seed) {
    QIL_REQUIRE(c, QIL_EINVAL_ARG, "fill_random: null handle");
    QIL_TRY(qil_ctx_activate(c->ctx));
    for (int64_t i = 0; i < c->n(); ++i) {
        double scale = 1.0 / sqrt(2.0 * (double)c->dims[(size_t)i]);
        QIL_TRY(qil_dev_fill_normal(c->ctx, c->dtype, c->site[(size_t)i], c->site_elems(i),
                                    seed * 0x9E3779B97F4A7C15ull + (uint64_t)i, scale));
    }
    return QIL_OK;
}
extern "C" int qil_mps_fill_random(qil_mps* psi, uint64_t seed) { return chain_fill_random(psi, seed); }
extern "C" int qil_mpo_fill_random(qil_mpo* W, uint64_t seed) { return chain_fill_random(W, seed); }
