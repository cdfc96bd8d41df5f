// Internal declarations shared by the libqilhip.so translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_complex.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <condition_variable>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "qilaplace_hip.h"
#include "qilaplace_hip_testing.h"

#include <atomic>

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE (a `static bool` granted it on the first GPU only,
// and a context on a second GPU of the same process then failed its launch).  One of these per kernel: the bytes granted so
// far on each device; concurrent callers may both set the attribute, which is harmless.
struct qil_lds_grant {
    static constexpr int kDevices = 64;
    std::atomic<size_t> bytes[kDevices] = {};
    hipError_t ensure(int device, const void* fn, size_t want) {
        const int d = device >= 0 && device < kDevices ? device : -1;
        if (d >= 0 && want <= bytes[d].load(std::memory_order_acquire)) return hipSuccess;
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
        if (e == hipSuccess && d >= 0) {
            size_t cur = bytes[d].load(std::memory_order_relaxed);
            while (cur < want && !bytes[d].compare_exchange_weak(cur, want, std::memory_order_release)) {}
        }
        return e;
    }
};

// ---------------------------------------------------------------- errors
void qil_set_error(const char* fmt, ...);
int qil_fail(int code, const char* fmt, ...);

#define QIL_HIP(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return qil_fail(_e == hipErrorOutOfMemory ? QIL_ENOMEM : QIL_EHIP,             \
                            "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),         \
                            __FILE__, __LINE__);                                           \
    } while (0)

#define QIL_TRY(expr)                 \
    do {                              \
        int _s = (expr);              \
        if (_s != QIL_OK) return _s;  \
    } while (0)

#define QIL_REQUIRE(cond, code, ...)                       \
    do {                                                   \
        if (!(cond)) return qil_fail((code), __VA_ARGS__); \
    } while (0)

// ---------------------------------------------------------------- context
struct qil_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    // caching pool: exact-size free lists (apply outputs recur with identical sizes)
    std::multimap<size_t, void*> free_blocks;
    struct live_block {
        size_t bytes;
        uint64_t serial;   // allocation order (error-path reclamation, see qil_call_scope)
        bool owned;        // attached to a chain handle: outlives the call that allocated it
    };
    std::map<void*, live_block> live_blocks;
    uint64_t alloc_serial = 0;
    int64_t fail_alloc_countdown = -1;   // fault injection (qil_context_fail_alloc_after); < 0 = off
    size_t bytes_in_use = 0, bytes_cached = 0;
    // pinned staging for small descriptor / bit uploads
    void* pinned = nullptr;
    size_t pinned_bytes = 0;
    struct qil_lockstep* lockstep = nullptr;   // set while this context is a slot of a running lock-step batch (qil_launch.h)
    int ls_slot = 0;              // ... and which one
    hipEvent_t sync_event = nullptr;   // qil_stream_sync inside a lock-step batch
    uint64_t progress_key = 0;    // where the chain driven through this context is (qil_progress)
    int cholqr_skip = 0;          // Cholesky QR attempts to skip after a refusal (qr_impl)
    bool qr_orthonormal = false;  // the last qr_impl took CholeskyQR2 with its first-order second pass: Q^H Q = I to O(|E|^2) < 1e-15 by construction (qr_reorthogonalise)
    double svd_deflate = 0.0;     // weight (relative) the one-factor SVD may drop with negligible rows of R: set by the truncating caller
    // CholeskyQR2 has R^-1 = X1 X2 at hand: a caller that is about to run the truncation certificate on R asks for it
    // (want_rinv) and takes the block over (rinv, valid for the R at rinv_for; the caller frees it)
    bool want_rinv = false;
    void* rinv = nullptr;
    uint64_t rinv_serial = 0;            // allocation serial of the parked block (the call scope frees it only if it still matches)
    const void* rinv_for = nullptr;
    // small device -> host read-backs without a copy command or a stream synchronisation (qil_read_back): kRbSlots slots of
    // kRbSlotBytes in pinned, device-visible memory + a ticket word a kernel writes behind the data; the host polls the word
    static constexpr int kRbSlots = 4;
    static constexpr size_t kRbSlotBytes = 8192;
    void* rb_host = nullptr;
    uint64_t rb_ticket = 0, rb_done = 0;                          // posted / seen complete
    bool dbg_times = false;       // QIL_BATCH_DEBUG: where a chain's host thread spends its time
    double dbg_rb_us = 0, dbg_ring_us = 0, dbg_alloc_us = 0;
    long long dbg_rb_n = 0, dbg_alloc_n = 0, dbg_alloc_miss = 0;
    // small host -> device uploads of a chain (permutations, scale vectors) without a copy command or an event: filled in a
    // pinned slot, moved by a (combinable) copy kernel; a slot is reused once a read-back posted after its consumers has
    // completed (qil_stage_*)
    static constexpr int kStSlots = 16;
    static constexpr size_t kStSlotBytes = 32768;
    void* st_host = nullptr;
    void* st_dev = nullptr;
    uint64_t st_born[kStSlots] = {};
    bool st_used[kStSlots] = {};
    int st_next = 0;
    void* dev_scratch = nullptr;  // per-call device workspace (stream-ordered reuse)
    size_t dev_scratch_bytes = 0;
    // ring of small host(pinned)/device descriptor slots for grouped launches: a slot is reused
    // only after the event recorded behind its last consumer has completed
    static constexpr int kDescSlots = 8;
    static constexpr size_t kDescSlotBytes = 1 << 16;
    void* desc_host = nullptr;
    void* desc_dev = nullptr;
    hipEvent_t desc_event[kDescSlots] = {};
    bool desc_used[kDescSlots] = {};
    int desc_next = 0;
    // timers
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    std::vector<hipEvent_t> event_pool;
    int num_cus = 256;
    // live chain handles: orphaned (ctx = nullptr, site pointers dropped) when the context is destroyed first,
    // so a handle released after its context touches nothing
    std::set<struct qil_chain*> chains;
    // communicators created on this context (qil_comm.hip): torn down with it, their handles stay valid and empty
    std::set<struct qil_comm*> comms;
    // Batch entry points (qil_run_batch): independent chains run concurrently on this context's own stream (calling
    // thread) and on worker contexts (own stream and pool, one host thread each), one batch at a time.  For the duration of
    // a batch the home context's cached blocks are LENT: every participant misses in its own cache first, then takes from
    // `lend_blocks` (all work enqueued on them is ordered before the batch by an event), then allocates; at the end the
    // remaining lent blocks return to the home cache and the workers keep theirs, so steady-state batches allocate nothing and
    // take (almost) nothing from the lender.
    std::vector<qil_context*> workers;
    std::mutex batch_mutex;
    qil_context* parent = nullptr;                   // worker -> home
    bool lending = false;                            // home: a batch is running
    std::multimap<size_t, void*> lend_blocks;        // guarded by pool_mutex
    size_t lend_cached = 0;                          // bytes in lend_blocks (guarded by pool_mutex)
    std::mutex pool_mutex;
};

// The stream for anything that is NOT a qil_klaunch: inside a lock-step batch the chain first waits until the launcher has
// issued everything it has queued (stream order), otherwise this is ctx->stream.
void qil_lockstep_drain(qil_context* ctx);
inline hipStream_t qil_stream(qil_context* ctx) {
    if (ctx->lockstep) qil_lockstep_drain(ctx);
    return ctx->stream;
}
// hipStreamSynchronize for chain code: inside a lock-step batch the slots share one stream that the launcher keeps feeding, so
// the chain waits for an event recorded behind ITS last operation instead of for the whole stream to run dry
hipError_t qil_stream_sync(qil_context* ctx);
hipError_t qil_event_sync(qil_context* ctx, hipEvent_t ev);    // hipEventSynchronize that tells the batch's launcher the chain is parked
int qil_ctx_activate(qil_context* ctx);  // hipSetDevice
int qil_ctx_alloc(qil_context* ctx, size_t bytes, void** out);
int qil_ctx_free(qil_context* ctx, void* p);
int qil_ctx_pinned(qil_context* ctx, size_t bytes, void** out);       // grows, stream-synchronising
int qil_ctx_dev_scratch(qil_context* ctx, size_t bytes, void** out);  // grows
// acquire a descriptor slot (host staging + device copy target); call qil_ctx_desc_commit after
// enqueueing the last kernel that reads the device side
int qil_ctx_desc_acquire(qil_context* ctx, size_t bytes, void** host, void** dev, int* slot);
int qil_ctx_desc_commit(qil_context* ctx, int slot);
int qil_ctx_event(qil_context* ctx, hipEvent_t* e);            // from the context's event pool
void qil_ctx_event_release(qil_context* ctx, hipEvent_t e);
// `bytes` (a multiple of 4, 4-byte aligned source) of device memory to the host, ordered after everything this context has
// launched: qil_read_back = post + wait; posted read-backs complete in order, at most kRbSlots - 1 may be outstanding.
// Larger blocks than a slot take the copy-command + stream-synchronisation route.
int qil_apply_shared_state(const qil_mpo* W, const qil_mps* psi, qil_mps** out);   // (qil_apply.hip) psi may live in another context of the device
int qil_lockstep_park(qil_context* ctx, const unsigned long long* word, unsigned long long ticket);   // (qil_context.hip)
int qil_read_back_post(qil_context* ctx, const void* dev_src, size_t bytes, uint64_t* ticket);
int qil_read_back_wait(qil_context* ctx, uint64_t ticket, void* host_dst, size_t bytes);
int qil_read_back(qil_context* ctx, void* host_dst, const void* dev_src, size_t bytes);
// acquire a slot (host = pinned side to fill, dev = what the kernels read), push = launch the copy, commit = the consumers
// have been launched
int qil_stage_acquire(qil_context* ctx, size_t bytes, void** host, void** dev, int* slot);
int qil_stage_push(qil_context* ctx, int slot, size_t bytes);
void qil_stage_commit(qil_context* ctx, int slot);
int qil_ctx_prof_begin(qil_context* ctx);
int qil_ctx_prof_end(qil_context* ctx);

// Batches of independent chains.  A chain of truncations is a latency chain of small dependent kernels that occupies a
// few percent of the chip; `fn` is run for every chain of the batch concurrently: chain j is handed (its pool blocks
// change owner, nothing is copied) to worker context j % nw of the chains' common home context, one host thread per worker
// drives it on that worker's stream, and the chains return to the home context before the call does.  The work of
// each chain is exactly what fn(chain) does alone.  All chains must belong to the same context; nw = min(nb,
// QIL_BATCH_WORKERS (default 8)).  Returns the first failing chain's status.
int qil_run_batch(struct qil_chain* const* items, int64_t nb, const std::function<int(struct qil_chain*)>& fn);
// The same for work that CREATES chains from read-only operands of the home context: fn(j, work) runs item j with `work`
// as its working context (stream + pool; operands may be shared between items and stay where they are); every chain
// bound to a worker when the batch ends -- moved there by place(j, slot) beforehand or created by fn -- returns to home.
int qil_run_batch_on(qil_context* home, int64_t nb, const std::function<void(int64_t, qil_context*)>& place,
                     const std::function<int(int64_t, qil_context*)>& fn);
void qil_comm_orphan(struct qil_comm* cm);   // (qil_comm.hip) the context of a communicator is going away
// hand a live pool block of `from` to `to` (bookkeeping only; the caller orders the streams)
void qil_ctx_transfer(qil_context* from, qil_context* to, void* p);
// hand a whole chain (its site blocks and its registration) to another context of the same device
void qil_chain_rebind(struct qil_chain* c, qil_context* to);
// device copy of psi owned by ctx (made on ctx's stream)
int qil_mps_clone_to(qil_context* ctx, const struct qil_mps* psi, struct qil_mps** out);

// ---------------------------------------------------------------- containers
static inline size_t qil_elem_size(int dtype) { return dtype == QIL_C64 ? 16 : 8; }

struct qil_chain {
    qil_context* ctx = nullptr;
    int dtype = QIL_F64;
    int paired = 0;
    int phys_rank = 1;               // 1 = MPS (A[a,s,b]), 2 = MPO (W[a,si,so,b])
    std::vector<int64_t> dims;       // n+1 bond dims incl. the two dim-1 edges
    std::vector<int64_t> site_ids;   // n labels
    std::vector<void*> site;         // device pointers, one allocation per site
    double amplitude = 1.0;
    qil_chain() = default;
    qil_chain(const qil_chain&) = delete;
    qil_chain& operator=(const qil_chain&) = delete;
    ~qil_chain() {
        if (ctx) ctx->chains.erase(this);
    }
    int64_t n() const { return (int64_t)site.size(); }
    int64_t site_elems(int64_t i) const {
        return dims[i] * (phys_rank == 1 ? 2 : 4) * dims[i + 1];
    }
    size_t site_bytes(int64_t i) const { return (size_t)site_elems(i) * qil_elem_size(dtype); }
};
struct qil_mps : qil_chain {};
struct qil_mpo : qil_chain {};

int qil_chain_alloc(qil_context* ctx, qil_chain* c, int64_t n, int dtype, int paired, int phys_rank,
                    const int64_t* bond_dims, const int64_t* site_ids);
int qil_chain_release(qil_chain* c);
// tie a handle to its context (registers it for orphaning on context destruction)
void qil_chain_bind(qil_chain* c, qil_context* ctx);
// replace site i's buffer (takes ownership of `p`), updating the bond dims
int qil_chain_set_site(qil_chain* c, int64_t i, void* p, int64_t dl, int64_t dr);
// attach a pool block to an EMPTY site slot of a chain under construction (ownership moves to the chain)
void qil_chain_adopt(qil_chain* c, int64_t i, void* p);

// Error-path reclamation.  Every C entry point opens one of these; temporaries come from the context pool and
// are released explicitly on the success path.  If the call FAILS (qil_fail ran on this thread while the scope
// was open), the destructor returns to the pool every block that was allocated during the call and is not
// owned by a chain handle -- so an early `return status` can never strand device memory.
unsigned qil_fail_count();
struct qil_call_scope {
    qil_context* ctx;
    uint64_t serial0;
    unsigned fails0;
    explicit qil_call_scope(qil_context* c);
    ~qil_call_scope();
    qil_call_scope(const qil_call_scope&) = delete;
    qil_call_scope& operator=(const qil_call_scope&) = delete;
};

// ---------------------------------------------------------------- device linear algebra (qil_linalg.hip)
// All matrices column-major on the device, dtype QIL_F64/QIL_C64.
// C[m x n] = opA(A) * opB(B); op: 0 = N, 1 = T, 2 = H, 3 = conj (no transpose).  alpha = 1, beta = 0.
int qil_dev_gemm(qil_context* ctx, int dtype, int opA, int opB, int64_t m, int64_t n, int64_t k,
                 const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc);
// The same product for a SKINNY op(A) (m <= 48 rows against thousands of columns: the bit-sorted coefficient read-out):
// 32 x 64 / 48 x 64 output tiles with two K tiles in flight instead of 64 x 64 with one, so that a 30-row operand does not pay
// for 64 rows of matrix-core work and the long operand streams at more than one tile's latency allows.
int qil_dev_gemm_skinny(qil_context* ctx, int dtype, int opA, int opB, int64_t m, int64_t n, int64_t k,
                        const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc);
// Strided batch of the same product (grid y = batch): operand/result b lives at base + b * stride elements;
// optionally B of batch b is shifted by b_sel[b * b_sel_step] * b_sel_stride further elements.
struct qil_gemm_batch {
    int64_t count = 1;
    int64_t a_bs = 0, b_bs = 0, c_bs = 0;
    const uint8_t* b_sel = nullptr;
    int64_t b_sel_step = 0, b_sel_stride = 0;
};
int qil_dev_gemm_batched(qil_context* ctx, int dtype, int opA, int opB, int64_t m, int64_t n, int64_t k,
                         const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                         const qil_gemm_batch* batch);
// |r_jj|^2 (j < n <= 1024) of a triangular factor on the device, to the host
int qil_dev_diag_abs2(qil_context* ctx, int dtype, const void* R, int64_t ldr, int64_t n, double* host_out);
// At (n x m, ldt) = A^T (conj = 0) or A^H (conj = 1)
int qil_dev_transpose(qil_context* ctx, int dtype, int conj, int64_t m, int64_t n, const void* A, int64_t lda,
                      void* At, int64_t ldt);
// Thin SVD of A (m x n, lda) on the device: U (m x r0), S (r0, host), Vh (r0 x n), r0 = min(m,n).
// A is destroyed.  Singular values sorted descending.
// negligible_rel > 0: columns with |a|^2 < negligible_rel |A|_F^2 are left alone by the rotations (rounding residue
// of a rank-deficient operand; callers that truncate pass a value far below their cutoff).
int qil_dev_svd(qil_context* ctx, int dtype, int64_t m, int64_t n, void* A, int64_t lda, void* U,
                int64_t ldu, double* S_host, void* Vh, int64_t ldvh, double negligible_rel = 0.0);
// SVD with ONE isometric factor for the gauge sweeps: B (p x q; destroyed when *handled) = Uiso diag(S) V^H with
// Uiso (p x k, k = min(p, q)) orthonormal columns in descending-S order, SVh (k x q) = diag(S) V^H (no division by S
// anywhere).  Serves the mid-size regime (97 <= k < 640, and smaller operands that do not fit the single-workgroup iteration);
// *handled = 0 leaves B intact for the general qil_dev_svd.
// cert_cutoff > 0: the caller truncates by that cutoff only and ignores S_host; when the triangular factor certifies that no
// singular value can be dropped, the thin QR is returned as the gauge step (*handled = 2, S_host untouched).
int qil_dev_svd_left(qil_context* ctx, int dtype, int64_t p, int64_t q, void* B, int64_t ldb, void* Uiso, int64_t ldu,
                     double* S_host, void* SVh, int64_t ldsvh, double negligible_rel, int* handled, double cert_cutoff = 0.0);
// The same certificate for operands of any size: thin QR of A (m >= n) or A^H (m < n) into Qout (max(m, n) x k, packed) and
// Rout (k x k, packed); *certified says whether a truncation at `cutoff` can drop anything.  A is left intact.
int qil_dev_qr_certified(qil_context* ctx, int dtype, int64_t m, int64_t n, const void* A, int64_t lda, double cutoff, void* Qout,
                         void* Rout, bool* certified);
// V (n x n, ldv) = I (the diagonal is written; the caller zeroes the rest)
int qil_dev_set_identity(qil_context* ctx, int dtype, void* V, int64_t ldv, int64_t n);
// Thin QR with non-negative real diagonal of R: A (m x n, m >= n) -> Q (m x n) in place; R (n x n) optional.
// orthonormal = true: Q^H Q is measured afterwards and Q re-factored while it is not the identity (numerically
// rank-deficient operands; one small GEMM and one stream synchronisation when nothing needs doing).
int qil_dev_qr_positive(qil_context* ctx, int dtype, int64_t m, int64_t n, void* A, int64_t lda,
                        void* R, int64_t ldr, bool orthonormal = false);
// device-to-device copies / zero fills as kernels (they ride the combined launches of a lock-step batch, qil_launch.h);
// pitches and widths in bytes
int qil_dev_copy2d(qil_context* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height);
int qil_dev_copy(qil_context* ctx, void* dst, const void* src, size_t bytes);
int qil_dev_zero2d(qil_context* ctx, void* dst, size_t pitch, size_t width, size_t height);
int qil_dev_zero(qil_context* ctx, void* dst, size_t bytes);
// ITensors truncation rule (host): number of singular values kept.
int64_t qil_truncation_rank(const double* S, int64_t n, double cutoff, bool use_cutoff, int64_t maxdim,
                            int64_t mindim);
// scale columns (side=1: A[:, j] *= s[j]) or rows (side=0: A[i, :] *= s[i]) by real s (device copy made)
int qil_dev_scale(qil_context* ctx, int dtype, int side, int64_t m, int64_t n, void* A, int64_t lda,
                  const double* s_host);
int qil_dev_fill_normal(qil_context* ctx, int dtype, void* p, int64_t n_elems, uint64_t seed,
                        double scale);
