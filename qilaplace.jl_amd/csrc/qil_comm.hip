// Multi-GPU verbs of the C ABI: the batched gather of a sweep's coefficient samples over RCCL / xGMI.
//
// The path shards by independent (signal, damping value) items dealt round-robin to one process per GPU; the ONLY exchange is
// one all-gather of the per-item coefficient batches at the end (64 x 1024 x 16 B = 1 MiB at cfg4: latency-bound, the ring
// bound of 7 links x ~153 GB/s is not approached).  The reference's callers loop serially (docs/src/tutorials/zt.jl:300-348,
// scripts/benchmark/zt_full_runtime.jl:151-221).  Until r04 this exchange existed only as torch.distributed in sweep.py, so
// a Julia host calling the library through `ccall` had no route to it.
//
// RCCL is NOT a link-time dependency (libqilhip.so links libamdhip64 only): qil_comm_create dlopens it -- QIL_RCCL_LIB, else the
// librccl.so sitting NEXT TO the libamdhip64 this process runs on (one HIP runtime per process: under PyTorch that is torch's
// bundled pair, otherwise /opt/rocm/lib), else the loader's search path -- and fails loudly when there is none.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <string>
#include <vector>

#include "qil_internal.h"

namespace {

struct RcclApi {
    void* handle = nullptr;
    std::string path;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;          // optional: a rank that fails before the collective releases its peers
};

std::mutex g_rccl_mutex;
RcclApi g_rccl;

int load_rccl(RcclApi** out) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (!g_rccl.handle) {
        std::vector<std::string> cands;
        if (const char* e = getenv("QIL_RCCL_LIB")) cands.emplace_back(e);
        Dl_info info;
        if (dladdr(reinterpret_cast<const void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t cut = dir.find_last_of('/');
            if (cut != std::string::npos) {
                cands.push_back(dir.substr(0, cut) + "/librccl.so");
                cands.push_back(dir.substr(0, cut) + "/librccl.so.1");
            }
        }
        cands.emplace_back("librccl.so.1");
        cands.emplace_back("librccl.so");
        std::string tried;
        for (const std::string& c : cands) {
            if (void* h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL)) {
                g_rccl.handle = h;
                g_rccl.path = c;
                break;
            }
            tried += (tried.empty() ? "" : ", ") + c;
        }
        if (!g_rccl.handle) return qil_fail(QIL_EHIP, "qil_comm: no RCCL library could be loaded (tried %s)", tried.c_str());
#define QIL_SYM(field, name)                                                                                  \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.handle, name));                     \
    if (!g_rccl.field) {                                                                                      \
        dlclose(g_rccl.handle);                                                                               \
        g_rccl.handle = nullptr;                                                                              \
        return qil_fail(QIL_EHIP, "qil_comm: %s has no symbol %s", g_rccl.path.c_str(), name);               \
    }
        QIL_SYM(GetUniqueId, "ncclGetUniqueId")
        QIL_SYM(CommInitRank, "ncclCommInitRank")
        QIL_SYM(CommDestroy, "ncclCommDestroy")
        QIL_SYM(AllGather, "ncclAllGather")
        QIL_SYM(GetErrorString, "ncclGetErrorString")
#undef QIL_SYM
        g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(g_rccl.handle, "ncclCommAbort"));
    }
    *out = &g_rccl;
    return QIL_OK;
}

}  // namespace

struct qil_comm {
    qil_context* ctx = nullptr;
    RcclApi* api = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

#define QIL_NCCL(api, expr)                                                                              \
    do {                                                                                                 \
        const ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) return qil_fail(QIL_EHIP, "RCCL: %s failed: %s", #expr, (api)->GetErrorString(r_)); \
    } while (0)

static_assert(QIL_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "qil_comm id size = ncclUniqueId");

extern "C" int qil_comm_unique_id(void* id_out) {
    QIL_REQUIRE(id_out, QIL_EINVAL_ARG, "qil_comm_unique_id: null out");
    RcclApi* api = nullptr;
    QIL_TRY(load_rccl(&api));
    ncclUniqueId id;
    QIL_NCCL(api, api->GetUniqueId(&id));
    memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return QIL_OK;
}

extern "C" int qil_comm_create(qil_context* ctx, int rank, int world, const void* id, qil_comm** out) {
    QIL_REQUIRE(ctx && id && out, QIL_EINVAL_ARG, "qil_comm_create: null argument");
    QIL_REQUIRE(world >= 1 && rank >= 0 && rank < world, QIL_EINVAL_ARG, "qil_comm_create: rank %d outside [0, %d)", rank, world);
    QIL_TRY(qil_ctx_activate(ctx));
    RcclApi* api = nullptr;
    QIL_TRY(load_rccl(&api));
    ncclUniqueId uid;
    memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    QIL_NCCL(api, api->CommInitRank(&c, world, uid, rank));
    qil_comm* cm = new qil_comm;
    cm->ctx = ctx;
    cm->api = api;
    cm->comm = c;
    cm->rank = rank;
    cm->world = world;
    ctx->comms.insert(cm);
    *out = cm;
    return QIL_OK;
}

// qil_context_destroy: the communicators created on the context go first (a host whose finalizers run in any order
// -- Julia's GC, Python at interpreter shutdown -- may release the context before the communicator; the handle then stays
// valid, holds nothing, and qil_comm_destroy only frees it).
void qil_comm_orphan(qil_comm* cm) {
    if (!cm) return;
    if (cm->comm && cm->ctx) {
        (void)qil_stream_sync(cm->ctx);
        (void)cm->api->CommDestroy(cm->comm);
    }
    cm->comm = nullptr;
    cm->ctx = nullptr;
}

extern "C" int qil_comm_destroy(qil_comm* comm) {
    if (!comm) return QIL_OK;
    int st = QIL_OK;
    if (comm->comm && comm->ctx) {
        (void)qil_ctx_activate(comm->ctx);
        (void)qil_stream_sync(comm->ctx);
        const ncclResult_t r = comm->api->CommDestroy(comm->comm);
        if (r != ncclSuccess) st = qil_fail(QIL_EHIP, "RCCL: ncclCommDestroy failed: %s", comm->api->GetErrorString(r));
    }
    if (comm->ctx) comm->ctx->comms.erase(comm);
    delete comm;
    return st;
}

extern "C" int qil_comm_info(const qil_comm* comm, int* rank, int* world) {
    QIL_REQUIRE(comm, QIL_EINVAL_ARG, "qil_comm_info: null communicator");
    if (rank) *rank = comm->rank;
    if (world) *world = comm->world;
    return QIL_OK;
}

// Static round-robin: rank r owns items r, r + world, ... (sweep.py shard_items); every rank contributes `per` =
// ceil(n_items / world) slots of `width` complex values (unused slots zero).  `gathered` = world blocks of per x width c64 in
// rank order; `out` = n_items x width c64 in item order.  Pure host code (tested on CPU against the gloo path).
extern "C" int qil_sweep_unshuffle(int world, int64_t n_items, int64_t width, const double* gathered, double* out) {
    QIL_REQUIRE(world >= 1 && n_items >= 0 && width >= 0, QIL_EINVAL_ARG, "qil_sweep_unshuffle: bad sizes");
    if (n_items == 0 || width == 0) return QIL_OK;
    QIL_REQUIRE(gathered && out, QIL_EINVAL_ARG, "qil_sweep_unshuffle: null buffer");
    const int64_t per = (n_items + world - 1) / world;
    for (int r = 0; r < world; ++r)
        for (int64_t slot = 0, i = r; i < n_items; ++slot, i += world)
            memcpy(out + 2 * width * i, gathered + 2 * width * (per * r + slot), (size_t)width * 16);
    return QIL_OK;
}

// gathered (world blocks of per x width c64, rank order) -> out (n_items x width c64, item order) on the device: the layout rule
// of qil_sweep_unshuffle as a kernel, so that a sweep's samples never leave HBM between the read-out and the gathered table
namespace {
typedef double qil_c2 __attribute__((ext_vector_type(2)));
__global__ void unshuffle_k(const qil_c2* __restrict__ g, qil_c2* __restrict__ out, int world, long long n_items, long long width, long long per) {
    const long long total = n_items * width;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const long long i = t / width, w = t - i * width;
        out[t] = g[(per * (i % world) + i / world) * width + w];
    }
}
}  // namespace

extern "C" int qil_sweep_unshuffle_device(qil_context* ctx, int world, int64_t n_items, int64_t width, const void* gathered_dev, void* out_dev) {
    QIL_REQUIRE(ctx && world >= 1 && n_items >= 0 && width >= 0, QIL_EINVAL_ARG, "qil_sweep_unshuffle_device: bad arguments");
    if (n_items == 0 || width == 0) return QIL_OK;
    QIL_REQUIRE(gathered_dev && out_dev, QIL_EINVAL_ARG, "qil_sweep_unshuffle_device: null buffer");
    QIL_TRY(qil_ctx_activate(ctx));
    const long long per = (n_items + world - 1) / world;
    const unsigned grid = (unsigned)std::min<long long>((n_items * width + 255) / 256, 4096);
    hipLaunchKernelGGL(unshuffle_k, dim3(grid), dim3(256), 0, qil_stream(ctx), (const qil_c2*)gathered_dev, (qil_c2*)out_dev, world,
                       (long long)n_items, (long long)width, per);
    QIL_HIP(hipGetLastError());
    return QIL_OK;
}

// A rank that fails BEFORE the collective would leave its peers blocked in ncclAllGather for ever: abort the
// communicator so that they fail instead (ncclCommAbort where the library has it), and leave a dead handle behind.
static int fail_before_collective(qil_comm* comm, int st) {
    if (st != QIL_OK && comm->comm && comm->api->CommAbort && comm->world > 1) {
        (void)comm->api->CommAbort(comm->comm);
        comm->comm = nullptr;
    }
    return st;
}

// local_dev: this rank's (ceil-share) x width c64 in slot order (slot k = item rank + k world), in HBM of the communicator's
// context; out_dev: n_items x width c64 in item order.  Stream-ordered on the context's stream, no host synchronisation.
// Every rank must pass the SAME n_items and width (they size the collective).
static int gather_device_impl(qil_comm* comm, int64_t n_items, int64_t width, const void* local_dev, void* out_dev) {
    qil_context* ctx = comm->ctx;
    const int world = comm->world;
    const int64_t per = (n_items + world - 1) / world;
    const int64_t mine = comm->rank < n_items ? (n_items - comm->rank + world - 1) / world : 0;
    const size_t block = (size_t)per * (size_t)width * 16;
    void *dsend = nullptr, *drecv = nullptr;
    int st = qil_ctx_alloc(ctx, block, &dsend);
    if (st == QIL_OK) st = qil_ctx_alloc(ctx, block * (size_t)world, &drecv);
    hipStream_t s = qil_stream(ctx);
    if (st == QIL_OK && hipMemsetAsync(dsend, 0, block, s) != hipSuccess) st = qil_fail(QIL_EHIP, "qil_gather_coefficients: hipMemsetAsync failed");
    if (st == QIL_OK && mine > 0 &&
        hipMemcpyAsync(dsend, local_dev, (size_t)mine * (size_t)width * 16, hipMemcpyDeviceToDevice, s) != hipSuccess)
        st = qil_fail(QIL_EHIP, "qil_gather_coefficients: staging this rank's samples failed");
    if (st != QIL_OK) return fail_before_collective(comm, st);
    // the one collective of the sweep: 2 * per * width doubles per rank
    QIL_NCCL(comm->api, comm->api->AllGather(dsend, drecv, (size_t)(2 * per * width), ncclFloat64, comm->comm, s));
    QIL_TRY(qil_sweep_unshuffle_device(ctx, world, n_items, width, drecv, out_dev));
    qil_ctx_free(ctx, dsend);                             // (pool blocks are recycled in stream order)
    qil_ctx_free(ctx, drecv);
    return QIL_OK;
}

extern "C" int qil_gather_coefficients_device(qil_comm* comm, int64_t n_items, int64_t width, const void* local_dev, void* out_dev) {
    QIL_REQUIRE(comm && comm->comm && comm->ctx, QIL_EINVAL_ARG, "qil_gather_coefficients_device: null or dead communicator");
    QIL_REQUIRE(n_items >= 0 && width >= 0, QIL_EINVAL_ARG, "qil_gather_coefficients_device: bad sizes");
    if (n_items == 0 || width == 0) return QIL_OK;
    QIL_REQUIRE(out_dev && (local_dev || comm->rank >= n_items), QIL_EINVAL_ARG, "qil_gather_coefficients_device: null buffer");
    QIL_TRY(qil_ctx_activate(comm->ctx));
    qil_call_scope call_scope(comm->ctx);
    return gather_device_impl(comm, n_items, width, local_dev, out_dev);
}

extern "C" int qil_gather_coefficients(qil_comm* comm, int64_t n_items, int64_t width, const double* local, double* out) {
    QIL_REQUIRE(comm && comm->comm && comm->ctx, QIL_EINVAL_ARG, "qil_gather_coefficients: null or dead communicator");
    QIL_REQUIRE(n_items >= 0 && width >= 0, QIL_EINVAL_ARG, "qil_gather_coefficients: bad sizes");
    if (n_items == 0 || width == 0) return QIL_OK;
    QIL_REQUIRE(local && out, QIL_EINVAL_ARG, "qil_gather_coefficients: null buffer");
    qil_context* ctx = comm->ctx;
    QIL_TRY(qil_ctx_activate(ctx));
    qil_call_scope call_scope(ctx);
    const int world = comm->world;
    const int64_t mine = comm->rank < n_items ? (n_items - comm->rank + world - 1) / world : 0;
    void *dloc = nullptr, *dout = nullptr;
    int st = qil_ctx_alloc(ctx, (size_t)std::max<int64_t>(mine, 1) * (size_t)width * 16, &dloc);
    if (st == QIL_OK) st = qil_ctx_alloc(ctx, (size_t)n_items * (size_t)width * 16, &dout);
    if (st == QIL_OK && mine > 0 &&
        hipMemcpyAsync(dloc, local, (size_t)mine * (size_t)width * 16, hipMemcpyHostToDevice, qil_stream(ctx)) != hipSuccess)
        st = qil_fail(QIL_EHIP, "qil_gather_coefficients: upload of this rank's samples failed");
    if (st != QIL_OK) return fail_before_collective(comm, st);
    QIL_TRY(gather_device_impl(comm, n_items, width, dloc, dout));
    QIL_HIP(hipMemcpyAsync(out, dout, (size_t)n_items * (size_t)width * 16, hipMemcpyDeviceToHost, qil_stream(ctx)));
    QIL_HIP(qil_stream_sync(ctx));
    qil_ctx_free(ctx, dloc);
    qil_ctx_free(ctx, dout);
    return QIL_OK;
}

// The body of a damping sweep ACROSS the ranks of a communicator (docs/src/tutorials/dt.jl:150-197, zt.jl:300-348):
// this rank's operators Ws[0..nw) are its round-robin share of n_items operators (slot k = item rank + k world, so nw must be
// that share's size); each product W psi is read out at the nb configurations, the samples stay in HBM, ONE ncclAllGather
// exchanges them and the table out[n_items x nb] (item order, complex) reaches the host in one copy on every rank.
int qil_apply_coefficient_sweep_dev(const qil_mpo* const* Ws, int64_t nw, const qil_mps* psi, int64_t nb, const uint8_t* bits, void* dout);

extern "C" int qil_apply_coefficient_sweep_gather(qil_comm* comm, const qil_mpo* const* Ws, int64_t nw, const qil_mps* psi, int64_t nb,
                                                  const uint8_t* bits, int64_t n_items, double* out) {
    QIL_REQUIRE(comm && comm->comm && comm->ctx, QIL_EINVAL_ARG, "apply_coefficient_sweep_gather: null or dead communicator");
    QIL_REQUIRE(psi && (nw == 0 || Ws) && n_items >= 0 && nb >= 0, QIL_EINVAL_ARG, "apply_coefficient_sweep_gather: bad arguments");
    QIL_REQUIRE(psi->ctx == comm->ctx, QIL_EINVAL_ARG, "apply_coefficient_sweep_gather: the state lives in another context than the communicator");
    const int world = comm->world;
    const int64_t mine = comm->rank < n_items ? (n_items - comm->rank + world - 1) / world : 0;
    QIL_REQUIRE(nw == mine, QIL_EINVAL_LENGTH, "apply_coefficient_sweep_gather: rank %d of %d owns %lld of %lld items, got %lld operators",
                comm->rank, world, (long long)mine, (long long)n_items, (long long)nw);
    if (n_items == 0 || nb == 0) return QIL_OK;
    QIL_REQUIRE(bits && out, QIL_EINVAL_ARG, "apply_coefficient_sweep_gather: null buffer");
    qil_context* ctx = comm->ctx;
    QIL_TRY(qil_ctx_activate(ctx));
    qil_call_scope call_scope(ctx);
    void *dloc = nullptr, *dout = nullptr;
    int st = qil_ctx_alloc(ctx, (size_t)std::max<int64_t>(mine, 1) * (size_t)nb * 16, &dloc);
    if (st == QIL_OK) st = qil_ctx_alloc(ctx, (size_t)n_items * (size_t)nb * 16, &dout);
    if (st == QIL_OK && mine > 0) st = qil_apply_coefficient_sweep_dev(Ws, nw, psi, nb, bits, dloc);
    if (st != QIL_OK) return fail_before_collective(comm, st);
    QIL_TRY(gather_device_impl(comm, n_items, nb, dloc, dout));
    QIL_HIP(hipMemcpyAsync(out, dout, (size_t)n_items * (size_t)nb * 16, hipMemcpyDeviceToHost, qil_stream(ctx)));
    QIL_HIP(qil_stream_sync(ctx));
    qil_ctx_free(ctx, dloc);
    qil_ctx_free(ctx, dout);
    return QIL_OK;
}
