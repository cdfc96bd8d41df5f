// Kernel launches that can be COMBINED across the independent chains of a batch (gfx950 only).
//
// A truncation chain (compress!, zip_to_compress_mpo, apply_compress, the encoders) is ~10^4 small dependent kernels; one
// chain keeps a few percent of the chip busy, and the runtime multiplexes the streams of concurrent chains onto 4 hardware
// queues -- 8 chains on 8 streams take 2x one chain (16 take 4x; GPU_MAX_HW_QUEUES = 8 / 16: 3x), whatever the host does.
// The way to more than 4 chains in flight is ONE launch for the same step of all chains:
//
//   * every kernel of the chain is written as a device function `body(blockIdx, gridDim, args...)` (the two parameters shadow
//     the built-ins, the body itself is what it was) wrapped by a functor F; `qil_k1<F>` is the plain kernel, `qil_kn<F>` the
//     table form: kernel argument = up to QIL_MAXB argument packs + grids, blockIdx.y selects the operand, blocks beyond an
//     operand's own grid leave at once;
//   * `qil_klaunch<F>(ctx, grid, block, lds, args...)` launches `qil_k1` -- or, when ctx is a slot of a running lock-step batch,
//     hands the launch to the batch's combiner (qil_context.hip): when every live chain of the batch has arrived, the launches
//     of the chains that are FURTHEST BEHIND (smallest progress key, set by the chain code at site / phase boundaries) go
//     out, one `qil_kn` launch per kernel class, on the one stream all slots share; chains that are ahead keep waiting, so
//     data-dependent extra work of one chain (one more sweep, a second QR) does not leave the chains out of phase for good.
//     The arithmetic of every operand is the single-operand kernel's, bit for bit.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>
#include <type_traits>

#include "qil_internal.h"

constexpr int QIL_MAXB = 16;             // operands per combined launch (kernel argument <= 4 KB); 8 -> 16: 64 chains 184 -> 127 ms, smaller batches unchanged
constexpr int QIL_PACK_MAX = 288;        // bytes of one argument pack

// ---------------------------------------------------------------- argument packs (trivially copyable tuples)
template <class... A>
struct qil_pack;
template <>
struct qil_pack<> {};
template <class H, class... R>
struct qil_pack<H, R...> {
    H h;
    qil_pack<R...> r;
};
template <class H, class... R>
inline qil_pack<H, R...> qil_make_pack(H h, R... r) {
    qil_pack<H, R...> p;
    p.h = h;
    if constexpr (sizeof...(R) > 0) p.r = qil_make_pack<R...>(r...);
    return p;
}
template <class F, class... Un>
__device__ __forceinline__ void qil_unpack(const uint3 b, const uint3 g, const qil_pack<>&, Un... un) {
    F::run(b, g, un...);
}
template <class F, class H, class... R, class... Un>
__device__ __forceinline__ void qil_unpack(const uint3 b, const uint3 g, const qil_pack<H, R...>& p, Un... un) {
    qil_unpack<F>(b, g, p.r, un..., p.h);
}

template <class... A>
struct qil_ktab {
    unsigned gx[QIL_MAXB], gy[QIL_MAXB], gz[QIL_MAXB];
    qil_pack<A...> it[QIL_MAXB];
};

template <class F, class... A>
__global__ __launch_bounds__(F::NT, F::MINW) void qil_k1(A... a) {
    F::run(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), make_uint3(gridDim.x, gridDim.y, gridDim.z), a...);
}
template <class F, class... A>
__global__ __launch_bounds__(F::NT, F::MINW) void qil_kn(const qil_ktab<A...> t) {
    const unsigned item = blockIdx.y, flat = blockIdx.x;
    const unsigned gx = t.gx[item], gy = t.gy[item], gz = t.gz[item];
    if (flat >= gx * gy * gz) return;
    const uint3 b = make_uint3(flat % gx, (flat / gx) % gy, flat / (gx * gy)), g = make_uint3(gx, gy, gz);
    qil_unpack<F>(b, g, t.it[item]);
}

// ---------------------------------------------------------------- the combiner side (qil_context.hip)
// Every chain of a lock-step batch runs on a host thread of its own and PRODUCES launch requests into its ring; the thread
// that called the batch entry point is the launcher: it looks at the heads of the rings, takes those of the chains that are
// furthest behind (a chain whose ring is empty counts with the key it is working at), groups them by kernel class and issues
// one table launch per class.  A chain only waits for the launcher where it needs stream order for something else (a copy
// to the host, an event, a synchronisation, a launch that does not go through qil_klaunch): qil_stream(ctx) drains its ring.
struct qil_launch_req {
    const void* kern = nullptr;          // identity of the kernel class (address of the table kernel)
    dim3 grid, block;
    size_t lds = 0;
    alignas(16) unsigned char blob[QIL_PACK_MAX];
    // launches the table kernel for `n` requests of this class on `s`
    int (*launch_group)(qil_launch_req* const* reqs, int n, hipStream_t s) = nullptr;
    uint64_t progress = 0;               // (progress key, position inside the key's segment): the launcher's order
    unsigned seq = 0;
};
struct qil_lockstep;                                      // per running batch (home context)
// the slot's ring: returns the request to fill (waits while the ring is full); qil_lockstep_commit publishes it
qil_launch_req* qil_lockstep_begin(qil_lockstep* ls, qil_context* ctx);
int qil_lockstep_commit(qil_lockstep* ls, qil_context* ctx);   // returns the chain's sticky launch status
// chain code: where this chain is (larger = further along); chains with the smallest key are served first.
// key = (pass, step, phase): qil_progress_step opens step `step` of the chain's next pass (new_pass) or of the current one and
// resets the phase; qil_progress_phase moves on inside the step (phases of a step in program order)
void qil_progress_step(qil_context* ctx, bool new_pass, long long step);
void qil_progress_phase(qil_context* ctx, int phase);

template <class F, class... A>
int qil_launch_group(qil_launch_req* const* reqs, int n, hipStream_t s) {
    static_assert(sizeof(qil_ktab<A...>) <= 4000, "table of argument packs must fit the kernel argument segment");
    qil_ktab<A...> t;
    unsigned maxflat = 0;
    for (int i = 0; i < n; ++i) {
        t.gx[i] = reqs[i]->grid.x;
        t.gy[i] = reqs[i]->grid.y;
        t.gz[i] = reqs[i]->grid.z;
        maxflat = std::max(maxflat, reqs[i]->grid.x * reqs[i]->grid.y * reqs[i]->grid.z);
        memcpy(&t.it[i], reqs[i]->blob, sizeof(qil_pack<A...>));
    }
    for (int i = n; i < QIL_MAXB; ++i) t.gx[i] = t.gy[i] = t.gz[i] = 0;
    // requests of one class may ask for different amounts of dynamic LDS (it follows the operand's shape): the launch takes the
    // largest -- an operand that gets more than it asked for runs the same code on the same data
    size_t lds = 0;
    for (int i = 0; i < n; ++i) lds = std::max(lds, reqs[i]->lds);
    hipLaunchKernelGGL((qil_kn<F, A...>), dim3(maxflat, (unsigned)n), reqs[0]->block, lds, s, t);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return qil_fail(QIL_EHIP, "combined launch failed: %s", hipGetErrorString(e));
    return QIL_OK;
}

template <class F, class... A>
int qil_klaunch(qil_context* ctx, dim3 grid, dim3 block, size_t lds, A... args) {
    static_assert((std::is_trivially_copyable<A>::value && ...), "kernel arguments must be trivially copyable");
    static_assert(sizeof(qil_pack<A...>) <= QIL_PACK_MAX, "argument pack too large for a combined launch");
    if (lds > 64 * 1024) {                                    // both forms, whenever an instantiation asks for more than before
        static qil_lds_grant g1, gn;                              // per device (qil_internal.h)
        QIL_HIP(g1.ensure(ctx->device, reinterpret_cast<const void*>(&qil_k1<F, A...>), lds));
        QIL_HIP(gn.ensure(ctx->device, reinterpret_cast<const void*>(&qil_kn<F, A...>), lds));
    }
    qil_lockstep* ls = ctx->lockstep;
    if (!ls) {
        hipLaunchKernelGGL((qil_k1<F, A...>), grid, block, lds, ctx->stream, args...);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return qil_fail(QIL_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
        return QIL_OK;
    }
    qil_launch_req* req = qil_lockstep_begin(ls, ctx);
    req->kern = reinterpret_cast<const void*>(&qil_kn<F, A...>);
    req->grid = grid;
    req->block = block;
    req->lds = lds;
    const qil_pack<A...> p = qil_make_pack<A...>(args...);
    memcpy(req->blob, &p, sizeof(p));
    req->launch_group = &qil_launch_group<F, A...>;
    return qil_lockstep_commit(ls, ctx);
}
