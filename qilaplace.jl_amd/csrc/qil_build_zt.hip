// build_zt_mpo ENTIRELY on the device behind one C verb.
//
//   build_zt_mpo(n, wr, sites_main, sites_copy; cutoff=1e-14, maxdim=1000)      src/transforms/zt_transformer.jl:41-112
//       :74      W_dt   = build_dt_mpo(n, wr)                 -> dt_build_persistent, one workgroup per damping value
//       :78-99   mpo_qft = paired-register QFT chain          -> chain_build_persistent, one workgroup, does not depend on wr
//       :103     W_zt   = apply(W_dt, mpo_qft)                -> mpo_compose_site per site (qil_apply_mpo_mpo)
//       :104     zip_to_compress_mpo(W_zt, "down")            -> qil_mpo_compress (exact QR gauge sweep, truncating SVD sweep)
//
// The two persistent builders occupy nb + 1 of the 256 CUs and neither depends on the other: they run CONCURRENTLY, the DT
// halves on the context's stream (calling thread) and the QFT chain on a worker stream (second host thread) -- the chain
// (52 ms at n = 24) disappears behind the DT build (130-157 ms).  For a sweep the QFT chain is built ONCE, every product
// shares it, and the nb compressions run as one batch (lock-step table launches from 5 values on).  Nothing but 2 x 2 gate
// entries is computed on the host.
#include <algorithm>
#include <vector>

#include "qil_internal.h"

int qil_build_chain_persistent(qil_context* ctx, int kind, int64_t n, double cutoff, int64_t maxdim, const int64_t* site_ids,
                               qil_mpo** out, int* fallback);
int qil_apply_mpo_mpo_shared(const qil_mpo* W1, const qil_mpo* W2, qil_mpo** out);
int qil_build_zt_qft_chain_generic(qil_context* ctx, int64_t n, double cutoff, int64_t maxdim, const int64_t* site_ids, qil_mpo** out);

extern "C" int qil_build_zt_mpo_batch(qil_context* ctx, int64_t n, int64_t nb, const double* wrs, double cutoff,
                                      int64_t maxdim, const int64_t* site_ids, qil_mpo** out) {
    QIL_REQUIRE(ctx && wrs && out, QIL_EINVAL_ARG, "build_zt_mpo: null argument");
    QIL_REQUIRE(n >= 1, QIL_EINVAL_ARG, "build_zt_mpo: n must be >= 1. Found n=%lld", (long long)n);
    QIL_REQUIRE(nb >= 1 && nb <= 4096, QIL_EINVAL_ARG, "build_zt_mpo: batch of %lld damping values", (long long)nb);
    QIL_REQUIRE(cutoff >= 0, QIL_EINVAL_ARG, "build_zt_mpo: cutoff must be >= 0");
    QIL_TRY(qil_ctx_activate(ctx));
    qil_call_scope call_scope(ctx);
    std::vector<qil_mpo*> dts((size_t)nb, nullptr), prods((size_t)nb, nullptr);
    qil_mpo* Q = nullptr;
    auto drop = [&]() {
        for (auto& W : dts)
            if (W) qil_mpo_destroy(W), W = nullptr;
        for (auto& W : prods)
            if (W) qil_mpo_destroy(W), W = nullptr;
        if (Q) qil_mpo_destroy(Q), Q = nullptr;
    };
    // :74 || :78-99 -- item 0 on the home stream, item 1 on a worker stream; chains a worker created return home with the batch
    int st = qil_run_batch_on(ctx, 2, nullptr, [&](int64_t j, qil_context* work) -> int {
        if (j == 0) return qil_build_dt_mpo_batch(work, n, nb, wrs, cutoff, maxdim, site_ids, dts.data());
        QIL_TRY(qil_ctx_activate(work));
        qil_call_scope scope(work);
        int fallback = 0;
        QIL_TRY(qil_build_chain_persistent(work, 1, n, cutoff, maxdim, site_ids, &Q, &fallback));
        if (fallback) QIL_TRY(qil_build_zt_qft_chain_generic(work, n, cutoff, maxdim, site_ids, &Q));
        return QIL_OK;
    });
    if (st != QIL_OK) {
        drop();
        return st;
    }
    // :103 -- "W_dt first, then the QFT chain": W_dt's output leg feeds the chain's input leg (apply.jl:163-171) -- and :104
    // (n == 1 returns the bare product, :66-70).  One value: on the context's stream.  Several: value b's product AND its
    // compression form one chain of the batch, on the slot that owns W_dt[b]; the QFT chain stays in the home context and is
    // read by every slot (64 values: the 64 x 48 product launches no longer queue up on one stream ahead of the batch).
    auto finish = [&](int64_t b) -> int {
        qil_mpo* P = nullptr;
        QIL_TRY(qil_apply_mpo_mpo_shared(dts[(size_t)b], Q, &P));
        prods[(size_t)b] = P;
        if (n > 1) QIL_TRY(qil_mpo_compress(P, 0, cutoff, maxdim));
        return QIL_OK;
    };
    if (nb == 1) {
        st = finish(0);
    } else {
        st = qil_run_batch_on(
            ctx, nb, [&](int64_t b, qil_context* slot) { qil_chain_rebind(dts[(size_t)b], slot); },
            [&](int64_t b, qil_context*) { return finish(b); });
    }
    for (auto& W : dts)
        if (W) qil_mpo_destroy(W), W = nullptr;
    if (st != QIL_OK) {
        drop();
        return st;
    }
    qil_mpo_destroy(Q);
    for (int64_t b = 0; b < nb; ++b) out[b] = prods[(size_t)b];
    return QIL_OK;
}
