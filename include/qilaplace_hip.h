/*
 * qilaplace_hip.h -- C ABI of libqilhip.so, the MI355X (gfx950) implementation of
 * QILaplace.jl's MPO x MPS apply / coefficient / compress / encode hot path.
 *
 * The reference (SUTD-MDQS/QILaplace.jl v0.1.1) has NO FFI layer: its boundary is
 * Julia multiple dispatch on ITensors-backed containers.  Each entry point below
 * names the reference method it stands in for (file:line relative to the
 * reference tree); INTEGRATION.md shows the `ccall` methods a maintainer adds so
 * `apply`, `*`, `coefficient`, `compress!`, `signal_mps` ... dispatch here.
 *
 * Data layout at the boundary (fixed, canonical; all COLUMN-MAJOR, first index
 * fastest, complex = interleaved (re, im) doubles):
 *     MPS site   A[alpha, s, beta]          dims (chi_l, 2, chi_r)
 *     MPO site   W[a, s_in, s_out, b]       dims (D_l, 2, 2, D_r)
 *         s_in  = the reference's primed leg  s'  (contracted with the MPS)
 *         s_out = the reference's unprimed leg s  (survives)    src/linalg/apply.jl:98-101
 * Edge tensors carry explicit dimension-1 bonds.  Paired-register objects
 * (ZTMPS / PairedSiteMPO) are passed as their interleaved 2n-tensor chain
 * main_1, copy_1, main_2, ... (src/mps.jl:421-444, src/linalg/apply.jl:16-32)
 * with `paired = 1`.
 *
 * Ownership: handles returned through `out` parameters belong to the caller and
 * must be released with the matching *_destroy.  The library never keeps a host
 * pointer after a call returns.  Device buffers belong to the handle.
 *
 * Errors: every function returns a qil_status; qil_last_error() gives the
 * thread-local message.  The host shims re-raise QIL_EINVAL_* as ArgumentError
 * and QIL_EDOMAIN as DomainError (Julia) / ValueError and ArithmeticError (Python).
 *
 * Threading: one HIP stream per qil_context; calls on one context are serialised
 * by the caller, calls on different contexts are independent.  There is no global
 * RNG (the reference's rsvd reseeds Julia's global RNG, src/linalg/rsvd.jl:74):
 * seeds are explicit parameters.
 */
#ifndef QILAPLACE_HIP_H
#define QILAPLACE_HIP_H

#include <stdint.h>

/* The library is built with -fvisibility=hidden: the entries declared QIL_API (here and in qilaplace_hip_testing.h) are
 * its whole dynamic symbol table -- no C++ internals, no template instantiations (tests/test_cabi_symbols.py). */
#ifndef QIL_API
#define QIL_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qil_context qil_context;
typedef struct qil_mps qil_mps;   /* SignalMPS (src/mps.jl:70-79) or ZTMPS chain (:98-117) */
typedef struct qil_mpo qil_mpo;   /* SingleSiteMPO / PairedSiteMPO (src/mpo.jl:26-74)       */

typedef enum { QIL_F64 = 0, QIL_C64 = 1 } qil_dtype;

typedef enum {
    QIL_OK = 0,
    QIL_EINVAL_LENGTH = 1, /* ArgumentError: site-count mismatch (apply.jl:76-80, 202-203; mps.jl:671-672) */
    QIL_EINVAL_SITES = 2,  /* ArgumentError: site-identity mismatch (apply.jl:81-85, 130)                 */
    QIL_EINVAL_CONFIG = 3, /* ArgumentError: bad bit configuration (mps.jl:612, 619-628, 634-643)         */
    QIL_EDOMAIN = 4,       /* DomainError: N < 2 in compress! (mps.jl:918), bad center (:800, :820)       */
    QIL_ENOMEM = 5,
    QIL_EHIP = 6,          /* a HIP runtime call failed; message carries hipGetErrorString                */
    QIL_EINVAL_ARG = 7,    /* null handle, bad dtype / method / direction (SignalConverters.jl:200-201)   */
    QIL_EEMPTY = 8         /* rsvd: left or right index set empty (rsvd.jl:56-60)                         */
} qil_status;

typedef enum { QIL_METHOD_SVD = 0, QIL_METHOD_RSVD = 1 } qil_method;     /* method=:svd / :rsvd   */
typedef enum { QIL_DIR_RIGHT = 0, QIL_DIR_LEFT = 1 } qil_direction;      /* :right / :left        */

/* "no cap": the reference's typemax(Int) defaults for maxdim */
#define QIL_MAXDIM_NONE INT64_MAX

/* ------------------------------------------------------------------ library / context */
QIL_API const char* qil_last_error(void);
QIL_API const char* qil_version(void);
/* number of visible HIP devices (does not initialise a context) */
QIL_API int qil_device_count(int* out);

/* One context = one device + one HIP stream + a caching device-memory pool.
 * `stream` may be NULL (the context creates its own non-blocking stream) or an
 * existing hipStream_t to enqueue on (e.g. torch's current stream).            */
QIL_API int qil_context_create(int device, void* stream, qil_context** out);
QIL_API int qil_context_destroy(qil_context* ctx);
QIL_API int qil_context_synchronize(qil_context* ctx);
/* release cached (free) device blocks back to the driver */
QIL_API int qil_context_trim(qil_context* ctx);
QIL_API int qil_context_mem_info(qil_context* ctx, int64_t* pool_bytes_in_use, int64_t* pool_bytes_cached,
                         int64_t* device_free, int64_t* device_total);
/* Host CPUs the batch runners of this process may keep busy: min(cgroup CPU quota, affinity mask) / LOCAL_WORLD_SIZE
 * (the ranks torch.distributed.run / bench.py place on this node), overridden by QIL_CPU_BUDGET.  The lock-step batch
 * entry points (qil_*_batch) never run more polling launcher threads than this minus one.  No reference counterpart
 * (the reference is single-threaded Julia + BLAS threads, benchmarking.md:12).                                      */
QIL_API int qil_host_cpu_budget(int* out);

/* (fault injection, pool accounting, HIP-event timers and the apply-kernel profile are not part of the boundary:
 * include/qilaplace_hip_testing.h) */

/* ------------------------------------------------------------------ containers (T1-T3) */
/* Construct from HOST tensors.  bond_dims: the n-1 internal bonds.  site_ids: n
 * labels playing the role of the reference's Index identities (may be NULL =>
 * 1..n).  site_ptrs[i]: host tensor i in the canonical layout above.
 * Replaces SignalMPS(data, sites, bonds; amplitude) src/mps.jl:121-146 and
 * ZTMPS(...) :148-184 (paired = 1, n even).                                       */
QIL_API int qil_mps_create(qil_context* ctx, int64_t n, int dtype, int paired, const int64_t* bond_dims,
                   const int64_t* site_ids, const void* const* site_ptrs, double amplitude,
                   qil_mps** out);
/* Same, but tensors left uninitialised on the device (fill through qil_mps_site_device_ptr). */
QIL_API int qil_mps_alloc(qil_context* ctx, int64_t n, int dtype, int paired, const int64_t* bond_dims,
                  const int64_t* site_ids, double amplitude, qil_mps** out);
QIL_API int qil_mps_destroy(qil_mps* psi);
QIL_API int qil_mps_clone(const qil_mps* psi, qil_mps** out);
QIL_API int qil_mps_nsites(const qil_mps* psi, int64_t* n);
QIL_API int qil_mps_dtype(const qil_mps* psi, int* dtype);
QIL_API int qil_mps_is_paired(const qil_mps* psi, int* paired);
QIL_API int qil_mps_bond_dims(const qil_mps* psi, int64_t* bond_dims /* n-1 */);
QIL_API int qil_mps_site_ids(const qil_mps* psi, int64_t* site_ids /* n */);
QIL_API int qil_mps_amplitude(const qil_mps* psi, double* amplitude);
QIL_API int qil_mps_set_amplitude(qil_mps* psi, double amplitude);
QIL_API int qil_mps_site_nbytes(const qil_mps* psi, int64_t i, int64_t* nbytes);
QIL_API int qil_mps_download_site(const qil_mps* psi, int64_t i, void* host_dst);
QIL_API int qil_mps_upload_site(qil_mps* psi, int64_t i, const void* host_src);
QIL_API int qil_mps_site_device_ptr(const qil_mps* psi, int64_t i, void** dev_ptr);
/* seeded device-side fill with i.i.d. N(0,1)/sqrt(2 chi_l) entries (synthetic workloads) */
QIL_API int qil_mps_fill_random(qil_mps* psi, uint64_t seed);

/* SingleSiteMPO(data, sites, bonds) src/mpo.jl:30-43 / PairedSiteMPO :62-73 (paired = 1) */
QIL_API int qil_mpo_create(qil_context* ctx, int64_t n, int dtype, int paired, const int64_t* bond_dims,
                   const int64_t* site_ids, const void* const* site_ptrs, qil_mpo** out);
QIL_API int qil_mpo_alloc(qil_context* ctx, int64_t n, int dtype, int paired, const int64_t* bond_dims,
                  const int64_t* site_ids, qil_mpo** out);
QIL_API int qil_mpo_destroy(qil_mpo* W);
QIL_API int qil_mpo_nsites(const qil_mpo* W, int64_t* n);
QIL_API int qil_mpo_dtype(const qil_mpo* W, int* dtype);
QIL_API int qil_mpo_is_paired(const qil_mpo* W, int* paired);
QIL_API int qil_mpo_bond_dims(const qil_mpo* W, int64_t* bond_dims /* n-1 */);
QIL_API int qil_mpo_site_ids(const qil_mpo* W, int64_t* site_ids /* n */);
QIL_API int qil_mpo_site_nbytes(const qil_mpo* W, int64_t i, int64_t* nbytes);
QIL_API int qil_mpo_download_site(const qil_mpo* W, int64_t i, void* host_dst);
QIL_API int qil_mpo_site_device_ptr(const qil_mpo* W, int64_t i, void** dev_ptr);
QIL_API int qil_mpo_fill_random(qil_mpo* W, uint64_t seed);

/* ------------------------------------------------------------------ apply (A1-A3) */
/* apply(W::SingleSiteMPO, psi::SignalMPS) src/linalg/apply.jl:75-122 and
 * apply(W::PairedSiteMPO, psi::ZTMPS) :201-218 (and `*`, :233-236).
 *   B_i[(a,alpha), s, (b,beta)] = sum_{s'} W_i[a, s', s, b] * A_i[alpha, s', beta]
 * written once, directly in the fused layout row = alpha + chi_l*a, col = beta + chi_r*b.
 * No truncation (the reference ignores cutoff/maxdim kwargs, apply.jl:75).  Output
 * shares psi's site ids and amplitude (apply.jl:121, :216); dtype = promote(W, psi).
 * Errors: QIL_EINVAL_LENGTH (apply.jl:76-80, 202-203), QIL_EINVAL_SITES (:81-85).  */
QIL_API int qil_apply(const qil_mpo* W, const qil_mps* psi, qil_mps** out);
/* Same, into an existing handle of identical shape/dtype (no allocation). */
QIL_API int qil_apply_into(const qil_mpo* W, const qil_mps* psi, qil_mps* out);
/* apply(W1, W2) MPO x MPO, "W1 first, then W2", window semantics of apply.jl:124-199
 * (paired: :220-230).  QIL_EINVAL_SITES when the supports are disjoint (:130).     */
QIL_API int qil_apply_mpo_mpo(const qil_mpo* W1, const qil_mpo* W2, qil_mpo** out);

/* ------------------------------------------------------------------ read-out (C1, C2, K3) */
/* coefficient(psi, cfg) src/mps.jl:669-693 for nb configurations at once.
 * bits: host, nb x n bytes, query-major, bits[q*n + i] in {0,1} for site i+1 (site 1
 * first = MSB of a signal index; paired: interleaved main_1, copy_1, ...).
 * out: host, nb complex doubles (re, im) = amplitude * prod_i A_i[:, bit_i, :].
 * Errors: QIL_EINVAL_CONFIG for a bit outside [0,1] (mps.jl:612).                  */
QIL_API int qil_coefficient_batch(const qil_mps* psi, int64_t nb, const uint8_t* bits, double* out);
/* Same with bit value 2 allowed = "sum over this site's physical index" (marginal / partial trace with
 * the all-ones vector).  Serves the coefficient-grid and Laplace-value scans of the tutorials
 * (docs/src/tutorials/dt.jl:187-197 sums N coefficient calls per value; zt.jl:283-309 scans 256 x 256
 * grids) with one chain per value instead of N. */
QIL_API int qil_coefficient_marginal_batch(const qil_mps* psi, int64_t nb, const uint8_t* bits, double* out);
/* <bits| W psi> without materialising W*psi (same numbers as
 * qil_coefficient_batch(qil_apply(W, psi))).                                      */
QIL_API int qil_apply_coefficient_batch(const qil_mpo* W, const qil_mps* psi, int64_t nb,
                                const uint8_t* bits, double* out);
/* The body of a damping sweep: for each of the nw operators (the reference's loop `W = build_dt_mpo(psi, wr);
 * out = W * psi; coefficient(out, ...)`, docs/src/tutorials/dt.jl:150-197, zt.jl:300-348) the product W_j psi is
 * materialised by the apply kernel (apply.jl:75-122) and read out at the same nb configurations (mps.jl:669-693).
 * out: host, nw x nb complex doubles, operator-major.  One upload, one download, one synchronisation for the batch.  */
QIL_API int qil_apply_coefficient_sweep(const qil_mpo* const* Ws, int64_t nw, const qil_mps* psi, int64_t nb,
                                const uint8_t* bits, double* out);
/* mps_to_vector(psi; reverse) src/mps.jl:716-743: 2^n values of psi's dtype, times amplitude. */
QIL_API int qil_mps_to_vector(const qil_mps* psi, int reverse, void* host_out);
/* Dense read-out of a sub-lattice of configurations: spec[i] = 0 / 1 fixes site i's bit, 2 sums the site
 * (marginal), 3 leaves it free; host_out receives the 2^(#free) coefficients (psi's dtype, times amplitude),
 * free sites in chain order, the first one the most significant bit (reverse = 0) or the least (reverse = 1).
 * All free = mps_to_vector (mps.jl:716-743), none free = coefficient (mps.jl:669-678); in between it is the
 * (k, l) grid scan of docs/src/tutorials/zt.jl:283-309 or the N-term Laplace sums of dt.jl:187-197 as one
 * contraction.  QIL_EINVAL_CONFIG for spec values > 3, QIL_EINVAL_LENGTH for more than 34 free sites.      */
QIL_API int qil_mps_block(const qil_mps* psi, const uint8_t* spec, int reverse, void* host_out);
/* norm(psi) src/mps.jl:754-771 (without amplitude). */
QIL_API int qil_norm(const qil_mps* psi, double* out);

/* ------------------------------------------------------------------ truncation (K1, K2) */
/* canonicalize!(psi, direction; center, cutoff=1e-12, maxdim) src/mps.jl:787-847.
 * center = 0 selects the default (N for :right, 1 for :left); 1-based otherwise.    */
QIL_API int qil_canonicalize(qil_mps* psi, int direction, int64_t center, double cutoff, int64_t maxdim);
/* compress!(psi; maxdim, tol=1e-12, sweeps=1) src/mps.jl:913-999.  In place.
 * Accuracy contract: same bond dimensions and amplitude as the reference's rule (cutoff = tol^2 / ((N-1) sweeps),
 * mps.jl:920; gauge passes at canonicalize!'s cutoff 1e-12), truncated state within 1e-9 of the CPU restatement's on
 * sampled coefficients (tests: test_compress_*, test_bench_truncate_operands_against_oracle).  One deliberate
 * deviation from "full SVD of every site": when a site's triangular factor is numerically rank-deficient (every
 * product bond before its truncation) the one-factor SVD first DROPS the rows of that factor whose summed squared
 * weight stays below 1e-6 of the caller's cutoff x |A|_F^2 and factors only the rest (svd_left_deflated).  A dropped
 * weight w costs sqrt(w) in amplitude, i.e. at most 1e-3 of what the cutoff itself is allowed to discard per site:
 * measured on the bond-1008 zT product (maxdim 64, tol 1e-8) the truncated state differs from the CPU restatement's
 * by 4e-10 of the scale with the rule and 1e-11 without it, against 1.8e-5 of truncation error of the algorithm
 * itself.                                                                                                            */
QIL_API int qil_compress(qil_mps* psi, int64_t maxdim, double tol, int sweeps);

/* zip_to_compress_mpo over a whole MPO, in place (src/transforms/dt_transformer.jl:167-288; the step the
 * reference runs on the MPO x MPO product in zt_transformer.jl:103-104).  direction 0 = "down" (exact gauge
 * sweep left -> right, truncating SVD sweep right -> left), 1 = "up" (mirror).  cutoff / maxdim follow the
 * ITensors truncation rule (maxdim <= 0: no cap).  QIL_EINVAL_ARG for any other direction (the reference's
 * `error("unknown direction")`).                                                                         */
QIL_API int qil_mpo_compress(qil_mpo* W, int direction, double cutoff, int64_t maxdim);

/* Batches of independent chains (the serial loops over signals / damping values of
 * scripts/benchmark/zt_full_runtime.jl:151-221 and docs/src/tutorials/zt.jl:300-348 call compress! /
 * zip_to_compress_mpo once per item).  Item j receives exactly qil_compress(items[j], ...) resp.
 * qil_mpo_compress(items[j], ...), in place; the nb chains run concurrently (each is a latency chain of small
 * factorisations that fills a few percent of the chip): up to 4 on streams of their own, larger batches as four lock-step
 * groups whose chains share ONE table launch per step; bit-identical to the item-by-item calls; the call
 * returns when all are done.  All items must live in one context and be distinct handles (QIL_EINVAL_ARG);
 * the first failing item's status is returned, the other items are still processed.                        */
QIL_API int qil_compress_batch(qil_mps* const* items, int64_t nb, int64_t maxdim, double tol, int sweeps);
QIL_API int qil_mpo_compress_batch(qil_mpo* const* items, int64_t nb, int direction, double cutoff, int64_t maxdim);

/* Fused apply-and-truncate: compress!(apply(W, psi); maxdim, tol, sweeps) (apply.jl:75-122 followed by
 * mps.jl:913-973) without materialising the (D chi)^2 product: psi is brought to right-canonical gauge by exact QRs, a zip-up
 * sweep with intermediate bond cap zip_maxdim (<= 0: max(1.5 maxdim, maxdim + 16)) builds a basis per bond (sketched on capped
 * bonds), ONE variational sweep replaces every site by the best tensor given the others, then the exact-gauge compress!
 * runs.  Same error codes as qil_apply / qil_compress.  Not a reference entry point (the reference's apply ignores its
 * cutoff / maxdim kwargs); qil_apply keeps that behaviour.  Accuracy against qil_apply + qil_compress (the exact route):
 * identical bond dimensions and a state error <= 2x the truncation's own on random flat-spectrum products; on transform
 * pipelines identical bonds at tol >= 1e-4 and, below that, bonds that are never larger with a SMALLER error than the exact
 * route, whose gauge passes carry canonicalize!'s fixed cutoff 1e-12 (tests/test_gpu_parity.py,
 * test_apply_compress_*_against_oracle).                                                                          */
QIL_API int qil_apply_compress(const qil_mpo* W, const qil_mps* psi, int64_t maxdim, double tol, int sweeps,
                       int64_t zip_maxdim, qil_mps** out);
/* The same for nb independent (operator, state) pairs of one context -- the (signal, damping value) items of a sweep;
 * Ws / psis entries may repeat (one operator on many signals, many operators on one signal).  outs[j] receives
 * exactly qil_apply_compress(Ws[j], psis[j], ...); the chains run concurrently on the context's streams (see
 * qil_compress_batch).  On failure the first failing item's status is returned and NO handle is handed out.      */
QIL_API int qil_apply_compress_batch(const qil_mpo* const* Ws, const qil_mps* const* psis, int64_t nb, int64_t maxdim,
                             double tol, int sweeps, int64_t zip_maxdim, qil_mps** outs);

/* ------------------------------------------------------------------ encode (E1-E4) */
/* signal_mps(x; method, cutoff, maxdim, k, p, q, random_seed, mindim)
 * src/signals/SignalConverters.jl:228-233.  x: len values of `dtype`, in host memory OR already in HBM (a device
 * pointer is recognised through unified addressing; the caller orders its producer before the call).       */
QIL_API int qil_signal_mps(qil_context* ctx, const void* x, int64_t len, int dtype, int method,
                   double cutoff, int64_t maxdim, int64_t k, int64_t p, int q, uint64_t seed,
                   int64_t mindim, qil_mps** out);
/* signal_ztmps(x; cutoff=1e-10, maxdim, kwargs...) SignalConverters.jl:247-283. */
QIL_API int qil_signal_ztmps(qil_context* ctx, const void* x, int64_t len, int dtype, int method,
                     double cutoff, int64_t maxdim, int64_t k, int64_t p, int q, uint64_t seed,
                     int64_t mindim, qil_mps** out);
/* nb signals of one length and dtype encoded concurrently on the context's streams -- the serial loop over signal kinds
 * of scripts/benchmark/zt_full_runtime.jl:151-221.  outs[j] receives exactly qil_signal_mps / qil_signal_ztmps(xs[j], ...).
 * On failure the first failing signal's status is returned and NO handle is handed out.                            */
QIL_API int qil_signal_mps_batch(qil_context* ctx, const void* const* xs, int64_t nb, int64_t len, int dtype, int method,
                         double cutoff, int64_t maxdim, int64_t k, int64_t p, int q, uint64_t seed, int64_t mindim,
                         qil_mps** outs);
QIL_API int qil_signal_ztmps_batch(qil_context* ctx, const void* const* xs, int64_t nb, int64_t len, int dtype, int method,
                           double cutoff, int64_t maxdim, int64_t k, int64_t p, int q, uint64_t seed, int64_t mindim,
                           qil_mps** outs);
/* rsvd(A, Linds...; k, p, q, random_seed, cutoff, maxdim, mindim) src/linalg/rsvd.jl:38-121
 * on the matricised operand A (m x n, host, column-major).  Outputs (host, caller
 * allocated for rank min(k+p, m, n)): U m x r, S r, Vh r x n; *rank = r kept.        */
QIL_API int qil_rsvd(qil_context* ctx, const void* A, int64_t m, int64_t n, int dtype, int64_t k,
             int64_t p, int q, uint64_t seed, double cutoff, int64_t maxdim, int64_t mindim,
             int64_t* rank, void* U, double* S, void* Vh);
/* truncated svd(A; cutoff, maxdim, mindim) with the ITensors truncation rule (the
 * call sites mps.jl:929,946; SignalConverters.jl:84,266).  Same output contract.   */
QIL_API int qil_svd_trunc(qil_context* ctx, const void* A, int64_t m, int64_t n, int dtype, double cutoff,
                  int64_t maxdim, int64_t mindim, int64_t* rank, void* U, double* S, void* Vh);

/* ------------------------------------------------------------------ transform producers (P2) */
/* build_dt_mpo(n, wr; cutoff=1e-14, maxdim=1000) src/transforms/dt_transformer.jl:312-407 for a BATCH of
 * damping values wr[0..nb): ONE kernel launch, one workgroup per damping value runs that value's whole chain of
 * zip_to_combine / zip_to_compress steps (:20-288) with the tensors being factorised resident in LDS.
 * out[nb] receives PairedSiteMPO handles (f64, 2n tensors), each with its own bond dimensions -- those of a
 * single build_dt_mpo call.  site_ids: the 2n labels of the operand the MPOs will act on (build_dt_mpo(psi::ZTMPS,
 * ...) builds on psi's own sites, dt_transformer.jl:409-412); NULL => 1..2n.  maxdim <= 0: no cap.
 * Bonds beyond the in-LDS capacity (truncated bond > 26; never at the reference's cutoffs) take a launch-per-step
 * route that pads every MPO of the batch to a common bond profile with zero components (same operators).
 * The persistent builders (this one, qil_build_qft_mpo, qil_build_zt_qft_chain, qil_build_zt_mpo_batch) run their truncation rule
 * at max(cutoff, 1e-28): directions below 1e-28 of a bond's weight are rounding residue of exactly rank-deficient product bonds
 * (the reference's LAPACK SVD would keep them at cutoff = 0 as noise-level singular values; the operator is the same to rounding). */
QIL_API int qil_build_dt_mpo_batch(qil_context* ctx, int64_t n, int64_t nb, const double* wrs, double cutoff,
                           int64_t maxdim, const int64_t* site_ids, qil_mpo** out);

/* build_qft_mpo(n, sites; cutoff=1e-14, maxdim=1000) src/transforms/qft_transformer.jl:121-165 ENTIRELY on the device: one
 * launch of one workgroup runs the whole chain of zip-ups (:13-66) and truncating zip-downs (:69-101) with the tensors in
 * LDS (bonds <= 8, <= 16 before a truncation); only the 2 x 2 gate blocks of control_Hphase_mpo (qft_gates.jl:43-97)
 * come from the host.  Result: a SingleSiteMPO handle (complex) with the reference's bond dimensions and, to rounding,
 * its dense operator (gauges differ).  *fallback = 1 (and no handle) when a bond exceeded the in-LDS capacity: the
 * caller takes the generic route (qil_apply_mpo_mpo + qil_mpo_compress per layer).                                    */
QIL_API int qil_build_qft_mpo(qil_context* ctx, int64_t n, double cutoff, int64_t maxdim, const int64_t* site_ids,
                      qil_mpo** out, int* fallback);
/* The paired-register QFT half of build_zt_mpo (src/transforms/zt_transformer.jl:78-99: identity extension, zip_to_combine
 * "down", zip_to_compress "down" per block control_Hphase_ztmps_mpo, zt_gates.jl:12-114), same persistent kernel; a
 * PairedSiteMPO handle over 2 n tensors main_1, copy_1, ...                                                          */
QIL_API int qil_build_zt_qft_chain(qil_context* ctx, int64_t n, double cutoff, int64_t maxdim, const int64_t* site_ids,
                           qil_mpo** out, int* fallback);

/* build_zt_mpo(n, wr, sites_main, sites_copy; cutoff=1e-14, maxdim=1000) src/transforms/zt_transformer.jl:41-112 for a BATCH
 * of damping values wr[0..nb), every step on the device: the DT halves (:74, one launch, one workgroup per value) and the
 * paired-register QFT chain (:78-99, one launch of one workgroup, built once for the whole batch) run CONCURRENTLY on two
 * streams of the context, then per value the MPO x MPO product apply(W_dt, mpo_qft) (:103) and zip_to_compress_mpo "down"
 * (:104; the nb compressions run as one batch).  out[nb] receives PairedSiteMPO handles (complex, 2n tensors) with the bond
 * dimensions of a single build_zt_mpo call each.  site_ids: the 2n labels main_1, copy_1, ... of the operand (build_zt_mpo(
 * psi::ZTMPS, wr), :107-111); NULL => 1..2n.  maxdim <= 0: no cap.  n == 1 returns the bare product (:66-70).
 * QIL_EINVAL_ARG for n < 1 (the reference's ArgumentError, :49).                                                        */
QIL_API int qil_build_zt_mpo_batch(qil_context* ctx, int64_t n, int64_t nb, const double* wrs, double cutoff,
                           int64_t maxdim, const int64_t* site_ids, qil_mpo** out);

/* C (m x n) = opA(A) * opB(B) on host operands, column-major; op: 0 = N, 1 = T, 2 = H, 3 = conj.
 * The f64-MFMA GEMM every contraction of the truncation/encode path goes through (the `*` of
 * mps.jl:930,947; rsvd.jl:79,89,93,98,114); exported as a utility and test hook.                */
QIL_API int qil_gemm(qil_context* ctx, int dtype, int opA, int opB, int64_t m, int64_t n, int64_t k,
             const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc);

/* Thin QR with non-negative real diagonal (qr(...; positive=true), rsvd.jl:83,90,94) of a host operand
 * A (m x n, m >= n, column-major): Q (m x n), R (n x n).  Utility / test hook.                       */
QIL_API int qil_qr_positive(qil_context* ctx, int dtype, int64_t m, int64_t n, const void* A, void* Q, void* R);
/* ------------------------------------------------------------------ multi-GPU: the batched gather */
/* Independent (signal, damping value) items are dealt round-robin to one process per GPU (item i belongs to rank
 * i mod world); nothing is exchanged until the end, when every rank needs all coefficient batches: ONE RCCL all-gather
 * over xGMI.  The reference's callers loop serially over the damping values (docs/src/tutorials/zt.jl:300-348,
 * scripts/benchmark/zt_full_runtime.jl:151-221); this is the verb a Julia host uses in place of torch.distributed.
 * RCCL is loaded at run time by qil_comm_unique_id / qil_comm_create (QIL_RCCL_LIB, else the librccl.so next to the
 * process's libamdhip64, else the loader path): libqilhip.so itself links the HIP runtime only.
 *
 * Rendezvous: rank 0 calls qil_comm_unique_id and ships the QIL_COMM_ID_BYTES bytes to the other ranks over any host
 * channel (a file, Distributed.jl, MPI); then EVERY rank calls qil_comm_create (collective: returns when all have).   */
#define QIL_COMM_ID_BYTES 128
typedef struct qil_comm qil_comm;
QIL_API int qil_comm_unique_id(void* id_out);
QIL_API int qil_comm_create(qil_context* ctx, int rank, int world, const void* id, qil_comm** out);
/* In any order with qil_context_destroy: a context that goes first tears its communicators down and leaves their handles
 * valid and empty (a GC'd host -- Julia finalizers, Python at shutdown -- cannot promise an order).                 */
QIL_API int qil_comm_destroy(qil_comm* comm);
QIL_API int qil_comm_info(const qil_comm* comm, int* rank, int* world);
/* local: this rank's items in its own order (item rank, rank + world, ...), each `width` complex values (interleaved
 * doubles), host memory.  out: n_items x width complex values in ITEM order, host memory, on every rank.  Collective. */
QIL_API int qil_gather_coefficients(qil_comm* comm, int64_t n_items, int64_t width, const double* local, double* out);
/* The same gather for samples that are already in HBM (a sweep's read-outs start there): local_dev = this rank's
 * ceil-share x width complex values in slot order, out_dev = n_items x width in item order, both DEVICE memory of the
 * communicator's context; stream-ordered on that context's stream, no host synchronisation, no PCIe trip.  Every rank must
 * pass the same n_items and width (they size the collective).  A rank that fails before the collective aborts the
 * communicator (ncclCommAbort) so that its peers fail instead of waiting for ever.                                        */
QIL_API int qil_gather_coefficients_device(qil_comm* comm, int64_t n_items, int64_t width, const void* local_dev, void* out_dev);
/* The body of a damping sweep across the ranks of a communicator (docs/src/tutorials/dt.jl:150-197, zt.jl:300-348): Ws[0..nw)
 * = THIS rank's round-robin share of n_items operators (slot k = item rank + k world; nw must equal that share, else
 * QIL_EINVAL_LENGTH); every W psi is read out at the nb configurations (as qil_apply_coefficient_sweep), the samples stay in HBM,
 * ONE all-gather exchanges them, and out[n_items x nb] (item order, complex, host) is filled on every rank.  Collective.       */
QIL_API int qil_apply_coefficient_sweep_gather(qil_comm* comm, const qil_mpo* const* Ws, int64_t nw, const qil_mps* psi, int64_t nb,
                                       const uint8_t* bits, int64_t n_items, double* out);
/* The layout rule of that gather as a host function (no GPU, no RCCL): `gathered` = world blocks of
 * ceil(n_items / world) x width complex values in rank order -> `out` in item order.                                 */
QIL_API int qil_sweep_unshuffle(int world, int64_t n_items, int64_t width, const double* gathered, double* out);
/* ... and as the kernel the device gather uses (device buffers of `ctx`, stream-ordered): same rule, tested against the host one. */
QIL_API int qil_sweep_unshuffle_device(qil_context* ctx, int world, int64_t n_items, int64_t width, const void* gathered_dev, void* out_dev);

#ifdef __cplusplus
}
#endif
#endif /* QILAPLACE_HIP_H */
