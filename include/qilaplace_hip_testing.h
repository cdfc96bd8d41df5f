/*
 * qilaplace_hip_testing.h -- test, fault-injection and measurement hooks of libqilhip.so.
 *
 * NOT part of the drop-in boundary: nothing here replaces a method of the reference.  The parity suite
 * (tests/), bench.py and tools/ use these entries; a Julia / C client of the boundary never needs them.  They are exported
 * by the same library so that the shipped binary is the one that is tested and measured.
 */
#ifndef QILAPLACE_HIP_TESTING_H
#define QILAPLACE_HIP_TESTING_H

#include "qilaplace_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Testing aid: the n-th pool allocation from now (0 = the next one) fails with QIL_ENOMEM; n < 0 switches the
 * injection off.  Used to check that a failing call leaves no device memory behind and its operands intact. */
QIL_API int qil_context_fail_alloc_after(qil_context* ctx, int64_t n);
/* Testing aid: pool bytes in use that no MPS/MPO handle owns.  Zero between calls -- every temporary is back in
 * the pool whether the last call succeeded or failed.                                                        */
QIL_API int qil_context_unowned_bytes(qil_context* ctx, int64_t* out);

/* HIP-event timing on the context's stream (hipEventRecord / hipEventElapsedTime). */
QIL_API int qil_timer_start(qil_context* ctx);
QIL_API int qil_timer_stop(qil_context* ctx, double* elapsed_ms);   /* synchronises the stop event */
/* Per-kernel profile: when enabled every launch of the site-contraction kernel is
 * bracketed by its own event pair; read returns launches and summed device ms since
 * the last reset (synchronises).                                                   */
QIL_API int qil_profile_enable(qil_context* ctx, int on);
QIL_API int qil_profile_read(qil_context* ctx, int64_t* n_launches, double* total_ms, int reset);

/* Diagnostic: the store-only HBM ceiling of THIS GPU -- `bytes` (>= 64 MiB) written `reps` times by each of four writers
 * (hipMemsetAsync, 256 KiB span per workgroup with plain / non-temporal stores, grid-stride fill), HIP events; the best rate
 * in GB/s and which writer reached it (0..3).  bench.py prints it beside the apply's roofline fraction so that lines measured
 * on different boxes of a pool can be compared.                                                                         */
QIL_API int qil_hbm_store_peak(qil_context* ctx, int64_t bytes, int reps, double* best_gbs, int* best_kind);

/* Diagnostic: device-resident time of the same GEMM (operands generated in HBM, HIP events). */
QIL_API int qil_gemm_device_time(qil_context* ctx, int dtype, int opA, int opB, int64_t m, int64_t n, int64_t k,
                         int reps, double* ms_per_call);

#ifdef __cplusplus
}
#endif
#endif /* QILAPLACE_HIP_TESTING_H */
