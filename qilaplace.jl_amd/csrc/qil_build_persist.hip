// Persistent builder of the Damping-Transform MPO: ONE launch per sweep of damping values, one workgroup per
// damping value runs the whole chain of build_dt_mpo.
//
//   build_dt_mpo(n, wr; cutoff=1e-14, maxdim=1000)   src/transforms/dt_transformer.jl:312-407
//   zip_to_combine_mpos :20-164, zip_to_compress_mpo :167-288, gate blocks src/circuits/dt_gates.jl:30-229
//
// A DT build is ~4500 dependent factorisations of matrices no larger than (4 * 36) x 36 -- pure latency.  The
// launch-per-step builder of qil_build.hip pays several kernel launches (and, for every truncation, a host round
// trip) per step: 0.44 s for n = 24 whatever the batch size.  Here the chain never leaves the compute unit: the
// tensors being factorised live in LDS, the rest of the (KB-sized) chain in a per-value global workspace that stays
// in L2, the gate blocks are generated in place from a per-value table of the damping factors, and every
// truncation decision (sort, ITensors rule) is taken on the device.  Each damping value keeps its TRUE bond
// dimensions (no padding to a batch maximum).
//
// Factorisations: QR = Householder reflectors (one barrier per column, Q formed barrier-free with one column per
// 16-lane DPP row held in registers) -- unconditionally orthonormal, so exactly rank-deficient product bonds need no
// special casing; SVD = one-sided Jacobi on the short side in LDS (jacobi_sweeps_nov), as in the launch-per-step
// builder.  The gauge sweep of zip_to_compress skips the sites the preceding zip has just left isometric (the
// reference re-factorises them: Q R with R = identity to rounding).
//
// Only gauge-invariant results are comparable with the reference (dense operator, bond dimensions).
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "qil_internal.h"
#include "qil_device_utils.h"

namespace {
using namespace qil_dev;

constexpr int PB_NT = 256;                          // threads per workgroup (4 waves, one per SIMD)
constexpr int PB_NG = PB_NT / 16;                   // 16-lane DPP rows
constexpr int PB_MAXL = 256;                        // chain length 2n
constexpr int PB_DMAX = 56;                         // hard cap of any bond inside the kernel
constexpr int PB_NCYC = 13;                          // 8 cycle categories + SVD statistics

struct PbArgs {
    int n, L, dcap;
    long long site_cap;                 // doubles per stored site = dcap * 4 * dcap
    double cutoff;
    long long maxdim;
    const double* gates;                // per value: [1/sqrt2, e^{-w/2}/sqrt2, rfm[0..n], rfc[0..n-2]]  (2n + 2)
    double* ws;                         // per value: 2 chains x L x site_cap
    int* dims_out;                      // per value: L + 1 bond dims (with the two edges)
    int* status;                        // per value: 0 ok, else the capacity check that failed
    int arena_doubles;
    unsigned long long* cycles;         // optional: per value PB_NCYC counters
};

struct PbState {
    int bd[PB_MAXL + 1];                // bd[i] = left bond of site i, bd[len] = 1
    double blk[16];
    int blk_dl, blk_dr;
    double kap[PB_DMAX], dif[PB_DMAX], beta[PB_DMAX];
    double sig[PB_DMAX], inv[PB_DMAX], nrp[PB_DMAX];
    int perm[PB_DMAX], act[PB_DMAX];
    int rank, rot, rot2, err, nact;
    double red[8];
    unsigned long long cyc[PB_NCYC], t0;
};

#define PB_FAIL(code)                      \
    do {                                   \
        if (threadIdx.x == 0) st.err = (code); \
    } while (0)

__device__ __forceinline__ void pb_tick(PbState& st, int cat, bool prof) {
    if (prof && threadIdx.x == 0) {
        const unsigned long long t = __builtin_readcyclecounter();
        st.cyc[cat] += t - st.t0;
        st.t0 = t;
    }
}

// ------------------------------------------------------------------ gate blocks (src/circuits/dt_gates.jl)
// W[a + dl*(s_in + 2*(s_out + 2*b))] += g[s_in + 2*s_out]
__device__ __forceinline__ void pb_put(double* W, int dl, int a, int b, double g0, double g1, double g2, double g3) {
    W[a + dl * (0 + 2 * (0 + 2 * b))] += g0;
    W[a + dl * (1 + 2 * (0 + 2 * b))] += g1;
    W[a + dl * (0 + 2 * (1 + 2 * b))] += g2;
    W[a + dl * (1 + 2 * (1 + 2 * b))] += g3;
}

// control_damping_mpo(n, k, wr), tensor `site` of 2k (dt_gates.jl:30-130)
__device__ void pb_main_site(int k, const double* g, int site, int* dl, int* dr, double* W) {
    const double is2 = g[0], e2is2 = g[1];
    for (int t = 0; t < 16; ++t) W[t] = 0.0;
    if (k == 1) {
        *dl = *dr = 1;
        if (site == 0) pb_put(W, 1, 0, 0, is2, is2, is2, e2is2);
        else pb_put(W, 1, 0, 0, 1, 0, 0, 1);
        return;
    }
    const int pair = site / 2 + 1;
    const bool main = (site & 1) == 0;
    if (pair < k) {
        if (main) {
            const double rf = g[2 + (k + 1 - pair)];                 // exp(-w 2^(pair - k - 1))
            *dl = pair == 1 ? 1 : 2;
            *dr = 2;
            pb_put(W, *dl, 0, 0, 1, 0, 0, 1);
            pb_put(W, *dl, pair == 1 ? 0 : 1, 1, 1, 0, 0, rf);
        } else {
            *dl = *dr = 2;
            pb_put(W, 2, 0, 0, 1, 0, 0, 1);
            pb_put(W, 2, 1, 1, 1, 0, 0, 1);
        }
        return;
    }
    if (main) {
        *dl = *dr = 2;
        pb_put(W, 2, 0, 0, is2, 0, is2, 0);                          // input bit 0: Hd[0, s_out]
        pb_put(W, 2, 1, 1, 0, is2, 0, e2is2);                        // input bit 1: Hd[1, s_out]
    } else {
        *dl = 2;
        *dr = 1;
        pb_put(W, 2, 0, 0, 1, 0, 0, 1);
        pb_put(W, 2, 1, 0, 1, 0, 0, 1);
    }
}

// control_damping_copy_mpo(n, k, wr), tensor `site` of 2(n-k+1) (dt_gates.jl:133-229)
__device__ void pb_copy_site(int n, int k, const double* g, int site, int* dl, int* dr, double* W) {
    const int Lp = n - k + 1;
    for (int t = 0; t < 16; ++t) W[t] = 0.0;
    if (Lp == 1) {
        *dl = *dr = 1;
        pb_put(W, 1, 0, 0, 1, 0, 0, 1);
        return;
    }
    const int j = site / 2 + 1;
    const bool main = (site & 1) == 0;
    if (j == 1) {
        if (main) {
            *dl = 1;
            *dr = 2;
            pb_put(W, 1, 0, 0, 1, 0, 0, 1);
        } else {
            *dl = *dr = 2;
            pb_put(W, 2, 0, 0, 1, 0, 0, 0);
            pb_put(W, 2, 0, 1, 0, 0, 0, 1);
        }
        return;
    }
    if (main) {
        const double rf = g[2 + (n + 1) + (j - 2)];                  // exp(-w 2^(j - 2))
        *dl = *dr = 2;
        pb_put(W, 2, 0, 0, 1, 0, 0, 1);
        pb_put(W, 2, 1, 1, 1, 0, 0, rf);
    } else {
        const bool last = j == Lp;
        *dl = 2;
        *dr = last ? 1 : 2;
        pb_put(W, 2, 0, 0, 1, 0, 0, 1);
        pb_put(W, 2, 1, last ? 0 : 1, 1, 0, 0, 1);
    }
}

// block tensor t of the current gate block into st.blk; part 2 works in the mirrored frame (site order reversed,
// bond axes swapped).  One thread; the caller's next barrier publishes it.
__device__ void pb_block_site(PbState& st, int part, int n, int k, const double* g, int t, int nsites) {
    if (threadIdx.x != 0) return;
    int dl, dr;
    if (part == 1) {
        pb_main_site(k, g, t, &dl, &dr, st.blk);
        st.blk_dl = dl;
        st.blk_dr = dr;
    } else {
        double tmp[16];
        pb_copy_site(n, k, g, nsites - 1 - t, &dl, &dr, tmp);
        for (int a = 0; a < dl; ++a)
            for (int io = 0; io < 4; ++io)
                for (int b = 0; b < dr; ++b) st.blk[b + dr * (io + 4 * a)] = tmp[a + dl * (io + 4 * b)];
        st.blk_dl = dr;
        st.blk_dr = dl;
    }
}

// ------------------------------------------------------------------ Householder QR in LDS
// Everything here is latency-bound on LDS round trips (~128 cycles each): every loop first issues ALL its loads into
// registers (fixed unroll MU = rows per lane, predicated), then computes -- a `for (r ...) acc += x[r] * y[r]` loop
// with a run-time trip count pays one LDS latency per iteration (measured: 2400 cycles per Jacobi round instead of ~600).
//
// A (m x n, m >= n, LDS).  On exit the strict upper triangle of A holds R's, st.beta its diagonal, and column j
// from the diagonal down holds u_j = x_j - beta_j e_j (H_j = I - kap_j u_j u_j^T).  Every 16-lane row derives the
// reflector of column j redundantly (no broadcast step) and updates its share of the trailing columns; one
// barrier per column.
template <int MU>
__device__ void pb_hh_factor_t(double* A, int lda, int m, int n, PbState& st) {
    const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;
    for (int j = 0; j < n; ++j) {
        double* x = A + lda * j;
        double xr[MU];
        double s2 = 0;
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            const int r = l16 + 16 * u;
            xr[u] = (r > j && r < m) ? x[r] : 0.0;
        }
        const double alpha = x[j];
#pragma unroll
        for (int u = 0; u < MU; ++u) s2 = fma(xr[u], xr[u], s2);
        s2 = row16_sum(s2);
        double beta = alpha, diff = 0.0, kappa = 0.0;
        if (s2 != 0.0) {
            beta = -copysign(sqrt(fma(alpha, alpha, s2)), alpha);
            diff = alpha - beta;
            kappa = -1.0 / (beta * diff);
        }
        if (kappa != 0.0)
            for (int c0 = j + 1 + grp; c0 < n; c0 += 2 * PB_NG) {
                // two trailing columns per trip: both columns' loads are in flight together
                const int c1 = c0 + PB_NG;
                const bool two = c1 < n;
                double* y0 = A + lda * c0;
                double* y1 = A + lda * (two ? c1 : c0);
                double a0[MU], a1[MU];
#pragma unroll
                for (int u = 0; u < MU; ++u) {
                    const int r = l16 + 16 * u;
                    a0[u] = (r > j && r < m) ? y0[r] : 0.0;
                    a1[u] = (r > j && r < m) ? y1[r] : 0.0;
                }
                const double yj0 = y0[j], yj1 = y1[j];
                double w0 = 0, w1 = 0;
#pragma unroll
                for (int u = 0; u < MU; ++u) {
                    w0 = fma(xr[u], a0[u], w0);
                    w1 = fma(xr[u], a1[u], w1);
                }
                w0 = row16_sum(w0);
                w1 = row16_sum(w1);
                const double f0 = kappa * fma(diff, yj0, w0), f1 = kappa * fma(diff, yj1, w1);
#pragma unroll
                for (int u = 0; u < MU; ++u) {
                    const int r = l16 + 16 * u;
                    if (r > j && r < m) {
                        y0[r] = fma(-f0, xr[u], a0[u]);
                        if (two) y1[r] = fma(-f1, xr[u], a1[u]);
                    }
                }
                if (l16 == 0) {
                    y0[j] = fma(-f0, diff, yj0);
                    if (two) y1[j] = fma(-f1, diff, yj1);
                }
            }
        __syncthreads();
        if (tid == 0) {                       // nobody reads A[j, j] again before the Q formation
            st.kap[j] = kappa;
            st.dif[j] = diff;
            st.beta[j] = beta;
            x[j] = diff;
        }
    }
    __syncthreads();
}

__device__ void pb_hh_factor(double* A, int lda, int m, int n, PbState& st) {
    if (m <= 32) pb_hh_factor_t<2>(A, lda, m, n, st);
    else if (m <= 64) pb_hh_factor_t<4>(A, lda, m, n, st);
    else if (m <= 112) pb_hh_factor_t<7>(A, lda, m, n, st);
    else pb_hh_factor_t<13>(A, lda, m, n, st);
}

// R (n x n upper triangular, ldr) from a factored A
__device__ void pb_hh_copy_r(const double* A, int lda, int n, const PbState& st, double* R, int ldr) {
    for (int c = threadIdx.x >> 4; c < n; c += PB_NG)
        for (int i = threadIdx.x & 15; i < n; i += 16)
            R[i + ldr * c] = i < c ? A[i + lda * c] : (i == c ? st.beta[c] : 0.0);
}

// Q (m x n) = H_0 ... H_{n-1} [I_n; 0], one column per 16-lane row, the column held in registers while the
// reflectors c, c-1, ..., 0 are applied: no barriers.  Column c costs c + 1 applications; the 16 rows take the
// columns in snake order.  The next reflector is loaded while the current one is applied.  MU = ceil(m / 16) bound.
template <int MU>
__device__ void pb_hh_form_q(const double* A, int lda, int m, int n, const PbState& st, double* Q, int ldq) {
    const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;
    for (int pass = 0; pass * PB_NG < n; ++pass) {
        const int slot = (pass & 1) ? PB_NG - 1 - grp : grp;
        const int c = n - 1 - (pass * PB_NG + slot);
        if (c < 0) continue;
        double q[MU], xv[MU], xn[MU];
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            const int r = l16 + 16 * u;
            q[u] = (r == c) ? 1.0 : 0.0;
            xv[u] = (r >= c && r < m) ? A[lda * c + r] : 0.0;
        }
        double kappa = st.kap[c];
        for (int j = c; j >= 0; --j) {
            double kn = 0.0;
            if (j > 0) {
                const double* x = A + lda * (j - 1);
                kn = st.kap[j - 1];
#pragma unroll
                for (int u = 0; u < MU; ++u) {
                    const int r = l16 + 16 * u;
                    xn[u] = (r >= j - 1 && r < m) ? x[r] : 0.0;
                }
            }
            if (kappa != 0.0) {
                double wv = 0;
#pragma unroll
                for (int u = 0; u < MU; ++u) wv = fma(xv[u], q[u], wv);
                const double f = kappa * row16_sum(wv);
#pragma unroll
                for (int u = 0; u < MU; ++u) q[u] = fma(-f, xv[u], q[u]);
            }
            kappa = kn;
#pragma unroll
            for (int u = 0; u < MU; ++u) xv[u] = xn[u];
        }
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            const int r = l16 + 16 * u;
            if (r < m) Q[r + (long long)ldq * c] = q[u];
        }
    }
}

__device__ void pb_form_q(const double* A, int lda, int m, int n, const PbState& st, double* Q, int ldq) {
    if (m <= 32) pb_hh_form_q<2>(A, lda, m, n, st, Q, ldq);
    else if (m <= 64) pb_hh_form_q<4>(A, lda, m, n, st, Q, ldq);
    else if (m <= 112) pb_hh_form_q<7>(A, lda, m, n, st, Q, ldq);
    else pb_hh_form_q<13>(A, lda, m, n, st, Q, ldq);
}

// C (m x n, ldc) = A (m x k, lda) * B (k x n, ldb); any of them LDS or global; C must not alias A or B.  One work
// item = one column of C x 64 rows (4 per lane of a 16-lane row); the k loop runs in steps of 4 with the step's
// 20 loads issued together.  Optional per-column scale of C.
__device__ void pb_gemm(const double* __restrict__ A, int lda, const double* __restrict__ Bm, int ldb,
                        double* __restrict__ C, int ldc, int m, int n, int k) {
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int nrb = (m + 63) >> 6;
    int rb = 0, j = grp;
    while (j >= n && rb < nrb) {
        j -= n;
        ++rb;
    }
    while (rb < nrb) {
        const int i0 = rb * 64 + l16;
        int ro[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) ro[u] = min(i0 + 16 * u, m - 1);
        const double* bj = Bm + (long long)ldb * j;
        double acc[4] = {0, 0, 0, 0};
        int kk = 0;
        for (; kk + 4 <= k; kk += 4) {
            double bv[4], av[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bv[t] = bj[kk + t];
                const double* ak = A + (long long)lda * (kk + t);
#pragma unroll
                for (int u = 0; u < 4; ++u) av[t][u] = ak[ro[u]];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = fma(av[t][u], bv[t], acc[u]);
        }
        if (kk < k) {
            double bv[3], av[3][4];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int kt = min(kk + t, k - 1);
                bv[t] = (kk + t < k) ? bj[kt] : 0.0;
                const double* ak = A + (long long)lda * kt;
#pragma unroll
                for (int u = 0; u < 4; ++u) av[t][u] = ak[ro[u]];
            }
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = fma(av[t][u], bv[t], acc[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + 16 * u < m) C[(long long)ldc * j + i0 + 16 * u] = acc[u];
        j += PB_NG;
        while (j >= n && rb < nrb) {
            j -= n;
            ++rb;
        }
    }
}

__device__ __forceinline__ void pb_copy(const double* src, double* dst, int count) {
    for (int t = threadIdx.x; t < count; t += PB_NT) dst[t] = src[t];
}

// ------------------------------------------------------------------ one-sided Jacobi in LDS
template <int G>
__device__ __forceinline__ double pb_gsum(double v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    if (G == 16) v += dpp_mov<0x140>(v);  // row_mirror
    return v;
}

// Rotation orthogonalising a column pair with |x|^2 = al, |y|^2 = be, x.y = g, from two reciprocal square roots:
// cos 2t = |be - al| / h, h = sqrt((be - al)^2 + 4 g^2); c = sqrt((1 + cos 2t) / 2); s = |g| / (h c), signed by
// (be - al).  The chain sits on the critical path of every round of a lone wave, so the angle comes from the raw hardware
// reciprocal square roots (~1e-8 relative: it only decides how completely this pair is annihilated, quadratic convergence
// absorbs it) and unitarity is restored exactly: with eps = c0^2 + s0^2 - 1 both are scaled by
// 1 / sqrt(1 + eps) = 1 - eps / 2 + 3 eps^2 / 8 + O(eps^3 < 1e-18).  (No V is accumulated here: the factors are rebuilt
// from the normalised rotated columns.)  Returns false when the pair passes |g| <= tol |x| |y|.
__device__ __forceinline__ bool pb_rotation(double al, double be, double g, double tol, double& c, double& s, bool& big) {
    const double g2 = g * g, ab = al * be;
    big = g2 > (kQuadraticOff * kQuadraticOff) * ab;
    if (!(g2 > tol * tol * ab) || g2 == 0.0) return false;
    const double d = be - al;
    const double rh = __builtin_amdgcn_rsq(fma(d, d, 4.0 * g2));
    const double c2 = fma(0.5 * fabs(d), rh, 0.5);
    const double rc = __builtin_amdgcn_rsq(c2);
    const double c0 = c2 * rc;
    const double s0 = fabs(g) * rh * rc;
    const double eps = fma(c0, c0, fma(s0, s0, -1.0));
    const double f = fma(eps, fma(eps, 0.375, -0.5), 1.0);
    c = c0 * f;
    s = copysign(s0 * f, d);
    return true;
}

// Sweeps of the round-robin tournament over the columns of A (m x n, LDS) until no pair rotates.  Before every
// sweep the columns whose squared norm is below `negligible` (rounding residue of a rank-deficient operand: no
// direction to converge to, far below any cutoff) leave the tournament, so later sweeps run over the numerical
// rank only.  One pair per G-lane DPP row with both columns in registers (MU = rows per lane), one barrier per round.
template <int G, int MU>
__device__ void pb_jacobi_t(double* A, int lda, int m, int n, PbState& st, double tol, double negligible,
                            int max_sweeps, bool prof) {
    const int tid = threadIdx.x, lane = tid & (G - 1), grp = tid / G;
    constexpr int NW = PB_NT / G;
    int sweeps = 0, rounds = 0, nact = n;
    for (; sweeps < max_sweeps; ++sweeps) {
        for (int j = grp; j < n; j += NW) {
            const double* a = A + lda * j;
            double xs[MU];
#pragma unroll
            for (int u = 0; u < MU; ++u) {
                const int r = lane + G * u;
                xs[u] = r < m ? a[r] : 0.0;
            }
            double v = 0;
#pragma unroll
            for (int u = 0; u < MU; ++u) v = fma(xs[u], xs[u], v);
            v = pb_gsum<G>(v);
            if (lane == 0) st.sig[j] = v;
        }
        if (tid == 0) {
            st.rot = 0;
            st.rot2 = 0;
        }
        __syncthreads();
        if (tid < n) {
            // rank of this column among the kept ones; all norms read into registers first (one LDS latency, not n)
            const double mine = st.sig[tid];
            const bool keep = mine >= negligible && mine > 0.0;
            int pos = 0, tot = 0;
#pragma unroll
            for (int q = 0; q < PB_DMAX; ++q) {
                const double o = q < n ? st.sig[q] : 0.0;
                const bool kq = o >= negligible && o > 0.0;
                // (tournament positions in column order.  r05 measured de Rijk's ordering -- positions by decreasing norm -- on
                // this kernel: the slowest damping value went from 4.80 to 5.20 sweeps, 75 to 83 rounds per SVD, 64 values at
                // n = 24 from 157 to 184 ms; removed again, gpurun evidence profiles/r05_dt_norm_order.txt)
                pos += kq && q < tid;
                tot += kq;
            }
            if (keep) {
                st.act[pos] = tid;
                st.nrp[pos] = mine;                 // squared norm by tournament position: read together with act[]
            }
            if (tid == 0) st.nact = tot;
        }
        __syncthreads();
        nact = st.nact;
        if (nact < 2) break;
        const int npad = nact + (nact & 1);
        for (int round = 0; round < npad - 1; ++round) {
            for (int i = grp; i < npad / 2; i += NW) {
                int p, q;
                if (i == 0) {
                    p = npad - 1;
                    q = round;
                } else {
                    p = round + i;
                    q = round + npad - 1 - i;
                    if (p >= npad - 1) p -= npad - 1;
                    if (q >= npad - 1) q -= npad - 1;
                }
                if (p >= nact || q >= nact) continue;
                const int cp = st.act[p], cq = st.act[q];
                const double al = st.nrp[p], be = st.nrp[q];
                double* ap = A + lda * cp;
                double* aq = A + lda * cq;
                double x[MU], y[MU];
#pragma unroll
                for (int u = 0; u < MU; ++u) {
                    const int r = lane + G * u;
                    x[u] = r < m ? ap[r] : 0.0;
                    y[u] = r < m ? aq[r] : 0.0;
                }
                // squared norms are carried in st.nrp (exact at the start of every sweep, updated with each rotation):
                // one reduction per pair instead of three
                double g = 0;
#pragma unroll
                for (int u = 0; u < MU; ++u) g = fma(x[u], y[u], g);
                g = pb_gsum<G>(g);
                if (al < negligible || be < negligible) continue;
                double c, sn;
                bool big;
                if (!pb_rotation(al, be, g, tol, c, sn, big)) continue;
                if (lane == 0) {                      // plain stores of the same value from every rotating pair
                    st.rot = 1;
                    if (big) st.rot2 = 1;
                }
                const double sg = g >= 0 ? sn : -sn;
#pragma unroll
                for (int u = 0; u < MU; ++u) {
                    const int r = lane + G * u;
                    const double xn = fma(-sg, y[u], c * x[u]), yn = fma(sg, x[u], c * y[u]);
                    x[u] = xn;
                    y[u] = yn;
                    if (r < m) {
                        ap[r] = xn;
                        aq[r] = yn;
                    }
                }
                // |x'|^2 = c^2 al + s^2 be - 2 c s |g|, |y'|^2 = s^2 al + c^2 be + 2 c s |g|; after strong cancellation
                // the norm is taken from the rotated registers instead
                const double cs2 = 2.0 * c * sn * fabs(g), c2 = c * c, s2 = sn * sn;
                double aln = fma(c2, al, fma(s2, be, -cs2)), ben = fma(s2, al, fma(c2, be, cs2));
                if (aln < 0.25 * al || ben < 0.25 * be) {
                    double ea = 0, eb = 0;
#pragma unroll
                    for (int u = 0; u < MU; ++u) {
                        ea = fma(x[u], x[u], ea);
                        eb = fma(y[u], y[u], eb);
                    }
                    aln = pb_gsum<G>(ea);
                    ben = pb_gsum<G>(eb);
                }
                if (lane == 0) {
                    st.nrp[p] = aln;
                    st.nrp[q] = ben;
                }
            }
            __syncthreads();
        }
        rounds += npad - 1;
        const int any = st.rot2 ? 2 : 0;
        __syncthreads();
        if (!(any & 2)) {
            ++sweeps;
            break;
        }
    }
    // singular values = norms of the rotated columns
    for (int j = grp; j < n; j += NW) {
        const double* a = A + lda * j;
        double xs[MU];
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            const int r = lane + G * u;
            xs[u] = r < m ? a[r] : 0.0;
        }
        double v = 0;
#pragma unroll
        for (int u = 0; u < MU; ++u) v = fma(xs[u], xs[u], v);
        v = pb_gsum<G>(v);
        if (lane == 0) st.sig[j] = sqrt(v);
    }
    __syncthreads();
    if (prof && tid == 0) {
        st.cyc[8] += 1;
        st.cyc[9] += n;
        st.cyc[10] += sweeps;
        st.cyc[11] += rounds;
        st.cyc[12] += nact;
    }
}

__device__ void pb_jacobi(double* A, int lda, int m, int n, PbState& st, double tol, double negligible, int max_sweeps,
                          bool prof) {
    if (n > 32) {                                     // more than 16 pairs a round: 8-lane rows, 32 pairs at once
        if (m <= 80) pb_jacobi_t<8, 10>(A, lda, m, n, st, tol, negligible, max_sweeps, prof);
        else if (m <= 112) pb_jacobi_t<8, 14>(A, lda, m, n, st, tol, negligible, max_sweeps, prof);
        else pb_jacobi_t<8, 28>(A, lda, m, n, st, tol, negligible, max_sweeps, prof);
    } else {
        if (m <= 32) pb_jacobi_t<16, 2>(A, lda, m, n, st, tol, negligible, max_sweeps, prof);
        else if (m <= 64) pb_jacobi_t<16, 4>(A, lda, m, n, st, tol, negligible, max_sweeps, prof);
        else if (m <= 80) pb_jacobi_t<16, 5>(A, lda, m, n, st, tol, negligible, max_sweeps, prof);
        else if (m <= 112) pb_jacobi_t<16, 7>(A, lda, m, n, st, tol, negligible, max_sweeps, prof);
        else pb_jacobi_t<16, 14>(A, lda, m, n, st, tol, negligible, max_sweeps, prof);
    }
}

// ------------------------------------------------------------------ the chain
struct PbChain {
    double* base;            // current chain buffer (global)
    long long cap;
    __device__ double* site(int i) const { return base + (long long)i * cap; }
};

// zip_to_combine "down": the block acts after M on sites 0 .. L2-1 (dt_transformer.jl:38-95), QR remainder
// carried to the right.  Sites 0 .. L2-1 come out isometric; the remainder is absorbed into site L2 (or L2-1).
__device__ void pb_zip(PbState& st, const PbArgs& a, PbChain ch, double* arena, int part, int k, const double* g,
                       int L2, int len, bool prof) {
    const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;
    const int TS = a.dcap * a.dcap;
    double* T = arena;                                         // remainder [R, Da, Dc]; the new one replaces it in place
    double* Wk = arena + TS;
    const int WS = a.arena_doubles - TS;
    if (tid == 0) T[0] = 1.0;
    int R = 1, Da = 1, Dc = 1;
    for (int t = 0; t < L2; ++t) {
        pb_block_site(st, part, a.n, k, g, t, L2);
        const int B1 = st.bd[t + 1];
        const int E = 4 * B1;
        __syncthreads();                                       // block tensor and T visible
        const int B2 = st.blk_dr;
        const int rows = 4 * R, cols = B1 * B2;
        const int nb = rows > cols ? cols : rows;
        // LDS plan.  all = 1: the old site (Ms), X for every block-bond value c and the core fit together.
        // all = 0 (bonds beyond ~20): X one c at a time, the old site read from the workspace (L2).
        const bool all = Da * E + R * E * Dc + rows * cols <= WS;
        if ((!all && R * E + rows * cols > WS) || nb * cols > TS || R * Da * Dc > TS || nb > a.dcap || rows > 16 * 13) {
            PB_FAIL(1);
            __syncthreads();
            return;
        }
        const double* Ms = ch.site(t);                         // Da x 4 x B1
        double* X = Wk;
        if (all) {
            pb_copy(ch.site(t), Wk, Da * E);
            Ms = Wk;
            X = Wk + Da * E;
            __syncthreads();
        }
        double* A = X + R * E * (all ? Dc : 1);                // rows x cols
        for (int c0 = 0; c0 < Dc; c0 += all ? Dc : 1) {
            const int c1 = all ? Dc : c0 + 1;
            // stage 1: X[r, e, c] = sum_a T[r, a, c] Ms[a, e]
            for (int c = c0; c < c1; ++c) pb_gemm(T + R * Da * c, R, Ms, Da, X + R * E * (c - c0), R, R, E, Da);
            __syncthreads();
            // stage 2: core[(r,i,o), (b1,b2)] (+)= sum_{c,m} X[r, (i,m,b1), c] Bk[c, m, o, b2]        (:54-61)
            for (int p = grp; p < 4 * cols; p += PB_NG) {
                const int i = p & 1, o = (p >> 1) & 1, col = p >> 2;
                const int b1 = col % B1, b2 = col / B1;
                for (int r0 = 0; r0 < R; r0 += 64) {
                    double acc[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = r0 + l16 + 16 * u;
                        acc[u] = (c0 > 0 && r < R) ? A[(r + R * (i + 2 * o)) + rows * col] : 0.0;
                    }
                    for (int c = c0; c < c1; ++c)
                        for (int m = 0; m < 2; ++m) {
                            const double bv = st.blk[c + Dc * (m + 2 * (o + 2 * b2))];
                            if (bv == 0.0) continue;
                            const double* xc = X + R * ((i + 2 * (m + 2 * b1)) + E * (c - c0));
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int r = r0 + l16 + 16 * u;
                                if (r < R) acc[u] = fma(xc[r], bv, acc[u]);
                            }
                        }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = r0 + l16 + 16 * u;
                        if (r < R) A[(r + R * (i + 2 * o)) + rows * col] = acc[u];
                    }
                }
            }
            __syncthreads();
        }
        pb_tick(st, 0, prof);
        double* dst = ch.site(t);
        if (rows > cols) {
            pb_hh_factor(A, rows, rows, cols, st);
            pb_hh_copy_r(A, rows, cols, st, T, cols);
            pb_form_q(A, rows, rows, cols, st, dst, rows);
        } else {
            for (int e = tid; e < rows * rows; e += PB_NT) dst[e] = (e % rows) == (e / rows) ? 1.0 : 0.0;
            pb_copy(A, T, rows * cols);
        }
        if (tid == 0) {
            st.bd[t] = R;
            st.bd[t + 1] = nb;
        }
        __syncthreads();
        pb_tick(st, 1, prof);
        R = nb;
        Da = B1;
        Dc = B2;
    }
    // the block has ended (Dc == 1): T is R x Da
    if (len > L2) {
        const int dr = st.bd[L2 + 1];
        if (Da * 4 * dr > WS || R > a.dcap) {
            PB_FAIL(2);
            __syncthreads();
            return;
        }
        pb_copy(ch.site(L2), Wk, Da * 4 * dr);
        __syncthreads();
        pb_gemm(T, R, Wk, Da, ch.site(L2), R, R, 4 * dr, Da);
        if (tid == 0) st.bd[L2] = R;
    } else {
        const int dl = st.bd[L2 - 1];
        if (dl * 4 * R > WS) {
            PB_FAIL(3);
            __syncthreads();
            return;
        }
        pb_copy(ch.site(L2 - 1), Wk, dl * 4 * R);
        __syncthreads();
        pb_gemm(Wk, dl * 4, T, R, ch.site(L2 - 1), dl * 4, dl * 4, Da, R);
        if (tid == 0) st.bd[L2] = Da;
    }
    __syncthreads();
    pb_tick(st, 2, prof);
}

// gauge sweep of zip_to_compress "down" over sites first .. len-2 (:186-203)
__device__ void pb_gauge(PbState& st, const PbArgs& a, PbChain ch, double* arena, int first, int len, bool prof) {
    const int tid = threadIdx.x;
    const int TS = a.dcap * a.dcap;
    double* Rm = arena;
    double* Wk = arena + TS;
    const int WS = a.arena_doubles - TS;
    for (int i = first; i + 1 < len; ++i) {
        const int m = st.bd[i] * 4, n = st.bd[i + 1], w2 = 4 * st.bd[i + 2];
        double* A = Wk;                    // m x n
        double* Nx = A + m * n;            // n x w2
        if (m * n + n * w2 > WS || n * n > TS || m > 16 * 13) {
            PB_FAIL(4);
            __syncthreads();
            return;
        }
        pb_copy(ch.site(i), A, m * n);
        pb_copy(ch.site(i + 1), Nx, n * w2);
        __syncthreads();
        if (m > n) {
            pb_hh_factor(A, m, m, n, st);
            pb_hh_copy_r(A, m, n, st, Rm, n);
            pb_form_q(A, m, m, n, st, ch.site(i), m);
            __syncthreads();
            pb_gemm(Rm, n, Nx, n, ch.site(i + 1), n, n, w2, n);
        } else {
            // fat site: Q = I_m (an isometry), R = the site itself; the bond shrinks to m
            double* dst = ch.site(i);
            for (int e = tid; e < m * m; e += PB_NT) dst[e] = (e % m) == (e / m) ? 1.0 : 0.0;
            pb_gemm(A, m, Nx, n, ch.site(i + 1), m, m, w2, n);
            if (tid == 0) st.bd[i + 1] = m;
        }
        __syncthreads();
        pb_tick(st, 3, prof);
    }
}

// truncating sweep of zip_to_compress "down" (:207-229).  The reference factorises the two-site core
// M[i-1] M[i]; with everything left of the bond isometric that core has the singular values and right
// singular vectors of the single tensor M[i] viewed as (bond | in, out, right bond): 4x smaller.
__device__ void pb_truncate(PbState& st, const PbArgs& a, PbChain ch, double* arena, int len, bool prof) {
    const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4, lane = tid & 63, wave = tid >> 6;
    for (int i = len - 1; i >= 1; --i) {
        const int d = st.bd[i], w = 4 * st.bd[i + 1], dl4 = 4 * st.bd[i - 1];
        const bool tall = d > w;
        const int cols = tall ? w : d, rows = tall ? d : w;
        const int ldw = rows | 1;
        // LDS plan: [Wk | Mo | VT] from the front, US at the very end; once US is built the front is dead and takes
        // the copy of M[i-1] for the last product
        double* Wk = arena;                                      // rows x cols, rotated in place
        double* Mo = Wk + ldw * cols;                            // d x w copy of M[i]
        double* VT = Mo + d * w;                                 // w x cols: Vh^T (non-tall)
        double* US = arena + a.arena_doubles - d * cols;         // d x cols
        double* Mp = arena;                                      // dl4 x d copy of M[i-1]
        if (ldw * cols + d * w + (tall ? 0 : w * cols) + d * cols > a.arena_doubles ||
            dl4 * d + d * cols > a.arena_doubles || cols > PB_DMAX) {
            PB_FAIL(5);
            __syncthreads();
            return;
        }
        const double* src = ch.site(i);
        double f = 0;
        for (int t = tid; t < d * w; t += PB_NT) {
            const double v = src[t];
            Mo[t] = v;
            const int r = t % d, c = t / d;
            if (tall) Wk[r + ldw * c] = v;
            else Wk[c + ldw * r] = v;
            f = fma(v, v, f);
        }
        f = wave_sum(f);
        if (lane == 0) st.red[wave] = f;
        __syncthreads();
        f = (st.red[0] + st.red[1]) + (st.red[2] + st.red[3]);
        pb_tick(st, 4, prof);
        pb_jacobi(Wk, ldw, rows, cols, st, 1e-15, 1e-30 * f, 40, prof);
        pb_tick(st, 5, prof);
        // stable descending order + the ITensors truncation rule (qil_truncation_rank).  Every column's thread finds its own sorted
        // position and leaves there its index, its square and its reciprocal (r04: the rule below used to re-read all 56 possible
        // entries through the permutation, three fully unrolled passes on one lane: 12.7 k cycles per SVD, MEASUREMENTS R04.11)
        if (tid < cols) {
            const double s = st.sig[tid];
            int pos = 0;
#pragma unroll
            for (int q = 0; q < PB_DMAX; ++q) {
                const double o = q < cols ? st.sig[q] : -1.0;
                pos += (o > s) || (o == s && q < tid);
            }
            st.perm[pos] = tid;
            st.nrp[pos] = s * s;
            st.inv[pos] = s > 0.0 ? 1.0 / s : 0.0;
        }
        __syncthreads();
        if (tid == 0) {
            // the rule on the sorted squares, sums in the host's order (ascending for the total, from the tail for the discarded
            // weight), eight entries loaded per trip and only as many trips as there are columns
            int kk = cols;
            if (!(st.nrp[0] > 0.0) || cols == 1) kk = 1;
            else {
                double terr = 0.0, scale = 0.0;
                for (int q0 = 0; q0 < cols; q0 += 8) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = q0 + u < cols ? st.nrp[q0 + u] : 0.0;
#pragma unroll
                    for (int u = 0; u < 8; ++u) scale += v[u];            // (+ 0.0 beyond the last column: exact)
                }
                if (scale == 0.0) scale = 1.0;
                const double lim = a.cutoff * scale;
                bool open = true;
                for (int q0 = ((cols - 1) >> 3) << 3; q0 >= 0 && open; q0 -= 8) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = q0 + u < cols ? st.nrp[q0 + u] : 0.0;
#pragma unroll
                    for (int u = 7; u >= 0; --u) {
                        const int q = q0 + u;
                        if (q >= 1 && q < cols && open) {
                            if ((long long)(q + 1) > a.maxdim || terr + v[u] <= lim) {
                                terr += v[u];
                                kk = q;
                            } else open = false;
                        }
                    }
                }
                if (kk < 1) kk = 1;
            }
            st.rank = kk;
        }
        __syncthreads();
        const int rk = st.rank;
        double* dsti = ch.site(i);                  // new M[i] = Vh (rk x w)
        if (!tall) {
            // M[i]^T = (Wk D^-1) D V^T : Vh = (Wk[:, perm] D^-1)^T ; U S = M[i] Vh^T
            for (int j = grp; j < rk; j += PB_NG) {
                const double* wj = Wk + ldw * st.perm[j];
                const double sc = st.inv[j];
                for (int c = l16; c < w; c += 16) VT[c + w * j] = wj[c] * sc;
            }
            __syncthreads();
            for (int t = tid; t < rk * w; t += PB_NT) {
                const int j = t % rk, c = t / rk;
                dsti[t] = VT[c + w * j];
            }
            pb_gemm(Mo, d, VT, w, US, d, d, rk, w);
        } else {
            // M[i] = (Wk D^-1) D V^T, Wk = M[i] rotated: U S = Wk[:, perm] ; Vh = D^-2 (U S)^T M[i]
            for (int t = tid; t < d * rk; t += PB_NT) {
                const int r = t % d, j = t / d;
                US[t] = Wk[r + ldw * st.perm[j]];
            }
            __syncthreads();
            for (int t = tid; t < rk * w; t += PB_NT) {
                const int j = t % rk, c = t / rk;
                const double* uj = US + d * j;
                const double* mc = Mo + d * c;
                double acc = 0;
                for (int r = 0; r < d; ++r) acc = fma(uj[r], mc[r], acc);
                dsti[t] = acc * st.inv[j] * st.inv[j];
            }
        }
        __syncthreads();
        pb_copy(ch.site(i - 1), Mp, dl4 * d);
        __syncthreads();
        pb_gemm(Mp, dl4, US, d, ch.site(i - 1), dl4, dl4, rk, d);           // M[i-1] <- M[i-1] (U S)
        if (tid == 0) st.bd[i] = rk;
        __syncthreads();
        pb_tick(st, 6, prof);
    }
}

// reverse the site order and swap the bond axes of every tensor: out[b, io, a] = in[a, io, b]
__device__ void pb_mirror(PbState& st, PbChain src, PbChain dst, int len, int* tmp_bd) {
    const int tid = threadIdx.x;
    for (int i = 0; i < len; ++i) {
        const int Da = st.bd[i], Db = st.bd[i + 1];
        const double* s = src.site(i);
        double* o = dst.site(len - 1 - i);
        for (int t = tid; t < Da * 4 * Db; t += PB_NT) {
            const int b = t % Db, io = (t / Db) & 3, aa = t / (4 * Db);
            o[t] = s[aa + Da * (io + 4 * b)];
        }
    }
    __syncthreads();
    for (int i = tid; i <= len; i += PB_NT) tmp_bd[i] = st.bd[len - i];
    __syncthreads();
    for (int i = tid; i <= len; i += PB_NT) st.bd[i] = tmp_bd[i];
    __syncthreads();
}

__global__ __launch_bounds__(PB_NT) void dt_build_persistent(PbArgs a) {
    extern __shared__ __attribute__((aligned(16))) double pb_arena[];
    __shared__ PbState st;
    __shared__ int tmp_bd[PB_MAXL + 1];
    const int tid = threadIdx.x, b = blockIdx.x;
    const bool prof = a.cycles != nullptr;
    const double* g = a.gates + (long long)b * (2 * a.n + 2);
    PbChain ch{a.ws + (long long)b * 2 * a.L * a.site_cap, a.site_cap};
    PbChain alt{ch.base + (long long)a.L * a.site_cap, a.site_cap};
    if (tid == 0) {
        st.err = 0;
        for (int c = 0; c < PB_NCYC; ++c) st.cyc[c] = 0;
        st.t0 = __builtin_readcyclecounter();
        st.bd[0] = st.bd[1] = st.bd[2] = 1;
        int dl, dr;
        pb_main_site(1, g, 0, &dl, &dr, st.blk);
        for (int t = 0; t < 4; ++t) ch.site(0)[t] = st.blk[t];
        pb_main_site(1, g, 1, &dl, &dr, st.blk);
        for (int t = 0; t < 4; ++t) ch.site(1)[t] = st.blk[t];
    }
    __syncthreads();
    int len = 2;
    for (int k = 2; k <= a.n && st.err == 0; ++k) {                          // part 1 (:351-390)
        if (tid < 2) {
            double* s = ch.site(len + tid);
            s[0] = 1.0;
            s[1] = 0.0;
            s[2] = 0.0;
            s[3] = 1.0;
            st.bd[len + 1 + tid] = 1;
        }
        len += 2;
        __syncthreads();
        pb_zip(st, a, ch, pb_arena, 1, k, g, len, len, prof);
        if (st.err) break;
        pb_truncate(st, a, ch, pb_arena, len, prof);                          // every site is isometric after the zip
    }
    if (a.n > 1 && st.err == 0) {                                             // part 2, mirrored frame (:396-405)
        pb_mirror(st, ch, alt, len, tmp_bd);
        for (int k = 1; k < a.n && st.err == 0; ++k) {
            const int ns = 2 * (a.n - k + 1);
            pb_zip(st, a, alt, pb_arena, 2, k, g, ns, len, prof);
            if (st.err) break;
            pb_gauge(st, a, alt, pb_arena, ns, len, prof);
            if (st.err) break;
            pb_truncate(st, a, alt, pb_arena, len, prof);
        }
        if (st.err == 0) pb_mirror(st, alt, ch, len, tmp_bd);
    }
    pb_tick(st, 7, prof);
    __syncthreads();
    for (int i = tid; i <= a.L; i += PB_NT) a.dims_out[(long long)b * (a.L + 1) + i] = st.bd[i];
    if (tid == 0) a.status[b] = st.err;
    if (prof && tid < PB_NCYC) a.cycles[(long long)b * PB_NCYC + tid] = st.cyc[tid];
}

// dst[b * L + i][0 .. dl*4*dr) = final chain site i of value b
__global__ void pb_copy_out(const double* __restrict__ ws, long long site_cap, int L, const int* __restrict__ dims,
                            double* const* __restrict__ dst) {
    const int i = blockIdx.x, b = blockIdx.y;
    const int* bd = dims + (long long)b * (L + 1);
    const int count = bd[i] * 4 * bd[i + 1];
    const double* s = ws + ((long long)b * 2 * L + i) * site_cap;
    double* o = dst[(long long)b * L + i];
    for (int t = threadIdx.x; t < count; t += blockDim.x) o[t] = s[t];
}

}  // namespace

static int persist_chunk(qil_context* ctx, int64_t n, int64_t nb, const double* wrs, double cutoff, int64_t maxdim,
                         const int64_t* site_ids, qil_mpo** out, int* fallback);

// Returns QIL_OK with *fallback = 1 when some value exceeded the in-LDS capacities (caller takes the
// launch-per-step route); out[] holds no handles in that case.  Large sweeps go through in chunks so that the
// per-value workspace (2 chains x 2n sites x 86 KB) stays within ~8 GiB.
int qil_build_dt_persistent(qil_context* ctx, int64_t n, int64_t nb, const double* wrs, double cutoff, int64_t maxdim,
                            const int64_t* site_ids, qil_mpo** out, int* fallback) {
    *fallback = 0;
    const int64_t per_value = 2 * (2 * n) * (52LL * 4 * 52) * (int64_t)sizeof(double);
    const int64_t chunk = std::max<int64_t>(256, (8LL << 30) / std::max<int64_t>(per_value, 1));
    for (int64_t off = 0; off < nb; off += chunk) {
        const int64_t cnt = std::min(chunk, nb - off);
        int st = persist_chunk(ctx, n, cnt, wrs + off, cutoff, maxdim, site_ids, out + off, fallback);
        if (st != QIL_OK || *fallback) {
            for (int64_t b = 0; b < off; ++b) {
                qil_mpo_destroy(out[b]);
                out[b] = nullptr;
            }
            return st;
        }
    }
    return QIL_OK;
}

static int persist_chunk(qil_context* ctx, int64_t n, int64_t nb, const double* wrs, double cutoff, int64_t maxdim,
                         const int64_t* site_ids, qil_mpo** out, int* fallback) {
    *fallback = 0;
    const int L = (int)(2 * n);
    if (L > PB_MAXL) {
        *fallback = 1;
        return QIL_OK;
    }
    const int B = (int)nb;
    int dcap = 52;                                    // truncated bonds up to 26 (damping 0.25 at n = 24: 25)
    if (const char* e = getenv("QIL_DT_DCAP")) dcap = std::max(8, std::min(PB_DMAX, atoi(e)));
    const long long site_cap = (long long)dcap * 4 * dcap;
    const int arena_doubles = (160 * 1024 - 8 * 1024) / 8;
    static qil_lds_grant grant;                                  // per device; a refusal sends the caller to the launch-per-step builder
    if (grant.ensure(ctx->device, reinterpret_cast<const void*>(&dt_build_persistent), (size_t)arena_doubles * 8) != hipSuccess) {
        (void)hipGetLastError();
        *fallback = 1;
        return QIL_OK;
    }
    const int gstride = 2 * (int)n + 2;
    std::vector<double> gates((size_t)B * gstride, 0.0);
    for (int b = 0; b < B; ++b) {
        double* g = gates.data() + (size_t)b * gstride;
        const double w = wrs[b], is2 = 1.0 / std::sqrt(2.0);
        g[0] = is2;
        g[1] = std::exp(-w / 2.0) * is2;
        for (int idx = 0; idx <= n; ++idx) g[2 + idx] = std::exp(-w * std::pow(2.0, -idx));
        for (int e = 0; e + 1 < n; ++e) g[2 + (n + 1) + e] = std::exp(-w * std::pow(2.0, e));
    }
    void *dg = nullptr, *ws = nullptr, *dmeta = nullptr, *dcyc = nullptr;
    const size_t meta_ints = (size_t)B * (L + 1) + (size_t)B;
    QIL_TRY(qil_ctx_alloc(ctx, gates.size() * sizeof(double), &dg));
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)B * 2 * L * site_cap * sizeof(double), &ws));
    QIL_TRY(qil_ctx_alloc(ctx, meta_ints * sizeof(int), &dmeta));
    const bool prof = getenv("QIL_DT_PROFILE") != nullptr;
    if (prof) {
        QIL_TRY(qil_ctx_alloc(ctx, (size_t)B * PB_NCYC * sizeof(unsigned long long), &dcyc));
    }
    QIL_HIP(hipMemcpyAsync(dg, gates.data(), gates.size() * sizeof(double), hipMemcpyHostToDevice, qil_stream(ctx)));
    PbArgs a;
    a.n = (int)n;
    a.L = L;
    a.dcap = dcap;
    a.site_cap = site_cap;
    // The tournament leaves columns below 1e-30 of the operand's weight alone (rounding residue of exactly rank-deficient product
    // bonds: not directions, hence not orthogonalised); a cutoff below that would KEEP them as columns of an "isometry" that is none
    // (found r06: cutoff = 0 gave wrong operators).  The rule therefore never runs below 1e-28: what it drops beyond the caller's
    // cutoff carries < 1e-28 of the weight, which no gauge-invariant quantity sees (same rule as the complex chain builder's zip).
    a.cutoff = std::max(cutoff, 1e-28);
    a.maxdim = maxdim <= 0 ? INT64_MAX : maxdim;
    a.gates = static_cast<const double*>(dg);
    a.ws = static_cast<double*>(ws);
    a.dims_out = static_cast<int*>(dmeta);
    a.status = a.dims_out + (size_t)B * (L + 1);
    a.arena_doubles = arena_doubles;
    a.cycles = static_cast<unsigned long long*>(dcyc);
    if (prof) QIL_TRY(qil_timer_start(ctx));
    hipLaunchKernelGGL(dt_build_persistent, dim3(B), dim3(PB_NT), (size_t)arena_doubles * 8, qil_stream(ctx), a);
    QIL_HIP(hipGetLastError());
    double kernel_ms = 0.0;
    if (prof) QIL_TRY(qil_timer_stop(ctx, &kernel_ms));
    std::vector<int> meta(meta_ints);
    QIL_HIP(hipMemcpyAsync(meta.data(), dmeta, meta_ints * sizeof(int), hipMemcpyDeviceToHost, qil_stream(ctx)));
    QIL_HIP(qil_stream_sync(ctx));           // also orders the pageable `gates` upload before its release
    if (prof) {
        std::vector<unsigned long long> cyc((size_t)B * PB_NCYC);
        QIL_HIP(hipMemcpy(cyc.data(), dcyc, cyc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        static const char* names[8] = {"zip-core", "zip-qr", "zip-absorb", "gauge-qr", "svd-load", "svd-jacobi",
                                             "svd-post", "other"};
        unsigned long long tot = 0;
        for (int c = 0; c < 8; ++c) tot += cyc[c];
        int nfail = 0;
        for (int b = 0; b < B; ++b) nfail += meta[(size_t)B * (L + 1) + b] != 0;
        fprintf(stderr, "[qil dt persistent] kernel %.3f ms, %d of %d values over capacity (code of value 0: %d); value 0, "
                "n=%lld: %.3f Mcycles:", kernel_ms, nfail, B, meta[(size_t)B * (L + 1)], (long long)n, tot * 1e-6);
        for (int c = 0; c < 8; ++c) fprintf(stderr, " %s %.1f%%", names[c], 100.0 * cyc[c] / std::max<double>(tot, 1));
        const double ns = std::max<double>((double)cyc[8], 1.0);
        fprintf(stderr, "; %llu SVDs, mean columns %.1f, sweeps %.2f, rounds %.1f, columns left in the last sweep %.1f\n",
                cyc[8], cyc[9] / ns, cyc[10] / ns, cyc[11] / ns, cyc[12] / ns);
        // per-value totals: is the launch as long as its slowest chain, or do all chains slow down together?
        unsigned long long tmin = ~0ull, tmax = 0;
        int bmax = 0;
        double tsum = 0;
        for (int b = 0; b < B; ++b) {
            unsigned long long tb = 0;
            for (int c = 0; c < 8; ++c) tb += cyc[(size_t)b * PB_NCYC + c];
            tsum += (double)tb;
            if (tb < tmin) tmin = tb;
            if (tb > tmax) {
                tmax = tb;
                bmax = b;
            }
        }
        fprintf(stderr, "[qil dt persistent] cycles per value: min %.1f M, mean %.1f M, max %.1f M (value %d, wr = %.3f); kernel time x 2.4 GHz = %.1f M; slowest value:",
                tmin * 1e-6, tsum / B * 1e-6, tmax * 1e-6, bmax, wrs[bmax], kernel_ms * 2.4);
        {
            const unsigned long long* cb = cyc.data() + (size_t)bmax * PB_NCYC;
            for (int c = 0; c < 8; ++c) fprintf(stderr, " %s %.1f%%", names[c], 100.0 * cb[c] / std::max<double>((double)tmax, 1));
            const double nsb = std::max<double>((double)cb[8], 1.0);
            fprintf(stderr, "; %llu SVDs, mean columns %.1f, sweeps %.2f, rounds %.1f (%.0f cycles per round)\n", cb[8], cb[9] / nsb, cb[10] / nsb, cb[11] / nsb,
                    (double)cb[5] / std::max<double>((double)cb[11], 1.0));
        }
        qil_ctx_free(ctx, dcyc);
    }
    const int* status = meta.data() + (size_t)B * (L + 1);
    bool over = false;
    for (int b = 0; b < B; ++b) over = over || status[b] != 0;
    if (over) {
        qil_ctx_free(ctx, dg);
        qil_ctx_free(ctx, ws);
        qil_ctx_free(ctx, dmeta);
        *fallback = 1;
        return QIL_OK;
    }
    // hand out one PairedSiteMPO per damping value with its own bond dimensions
    std::vector<double*> ptrs((size_t)B * L);
    std::vector<int64_t> bonds((size_t)std::max(L - 1, 1));
    int st = QIL_OK;
    int made = 0;
    for (int b = 0; b < B && st == QIL_OK; ++b) {
        const int* bd = meta.data() + (size_t)b * (L + 1);
        for (int i = 0; i + 1 < L; ++i) bonds[(size_t)i] = bd[i + 1];
        qil_mpo* W = nullptr;
        st = qil_mpo_alloc(ctx, L, QIL_F64, 1, bonds.data(), site_ids, &W);
        if (st != QIL_OK) break;
        out[b] = W;
        ++made;
        for (int i = 0; i < L; ++i) ptrs[(size_t)b * L + i] = static_cast<double*>(W->site[(size_t)i]);
    }
    void* dptr = nullptr;
    if (st == QIL_OK) st = qil_ctx_alloc(ctx, ptrs.size() * sizeof(double*), &dptr);
    if (st == QIL_OK) {
        hipError_t e = hipMemcpyAsync(dptr, ptrs.data(), ptrs.size() * sizeof(double*), hipMemcpyHostToDevice, qil_stream(ctx));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(pb_copy_out, dim3(L, B), dim3(256), 0, qil_stream(ctx), (const double*)ws, site_cap, L,
                               (const int*)dmeta, (double* const*)dptr);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = qil_stream_sync(ctx);
        if (e != hipSuccess) st = qil_fail(QIL_EHIP, "build_dt_mpo: copy-out failed: %s", hipGetErrorString(e));
    }
    if (dptr) qil_ctx_free(ctx, dptr);
    qil_ctx_free(ctx, dg);
    qil_ctx_free(ctx, ws);
    qil_ctx_free(ctx, dmeta);
    if (st != QIL_OK) {
        for (int b = 0; b < made; ++b) {
            qil_mpo_destroy(out[b]);
            out[b] = nullptr;
        }
    }
    return st;
}
