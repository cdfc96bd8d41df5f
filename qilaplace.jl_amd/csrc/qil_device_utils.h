// Device-side helpers shared by the dense linear algebra (qil_linalg.hip) and the batched MPO builder
// (qil_build.hip): complex arithmetic, DPP cross-lane sums, Jacobi rotations and the in-workgroup
// one-sided Jacobi sweep loop.  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>

namespace qil_dev {

struct c64 {
    double re, im;
};

__device__ __forceinline__ double conj_t(double v) { return v; }
__device__ __forceinline__ c64 conj_t(c64 v) { return c64{v.re, -v.im}; }
__device__ __forceinline__ double fma_t(double a, double b, double acc) { return fma(a, b, acc); }
__device__ __forceinline__ c64 fma_t(c64 a, c64 b, c64 acc) {
    acc.re = fma(a.re, b.re, acc.re);
    acc.re = fma(-a.im, b.im, acc.re);
    acc.im = fma(a.re, b.im, acc.im);
    acc.im = fma(a.im, b.re, acc.im);
    return acc;
}
__device__ __forceinline__ double abs2_t(double v) { return v * v; }
__device__ __forceinline__ double abs2_t(c64 v) { return v.re * v.re + v.im * v.im; }
__device__ __forceinline__ double scale_t(double v, double s) { return v * s; }
__device__ __forceinline__ c64 scale_t(c64 v, double s) { return c64{v.re * s, v.im * s}; }
__device__ __forceinline__ double sub_t(double a, double b) { return a - b; }
__device__ __forceinline__ c64 sub_t(c64 a, c64 b) { return c64{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ double add_t(double a, double b) { return a + b; }
__device__ __forceinline__ c64 add_t(c64 a, c64 b) { return c64{a.re + b.re, a.im + b.im}; }

// ------------------------------------------------------------------ block reductions
// Cross-lane sums on the DPP path (no LDS crossbar): quad_perm butterflies inside each quad, then
// row_half_mirror / row_mirror fold the 8- and 16-lane halves; every lane of a 16-lane row ends with the
// row total.  A ds_bpermute-based __shfl_xor costs ~50+ cycles per step; a DPP move costs a VALU slot.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x140>(v);  // row_mirror
    return v;
}
__device__ __forceinline__ double read_lane_d(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v = row16_sum(v);
    return (read_lane_d(v, 0) + read_lane_d(v, 16)) + (read_lane_d(v, 32) + read_lane_d(v, 48));
}
// Totals of TWO values over the 64 lanes, both delivered to every lane, for the price of little more than one: a
// v_permlane32_swap folds the halves so that lanes 0-31 carry a and lanes 32-63 carry b, the four DPP steps and one
// v_permlane16_swap reduce each half, a last v_permlane32_swap broadcasts both totals (22 instructions against 2 x 25).
// The whole wave must be active.
__device__ __forceinline__ double swap_fold32(double x, double y) {   // lanes l < 32: x[l] + x[l + 32]; l >= 32: y[l - 32] + y[l]
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void wave_sum2(double& a, double& b) {
    double t = row16_sum(swap_fold32(a, b));
    {   // rows 0 + 1 and 2 + 3
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(t), (unsigned)__double2loint(t), false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(t), (unsigned)__double2hiint(t), false, false);
        t = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    }
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(t), (unsigned)__double2loint(t), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(t), (unsigned)__double2hiint(t), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}
// every lane of a 32-lane half ends with that half's total
__device__ __forceinline__ double half32_sum(double v) {
    v = row16_sum(v);
    const double lo = read_lane_d(v, 0) + read_lane_d(v, 16), hi = read_lane_d(v, 32) + read_lane_d(v, 48);
    return (threadIdx.x & 32) ? hi : lo;
}
template <int G>
__device__ __forceinline__ double group_sum(double v) {
    return G == 16 ? row16_sum(v) : G == 32 ? half32_sum(v) : wave_sum(v);
}

// sums `NV` doubles per thread across a 256-thread workgroup; result valid in all threads
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* lds /* NV * 4 */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < NV; ++i) lds[i * 4 + wave] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = lds[i * 4 + 0] + lds[i * 4 + 1] + lds[i * 4 + 2] + lds[i * 4 + 3];
}

// ------------------------------------------------------------------ one-sided Jacobi SVD
// One round of the round-robin tournament: workgroup i orthogonalises columns (p, q) of A
// (m x n) and applies the same rotation to V (n x n).
__device__ __forceinline__ void rotate_pair(double& x, double& y, double c, double s, double pr, double) {
    // pr = sign(x.y): the rotation angle is computed for |gamma|
    const double xn = c * x - s * pr * y, yn = s * pr * x + c * y;
    x = xn;
    y = yn;
}
__device__ __forceinline__ void rotate_pair(c64& x, c64& y, double c, double s, double pr, double pi) {
    // x' = c x - s e^{-i phi} y ;  y' = s e^{i phi} x + c y,  e^{i phi} = pr + i pi
    const c64 ey{pr * y.re + pi * y.im, pr * y.im - pi * y.re};  // e^{-i phi} y
    const c64 ex{pr * x.re - pi * x.im, pr * x.im + pi * x.re};  // e^{ i phi} x
    const c64 xn{c * x.re - s * ey.re, c * x.im - s * ey.im};
    const c64 yn{s * ex.re + c * y.re, s * ex.im + c * y.im};
    x = xn;
    y = yn;
}
__device__ __forceinline__ void dot_parts(double x, double y, double& gr, double& gi) {
    gr += x * y;
    (void)gi;
}
__device__ __forceinline__ void dot_parts(c64 x, c64 y, double& gr, double& gi) {
    // conj(x) * y
    gr += x.re * y.re + x.im * y.im;
    gi += x.re * y.im - x.im * y.re;
}

// Rotation that orthogonalises a column pair with alpha = |x|^2, beta = |y|^2, gamma = x^H y = gr + i gi.
// Returns false when the pair already passes |gamma| <= tol |x| |y|.  The rotation is a dependent chain of
// double-precision sqrt / divide sequences executed once per pair per round -- on the latency-bound
// in-LDS paths it IS the round time, so it is kept to one divide, one sqrt + divide and one rsqrt
// (real: |gamma| and its sign need neither a sqrt nor a divide).
// 1 / sqrt(x) to within an ulp: the library rsqrt plus one Newton step (3 FMAs).  The rotations must be
// unitary to rounding level -- a 2-3 ulp bias in c accumulates over the thousands of rotations V receives.
__device__ __forceinline__ double rsqrt_refined(double x) {
    const double y = rsqrt(x);
    return y * fma(-0.5 * x * y, y, 1.5);
}

// 1 / x from the hardware reciprocal plus one Newton step (2 FMAs): the IEEE divide sequence is ~4x longer
// and sits on the critical path of every rotation.
__device__ __forceinline__ double rcp_refined(double x) {
    const double y = __builtin_amdgcn_rcp(x);
    return y * fma(-x, y, 2.0);
}

// kQuadraticOff: once every pair a sweep meets is already orthogonal to this relative level BEFORE its rotation, that sweep
// (quadratic convergence) leaves all of them at rounding level: the iteration stops there instead of spending one more
// sweep on a handful of 1e-15-level rotations and another one on finding nothing to do.  `big` reports whether this
// pair was above that level.
constexpr double kQuadraticOff = 1e-9;
// Rank-deficient operands (every product bond before its truncation) leave columns that are pure rounding residue:
// |a|^2 below 1e-30 |A|_F^2.  Such a column has no direction to converge to -- each rotation of the genuine columns
// re-randomises it -- and keeps full sweeps going (builder slices: 14.7 -> 4.5 sweeps when it is left alone).  When the
// caller truncates with a cutoff these columns are far below it (the threshold also stays 100x under the cutoff), so
// pairs involving one are skipped; without a cutoff the threshold is 0 and nothing is skipped.
constexpr double kNegligibleColumn = 1e-30;

template <bool CX>
__device__ __forceinline__ bool jacobi_rotation(double al, double be, double gr, double gi, double tol,
                                                double& c, double& s, double& pr, double& pi, bool& big) {
    double half_inv_g;
    big = (gr * gr + gi * gi) > kQuadraticOff * kQuadraticOff * al * be;
    if (CX) {
        const double g2 = gr * gr + gi * gi;
        if (!(g2 > tol * tol * al * be) || g2 == 0.0) return false;
        const double inv_g = rsqrt_refined(g2);
        pr = gr * inv_g;
        pi = gi * inv_g;
        half_inv_g = 0.5 * inv_g;
    } else {
        const double g = fabs(gr);
        if (!(g * g > tol * tol * al * be) || g == 0.0) return false;
        pr = gr >= 0 ? 1.0 : -1.0;
        pi = 0.0;
        half_inv_g = 0.5 * rcp_refined(g);
    }
    // t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)); its accuracy only decides how well THIS pair is annihilated,
    // unitarity rests on c alone (s = c t, c = 1 / sqrt(1 + t^2) to an ulp)
    const double zeta = (be - al) * half_inv_g;
    const double az = fmin(fabs(zeta), 1e150);
    const double u = fma(az, az, 1.0);
    const double t = (zeta >= 0 ? 1.0 : -1.0) * rcp_refined(az + u * rsqrt_refined(u));
    c = rsqrt_refined(fma(t, t, 1.0));
    s = c * t;
    return true;
}

// Rotation of a column pair from |x|^2 = al, |y|^2 = be, x^H y = gr + i gi:  x' = c x - conj(sg) y,  y' = sg x + c y with
// sg = s e^{i phi} = (sr, si), e^{i phi} = g / |g|, s signed by be - al; sabs = |sg|, gabs ~ |g|.
// The dependent chain of this computation sits on the critical path of every inner round, so it is kept short: the angle comes
// from the raw hardware reciprocal square roots (relative error ~1e-8, which only decides how completely THIS pair is
// annihilated -- quadratic convergence absorbs it), and unitarity, which must hold to rounding because the errors of
// thousands of rotations add up in the singular values, is restored exactly afterwards: with eps = c0^2 + |sg0|^2 - 1
// (|eps| < 1e-6) both are scaled by 1 / sqrt(1 + eps) = 1 - eps / 2 + 3 eps^2 / 8 + O(eps^3 < 1e-18).
template <bool CX>
__device__ __forceinline__ bool rotation_fast(double al, double be, double gr, double gi, double tol, double& c, double& sr,
                                              double& si, double& sabs, double& gabs, bool& big) {
    const double g2 = CX ? fma(gr, gr, gi * gi) : gr * gr;
    const double ab = al * be;
    big = g2 > kQuadraticOff * kQuadraticOff * ab;
    if (!(g2 > tol * tol * ab) || g2 == 0.0) return false;
    const double d = be - al;
    const double rh = __builtin_amdgcn_rsq(fma(d, d, 4.0 * g2));
    const double c2 = fma(0.5 * fabs(d), rh, 0.5);
    const double rc = __builtin_amdgcn_rsq(c2);
    const double c0 = c2 * rc;
    const double q = copysign(rh * rc, d);            // sg0 = q g
    const double sr0 = q * gr, si0 = CX ? q * gi : 0.0;
    const double s02 = CX ? fma(sr0, sr0, si0 * si0) : sr0 * sr0;
    const double eps = fma(c0, c0, s02 - 1.0);
    const double f = fma(eps, fma(eps, 0.375, -0.5), 1.0);
    c = c0 * f;
    sr = sr0 * f;
    si = si0 * f;
    // |sg| and |g| only feed the carried norms (re-taken from the data at every staging): raw accuracy is enough
    gabs = CX ? g2 * __builtin_amdgcn_rsq(g2) : fabs(gr);
    sabs = fabs(q) * gabs * f;
    return true;
}
__device__ __forceinline__ void rotate_pair_sg(double& x, double& y, double c, double sr, double) {
    const double xn = fma(c, x, -sr * y), yn = fma(sr, x, c * y);
    x = xn;
    y = yn;
}
__device__ __forceinline__ void rotate_pair_sg(c64& x, c64& y, double c, double sr, double si) {
    // x' = c x - conj(sg) y ;  y' = sg x + c y
    const c64 xn{fma(c, x.re, -fma(sr, y.re, si * y.im)), fma(c, x.im, -fma(sr, y.im, -si * y.re))};
    const c64 yn{fma(c, y.re, fma(sr, x.re, -si * x.im)), fma(c, y.im, fma(sr, x.im, si * x.re))};
    x = xn;
    y = yn;
}

// Whole one-sided Jacobi SVD iteration in ONE launch of ONE 1024-thread workgroup: V = I, sweeps of the
// round-robin tournament until no pair rotates, then the column norms.  Each wave owns whole column
// pairs (lanes stride over rows, shuffle reductions), pairs of a round are disjoint, rounds are
// separated by a workgroup barrier.  Used when the rotated side is small (<= 128 columns): there the
// multi-launch form is bounded by ~(n-1) x sweeps kernel boundaries, not by work.  When A and V fit
// the CU's 160 KiB LDS (LDS = true) they are staged there for the whole iteration, so every round
// trip of the rotation is an LDS access instead of an L2 one.
template <class T, int G, int NT = 1024>
__device__ __forceinline__ void jacobi_sweeps(T* A, int lda, int m, T* V, int ldv, int n, double tol,
                                              int max_sweeps, int* s_rot, double negligible = 0.0) {
    // a column pair is owned by a group of G lanes (16 = one DPP row, or the whole wave); NT = workgroup size
    const int tid = threadIdx.x, lane = tid & (G - 1), wave = tid / G;
    constexpr int NW = NT / G;
    const int npad = n + (n & 1);
    for (int sweep = 0; sweep < max_sweeps && n > 1; ++sweep) {
        if (tid == 0) *s_rot = 0;
        __syncthreads();
        for (int round = 0; round < npad - 1; ++round) {
            for (int i = wave; i < npad / 2; i += NW) {
                int p, q;
                if (i == 0) {
                    p = npad - 1;
                    q = round;
                } else {
                    // both sums lie in [0, 2 (npad - 1)): one conditional subtraction, not a runtime modulo
                    p = round + i;
                    q = round + npad - 1 - i;
                    if (p >= npad - 1) p -= npad - 1;
                    if (q >= npad - 1) q -= npad - 1;
                }
                if (p >= n || q >= n) continue;
                if (p > q) {
                    const int t = p;
                    p = q;
                    q = t;
                }
                T* ap = A + lda * p;
                T* aq = A + lda * q;
                double al = 0, be = 0, gr = 0, gi = 0;
                for (int r = lane; r < m; r += G) {
                    const T x = ap[r], y = aq[r];
                    al += abs2_t(x);
                    be += abs2_t(y);
                    dot_parts(x, y, gr, gi);
                }
                al = group_sum<G>(al);
                be = group_sum<G>(be);
                gr = group_sum<G>(gr);
                if (sizeof(T) == 16) gi = group_sum<G>(gi);
                if (al < negligible || be < negligible) continue;     // rounding residue: see kNegligibleColumn
                double c, sr, si, sabs, gabs;
                bool big;
                if (!rotation_fast<sizeof(T) == 16>(al, be, gr, gi, tol, c, sr, si, sabs, gabs, big)) continue;
                if (lane == 0) atomicOr(s_rot, big ? 3 : 1);
                for (int r = lane; r < m; r += G) {
                    T x = ap[r], y = aq[r];
                    rotate_pair_sg(x, y, c, sr, si);
                    ap[r] = x;
                    aq[r] = y;
                }
                T* vp = V + ldv * p;
                T* vq = V + ldv * q;
                for (int r = lane; r < n; r += G) {
                    T x = vp[r], y = vq[r];
                    rotate_pair_sg(x, y, c, sr, si);
                    vp[r] = x;
                    vq[r] = y;
                }
            }
            __threadfence_block();
            __syncthreads();
        }
        const int any = *s_rot;
        __syncthreads();
        if (!(any & 2)) break;      // nothing rotated, or only pairs already below the quadratic-phase level
    }
}


// One-sided Jacobi on every slice: the columns of A (m x n, n <= m) are rotated in place until mutually
// orthogonal; norms (n) receives the column norms = singular values.  No V accumulation: the callers
// recover the other factor with one small GEMM.  A lives in LDS for the whole iteration when it fits.
template <class T, int G, int NT>
__device__ __forceinline__ void jacobi_sweeps_nov(T* A, int lda, int m, int n, double tol, int max_sweeps,
                                                  int* s_rot, double negligible) {
    const int tid = threadIdx.x, lane = tid & (G - 1), grp = tid / G;
    constexpr int NW = NT / G;
    const int npad = n + (n & 1);
    for (int sweep = 0; sweep < max_sweeps && n > 1; ++sweep) {
        if (tid == 0) *s_rot = 0;
        __syncthreads();
        for (int round = 0; round < npad - 1; ++round) {
            for (int i = grp; i < npad / 2; i += NW) {
                int p, q;
                if (i == 0) {
                    p = npad - 1;
                    q = round;
                } else {
                    // both sums lie in [0, 2 (npad - 1)): one conditional subtraction, not a runtime modulo
                    p = round + i;
                    q = round + npad - 1 - i;
                    if (p >= npad - 1) p -= npad - 1;
                    if (q >= npad - 1) q -= npad - 1;
                }
                if (p >= n || q >= n) continue;
                T* ap = A + lda * p;
                T* aq = A + lda * q;
                double al = 0, be = 0, gr = 0, gi = 0;
                for (int r = lane; r < m; r += G) {
                    const T x = ap[r], y = aq[r];
                    al += abs2_t(x);
                    be += abs2_t(y);
                    dot_parts(x, y, gr, gi);
                }
                al = group_sum<G>(al);
                be = group_sum<G>(be);
                gr = group_sum<G>(gr);
                // a column below 1e-15 of the slice's Frobenius norm is rounding residue of a rank-deficient
                // slice: it has no direction to converge to and would keep the sweeps going (6 instead of 8.5
                // sweeps on the builder's slices); its norm is far below any truncation cutoff
                if (al < negligible || be < negligible) continue;
                double c, sn, pr, pi_unused;
                bool big;
                if (!jacobi_rotation<false>(al, be, gr, 0.0, tol, c, sn, pr, pi_unused, big)) continue;
                if (lane == 0) atomicOr(s_rot, big ? 3 : 1);
                for (int r = lane; r < m; r += G) {
                    T x = ap[r], y = aq[r];
                    rotate_pair(x, y, c, sn, pr, 0.0);
                    ap[r] = x;
                    aq[r] = y;
                }
            }
            __threadfence_block();
            __syncthreads();
        }
        const int any = *s_rot;
        __syncthreads();
        if (!(any & 2)) break;
    }
}


}  // namespace qil_dev
