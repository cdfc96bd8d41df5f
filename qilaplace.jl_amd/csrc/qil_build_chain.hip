// Persistent builder of the COMPLEX transform chains: the QFT MPO and the paired-register QFT half of the zT MPO, each in
// ONE launch of ONE workgroup ("a persistent complex builder for the QFT chains").
//
//   build_qft_mpo(n; cutoff=1e-14, maxdim=1000)   src/transforms/qft_transformer.jl:121-165
//       zip_up_mpos :13-66 (QR-type factorisation from the last site upwards, nothing dropped), zip_down_mpos :69-101
//       (truncating SVD sweep from the orthogonality centre downwards), blocks control_Hphase_mpo src/circuits/qft_gates.jl:43-97
//   the paired QFT chain of build_zt_mpo          src/transforms/zt_transformer.jl:78-99
//       identity extension :81-95, zip_to_combine_mpos "down" (dt_transformer.jl:38-95), zip_to_compress_mpo "down"
//       (:185-230), blocks control_Hphase_ztmps_mpo src/circuits/zt_gates.jl:12-114
//
// Both are "multiply the chain by a bond-2 block on a window of sites, exactly, then truncate" repeated n - 1 times on tensors
// with bonds <= 8 (<= 16 before the truncation): ~550 (QFT, n = 24) / ~1 200 (paired chain) dependent factorisations of
// matrices no larger than 64 x 32 -- pure latency, like the DT build (qil_build_persist.hip).  The r03 device route assembled
// them from generic MPO x MPO and compression calls (5 ms of launches and read-backs per layer: 128 ms at n = 24, slower than
// numpy on the host).  Here the chain lives in a global workspace that stays in L2, the tensors being factorised in LDS, the gate
// blocks come as a table from the host (2 x 2 matrices), and every truncation decision is taken on the device.
//
// One frame serves both: a layer zips the block onto sites 0 .. L2-1 from the left (remainder carried to the right and absorbed
// into the next site), then truncates from `start` down to site 1 keeping right-isometric sites.  The paired chain is built in
// its natural order; the QFT chain -- whose reference sweeps run the other way -- in the MIRRORED frame (site order reversed,
// bond axes swapped), and is mirrored back on the way out.
//
// One factorisation primitive: complex one-sided Jacobi in LDS (no rotation matrix is accumulated; the second factor is a
// small product with the untouched copy of the operand).  The zip uses it as its "QR-type" step: Q = normalised rotated columns
// with non-negligible norm (an isometry; exactly dependent directions of a product bond carry weight < 1e-30 and are dropped,
// which no gauge-invariant quantity sees), remainder = Q^H A.  Only gauge-invariant results are comparable with the reference:
// dense operator and bond dimensions (tests/test_gpu_parity.py::test_persistent_qft_builders).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "qil_internal.h"
#include "qil_device_utils.h"

namespace {
using namespace qil_dev;

constexpr int CB_NT = 256;
constexpr int CB_NG = CB_NT / 16;
constexpr int CB_DCAP = 16;                          // largest bond inside the kernel (before truncation)
constexpr int CB_MAXL = 256;
constexpr long long CB_SITE_CAP = (long long)CB_DCAP * 4 * CB_DCAP;      // c64 entries per stored site

struct CbStep {
    int L2;            // the block acts on sites 0 .. L2-1 (frame order)
    int blk;           // first block tensor of this layer in the table
    int extend;        // append two identity sites (dim-1 bonds) first (zt_transformer.jl:81-95)
    int start;         // the truncating sweep runs from this site down to site 1
};

struct CbArgs {
    int nsteps, len0;
    const CbStep* steps;
    const c64* blocks;           // per block tensor 16 entries W[a + dl (s_in + 2 (s_out + 2 b))]
    const int* blkdims;          // per block tensor (dl, dr)
    c64* ws;                     // CB_MAXL x CB_SITE_CAP
    double cutoff;
    long long maxdim;
    int* dims_out;               // len + 1 bond dimensions with both edges; dims_out[CB_MAXL + 1] = status, [CB_MAXL + 2] = len
};

struct CbState {
    int bd[CB_MAXL + 1];
    c64 blk[16];
    int blk_dl, blk_dr;
    double sig[2 * CB_DCAP], inv[2 * CB_DCAP];
    int perm[2 * CB_DCAP];
    int rank, rot, err;
    double red[4];
};

__device__ __forceinline__ c64 cmul(c64 a, c64 b) { return c64{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ c64 cmulc(c64 a, c64 b) { return c64{a.re * b.re + a.im * b.im, a.re * b.im - a.im * b.re}; }   // conj(a) b

// sum_k op(a[k sa]) b[k sb], k < n, with the loads of four terms in flight before their first use (a rolled loop pays one LDS
// latency per term -- these sums are on the critical path of every site of every layer)
template <bool CONJ_A>
__device__ __forceinline__ c64 cb_dot(const c64* __restrict__ a, int sa, const c64* __restrict__ b, int sb, int n) {
    c64 acc{0, 0};
    for (int k0 = 0; k0 < n; k0 += 4) {
        c64 av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u < n ? k0 + u : n - 1;
            av[u] = a[k * sa];
            bv[u] = scale_t(b[k * sb], k0 + u < n ? 1.0 : 0.0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = fma_t(CONJ_A ? conj_t(av[u]) : av[u], bv[u], acc);
    }
    return acc;
}

// One-sided Jacobi on the columns of A (m x n, LDS, column stride lda); on exit the columns are mutually orthogonal and
// st.sig holds their norms.  One column pair per 16-lane row with both columns in registers (MU rows per lane, every LDS
// load of a phase issued before the first use: a rolled `for r` loop pays one LDS latency per element); columns below
// `negligible` (squared norm) are rounding residue of a rank-deficient operand and are left alone.
template <int MU>
__device__ void cb_jacobi_t(c64* A, int lda, int m, int n, CbState& st, double tol, double negligible) {
    const int tid = threadIdx.x, lane = tid & 15, grp = tid >> 4;
    const int npad = n + (n & 1);
    for (int sweep = 0; sweep < 40 && n > 1; ++sweep) {
        if (tid == 0) st.rot = 0;
        __syncthreads();
        for (int round = 0; round < npad - 1; ++round) {
            for (int i = grp; i < npad / 2; i += CB_NG) {
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
                if (p >= n || q >= n) continue;
                c64* ap = A + lda * p;
                c64* aq = A + lda * q;
                c64 x[MU], y[MU];
#pragma unroll
                for (int u = 0; u < MU; ++u) {
                    const int r = lane + 16 * u;
                    const int rr = r < m ? r : m - 1;
                    const double ok = r < m ? 1.0 : 0.0;
                    x[u] = scale_t(ap[rr], ok);
                    y[u] = scale_t(aq[rr], ok);
                }
                double al = 0, be = 0, gr = 0, gi = 0;
#pragma unroll
                for (int u = 0; u < MU; ++u) {
                    al += abs2_t(x[u]);
                    be += abs2_t(y[u]);
                    dot_parts(x[u], y[u], gr, gi);
                }
                al = row16_sum(al);
                be = row16_sum(be);
                gr = row16_sum(gr);
                gi = row16_sum(gi);
                if (al < negligible || be < negligible) continue;
                double c, sr, si, sabs, gabs;
                bool big;
                if (!rotation_fast<true>(al, be, gr, gi, tol, c, sr, si, sabs, gabs, big)) continue;
                if (lane == 0) atomicOr(&st.rot, big ? 3 : 1);
#pragma unroll
                for (int u = 0; u < MU; ++u) {
                    const int r = lane + 16 * u;
                    rotate_pair_sg(x[u], y[u], c, sr, si);
                    if (r < m) {
                        ap[r] = x[u];
                        aq[r] = y[u];
                    }
                }
            }
            __syncthreads();
        }
        const int any = st.rot;
        __syncthreads();
        if (!(any & 2)) break;
    }
    for (int j = grp; j < n; j += CB_NG) {
        const c64* a = A + lda * j;
        double v = 0;
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            const int r = lane + 16 * u;
            v += r < m ? abs2_t(a[r]) : 0.0;
        }
        v = row16_sum(v);
        if (lane == 0) st.sig[j] = sqrt(v);
    }
    __syncthreads();
}
__device__ void cb_jacobi(c64* A, int lda, int m, int n, CbState& st, double tol, double negligible) {
    if (m <= 16) cb_jacobi_t<1>(A, lda, m, n, st, tol, negligible);
    else if (m <= 32) cb_jacobi_t<2>(A, lda, m, n, st, tol, negligible);
    else cb_jacobi_t<4>(A, lda, m, n, st, tol, negligible);           // m <= 4 CB_DCAP = 64
}

// stable descending order of st.sig[0 .. n) into st.perm, reciprocals into st.inv (by sorted position)
__device__ void cb_sort(CbState& st, int n) {
    const int tid = threadIdx.x;
    if (tid < n) {
        const double s = st.sig[tid];
        int pos = 0;
        for (int q = 0; q < n; ++q) {
            const double o = st.sig[q];
            pos += (o > s) || (o == s && q < tid);
        }
        st.perm[pos] = tid;
    }
    __syncthreads();
    if (tid < n) {
        const double s = st.sig[st.perm[tid]];
        st.inv[tid] = s > 0.0 ? 1.0 / s : 0.0;
    }
    __syncthreads();
}

__global__ __launch_bounds__(CB_NT) void chain_build_persistent(CbArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cb_raw[];
    __shared__ CbState st;
    c64* arena = reinterpret_cast<c64*>(cb_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int TS = CB_DCAP * CB_DCAP * 2;                     // remainder T[r, a, c]
    c64* T = arena;
    c64* Tn = T + TS;
    c64* Ms = Tn + TS;                                            // old site, Da x 4 x B1
    c64* A = Ms + CB_SITE_CAP;                                    // the matrix being factorised (<= 64 x 32 / 16 x 64 transposed)
    c64* A0 = A + 4 * CB_DCAP * 2 * CB_DCAP;                      // its untouched copy
    c64* Mp = A0 + 4 * CB_DCAP * 2 * CB_DCAP;                     // neighbour site for the absorb products
    auto site = [&](int i) { return a.ws + (long long)i * CB_SITE_CAP; };
    int len = a.len0;
    if (tid == 0) st.err = 0;
    for (int i = tid; i <= len; i += CB_NT) st.bd[i] = a.dims_out[i];           // the host wrote the initial bonds there
    __syncthreads();
#define CB_FAIL(code)                  \
    do {                               \
        if (tid == 0) st.err = (code); \
        __syncthreads();               \
        goto done;                     \
    } while (0)
    for (int s = 0; s < a.nsteps; ++s) {
        const CbStep step = a.steps[s];
        if (step.extend) {
            if (len + 2 > CB_MAXL) CB_FAIL(7);
            if (tid < 2) {
                c64* sp = site(len + tid);
                sp[0] = c64{1, 0};
                sp[1] = c64{0, 0};
                sp[2] = c64{0, 0};
                sp[3] = c64{1, 0};
                st.bd[len + 1 + tid] = 1;
            }
            len += 2;
            __syncthreads();
        }
        // ---------------- zip: the block acts after the chain on sites 0 .. L2-1 (dt_transformer.jl:54-75 in this frame)
        const int L2 = step.L2;
        if (tid == 0) T[0] = c64{1, 0};
        int R = 1, Da = 1, Dc = 1;
        __syncthreads();
        for (int t = 0; t < L2; ++t) {
            const int B1 = st.bd[t + 1];
            if (tid < 16) st.blk[tid] = a.blocks[(long long)(step.blk + t) * 16 + tid];
            if (tid == 0) {
                st.blk_dl = a.blkdims[2 * (step.blk + t)];
                st.blk_dr = a.blkdims[2 * (step.blk + t) + 1];
            }
            if (B1 > CB_DCAP || Da * 4 * B1 > (int)CB_SITE_CAP) CB_FAIL(1);    // (before the load: Ms holds one stored site)
            for (int e = tid; e < Da * 4 * B1; e += CB_NT) Ms[e] = site(t)[e];
            __syncthreads();
            const int B2 = st.blk_dr;
            const int rows = 4 * R, cols = B1 * B2;
            if (st.blk_dl != Dc || R > CB_DCAP || B1 > CB_DCAP || rows > 4 * CB_DCAP || cols > 2 * CB_DCAP) CB_FAIL(1);
            // core[(r, i, o), (b1, b2)] = sum_{a, c, m} T[r, a, c] Ms[a, i, m, b1] Bk[c, m, o, b2] in two stages:
            // X[r, (i, m, b1), c] = sum_a T[r, a, c] Ms[a, (i, m, b1)], then the block's <= 4 non-zero terms per entry
            const int E = 4 * B1;
            c64* X = Mp;
            if (R * E * Dc > (int)CB_SITE_CAP) CB_FAIL(1);
            for (int e = tid; e < R * E * Dc; e += CB_NT) {
                const int r = e % R, ee = (e / R) % E, c = e / (R * E);
                X[e] = cb_dot<false>(T + r + R * Da * c, R, Ms + Da * ee, 1, Da);
            }
            __syncthreads();
            for (int e = tid; e < rows * cols; e += CB_NT) {
                const int row = e % rows, col = e / rows;
                const int r = row % R, i = (row / R) & 1, o = row / (2 * R);
                const int b1 = col % B1, b2 = col / B1;
                c64 xv[4], bv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = (u >> 1) < Dc ? (u >> 1) : 0, mm = u & 1;
                    xv[u] = X[r + R * ((i + 2 * (mm + 2 * b1)) + E * c)];
                    bv[u] = scale_t(st.blk[c + Dc * (mm + 2 * (o + 2 * b2))], (u >> 1) < Dc ? 1.0 : 0.0);
                }
                c64 acc{0, 0};
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = fma_t(xv[u], bv[u], acc);
                A[e] = acc;
                A0[e] = acc;
            }
            __syncthreads();
            c64* dst = site(t);
            int nb;
            if ((rows > cols ? cols : rows) * cols > TS + (int)CB_SITE_CAP) CB_FAIL(6);   // (Tn may run into Ms, never beyond it)
            if (rows > cols) {
                // "QR-type" step through the Jacobi primitive: Q = normalised rotated columns, remainder = Q^H core
                double f = 0;
                for (int e = tid; e < rows * cols; e += CB_NT) f += abs2_t(A[e]);
                f = wave_sum(f);
                if (lane == 0) st.red[wave] = f;
                __syncthreads();
                f = (st.red[0] + st.red[1]) + (st.red[2] + st.red[3]);
                cb_jacobi(A, rows, rows, cols, st, 1e-15, 1e-30 * f);
                cb_sort(st, cols);
                if (tid == 0) {
                    // keep the columns the sweeps have orthogonalised: everything above 1e-28 |A|_F^2 (the tournament leaves
                    // columns below 1e-30 |A|_F^2 alone -- they are rounding residue, not directions; what is dropped here
                    // carries less than 1e-28 of the weight, 13 orders below any cutoff of the truncating sweep that follows)
                    int k = 0;
                    while (k < cols && st.sig[st.perm[k]] * st.sig[st.perm[k]] > 1e-28 * f) ++k;
                    st.rank = k < 1 ? 1 : k;
                }
                __syncthreads();
                nb = st.rank;
                if (nb > CB_DCAP || rows * nb > (int)CB_SITE_CAP) CB_FAIL(6);   // (before the store: a stored site holds CB_SITE_CAP entries)
                for (int e = tid; e < rows * nb; e += CB_NT) {
                    const int row = e % rows, j = e / rows;
                    dst[e] = scale_t(A[row + rows * st.perm[j]], st.inv[j]);           // (r, i, o | new bond)
                }
                // Tn[j, col] = sum_row conj(Q[row, j]) A0[row, col]
                for (int e = tid; e < nb * cols; e += CB_NT) {
                    const int j = e % nb, col = e / nb;
                    Tn[e] = scale_t(cb_dot<true>(A + rows * st.perm[j], 1, A0 + rows * col, 1, rows), st.inv[j]);
                }
            } else {
                // fat core: Q = identity on the rows, the core itself is the remainder
                nb = rows;
                if (nb > CB_DCAP || rows * rows > (int)CB_SITE_CAP) CB_FAIL(6);
                for (int e = tid; e < rows * rows; e += CB_NT) dst[e] = (e % rows) == (e / rows) ? c64{1, 0} : c64{0, 0};
                for (int e = tid; e < rows * cols; e += CB_NT) Tn[e] = A0[e];
            }
            __syncthreads();
            if (nb > CB_DCAP || nb * cols > TS) CB_FAIL(6);          // (the next remainder would not fit)
            for (int e = tid; e < nb * cols; e += CB_NT) T[e] = Tn[e];
            if (tid == 0) {
                st.bd[t] = R;
                st.bd[t + 1] = nb;
            }
            __syncthreads();
            R = nb;
            Da = B1;
            Dc = B2;
        }
        // the block has ended (Dc == 1): T is R x Da
        if (Dc != 1) CB_FAIL(2);
        if (len > L2) {
            const int dr = st.bd[L2 + 1];
            if (Da * 4 * dr > CB_SITE_CAP || R > CB_DCAP) CB_FAIL(3);
            for (int e = tid; e < Da * 4 * dr; e += CB_NT) Mp[e] = site(L2)[e];
            __syncthreads();
            c64* dst = site(L2);
            for (int e = tid; e < R * 4 * dr; e += CB_NT) {
                const int r = e % R, rest = e / R;
                dst[e] = cb_dot<false>(T + r, R, Mp + Da * rest, 1, Da);
            }
            if (tid == 0) st.bd[L2] = R;
        } else {
            const int dl4 = st.bd[L2 - 1] * 4;
            if (dl4 * R > CB_SITE_CAP) CB_FAIL(4);
            for (int e = tid; e < dl4 * R; e += CB_NT) Mp[e] = site(L2 - 1)[e];
            __syncthreads();
            c64* dst = site(L2 - 1);
            for (int e = tid; e < dl4 * Da; e += CB_NT) {
                const int row = e % dl4, b = e / dl4;
                dst[e] = cb_dot<false>(Mp + row, dl4, T + R * b, 1, R);
            }
            if (tid == 0) st.bd[L2] = Da;
        }
        __syncthreads();
        // ---------------- truncating sweep start .. 1: every site to the left of the centre is an isometry, every site to its
        // right a right isometry, so the centre's singular values are the bond's (qft_transformer.jl:79-89, dt_transformer.jl:207-229)
        for (int i = step.start; i >= 1; --i) {
            const int d = st.bd[i], w = 4 * st.bd[i + 1], dl4 = 4 * st.bd[i - 1];
            if (d > 2 * CB_DCAP || w > 4 * CB_DCAP || d * w > 4 * CB_DCAP * 2 * CB_DCAP || dl4 * d > CB_SITE_CAP ||
                (d < w ? d : w) * w > CB_SITE_CAP || d * (d < w ? d : w) > CB_SITE_CAP)
                CB_FAIL(5);
            const bool tall = d > w;
            const int cols = tall ? w : d, rows = tall ? d : w;
            const c64* src = site(i);
            double f = 0;
            for (int e = tid; e < d * w; e += CB_NT) {
                const c64 v = src[e];
                A0[e] = v;                                           // M[i] as d x w
                const int r = e % d, c = e / d;
                if (tall) A[r + rows * c] = v;
                else A[c + rows * r] = conj_t(v);                    // M[i]^H as w x d
                f += abs2_t(v);
            }
            f = wave_sum(f);
            if (lane == 0) st.red[wave] = f;
            __syncthreads();
            f = (st.red[0] + st.red[1]) + (st.red[2] + st.red[3]);
            cb_jacobi(A, rows, rows, cols, st, 1e-15, 1e-30 * f);
            cb_sort(st, cols);
            if (tid == 0) {
                // qil_truncation_rank: the ITensors rule on the sorted squares, sums in the host's order
                int kk = cols;
                const double p0 = st.sig[st.perm[0]] * st.sig[st.perm[0]];
                if (!(p0 > 0.0) || cols == 1) kk = 1;
                else {
                    double scale = 0.0, terr = 0.0;
                    for (int q = 0; q < cols; ++q) scale += st.sig[st.perm[q]] * st.sig[st.perm[q]];
                    if (scale == 0.0) scale = 1.0;
                    const double lim = a.cutoff * scale;
                    for (int q = cols - 1; q >= 1; --q) {
                        const double p2 = st.sig[st.perm[q]] * st.sig[st.perm[q]];
                        if ((long long)(q + 1) > a.maxdim || terr + p2 <= lim) {
                            terr += p2;
                            kk = q;
                        } else break;
                    }
                    if (kk < 1) kk = 1;
                }
                st.rank = kk;
            }
            __syncthreads();
            const int rk = st.rank;
            c64* dsti = site(i);                                     // new M[i] = Vh (rk x w), right isometry
            c64* US = Ms;                                            // d x rk (the zip's staging area is free here)
            if (!tall) {
                // M[i]^H = U' S V'^H with U' = normalised rotated columns: Vh(M[i]) = U'^H, U S (M[i]) = M[i] U'
                for (int e = tid; e < rk * w; e += CB_NT) {
                    const int j = e % rk, c = e / rk;
                    dsti[e] = scale_t(conj_t(A[c + rows * st.perm[j]]), st.inv[j]);
                }
                for (int e = tid; e < d * rk; e += CB_NT) {
                    const int r = e % d, j = e / d;
                    US[e] = scale_t(cb_dot<false>(A0 + r, d, A + rows * st.perm[j], 1, w), st.inv[j]);
                }
            } else {
                // rotated columns of M[i] are U S; Vh = S^-2 (U S)^H M[i]
                for (int e = tid; e < d * rk; e += CB_NT) US[e] = A[(e % d) + rows * st.perm[e / d]];
                for (int e = tid; e < rk * w; e += CB_NT) {
                    const int j = e % rk, c = e / rk;
                    dsti[e] = scale_t(cb_dot<true>(A + rows * st.perm[j], 1, A0 + d * c, 1, d), st.inv[j] * st.inv[j]);
                }
            }
            for (int e = tid; e < dl4 * d; e += CB_NT) Mp[e] = site(i - 1)[e];
            __syncthreads();
            c64* dstp = site(i - 1);                                 // M[i-1] <- M[i-1] (U S)
            for (int e = tid; e < dl4 * rk; e += CB_NT) {
                const int row = e % dl4, j = e / dl4;
                dstp[e] = cb_dot<false>(Mp + row, dl4, US + d * j, 1, d);
            }
            if (tid == 0) st.bd[i] = rk;
            __syncthreads();
        }
    }
done:
    __syncthreads();
    for (int i = tid; i <= len; i += CB_NT) a.dims_out[i] = st.bd[i];
    if (tid == 0) {
        a.dims_out[CB_MAXL + 1] = st.err;
        a.dims_out[CB_MAXL + 2] = len;
    }
#undef CB_FAIL
}

// dst[i] = final chain site i; mirrored: site order reversed, bond axes swapped back
__global__ void cb_copy_out(const c64* __restrict__ ws, int L, const int* __restrict__ bd, c64* const* __restrict__ dst, int mirrored) {
    const int i = blockIdx.x;
    const int src = mirrored ? L - 1 - i : i;
    const int Da = bd[src], Db = bd[src + 1];
    const c64* s = ws + (long long)src * CB_SITE_CAP;
    c64* o = dst[i];
    for (int t = threadIdx.x; t < Da * 4 * Db; t += blockDim.x) {
        if (!mirrored) o[t] = s[t];
        else {
            // out[b, io, a] = in[a, io, b]: out has left bond Db
            const int b = t % Db, io = (t / Db) & 3, aa = t / (4 * Db);
            o[t] = s[aa + Da * (io + 4 * b)];
        }
    }
}

// ------------------------------------------------------------------ gate blocks on the host (2 x 2 matrices M[s_in, s_out])
struct Blk {
    int dl, dr;
    c64 w[16];
};
static void put(Blk& b, int a, int bb, const c64 g[4]) {          // g[s_in + 2 s_out]
    for (int si = 0; si < 2; ++si)
        for (int so = 0; so < 2; ++so) {
            c64& e = b.w[a + b.dl * (si + 2 * (so + 2 * bb))];
            e.re += g[si + 2 * so].re;
            e.im += g[si + 2 * so].im;
        }
}
static Blk mk(int dl, int dr) {
    Blk b;
    b.dl = dl;
    b.dr = dr;
    for (auto& e : b.w) e = c64{0, 0};
    return b;
}
static const double kIs2 = 0.70710678118654752440;
static void gI(c64 g[4]) { g[0] = c64{1, 0}, g[1] = c64{0, 0}, g[2] = c64{0, 0}, g[3] = c64{1, 0}; }
static void gH(c64 g[4]) { g[0] = c64{kIs2, 0}, g[1] = c64{kIs2, 0}, g[2] = c64{kIs2, 0}, g[3] = c64{-kIs2, 0}; }
static void gP(double theta, c64 g[4]) {                          // diag(1, e^{-i theta})   qft_gates.jl:24-30
    g[0] = c64{1, 0}, g[1] = c64{0, 0}, g[2] = c64{0, 0}, g[3] = c64{std::cos(theta), -std::sin(theta)};
}
static void gHproj(int c, c64 g[4]) {                             // (H Pi_c)[s_in, s_out] = H[s_in, c] delta(s_out, c)
    const double h[2][2] = {{kIs2, kIs2}, {kIs2, -kIs2}};
    for (int si = 0; si < 2; ++si)
        for (int so = 0; so < 2; ++so) g[si + 2 * so] = c64{so == c ? h[si][c] : 0.0, 0};
}
static void gprojH(int c, c64 g[4]) {                             // (Pi_c H)[s_in, s_out] = delta(s_in, c) H[c, s_out]
    const double h[2][2] = {{kIs2, kIs2}, {kIs2, -kIs2}};
    for (int si = 0; si < 2; ++si)
        for (int so = 0; so < 2; ++so) g[si + 2 * so] = c64{si == c ? h[c][so] : 0.0, 0};
}
// control_Hphase_mpo(k)  (qft_gates.jl:43-97): H then project the output of site 1; P(2 pi / 2^l) on site l
static std::vector<Blk> qft_block(int k) {
    std::vector<Blk> out;
    c64 g[4];
    if (k == 1) {
        Blk b = mk(1, 1);
        gH(g), put(b, 0, 0, g);
        out.push_back(b);
        return out;
    }
    Blk f = mk(1, 2);
    gHproj(0, g), put(f, 0, 0, g);
    gHproj(1, g), put(f, 0, 1, g);
    out.push_back(f);
    for (int l = 2; l < k; ++l) {
        Blk b = mk(2, 2);
        gI(g), put(b, 0, 0, g);
        gP(2.0 * M_PI / std::pow(2.0, l), g), put(b, 1, 1, g);
        out.push_back(b);
    }
    Blk e = mk(2, 1);
    gI(g), put(e, 0, 0, g);
    gP(2.0 * M_PI / std::pow(2.0, k), g), put(e, 1, 0, g);
    out.push_back(e);
    return out;
}
// control_Hphase_ztmps_mpo(k)  (zt_gates.jl:12-114): control = input bit of copy_k (project, then H); main sites carry the bond
static std::vector<Blk> zt_block(int k) {
    std::vector<Blk> out;
    c64 g[4];
    if (k == 1) {
        Blk m0 = mk(1, 1), c0 = mk(1, 1);
        gI(g), put(m0, 0, 0, g);
        gH(g), put(c0, 0, 0, g);
        out.push_back(m0), out.push_back(c0);
        return out;
    }
    Blk m1 = mk(1, 2);
    gI(g), put(m1, 0, 0, g), put(m1, 0, 1, g);
    out.push_back(m1);
    Blk c1 = mk(2, 2);
    gI(g), put(c1, 0, 0, g);
    gP(2.0 * M_PI / std::pow(2.0, k), g), put(c1, 1, 1, g);
    out.push_back(c1);
    for (int j = 2; j < k; ++j) {
        Blk m = mk(2, 2), c = mk(2, 2);
        gI(g), put(m, 0, 0, g), put(m, 1, 1, g);
        put(c, 0, 0, g);
        gP(2.0 * M_PI / std::pow(2.0, k - j + 1), g), put(c, 1, 1, g);
        out.push_back(m), out.push_back(c);
    }
    Blk ml = mk(2, 2);
    gI(g), put(ml, 0, 0, g), put(ml, 1, 1, g);
    out.push_back(ml);
    Blk cl = mk(2, 1);
    gprojH(0, g), put(cl, 0, 0, g);
    gprojH(1, g), put(cl, 1, 0, g);
    out.push_back(cl);
    return out;
}
static Blk mirror_blk(const Blk& b) {                             // out[bb, io, a] = in[a, io, bb]
    Blk o = mk(b.dr, b.dl);
    for (int a = 0; a < b.dl; ++a)
        for (int io = 0; io < 4; ++io)
            for (int bb = 0; bb < b.dr; ++bb) o.w[bb + b.dr * (io + 4 * a)] = b.w[a + b.dl * (io + 4 * bb)];
    return o;
}

}  // namespace

// kind 0: build_qft_mpo(n) (n sites, mirrored frame inside); kind 1: the paired QFT chain of build_zt_mpo (2 n sites).
// *fallback = 1 when a bond exceeded the in-LDS capacity (the caller takes the generic device route).
int qil_build_chain_persistent(qil_context* ctx, int kind, int64_t n, double cutoff, int64_t maxdim, const int64_t* site_ids,
                               qil_mpo** out, int* fallback) {
    *fallback = 0;
    const int L = (int)(kind == 0 ? n : 2 * n);
    if (L > CB_MAXL || n < 1) {
        *fallback = 1;
        return QIL_OK;
    }
    std::vector<Blk> table;
    std::vector<CbStep> steps;
    std::vector<Blk> init;
    if (kind == 0) {
        auto b0 = qft_block((int)n);
        for (int t = (int)b0.size() - 1; t >= 0; --t) init.push_back(mirror_blk(b0[(size_t)t]));
        for (int it = 1; it < n; ++it) {
            auto b = qft_block((int)n - it);
            CbStep s{(int)n - it, (int)table.size(), 0, (int)n - it};
            for (int t = (int)b.size() - 1; t >= 0; --t) table.push_back(mirror_blk(b[(size_t)t]));
            steps.push_back(s);
        }
    } else {
        init = zt_block(1);
        for (int k = 2; k <= n; ++k) {
            auto b = zt_block(k);
            CbStep s{2 * k, (int)table.size(), 1, 2 * k - 1};
            for (auto& t : b) table.push_back(t);
            steps.push_back(s);
        }
    }
    const int len0 = (int)init.size();
    std::vector<c64> blocks(std::max<size_t>(table.size(), 1) * 16);
    std::vector<int> blkdims(std::max<size_t>(table.size(), 1) * 2, 1);
    for (size_t t = 0; t < table.size(); ++t) {
        for (int e = 0; e < 16; ++e) blocks[t * 16 + e] = table[t].w[e];
        blkdims[2 * t] = table[t].dl;
        blkdims[2 * t + 1] = table[t].dr;
    }
    std::vector<int> meta(CB_MAXL + 3, 1);
    meta[0] = 1;
    for (int i = 0; i < len0; ++i) meta[(size_t)i + 1] = init[(size_t)i].dr;
    void *dblk = nullptr, *ddim = nullptr, *dstep = nullptr, *ws = nullptr, *dmeta = nullptr;
    QIL_TRY(qil_ctx_alloc(ctx, blocks.size() * sizeof(c64), &dblk));
    QIL_TRY(qil_ctx_alloc(ctx, blkdims.size() * sizeof(int), &ddim));
    QIL_TRY(qil_ctx_alloc(ctx, std::max<size_t>(steps.size(), 1) * sizeof(CbStep), &dstep));
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)CB_MAXL * CB_SITE_CAP * sizeof(c64), &ws));
    QIL_TRY(qil_ctx_alloc(ctx, meta.size() * sizeof(int), &dmeta));
    auto release = [&]() {
        for (void* p : {dblk, ddim, dstep, ws, dmeta})
            if (p) qil_ctx_free(ctx, p);
    };
    hipStream_t s = qil_stream(ctx);
    QIL_HIP(hipMemcpyAsync(dblk, blocks.data(), blocks.size() * sizeof(c64), hipMemcpyHostToDevice, s));
    QIL_HIP(hipMemcpyAsync(ddim, blkdims.data(), blkdims.size() * sizeof(int), hipMemcpyHostToDevice, s));
    if (!steps.empty()) QIL_HIP(hipMemcpyAsync(dstep, steps.data(), steps.size() * sizeof(CbStep), hipMemcpyHostToDevice, s));
    QIL_HIP(hipMemcpyAsync(dmeta, meta.data(), meta.size() * sizeof(int), hipMemcpyHostToDevice, s));
    // the initial chain: init[i].w is already the stored layout W[a + dl (io + 4 b)]
    std::vector<c64> init_flat((size_t)len0 * 16);
    for (int i = 0; i < len0; ++i)
        for (int e = 0; e < 16; ++e) init_flat[(size_t)i * 16 + e] = init[(size_t)i].w[e];
    for (int i = 0; i < len0; ++i)
        QIL_HIP(hipMemcpyAsync(static_cast<c64*>(ws) + (long long)i * CB_SITE_CAP, init_flat.data() + (size_t)i * 16,
                               (size_t)init[(size_t)i].dl * 4 * init[(size_t)i].dr * sizeof(c64), hipMemcpyHostToDevice, s));
    constexpr size_t lds_bytes = (size_t)(2 * CB_DCAP * CB_DCAP * 2 + CB_SITE_CAP + 2 * (4 * CB_DCAP * 2 * CB_DCAP) + CB_SITE_CAP) * sizeof(c64);
    static qil_lds_grant grant;                                  // per device; a refusal sends the caller down the generic route
    if (grant.ensure(ctx->device, reinterpret_cast<const void*>(&chain_build_persistent), lds_bytes) != hipSuccess) {
        (void)hipGetLastError();
        release();
        *fallback = 1;
        return QIL_OK;
    }
    CbArgs a;
    a.nsteps = (int)steps.size();
    a.len0 = len0;
    a.steps = static_cast<const CbStep*>(dstep);
    a.blocks = static_cast<const c64*>(dblk);
    a.blkdims = static_cast<const int*>(ddim);
    a.ws = static_cast<c64*>(ws);
    a.cutoff = std::max(cutoff, 1e-28);              // (never below what the sweeps orthogonalise: see qil_build_persist.hip, persist_chunk)
    a.maxdim = maxdim <= 0 ? INT64_MAX : maxdim;
    a.dims_out = static_cast<int*>(dmeta);
    hipLaunchKernelGGL(chain_build_persistent, dim3(1), dim3(CB_NT), lds_bytes, s, a);
    QIL_HIP(hipGetLastError());
    QIL_HIP(hipMemcpyAsync(meta.data(), dmeta, meta.size() * sizeof(int), hipMemcpyDeviceToHost, s));
    QIL_HIP(qil_stream_sync(ctx));           // also orders the pageable uploads before their release
    if (meta[CB_MAXL + 1] != 0 || meta[CB_MAXL + 2] != L) {
        release();
        *fallback = 1;
        return QIL_OK;
    }
    std::vector<int64_t> bonds((size_t)std::max(L - 1, 1));
    for (int i = 0; i + 1 < L; ++i) bonds[(size_t)i] = kind == 0 ? meta[(size_t)(L - 1 - i)] : meta[(size_t)i + 1];
    qil_mpo* W = nullptr;
    int st = qil_mpo_alloc(ctx, L, QIL_C64, kind == 1 ? 1 : 0, bonds.data(), site_ids, &W);
    void* dptr = nullptr;
    if (st == QIL_OK) st = qil_ctx_alloc(ctx, (size_t)L * sizeof(c64*), &dptr);
    if (st == QIL_OK) {
        std::vector<c64*> ptrs((size_t)L);
        for (int i = 0; i < L; ++i) ptrs[(size_t)i] = static_cast<c64*>(W->site[(size_t)i]);
        hipError_t e = hipMemcpyAsync(dptr, ptrs.data(), ptrs.size() * sizeof(c64*), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(cb_copy_out, dim3(L), dim3(256), 0, s, (const c64*)ws, L, (const int*)dmeta, (c64* const*)dptr, kind == 0 ? 1 : 0);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = qil_stream_sync(ctx);
        if (e != hipSuccess) st = qil_fail(QIL_EHIP, "persistent chain builder: copy-out failed: %s", hipGetErrorString(e));
    }
    if (dptr) qil_ctx_free(ctx, dptr);
    release();
    if (st != QIL_OK) {
        if (W) qil_mpo_destroy(W);
        return st;
    }
    *out = W;
    return QIL_OK;
}

// The paired QFT chain by the GENERIC device route (layers of qil_apply_mpo_mpo + qil_mpo_compress "down"): what the persistent
// kernel's callers take when a bond exceeds its in-LDS capacity (cutoffs far below the reference's default, chains longer than
// CB_MAXL).  zt_transformer.jl:78-99: block k acts on sites 1..2k, the chain is extended by an identity pair first -- the window
// product of apply(W1, W2) pads the shorter operand exactly like that (apply.jl:141-147).
int qil_build_zt_qft_chain_generic(qil_context* ctx, int64_t n, double cutoff, int64_t maxdim, const int64_t* site_ids, qil_mpo** out) {
    std::vector<int64_t> ids((size_t)(2 * n));
    for (int64_t i = 0; i < 2 * n; ++i) ids[(size_t)i] = site_ids ? site_ids[i] : i + 1;
    auto upload = [&](const std::vector<Blk>& b, qil_mpo** W) {
        const int64_t L = (int64_t)b.size();
        std::vector<int64_t> bonds((size_t)std::max<int64_t>(L - 1, 1), 1);
        std::vector<const void*> ptrs((size_t)L);
        for (int64_t i = 0; i < L; ++i) {
            if (i + 1 < L) bonds[(size_t)i] = b[(size_t)i].dr;
            ptrs[(size_t)i] = b[(size_t)i].w;                      // Blk::w is the stored layout W[a + dl (io + 4 b)], packed
        }
        return qil_mpo_create(ctx, L, QIL_C64, 1, bonds.data(), ids.data(), ptrs.data(), W);
    };
    qil_mpo* Q = nullptr;
    QIL_TRY(upload(zt_block(1), &Q));
    for (int k = 2; k <= n; ++k) {
        qil_mpo *B = nullptr, *P = nullptr;
        int st = upload(zt_block(k), &B);
        if (st == QIL_OK) st = qil_apply_mpo_mpo(Q, B, &P);
        if (B) qil_mpo_destroy(B);
        qil_mpo_destroy(Q);
        Q = P;
        if (st == QIL_OK) st = qil_mpo_compress(Q, 0, cutoff, maxdim);
        if (st != QIL_OK) {
            if (Q) qil_mpo_destroy(Q);
            return st;
        }
    }
    *out = Q;
    return QIL_OK;
}

// build_qft_mpo(n, sites; cutoff, maxdim) on the device (qft_transformer.jl:121-165)
extern "C" int qil_build_qft_mpo(qil_context* ctx, int64_t n, double cutoff, int64_t maxdim, const int64_t* site_ids, qil_mpo** out,
                                 int* fallback) {
    QIL_REQUIRE(ctx && out && fallback, QIL_EINVAL_ARG, "build_qft_mpo: null argument");
    QIL_REQUIRE(n >= 1, QIL_EINVAL_ARG, "build_qft_mpo: Number of qubits 'n' must be at least 1. Found n=%lld", (long long)n);
    QIL_REQUIRE(cutoff >= 0, QIL_EINVAL_ARG, "build_qft_mpo: cutoff must be >= 0");
    QIL_TRY(qil_ctx_activate(ctx));
    qil_call_scope call_scope(ctx);
    return qil_build_chain_persistent(ctx, 0, n, cutoff, maxdim, site_ids, out, fallback);
}

// the paired-register QFT chain of build_zt_mpo (zt_transformer.jl:78-99) on the device: 2 n sites main_1, copy_1, ...
extern "C" int qil_build_zt_qft_chain(qil_context* ctx, int64_t n, double cutoff, int64_t maxdim, const int64_t* site_ids,
                                      qil_mpo** out, int* fallback) {
    QIL_REQUIRE(ctx && out && fallback, QIL_EINVAL_ARG, "build_zt_mpo: null argument");
    QIL_REQUIRE(n >= 1, QIL_EINVAL_ARG, "build_zt_mpo: n must be >= 1. Found n=%lld", (long long)n);
    QIL_REQUIRE(cutoff >= 0, QIL_EINVAL_ARG, "build_zt_mpo: cutoff must be >= 0");
    QIL_TRY(qil_ctx_activate(ctx));
    qil_call_scope call_scope(ctx);
    return qil_build_chain_persistent(ctx, 1, n, cutoff, maxdim, site_ids, out, fallback);
}
