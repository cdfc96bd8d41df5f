// Truncation and encoding on the device: canonicalize! (K2), compress! (K1), truncated SVD,
// rsvd (E3), signal_mps (E1/E2), signal_ztmps (E4).
//
//   canonicalize!   src/mps.jl:787-847 (ZTMPS :866-901)
//   compress!       src/mps.jl:913-973 (ZTMPS :975-999)
//   rsvd            src/linalg/rsvd.jl:38-121
//   signal_mps      src/signals/SignalConverters.jl:16-46, 49-104 (:svd), 107-196 (:rsvd), 228-233
//   signal_ztmps    src/signals/SignalConverters.jl:247-283
//
// Matricisations are chosen so that every reshape is free in the canonical column-major layouts
// and the one-sided Jacobi SVD always rotates the SHORT side (see the notes at each call site).
#include <algorithm>
#include <chrono>
#include <cmath>

#include "qil_internal.h"
#include "qil_launch.h"

namespace {

struct c64 {
    double re, im;
};

constexpr int64_t kNoCap = INT64_MAX;

template <class F>
struct Defer {
    F f;
    ~Defer() { f(); }
};
template <class F>
Defer<F> defer(F f) {
    return Defer<F>{f};
}

// dst (rows x cols, ldd) <- src (rows x cols, lds): strided device copy
int copy_2d(qil_context* ctx, int dtype, int64_t rows, int64_t cols, const void* src, int64_t lds_, void* dst,
            int64_t ldd) {
    if (rows == 0 || cols == 0) return QIL_OK;
    const size_t e = qil_elem_size(dtype);
    QIL_TRY(qil_dev_copy2d(ctx, dst, (size_t)ldd * e, src, (size_t)lds_ * e, (size_t)rows * e, (size_t)cols));
    return QIL_OK;
}

// out[2 * block] += |A - D|_F^2 over this block's elements, out[2 * block + 1] += |A|_F^2   (A: lda, D: ldd; doubles viewed
// as reals: nre = 1 real / 2 complex values per element)
__device__ __forceinline__ void residual_sumsq_body(const uint3 blockIdx, const uint3 gridDim, const double* __restrict__ A, long long lda, const double* __restrict__ D,
                                                       long long ldd, long long m, long long n, int nre,
                                                       double* __restrict__ out) {
    __shared__ double red[8];
    double r2 = 0, a2 = 0;
    const long long rows = m * nre;
    for (long long t = blockIdx.x * 256LL + threadIdx.x; t < rows * n; t += (long long)gridDim.x * 256) {
        const long long i = t % rows, j = t / rows;
        const double a = A[i + lda * nre * j], d = D[i + ldd * nre * j];
        r2 = fma(a - d, a - d, r2);
        a2 = fma(a, a, a2);
    }
    for (int o = 32; o > 0; o >>= 1) {
        r2 += __shfl_xor(r2, o);
        a2 += __shfl_xor(a2, o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[wave] = r2;
        red[4 + wave] = a2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        out[2 * blockIdx.x + 1] = (red[4] + red[5]) + (red[6] + red[7]);
    }
}
struct residual_sumsq_k {
    static constexpr int NT = 256, MINW = 1;
    template <class... QA>
    static __device__ __forceinline__ void run(const uint3 b, const uint3 g, QA... a) {
        residual_sumsq_body(b, g, a...);
    }
};

int svd_trunc_dev(qil_context* ctx, int dtype, int64_t m, int64_t n, void* A, int64_t lda, double cutoff,
                  bool use_cutoff, int64_t maxdim, int64_t mindim, int absorb, int64_t* rank, void** U_out,
                  void** Vh_out, std::vector<double>* S_out);

// Low-rank fast path of a TRUNCATING SVD of a large operand (every product bond before its truncation is heavily
// rank-deficient: bond 1008 of the zT product carries ~60 singular values above 1e-6 of the largest).  Range finder
// with one power iteration, Q = orth(A (A^H (A Omega))) (k = 128 columns, then 256), B = Q^H A, and the residual
// |A - Q B|_F^2 is MEASURED: the path is taken only if it is below 1e-8 of the weight the caller's cutoff allows to be
// discarded -- then the truncation rule sees the same decision as on A itself -- and the truncated SVD of the k x n
// matrix B gives the factors (U = Q U_B).  Otherwise *done = 0 and A is intact.
static int svd_trunc_lowrank(qil_context* ctx, int dtype, int64_t m, int64_t n, void* A, int64_t lda, double cutoff,
                             int64_t maxdim, int64_t mindim, int absorb, int64_t* rank, void** U_out, void** Vh_out,
                             std::vector<double>* S_out, int* done) {
    *done = 0;
    const int64_t r0 = std::min(m, n);
    if (r0 < 384 || !(cutoff > 0)) return QIL_OK;
    const size_t e = qil_elem_size(dtype);
    const int cj = dtype == QIL_C64 ? 2 : 1;
    const int nre = dtype == QIL_C64 ? 2 : 1;
    for (int64_t k : {(int64_t)128, (int64_t)256}) {
        if (k * 3 > r0) break;
        void *Om = nullptr, *Y = nullptr, *Z = nullptr, *B = nullptr, *D = nullptr, *part = nullptr;
        auto release = [&]() {
            for (void* b : {Om, Y, Z, B, D, part})
                if (b) qil_ctx_free(ctx, b);
        };
        QIL_TRY(qil_ctx_alloc(ctx, (size_t)(n * k) * e, &Om));
        QIL_TRY(qil_dev_fill_normal(ctx, dtype, Om, n * k, 0x10a4c0deull + (uint64_t)k, 1.0));
        QIL_TRY(qil_ctx_alloc(ctx, (size_t)(m * k) * e, &Y));
        QIL_TRY(qil_ctx_alloc(ctx, (size_t)(n * k) * e, &Z));
        QIL_TRY(qil_dev_gemm(ctx, dtype, 0, 0, m, k, n, A, lda, Om, n, Y, m));                       // Y = A Omega
        QIL_TRY(qil_dev_qr_positive(ctx, dtype, m, k, Y, m, nullptr, 0, false));
        QIL_TRY(qil_dev_gemm(ctx, dtype, cj, 0, n, k, m, A, lda, Y, m, Z, n));                       // Z = A^H Y
        QIL_TRY(qil_dev_qr_positive(ctx, dtype, n, k, Z, n, nullptr, 0, false));
        QIL_TRY(qil_dev_gemm(ctx, dtype, 0, 0, m, k, n, A, lda, Z, n, Y, m));                        // Y = A Z
        QIL_TRY(qil_dev_qr_positive(ctx, dtype, m, k, Y, m, nullptr, 0, true));                      // Q (orthonormal)
        QIL_TRY(qil_ctx_alloc(ctx, (size_t)(k * n) * e, &B));
        QIL_TRY(qil_dev_gemm(ctx, dtype, cj, 0, k, n, m, Y, m, A, lda, B, k));                       // B = Q^H A
        QIL_TRY(qil_ctx_alloc(ctx, (size_t)(m * n) * e, &D));
        QIL_TRY(qil_dev_gemm(ctx, dtype, 0, 0, m, n, k, Y, m, B, k, D, m));                          // D = Q B
        const unsigned nblk_ = (unsigned)std::min<long long>((m * n * nre + 255) / 256, 1024);
        QIL_TRY(qil_ctx_alloc(ctx, (size_t)nblk_ * 2 * sizeof(double), &part));
        QIL_TRY((qil_klaunch<residual_sumsq_k>(ctx, dim3(nblk_), dim3(256), 0, (const double*)A, (long long)lda, (const double*)D, (long long)m, (long long)m, (long long)n, nre, (double*)part)));
        std::vector<double> hp((size_t)nblk_ * 2);
        QIL_TRY(qil_read_back(ctx, hp.data(), part, hp.size() * sizeof(double)));
        double rho = 0, tot = 0;
        for (unsigned b = 0; b < nblk_; ++b) {
            rho += hp[2 * b];
            tot += hp[2 * b + 1];
        }
        if (getenv("QIL_SVD_DEBUG")) fprintf(stderr, "[svd-lowrank] %lld x %lld, k = %lld: residual^2 / total = %.3e (cutoff %.1e)\n",
                                             (long long)m, (long long)n, (long long)k, tot > 0 ? rho / tot : 0.0, cutoff);
        if (!(rho <= 1e-8 * cutoff * tot)) {
            release();
            // a sketch twice as wide cannot help an operand that is nowhere near low rank (flat spectra leave > 50 % of
            // the weight outside 128 columns): only residuals that are already small earn the second attempt
            if (rho > 1e-4 * tot) break;
            continue;
        }
        int64_t r = 0;
        void *Ub = nullptr, *Vh = nullptr;
        std::vector<double> S;
        int st = svd_trunc_dev(ctx, dtype, k, n, B, k, cutoff, true, maxdim, mindim, absorb, &r, &Ub, &Vh, &S);
        if (st != QIL_OK) return st;
        void* U = nullptr;
        QIL_TRY(qil_ctx_alloc(ctx, (size_t)(m * std::min(m, n)) * e, &U));     // callers index U with ld m only
        QIL_TRY(qil_dev_gemm(ctx, dtype, 0, 0, m, r, k, Y, m, Ub, k, U, m));                         // U = Q U_B
        qil_ctx_free(ctx, Ub);
        release();
        *rank = r;
        *U_out = U;
        *Vh_out = Vh;
        if (S_out) *S_out = std::move(S);
        *done = 1;
        return QIL_OK;
    }
    return QIL_OK;
}

// Truncated SVD of the device matrix A (m x n, lda; destroyed).  Outputs are fresh pool blocks:
//   U  (m x r, ld m)   -- optionally scaled by S (absorb = 1)
//   Vh (r x n, ld r)   -- optionally scaled by S (absorb = 2)
int svd_trunc_dev(qil_context* ctx, int dtype, int64_t m, int64_t n, void* A, int64_t lda, double cutoff,
                  bool use_cutoff, int64_t maxdim, int64_t mindim, int absorb, int64_t* rank, void** U_out,
                  void** Vh_out, std::vector<double>* S_out) {
    const int64_t r0 = std::min(m, n);
    const size_t e = qil_elem_size(dtype);
    // Gauge sweeps that truncate by cutoff only and do not read the singular values (canonicalize!(cutoff), the first pass of
    // compress!): where the site's triangular factor certifies that nothing can be dropped, the thin QR is the gauge step
    // (qil_linalg.hip, "nothing can be truncated" certificate).  97..639 columns: inside the one-factor SVD, which has the QR
    // at hand; from 640 columns on: here, before the GEMM-shaped block Jacobi.
    const double cert_cutoff = (!S_out && use_cutoff && cutoff > 0.0 && r0 <= maxdim && (absorb == 1 || absorb == 2)) ? cutoff : 0.0;
    if (cert_cutoff > 0.0 && r0 >= 640) {
        const int64_t rows = std::max(m, n);
        void *Qb = nullptr, *Rb = nullptr;
        QIL_TRY(qil_ctx_alloc(ctx, (size_t)(rows * r0) * e, &Qb));
        QIL_TRY(qil_ctx_alloc(ctx, (size_t)(r0 * r0) * e, &Rb));
        bool certified = false;
        QIL_TRY(qil_dev_qr_certified(ctx, dtype, m, n, A, lda, cert_cutoff, Qb, Rb, &certified));
        if (getenv("QIL_SVD_DEBUG"))
            fprintf(stderr, "[svd-cert] %lld x %lld (lda %lld), absorb %d: %s\n", (long long)m, (long long)n, (long long)lda, absorb,
                    certified ? "QR gauge" : "declined");
        if (certified) {
            void *U = nullptr, *Vh = nullptr;
            QIL_TRY(qil_ctx_alloc(ctx, (size_t)(m * r0) * e, &U));
            QIL_TRY(qil_ctx_alloc(ctx, (size_t)(r0 * n) * e, &Vh));
            if (absorb == 2) {                       // U isometric, the rest absorbed
                if (m >= n) {                        // A = Q R
                    QIL_TRY(copy_2d(ctx, dtype, m, r0, Qb, rows, U, m));
                    QIL_TRY(copy_2d(ctx, dtype, r0, n, Rb, r0, Vh, r0));
                } else {                             // every row direction is kept: U = I, the rest = A
                    QIL_TRY(qil_dev_zero(ctx, U, (size_t)(m * r0) * e));
                    QIL_TRY(qil_dev_set_identity(ctx, dtype, U, m, r0));
                    QIL_TRY(copy_2d(ctx, dtype, m, n, A, lda, Vh, r0));
                }
            } else {                                 // Vh isometric, U S absorbed
                if (m < n) {                         // A^H = Q R  =>  A = R^H Q^H   (m == n: the factor is of A itself)
                    QIL_TRY(qil_dev_transpose(ctx, dtype, 1, rows, r0, Qb, rows, Vh, r0));
                    QIL_TRY(qil_dev_transpose(ctx, dtype, 1, r0, r0, Rb, r0, U, m));
                } else {                             // every column direction is kept: Vh = I, U S = A
                    QIL_TRY(qil_dev_zero(ctx, Vh, (size_t)(r0 * n) * e));
                    QIL_TRY(qil_dev_set_identity(ctx, dtype, Vh, r0, r0));
                    QIL_TRY(copy_2d(ctx, dtype, m, n, A, lda, U, m));
                }
            }
            qil_ctx_free(ctx, Qb);
            qil_ctx_free(ctx, Rb);
            *rank = r0;
            *U_out = U;
            *Vh_out = Vh;
            return QIL_OK;
        }
        qil_ctx_free(ctx, Qb);
        qil_ctx_free(ctx, Rb);
    }
    if (use_cutoff) {
        int done = 0;
        QIL_TRY(svd_trunc_lowrank(ctx, dtype, m, n, A, lda, cutoff, maxdim, mindim, absorb, rank, U_out, Vh_out, S_out, &done));
        if (done) return QIL_OK;
    }
    void *U = nullptr, *Vh = nullptr;
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(m * r0) * e, &U));
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(r0 * n) * e, &Vh));
    std::vector<double> S((size_t)r0);
    // with a truncating cutoff, columns 100x below it (and never above 1e-30 |A|_F^2) are not worth rotating
    const double negl_rel = use_cutoff && cutoff > 0 ? std::min(1e-30, 1e-2 * cutoff) : 0.0;
    // Gauge sweeps keep ONE factor as the site (isometric) and multiply the other into the neighbour: the mid-size path
    // that accumulates no rotation matrix (qil_dev_svd_left) serves them; everything else takes the general SVD.
    int handled = 0;
    struct deflate_scope {                                       // the one-factor SVD may drop 1e-6 of what the cutoff allows
        qil_context* c;
        ~deflate_scope() { c->svd_deflate = 0.0; }
    } dscope{ctx};
    // (the weight w the one-factor SVD may drop costs sqrt(w) in amplitude.  Measured on the exact compress! of the bond-1008 zT
    // product, w = 1e-3 / 1e-5 / 1e-6 / 1e-8 of the cutoff: 253-260 / 282 / 266-273 / 270 ms (295 without), state against the
    // CPU oracle's compress! 1.2e-7 / 1.6e-9 / 4e-10 / 3e-10 (1e-11 without; the algorithm's own error there is 1.8e-5))
    ctx->svd_deflate = (use_cutoff && cutoff > 0.0) ? 1e-6 * cutoff : 0.0;
    if (absorb == 2) {            // U isometric, S Vh absorbed
        QIL_TRY(qil_dev_svd_left(ctx, dtype, m, n, A, lda, U, m, S.data(), Vh, r0, negl_rel, &handled, cert_cutoff));
    } else if (absorb == 1) {     // Vh isometric, U S absorbed: the same problem on A^H
        void *At = nullptr, *Vi = nullptr, *SU = nullptr;
        QIL_TRY(qil_ctx_alloc(ctx, (size_t)(m * n) * e, &At));
        QIL_TRY(qil_dev_transpose(ctx, dtype, 1, m, n, A, lda, At, n));                         // A^H (n x m)
        QIL_TRY(qil_ctx_alloc(ctx, (size_t)(n * r0) * e, &Vi));
        QIL_TRY(qil_ctx_alloc(ctx, (size_t)(r0 * m) * e, &SU));
        QIL_TRY(qil_dev_svd_left(ctx, dtype, n, m, At, n, Vi, n, S.data(), SU, r0, negl_rel, &handled, cert_cutoff));
        if (handled) {
            QIL_TRY(qil_dev_transpose(ctx, dtype, 1, n, r0, Vi, n, Vh, r0));                    // Vh = V^H   (r0 x n)
            QIL_TRY(qil_dev_transpose(ctx, dtype, 1, r0, m, SU, r0, U, m));                     // U S = (S U^H)^H
        }
        qil_ctx_free(ctx, At);
        qil_ctx_free(ctx, Vi);
        qil_ctx_free(ctx, SU);
    }
    if (!handled) QIL_TRY(qil_dev_svd(ctx, dtype, m, n, A, lda, U, m, S.data(), Vh, r0, negl_rel));
    // handled == 2: certified that nothing can be truncated (S was not computed)
    const int64_t r = handled == 2 ? r0 : qil_truncation_rank(S.data(), r0, cutoff, use_cutoff, maxdim, mindim);
    if (!handled && absorb == 1) QIL_TRY(qil_dev_scale(ctx, dtype, 1, m, r, U, m, S.data()));
    if (!handled && absorb == 2) QIL_TRY(qil_dev_scale(ctx, dtype, 0, r, n, Vh, r0, S.data()));
    if (r < r0) {  // compact Vh rows to leading dimension r
        void* Vc = nullptr;
        QIL_TRY(qil_ctx_alloc(ctx, (size_t)(r * n) * e, &Vc));
        QIL_TRY(copy_2d(ctx, dtype, r, n, Vh, r0, Vc, r));
        qil_ctx_free(ctx, Vh);
        Vh = Vc;
    }
    S.resize((size_t)r);
    *rank = r;
    *U_out = U;  // first r columns are the kept ones (contiguous)
    *Vh_out = Vh;
    if (S_out) *S_out = std::move(S);
    return QIL_OK;
}

// ---------------------------------------------------------------- canonicalize!
// Works on any chain: the physical block is 2 (MPS site A[a,s,b]) or 4 (MPO site W[a,s',s,b]) wide and sits
// between the two bonds in memory, so "rows (a, phys) | cols b" and "rows a | cols (phys, b)" are the site
// buffer as it lies in both cases.
// Exact gauge sweep by thin QR (no truncation): the first pass of zip_to_compress_mpo (dt_transformer.jl:190, :237
// use `qr`, only the return pass uses `svd`).  Same site layouts as canonicalize_impl below; a site whose matrix is
// wider than tall has no thin QR that shrinks nothing, so it takes the SVD route with cutoff 0.
static int gauge_qr_site_right(qil_chain* psi, int64_t i, int64_t pd) {
    qil_context* ctx = psi->ctx;
    const int dt = psi->dtype;
    const size_t e = qil_elem_size(dt);
    const int64_t cl = psi->dims[(size_t)i], cr = psi->dims[(size_t)i + 1], cr2 = psi->dims[(size_t)i + 2];
    void *Rf = nullptr, *next = nullptr;
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(cr * cr) * e, &Rf));
    QIL_TRY(qil_dev_qr_positive(ctx, dt, pd * cl, cr, psi->site[(size_t)i], pd * cl, Rf, cr, true));   // Q in place
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(cr * pd * cr2) * e, &next));
    QIL_TRY(qil_dev_gemm(ctx, dt, 0, 0, cr, pd * cr2, cr, Rf, cr, psi->site[(size_t)i + 1], cr, next, cr));
    qil_ctx_free(ctx, Rf);
    QIL_TRY(qil_chain_set_site(psi, i + 1, next, cr, cr2));
    return QIL_OK;
}

static int gauge_qr_site_left(qil_chain* psi, int64_t i, int64_t pd) {
    qil_context* ctx = psi->ctx;
    const int dt = psi->dtype;
    const size_t e = qil_elem_size(dt);
    const int64_t cl = psi->dims[(size_t)i], cr = psi->dims[(size_t)i + 1], cl0 = psi->dims[(size_t)i - 1];
    // A (cl x pd cr) = L Q with orthonormal rows of Q:  A^H = Qt Rt  =>  Q = Qt^H, L = Rt^H
    void *Ah = nullptr, *Rt = nullptr, *prev = nullptr;
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(pd * cr * cl) * e, &Ah));
    QIL_TRY(qil_dev_transpose(ctx, dt, 1, cl, pd * cr, psi->site[(size_t)i], cl, Ah, pd * cr));
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(cl * cl) * e, &Rt));
    QIL_TRY(qil_dev_qr_positive(ctx, dt, pd * cr, cl, Ah, pd * cr, Rt, cl, true));
    QIL_TRY(qil_dev_transpose(ctx, dt, 1, pd * cr, cl, Ah, pd * cr, psi->site[(size_t)i], cl));
    qil_ctx_free(ctx, Ah);
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(cl0 * pd * cl) * e, &prev));
    QIL_TRY(qil_dev_gemm(ctx, dt, 0, 2, pd * cl0, cl, cl, psi->site[(size_t)i - 1], pd * cl0, Rt, cl, prev, pd * cl0));
    qil_ctx_free(ctx, Rt);
    QIL_TRY(qil_chain_set_site(psi, i - 1, prev, cl0, cl));
    return QIL_OK;
}

// One step of the right-to-left gauge sweep (mps.jl:822-837): site i becomes a right isometry (rows alpha | cols
// (phys, beta)), its left factor is multiplied into site i-1.  gauge_qr: exact thin QR where the site is wide enough,
// truncated SVD otherwise (and always when gauge_qr is off).
static int gauge_site_left(qil_chain* psi, int64_t i, double cutoff, int64_t maxdim, bool gauge_qr) {
    const int64_t pd = psi->phys_rank == 1 ? 2 : 4;
    qil_context* ctx = psi->ctx;
    const int dt = psi->dtype;
    const size_t e = qil_elem_size(dt);
    const int64_t cl = psi->dims[(size_t)i], cr = psi->dims[(size_t)i + 1];
    const int64_t cl0 = psi->dims[(size_t)i - 1];
    if (gauge_qr && pd * cr >= cl) return gauge_qr_site_left(psi, i, pd);
    int64_t r = 0;
    void *US = nullptr, *Vh = nullptr;
    // rows alpha | cols (s, beta)
    QIL_TRY(svd_trunc_dev(ctx, dt, cl, pd * cr, psi->site[(size_t)i], cl, cutoff, true, maxdim, 1, 1, &r, &US, &Vh,
                          nullptr));
    void* prev = nullptr;
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(cl0 * pd * r) * e, &prev));
    QIL_TRY(qil_dev_gemm(ctx, dt, 0, 0, pd * cl0, r, cl, psi->site[(size_t)i - 1], pd * cl0, US, cl, prev, pd * cl0));
    qil_ctx_free(ctx, US);
    QIL_TRY(qil_chain_set_site(psi, i, Vh, r, cr));
    QIL_TRY(qil_chain_set_site(psi, i - 1, prev, cl0, r));
    return QIL_OK;
}

int canonicalize_impl(qil_chain* psi, int direction, int64_t center, double cutoff, int64_t maxdim,
                      bool gauge_qr = false) {
    const int64_t pd = psi->phys_rank == 1 ? 2 : 4;
    qil_context* ctx = psi->ctx;
    const int64_t N = psi->n();
    const int dt = psi->dtype;
    const size_t e = qil_elem_size(dt);
    if (direction == QIL_DIR_RIGHT) {
        const int64_t c = center == 0 ? N : center;
        QIL_REQUIRE(c >= 1 && c <= N, QIL_EDOMAIN, "Center out of range [1,%lld]", (long long)N);
        for (int64_t i = 0; i + 1 < c; ++i) {  // mps.jl:802-817
            qil_progress_step(ctx, i == 0, i);                                  // lock-step batches: where this chain is
            const int64_t cl = psi->dims[(size_t)i], cr = psi->dims[(size_t)i + 1];
            const int64_t cr2 = psi->dims[(size_t)i + 2];
            if (gauge_qr && pd * cl >= cr) {
                QIL_TRY(gauge_qr_site_right(psi, i, pd));
                continue;
            }
            int64_t r = 0;
            void *U = nullptr, *SV = nullptr;
            // rows (alpha, s) | cols beta : the site buffer as it lies
            QIL_TRY(svd_trunc_dev(ctx, dt, pd * cl, cr, psi->site[(size_t)i], pd * cl, cutoff, true, maxdim, 1, 2, &r,
                                  &U, &SV, nullptr));
            void* next = nullptr;
            QIL_TRY(qil_ctx_alloc(ctx, (size_t)(r * pd * cr2) * e, &next));
            QIL_TRY(qil_dev_gemm(ctx, dt, 0, 0, r, pd * cr2, cr, SV, r, psi->site[(size_t)i + 1], cr, next, r));
            qil_ctx_free(ctx, SV);
            QIL_TRY(qil_chain_set_site(psi, i, U, cl, r));
            QIL_TRY(qil_chain_set_site(psi, i + 1, next, r, cr2));
        }
    } else if (direction == QIL_DIR_LEFT) {
        const int64_t c = center == 0 ? 1 : center;
        QIL_REQUIRE(c >= 1 && c <= N, QIL_EDOMAIN, "Center out of range [1,%lld]", (long long)N);
        for (int64_t i = N - 1; i >= c; --i) {                                                             // mps.jl:822-837
            qil_progress_step(ctx, i == N - 1, N - 1 - i);
            QIL_TRY(gauge_site_left(psi, i, cutoff, maxdim, gauge_qr));
        }
    } else {
        return qil_fail(QIL_EINVAL_ARG, "Direction must be :right or :left");
    }
    return QIL_OK;
}

// ---------------------------------------------------------------- helper kernels (layout permutations)
// site A[alpha + cl*(s + 2*k)] = Vyh[k + ldv*(s + 2*alpha)]      (signal_mps :svd, see below)
template <class T>
__device__ __forceinline__ void site_from_vh_body(const uint3 blockIdx, const uint3 gridDim, const T* __restrict__ Vyh, long long ldv, int cl, int k, T* __restrict__ A) {
    const long long total = 2LL * cl * k;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int alpha = (int)(t % cl);
        const long long u = t / cl;
        const int s = (int)(u & 1);
        const int kk = (int)(u >> 1);
        A[t] = Vyh[kk + ldv * (s + 2LL * alpha)];
    }
}
template <class T>
struct site_from_vh_k {
    static constexpr int NT = 1024, MINW = 1;
    template <class... QA>
    static __device__ __forceinline__ void run(const uint3 b, const uint3 g, QA... a) {
        site_from_vh_body<T>(b, g, a...);
    }
};

// canonical site A[lb + cl*(s + 2*rb)] from the chunk layout X[rb + cr*(s + 2*lb)]
template <class T>
__device__ __forceinline__ void site_from_chunk_body(const uint3 blockIdx, const uint3 gridDim, const T* __restrict__ X, int cl, int cr, T* __restrict__ A) {
    const long long total = 2LL * cl * cr;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int lb = (int)(t % cl);
        const long long u = t / cl;
        const int s = (int)(u & 1);
        const int rb = (int)(u >> 1);
        A[t] = X[rb + (long long)cr * (s + 2LL * lb)];
    }
}
template <class T>
struct site_from_chunk_k {
    static constexpr int NT = 1024, MINW = 1;
    template <class... QA>
    static __device__ __forceinline__ void run(const uint3 b, const uint3 g, QA... a) {
        site_from_chunk_body<T>(b, g, a...);
    }
};

// T[(alpha, m), (c, beta)] = A[alpha, m, beta] * delta(m, c)   (signal_ztmps, SignalConverters.jl:263)
template <class T>
__device__ __forceinline__ void fuse_delta_body(const uint3 blockIdx, const uint3 gridDim, const T* __restrict__ A, int cl, int cr, T* __restrict__ Tm) {
    const long long total = 4LL * cl * cr;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const long long row = t % (2LL * cl), col = t / (2LL * cl);
        const int alpha = (int)(row % cl), m = (int)(row / cl);
        const int c = (int)(col & 1);
        const long long beta = col >> 1;
        Tm[t] = (m == c) ? A[alpha + (long long)cl * (m + 2 * beta)] : T{};
    }
}
template <class T>
struct fuse_delta_k {
    static constexpr int NT = 1024, MINW = 1;
    template <class... QA>
    static __device__ __forceinline__ void run(const uint3 b, const uint3 g, QA... a) {
        fuse_delta_body<T>(b, g, a...);
    }
};

inline unsigned nblk(long long total) { return (unsigned)std::min<long long>((total + 255) / 256, 65536); }

// per-workgroup partial sums of squares (fixed grid => the host-side final sum has a fixed order)
__device__ __forceinline__ void sumsq_partial_body(const uint3 blockIdx, const uint3 gridDim, const double* __restrict__ x, long long n,
                                                     double* __restrict__ part) {
    __shared__ double red[4];
    double v = 0;
    for (long long t = blockIdx.x * 256LL + threadIdx.x; t < n; t += (long long)gridDim.x * 256) v += x[t] * x[t];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
struct sumsq_partial_k {
    static constexpr int NT = 256, MINW = 1;
    template <class... QA>
    static __device__ __forceinline__ void run(const uint3 b, const uint3 g, QA... a) {
        sumsq_partial_body(b, g, a...);
    }
};
__device__ __forceinline__ void scale_inplace_body(const uint3 blockIdx, const uint3 gridDim, double* __restrict__ x, long long n, double s) {
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n;
         t += (long long)gridDim.x * blockDim.x)
        x[t] *= s;
}
struct scale_inplace_k {
    static constexpr int NT = 1024, MINW = 1;
    template <class... QA>
    static __device__ __forceinline__ void run(const uint3 b, const uint3 g, QA... a) {
        scale_inplace_body(b, g, a...);
    }
};
// y = s x (out of place: the caller's device-resident samples are read once for the norm and once here)
__device__ __forceinline__ void scale_copy_body(const uint3 blockIdx, const uint3 gridDim, const double* __restrict__ x, double* __restrict__ y, long long n, double s) {
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n;
         t += (long long)gridDim.x * blockDim.x)
        y[t] = x[t] * s;
}
struct scale_copy_k {
    static constexpr int NT = 1024, MINW = 1;
    template <class... QA>
    static __device__ __forceinline__ void run(const uint3 b, const uint3 g, QA... a) {
        scale_copy_body(b, g, a...);
    }
};

// ---------------------------------------------------------------- rsvd on a device operand
// `Z` holds M^T (plain transpose) column-major: Z is (n x m) for the m x n operand M ("M stored
// row-major").  Produces, with k = kept rank,
//   left  = U_M^T        (k x m, ld k)      -- U_M = Q * Uhat
//   right = (S V_M^h)^T  (n x k, ld n)
// Steps follow src/linalg/rsvd.jl:72-114; the small SVD is taken of B^T = Z conj(Q) (n x l), whose
// short side is l, so the Jacobi rotations act on l columns and (S V^h)^T falls out of the rotated
// work matrix directly.
int rsvd_rowmajor(qil_context* ctx, int dt, int64_t m, int64_t n, const void* Z, int64_t k, int64_t p, int q,
                  uint64_t seed, double cutoff, int64_t maxdim, int64_t mindim, int64_t* rank, void** left,
                  void** right, std::vector<double>* S_out) {
    QIL_REQUIRE(m >= 1 && n >= 1, QIL_EEMPTY, "In `rsvd`, left or right index set is empty.");
    const size_t e = qil_elem_size(dt);
    const int64_t l0 = std::min(std::min(k + p, m), n);  // rsvd.jl:72
    if (l0 == std::min(m, n)) {
        // The sketch is as wide as the short side: Q spans the whole range of M and the procedure returns the exact
        // truncated SVD (up to rounding) after 2 + 2q products and 1 + 2q QRs of matrices no larger than M itself.
        // Take the SVD directly: Z = M^T = Uz Sz Vz^h  =>  U_M^T = Vz^h,  (S V_M^h)^T = Uz Sz.  Every deep split of
        // the bisection encoder ends up here (SignalConverters.jl:161: m or n <= k + p).
        void* Zc = nullptr;
        QIL_TRY(qil_ctx_alloc(ctx, (size_t)(m * n) * e, &Zc));
        QIL_TRY(qil_dev_copy(ctx, Zc, Z, (size_t)(m * n) * e));
        int64_t r = 0;
        void *UzS = nullptr, *Vzh = nullptr;
        QIL_TRY(svd_trunc_dev(ctx, dt, n, m, Zc, n, cutoff, true, maxdim, mindim, 1, &r, &UzS, &Vzh, S_out));
        qil_ctx_free(ctx, Zc);
        *rank = r;
        *left = Vzh;    // r x m, ld r
        *right = UzS;   // n x (>= r) columns, ld n
        return QIL_OK;
    }
    void *Om = nullptr, *Y = nullptr, *Zq = nullptr, *Bt = nullptr;
    const bool dbg = getenv("QIL_RSVD_DEBUG") != nullptr && m * n >= ((1LL << 24));
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!dbg) return;
        (void)qil_stream_sync(ctx);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[rsvd] %lld x %lld, l = %lld: %s %.2f ms\n", (long long)m, (long long)n, (long long)l0, what,
                std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(n * l0) * e, &Om));
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(m * l0) * e, &Y));
    QIL_TRY(qil_dev_fill_normal(ctx, dt, Om, n * l0, seed, 1.0));                        // rsvd.jl:74-76
    lap("omega");
    QIL_TRY(qil_dev_gemm(ctx, dt, 1, 0, m, l0, n, Z, n, Om, n, Y, m));                   // Y = M Omega (:79)
    lap("Y = M Omega");
    // the basis that B = Q^H M and U = Q Uhat are built from (the last QR) must be orthonormal even when the sketch is
    // wider than the rank of M; the power iteration's intermediate bases only stabilise it
    // Deflation of a sketch wider than the operand's numerical rank (r05).  The signals this library is for compress to bonds of
    // 4 ... 20 while the sketch has k + p = 25 ... 133 columns: Y = M Omega then has numerical rank r << l, the columns of Q
    // beyond a basis of range(Y) are orthonormalised rounding noise -- orthogonal to range(M), so their rows of B = Q^H M
    // vanish to rounding and no cutoff >= 1e-24 keeps them -- and every later product and QR carried them at full price (n = 30,
    // rank-4 signal: 6 products of 32768 x 32768 x 133 where 32768 x 32768 x 8 do).  Columns j with |r_jj| <= 1e-13 max |r_ii|
    // leave the basis (R's diagonal is read back: one small transfer); the kept ones are a basis of range(Y) because QR without
    // pivoting gives r_jj = 0 exactly for a column that lies in the span of its predecessors.
    int64_t l = l0;
    {
        void* Rq = nullptr;
        QIL_TRY(qil_ctx_alloc(ctx, (size_t)(l0 * l0) * e, &Rq));
        QIL_TRY(qil_dev_qr_positive(ctx, dt, m, l0, Y, m, Rq, l0, q == 0));             // Q (:83)
        lap("qr(Y)");
        if (l0 <= 1024 && l0 >= 8) {
            std::vector<double> d2((size_t)l0);
            QIL_TRY(qil_dev_diag_abs2(ctx, dt, Rq, l0, l0, d2.data()));
            double dmax = 0;
            for (double v : d2) dmax = std::max(dmax, v);
            std::vector<int64_t> keep;
            for (int64_t j = 0; j < l0; ++j)
                if (d2[(size_t)j] > 1e-26 * dmax && d2[(size_t)j] > 0.0) keep.push_back(j);
            if (keep.empty()) keep.push_back(0);
            // never fewer columns than the caller's mindim keeps (the reference's svd(B; mindim) returns mindim
            // columns whatever the spectrum, rsvd.jl:103-111): dropped columns come back in index order (they are orthonormal
            // columns of Q like the others)
            if ((int64_t)keep.size() < std::min<int64_t>(mindim, l0)) {
                std::vector<char> in((size_t)l0, 0);
                for (int64_t j : keep) in[(size_t)j] = 1;
                for (int64_t j = 0; j < l0 && (int64_t)keep.size() < std::min<int64_t>(mindim, l0); ++j)
                    if (!in[(size_t)j]) keep.push_back(j);
                std::sort(keep.begin(), keep.end());
            }
            if ((int64_t)keep.size() < l0) {
                for (size_t t = 0; t < keep.size(); ++t)                                 // compact the kept columns to the front (usually a prefix already)
                    if (keep[t] != (int64_t)t)
                        QIL_TRY(qil_dev_copy(ctx, static_cast<char*>(Y) + (size_t)t * (size_t)m * e,
                                             static_cast<const char*>(Y) + (size_t)keep[t] * (size_t)m * e, (size_t)m * e));
                l = (int64_t)keep.size();
                if (dbg) fprintf(stderr, "[rsvd] %lld x %lld: sketch of %lld columns has numerical rank %lld: deflated\n", (long long)m, (long long)n, (long long)l0, (long long)l);
            }
        }
        qil_ctx_free(ctx, Rq);
    }
    if (q > 0) QIL_TRY(qil_ctx_alloc(ctx, (size_t)(n * l) * e, &Zq));
    for (int it = 0; it < q; ++it) {                                                    // :86-95
        QIL_TRY(qil_dev_gemm(ctx, dt, 3, 0, n, l, m, Z, n, Y, m, Zq, n));               // M^H Q = conj(Z) Q
        QIL_TRY(qil_dev_qr_positive(ctx, dt, n, l, Zq, n, nullptr, 0));
        QIL_TRY(qil_dev_gemm(ctx, dt, 1, 0, m, l, n, Z, n, Zq, n, Y, m));               // M Qz
        QIL_TRY(qil_dev_qr_positive(ctx, dt, m, l, Y, m, nullptr, 0, it + 1 == q));
        lap("power iteration (2 products + 2 QRs)");
    }
    qil_ctx_free(ctx, Om);
    if (Zq) qil_ctx_free(ctx, Zq);
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(n * l) * e, &Bt));
    QIL_TRY(qil_dev_gemm(ctx, dt, 0, 3, n, l, m, Z, n, Y, m, Bt, n));                   // B^T = Z conj(Q) (:98)
    lap("B^T = Z conj(Q)");
    // B^T = Ub Sb Vbh  =>  Uhat = Vbh^T,  (S V^h)^T = Ub Sb
    int64_t r = 0;
    void *UbS = nullptr, *Vbh = nullptr;
    QIL_TRY(svd_trunc_dev(ctx, dt, n, l, Bt, n, cutoff, true, maxdim, mindim, 1, &r, &UbS, &Vbh, S_out));  // :103
    qil_ctx_free(ctx, Bt);
    void* L = nullptr;
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(r * m) * e, &L));
    lap("svd(B^T)");
    QIL_TRY(qil_dev_gemm(ctx, dt, 0, 1, r, m, l, Vbh, r, Y, m, L, r));                  // U_M^T = Vbh Q^T (:114)
    lap("U^T = Vbh Q^T");
    qil_ctx_free(ctx, Vbh);
    qil_ctx_free(ctx, Y);
    *rank = r;
    *left = L;
    *right = UbS;  // first r columns, ld n
    return QIL_OK;
}

struct EncodeParams {
    int method;
    double cutoff;
    int64_t maxdim, k, p;
    int q;
    uint64_t seed;
    int64_t mindim;
};

// _tensor_to_mps_rsvd recursion (SignalConverters.jl:145-184).  Chunk layout (fastest -> slowest):
// (rb, s_last, ..., s_first, lb) == M^T column-major for the split (lb, left sites | right sites, rb),
// with M's row index ordered (s_mid fastest, ..., s_first, lb) and its column index (rb fastest,
// s_last, ..., s_mid+1).  Both children come out in the same chunk layout with no transposition:
// left = U_M^T (k x rows), right = (S V^h)^T (cols x k).
// Sub-trees of the bisection are independent: with `defer`, the children of the nodes at depth `defer_depth - 1` are not
// descended into but recorded, and the caller runs them concurrently (qil_run_batch_on) -- below the first splits every
// node is a short chain of small factorisations.
struct TtTask {
    void* X;
    int64_t lb, rb, first, last;
};
int compress_tt(qil_context* ctx, int dt, void* X, int64_t lb, int64_t rb, int64_t first, int64_t last,
                const EncodeParams& P, std::vector<void*>& sites, std::vector<int64_t>& dims, int depth = 0,
                std::vector<TtTask>* defer = nullptr, int defer_depth = 0) {
    const size_t e = qil_elem_size(dt);
    if (first == last) {
        void* A = nullptr;
        QIL_TRY(qil_ctx_alloc(ctx, (size_t)(lb * 2 * rb) * e, &A));
        if (dt == QIL_C64)
            QIL_TRY((qil_klaunch<site_from_chunk_k<c64>>(ctx, dim3(nblk(2 * lb * rb)), dim3(256), 0, (const c64*)X, (int)lb, (int)rb, (c64*)A)));
        else
            QIL_TRY((qil_klaunch<site_from_chunk_k<double>>(ctx, dim3(nblk(2 * lb * rb)), dim3(256), 0, (const double*)X, (int)lb, (int)rb, (double*)A)));
        QIL_HIP(hipGetLastError());
        sites[(size_t)first] = A;
        dims[(size_t)first] = lb;
        dims[(size_t)first + 1] = rb;
        qil_ctx_free(ctx, X);
        return QIL_OK;
    }
    const int64_t mid = (first + last + 1) / 2 - 1;  // 1-based (first+last-1) div 2, SignalConverters.jl:161
    const int64_t m = lb << (mid - first + 1), n = rb << (last - mid);
    int64_t r = 0;
    void *L = nullptr, *R = nullptr;
    // explicit cutoff/maxdim override the rsvd defaults (SignalConverters.jl:133)
    QIL_TRY(rsvd_rowmajor(ctx, dt, m, n, X, P.k, P.p, P.q, P.seed, P.cutoff, P.maxdim, P.mindim, &r, &L, &R,
                          nullptr));
    qil_ctx_free(ctx, X);
    if (defer && depth + 1 == defer_depth) {
        defer->push_back(TtTask{L, lb, r, first, mid});
        defer->push_back(TtTask{R, r, rb, mid + 1, last});
        return QIL_OK;
    }
    QIL_TRY(compress_tt(ctx, dt, L, lb, r, first, mid, P, sites, dims, depth + 1, defer, defer_depth));
    QIL_TRY(compress_tt(ctx, dt, R, r, rb, mid + 1, last, P, sites, dims, depth + 1, defer, defer_depth));
    return QIL_OK;
}

// The whole bisection.  The nodes of one level are independent, and so are the sub-trees below any level: the first
// `par_depth` levels run level by level, the nodes of a level concurrently on the context's streams (qil_run_batch_on), and
// the 2^par_depth sub-trees below them concurrently, each as a sequential recursion.  Same kernels on the same operands in
// every order, so the MPS is bit-identical to the sequential recursion (QIL_ENCODE_PAR_DEPTH=0).
static int compress_tt_root(qil_context* ctx, int dt, void* X, int64_t n, const EncodeParams& P, std::vector<void*>& sites,
                            std::vector<int64_t>& dims) {
    const int par_depth = getenv("QIL_ENCODE_PAR_DEPTH") ? atoi(getenv("QIL_ENCODE_PAR_DEPTH")) : 3;   // tuning aid (read per call); 0 = off
    // (a context that is already one slot of a batch -- qil_signal_mps_batch -- encodes sequentially)
    if (par_depth <= 0 || n < 16 || ctx->parent || ctx->lending) return compress_tt(ctx, dt, X, 1, 1, 0, n - 1, P, sites, dims);
    std::vector<TtTask> frontier{TtTask{X, 1, 1, 0, n - 1}};
    int s = QIL_OK;
    for (int level = 0; level <= par_depth && !frontier.empty() && s == QIL_OK; ++level) {
        const bool whole = level == par_depth;               // last pass: every task is a whole sub-tree
        const int64_t nt = (int64_t)frontier.size();
        std::vector<qil_context*> where((size_t)nt, ctx);
        std::vector<std::vector<int64_t>> tdims((size_t)nt, std::vector<int64_t>(dims.size(), 1));
        std::vector<std::vector<TtTask>> kids((size_t)nt);
        auto node = [&](int64_t j, qil_context* work) {
            const TtTask& t = frontier[(size_t)j];
            const int st = whole ? compress_tt(work, dt, t.X, t.lb, t.rb, t.first, t.last, P, sites, tdims[(size_t)j])
                                 : compress_tt(work, dt, t.X, t.lb, t.rb, t.first, t.last, P, sites, tdims[(size_t)j], 0,
                                               &kids[(size_t)j], 1);
            if (st != QIL_OK) (void)qil_ctx_free(work, t.X);     // "unknown block" if the recursion released it already
            return st;
        };
        if (nt == 1)
            s = node(0, ctx);
        else
            s = qil_run_batch_on(
                ctx, nt,
                [&](int64_t j, qil_context* slot) {
                    qil_ctx_transfer(ctx, slot, frontier[(size_t)j].X);
                    where[(size_t)j] = slot;
                },
                node);
        // everything the level produced comes home: finished sites, and the operands of the next level
        std::vector<TtTask> next;
        for (int64_t j = 0; j < nt; ++j) {
            const TtTask& t = frontier[(size_t)j];
            for (int64_t i = t.first; i <= t.last; ++i)
                if (sites[(size_t)i] && (whole || t.first == t.last)) {
                    qil_ctx_transfer(where[(size_t)j], ctx, sites[(size_t)i]);
                    dims[(size_t)i] = tdims[(size_t)j][(size_t)i];
                    dims[(size_t)i + 1] = tdims[(size_t)j][(size_t)i + 1];
                }
            for (const TtTask& c : kids[(size_t)j]) {
                qil_ctx_transfer(where[(size_t)j], ctx, c.X);
                next.push_back(c);
            }
        }
        frontier.swap(next);
    }
    if (s != QIL_OK)
        for (const TtTask& t : frontier) (void)qil_ctx_free(ctx, t.X);
    return s;
}

// _tensor_to_mps_svd (SignalConverters.jl:77-98).  The carried tensor X[(s_n..s_i), alpha] is viewed
// for free as Y[low, (s_i, alpha)] = M^T (L x 2r): the Jacobi SVD rotates its 2r columns, the site
// tensor is V_y^h permuted, and the carry S V^h is U_y S -- already in the layout of the next step.
int svd_sweep(qil_context* ctx, int dt, void* X, int64_t n, const EncodeParams& P, std::vector<void*>& sites,
              std::vector<int64_t>& dims) {
    const size_t e = qil_elem_size(dt);
    int64_t r = 1;
    dims[0] = 1;
    for (int64_t i = 0; i + 1 < n; ++i) {
        const int64_t L = 1LL << (n - 1 - i);
        int64_t k = 0;
        void *UyS = nullptr, *Vyh = nullptr;
        QIL_TRY(svd_trunc_dev(ctx, dt, L, 2 * r, X, L, P.cutoff, true, P.maxdim, 1, 1, &k, &UyS, &Vyh, nullptr));
        qil_ctx_free(ctx, X);
        void* A = nullptr;
        QIL_TRY(qil_ctx_alloc(ctx, (size_t)(r * 2 * k) * e, &A));
        if (dt == QIL_C64)
            QIL_TRY((qil_klaunch<site_from_vh_k<c64>>(ctx, dim3(nblk(2 * r * k)), dim3(256), 0, (const c64*)Vyh, (long long)k, (int)r, (int)k, (c64*)A)));
        else
            QIL_TRY((qil_klaunch<site_from_vh_k<double>>(ctx, dim3(nblk(2 * r * k)), dim3(256), 0, (const double*)Vyh, (long long)k, (int)r, (int)k, (double*)A)));
        QIL_HIP(hipGetLastError());
        qil_ctx_free(ctx, Vyh);
        sites[(size_t)i] = A;
        dims[(size_t)i + 1] = k;
        X = UyS;  // (L x k): X'[(s_n..s_{i+1}), alpha']
        r = k;
    }
    // last site: X[(s_n), alpha] -> A[alpha, s_n, 1]  == chunk layout with rb = 1
    void* A = nullptr;
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(r * 2) * e, &A));
    if (dt == QIL_C64)
        QIL_TRY((qil_klaunch<site_from_chunk_k<c64>>(ctx, dim3(1), dim3(256), 0, (const c64*)X, (int)r, 1, (c64*)A)));
    else
        QIL_TRY((qil_klaunch<site_from_chunk_k<double>>(ctx, dim3(1), dim3(256), 0, (const double*)X, (int)r, 1, (double*)A)));
    QIL_HIP(hipGetLastError());
    qil_ctx_free(ctx, X);
    sites[(size_t)n - 1] = A;
    dims[(size_t)n] = 1;
    return QIL_OK;
}

int signal_mps_impl(qil_context* ctx, const void* x, int64_t len, int dtype, const EncodeParams& P,
                    qil_mps** out) {
    QIL_REQUIRE(ctx && x && out, QIL_EINVAL_ARG, "signal_mps: null argument");
    QIL_REQUIRE(dtype == QIL_F64 || dtype == QIL_C64, QIL_EINVAL_ARG, "signal_mps: unknown dtype %d", dtype);
    QIL_REQUIRE(P.method == QIL_METHOD_SVD || P.method == QIL_METHOD_RSVD, QIL_EINVAL_ARG,
                "tensor_to_mps: unknown method %d. Use :svd or :rsvd.", P.method);
    QIL_REQUIRE(len >= 1, QIL_EINVAL_LENGTH, "signal_mps: empty signal");
    QIL_TRY(qil_ctx_activate(ctx));
    qil_call_scope call_scope(ctx);
    // _array_to_tensor (SignalConverters.jl:16-46): n = round(log2 N), zero-fill, normalise
    const int64_t n = std::max<int64_t>(1, (int64_t)std::llround(std::log2((double)len)));
    const int64_t N = 1LL << n;
    QIL_REQUIRE(len <= N, QIL_EINVAL_LENGTH,
                "_array_to_tensor: Length of signal vector must be a power of 2 (got %lld)", (long long)len);
    // upload the raw samples, zero-fill the tail, then norm + scale ON THE DEVICE (the host never makes
    // a pass over the 2^n samples)
    const int ncomp = dtype == QIL_C64 ? 2 : 1;
    const size_t e = qil_elem_size(dtype);
    void* X = nullptr;
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)N * e, &X));
    // `x` may live on the host or already in HBM (unified addressing resolves the direction).  Samples that are ALREADY on this
    // device and fill the register (an n = 30 signal: 8.6 GB) are not copied first: the norm reads them in place and the scaled
    // copy is written in one pass (r05: copy + norm + scale in place moved 43 GB through HBM, 9 ms; now 26 GB)
    bool in_place = false;
    if (len == N) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, x) == hipSuccess && at.type == hipMemoryTypeDevice && at.device == ctx->device) in_place = true;
        else (void)hipGetLastError();
    }
    if (!in_place) {
        QIL_HIP(hipMemcpyAsync(X, x, (size_t)len * e, hipMemcpyDefault, qil_stream(ctx)));
        if (len < N)
            QIL_TRY(qil_dev_zero(ctx, static_cast<char*>(X) + (size_t)len * e, (size_t)(N - len) * e));
    }
    const void* src = in_place ? x : X;
    void* part = nullptr;
    constexpr int kPartBlocks = 1024;
    QIL_TRY(qil_ctx_alloc(ctx, kPartBlocks * sizeof(double), &part));
    QIL_TRY((qil_klaunch<sumsq_partial_k>(ctx, dim3(kPartBlocks), dim3(256), 0, (const double*)src, (long long)(N * ncomp), (double*)part)));
    std::vector<double> ph(kPartBlocks);
    QIL_TRY(qil_read_back(ctx, ph.data(), part, kPartBlocks * sizeof(double)));   // also completes the upload of caller memory `x`
    qil_ctx_free(ctx, part);
    double ss = 0;
    for (double v : ph) ss += v;                  // fixed order: deterministic
    const double amp = std::sqrt(ss);
    if (!(amp > 0 && std::isfinite(amp))) {
        qil_ctx_free(ctx, X);
        return qil_fail(QIL_EINVAL_ARG, "signal_mps: signal has zero or non-finite norm");
    }
    if (in_place) {
        QIL_TRY((qil_klaunch<scale_copy_k>(ctx, dim3(nblk(N * ncomp)), dim3(256), 0, (const double*)x, (double*)X, (long long)(N * ncomp), 1.0 / amp)));
        QIL_HIP(qil_stream_sync(ctx));             // `x` is caller memory: read completely before anything else can return
    } else
        QIL_TRY((qil_klaunch<scale_inplace_k>(ctx, dim3(nblk(N * ncomp)), dim3(256), 0, (double*)X, (long long)(N * ncomp), 1.0 / amp)));
    QIL_HIP(hipGetLastError());
    std::vector<void*> sites((size_t)n, nullptr);
    std::vector<int64_t> dims((size_t)n + 1, 1);
    int s;
    if (n == 1) {
        // single site: A[1, s, 1] = x_hat[s]
        sites[0] = X;
        s = QIL_OK;
    } else if (P.method == QIL_METHOD_SVD) {
        s = svd_sweep(ctx, dtype, X, n, P, sites, dims);
    } else {
        s = compress_tt_root(ctx, dtype, X, n, P, sites, dims);
    }
    if (s != QIL_OK) {
        for (void* p : sites)
            if (p) qil_ctx_free(ctx, p);
        return s;
    }
    qil_mps* psi = new qil_mps();
    qil_chain_bind(psi, ctx);
    psi->dtype = dtype;
    psi->paired = 0;
    psi->phys_rank = 1;
    psi->dims = dims;
    psi->site.assign(sites.size(), nullptr);
    for (size_t i = 0; i < sites.size(); ++i) qil_chain_adopt(psi, (int64_t)i, sites[i]);
    psi->site_ids.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) psi->site_ids[(size_t)i] = i + 1;
    psi->amplitude = amp;
    *out = psi;
    return QIL_OK;
}


// ---------------------------------------------------------------- fused apply-and-truncate (zip-up) kernels
__device__ __forceinline__ c64 zmul(c64 a, c64 b) { return c64{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ c64 zmul(c64 a, double b) { return c64{a.re * b, a.im * b}; }
__device__ __forceinline__ double zmul(double a, double b) { return a * b; }
__device__ __forceinline__ c64 zadd(c64 a, c64 b) { return c64{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ double zadd(double a, double b) { return a + b; }

// X[r, a, s', beta] = sum_alpha Rm[r, alpha, a] * A[alpha, s', beta]      (Rm index: r + R*(alpha + cl*a))
template <class TO, class TA>
__device__ __forceinline__ void zip_stage1_body(const uint3 blockIdx, const uint3 gridDim, const TO* __restrict__ Rm, const TA* __restrict__ A, TO* __restrict__ X, int R, int Dl,
                           int cl, int cr) {
    const long long total = (long long)R * Dl * 2 * cr;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        long long u = t;
        const int r = (int)(u % R);
        u /= R;
        const int a = (int)(u % Dl);
        u /= Dl;
        const int sp = (int)(u & 1);
        const int beta = (int)(u >> 1);
        const TO* rp = Rm + r + (long long)R * cl * a;
        const TA* ap = A + (long long)cl * (sp + 2LL * beta);
        TO acc{};
        for (int al = 0; al < cl; ++al) acc = zadd(acc, zmul(rp[(long long)R * al], ap[al]));
        X[t] = acc;
    }
}
template <class TO, class TA>
struct zip_stage1_k {
    static constexpr int NT = 1024, MINW = 1;
    template <class... QA>
    static __device__ __forceinline__ void run(const uint3 b, const uint3 g, QA... a) {
        zip_stage1_body<TO, TA>(b, g, a...);
    }
};

// theta[(r, s), (beta, b)] = sum_{a, s'} X[r, a, s', beta] * W[a, s', s, b]
template <class TO, class TW>
__device__ __forceinline__ void zip_stage2_body(const uint3 blockIdx, const uint3 gridDim, const TO* __restrict__ X, const TW* __restrict__ W, TO* __restrict__ theta, int R, int Dl,
                           int Dr, int cr) {
    const long long total = (long long)R * 2 * cr * Dr;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        long long u = t;
        const int r = (int)(u % R);
        u /= R;
        const int s_ = (int)(u & 1);
        u >>= 1;
        const int beta = (int)(u % cr);
        const int b = (int)(u / cr);
        TO acc{};
        for (int sp = 0; sp < 2; ++sp) {
            const TO* xp = X + r + (long long)R * Dl * (sp + 2LL * beta);            // + R * a
            const TW* wp = W + (long long)Dl * (sp + 2 * (s_ + 2LL * b));            // + a
            for (int a = 0; a < Dl; ++a) acc = zadd(acc, zmul(xp[(long long)R * a], wp[a]));
        }
        theta[t] = acc;
    }
}
template <class TO, class TW>
struct zip_stage2_k {
    static constexpr int NT = 1024, MINW = 1;
    template <class... QA>
    static __device__ __forceinline__ void run(const uint3 b, const uint3 g, QA... a) {
        zip_stage2_body<TO, TW>(b, g, a...);
    }
};

template <class TO, class TW, class TA>
int zip_theta(qil_context* ctx, const void* Rm, const void* W, const void* A, int R, int Dl, int Dr, int cl, int cr,
              void** theta_out) {
    void *X = nullptr, *th = nullptr;
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)R * Dl * 2 * cr * sizeof(TO), &X));
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)R * 2 * cr * Dr * sizeof(TO), &th));
    QIL_TRY((qil_klaunch<zip_stage1_k<TO, TA>>(ctx, dim3(nblk((long long)R * Dl * 2 * cr)), dim3(256), 0, (const TO*)Rm, (const TA*)A, (TO*)X, R, Dl, cl, cr)));
    QIL_TRY((qil_klaunch<zip_stage2_k<TO, TW>>(ctx, dim3(nblk((long long)R * 2 * cr * Dr)), dim3(256), 0, (const TO*)X, (const TW*)W, (TO*)th, R, Dl, Dr, cr)));
    QIL_HIP(hipGetLastError());
    qil_ctx_free(ctx, X);
    *theta_out = th;
    return QIL_OK;
}

// Z[(alpha, a), (s, r')] = sum_{s', b} Y[alpha, s', b, r'] W[a, s', s, b]     (right environment of the fit sweep)
//   Y index (alpha + cl s') + 2 cl (b + Dr r'),  W index a + Dl (s' + 2 (s + 2 b)),  Z index (alpha + cl a) + cl Dl (s + 2 r')
template <class T>
__device__ __forceinline__ void fit_env_stage_body(const uint3 blockIdx, const uint3 gridDim, const T* __restrict__ Y, const T* __restrict__ W, T* __restrict__ Z, int cl, int Dl,
                              int Dr, int rp) {
    const long long total = (long long)cl * Dl * 2 * rp;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        long long u = t;
        const int alpha = (int)(u % cl);
        u /= cl;
        const int a = (int)(u % Dl);
        u /= Dl;
        const int s_ = (int)(u & 1);
        const int r = (int)(u >> 1);
        T acc{};
        for (int sp = 0; sp < 2; ++sp) {
            const T* yp = Y + (alpha + (long long)cl * sp) + 2LL * cl * Dr * r;        // + 2 cl b
            const T* wp = W + a + (long long)Dl * (sp + 2 * s_);                       // + 4 Dl b
            for (int b = 0; b < Dr; ++b) acc = zadd(acc, zmul(yp[2LL * cl * b], wp[4LL * Dl * b]));
        }
        Z[t] = acc;
    }
}
template <class T>
struct fit_env_stage_k {
    static constexpr int NT = 1024, MINW = 1;
    template <class... QA>
    static __device__ __forceinline__ void run(const uint3 b, const uint3 g, QA... a) {
        fit_env_stage_body<T>(b, g, a...);
    }
};

__device__ __forceinline__ void widen_f64_body(const uint3 blockIdx, const uint3 gridDim, const double* __restrict__ in, c64* __restrict__ out, long long n) {
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x)
        out[t] = c64{in[t], 0.0};
}
struct widen_f64_k {
    static constexpr int NT = 1024, MINW = 1;
    template <class... QA>
    static __device__ __forceinline__ void run(const uint3 b, const uint3 g, QA... a) {
        widen_f64_body(b, g, a...);
    }
};

}  // namespace

// ---------------------------------------------------------------- exported
extern "C" int qil_canonicalize(qil_mps* psi, int direction, int64_t center, double cutoff, int64_t maxdim) {
    QIL_REQUIRE(psi, QIL_EINVAL_ARG, "canonicalize!: null handle");
    QIL_REQUIRE(direction == QIL_DIR_RIGHT || direction == QIL_DIR_LEFT, QIL_EINVAL_ARG,
                "Direction must be :right or :left");
    QIL_TRY(qil_ctx_activate(psi->ctx));
    qil_call_scope call_scope(psi->ctx);
    // the ZTMPS method forwards `center` unchanged to the 2n-site chain (mps.jl:880-881)
    return canonicalize_impl(psi, direction, center, cutoff, maxdim);
}

static int compress_impl(qil_mps* psi, int64_t maxdim, double tol, int sweeps, bool right_canonical);

// Fused apply-and-truncate: compress!(apply(W, psi); maxdim, tol, sweeps) WITHOUT ever writing the
// (D chi)^2 product tensors.
//   1. zip-up: psi is brought to right-canonical gauge (a copy); one left-to-right sweep carries the left
//      environment L[r, alpha, a] = <phi_{<i} | W psi_{<i}> and per site forms only
//          theta[(r, s), (beta, b)] = sum L[r, alpha, a] W[a, s', s, b] A[alpha, s', beta]      (2 r x D chi)
//      whose truncated SVD (cap zip_maxdim = 2 maxdim, 100x tighter cutoff) gives the site phi_i (U) and the next
//      environment (S V^h).  Its truncations see only psi's gauge, not the operator's right part: on flat-spectrum
//      operands they discard weight the exact route keeps (1e-2 in 5 of 24 random products).
//   2. one variational sweep right to left repairs that: with the left environments of step 1 and right environments
//      R[(beta, b), r'] = <phi_{>i} | W psi_{>i}> built on the way, every site is replaced by the best tensor given the
//      others,  phi_i = L_i A_i W_i R_i  (two small contraction kernels + one GEMM), and re-gauged by a thin QR.  No SVD,
//      nothing of size (D chi)^2.  After it the state agrees with compress!(apply(W, psi)) of the CPU oracle to rounding on
//      22 of the 24 random products and within the truncation's own error on the other two (numpy prototype and
//      tests/test_gpu_parity.py::test_apply_compress_random_products_against_oracle).
//   3. the exact-gauge compress! (src/mps.jl:913-973) fixes the reference's post-conditions (bonds <= maxdim by the
//      ITensors rule, unit norm in the tensors, norm moved into `amplitude`).
// Cost O(n r D chi (D + chi + r)) instead of O(n (D chi)^3).
// `ctx` = working context (stream + pool) of the call: psi's own, or a worker of it (qil_apply_compress_batch); the
// operands are only read
static int apply_compress_on(qil_context* ctx, const qil_mpo* W, const qil_mps* psi, int64_t maxdim, double tol, int sweeps,
                             int64_t zip_maxdim, qil_mps** out) {
    QIL_REQUIRE(W && psi && out, QIL_EINVAL_ARG, "apply_compress: null argument");
    QIL_REQUIRE(W->ctx == psi->ctx, QIL_EINVAL_ARG, "apply: MPO and MPS belong to different contexts");
    QIL_REQUIRE(W->paired == psi->paired, QIL_EINVAL_ARG, "apply: cannot mix paired and single-register operands");
    QIL_REQUIRE(W->n() == psi->n(), QIL_EINVAL_LENGTH,
                "apply: MPO and MPS must have the same number of sites. Found length(W)=%lld, length(psi)=%lld",
                (long long)W->n(), (long long)psi->n());
    QIL_REQUIRE(W->site_ids == psi->site_ids, QIL_EINVAL_SITES, "apply: MPO and MPS must have the same site indices.");
    const int64_t N = psi->n();
    QIL_REQUIRE(N >= 2, QIL_EDOMAIN, "SignalMPS must have at least 2 sites.");
    QIL_REQUIRE(sweeps >= 1, QIL_EINVAL_ARG, "compress!: sweeps must be >= 1");
    QIL_TRY(qil_ctx_activate(ctx));
    qil_call_scope call_scope(ctx);
    if (maxdim <= 0) maxdim = kNoCap;
    // intermediate bond cap of the zip-up and the variational sweep: maxdim + 16.  With the variational sweep behind it the zip-up
    // only has to deliver a basis that contains the kept space: an oversampling of 16 columns gives the same verdicts as 2 maxdim
    // (r01), max(1.5 maxdim, maxdim + 16) (r02/r03) -- 200 random flat-spectrum products with maxdim 8 ... 128 each, 0 bad
    // (tools/_fuzz_product_compress.py, QIL_FUZZ_ZIP=plus16) -- and the oracle tests, while an oversampling of 4 (1.5 maxdim at
    // maxdim = 8) fails 2 of 200.  Bench product, maxdim 64: cap 96 117 ms, cap 80 108.6 ms (MEASUREMENTS R04.7).
    // The evidence covers maxdim <= 128 with flat spectra only: beyond that a relative floor of 1.125 maxdim
    // keeps the oversampling proportional (maxdim 256: 288, 512: 576) until a slowly-decaying case at that size is fuzzed.
    const int64_t zip_over = 16;
    if (zip_maxdim <= 0) zip_maxdim = maxdim > kNoCap / 2 ? kNoCap : std::max(maxdim + zip_over, maxdim + (maxdim + 7) / 8);
    const double cutoff = tol * tol / ((double)(N - 1) * sweeps);
    const double zip_cutoff = cutoff * 1e-2;
    const bool wc = W->dtype == QIL_C64, ac = psi->dtype == QIL_C64;
    const int odt = (wc || ac) ? QIL_C64 : QIL_F64;
    const size_t e = qil_elem_size(odt);
    // right-canonical copy of psi: the zip's truncations then see (nearly) orthonormal environments
    qil_mps* phi = nullptr;
    QIL_TRY(qil_mps_clone_to(ctx, psi, &phi));
    qil_mps* res = nullptr;
    std::vector<void*> tmp;                          // every pool block this call owns outside a handle
    std::vector<void*> Asite((size_t)N), Wsite((size_t)N), Lenv((size_t)N, nullptr);
    std::vector<int> Rdim((size_t)N + 1, 1);
    auto cleanup = [&](int code) {
        for (void* p : tmp) qil_ctx_free(ctx, p);
        if (phi) qil_mps_destroy(phi);
        if (code != QIL_OK && res) qil_mps_destroy(res);
        return code;
    };
    auto take = [&](size_t bytes, void** p) {
        int s = qil_ctx_alloc(ctx, bytes, p);
        if (s == QIL_OK) tmp.push_back(*p);
        return s;
    };
    auto drop = [&](void* p) {
        for (size_t t = tmp.size(); t-- > 0;)
            if (tmp[t] == p) {
                tmp.erase(tmp.begin() + (long)t);
                break;
            }
        qil_ctx_free(ctx, p);
    };
    auto forget = [&](void* p) {                     // ownership moved into a handle
        for (size_t t = tmp.size(); t-- > 0;)
            if (tmp[t] == p) {
                tmp.erase(tmp.begin() + (long)t);
                break;
            }
    };
    int st = canonicalize_impl(phi, QIL_DIR_LEFT, 0, 0.0, kNoCap, true);      // exact thin-QR gauge: psi loses nothing
    if (st != QIL_OK) return cleanup(st);
    // operands in the output dtype (a real operand of a complex product is widened once: KB..MB)
    for (int64_t i = 0; i < N && st == QIL_OK; ++i) {
        Asite[(size_t)i] = phi->site[(size_t)i];
        Wsite[(size_t)i] = W->site[(size_t)i];
        if (odt == QIL_C64 && !ac) {
            const long long ne = phi->site_elems(i);
            void* p = nullptr;
            if ((st = take((size_t)ne * e, &p)) != QIL_OK) break;
            QIL_TRY((qil_klaunch<widen_f64_k>(ctx, dim3(nblk(ne)), dim3(256), 0, (const double*)phi->site[(size_t)i], (c64*)p, ne)));
            Asite[(size_t)i] = p;
        }
        if (odt == QIL_C64 && !wc) {
            const long long ne = W->site_elems(i);
            void* p = nullptr;
            if ((st = take((size_t)ne * e, &p)) != QIL_OK) break;
            QIL_TRY((qil_klaunch<widen_f64_k>(ctx, dim3(nblk(ne)), dim3(256), 0, (const double*)W->site[(size_t)i], (c64*)p, ne)));
            Wsite[(size_t)i] = p;
        }
    }
    if (st != QIL_OK) return cleanup(st);
    auto theta_of = [&](int64_t i, void** theta) {
        const int Dl = (int)W->dims[(size_t)i], Dr = (int)W->dims[(size_t)i + 1];
        const int cl = (int)phi->dims[(size_t)i], cr = (int)phi->dims[(size_t)i + 1];
        int s2 = odt == QIL_F64
                     ? zip_theta<double, double, double>(ctx, Lenv[(size_t)i], Wsite[(size_t)i], Asite[(size_t)i],
                                                         Rdim[(size_t)i], Dl, Dr, cl, cr, theta)
                     : zip_theta<c64, c64, c64>(ctx, Lenv[(size_t)i], Wsite[(size_t)i], Asite[(size_t)i], Rdim[(size_t)i], Dl,
                                                Dr, cl, cr, theta);
        if (s2 == QIL_OK) tmp.push_back(*theta);
        return s2;
    };
    auto one = [&](void** p) {                       // 1 x 1 environment
        int s2 = take(e, p);
        const double v[2] = {1.0, 0.0};
        if (s2 == QIL_OK && hipMemcpyAsync(*p, v, e, hipMemcpyHostToDevice, qil_stream(ctx)) != hipSuccess)
            s2 = qil_fail(QIL_EHIP, "apply_compress: upload failed");
        if (s2 == QIL_OK && qil_stream_sync(ctx) != hipSuccess) s2 = qil_fail(QIL_EHIP, "sync failed");
        return s2;
    };
    res = new qil_mps();
    qil_chain_bind(res, ctx);
    res->dtype = odt;
    res->paired = psi->paired;
    res->phys_rank = 1;
    res->dims.assign((size_t)N + 1, 1);
    res->site.assign((size_t)N, nullptr);
    res->site_ids = psi->site_ids;
    res->amplitude = psi->amplitude;
    const bool fdbg = getenv("QIL_FUSED_DEBUG") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!fdbg) return;
        (void)qil_stream_sync(ctx);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[apply_compress] %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    // sketch matrix for capped bonds (one for the whole call: any sub-block of a Gaussian matrix is Gaussian)
    static const bool sketch = true;   // tuning aid
    void* Om = nullptr;
    int64_t om_ld = 1;
    if (sketch && zip_maxdim < kNoCap / 2) {
        for (int64_t i = 0; i + 1 < N; ++i) om_ld = std::max<int64_t>(om_ld, phi->dims[(size_t)i + 1] * W->dims[(size_t)i + 1]);
        if (zip_maxdim < om_ld) {
            if ((st = take((size_t)(om_ld * zip_maxdim) * e, &Om)) != QIL_OK) return cleanup(st);
            if ((st = qil_dev_fill_normal(ctx, odt, Om, om_ld * zip_maxdim, 0x51b0e7c5ull, 1.0)) != QIL_OK) return cleanup(st);
        }
    }
    // ---- 1. zip-up, keeping the left environments
    if ((st = one(&Lenv[0])) != QIL_OK) return cleanup(st);
    for (int64_t i = 0; i < N; ++i) {
        const int Dr = (int)W->dims[(size_t)i + 1], cr = (int)phi->dims[(size_t)i + 1];
        const int R = Rdim[(size_t)i];
        void* theta = nullptr;
        if ((st = theta_of(i, &theta)) != QIL_OK) return cleanup(st);
        if (i + 1 == N) {                       // last site: theta is (R, 2, 1)
            forget(theta);
            qil_chain_adopt(res, i, theta);
            res->dims[(size_t)i] = R;
            break;
        }
        int64_t r = 0;
        void *U = nullptr, *SV = nullptr;
        const int64_t rows = 2LL * R, Pc = (int64_t)cr * Dr;
        if (sketch && zip_maxdim < std::min(rows, Pc)) {
            // capped bond: an orthonormal basis of theta Omega (Omega: Pc x zip_maxdim, seeded Gaussian) instead of
            // theta's truncated SVD -- the variational sweep below re-optimises every site anyway, and with it the
            // sketched zip-up reproduces the oracle's compress!(apply) on all 24 random products exactly as the SVD one
            // does (numpy prototype, then tests); one GEMM + one thin QR + one GEMM instead of a 2r x D chi SVD
            r = zip_maxdim;
            void* Y = nullptr;
            if ((st = qil_ctx_alloc(ctx, (size_t)(rows * r) * e, &Y)) != QIL_OK) return cleanup(st);
            st = qil_dev_gemm(ctx, odt, 0, 0, rows, r, Pc, theta, rows, Om, om_ld, Y, rows);
            if (st == QIL_OK) st = qil_dev_qr_positive(ctx, odt, rows, r, Y, rows, nullptr, 0, true);
            if (st == QIL_OK) st = qil_ctx_alloc(ctx, (size_t)(r * Pc) * e, &SV);
            if (st == QIL_OK) st = qil_dev_gemm(ctx, odt, odt == QIL_C64 ? 2 : 1, 0, r, Pc, rows, Y, rows, theta, rows, SV, r);
            if (st != QIL_OK) return cleanup(st);
            U = Y;
        } else {
            st = svd_trunc_dev(ctx, odt, rows, Pc, theta, rows, zip_cutoff, true, zip_maxdim, 1, 2, &r, &U, &SV, nullptr);
            if (st != QIL_OK) return cleanup(st);
        }
        drop(theta);
        qil_chain_adopt(res, i, U);             // [R, s, r]
        res->dims[(size_t)i] = R;
        res->dims[(size_t)i + 1] = r;
        tmp.push_back(SV);
        Lenv[(size_t)i + 1] = SV;               // [r, (beta, b)] == L[r, alpha, a] of the next site
        Rdim[(size_t)i + 1] = (int)r;
    }
    lap("gauge + zip-up");
    // ---- 2. variational sweep right to left
    void* Renv = nullptr;                       // [(beta, b), r']
    int rp = 1;
    if ((st = one(&Renv)) != QIL_OK) return cleanup(st);
    for (int64_t i = N - 1; i >= 0; --i) {
        const int Dl = (int)W->dims[(size_t)i], Dr = (int)W->dims[(size_t)i + 1];
        const int cl = (int)phi->dims[(size_t)i], cr = (int)phi->dims[(size_t)i + 1];
        const int R = Rdim[(size_t)i];
        void *theta = nullptr, *T = nullptr;
        if ((st = theta_of(i, &theta)) != QIL_OK) return cleanup(st);
        if ((st = take((size_t)2 * R * rp * e, &T)) != QIL_OK) return cleanup(st);
        st = qil_dev_gemm(ctx, odt, 0, 0, 2LL * R, rp, (int64_t)cr * Dr, theta, 2LL * R, Renv, (int64_t)cr * Dr, T, 2LL * R);
        if (st != QIL_OK) return cleanup(st);
        drop(theta);
        forget(T);
        if ((st = qil_chain_set_site(res, i, T, R, rp)) != QIL_OK) return cleanup(st);
        if (i == 0) break;
        if ((st = gauge_site_left(res, i, 0.0, kNoCap, true)) != QIL_OK) return cleanup(st);
        const int K = (int)res->dims[(size_t)i];
        void *Y = nullptr, *Z = nullptr, *Rn = nullptr;
        if ((st = take((size_t)2 * cl * Dr * rp * e, &Y)) != QIL_OK) return cleanup(st);
        st = qil_dev_gemm(ctx, odt, 0, 0, 2LL * cl, (int64_t)Dr * rp, cr, Asite[(size_t)i], 2LL * cl, Renv, cr, Y, 2LL * cl);
        if (st != QIL_OK) return cleanup(st);
        if ((st = take((size_t)cl * Dl * 2 * rp * e, &Z)) != QIL_OK) return cleanup(st);
        if (odt == QIL_F64)
            QIL_TRY((qil_klaunch<fit_env_stage_k<double>>(ctx, dim3(nblk((long long)cl * Dl * 2 * rp)), dim3(256), 0, (const double*)Y, (const double*)Wsite[(size_t)i], (double*)Z, cl, Dl, Dr, rp)));
        else
            QIL_TRY((qil_klaunch<fit_env_stage_k<c64>>(ctx, dim3(nblk((long long)cl * Dl * 2 * rp)), dim3(256), 0, (const c64*)Y, (const c64*)Wsite[(size_t)i], (c64*)Z, cl, Dl, Dr, rp)));
        if (hipGetLastError() != hipSuccess) return cleanup(qil_fail(QIL_EHIP, "apply_compress: launch failed"));
        if ((st = take((size_t)cl * Dl * K * e, &Rn)) != QIL_OK) return cleanup(st);
        // R_{i-1}[(alpha, a), k] = sum_{s, r'} Z[(alpha, a), (s, r')] conj(phi_i[k, (s, r')])
        st = qil_dev_gemm(ctx, odt, 0, 2, (int64_t)cl * Dl, K, 2LL * rp, Z, (int64_t)cl * Dl, res->site[(size_t)i], K, Rn,
                          (int64_t)cl * Dl);
        if (st != QIL_OK) return cleanup(st);
        drop(Y);
        drop(Z);
        drop(Renv);
        Renv = Rn;
        rp = K;
    }
    lap("variational sweep");
    (void)cleanup(QIL_OK);
    phi = nullptr;
    tmp.clear();
    // ---- 3. exact-gauge truncation
    st = compress_impl(res, maxdim, tol, sweeps, true);        // the variational sweep left res right-canonical
    lap("compress!");
    if (st != QIL_OK) {
        qil_mps_destroy(res);
        return st;
    }
    *out = res;
    return QIL_OK;
}

extern "C" int qil_apply_compress(const qil_mpo* W, const qil_mps* psi, int64_t maxdim, double tol, int sweeps,
                                  int64_t zip_maxdim, qil_mps** out) {
    QIL_REQUIRE(W && psi && out, QIL_EINVAL_ARG, "apply_compress: null argument");
    return apply_compress_on(psi->ctx, W, psi, maxdim, tol, sweeps, zip_maxdim, out);
}

extern "C" int qil_apply_compress_batch(const qil_mpo* const* Ws, const qil_mps* const* psis, int64_t nb, int64_t maxdim,
                                        double tol, int sweeps, int64_t zip_maxdim, qil_mps** outs) {
    QIL_REQUIRE(nb >= 0 && ((Ws && psis && outs) || nb == 0), QIL_EINVAL_ARG, "apply_compress_batch: null argument");
    if (nb == 0) return QIL_OK;
    for (int64_t j = 0; j < nb; ++j) {
        outs[j] = nullptr;
        QIL_REQUIRE(Ws[j] && psis[j], QIL_EINVAL_ARG, "apply_compress_batch: item %lld is null", (long long)j);
        QIL_REQUIRE(Ws[j]->ctx && Ws[j]->ctx == psis[0]->ctx && psis[j]->ctx == psis[0]->ctx, QIL_EINVAL_ARG,
                    "apply_compress_batch: item %lld lives in another context", (long long)j);
    }
    const int st = qil_run_batch_on(psis[0]->ctx, nb, nullptr, [&](int64_t j, qil_context* work) {
        return apply_compress_on(work, Ws[j], psis[j], maxdim, tol, sweeps, zip_maxdim, &outs[j]);
    });
    if (st != QIL_OK)                                   // all or nothing: a failed batch hands out no handles
        for (int64_t j = 0; j < nb; ++j) {
            if (outs[j]) qil_mps_destroy(outs[j]);
            outs[j] = nullptr;
        }
    return st;
}

static int compress_impl(qil_mps* psi, int64_t maxdim, double tol, int sweeps, bool right_canonical);

extern "C" int qil_compress(qil_mps* psi, int64_t maxdim, double tol, int sweeps) {
    return compress_impl(psi, maxdim, tol, sweeps, false);
}

// right_canonical: the caller guarantees every site but the first is a right isometry (the fused apply-and-truncate
// after its variational sweep).  canonicalize!(:left) of such a state can neither truncate nor change anything but the
// gauge (all its singular values are 1), so it is not run.
static int compress_impl(qil_mps* psi, int64_t maxdim, double tol, int sweeps, bool right_canonical) {
    QIL_REQUIRE(psi, QIL_EINVAL_ARG, "compress!: null handle");
    const int64_t N = psi->n();
    QIL_REQUIRE(N >= 2, QIL_EDOMAIN, "SignalMPS must have at least 2 sites.");   // mps.jl:918
    QIL_REQUIRE(sweeps >= 1, QIL_EINVAL_ARG, "compress!: sweeps must be >= 1");
    qil_context* ctx = psi->ctx;
    QIL_TRY(qil_ctx_activate(ctx));
    qil_call_scope call_scope(ctx);
    const int dt = psi->dtype;
    const double cutoff = tol * tol / ((double)(N - 1) * sweeps);               // mps.jl:920
    if (!right_canonical) QIL_TRY(canonicalize_impl(psi, QIL_DIR_LEFT, 0, 1e-12, kNoCap));            // mps.jl:923
    // The reference truncates the two-site tensor psi[j] psi[j+1] (mps.jl:929, :946).  In canonical gauge --
    // which every step of the sweep maintains -- that tensor has the same singular values and the same kept
    // subspace as the single site next to the orthogonality centre (its neighbour is an isometry), so each
    // sweep is run as a truncating gauge sweep on one-site matrices: (2 chi_l x chi) instead of
    // (2 chi_l x 2 chi_r), no theta product, and the Jacobi SVD rotates half as many columns.
    for (int sw = 0; sw < sweeps; ++sw) {
        QIL_TRY(canonicalize_impl(psi, QIL_DIR_RIGHT, 0, cutoff, maxdim));      // L -> R: U | S V   (mps.jl:927-942)
        QIL_TRY(canonicalize_impl(psi, QIL_DIR_LEFT, 0, cutoff, maxdim));       // R -> L: U S | V   (mps.jl:944-959)
    }
    // mps.jl:963 re-gauges once more (canonicalize!(:left), cutoff 1e-12).  The sweep that just ended left every site
    // but the first an isometry (all its singular values are 1), so that pass can neither truncate nor change anything
    // but the gauge -- it is not run.
    double nrm = 0;
    QIL_TRY(qil_norm(psi, &nrm));                                               // mps.jl:967-971
    if (nrm != 0) {
        psi->amplitude *= nrm;
        const double inv = 1.0 / nrm;
        std::vector<double> s((size_t)psi->dims[1], inv);
        QIL_TRY(qil_dev_scale(ctx, dt, 1, 2 * psi->dims[0], psi->dims[1], psi->site[0], 2 * psi->dims[0], s.data()));
    }
    return QIL_OK;
}

// zip_to_compress_mpo over a whole MPO (dt_transformer.jl:167-288; called on the MPO x MPO product in
// zt_transformer.jl:103-104): an exact gauge sweep towards one end, then a truncating sweep back.  In the gauged
// chain the reference's two-site core has the singular values of the single site next to the centre, so the
// truncating sweep is the same one-site sweep compress! uses.  direction 0 = "down" (gauge left -> right, truncate
// right -> left), 1 = "up" (mirror image).  The gauge sweep runs with cutoff 0: only exactly-zero singular
// values (rank-deficient bonds of a product) are dropped.
extern "C" int qil_mpo_compress(qil_mpo* W, int direction, double cutoff, int64_t maxdim) {
    QIL_REQUIRE(W, QIL_EINVAL_ARG, "mpo_compress: null argument");
    QIL_REQUIRE(direction == 0 || direction == 1, QIL_EINVAL_ARG,
                "zip_to_compress_mpo: unknown direction %d (0 = down, 1 = up)", direction);
    QIL_REQUIRE(cutoff >= 0, QIL_EINVAL_ARG, "mpo_compress: cutoff must be >= 0");
    if (W->n() < 2) return QIL_OK;
    QIL_TRY(qil_ctx_activate(W->ctx));
    qil_call_scope call_scope(W->ctx);
    if (maxdim <= 0) maxdim = kNoCap;
    const int gauge = direction == 0 ? QIL_DIR_RIGHT : QIL_DIR_LEFT;
    const int trunc = direction == 0 ? QIL_DIR_LEFT : QIL_DIR_RIGHT;
    // exact gauge pass by thin QRs, like the reference's (an SVD here would drop exactly-zero singular values and change
    // the bond dimensions the truncating pass starts from)
    QIL_TRY(canonicalize_impl(W, gauge, 0, 0.0, kNoCap, true));
    QIL_TRY(canonicalize_impl(W, trunc, 0, cutoff, maxdim));
    return QIL_OK;
}

extern "C" int qil_compress_batch(qil_mps* const* items, int64_t nb, int64_t maxdim, double tol, int sweeps) {
    QIL_REQUIRE(nb >= 0 && (items || nb == 0), QIL_EINVAL_ARG, "compress_batch: null item array");
    std::vector<qil_chain*> chains(items, items + nb);
    return qil_run_batch(chains.data(), nb, [&](qil_chain* c) {
        return compress_impl(static_cast<qil_mps*>(c), maxdim, tol, sweeps, false);
    });
}

extern "C" int qil_mpo_compress_batch(qil_mpo* const* items, int64_t nb, int direction, double cutoff, int64_t maxdim) {
    QIL_REQUIRE(nb >= 0 && (items || nb == 0), QIL_EINVAL_ARG, "mpo_compress_batch: null item array");
    std::vector<qil_chain*> chains(items, items + nb);
    return qil_run_batch(chains.data(), nb, [&](qil_chain* c) {
        return qil_mpo_compress(static_cast<qil_mpo*>(c), direction, cutoff, maxdim);
    });
}

extern "C" int qil_svd_trunc(qil_context* ctx, const void* A, int64_t m, int64_t n, int dtype, double cutoff,
                             int64_t maxdim, int64_t mindim, int64_t* rank, void* U, double* S, void* Vh) {
    QIL_REQUIRE(ctx && A && rank && U && S && Vh, QIL_EINVAL_ARG, "svd: null argument");
    QIL_REQUIRE(m >= 1 && n >= 1, QIL_EEMPTY, "svd: empty matrix");
    QIL_TRY(qil_ctx_activate(ctx));
    qil_call_scope call_scope(ctx);
    const size_t e = qil_elem_size(dtype);
    void* dA = nullptr;
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(m * n) * e, &dA));
    QIL_HIP(hipMemcpyAsync(dA, A, (size_t)(m * n) * e, hipMemcpyHostToDevice, qil_stream(ctx)));
    QIL_HIP(qil_stream_sync(ctx));
    int64_t r = 0;
    void *dU = nullptr, *dVh = nullptr;
    std::vector<double> Sv;
    QIL_TRY(svd_trunc_dev(ctx, dtype, m, n, dA, m, cutoff, cutoff >= 0, maxdim, mindim, 0, &r, &dU, &dVh, &Sv));
    QIL_HIP(hipMemcpyAsync(U, dU, (size_t)(m * r) * e, hipMemcpyDeviceToHost, qil_stream(ctx)));
    QIL_TRY(qil_read_back(ctx, Vh, dVh, (size_t)(r * n) * e));
    for (int64_t i = 0; i < r; ++i) S[i] = Sv[(size_t)i];
    *rank = r;
    qil_ctx_free(ctx, dA);
    qil_ctx_free(ctx, dU);
    qil_ctx_free(ctx, dVh);
    return QIL_OK;
}

// rsvd(A, Linds...; kwargs) on a host operand A (m x n, column-major): src/linalg/rsvd.jl:38-121.
extern "C" int qil_rsvd(qil_context* ctx, const void* A, int64_t m, int64_t n, int dtype, int64_t k, int64_t p,
                        int q, uint64_t seed, double cutoff, int64_t maxdim, int64_t mindim, int64_t* rank,
                        void* U, double* S, void* Vh) {
    QIL_REQUIRE(ctx && rank && U && S && Vh, QIL_EINVAL_ARG, "rsvd: null argument");
    QIL_REQUIRE(m >= 1 && n >= 1 && A, QIL_EEMPTY,
                "In `rsvd`, left or right index set is empty.");               // rsvd.jl:56-60
    QIL_REQUIRE(k >= 1 && p >= 0 && q >= 0, QIL_EINVAL_ARG, "rsvd: need k >= 1, p >= 0, q >= 0");
    QIL_TRY(qil_ctx_activate(ctx));
    qil_call_scope call_scope(ctx);
    const size_t e = qil_elem_size(dtype);
    void *dA = nullptr, *Z = nullptr;
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(m * n) * e, &dA));
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(m * n) * e, &Z));
    QIL_HIP(hipMemcpyAsync(dA, A, (size_t)(m * n) * e, hipMemcpyHostToDevice, qil_stream(ctx)));
    QIL_HIP(qil_stream_sync(ctx));
    QIL_TRY(qil_dev_transpose(ctx, dtype, 0, m, n, dA, m, Z, n));               // Z = A^T ("A row-major")
    qil_ctx_free(ctx, dA);
    if (maxdim <= 0) maxdim = k;                                                 // maxdim = k default (rsvd.jl:47)
    int64_t r = 0;
    void *L = nullptr, *R = nullptr;
    std::vector<double> Sv;
    QIL_TRY(rsvd_rowmajor(ctx, dtype, m, n, Z, k, p, q, seed, cutoff, maxdim, mindim < 1 ? 1 : mindim, &r, &L, &R,
                          &Sv));
    qil_ctx_free(ctx, Z);
    // L = U^T (r x m); R = (S V^h)^T (n x r).  Return U (m x r) and V^h (r x n) = (R diag(1/S))^T.
    std::vector<double> inv((size_t)r);
    for (int64_t i = 0; i < r; ++i) inv[(size_t)i] = Sv[(size_t)i] > 0 ? 1.0 / Sv[(size_t)i] : 0.0;
    QIL_TRY(qil_dev_scale(ctx, dtype, 1, n, r, R, n, inv.data()));
    void *dU = nullptr, *dVh = nullptr;
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(m * r) * e, &dU));
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(r * n) * e, &dVh));
    QIL_TRY(qil_dev_transpose(ctx, dtype, 0, r, m, L, r, dU, m));
    QIL_TRY(qil_dev_transpose(ctx, dtype, 0, n, r, R, n, dVh, r));
    QIL_HIP(hipMemcpyAsync(U, dU, (size_t)(m * r) * e, hipMemcpyDeviceToHost, qil_stream(ctx)));
    QIL_TRY(qil_read_back(ctx, Vh, dVh, (size_t)(r * n) * e));
    for (int64_t i = 0; i < r; ++i) S[i] = Sv[(size_t)i];
    *rank = r;
    qil_ctx_free(ctx, L);
    qil_ctx_free(ctx, R);
    qil_ctx_free(ctx, dU);
    qil_ctx_free(ctx, dVh);
    return QIL_OK;
}

extern "C" int qil_signal_mps(qil_context* ctx, const void* x, int64_t len, int dtype, int method, double cutoff,
                              int64_t maxdim, int64_t k, int64_t p, int q, uint64_t seed, int64_t mindim,
                              qil_mps** out) {
    EncodeParams P{method, cutoff, maxdim <= 0 ? kNoCap : maxdim, k, p, q, seed, mindim < 1 ? 1 : mindim};
    return signal_mps_impl(ctx, x, len, dtype, P, out);
}

extern "C" int qil_signal_ztmps(qil_context* ctx, const void* x, int64_t len, int dtype, int method,
                                double cutoff, int64_t maxdim, int64_t k, int64_t p, int q, uint64_t seed,
                                int64_t mindim, qil_mps** out) {
    QIL_REQUIRE(ctx && out, QIL_EINVAL_ARG, "signal_ztmps: null argument");
    qil_call_scope call_scope(ctx);
    EncodeParams P{method, cutoff, maxdim <= 0 ? kNoCap : maxdim, k, p, q, seed, mindim < 1 ? 1 : mindim};
    qil_mps* sig = nullptr;
    QIL_TRY(signal_mps_impl(ctx, x, len, dtype, P, &sig));                        // SignalConverters.jl:251
    const int64_t n = sig->n();
    const size_t e = qil_elem_size(dtype);
    qil_mps* zt = new qil_mps();
    qil_chain_bind(zt, ctx);
    zt->dtype = dtype;
    zt->paired = 1;
    zt->phys_rank = 1;
    zt->dims.assign((size_t)(2 * n) + 1, 1);
    zt->site.assign((size_t)(2 * n), nullptr);
    zt->site_ids.resize((size_t)(2 * n));
    for (int64_t i = 0; i < 2 * n; ++i) zt->site_ids[(size_t)i] = i + 1;
    zt->amplitude = sig->amplitude;
    int status = QIL_OK;
    for (int64_t i = 0; i < n && status == QIL_OK; ++i) {                          // :261-276
        const int64_t cl = sig->dims[(size_t)i], cr = sig->dims[(size_t)i + 1];
        void* T = nullptr;
        status = qil_ctx_alloc(ctx, (size_t)(4 * cl * cr) * e, &T);
        if (status != QIL_OK) break;
        if (dtype == QIL_C64)
            QIL_TRY((qil_klaunch<fuse_delta_k<c64>>(ctx, dim3(nblk(4 * cl * cr)), dim3(256), 0, (const c64*)sig->site[(size_t)i], (int)cl, (int)cr, (c64*)T)));
        else
            QIL_TRY((qil_klaunch<fuse_delta_k<double>>(ctx, dim3(nblk(4 * cl * cr)), dim3(256), 0, (const double*)sig->site[(size_t)i], (int)cl, (int)cr, (double*)T)));
        int64_t r = 0;
        void *U = nullptr, *SV = nullptr;
        status = svd_trunc_dev(ctx, dtype, 2 * cl, 2 * cr, T, 2 * cl, P.cutoff, true, P.maxdim, 1, 2, &r, &U, &SV,
                               nullptr);
        qil_ctx_free(ctx, T);
        if (status != QIL_OK) break;
        qil_chain_adopt(zt, 2 * i, U);        // core_main [b_{i-1}, s_main, c]
        qil_chain_adopt(zt, 2 * i + 1, SV);   // core_copy [c, s_copy, b_i]
        zt->dims[(size_t)(2 * i)] = cl;
        zt->dims[(size_t)(2 * i + 1)] = r;
        zt->dims[(size_t)(2 * i + 2)] = cr;
    }
    qil_mps_destroy(sig);
    if (status != QIL_OK) {
        qil_mps_destroy(zt);
        return status;
    }
    *out = zt;
    return QIL_OK;
}

// nb signals of one length encoded concurrently (the serial loop over signal kinds of
// scripts/benchmark/zt_full_runtime.jl:151-221): outs[j] = signal_mps / signal_ztmps(xs[j]; ...), each encoder on one slot
// of the context's streams.  All or nothing on failure.
static int signal_batch(qil_context* ctx, const void* const* xs, int64_t nb, int64_t len, int dtype, int method, double cutoff,
                        int64_t maxdim, int64_t k, int64_t p, int q, uint64_t seed, int64_t mindim, int paired,
                        qil_mps** outs) {
    QIL_REQUIRE(ctx && nb >= 0 && ((xs && outs) || nb == 0), QIL_EINVAL_ARG, "signal batch: null argument");
    for (int64_t j = 0; j < nb; ++j) {
        outs[j] = nullptr;
        QIL_REQUIRE(xs[j], QIL_EINVAL_ARG, "signal batch: signal %lld is null", (long long)j);
    }
    const int st = qil_run_batch_on(ctx, nb, nullptr, [&](int64_t j, qil_context* work) {
        return paired ? qil_signal_ztmps(work, xs[j], len, dtype, method, cutoff, maxdim, k, p, q, seed, mindim, &outs[j])
                      : qil_signal_mps(work, xs[j], len, dtype, method, cutoff, maxdim, k, p, q, seed, mindim, &outs[j]);
    });
    if (st != QIL_OK)
        for (int64_t j = 0; j < nb; ++j) {
            if (outs[j]) qil_mps_destroy(outs[j]);
            outs[j] = nullptr;
        }
    return st;
}

extern "C" int qil_signal_mps_batch(qil_context* ctx, const void* const* xs, int64_t nb, int64_t len, int dtype, int method,
                                    double cutoff, int64_t maxdim, int64_t k, int64_t p, int q, uint64_t seed, int64_t mindim,
                                    qil_mps** outs) {
    return signal_batch(ctx, xs, nb, len, dtype, method, cutoff, maxdim, k, p, q, seed, mindim, 0, outs);
}

extern "C" int qil_signal_ztmps_batch(qil_context* ctx, const void* const* xs, int64_t nb, int64_t len, int dtype, int method,
                                      double cutoff, int64_t maxdim, int64_t k, int64_t p, int q, uint64_t seed,
                                      int64_t mindim, qil_mps** outs) {
    return signal_batch(ctx, xs, nb, len, dtype, method, cutoff, maxdim, k, p, q, seed, mindim, 1, outs);
}
