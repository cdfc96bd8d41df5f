// libqilcpu.so -- CPU (C++ / OpenMP) implementation of the apply path behind the SAME C ABI as libqilhip.so
// (include/qilaplace_hip.h), for the box-side CPU baseline of bench.py (SURVEY.md 8d "CPU baseline beside it" (1)).
//
// TEST / BASELINE INFRASTRUCTURE (lives under oracle/): never linked into or loaded by the product path (qilaplace.jl_amd loads
// libqilhip.so only; tests/test_cabi_symbols.py checks that).  It implements the subset of the header the baseline
// needs -- containers on host memory, apply, coefficient -- with the reference's semantics:
//
//   apply(W, psi)            src/linalg/apply.jl:75-122   B[(a,alpha), s, (b,beta)] = sum_s' W[a,s',s,b] A[alpha,s',beta]
//   coefficient(psi, bits)   src/mps.jl:669-678
//
// The reference runs three passes per site (contraction :101, left combiner :114, right combiner :118); here the fused
// tensor is written once, straight into the fused layout, with the (s, column) loop spread over the OpenMP team and
// the contiguous alpha run vectorised -- what a competent CPU port of the same contraction does.  Threads: OpenMP,
// qilcpu_set_threads(n) (0 = all cores).
#include <omp.h>

#include <complex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "qilaplace_hip.h"

namespace {
thread_local char g_err[512] = "";
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
size_t esize(int dtype) { return dtype == QIL_C64 ? 16 : 8; }
typedef std::complex<double> c64;
}  // namespace

struct qil_context {
    int threads = 0;
};
struct qil_chain_cpu {
    qil_context* ctx = nullptr;
    int dtype = QIL_F64, paired = 0, phys = 2;
    std::vector<int64_t> dims, site_ids;
    std::vector<void*> site;
    double amplitude = 1.0;
    int64_t n() const { return (int64_t)site.size(); }
    size_t site_bytes(int64_t i) const { return (size_t)(dims[(size_t)i] * phys * dims[(size_t)i + 1]) * esize(dtype); }
    ~qil_chain_cpu() {
        for (void* p : site) free(p);
    }
};
struct qil_mps : qil_chain_cpu {};
struct qil_mpo : qil_chain_cpu {};

namespace {
template <class H>
int chain_new(qil_context* ctx, int64_t n, int dtype, int paired, int phys, const int64_t* bonds, const int64_t* ids,
              const void* const* ptrs, H** out) {
    if (!ctx || !out) return fail(QIL_EINVAL_ARG, "null argument");
    if (n < 1) return fail(QIL_EINVAL_LENGTH, "a tensor chain needs at least one site (got %lld)", (long long)n);
    if (paired && n % 2) return fail(QIL_EINVAL_LENGTH, "paired chains need an even number of tensors (got %lld)", (long long)n);
    H* h = new H();
    h->ctx = ctx;
    h->dtype = dtype;
    h->paired = paired;
    h->phys = phys;
    h->dims.assign((size_t)n + 1, 1);
    for (int64_t i = 0; i + 1 < n; ++i) h->dims[(size_t)i + 1] = bonds[i];
    h->site_ids.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) h->site_ids[(size_t)i] = ids ? ids[i] : i + 1;
    h->site.assign((size_t)n, nullptr);
    for (int64_t i = 0; i < n; ++i) {
        const size_t bytes = h->site_bytes(i);
        void* p = nullptr;
        if (posix_memalign(&p, 64, bytes ? bytes : 64) != 0) {
            delete h;
            return fail(QIL_ENOMEM, "out of host memory (%zu bytes)", bytes);
        }
        h->site[(size_t)i] = p;
        if (ptrs) memcpy(p, ptrs[i], bytes);
    }
    *out = h;
    return QIL_OK;
}

// one site of apply into `out` (R x 2 x C, R = Dl cl, C = Dr cr), fused layout row = alpha + cl a, col = beta + cr b
template <class TW, class TA, class TO>
void apply_site(const TW* __restrict__ W, const TA* __restrict__ A, TO* __restrict__ out, int64_t Dl, int64_t Dr,
                int64_t cl, int64_t cr, int threads) {
    const int64_t R = Dl * cl, C = Dr * cr;
#pragma omp parallel for collapse(2) schedule(static) num_threads(threads)
    for (int64_t col = 0; col < C; ++col)
        for (int s = 0; s < 2; ++s) {
            const int64_t beta = col % cr, b = col / cr;
            const TA* a0 = A + cl * (0 + 2 * beta);
            const TA* a1 = A + cl * (1 + 2 * beta);
            TO* o = out + R * (s + 2 * col);
            for (int64_t a = 0; a < Dl; ++a) {
                const TO w0 = W[a + Dl * (0 + 2 * (s + 2 * b))], w1 = W[a + Dl * (1 + 2 * (s + 2 * b))];
                TO* oa = o + cl * a;
#pragma omp simd
                for (int64_t al = 0; al < cl; ++al) oa[al] = w0 * TO(a0[al]) + w1 * TO(a1[al]);
            }
        }
}

int nthreads(const qil_context* ctx) { return ctx->threads > 0 ? ctx->threads : omp_get_max_threads(); }

void apply_site_any(const qil_mpo* W, const qil_mps* psi, int64_t i, void* out) {
    const int64_t Dl = W->dims[(size_t)i], Dr = W->dims[(size_t)i + 1], cl = psi->dims[(size_t)i], cr = psi->dims[(size_t)i + 1];
    const int th = nthreads(psi->ctx);
    const bool wc = W->dtype == QIL_C64, ac = psi->dtype == QIL_C64;
    const void *w = W->site[(size_t)i], *a = psi->site[(size_t)i];
    if (wc && ac) apply_site((const c64*)w, (const c64*)a, (c64*)out, Dl, Dr, cl, cr, th);
    else if (wc) apply_site((const c64*)w, (const double*)a, (c64*)out, Dl, Dr, cl, cr, th);
    else if (ac) apply_site((const double*)w, (const c64*)a, (c64*)out, Dl, Dr, cl, cr, th);
    else apply_site((const double*)w, (const double*)a, (double*)out, Dl, Dr, cl, cr, th);
}
}  // namespace

extern "C" {
const char* qil_last_error(void) { return g_err; }
const char* qil_version(void) { return "qilcpu 0.1.0 (C++/OpenMP baseline)"; }
int qil_context_create(int, void*, qil_context** out) {
    if (!out) return fail(QIL_EINVAL_ARG, "null argument");
    *out = new qil_context();
    return QIL_OK;
}
int qil_context_destroy(qil_context* ctx) {
    delete ctx;
    return QIL_OK;
}
int qil_context_synchronize(qil_context*) { return QIL_OK; }
// CPU-only extension: OpenMP team size of every call on this context (0 = all cores); returns the size in effect
int qilcpu_set_threads(qil_context* ctx, int n) {
    ctx->threads = n;
    return nthreads(ctx);
}

int qil_mps_create(qil_context* ctx, int64_t n, int dtype, int paired, const int64_t* bond_dims, const int64_t* site_ids,
                   const void* const* site_ptrs, double amplitude, qil_mps** out) {
    int st = chain_new<qil_mps>(ctx, n, dtype, paired, 2, bond_dims, site_ids, site_ptrs, out);
    if (st == QIL_OK) (*out)->amplitude = amplitude;
    return st;
}
int qil_mps_alloc(qil_context* ctx, int64_t n, int dtype, int paired, const int64_t* bond_dims, const int64_t* site_ids,
                  double amplitude, qil_mps** out) {
    int st = chain_new<qil_mps>(ctx, n, dtype, paired, 2, bond_dims, site_ids, nullptr, out);
    if (st == QIL_OK) (*out)->amplitude = amplitude;
    return st;
}
int qil_mps_destroy(qil_mps* p) {
    delete p;
    return QIL_OK;
}
int qil_mps_nsites(const qil_mps* p, int64_t* n) {
    *n = p->n();
    return QIL_OK;
}
int qil_mps_bond_dims(const qil_mps* p, int64_t* b) {
    for (int64_t i = 0; i + 1 < p->n(); ++i) b[i] = p->dims[(size_t)i + 1];
    return QIL_OK;
}
int qil_mps_site_nbytes(const qil_mps* p, int64_t i, int64_t* nb) {
    *nb = (int64_t)p->site_bytes(i);
    return QIL_OK;
}
int qil_mps_download_site(const qil_mps* p, int64_t i, void* dst) {
    memcpy(dst, p->site[(size_t)i], p->site_bytes(i));
    return QIL_OK;
}
int qil_mps_site_device_ptr(const qil_mps* p, int64_t i, void** ptr) {       // host pointer here
    *ptr = p->site[(size_t)i];
    return QIL_OK;
}
int qil_mpo_create(qil_context* ctx, int64_t n, int dtype, int paired, const int64_t* bond_dims, const int64_t* site_ids,
                   const void* const* site_ptrs, qil_mpo** out) {
    return chain_new<qil_mpo>(ctx, n, dtype, paired, 4, bond_dims, site_ids, site_ptrs, out);
}
int qil_mpo_alloc(qil_context* ctx, int64_t n, int dtype, int paired, const int64_t* bond_dims, const int64_t* site_ids,
                  qil_mpo** out) {
    return chain_new<qil_mpo>(ctx, n, dtype, paired, 4, bond_dims, site_ids, nullptr, out);
}
int qil_mpo_destroy(qil_mpo* p) {
    delete p;
    return QIL_OK;
}
int qil_mpo_site_device_ptr(const qil_mpo* p, int64_t i, void** ptr) {
    *ptr = p->site[(size_t)i];
    return QIL_OK;
}

static int check_pair(const qil_mpo* W, const qil_mps* psi) {
    if (!W || !psi) return fail(QIL_EINVAL_ARG, "apply: null handle");
    if (W->n() != psi->n())
        return fail(QIL_EINVAL_LENGTH, "apply: MPO and MPS must have the same number of sites. Found length(W)=%lld, length(psi)=%lld",
                    (long long)W->n(), (long long)psi->n());                                        // apply.jl:76-80
    if (W->site_ids != psi->site_ids) return fail(QIL_EINVAL_SITES, "apply: MPO and MPS must have the same site indices.");  // :81-85
    return QIL_OK;
}

int qil_apply(const qil_mpo* W, const qil_mps* psi, qil_mps** out) {
    int st = check_pair(W, psi);
    if (st != QIL_OK) return st;
    const int64_t n = psi->n();
    std::vector<int64_t> bonds((size_t)(n > 1 ? n - 1 : 1));
    for (int64_t i = 0; i + 1 < n; ++i) bonds[(size_t)i] = W->dims[(size_t)i + 1] * psi->dims[(size_t)i + 1];
    const int odt = (W->dtype == QIL_C64 || psi->dtype == QIL_C64) ? QIL_C64 : QIL_F64;
    qil_mps* res = nullptr;
    st = qil_mps_alloc(psi->ctx, n, odt, psi->paired, bonds.data(), psi->site_ids.data(), psi->amplitude, &res);
    if (st != QIL_OK) return st;
    for (int64_t i = 0; i < n; ++i) apply_site_any(W, psi, i, res->site[(size_t)i]);
    *out = res;
    return QIL_OK;
}

// CPU-only extension: first touch of a caller buffer by the OpenMP team that will write it (NUMA placement)
int qilcpu_first_touch(qil_context* ctx, void* buf, int64_t bytes) {
    char* p = static_cast<char*>(buf);
    const int th = nthreads(ctx);
    const int64_t page = 1 << 21;
#pragma omp parallel for schedule(static) num_threads(th)
    for (int64_t off = 0; off < bytes; off += page) memset(p + off, 0, (size_t)(bytes - off < page ? bytes - off : page));
    return QIL_OK;
}

// CPU-only extension for bounded timing: site i of apply(W, psi) into a caller buffer of
// esize(promote) * (Dl cl) * 2 * (Dr cr) bytes (the 80 GB result of the metric configuration need not exist at once)
int qilcpu_apply_site(const qil_mpo* W, const qil_mps* psi, int64_t i, void* out) {
    int st = check_pair(W, psi);
    if (st != QIL_OK) return st;
    if (i < 0 || i >= psi->n() || !out) return fail(QIL_EINVAL_ARG, "apply_site: bad site / buffer");
    apply_site_any(W, psi, i, out);
    return QIL_OK;
}

int qil_coefficient_batch(const qil_mps* psi, int64_t nb, const uint8_t* bits, double* out) {   // mps.jl:669-678
    if (!psi || (nb && (!bits || !out))) return fail(QIL_EINVAL_ARG, "coefficient: null argument");
    const int64_t n = psi->n();
    for (int64_t t = 0; t < nb * n; ++t)
        if (bits[t] > 1) return fail(QIL_EINVAL_CONFIG, "coefficient: bit value %d outside [0,1]", (int)bits[t]);
    const bool cx = psi->dtype == QIL_C64;
#pragma omp parallel for schedule(dynamic) num_threads(nthreads(psi->ctx))
    for (int64_t q = 0; q < nb; ++q) {
        std::vector<c64> v(1, c64(1.0, 0.0)), nv;
        for (int64_t i = 0; i < n; ++i) {
            const int64_t cl = psi->dims[(size_t)i], cr = psi->dims[(size_t)i + 1];
            const int bit = bits[q * n + i];
            nv.assign((size_t)cr, c64(0, 0));
            for (int64_t be = 0; be < cr; ++be) {
                c64 acc(0, 0);
                if (cx) {
                    const c64* a = (const c64*)psi->site[(size_t)i] + cl * (bit + 2 * be);
                    for (int64_t al = 0; al < cl; ++al) acc += v[(size_t)al] * a[al];
                } else {
                    const double* a = (const double*)psi->site[(size_t)i] + cl * (bit + 2 * be);
                    for (int64_t al = 0; al < cl; ++al) acc += v[(size_t)al] * a[al];
                }
                nv[(size_t)be] = acc;
            }
            v.swap(nv);
        }
        out[2 * q] = psi->amplitude * v[0].real();
        out[2 * q + 1] = psi->amplitude * v[0].imag();
    }
    return QIL_OK;
}
}  // extern "C"
