// Read-out kernels: coefficient (C1), lazy <bits|W psi>, mps_to_vector (C2), norm (K3).
//
//   coefficient(psi, cfg)   src/mps.jl:669-678   amplitude * prod_i A_i[:, cfg_i, :]
//   mps_to_vector           src/mps.jl:716-743
//   norm                    src/mps.jl:754-771
//
// The reference contracts the row vector with the WHOLE site tensor and projects afterwards
// (mps.jl:675); here the physical slice is selected first (half the bytes, same arithmetic).
// Every query is a left-to-right vector-matrix chain: HBM-read bound on the selected slices.
// One workgroup owns one query and walks all sites inside ONE launch; each output entry is a
// dot product along the contiguous alpha axis, reduced with wavefront shuffles.
#include "qil_internal.h"
#include "qil_launch.h"
#include <set>
#include <mutex>
#include <map>

namespace {

struct c64 {
    double re, im;
};

__device__ __forceinline__ double cmul_add(double acc, double a, double b) { return fma(a, b, acc); }
__device__ __forceinline__ c64 cmul_add(c64 acc, c64 a, c64 b) {
    acc.re = fma(a.re, b.re, acc.re);
    acc.re = fma(-a.im, b.im, acc.re);
    acc.im = fma(a.re, b.im, acc.im);
    acc.im = fma(a.im, b.re, acc.im);
    return acc;
}
__device__ __forceinline__ c64 cmul_add(c64 acc, c64 a, double b) {
    acc.re = fma(a.re, b, acc.re);
    acc.im = fma(a.im, b, acc.im);
    return acc;
}
__device__ __forceinline__ c64 cmul_add(c64 acc, double a, c64 b) {
    acc.re = fma(a, b.re, acc.re);
    acc.im = fma(a, b.im, acc.im);
    return acc;
}
__device__ __forceinline__ double shfl_xor_t(double v, int m) { return __shfl_xor(v, m, 64); }
__device__ __forceinline__ c64 shfl_xor_t(c64 v, int m) {
    return c64{__shfl_xor(v.re, m, 64), __shfl_xor(v.im, m, 64)};
}
__device__ __forceinline__ double add_t(double a, double b) { return a + b; }
__device__ __forceinline__ c64 add_t(c64 a, c64 b) { return c64{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ double one_t(double) { return 1.0; }
__device__ __forceinline__ c64 one_t(c64) { return c64{1.0, 0.0}; }
__device__ __forceinline__ c64 to_c64(double v) { return c64{v, 0.0}; }
__device__ __forceinline__ c64 to_c64(c64 v) { return v; }
template <class TD>
__device__ __forceinline__ TD cast_elem(double v);
template <>
__device__ __forceinline__ double cast_elem<double>(double v) { return v; }
template <>
__device__ __forceinline__ c64 cast_elem<c64>(double v) { return c64{v, 0.0}; }
template <class TD>
__device__ __forceinline__ TD cast_elem(c64 v) { return v; }
__device__ __forceinline__ double conj_t(double v) { return v; }
__device__ __forceinline__ c64 conj_t(c64 v) { return c64{v.re, -v.im}; }

struct ChainSite {
    const void* A;  // MPS site
    const void* W;  // MPO site (lazy path) or null
    int cl, cr, Dl, Dr;
};

constexpr int kThreads = 256;

// v_out[beta] = sum_alpha v_in[alpha] * A[alpha, bit, beta]; groups of G lanes share one beta.
template <class T>
__global__ __launch_bounds__(kThreads) void coefficient_chain(const ChainSite* __restrict__ sites, int n,
                                                              const uint8_t* __restrict__ bits,
                                                              T* __restrict__ scratch, long long maxchi,
                                                              c64* __restrict__ out, double amplitude) {
    const long long q = blockIdx.x;
    T* v_in = scratch + (2 * q) * maxchi;
    T* v_out = v_in + maxchi;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int nwaves = kThreads / 64;
    if (threadIdx.x == 0) v_in[0] = one_t(T{});
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        const ChainSite S = sites[i];
        const int bit = bits[q * n + i];
        // bit == 2: the site's physical index is summed (marginal): v' = v (A[:,0,:] + A[:,1,:])
        const bool both = bit == 2;
        const T* __restrict__ M = static_cast<const T*>(S.A) + (long long)S.cl * (both ? 0 : bit);
        int G = 64;  // lanes per dot product: smallest power of two >= cl (cap 64)
        while (G > 1 && (G >> 1) >= S.cl) G >>= 1;
        const int per_wave = 64 / G;
        const int grp = lane / G, gl = lane - grp * G;
        for (int beta = wave * per_wave + grp; beta < S.cr; beta += nwaves * per_wave) {
            const T* col = M + 2LL * S.cl * beta;
            T acc{};
            if (both)
                for (int al = gl; al < S.cl; al += G) acc = cmul_add(acc, v_in[al], add_t(col[al], col[al + S.cl]));
            else
                for (int al = gl; al < S.cl; al += G) acc = cmul_add(acc, v_in[al], col[al]);
            for (int m = G >> 1; m >= 1; m >>= 1) acc = add_t(acc, shfl_xor_t(acc, m));
            if (gl == 0) v_out[beta] = acc;
        }
        // the block's own global writes become visible to its other waves here
        __threadfence_block();
        __syncthreads();
        T* t = v_in;
        v_in = v_out;
        v_out = t;
    }
    if (threadIdx.x == 0) {
        c64 r = to_c64(v_in[0]);
        out[q] = c64{r.re * amplitude, r.im * amplitude};
    }
}

// Lazy path: carry M[a, alpha] (Dl x cl) per query.
//   M'[b, beta] = sum_{s'} sum_{a, alpha} W[a, s', bit, b] M[a, alpha] A[alpha, s', beta]
// two stages per site, both through global scratch owned by the workgroup:
//   X[s'][b, alpha] = sum_a W[a, s', bit, b] M[a, alpha]
//   M'[b, beta]     = sum_{s'} sum_alpha X[s'][b, alpha] A[alpha, s', beta]
template <class TW, class TA>
__global__ __launch_bounds__(kThreads) void lazy_coefficient_chain(const ChainSite* __restrict__ sites, int n,
                                                                   const uint8_t* __restrict__ bits,
                                                                   c64* __restrict__ scratch, long long msz,
                                                                   c64* __restrict__ out, double amplitude) {
    const long long q = blockIdx.x;
    c64* M = scratch + (4 * q) * msz;   // [alpha + cl * a]  (alpha fastest)
    c64* Mn = M + msz;
    c64* X = M + 2 * msz;               // 2 * msz: X[s'][alpha + cl * b]
    if (threadIdx.x == 0) M[0] = c64{1.0, 0.0};
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        const ChainSite S = sites[i];
        const int bit = bits[q * n + i];
        const TW* __restrict__ W = static_cast<const TW*>(S.W);
        const TA* __restrict__ A = static_cast<const TA*>(S.A);
        // stage 1: X[sp][alpha + cl*b] = sum_a W[a + Dl*(sp + 2*(bit + 2*b))] * M[alpha + cl*a]
        const long long n1 = 2LL * S.cl * S.Dr;
        for (long long t = threadIdx.x; t < n1; t += kThreads) {
            const int alpha = (int)(t % S.cl);
            long long u = t / S.cl;
            const int b = (int)(u % S.Dr);
            const int sp = (int)(u / S.Dr);
            const TW* w = W + (long long)S.Dl * (sp + 2 * (bit + 2LL * b));
            c64 acc{};
            for (int a = 0; a < S.Dl; ++a) acc = cmul_add(acc, M[alpha + (long long)S.cl * a], w[a]);
            X[sp * msz + alpha + (long long)S.cl * b] = acc;
        }
        __threadfence_block();
        __syncthreads();
        // stage 2: Mn[beta + cr*b] = sum_sp sum_alpha X[sp][alpha + cl*b] * A[alpha + cl*(sp + 2*beta)]
        const long long n2 = (long long)S.cr * S.Dr;
        for (long long t = threadIdx.x; t < n2; t += kThreads) {
            const int beta = (int)(t % S.cr);
            const int b = (int)(t / S.cr);
            c64 acc{};
            for (int sp = 0; sp < 2; ++sp) {
                const c64* x = X + sp * msz + (long long)S.cl * b;
                const TA* acol = A + (long long)S.cl * (sp + 2LL * beta);
                for (int alpha = 0; alpha < S.cl; ++alpha) acc = cmul_add(acc, x[alpha], acol[alpha]);
            }
            Mn[beta + (long long)S.cr * b] = acc;
        }
        __threadfence_block();
        __syncthreads();
        c64* t = M;
        M = Mn;
        Mn = t;
    }
    if (threadIdx.x == 0) out[q] = c64{M[0].re * amplitude, M[0].im * amplitude};
}

// One step of the dense contraction: T_k[beta + cr*idx'] = sum_alpha T_{k-1}[alpha + cl*idx] * A[alpha, s, beta]
//   reverse = 0: idx' = 2*idx + s (site 1 = MSB);  reverse = 1: idx' = idx + s * 2^(k-1) (site 1 = LSB)
template <class T>
__global__ void select_slice(const T* __restrict__ Tm, long long nb, int cr, const uint8_t* __restrict__ bits,
                             int n, int site, T* __restrict__ Vn) {
    const long long total = nb * cr;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const long long q = t % nb;
        const long long beta = t / nb;
        const int bit = bits[q * n + site];
        Vn[t] = bit == 2 ? add_t(Tm[q + nb * (2 * beta)], Tm[q + nb * (1 + 2 * beta)])
                         : Tm[q + nb * (bit + 2 * beta)];
    }
}

// The same three steps in the table form of qil_launch.h: in a lock-step batch (the per-operator read-outs of a damping sweep)
// the `select_slice` of up to 16 chains is ONE launch instead of one per chain and site (r04: 3 072 launches of 3.5 us per sweep
// of 64 operators, each of which also drained its chain's ring to get onto the stream).
template <class T>
__device__ __forceinline__ void select_slice_body(const uint3 blockIdx, const uint3 gridDim, const T* __restrict__ Tm, long long nb, int cr,
                                                  const uint8_t* __restrict__ bits, int n, int site, T* __restrict__ Vn) {
    const long long total = nb * cr;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const long long q = t % nb;
        const long long beta = t / nb;
        const int bit = bits[q * n + site];
        Vn[t] = bit == 2 ? add_t(Tm[q + nb * (2 * beta)], Tm[q + nb * (1 + 2 * beta)]) : Tm[q + nb * (bit + 2 * beta)];
    }
}
template <class T>
struct select_slice_k {
    static constexpr int NT = 256, MINW = 1;
    template <class... QA>
    static __device__ __forceinline__ void run(const uint3 b, const uint3 g, QA... a) {
        select_slice_body<T>(b, g, a...);
    }
};
// Bit-sorted read-out: rows of Tm are in site i's order (queries with bit 0 first), the next site wants its own order:
// Vn[r + nb * beta] = Tm[map[r] + nb * beta]
template <class T>
__device__ __forceinline__ void gather_rows_body(const uint3 blockIdx, const uint3 gridDim, const T* __restrict__ Tm, long long nb, int cr,
                                                 const int* __restrict__ map, T* __restrict__ Vn) {
    const long long total = nb * cr;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const long long r = t % nb, beta = t / nb;
        Vn[t] = Tm[map[r] + nb * beta];
    }
}
template <class T>
struct gather_rows_k {
    static constexpr int NT = 256, MINW = 1;
    template <class... QA>
    static __device__ __forceinline__ void run(const uint3 b, const uint3 g, QA... a) {
        gather_rows_body<T>(b, g, a...);
    }
};
template <class T>
__device__ __forceinline__ void fill_ones_body(const uint3 blockIdx, const uint3 gridDim, T* __restrict__ v, long long n) {
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) v[t] = one_t(T{});
}
template <class T>
struct fill_ones_k {
    static constexpr int NT = 256, MINW = 1;
    template <class... QA>
    static __device__ __forceinline__ void run(const uint3 b, const uint3 g, QA... a) {
        fill_ones_body<T>(b, g, a...);
    }
};
template <class T>
__device__ __forceinline__ void finish_coeff_body(const uint3 blockIdx, const uint3 gridDim, const T* __restrict__ v, long long nb,
                                                  double amplitude, c64* __restrict__ out) {
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < nb; t += (long long)gridDim.x * blockDim.x) {
        const c64 r = to_c64(v[t]);
        out[t] = c64{r.re * amplitude, r.im * amplitude};
    }
}
template <class T>
struct finish_coeff_k {
    static constexpr int NT = 256, MINW = 1;
    template <class... QA>
    static __device__ __forceinline__ void run(const uint3 b, const uint3 g, QA... a) {
        finish_coeff_body<T>(b, g, a...);
    }
};

template <class T>
__global__ void fill_ones(T* __restrict__ v, long long n) {
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n;
         t += (long long)gridDim.x * blockDim.x)
        v[t] = one_t(T{});
}

template <class T>
__global__ void finish_coeff(const T* __restrict__ v, long long nb, double amplitude, c64* __restrict__ out) {
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < nb;
         t += (long long)gridDim.x * blockDim.x) {
        const c64 r = to_c64(v[t]);
        out[t] = c64{r.re * amplitude, r.im * amplitude};
    }
}


// ---- dense read-out of a sub-lattice of configurations (grid scans) --------------------------------------
template <class T>
__global__ void slice_sum(const T* __restrict__ A, int cl, int cr, T* __restrict__ out) {
    // out[alpha, beta] = A[alpha, 0, beta] + A[alpha, 1, beta]
    const long long total = (long long)cl * cr;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int al = (int)(t % cl);
        const long long be = t / cl;
        out[t] = add_t(A[al + (long long)cl * (2 * be)], A[al + (long long)cl * (2 * be + 1)]);
    }
}
template <class T>
__global__ void bit_reverse_scale(const T* __restrict__ in, T* __restrict__ out, int nbits, double scale, int rev) {
    const long long N = 1LL << nbits;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < N; t += (long long)gridDim.x * blockDim.x) {
        long long d = t;
        if (rev) d = nbits == 0 ? 0 : (long long)(__brevll((unsigned long long)t) >> (64 - nbits));
        T v = in[t];
        reinterpret_cast<double*>(&v)[0] *= scale;
        if (sizeof(T) == 16) reinterpret_cast<double*>(&v)[1] *= scale;
        out[d] = v;
    }
}

// ---- lazy read-out, GEMM form (many queries, large chi * D) ----------------------------------------------
template <class TS>
__global__ void widen_to_c64(const TS* __restrict__ src, c64* __restrict__ dst, long long n) {
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x)
        dst[t] = to_c64(src[t]);
}
// W[a, s', s, b] -> Wp[a, s', b, s] (output bit slowest), optionally widened: each output-bit slice becomes one
// contiguous (D_l x 2 D_r) operand
template <class TS, class TD>
__global__ void mpo_site_bit_major(const TS* __restrict__ W, TD* __restrict__ Wp, int Dl, int Dr) {
    const long long total = 4LL * Dl * Dr;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int a = (int)(t % Dl);
        long long u = t / Dl;
        const int sp = (int)(u & 1);
        u >>= 1;
        const int b = (int)(u % Dr);
        const int bit = (int)(u / Dr);
        Wp[t] = cast_elem<TD>(W[a + (long long)Dl * (sp + 2 * (bit + 2LL * b))]);
    }
}
template <class T>
__global__ void lazy_finish(const T* __restrict__ M, long long nb, double amplitude, c64* __restrict__ out) {
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < nb; q += (long long)gridDim.x * blockDim.x) {
        const c64 v = to_c64(M[q]);
        out[q] = c64{v.re * amplitude, v.im * amplitude};
    }
}

// All queries advance together; per site two strided-batch MFMA GEMMs (batch = query):
//   X_q[alpha, (s', b)] = M_q[alpha, a] W[a, (s', b) | s = bit_q]               (chi_l x D_l) (D_l x 2 D_r)
//   M'_q[beta, b]     = sum_{(alpha, s')} A[(alpha, s'), beta] X_q[(alpha, s'), b]  (chi_r x 2 chi_l) (2 chi_l x D_r)
// The MPO site is re-laid once per site with the output bit slowest (a few hundred KB), so each query's
// output-bit slice is one contiguous operand picked by the per-batch operand shift of the GEMM; A is used
// exactly as it lies in HBM, and X_q comes out of the first product in the layout the second one reads.
int lazy_gemm_path(qil_context* ctx, const qil_mpo* W, const qil_mps* psi, int64_t nb, const uint8_t* dbits,
                   c64* dout) {
    const int64_t n = psi->n();
    const bool wc = W->dtype == QIL_C64, ac = psi->dtype == QIL_C64;
    const int dt = (wc || ac) ? QIL_C64 : QIL_F64;
    const size_t e = qil_elem_size(dt);
    long long maxM = 1, maxX = 1, maxW = 1, maxA = 1;
    for (int64_t i = 0; i < n; ++i) {
        const long long cl = psi->dims[(size_t)i], cr = psi->dims[(size_t)i + 1];
        const long long Dl = W->dims[(size_t)i], Dr = W->dims[(size_t)i + 1];
        maxM = std::max(maxM, std::max(cl * Dl, cr * Dr));
        maxX = std::max(maxX, cl * 2 * Dr);
        maxW = std::max(maxW, Dl * 4 * Dr);
        maxA = std::max(maxA, cl * 2 * cr);
    }
    // queries per pass: bounded scratch (X is the big one) and the grid's batch limit
    const long long per_query = (2 * maxM + maxX) * (long long)e;
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(nb, 32768), (8LL << 30) / per_query));
    void *M0 = nullptr, *M1 = nullptr, *X = nullptr, *Wc = nullptr, *Ac = nullptr;
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(chunk * maxM) * e, &M0));
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(chunk * maxM) * e, &M1));
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(chunk * maxX) * e, &X));
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)maxW * e, &Wc));
    if (dt == QIL_C64 && !ac) QIL_TRY(qil_ctx_alloc(ctx, (size_t)maxA * 16, &Ac));
    int st = QIL_OK;
    for (int64_t q0 = 0; q0 < nb && st == QIL_OK; q0 += chunk) {
        const int64_t nq = std::min<int64_t>(chunk, nb - q0);
        const unsigned g1 = (unsigned)std::min<long long>((nq + 255) / 256, 4096);
        if (dt == QIL_C64) hipLaunchKernelGGL(fill_ones<c64>, dim3(g1), dim3(256), 0, qil_stream(ctx), (c64*)M0, (long long)nq);
        else hipLaunchKernelGGL(fill_ones<double>, dim3(g1), dim3(256), 0, qil_stream(ctx), (double*)M0, (long long)nq);
        void *Mc = M0, *Mn = M1;
        for (int64_t i = 0; i < n && st == QIL_OK; ++i) {
            const long long cl = psi->dims[(size_t)i], cr = psi->dims[(size_t)i + 1];
            const long long Dl = W->dims[(size_t)i], Dr = W->dims[(size_t)i + 1];
            const void* Ap = psi->site[(size_t)i];
            {
                const unsigned g = (unsigned)std::min<long long>((Dl * 4 * Dr + 255) / 256, 4096);
                const void* Ws = W->site[(size_t)i];
                if (dt == QIL_F64)
                    hipLaunchKernelGGL((mpo_site_bit_major<double, double>), dim3(g), dim3(256), 0, qil_stream(ctx),
                                       (const double*)Ws, (double*)Wc, (int)Dl, (int)Dr);
                else if (wc)
                    hipLaunchKernelGGL((mpo_site_bit_major<c64, c64>), dim3(g), dim3(256), 0, qil_stream(ctx), (const c64*)Ws,
                                       (c64*)Wc, (int)Dl, (int)Dr);
                else
                    hipLaunchKernelGGL((mpo_site_bit_major<double, c64>), dim3(g), dim3(256), 0, qil_stream(ctx),
                                       (const double*)Ws, (c64*)Wc, (int)Dl, (int)Dr);
            }
            if (Ac) {
                hipLaunchKernelGGL(widen_to_c64<double>, dim3((unsigned)std::min<long long>((cl * 2 * cr + 255) / 256, 4096)),
                                   dim3(256), 0, qil_stream(ctx), (const double*)Ap, (c64*)Ac, cl * 2 * cr);
                Ap = Ac;
            }
            qil_gemm_batch b1, b2;
            b1.count = nq;
            b1.a_bs = cl * Dl;
            b1.c_bs = cl * 2 * Dr;
            b1.b_sel = dbits + q0 * n + i;
            b1.b_sel_step = n;
            b1.b_sel_stride = Dl * 2 * Dr;
            st = qil_dev_gemm_batched(ctx, dt, 0, 0, cl, 2 * Dr, Dl, Mc, cl, Wc, Dl, X, cl, &b1);
            if (st != QIL_OK) break;
            b2.count = nq;
            b2.b_bs = cl * 2 * Dr;
            b2.c_bs = cr * Dr;
            st = qil_dev_gemm_batched(ctx, dt, 1, 0, cr, Dr, 2 * cl, Ap, 2 * cl, X, 2 * cl, Mn, cr, &b2);
            std::swap(Mc, Mn);
        }
        if (st != QIL_OK) break;
        if (dt == QIL_C64)
            hipLaunchKernelGGL(lazy_finish<c64>, dim3(g1), dim3(256), 0, qil_stream(ctx), (const c64*)Mc, (long long)nq,
                               psi->amplitude, dout + q0);
        else
            hipLaunchKernelGGL(lazy_finish<double>, dim3(g1), dim3(256), 0, qil_stream(ctx), (const double*)Mc, (long long)nq,
                               psi->amplitude, dout + q0);
        if (hipGetLastError() != hipSuccess) st = qil_fail(QIL_EHIP, "lazy coefficient (GEMM form): launch failed");
    }
    qil_ctx_free(ctx, M0);
    qil_ctx_free(ctx, M1);
    qil_ctx_free(ctx, X);
    qil_ctx_free(ctx, Wc);
    if (Ac) qil_ctx_free(ctx, Ac);
    return st;
}

int upload_bits(qil_context* ctx, int64_t nb, int64_t n, const uint8_t* bits, uint8_t** dbits, int max_bit = 1) {
    for (int64_t t = 0; t < nb * n; ++t)
        QIL_REQUIRE(bits[t] <= max_bit, QIL_EINVAL_CONFIG, "coefficient: bit value %d outside [0,%d]", (int)bits[t],
                    max_bit);
    void* p = nullptr;
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(nb * n), &p));
    hipError_t e = hipMemcpyAsync(p, bits, (size_t)(nb * n), hipMemcpyHostToDevice, qil_stream(ctx));
    if (e == hipSuccess) e = qil_stream_sync(ctx);  // `bits` is caller memory
    if (e != hipSuccess) {
        qil_ctx_free(ctx, p);
        return qil_fail(QIL_EHIP, "bit upload failed: %s", hipGetErrorString(e));
    }
    *dbits = static_cast<uint8_t*>(p);
    return QIL_OK;
}

}  // namespace

static int coefficient_impl(const qil_mps* psi, int64_t nb, const uint8_t* bits, double* out, int max_bit);

extern "C" int qil_coefficient_batch(const qil_mps* psi, int64_t nb, const uint8_t* bits, double* out) {
    return coefficient_impl(psi, nb, bits, out, 1);
}

// Marginals: bit value 2 sums the site's physical index (a partial trace with the all-ones vector), so
// e.g. a Laplace value L(s_k) = dt sqrt(N) sum_j <k, j | psi>  (docs/src/tutorials/dt.jl:187-197: N
// coefficient calls per value) is ONE chain with every copy-site bit set to 2.
extern "C" int qil_coefficient_marginal_batch(const qil_mps* psi, int64_t nb, const uint8_t* bits, double* out) {
    return coefficient_impl(psi, nb, bits, out, 2);
}

// Enqueue the read-out of nb configurations (device bits) of psi into dout (nb complex, device); no
// synchronisation, every temporary goes back to the pool in stream order.
// Plan of the bit-sorted GEMM read-out (built on the host from the caller's bits, shared by every chain read at the same
// configurations): before site i the query vectors are stored in order_i = queries with bit_i = 0 first (stable), so the site
// costs two products with ONE slice each -- n0[i] x chi_l by chi_l x chi_r and (nb - n0[i]) x chi_l by chi_l x chi_r -- instead
// of one nb x chi_l by chi_l x 2 chi_r product that computes both slices for every query and throws half away.  map[i][r] = row
// of site i's result that becomes row r of site i + 1's operand (the last map restores the caller's order).
struct SortPlan {
    int* dmap = nullptr;                 // [n][nb], device
    std::vector<int64_t> n0;             // per site: queries with bit 0
};
static int build_sort_plan(qil_context* ctx, int64_t nb, int64_t n, const uint8_t* bits, SortPlan* plan) {
    std::vector<int> map((size_t)(n * nb)), order((size_t)nb), next((size_t)nb), pos((size_t)nb);
    plan->n0.assign((size_t)n, 0);
    auto order_of = [&](int64_t site, std::vector<int>& o) -> int64_t {
        int64_t z = 0;
        if (site >= n) {
            for (int64_t q = 0; q < nb; ++q) o[(size_t)q] = (int)q;
            return nb;
        }
        for (int64_t q = 0; q < nb; ++q)
            if (bits[q * n + site] == 0) o[(size_t)z++] = (int)q;
        int64_t w = z;
        for (int64_t q = 0; q < nb; ++q)
            if (bits[q * n + site] != 0) o[(size_t)w++] = (int)q;
        return z;
    };
    plan->n0[0] = order_of(0, order);
    for (int64_t i = 0; i < n; ++i) {
        for (int64_t r = 0; r < nb; ++r) pos[(size_t)order[(size_t)r]] = (int)r;
        const int64_t z = order_of(i + 1, next);
        if (i + 1 < n) plan->n0[(size_t)i + 1] = z;
        for (int64_t r = 0; r < nb; ++r) map[(size_t)(i * nb + r)] = pos[(size_t)next[(size_t)r]];
        order.swap(next);
    }
    void* p = nullptr;
    QIL_TRY(qil_ctx_alloc(ctx, map.size() * sizeof(int), &p));
    hipError_t e = hipMemcpyAsync(p, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice, qil_stream(ctx));
    if (e == hipSuccess) e = qil_stream_sync(ctx);         // `map` is a local
    if (e != hipSuccess) {
        qil_ctx_free(ctx, p);
        return qil_fail(QIL_EHIP, "read-out plan upload failed: %s", hipGetErrorString(e));
    }
    plan->dmap = static_cast<int*>(p);
    return QIL_OK;
}
// queries x bonds from which the sorted GEMM form is taken (same crossover as the GEMM form itself)
static bool wants_sort_plan(const qil_mps* psi, int64_t nb, int max_bit) {
    if (max_bit > 1 || nb < 4) return false;                 // a marginal (bit 2) needs both slices of its site
    long long maxchi = 1;
    for (int64_t i = 0; i < psi->n(); ++i) maxchi = std::max<long long>(maxchi, psi->dims[(size_t)i + 1]);
    return maxchi >= 128 || (nb >= 1024 && maxchi >= 16);
}

static int coefficient_enqueue(const qil_mps* psi, int64_t nb, const uint8_t* dbits, void* dout, const SortPlan* plan = nullptr) {
    qil_context* ctx = psi->ctx;
    const int64_t n = psi->n();
    std::vector<ChainSite> tab((size_t)n);
    long long maxchi = 1;
    for (int64_t i = 0; i < n; ++i) {
        tab[(size_t)i] = ChainSite{psi->site[(size_t)i], nullptr, (int)psi->dims[(size_t)i],
                                   (int)psi->dims[(size_t)i + 1], 1, 1};
        maxchi = std::max<long long>(maxchi, psi->dims[(size_t)i + 1]);
    }
    const size_t esz = qil_elem_size(psi->dtype);
    // crossover between one-workgroup-per-query chains and the all-queries-together GEMM path (tuning aid:
    // QIL_COEFF_GEMM_MINCHI)
    // Measured on a 256 x 256 grid scan (65,536 queries): bond 504 chains 0.40 s vs GEMM 0.09 s; bond 23
    // chains 4.9 ms vs GEMM 3.2 ms -- many queries favour the GEMM path at any bond dimension.  Few queries, 40
    // complex sites: bond 64 chains 0.37 ms vs GEMM 0.55 ms, bond 128 1.07 vs 0.81 ms, bond 512 9.2 vs 2.1 ms.
    static const long long min_chi = 128;
    if (nb >= 4 && (maxchi >= min_chi || (nb >= 1024 && maxchi >= 16))) {
        // Large bonds: all queries advance together, one f64-MFMA GEMM per site.  The site tensor is
        // read ONCE for the whole batch: T (nb x 2 chi_r) = V (nb x chi_l) * A_i (chi_l x 2 chi_r),
        // then each query keeps the column block of its own bit.
        void *V = nullptr, *Vn = nullptr, *Tm = nullptr;
        QIL_TRY(qil_ctx_alloc(ctx, (size_t)(nb * maxchi) * esz, &V));
        QIL_TRY(qil_ctx_alloc(ctx, (size_t)(nb * maxchi) * esz, &Vn));
        QIL_TRY(qil_ctx_alloc(ctx, (size_t)(2 * nb * maxchi) * esz, &Tm));
        const bool cx = psi->dtype == QIL_C64;
        const unsigned g1 = (unsigned)std::min<long long>((nb + 255) / 256, 4096);
        if (cx) QIL_TRY((qil_klaunch<fill_ones_k<c64>>(ctx, dim3(g1), dim3(256), 0, (c64*)V, (long long)nb)));
        else QIL_TRY((qil_klaunch<fill_ones_k<double>>(ctx, dim3(g1), dim3(256), 0, (double*)V, (long long)nb)));
        for (int64_t i = 0; i < n; ++i) {
            const int64_t cl = psi->dims[(size_t)i], cr = psi->dims[(size_t)i + 1];
            const unsigned g = (unsigned)std::min<long long>((nb * cr + 255) / 256, 65536);
            if (plan && plan->dmap) {
                // bit-sorted: rows [0, n0) take slice 0, rows [n0, nb) slice 1 (slice s of A[alpha, s, beta]: offset s chi_l, ld 2 chi_l)
                const int64_t z = plan->n0[(size_t)i];
                const char* site = static_cast<const char*>(psi->site[(size_t)i]);
                for (int sl = 0; sl < 2; ++sl) {
                    const int64_t r0 = sl ? z : 0, rows = sl ? nb - z : z;
                    if (rows == 0) continue;
                    const void* Vs = static_cast<const char*>(V) + (size_t)r0 * esz;
                    void* Ts = static_cast<char*>(Tm) + (size_t)r0 * esz;
                    const void* As = site + (size_t)(sl * cl) * esz;
                    if (rows <= 48) QIL_TRY(qil_dev_gemm_skinny(ctx, psi->dtype, 0, 0, rows, cr, cl, Vs, nb, As, 2 * cl, Ts, nb));
                    else QIL_TRY(qil_dev_gemm(ctx, psi->dtype, 0, 0, rows, cr, cl, Vs, nb, As, 2 * cl, Ts, nb));
                }
                const int* map = plan->dmap + (size_t)(i * nb);
                if (cx) QIL_TRY((qil_klaunch<gather_rows_k<c64>>(ctx, dim3(g), dim3(256), 0, (const c64*)Tm, (long long)nb, (int)cr, map, (c64*)Vn)));
                else QIL_TRY((qil_klaunch<gather_rows_k<double>>(ctx, dim3(g), dim3(256), 0, (const double*)Tm, (long long)nb, (int)cr, map, (double*)Vn)));
                std::swap(V, Vn);
                continue;
            }
            QIL_TRY(qil_dev_gemm(ctx, psi->dtype, 0, 0, nb, 2 * cr, cl, V, nb, psi->site[(size_t)i], cl, Tm, nb));
            if (cx)
                QIL_TRY((qil_klaunch<select_slice_k<c64>>(ctx, dim3(g), dim3(256), 0, (const c64*)Tm, (long long)nb, (int)cr,
                                                          (const uint8_t*)dbits, (int)n, (int)i, (c64*)Vn)));
            else
                QIL_TRY((qil_klaunch<select_slice_k<double>>(ctx, dim3(g), dim3(256), 0, (const double*)Tm, (long long)nb, (int)cr,
                                                             (const uint8_t*)dbits, (int)n, (int)i, (double*)Vn)));
            std::swap(V, Vn);
        }
        if (cx) QIL_TRY((qil_klaunch<finish_coeff_k<c64>>(ctx, dim3(g1), dim3(256), 0, (const c64*)V, (long long)nb, psi->amplitude, (c64*)dout)));
        else QIL_TRY((qil_klaunch<finish_coeff_k<double>>(ctx, dim3(g1), dim3(256), 0, (const double*)V, (long long)nb, psi->amplitude, (c64*)dout)));
        qil_ctx_free(ctx, V);
        qil_ctx_free(ctx, Vn);
        qil_ctx_free(ctx, Tm);
        return QIL_OK;
    }
    void *scratch = nullptr, *pin = nullptr, *dtab = nullptr;
    int slot = 0;
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(2 * nb * maxchi) * esz, &scratch));
    QIL_TRY(qil_ctx_desc_acquire(ctx, tab.size() * sizeof(ChainSite), &pin, &dtab, &slot));
    memcpy(pin, tab.data(), tab.size() * sizeof(ChainSite));
    QIL_HIP(hipMemcpyAsync(dtab, pin, tab.size() * sizeof(ChainSite), hipMemcpyHostToDevice, qil_stream(ctx)));
    if (psi->dtype == QIL_C64)
        hipLaunchKernelGGL(coefficient_chain<c64>, dim3((unsigned)nb), dim3(kThreads), 0, qil_stream(ctx),
                           (const ChainSite*)dtab, (int)n, dbits, (c64*)scratch, maxchi, (c64*)dout,
                           psi->amplitude);
    else
        hipLaunchKernelGGL(coefficient_chain<double>, dim3((unsigned)nb), dim3(kThreads), 0, qil_stream(ctx),
                           (const ChainSite*)dtab, (int)n, dbits, (double*)scratch, maxchi, (c64*)dout,
                           psi->amplitude);
    QIL_HIP(hipGetLastError());
    QIL_TRY(qil_ctx_desc_commit(ctx, slot));
    qil_ctx_free(ctx, scratch);
    return QIL_OK;
}

static int coefficient_impl(const qil_mps* psi, int64_t nb, const uint8_t* bits, double* out, int max_bit) {
    QIL_REQUIRE(psi && (nb == 0 || (bits && out)), QIL_EINVAL_ARG, "coefficient: null argument");
    if (nb == 0) return QIL_OK;
    qil_context* ctx = psi->ctx;
    QIL_TRY(qil_ctx_activate(ctx));
    qil_call_scope call_scope(ctx);
    uint8_t* dbits = nullptr;
    QIL_TRY(upload_bits(ctx, nb, psi->n(), bits, &dbits, max_bit));
    SortPlan plan;
    if (wants_sort_plan(psi, nb, max_bit)) QIL_TRY(build_sort_plan(ctx, nb, psi->n(), bits, &plan));
    void* dout = nullptr;
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)nb * 16, &dout));
    QIL_TRY(coefficient_enqueue(psi, nb, dbits, dout, &plan));
    if (plan.dmap) qil_ctx_free(ctx, plan.dmap);
    QIL_HIP(hipMemcpyAsync(out, dout, (size_t)nb * 16, hipMemcpyDeviceToHost, qil_stream(ctx)));
    QIL_HIP(qil_stream_sync(ctx));
    qil_ctx_free(ctx, dout);
    qil_ctx_free(ctx, dbits);
    return QIL_OK;
}

// The body of a damping sweep (BASELINE.json configs[3]; the reference loops `W = build_dt_mpo(psi, wr); out = W * psi;
// coefficient(out, ...)` per damping value, docs/src/tutorials/dt.jl:150-197, zt.jl:300-348): for every operator of the
// batch the product W_j psi is materialised by the apply kernel and read out at the same nb configurations.  One bit
// upload, one download and ONE host synchronisation for the whole batch; each product's blocks return to the pool in
// stream order and serve the next one.
// The sweep body with its results LEFT IN HBM: dout = nw x nb complex values (operator-major) in psi's context, complete when
// the call returns (every slot's stream has been synchronised by the batch runner / the home stream holds the rest in order).
// qil_apply_coefficient_sweep copies them to the host; qil_apply_coefficient_sweep_gather (qil_comm.hip) all-gathers them first.
int qil_apply_coefficient_sweep_dev(const qil_mpo* const* Ws, int64_t nw, const qil_mps* psi, int64_t nb, const uint8_t* bits,
                                    void* dout) {
    qil_context* ctx = psi->ctx;
    uint8_t* dbits = nullptr;
    QIL_TRY(upload_bits(ctx, nb, psi->n(), bits, &dbits, 1));
    SortPlan plan;                                        // one plan for every operator's read-out (same configurations)
    if (nb >= 4) QIL_TRY(build_sort_plan(ctx, nb, psi->n(), bits, &plan));
    bool distinct = true;                                 // operators change context for the batch: each must be its own handle
    {
        std::set<const qil_mpo*> seen;
        for (int64_t j = 0; j < nw; ++j) {
            QIL_REQUIRE(Ws[j], QIL_EINVAL_ARG, "apply_coefficient_sweep: null operator %lld", (long long)j);
            QIL_REQUIRE(Ws[j]->ctx == ctx, QIL_EINVAL_ARG, "apply: MPO and MPS belong to different contexts");
            distinct = distinct && seen.insert(Ws[j]).second;
        }
    }
    auto one = [&](const qil_mpo* W, const qil_mps* state, int64_t j) {
        qil_mps* prod = nullptr;
        QIL_TRY(qil_apply_shared_state(W, state, &prod));
        const int st = coefficient_enqueue(prod, nb, dbits, static_cast<char*>(dout) + (size_t)(j * nb) * 16, &plan);
        qil_mps_destroy(prod);
        return st;
    };
    static const bool concurrent = true;   // tuning aid
    if (concurrent && distinct && nw >= 4) {
        // every value's product + read-out is a chain of ~100 small launches: the values run concurrently on the context's
        // streams.  The operators move to their slot for the duration of the call (bookkeeping only); the state, the bits and the
        // results are shared device buffers (the state is read-only here and every slot's stream waits for the home stream first).
        const int st = qil_run_batch_on(
            ctx, nw, [&](int64_t j, qil_context* slot) { qil_chain_rebind(const_cast<qil_mpo*>(Ws[j]), slot); },
            [&](int64_t j, qil_context*) { return one(Ws[j], psi, j); });
        QIL_TRY(st);
    } else {
        for (int64_t j = 0; j < nw; ++j) QIL_TRY(one(Ws[j], psi, j));
    }
    qil_ctx_free(ctx, dbits);
    if (plan.dmap) qil_ctx_free(ctx, plan.dmap);
    return QIL_OK;
}

extern "C" int qil_apply_coefficient_sweep(const qil_mpo* const* Ws, int64_t nw, const qil_mps* psi, int64_t nb,
                                           const uint8_t* bits, double* out) {
    QIL_REQUIRE(Ws && psi && (nb == 0 || (bits && out)), QIL_EINVAL_ARG, "apply_coefficient_sweep: null argument");
    if (nw == 0 || nb == 0) return QIL_OK;
    qil_context* ctx = psi->ctx;
    QIL_TRY(qil_ctx_activate(ctx));
    qil_call_scope call_scope(ctx);
    void* dout = nullptr;
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(nw * nb) * 16, &dout));
    QIL_TRY(qil_apply_coefficient_sweep_dev(Ws, nw, psi, nb, bits, dout));
    QIL_HIP(hipMemcpyAsync(out, dout, (size_t)(nw * nb) * 16, hipMemcpyDeviceToHost, qil_stream(ctx)));
    QIL_HIP(qil_stream_sync(ctx));
    qil_ctx_free(ctx, dout);
    return QIL_OK;
}

extern "C" int qil_apply_coefficient_batch(const qil_mpo* W, const qil_mps* psi, int64_t nb,
                                           const uint8_t* bits, double* out) {
    QIL_REQUIRE(W && psi && (nb == 0 || (bits && out)), QIL_EINVAL_ARG, "apply_coefficient: null argument");
    QIL_REQUIRE(W->ctx == psi->ctx, QIL_EINVAL_ARG, "apply: MPO and MPS belong to different contexts");
    QIL_REQUIRE(W->n() == psi->n(), QIL_EINVAL_LENGTH,
                "apply: MPO and MPS must have the same number of sites. Found length(W)=%lld, length(psi)=%lld",
                (long long)W->n(), (long long)psi->n());
    QIL_REQUIRE(W->site_ids == psi->site_ids, QIL_EINVAL_SITES,
                "apply: MPO and MPS must have the same site indices.");
    if (nb == 0) return QIL_OK;
    qil_context* ctx = psi->ctx;
    QIL_TRY(qil_ctx_activate(ctx));
    qil_call_scope call_scope(ctx);
    const int64_t n = psi->n();
    std::vector<ChainSite> tab((size_t)n);
    long long msz = 1;
    for (int64_t i = 0; i < n; ++i) {
        tab[(size_t)i] = ChainSite{psi->site[(size_t)i], W->site[(size_t)i], (int)psi->dims[(size_t)i],
                                   (int)psi->dims[(size_t)i + 1], (int)W->dims[(size_t)i],
                                   (int)W->dims[(size_t)i + 1]};
        // M: Dr*cr, X[s']: Dr*cl
        msz = std::max<long long>(msz, W->dims[(size_t)i + 1] * psi->dims[(size_t)i + 1]);
        msz = std::max<long long>(msz, W->dims[(size_t)i + 1] * psi->dims[(size_t)i]);
    }
    uint8_t* dbits = nullptr;
    QIL_TRY(upload_bits(ctx, nb, n, bits, &dbits));
    void *scratch = nullptr, *dout = nullptr, *pin = nullptr, *dtab = nullptr;
    int slot = 0;
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)nb * 16, &dout));
    // Many queries on a wide product bond: the per-query chains (one workgroup each, vector ALU) give way to
    // batched MFMA GEMMs.  Tuning aid: QIL_LAZY_GEMM_MIN = smallest chi * D that takes the GEMM form.
    static const long long lazy_min = 1024;   // measured crossover: 1.0 vs 2.0 ms at 1024, 0.85 vs 0.77 at 512
    if (nb >= 16 && msz >= lazy_min) {
        int st = lazy_gemm_path(ctx, W, psi, nb, dbits, (c64*)dout);
        if (st == QIL_OK && hipMemcpyAsync(out, dout, (size_t)nb * 16, hipMemcpyDeviceToHost, qil_stream(ctx)) != hipSuccess)
            st = qil_fail(QIL_EHIP, "apply_coefficient: download failed");
        if (st == QIL_OK && qil_stream_sync(ctx) != hipSuccess) st = qil_fail(QIL_EHIP, "sync failed");
        qil_ctx_free(ctx, dout);
        qil_ctx_free(ctx, dbits);
        return st;
    }
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(4 * nb * msz) * 16, &scratch));
    QIL_TRY(qil_ctx_desc_acquire(ctx, tab.size() * sizeof(ChainSite), &pin, &dtab, &slot));
    memcpy(pin, tab.data(), tab.size() * sizeof(ChainSite));
    QIL_HIP(hipMemcpyAsync(dtab, pin, tab.size() * sizeof(ChainSite), hipMemcpyHostToDevice, qil_stream(ctx)));
    const bool wc = W->dtype == QIL_C64, ac = psi->dtype == QIL_C64;
#define LAUNCH_LAZY(TW, TA)                                                                             \
    hipLaunchKernelGGL((lazy_coefficient_chain<TW, TA>), dim3((unsigned)nb), dim3(kThreads), 0, qil_stream(ctx), \
                       (const ChainSite*)dtab, (int)n, dbits, (c64*)scratch, msz, (c64*)dout, psi->amplitude)
    if (wc && ac) LAUNCH_LAZY(c64, c64);
    else if (wc) LAUNCH_LAZY(c64, double);
    else if (ac) LAUNCH_LAZY(double, c64);
    else LAUNCH_LAZY(double, double);
#undef LAUNCH_LAZY
    QIL_HIP(hipGetLastError());
    QIL_TRY(qil_ctx_desc_commit(ctx, slot));
    QIL_HIP(hipMemcpyAsync(out, dout, (size_t)nb * 16, hipMemcpyDeviceToHost, qil_stream(ctx)));
    QIL_HIP(qil_stream_sync(ctx));
    qil_ctx_free(ctx, scratch);
    qil_ctx_free(ctx, dout);
    qil_ctx_free(ctx, dbits);
    return QIL_OK;
}

// All 2^F coefficients of the configurations that agree with `spec` on the fixed sites -- the (k, l) grid scans
// of docs/src/tutorials/zt.jl:283-309 and the Laplace-value sums of dt.jl:187-197 as ONE dense contraction
// instead of 2^F chains.  spec[i]: 0 / 1 = the site's bit is fixed, 2 = the site is summed (marginal),
// 3 = the site is free.  The running tensor T[idx, beta] (idx over the free sites so far) advances by one MFMA
// GEMM per site: a fixed site multiplies by the slice A[:, b, :], a summed site by A[:, 0, :] + A[:, 1, :], a
// free site by the whole site viewed as (chi_l x 2 chi_r) -- the product's (idx, s, beta) order IS the next
// T[(idx, s), beta], nothing is permuted.  Output index: free sites in chain order, first free site = most
// significant bit (reverse = 0, as mps_to_vector) or least significant (reverse = 1).
extern "C" int qil_mps_block(const qil_mps* psi, const uint8_t* spec, int reverse, void* host_out) {
    QIL_REQUIRE(psi && spec && host_out, QIL_EINVAL_ARG, "mps_block: null argument");
    qil_context* ctx = psi->ctx;
    QIL_TRY(qil_ctx_activate(ctx));
    qil_call_scope call_scope(ctx);
    const int64_t n = psi->n();
    int nfree = 0;
    for (int64_t i = 0; i < n; ++i) {
        QIL_REQUIRE(spec[i] <= 3, QIL_EINVAL_CONFIG, "mps_block: spec value %d outside [0,3]", (int)spec[i]);
        nfree += spec[i] == 3;
    }
    QIL_REQUIRE(nfree <= 34, QIL_EINVAL_LENGTH, "mps_block: %d free sites is too many for a dense block", nfree);
    const int dt = psi->dtype;
    const size_t e = qil_elem_size(dt);
    long long maxel = 1, rows = 1, maxslice = 1;
    for (int64_t i = 0; i < n; ++i) {
        if (spec[i] == 3) rows *= 2;
        maxel = std::max<long long>(maxel, rows * psi->dims[(size_t)i + 1]);
        maxslice = std::max<long long>(maxslice, psi->dims[(size_t)i] * psi->dims[(size_t)i + 1]);
    }
    void *bufA = nullptr, *bufB = nullptr, *sl = nullptr;
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)maxel * e, &bufA));
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)maxel * e, &bufB));
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)maxslice * e, &sl));
    const double one[2] = {1.0, 0.0};
    QIL_HIP(hipMemcpyAsync(bufA, one, e, hipMemcpyHostToDevice, qil_stream(ctx)));
    QIL_HIP(qil_stream_sync(ctx));
    // the fixed / summed sites BEHIND the last free one fold into a right boundary vector first (O(chi^2) each,
    // instead of dragging the 2^F rows of T through them)
    int64_t last_free = -1;
    for (int64_t i = 0; i < n; ++i)
        if (spec[i] == 3) last_free = i;
    long long maxchi = 1;
    for (int64_t i = 0; i <= n; ++i) maxchi = std::max<long long>(maxchi, psi->dims[(size_t)i]);
    void *rv = nullptr, *rv2 = nullptr;
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)maxchi * e, &rv));
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)maxchi * e, &rv2));
    QIL_HIP(hipMemcpyAsync(rv, one, e, hipMemcpyHostToDevice, qil_stream(ctx)));
    QIL_HIP(qil_stream_sync(ctx));
    auto site_slice = [&](int64_t i, const void** B, long long* ldb) -> int {   // the (chi_l x chi_r) factor of site i
        const long long cl = psi->dims[(size_t)i], cr = psi->dims[(size_t)i + 1];
        const char* A = static_cast<const char*>(psi->site[(size_t)i]);
        if (spec[i] == 2) {
            const unsigned g = (unsigned)std::min<long long>((cl * cr + 255) / 256, 4096);
            if (dt == QIL_C64)
                hipLaunchKernelGGL(slice_sum<c64>, dim3(g), dim3(256), 0, qil_stream(ctx), (const c64*)A, (int)cl, (int)cr, (c64*)sl);
            else
                hipLaunchKernelGGL(slice_sum<double>, dim3(g), dim3(256), 0, qil_stream(ctx), (const double*)A, (int)cl, (int)cr,
                                   (double*)sl);
            *B = sl;
            *ldb = cl;
        } else {
            *B = A + (size_t)spec[i] * (size_t)cl * e;
            *ldb = 2 * cl;
        }
        return QIL_OK;
    };
    for (int64_t i = n - 1; i > last_free; --i) {
        const long long cl = psi->dims[(size_t)i], cr = psi->dims[(size_t)i + 1];
        const void* B = nullptr;
        long long ldb = 0;
        QIL_TRY(site_slice(i, &B, &ldb));
        QIL_TRY(qil_dev_gemm(ctx, dt, 0, 0, cl, 1, cr, B, ldb, rv, cr, rv2, cl));
        std::swap(rv, rv2);
    }
    void *cur = bufA, *nxt = bufB;
    rows = 1;
    for (int64_t i = 0; i <= last_free; ++i) {
        const long long cl = psi->dims[(size_t)i], cr = psi->dims[(size_t)i + 1];
        if (spec[i] == 3) {
            QIL_TRY(qil_dev_gemm(ctx, dt, 0, 0, rows, 2 * cr, cl, cur, rows, psi->site[(size_t)i], cl, nxt, rows));
            rows *= 2;
        } else {
            const void* B = nullptr;
            long long ldb = 0;
            QIL_TRY(site_slice(i, &B, &ldb));
            QIL_TRY(qil_dev_gemm(ctx, dt, 0, 0, rows, cr, cl, cur, rows, B, ldb, nxt, rows));
        }
        std::swap(cur, nxt);
    }
    {   // close with the boundary vector: T (rows x chi) r (chi x 1)
        const long long cb = psi->dims[(size_t)(last_free + 1)];
        QIL_TRY(qil_dev_gemm(ctx, dt, 0, 0, rows, 1, cb, cur, rows, rv, cb, nxt, rows));
        std::swap(cur, nxt);
    }
    // natural order: the first free site is the LOWEST bit of idx
    const unsigned g = (unsigned)std::min<long long>((rows + 255) / 256, 65536);
    if (dt == QIL_C64)
        hipLaunchKernelGGL(bit_reverse_scale<c64>, dim3(g), dim3(256), 0, qil_stream(ctx), (const c64*)cur, (c64*)nxt, nfree,
                           psi->amplitude, reverse ? 0 : 1);
    else
        hipLaunchKernelGGL(bit_reverse_scale<double>, dim3(g), dim3(256), 0, qil_stream(ctx), (const double*)cur, (double*)nxt,
                           nfree, psi->amplitude, reverse ? 0 : 1);
    QIL_HIP(hipGetLastError());
    QIL_HIP(hipMemcpyAsync(host_out, nxt, (size_t)rows * e, hipMemcpyDeviceToHost, qil_stream(ctx)));
    QIL_HIP(qil_stream_sync(ctx));
    qil_ctx_free(ctx, bufA);
    qil_ctx_free(ctx, bufB);
    qil_ctx_free(ctx, sl);
    qil_ctx_free(ctx, rv);
    qil_ctx_free(ctx, rv2);
    return QIL_OK;
}

extern "C" int qil_mps_to_vector(const qil_mps* psi, int reverse, void* host_out) {
    // every site free: the dense block read-out above (one MFMA GEMM per site; at n = 24 with 4096-wide bonds 0.73 s
    // of vector-ALU steps became a few tens of ms)
    QIL_REQUIRE(psi && host_out, QIL_EINVAL_ARG, "mps_to_vector: null argument");
    QIL_REQUIRE(psi->n() <= 34, QIL_EINVAL_LENGTH, "mps_to_vector: %lld sites is too many for a dense vector",
                (long long)psi->n());
    std::vector<uint8_t> spec((size_t)psi->n(), (uint8_t)3);
    return qil_mps_block(psi, spec.data(), reverse, host_out);
}

// norm(psi) = sqrt(|<psi|psi>|): E' = A^H (E A) per site, two GEMMs on the matricised site tensor.
extern "C" int qil_norm(const qil_mps* psi, double* out) {
    QIL_REQUIRE(psi && out, QIL_EINVAL_ARG, "norm: null argument");
    qil_context* ctx = psi->ctx;
    QIL_TRY(qil_ctx_activate(ctx));
    qil_call_scope call_scope(ctx);
    const int64_t n = psi->n();
    const size_t esz = qil_elem_size(psi->dtype);
    long long maxchi = 1;
    for (int64_t i = 0; i <= n; ++i) maxchi = std::max<long long>(maxchi, psi->dims[(size_t)i]);
    void *E = nullptr, *En = nullptr, *T = nullptr;
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(maxchi * maxchi) * esz, &E));
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(maxchi * maxchi) * esz, &En));
    QIL_TRY(qil_ctx_alloc(ctx, (size_t)(2 * maxchi * maxchi) * esz, &T));
    const double one[2] = {1.0, 0.0};
    QIL_HIP(hipMemcpyAsync(E, one, esz, hipMemcpyHostToDevice, qil_stream(ctx)));
    QIL_HIP(qil_stream_sync(ctx));
    for (int64_t i = 0; i < n; ++i) {
        const int64_t cl = psi->dims[(size_t)i], cr = psi->dims[(size_t)i + 1];
        // T (cl x 2cr) = E (cl x cl) * A (cl x 2cr);   E[alpha', alpha]
        QIL_TRY(qil_dev_gemm(ctx, psi->dtype, 0, 0, cl, 2 * cr, cl, E, cl, psi->site[(size_t)i], cl, T, cl));
        // E' (cr x cr) = A^H ((2cl) x cr)^H * T ((2cl) x cr);  E'[beta', beta]
        QIL_TRY(qil_dev_gemm(ctx, psi->dtype, 2, 0, cr, cr, 2 * cl, psi->site[(size_t)i], 2 * cl, T, 2 * cl, En, cr));
        std::swap(E, En);
    }
    double h[2] = {0, 0};
    QIL_HIP(hipMemcpyAsync(h, E, esz, hipMemcpyDeviceToHost, qil_stream(ctx)));
    QIL_HIP(qil_stream_sync(ctx));
    *out = sqrt(sqrt(h[0] * h[0] + h[1] * h[1]));
    qil_ctx_free(ctx, E);
    qil_ctx_free(ctx, En);
    qil_ctx_free(ctx, T);
    return QIL_OK;
}
