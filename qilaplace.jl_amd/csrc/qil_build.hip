// Batched device builder of the Damping-Transform MPO for a sweep of damping values.
//
//   build_dt_mpo(n, wr; cutoff=1e-14, maxdim=1000)   src/transforms/dt_transformer.jl:312-407
//   zip_to_combine_mpos :20-164, zip_to_compress_mpo :167-288, gate blocks src/circuits/dt_gates.jl:30-229
//
// The reference (and the host builder) runs one chain of ~5000 tiny QR/SVD factorizations per damping
// value, one after another.  Here ALL damping values advance together: every tensor is a batch (one slice
// per sigma, identical shapes), every step is one launch with one workgroup per sigma, and each
// factorization runs entirely inside its workgroup (Gram-Schmidt QR, one-sided Jacobi SVD staged in LDS).
// Ranks differ between sigmas after a truncation; the batch keeps a common shape by padding every slice to
// the batch maximum with zero columns/rows (a zero bond component carries nothing, the operator is
// unchanged).  Only gauge-invariant results are comparable with the reference (dense operator, bond dims).
#include <algorithm>
#include <cmath>
#include <numeric>

#include "qil_internal.h"
#include "qil_device_utils.h"

namespace {
using namespace qil_dev;

struct BSite {
    void* p = nullptr;
    int dl = 1, dr = 1;
    long long elems() const { return (long long)dl * 4 * dr; }
};
struct BChain {
    qil_context* ctx = nullptr;
    int B = 0;
    std::vector<BSite> s;
};

inline unsigned nblk(long long total) { return (unsigned)std::min<long long>((total + 255) / 256, 4096); }

int balloc(qil_context* ctx, int B, int dl, int dr, BSite* out) {
    out->dl = dl;
    out->dr = dr;
    return qil_ctx_alloc(ctx, (size_t)B * out->elems() * sizeof(double), &out->p);
}
void bfree(qil_context* ctx, BSite& s) {
    if (s.p) qil_ctx_free(ctx, s.p);
    s.p = nullptr;
}

// ---------------------------------------------------------------- kernels (one grid.y slice per sigma)
// core[r, i, o, b1, b2] = sum_{a,c,m} T[r,a,c] M[a,i,m,b1] Bk[c,m,o,b2]      (dt_transformer.jl:54-61)
__global__ void bz_core(const double* __restrict__ Tm, const double* __restrict__ M,
                        const double* __restrict__ Bk, double* __restrict__ core, int R, int Da, int Dc, int B1,
                        int B2) {
    const long long per = (long long)R * 4 * B1 * B2;
    const double* t0 = Tm + (long long)blockIdx.y * R * Da * Dc;
    const double* m0 = M + (long long)blockIdx.y * Da * 4 * B1;
    const double* b0 = Bk + (long long)blockIdx.y * Dc * 4 * B2;
    double* c0 = core + (long long)blockIdx.y * per;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < per;
         t += (long long)gridDim.x * blockDim.x) {
        long long u = t;
        const int r = (int)(u % R);
        u /= R;
        const int i = (int)(u & 1), o = (int)((u >> 1) & 1);
        u >>= 2;
        const int b1 = (int)(u % B1), b2 = (int)(u / B1);
        double acc = 0;
        for (int c = 0; c < Dc; ++c)
            for (int m = 0; m < 2; ++m) {
                const double bv = b0[c + Dc * (m + 2 * (o + 2 * b2))];
                if (bv == 0.0) continue;
                double part = 0;
                for (int a = 0; a < Da; ++a) part = fma(t0[r + R * (a + Da * c)], m0[a + Da * (i + 2 * (m + 2 * b1))], part);
                acc = fma(part, bv, acc);
            }
        c0[t] = acc;
    }
}

// C (m x n) = A (m x k) * B (k x n), column-major, one slice per sigma
__global__ void bgemm(const double* __restrict__ A, const double* __restrict__ Bm, double* __restrict__ C, int m,
                      int n, int k) {
    const double* a0 = A + (long long)blockIdx.y * m * k;
    const double* b0 = Bm + (long long)blockIdx.y * k * n;
    double* c0 = C + (long long)blockIdx.y * m * n;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < (long long)m * n;
         t += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(t % m), j = (int)(t / m);
        double acc = 0;
        for (int kk = 0; kk < k; ++kk) acc = fma(a0[i + (long long)m * kk], b0[kk + (long long)k * j], acc);
        c0[t] = acc;
    }
}

__global__ void bidentity(double* __restrict__ Q, int m) {
    double* q0 = Q + (long long)blockIdx.y * m * m;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < m * m; t += gridDim.x * blockDim.x)
        q0[t] = (t % m) == (t / m) ? 1.0 : 0.0;
}

// out[b, i, o, a] = in[a, i, o, b]   (mirror of one site: swap the two bond axes)
__global__ void bmirror(const double* __restrict__ in, double* __restrict__ out, int Da, int Db) {
    const long long per = (long long)Da * 4 * Db;
    const double* i0 = in + (long long)blockIdx.y * per;
    double* o0 = out + (long long)blockIdx.y * per;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < per;
         t += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(t % Db);
        const int io = (int)((t / Db) & 3);
        const int a = (int)(t / (4LL * Db));
        o0[t] = i0[a + (long long)Da * (io + 4LL * b)];
    }
}

__global__ void btranspose(const double* __restrict__ A, double* __restrict__ At, int m, int n) {
    const double* a0 = A + (long long)blockIdx.y * m * n;
    double* t0 = At + (long long)blockIdx.y * m * n;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < (long long)m * n;
         t += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(t % n), i = (int)(t / n);
        t0[t] = a0[i + (long long)m * j];
    }
}

// Thin QR of a tall slice (m > n) by CGS2 inside one workgroup; Q overwrites A, R is n x n.  A column
// whose residual is below 1e-13 of its own norm is numerically dependent: it is dropped as a ZERO column
// (zero R diagonal), which keeps the batch shape and leaves A = Q R intact.
template <bool LDS>
__global__ __launch_bounds__(256) void bgs_fused(double* __restrict__ A, double* __restrict__ Rm, int m, int n) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* c = reinterpret_cast<double*>(smem_raw);      // n
    __shared__ double red[8];
    double* a0 = A + (long long)blockIdx.x * m * n;
    double* r0 = Rm + (long long)blockIdx.x * n * n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the slice stays in LDS for the whole factorisation when it fits (every phase below is a dependent
    // round trip to wherever the slice lives: ~3x shorter from LDS than from L2)
    double* Aw = a0;
    long long la = m;
    if (LDS) {
        la = m | 1;
        Aw = c + (n + (n & 1));
        for (int t = tid; t < m * n; t += 256) Aw[(t % m) + la * (t / m)] = a0[t];
    }
    for (int t = tid; t < n * n; t += 256) r0[t] = 0.0;
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        double* y = Aw + la * j;
        double v = 0;
        for (int r = tid; r < m; r += 256) v += y[r] * y[r];
        v = wave_sum(v);
        if (lane == 0) red[wave] = v;
        // first projection's dot products share the barrier with the norm
        for (int pass = 0; pass < 2 && j > 0; ++pass) {
            if (m <= 256) {
                // short columns: one 16-lane DPP row per previous column (16 dot products in flight, 4-step reductions)
                const int l16 = tid & 15;
                for (int i = tid >> 4; i < j; i += 16) {
                    const double* qi = Aw + la * i;
                    double g = 0;
                    for (int r = l16; r < m; r += 16) g = fma(qi[r], y[r], g);
                    g = row16_sum(g);
                    if (l16 == 0) c[i] = g;
                }
            } else {
                for (int i = wave; i < j; i += 4) {
                    const double* qi = Aw + la * i;
                    double g = 0;
                    for (int r = lane; r < m; r += 64) g = fma(qi[r], y[r], g);
                    g = wave_sum(g);
                    if (lane == 0) c[i] = g;
                }
            }
            __syncthreads();
            double w = 0;
            for (int r = tid; r < m; r += 256) {
                // four independent chains, the loads of a group issued together
                double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
                int i = 0;
                for (; i + 4 <= j; i += 4) {
                    const double a0 = Aw[r + la * i], a1 = Aw[r + la * (i + 1)], a2 = Aw[r + la * (i + 2)],
                                 a3 = Aw[r + la * (i + 3)];
                    s0 = fma(a0, c[i], s0);
                    s1 = fma(a1, c[i + 1], s1);
                    s2 = fma(a2, c[i + 2], s2);
                    s3 = fma(a3, c[i + 3], s3);
                }
                for (; i < j; ++i) s0 = fma(Aw[r + la * i], c[i], s0);
                const double acc = y[r] - ((s0 + s1) + (s2 + s3));
                y[r] = acc;
                w = fma(acc, acc, w);
            }
            for (int i = tid; i < j; i += 256) r0[i + n * j] += c[i];
            if (pass == 1) {                       // residual norm comes with the last update
                w = wave_sum(w);
                if (lane == 0) red[4 + wave] = w;
            }
            __threadfence_block();
            __syncthreads();
        }
        if (j == 0) {
            if (lane == 0) red[4 + wave] = v;
            __syncthreads();
        }
        const double nrm0 = sqrt((red[0] + red[1]) + (red[2] + red[3]));
        const double nrm = sqrt((red[4] + red[5]) + (red[6] + red[7]));
        const bool dep = !(nrm > 1e-13 * nrm0) || nrm0 == 0.0;
        const double inv = dep ? 0.0 : 1.0 / nrm;
        for (int r = tid; r < m; r += 256) y[r] *= inv;
        if (tid == 0) r0[j + n * j] = dep ? 0.0 : nrm;
        __threadfence_block();
        __syncthreads();
    }
    if (LDS)
        for (int t = tid; t < m * n; t += 256) a0[t] = Aw[(t % m) + la * (t / m)];
}

template <bool LDS, int NT>
__global__ __launch_bounds__(NT) void bjacobi(double* __restrict__ A, double* __restrict__ norms, int m, int n,
                                              double tol) {
    extern __shared__ __attribute__((aligned(16))) char jf_smem[];
    __shared__ int s_rot;
    double* a0 = A + (long long)blockIdx.x * m * n;
    double* n0 = norms + (long long)blockIdx.x * n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* Aw = a0;
    int la = m;
    if (LDS) {
        la = m | 1;
        Aw = reinterpret_cast<double*>(jf_smem);
        for (int t = tid; t < m * n; t += NT) Aw[(t % m) + la * (t / m)] = a0[t];
        __syncthreads();
    }
    __shared__ double s_fro[NT / 64];
    double f = 0;
    for (int t = tid; t < m * n; t += NT) {
        const double x = Aw[(t % m) + la * (t / m)];
        f = fma(x, x, f);
    }
    f = wave_sum(f);
    if (lane == 0) s_fro[wave] = f;
    __syncthreads();
    f = 0;
    for (int w = 0; w < NT / 64; ++w) f += s_fro[w];
    const double negligible = 1e-30 * f;
    if (m <= 128)
        jacobi_sweeps_nov<double, 16, NT>(Aw, la, m, n, tol, 40, &s_rot, negligible);
    else
        jacobi_sweeps_nov<double, 64, NT>(Aw, la, m, n, tol, 40, &s_rot, negligible);
    for (int j = wave; j < n; j += NT / 64) {
        const double* a = Aw + la * j;
        double v = 0;
        for (int r = lane; r < m; r += 64) v += a[r] * a[r];
        v = wave_sum(v);
        if (lane == 0) n0[j] = sqrt(v);
    }
    if (LDS) {
        __syncthreads();
        for (int t = tid; t < m * n; t += NT) a0[t] = Aw[(t % m) + la * (t / m)];
    }
}

// C (m x n) = A (m x k) * B^T, B given as (n x k)
__global__ void bgemm_nt(const double* __restrict__ A, const double* __restrict__ Bm, double* __restrict__ C, int m,
                         int n, int k) {
    const double* a0 = A + (long long)blockIdx.y * m * k;
    const double* b0 = Bm + (long long)blockIdx.y * n * k;
    double* c0 = C + (long long)blockIdx.y * m * n;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < (long long)m * n;
         t += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(t % m), j = (int)(t / m);
        double acc = 0;
        for (int kk = 0; kk < k; ++kk) acc = fma(a0[i + (long long)m * kk], b0[j + (long long)n * kk], acc);
        c0[t] = acc;
    }
}

// C (m x n) = diag(s^2) A^T B, A given as (k x m), B (k x n); s[j] per sigma (stride sstride)
__global__ void bgemm_tn_scaled(const double* __restrict__ A, const double* __restrict__ Bm,
                                const double* __restrict__ s, double* __restrict__ C, int m, int n, int k,
                                int sstride) {
    const double* a0 = A + (long long)blockIdx.y * k * m;
    const double* b0 = Bm + (long long)blockIdx.y * k * n;
    const double* s0 = s + (long long)blockIdx.y * sstride;
    double* c0 = C + (long long)blockIdx.y * m * n;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < (long long)m * n;
         t += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(t % m), j = (int)(t / m);
        double acc = 0;
        for (int kk = 0; kk < k; ++kk) acc = fma(a0[kk + (long long)k * i], b0[kk + (long long)k * j], acc);
        c0[t] = acc * s0[i] * s0[i];
    }
}

// dst[i, j] (tr = 0) or dst[j, i] (tr = 1)  =  j < rank ? src[i, perm[j]] * scale[j] : 0     per sigma
__global__ void bgather(const double* __restrict__ src, int rows, int scols, const int* __restrict__ perm,
                        const double* __restrict__ scale, const int* __restrict__ rank, double* __restrict__ dst,
                        int rmax, int tr) {
    const double* s0 = src + (long long)blockIdx.y * rows * scols;
    const int* p0 = perm + (long long)blockIdx.y * scols;
    const double* sc0 = scale ? scale + (long long)blockIdx.y * scols : nullptr;
    const int rk = rank[blockIdx.y];
    double* d0 = dst + (long long)blockIdx.y * rows * rmax;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < (long long)rows * rmax;
         t += (long long)gridDim.x * blockDim.x) {
        int i, j;
        if (tr) {
            j = (int)(t % rmax);
            i = (int)(t / rmax);
        } else {
            i = (int)(t % rows);
            j = (int)(t / rows);
        }
        double v = 0.0;
        if (j < rk) {
            v = s0[i + (long long)rows * p0[j]];
            if (sc0) v *= sc0[j];
        }
        d0[t] = v;
    }
}

// ---------------------------------------------------------------- host orchestration
struct Builder {
    qil_context* ctx;
    int B;
    double cutoff;
    long long maxdim;
    // pinned host staging for the per-step singular values (down) and truncation decisions (up): two slots used
    // alternately, so a step's upload needs no second stream synchronisation (the NEXT step's download sync covers it)
    // and the device never idles while the host enqueues the following step
    char* pin = nullptr;
    size_t pin_slot_bytes = 0;
    int pin_next = 0;

    ~Builder() {
        if (pin) (void)hipHostFree(pin);
    }
    int pinned_slot(size_t bytes, char** out) {
        if (bytes > pin_slot_bytes) {
            QIL_HIP(qil_stream_sync(ctx));
            if (pin) QIL_HIP(hipHostFree(pin));
            pin = nullptr;
            pin_slot_bytes = std::max<size_t>(2 * bytes, 1 << 16);
            QIL_HIP(hipHostMalloc(reinterpret_cast<void**>(&pin), 2 * pin_slot_bytes, hipHostMallocDefault));
        }
        *out = pin + (size_t)pin_next * pin_slot_bytes;
        pin_next ^= 1;
        return QIL_OK;
    }

    int upload(const std::vector<double>& h, void** dev) {
        QIL_TRY(qil_ctx_alloc(ctx, h.size() * sizeof(double), dev));
        QIL_HIP(hipMemcpyAsync(*dev, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, qil_stream(ctx)));
        QIL_HIP(qil_stream_sync(ctx));
        return QIL_OK;
    }

    // block tensors of one gate block for all sigmas: gen(sigma_index, site) -> (dl, dr, values[dl*4*dr])
    template <class Gen>
    int make_block(int nsites, Gen gen, std::vector<BSite>* out) {
        out->assign((size_t)nsites, BSite{});
        for (int k = 0; k < nsites; ++k) {
            int dl = 0, dr = 0;
            std::vector<double> all;
            for (int b = 0; b < B; ++b) {
                std::vector<double> one;
                gen(b, k, &dl, &dr, &one);
                all.insert(all.end(), one.begin(), one.end());
            }
            (*out)[(size_t)k].dl = dl;
            (*out)[(size_t)k].dr = dr;
            QIL_TRY(upload(all, &(*out)[(size_t)k].p));
        }
        return QIL_OK;
    }

    int qr_tall_or_identity(double* Mat, int m, int n, BSite* Qsite_out_p, void** R_out, int* nb_out) {
        // Mat: B slices of m x n.  m <= n: Q = I_m, R = Mat (valid factorisation, no work);
        // m > n: CGS2 in place, R n x n.
        (void)Qsite_out_p;
        if (m <= n) {
            *nb_out = m;
            *R_out = nullptr;  // caller uses Mat itself as R
            return QIL_OK;
        }
        void* R = nullptr;
        QIL_TRY(qil_ctx_alloc(ctx, (size_t)B * n * n * sizeof(double), &R));
        const size_t lds = ((size_t)(n + (n & 1)) + (size_t)(m | 1) * n) * sizeof(double);
        if (lds <= 60 * 1024)
            hipLaunchKernelGGL(bgs_fused<true>, dim3(B), dim3(256), lds, qil_stream(ctx), Mat, (double*)R, m, n);
        else
            hipLaunchKernelGGL(bgs_fused<false>, dim3(B), dim3(256), (size_t)n * sizeof(double), qil_stream(ctx), Mat,
                               (double*)R, m, n);
        QIL_HIP(hipGetLastError());
        *R_out = R;
        *nb_out = n;
        return QIL_OK;
    }

    // zip_to_combine "down": B-block acts after M (dt_transformer.jl:38-95)
    int zip_lr(std::vector<BSite>& M, const std::vector<BSite>& Blk) {
        const int L2 = (int)Blk.size(), L1 = (int)M.size();
        void* T = nullptr;  // [r, a, c]
        int R = 1, Da = 1, Dc = 1;
        {
            std::vector<double> ones((size_t)B, 1.0);
            QIL_TRY(upload(ones, &T));
        }
        for (int k = 0; k < L2; ++k) {
            const int B1 = M[(size_t)k].dr, B2 = Blk[(size_t)k].dr;
            QIL_REQUIRE(M[(size_t)k].dl == Da && Blk[(size_t)k].dl == Dc, QIL_EINVAL_ARG, "zip: bond mismatch");
            const int rows = R * 4, cols = B1 * B2;
            void* core = nullptr;
            QIL_TRY(qil_ctx_alloc(ctx, (size_t)B * rows * cols * sizeof(double), &core));
            hipLaunchKernelGGL(bz_core, dim3(nblk((long long)rows * cols), B), dim3(256), 0, qil_stream(ctx),
                               (const double*)T, (const double*)M[(size_t)k].p, (const double*)Blk[(size_t)k].p,
                               (double*)core, R, Da, Dc, B1, B2);
            qil_ctx_free(ctx, T);
            void* Rm = nullptr;
            int nb = 0;
            QIL_TRY(qr_tall_or_identity((double*)core, rows, cols, nullptr, &Rm, &nb));
            bfree(ctx, M[(size_t)k]);
            if (Rm) {                       // tall: Q = core (in place), T = R
                M[(size_t)k].p = core;
                T = Rm;
            } else {                        // fat/square: Q = I, T = core
                void* Q = nullptr;
                QIL_TRY(qil_ctx_alloc(ctx, (size_t)B * rows * rows * sizeof(double), &Q));
                hipLaunchKernelGGL(bidentity, dim3(nblk((long long)rows * rows), B), dim3(256), 0, qil_stream(ctx),
                                   (double*)Q, rows);
                M[(size_t)k].p = Q;
                T = core;
            }
            M[(size_t)k].dl = R;
            M[(size_t)k].dr = nb;
            R = nb;
            Da = B1;
            Dc = B2;
        }
        QIL_REQUIRE(Dc == 1, QIL_EINVAL_ARG, "zip: block does not end with a dimension-1 bond");
        // T is [R, Da] per sigma (b2 = 1)
        if (L1 > L2) {
            BSite& nx = M[(size_t)L2];
            void* out = nullptr;
            QIL_TRY(qil_ctx_alloc(ctx, (size_t)B * R * 4 * nx.dr * sizeof(double), &out));
            hipLaunchKernelGGL(bgemm, dim3(nblk((long long)R * 4 * nx.dr), B), dim3(256), 0, qil_stream(ctx),
                               (const double*)T, (const double*)nx.p, (double*)out, R, 4 * nx.dr, Da);
            bfree(ctx, nx);
            nx.p = out;
            nx.dl = R;
        } else {
            BSite& lt = M[(size_t)L2 - 1];
            void* out = nullptr;
            QIL_TRY(qil_ctx_alloc(ctx, (size_t)B * lt.dl * 4 * Da * sizeof(double), &out));
            hipLaunchKernelGGL(bgemm, dim3(nblk((long long)lt.dl * 4 * Da), B), dim3(256), 0, qil_stream(ctx),
                               (const double*)lt.p, (const double*)T, (double*)out, lt.dl * 4, Da, R);
            bfree(ctx, lt);
            lt.p = out;
            lt.dr = Da;
        }
        QIL_HIP(hipGetLastError());
        qil_ctx_free(ctx, T);
        return QIL_OK;
    }

    // zip_to_compress "down": QR gauge sweep L->R, truncating two-site SVD sweep R->L (:185-230)
    int compress_lr(std::vector<BSite>& M) {
        const int L = (int)M.size();
        for (int i = 0; i + 1 < L; ++i) {
            const int m = M[(size_t)i].dl * 4, n = M[(size_t)i].dr;
            BSite& nx = M[(size_t)i + 1];
            if (m <= n) {
                // fat site: Q = I_m (an isometry), R = the site itself; the bond shrinks to m
                void *out = nullptr, *Q = nullptr;
                QIL_TRY(qil_ctx_alloc(ctx, (size_t)B * m * 4 * nx.dr * sizeof(double), &out));
                hipLaunchKernelGGL(bgemm, dim3(nblk((long long)m * 4 * nx.dr), B), dim3(256), 0, qil_stream(ctx),
                                   (const double*)M[(size_t)i].p, (const double*)nx.p, (double*)out, m, 4 * nx.dr, n);
                QIL_TRY(qil_ctx_alloc(ctx, (size_t)B * m * m * sizeof(double), &Q));
                hipLaunchKernelGGL(bidentity, dim3(nblk((long long)m * m), B), dim3(256), 0, qil_stream(ctx), (double*)Q, m);
                bfree(ctx, M[(size_t)i]);
                M[(size_t)i].p = Q;
                M[(size_t)i].dr = m;
                bfree(ctx, nx);
                nx.p = out;
                nx.dl = m;
                continue;
            }
            void* Rm = nullptr;
            int nb = 0;
            QIL_TRY(qr_tall_or_identity((double*)M[(size_t)i].p, m, n, nullptr, &Rm, &nb));
            void* out = nullptr;
            QIL_TRY(qil_ctx_alloc(ctx, (size_t)B * n * 4 * nx.dr * sizeof(double), &out));
            hipLaunchKernelGGL(bgemm, dim3(nblk((long long)n * 4 * nx.dr), B), dim3(256), 0, qil_stream(ctx),
                               (const double*)Rm, (const double*)nx.p, (double*)out, n, 4 * nx.dr, n);
            qil_ctx_free(ctx, Rm);
            bfree(ctx, nx);
            nx.p = out;
        }
        // Truncating sweep R -> L.  The reference factorises the two-site core M[i-1] M[i] (:207-229); with
        // everything left of the bond in isometric gauge (the QR sweep above) that core has the SAME singular
        // values and right singular vectors as the single tensor M[i] viewed as (bond | in, out, right bond),
        // so the SVD is taken of that 4x smaller matrix: Jacobi rotates its <= D columns (D = bond) instead of
        // 4 D, with no V accumulation -- Vh are the normalised rotated columns and U S = M[i] Vh^T is one GEMM.
        for (int i = L - 1; i >= 1; --i) {
            BSite& lf = M[(size_t)i - 1];
            BSite& rt = M[(size_t)i];
            const int d = rt.dl, w = 4 * rt.dr;            // M[i] as d x w
            const int cols = std::min(d, w), rows = std::max(d, w);
            const bool tall = d > w;                        // rare (only near the right edge)
            void* Wk = nullptr;                             // rows x cols work matrix whose columns get rotated
            QIL_TRY(qil_ctx_alloc(ctx, (size_t)B * rows * cols * sizeof(double), &Wk));
            if (!tall)
                hipLaunchKernelGGL(btranspose, dim3(nblk((long long)d * w), B), dim3(256), 0, qil_stream(ctx),
                                   (const double*)rt.p, (double*)Wk, d, w);                       // Wk = M[i]^T (w x d)
            else
                QIL_HIP(hipMemcpyAsync(Wk, rt.p, (size_t)B * d * w * sizeof(double), hipMemcpyDeviceToDevice,
                                       qil_stream(ctx)));
            void* nrm = nullptr;
            QIL_TRY(qil_ctx_alloc(ctx, (size_t)B * cols * sizeof(double), &nrm));
            const size_t lds = (size_t)((rows | 1) * cols) * sizeof(double);
            if (cols <= 32 && rows <= 128 && lds <= 60 * 1024) {
                hipLaunchKernelGGL((bjacobi<true, 256>), dim3(B), dim3(256), lds, qil_stream(ctx), (double*)Wk,
                                   (double*)nrm, rows, cols, 1e-15);
            } else if (lds <= 150 * 1024) {
                static bool attr = false;
                if (!attr) {
                    QIL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bjacobi<true, 1024>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
                    attr = true;
                }
                hipLaunchKernelGGL((bjacobi<true, 1024>), dim3(B), dim3(1024), lds, qil_stream(ctx), (double*)Wk,
                                   (double*)nrm, rows, cols, 1e-15);
            } else {
                hipLaunchKernelGGL((bjacobi<false, 1024>), dim3(B), dim3(1024), 0, qil_stream(ctx), (double*)Wk,
                                   (double*)nrm, rows, cols, 1e-15);
            }
            QIL_HIP(hipGetLastError());
            // slot layout: sig (B cols doubles, down) | inv (B cols doubles) | perm (B cols ints) | rank (B ints)  (up)
            const size_t nsig = (size_t)B * cols;
            const size_t up_bytes = nsig * sizeof(double) + nsig * sizeof(int) + (size_t)B * sizeof(int);
            char* slot = nullptr;
            QIL_TRY(pinned_slot(nsig * sizeof(double) + up_bytes, &slot));
            double* sig = reinterpret_cast<double*>(slot);
            double* inv = sig + nsig;
            int* perm = reinterpret_cast<int*>(inv + nsig);
            int* rank = perm + nsig;
            QIL_HIP(hipMemcpyAsync(sig, nrm, nsig * sizeof(double), hipMemcpyDeviceToHost, qil_stream(ctx)));
            QIL_HIP(qil_stream_sync(ctx));
            std::vector<double> S((size_t)cols);
            int rmax = 1;
            for (int b = 0; b < B; ++b) {
                int* p = perm + (size_t)b * cols;
                const double* sg = sig + (size_t)b * cols;
                std::iota(p, p + cols, 0);
                std::stable_sort(p, p + cols, [&](int x, int y) { return sg[x] > sg[y]; });
                for (int j = 0; j < cols; ++j) {
                    S[(size_t)j] = sg[p[j]];
                    inv[(size_t)b * cols + j] = S[(size_t)j] > 0 ? 1.0 / S[(size_t)j] : 0.0;
                }
                rank[b] = (int)qil_truncation_rank(S.data(), cols, cutoff, true, maxdim, 1);
                rmax = std::max(rmax, rank[b]);
            }
            void* dpack = nullptr;
            QIL_TRY(qil_ctx_alloc(ctx, up_bytes, &dpack));
            QIL_HIP(hipMemcpyAsync(dpack, inv, up_bytes, hipMemcpyHostToDevice, qil_stream(ctx)));
            const double* dinv = static_cast<const double*>(dpack);
            const int* dperm = reinterpret_cast<const int*>(dinv + nsig);
            const int* drank = dperm + nsig;
            void *Vh = nullptr, *US = nullptr, *nl = nullptr;   // Vh: rmax x w ; US: d x rmax
            QIL_TRY(qil_ctx_alloc(ctx, (size_t)B * rmax * w * sizeof(double), &Vh));
            QIL_TRY(qil_ctx_alloc(ctx, (size_t)B * d * rmax * sizeof(double), &US));
            if (!tall) {
                // M[i]^T = (Wk D^-1) D V^T : Vh = (Wk[:, perm] D^-1)^T (rmax x w) ; U S = M[i] Vh^T (d x rmax)
                hipLaunchKernelGGL(bgather, dim3(nblk((long long)w * rmax), B), dim3(256), 0, qil_stream(ctx),
                                   (const double*)Wk, w, cols, (const int*)dperm, (const double*)dinv,
                                   (const int*)drank, (double*)Vh, rmax, 1);
                hipLaunchKernelGGL(bgemm_nt, dim3(nblk((long long)d * rmax), B), dim3(256), 0, qil_stream(ctx),
                                   (const double*)rt.p, (const double*)Vh, (double*)US, d, rmax, w);
            } else {
                // M[i] = (Wk D^-1) D V^T with Wk = M[i] rotated (d x w): U S = Wk[:, perm] ; Vh = D^-2 (U S)^T M[i]
                hipLaunchKernelGGL(bgather, dim3(nblk((long long)d * rmax), B), dim3(256), 0, qil_stream(ctx),
                                   (const double*)Wk, d, cols, (const int*)dperm, (const double*)nullptr,
                                   (const int*)drank, (double*)US, rmax, 0);
                hipLaunchKernelGGL(bgemm_tn_scaled, dim3(nblk((long long)rmax * w), B), dim3(256), 0, qil_stream(ctx),
                                   (const double*)US, (const double*)rt.p, (const double*)dinv, (double*)Vh, rmax, w, d,
                                   cols);
            }
            // M[i-1] <- M[i-1] (U S)
            QIL_TRY(qil_ctx_alloc(ctx, (size_t)B * lf.dl * 4 * rmax * sizeof(double), &nl));
            hipLaunchKernelGGL(bgemm, dim3(nblk((long long)lf.dl * 4 * rmax), B), dim3(256), 0, qil_stream(ctx),
                               (const double*)lf.p, (const double*)US, (double*)nl, lf.dl * 4, rmax, d);
            QIL_HIP(hipGetLastError());
            qil_ctx_free(ctx, Wk);
            qil_ctx_free(ctx, nrm);
            qil_ctx_free(ctx, dpack);
            qil_ctx_free(ctx, US);
            bfree(ctx, lf);
            bfree(ctx, rt);
            lf.p = nl;
            lf.dr = rmax;
            rt.p = Vh;
            rt.dl = rmax;
        }
        return QIL_OK;
    }

    int mirror(std::vector<BSite>& M) {
        std::vector<BSite> out(M.size());
        for (size_t i = 0; i < M.size(); ++i) {
            BSite& src = M[M.size() - 1 - i];
            QIL_TRY(balloc(ctx, B, src.dr, src.dl, &out[i]));
            hipLaunchKernelGGL(bmirror, dim3(nblk(src.elems()), B), dim3(256), 0, qil_stream(ctx), (const double*)src.p,
                               (double*)out[i].p, src.dl, src.dr);
        }
        QIL_HIP(hipGetLastError());
        for (auto& s : M) bfree(ctx, s);
        M = out;
        return QIL_OK;
    }
};

// ---- gate blocks (host, tiny): src/circuits/dt_gates.jl
void put(std::vector<double>& W, int dl, int a, int b, const double g[4]) {
    // g[s_in + 2*s_out]; W[a + dl*(s_in + 2*(s_out + 2*b))]
    for (int si = 0; si < 2; ++si)
        for (int so = 0; so < 2; ++so) W[(size_t)(a + dl * (si + 2 * (so + 2 * b)))] += g[si + 2 * so];
}
const double kI[4] = {1, 0, 0, 1};

// control_damping_mpo(n, k, wr): 2k tensors (dt_gates.jl:30-130)
void dt_main_site(int k, double w, int site, int* dl, int* dr, std::vector<double>* out) {
    const double e2 = std::exp(-w / 2.0), is2 = 1.0 / std::sqrt(2.0);
    const double Hd[4] = {is2, is2, is2, e2 * is2};               // dampedH[s_in + 2 s_out] (symmetric)
    if (k == 1) {
        *dl = *dr = 1;
        out->assign(4, 0.0);
        put(*out, 1, 0, 0, site == 0 ? Hd : kI);
        return;
    }
    const int pair = site / 2 + 1;                                // 1-based pair index l
    const bool main = site % 2 == 0;
    if (pair < k) {
        if (main) {
            const double rf = std::exp(-w * std::pow(2.0, pair - k - 1));
            const double Rg[4] = {1, 0, 0, rf};
            *dl = pair == 1 ? 1 : 2;
            *dr = 2;
            out->assign((size_t)(*dl * 4 * *dr), 0.0);
            put(*out, *dl, 0, 0, kI);
            put(*out, *dl, pair == 1 ? 0 : 1, 1, Rg);
        } else {
            *dl = *dr = 2;
            out->assign(16, 0.0);
            put(*out, 2, 0, 0, kI);
            put(*out, 2, 1, 1, kI);
        }
        return;
    }
    if (main) {
        // (Pi_c @ Hd)[s_in, s_out] = delta(s_in, c) Hd[c, s_out] on bond values (c, c)
        *dl = *dr = 2;
        out->assign(16, 0.0);
        const double P0[4] = {Hd[0], 0, Hd[2], 0};                // s_in = 0 row: Hd[0, s_out]
        const double P1[4] = {0, Hd[1], 0, Hd[3]};                // s_in = 1 row: Hd[1, s_out]
        put(*out, 2, 0, 0, P0);
        put(*out, 2, 1, 1, P1);
    } else {
        *dl = 2;
        *dr = 1;
        out->assign(8, 0.0);
        put(*out, 2, 0, 0, kI);
        put(*out, 2, 1, 0, kI);
    }
}

// control_damping_copy_mpo(n, k, wr): pairs k..n, 2(n-k+1) tensors (dt_gates.jl:133-229)
void dt_copy_site(int n, int k, double w, int site, int* dl, int* dr, std::vector<double>* out) {
    const int Lp = n - k + 1;
    if (Lp == 1) {
        *dl = *dr = 1;
        out->assign(4, 0.0);
        put(*out, 1, 0, 0, kI);
        return;
    }
    const int j = site / 2 + 1;                                   // relative pair index
    const bool main = site % 2 == 0;
    if (j == 1) {
        if (main) {
            *dl = 1;
            *dr = 2;
            out->assign(8, 0.0);
            put(*out, 1, 0, 0, kI);
        } else {
            const double P0[4] = {1, 0, 0, 0}, P1[4] = {0, 0, 0, 1};
            *dl = *dr = 2;
            out->assign(16, 0.0);
            put(*out, 2, 0, 0, P0);
            put(*out, 2, 0, 1, P1);
        }
        return;
    }
    if (main) {
        const double rf = std::exp(-w * std::pow(2.0, j - 2));
        const double Rg[4] = {1, 0, 0, rf};
        *dl = *dr = 2;
        out->assign(16, 0.0);
        put(*out, 2, 0, 0, kI);
        put(*out, 2, 1, 1, Rg);
    } else {
        const bool last = j == Lp;
        *dl = 2;
        *dr = last ? 1 : 2;
        out->assign((size_t)(2 * 4 * *dr), 0.0);
        put(*out, 2, 0, 0, kI);
        put(*out, 2, 1, last ? 0 : 1, kI);
    }
}

}  // namespace

int qil_build_dt_persistent(qil_context* ctx, int64_t n, int64_t nb, const double* wrs, double cutoff, int64_t maxdim,
                            const int64_t* site_ids, qil_mpo** out, int* fallback);

extern "C" int qil_build_dt_mpo_batch(qil_context* ctx, int64_t n, int64_t nb, const double* wrs, double cutoff,
                                      int64_t maxdim, const int64_t* site_ids, qil_mpo** out) {
    QIL_REQUIRE(ctx && wrs && out, QIL_EINVAL_ARG, "build_dt_mpo: null argument");
    QIL_REQUIRE(n >= 1, QIL_EINVAL_ARG, "build_dt_mpo: n must be >= 1. Found n=%lld", (long long)n);
    QIL_REQUIRE(nb >= 1 && nb <= 4096, QIL_EINVAL_ARG, "build_dt_mpo: batch of %lld damping values", (long long)nb);
    QIL_TRY(qil_ctx_activate(ctx));
    qil_call_scope call_scope(ctx);
    // default route: the persistent kernel (one launch, one workgroup per damping value, true bond dimensions per
    // value); the launch-per-step route below serves bonds beyond its in-LDS capacity and QIL_DT_BUILDER=launches
    const char* mode = getenv("QIL_DT_BUILDER");
    if (!(mode && strcmp(mode, "launches") == 0)) {
        int fallback = 0;
        QIL_TRY(qil_build_dt_persistent(ctx, n, nb, wrs, cutoff, maxdim, site_ids, out, &fallback));
        if (!fallback) return QIL_OK;
    }
    // (the rule never runs below 1e-28: the batched Jacobi leaves rounding residue of rank-deficient bonds un-orthogonalised, see
    // qil_build_persist.hip persist_chunk)
    Builder bd{ctx, (int)nb, std::max(cutoff, 1e-28), maxdim <= 0 ? INT64_MAX : maxdim};
    const int B = (int)nb;
    std::vector<BSite> M;
    int st = bd.make_block(2, [&](int b, int site, int* dl, int* dr, std::vector<double>* o) {
        dt_main_site(1, wrs[b], site, dl, dr, o);
    }, &M);
    if (st != QIL_OK) return st;
    auto fail = [&](int code) {
        for (auto& s : M) bfree(ctx, s);
        return code;
    };
    for (int k = 2; k <= n; ++k) {                                            // part 1 (:351-390)
        for (int t = 0; t < 2; ++t) {                                         // identity pair, dim-1 bonds
            BSite s;
            std::vector<double> eye;
            for (int b = 0; b < B; ++b) eye.insert(eye.end(), {1.0, 0.0, 0.0, 1.0});
            s.dl = s.dr = 1;
            if ((st = bd.upload(eye, &s.p)) != QIL_OK) return fail(st);
            M.push_back(s);
        }
        std::vector<BSite> blk;
        st = bd.make_block(2 * k, [&](int b, int site, int* dl, int* dr, std::vector<double>* o) {
            dt_main_site(k, wrs[b], site, dl, dr, o);
        }, &blk);
        if (st == QIL_OK) st = bd.zip_lr(M, blk);
        for (auto& s : blk) bfree(ctx, s);
        if (st == QIL_OK) st = bd.compress_lr(M);
        if (st != QIL_OK) return fail(st);
    }
    if (n > 1) {                                                              // part 2 in the mirrored frame (:396-405)
        if ((st = bd.mirror(M)) != QIL_OK) return fail(st);
        for (int k = 1; k < n; ++k) {
            const int ns = 2 * (int)(n - k + 1);
            std::vector<BSite> blk;
            st = bd.make_block(ns, [&](int b, int site, int* dl, int* dr, std::vector<double>* o) {
                dt_copy_site((int)n, k, wrs[b], site, dl, dr, o);
            }, &blk);
            if (st == QIL_OK) st = bd.mirror(blk);
            if (st == QIL_OK) st = bd.zip_lr(M, blk);
            for (auto& s : blk) bfree(ctx, s);
            if (st == QIL_OK) st = bd.compress_lr(M);
            if (st != QIL_OK) return fail(st);
        }
        if ((st = bd.mirror(M)) != QIL_OK) return fail(st);
    }
    // hand out one PairedSiteMPO per damping value (zero-padded to the batch's common bond profile)
    const int L = (int)M.size();
    std::vector<int64_t> bonds((size_t)std::max(L - 1, 1));
    for (int i = 0; i + 1 < L; ++i) bonds[(size_t)i] = M[(size_t)i].dr;
    for (int b = 0; b < B; ++b) {
        qil_mpo* W = nullptr;
        st = qil_mpo_alloc(ctx, L, QIL_F64, 1, bonds.data(), site_ids, &W);
        if (st != QIL_OK) return fail(st);
        for (int i = 0; i < L; ++i) {
            const size_t bytes = (size_t)M[(size_t)i].elems() * sizeof(double);
            hipError_t e = hipMemcpyAsync(W->site[(size_t)i], static_cast<char*>(M[(size_t)i].p) + (size_t)b * bytes, bytes,
                                          hipMemcpyDeviceToDevice, qil_stream(ctx));
            if (e != hipSuccess) return fail(qil_fail(QIL_EHIP, "copy failed: %s", hipGetErrorString(e)));
        }
        out[b] = W;
    }
    QIL_HIP(qil_stream_sync(ctx));
    for (auto& s : M) bfree(ctx, s);
    return QIL_OK;
}
