// MPO x MPS apply (A1/A2) and MPO x MPO apply (A3) for gfx950.
//
// Replaces src/linalg/apply.jl:75-122 (+ :201-218, :124-199).  The reference makes three
// passes over every output site tensor (K=2 contraction :101, right combiner :114, left
// combiner :118); here each output element is produced once and stored once, directly in
// its fused layout
//     B[row, s, col],  row = alpha + chi_l * a,  col = beta + chi_r * b     (column-major)
//     B[row, s, col] = sum_{s'} W[a, s', s, b] * A[alpha, s', beta].
//
// Roofline: K = 2 contraction => 0.37-0.87 flop/B => HBM-STORE bound.  The
// kernel is organised around the store stream:
//   * one lane owns one output ROW; for fixed (s, col) the rows are contiguous in memory, so
//     every wave-level store is 64 x 16 B = 1 KiB contiguous (c64) -- full-line coalesced;
//   * the lane keeps its A[alpha, :, beta-tile] values in registers for the whole block, and
//     streams over the MPO bond b; the workgroup's slab of the MPO site W[a_lo..a_hi, :, :, b-chunk]
//     (<= 16 KiB) is staged in LDS once, so the inner loop reads 4 LDS entries (wave-broadcast when
//     chi_l >= 64) and issues nothing but stores to global memory;
//   * ALL sites of one apply are issued as ONE grouped launch (a device-side site table and a
//     block -> (site, tile) map), so small-chi applies are not launch-bound and large ones
//     keep > 2000 independent workgroups per site in flight over the 256 CUs.
#include "qil_internal.h"

namespace {

struct c64 {
    double re, im;
};

struct ApplySite {
    const void* W;
    const void* A;
    void* B;
    int Dl, Dr, cl, cr;
    long long R;           // Dl * cl   (rows)
    int row_tiles;         // ceil(R / ROWS)
    int beta_tiles;        // ceil(cr / TB)
    int b_chunks;          // ceil(Dr / NB)
    int pad;
    long long block_begin; // first workgroup of this site in the grouped grid
};

constexpr int kRows = 256;  // rows per workgroup = threads per workgroup
constexpr int kTB = 8;      // beta values cached in registers per lane
constexpr int kNB = 16;     // MPO right-bond values streamed per workgroup

template <class T>
struct is_complex {
    static constexpr bool value = false;
};
template <>
struct is_complex<c64> {
    static constexpr bool value = true;
};

template <class TW, class TA>
struct out_type {
    using type = c64;
};
template <>
struct out_type<double, double> {
    using type = double;
};

__device__ __forceinline__ double mad2(double w0, double a0, double w1, double a1) {
    return fma(w1, a1, w0 * a0);
}
__device__ __forceinline__ c64 mad2(c64 w0, double a0, c64 w1, double a1) {
    return c64{fma(w1.re, a1, w0.re * a0), fma(w1.im, a1, w0.im * a0)};
}
__device__ __forceinline__ c64 mad2(double w0, c64 a0, double w1, c64 a1) {
    return c64{fma(w1, a1.re, w0 * a0.re), fma(w1, a1.im, w0 * a0.im)};
}
__device__ __forceinline__ c64 mad2(c64 w0, c64 a0, c64 w1, c64 a1) {
    double re = w0.re * a0.re;
    re = fma(-w0.im, a0.im, re);
    re = fma(w1.re, a1.re, re);
    re = fma(-w1.im, a1.im, re);
    double im = w0.re * a0.im;
    im = fma(w0.im, a0.re, im);
    im = fma(w1.re, a1.im, im);
    im = fma(w1.im, a1.re, im);
    return c64{re, im};
}

template <bool NT>
__device__ __forceinline__ void store_out(double* p, double v) {
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}
template <bool NT>
__device__ __forceinline__ void store_out(c64* p, c64 v) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 t = {v.re, v.im};
    if (NT) __builtin_nontemporal_store(t, reinterpret_cast<d2*>(p)); else *reinterpret_cast<d2*>(p) = t;
}

// RPL = output rows per lane: 1 for complex results (one 16-B store per element), 2 for real results (two
// adjacent rows packed into one 16-B store), so every wave-level store is 1 KiB contiguous either way.
template <class TO>
struct rows_per_lane {
    static constexpr int value = sizeof(TO) == 16 ? 1 : 2;
};

template <bool NT>
__device__ __forceinline__ void store_pair(double* p, double v0, double v1) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 t = {v0, v1};
    if (NT) __builtin_nontemporal_store(t, reinterpret_cast<d2*>(p)); else *reinterpret_cast<d2*>(p) = t;
}

template <class TW, class TA, bool NT, int NBV, bool WLDS>
__global__ __launch_bounds__(kRows) void site_apply_grouped(const ApplySite* __restrict__ sites, int nsites) {
    using TO = typename out_type<TW, TA>::type;
    constexpr int RPL = rows_per_lane<TO>::value;
    constexpr int kTileRows = kRows * RPL;
    // ---- block -> site (wave-uniform binary search over the prefix table)
    const long long blk = blockIdx.x;
    int lo = 0, hi = nsites - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (sites[mid].block_begin <= blk) lo = mid; else hi = mid - 1;
    }
    const ApplySite S = sites[lo];
    long long local = blk - S.block_begin;
    // row tile fastest: concurrently resident workgroups cover whole output columns
    const int row_tile = (int)(local % S.row_tiles);
    local /= S.row_tiles;
    const int beta_tile = (int)(local % S.beta_tiles);
    const int b_chunk = (int)(local / S.beta_tiles);

    const long long R = S.R;
    const long long r_first = (long long)row_tile * kTileRows + (long long)threadIdx.x * RPL;
    const bool valid = r_first < R;
    if (!WLDS && !valid) return;
    long long rr[RPL];
    int a[RPL], alpha[RPL];
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
        rr[k] = min(r_first + k, R - 1);   // clamped: idle lanes (WLDS) and the odd last row stay on a real row
        a[k] = (int)(rr[k] / S.cl);
        alpha[k] = (int)(rr[k] - (long long)a[k] * S.cl);
    }
    const bool second = RPL == 2 && r_first + 1 < R;              // this lane's second row exists
    const bool packed = RPL == 2 && second && (R & 1) == 0;       // 16-B aligned pair for every column
    const int beta0 = beta_tile * kTB;
    const int nbeta = min(kTB, S.cr - beta0);
    const int b0 = b_chunk * NBV;
    const int b1 = min(b0 + NBV, S.Dr);

    const TA* __restrict__ A = static_cast<const TA*>(S.A);
    const TW* __restrict__ W = static_cast<const TW*>(S.W);
    TO* __restrict__ B = static_cast<TO*>(S.B);

    // ---- this lane's slice of the MPS site: A[alpha, s', beta0 .. beta0+TB)
    TA A0[RPL][kTB], A1[RPL][kTB];
#pragma unroll
    for (int k = 0; k < RPL; ++k)
#pragma unroll
        for (int t = 0; t < kTB; ++t) {
            if (t < nbeta) {
                const long long off = alpha[k] + (long long)S.cl * (2LL * (beta0 + t));
                A0[k][t] = A[off];
                A1[k][t] = A[off + S.cl];
            } else {
                A0[k][t] = TA{};
                A1[k][t] = TA{};
            }
        }

    const long long wstride = (long long)S.Dl;  // W[a, si, so, b]: a + Dl*(si + 2*(so + 2*b))
    // ---- WLDS: stage this workgroup's slab of the MPO site, W[a_lo..a_hi, :, :, b0..b1), in LDS once
    constexpr int kWCap = 16384 / (int)sizeof(TW);
    __shared__ TW wtile[WLDS ? kWCap : 1];
    const int a_lo = (int)(((long long)row_tile * kTileRows) / S.cl);
    const int a_hi = (int)(min((long long)row_tile * kTileRows + kTileRows - 1, R - 1) / S.cl);
    const int na = a_hi - a_lo + 1;
    const bool staged = WLDS && na * 4 * (b1 - b0) <= kWCap;
    if (WLDS) {
        if (staged)
            for (int idx = threadIdx.x; idx < na * 4 * (b1 - b0); idx += kRows) {
                const int al = idx % na, q = (idx / na) & 3, bl = idx / (4 * na);
                wtile[idx] = W[(a_lo + al) + wstride * (q + 4LL * (b0 + bl))];
            }
        __syncthreads();
        if (!valid) return;
    }
    for (int b = b0; b < b1; ++b) {
        TW w00[RPL], w10[RPL], w01[RPL], w11[RPL];
#pragma unroll
        for (int k = 0; k < RPL; ++k) {
            if (staged) {
                const TW* wl = wtile + (a[k] - a_lo) + na * 4 * (b - b0);
                w00[k] = wl[0];
                w10[k] = wl[na];
                w01[k] = wl[2 * na];
                w11[k] = wl[3 * na];
            } else {
                const TW* wp = W + a[k] + wstride * (4LL * b);
                w00[k] = wp[0];                     // s_in=0, s_out=0
                w10[k] = wp[wstride];               // s_in=1, s_out=0
                w01[k] = wp[2 * wstride];           // s_in=0, s_out=1
                w11[k] = wp[3 * wstride];           // s_in=1, s_out=1
            }
        }
        TO* bp = B + r_first + R * (2LL * ((long long)beta0 + (long long)S.cr * b));
#pragma unroll
        for (int t = 0; t < kTB; ++t) {
            if (nbeta == kTB || t < nbeta) {
                const TO v0 = mad2(w00[0], A0[0][t], w10[0], A1[0][t]);
                const TO v1 = mad2(w01[0], A0[0][t], w11[0], A1[0][t]);
                if constexpr (RPL == 2) {
                    const TO u0 = mad2(w00[1], A0[1][t], w10[1], A1[1][t]);
                    const TO u1 = mad2(w01[1], A0[1][t], w11[1], A1[1][t]);
                    if (packed) {
                        store_pair<NT>(bp, v0, u0);
                        store_pair<NT>(bp + R, v1, u1);
                    } else {
                        store_out<NT>(bp, v0);
                        store_out<NT>(bp + R, v1);
                        if (second) {
                            store_out<NT>(bp + 1, u0);
                            store_out<NT>(bp + R + 1, u1);
                        }
                    }
                } else {
                    store_out<NT>(bp, v0);
                    store_out<NT>(bp + R, v1);
                }
            }
            bp += 2 * R;
        }
    }
}

// ---- MPO x MPO site composition: out[(a1,a2), i, o, (b1,b2)] = sum_m W1[a1,i,m,b1] W2[a2,m,o,b2]
// (W1 acts first; W1 bond fastest in the fused bonds).  Tiny tensors: one thread per output element.
template <class T1, class T2>
__global__ void mpo_compose_site(const T1* __restrict__ W1, const T2* __restrict__ W2,
                                 typename out_type<T1, T2>::type* __restrict__ out, int D1l, int D1r, int D2l,
                                 int D2r) {
    const long long Dl = (long long)D1l * D2l, Dr = (long long)D1r * D2r;
    const long long total = Dl * 4 * Dr;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        long long t = idx;
        const int af = (int)(t % Dl);
        t /= Dl;
        const int i = (int)(t & 1);
        const int o = (int)((t >> 1) & 1);
        const int bf = (int)(t >> 2);
        const int a1 = af % D1l, a2 = af / D1l;
        const int b1 = bf % D1r, b2 = bf / D1r;
        const T1 x0 = W1[a1 + (long long)D1l * (i + 2 * (0 + 2LL * b1))];
        const T1 x1 = W1[a1 + (long long)D1l * (i + 2 * (1 + 2LL * b1))];
        const T2 y0 = W2[a2 + (long long)D2l * (0 + 2 * (o + 2LL * b2))];
        const T2 y1 = W2[a2 + (long long)D2l * (1 + 2 * (o + 2LL * b2))];
        out[idx] = mad2(x0, y0, x1, y1);
    }
}

__global__ void widen_real_to_complex(const double* __restrict__ in, c64* __restrict__ out, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        out[i] = c64{in[i], 0.0};
}

int check_apply_operands(const qil_mpo* W, const qil_mps* psi, bool shared_state = false) {
    QIL_REQUIRE(W && psi, QIL_EINVAL_ARG, "apply: null handle");
    // shared_state: the batch runner's items read ONE state that stays in the home context (read-only for the duration of the
    // call, ordered behind the home stream by the batch's `ready` event) while their operators sit in the slots' contexts
    QIL_REQUIRE(shared_state || W->ctx == psi->ctx, QIL_EINVAL_ARG, "apply: MPO and MPS belong to different contexts");
    if (W->paired || psi->paired) {
        // apply(W::PairedSiteMPO, psi::ZTMPS): length(W.data) == 2 * length(psi.sites_main)  (apply.jl:202-203)
        QIL_REQUIRE(W->paired && psi->paired, QIL_EINVAL_ARG,
                    "apply: cannot mix paired and single-register operands");
        QIL_REQUIRE(W->n() == psi->n(), QIL_EINVAL_LENGTH, "apply: MPO and MPS must have compatible sizes.");
    } else {
        QIL_REQUIRE(W->n() == psi->n(), QIL_EINVAL_LENGTH,
                    "apply: MPO and MPS must have the same number of sites. Found length(W)=%lld, "
                    "length(psi)=%lld",
                    (long long)W->n(), (long long)psi->n());
    }
    QIL_REQUIRE(W->site_ids == psi->site_ids, QIL_EINVAL_SITES,
                "apply: MPO and MPS must have the same site indices.");
    return QIL_OK;
}

int launch_apply(const qil_mpo* W, const qil_mps* psi, qil_mps* out) {
    qil_context* ctx = W->ctx;
    const int64_t n = W->n();
    // the one shipped shape: MPO slab staged in LDS, non-temporal stores, 16 b per workgroup (measured against it in r02 and
    // removed: plain stores -3 %, 32-wide b chunks -3 %, MPO entries straight from L1/L2 -1.5 %, 8-wide chunks -2 %)
    const int nbv = kNB;
    // real x real results pack two rows per lane (16-B stores): 512-row tiles
    const int tile_rows = (W->dtype == QIL_F64 && psi->dtype == QIL_F64) ? 2 * kRows : kRows;
    std::vector<ApplySite> tab((size_t)n);
    long long blocks = 0;
    for (int64_t i = 0; i < n; ++i) {
        ApplySite& s = tab[(size_t)i];
        s.W = W->site[(size_t)i];
        s.A = psi->site[(size_t)i];
        s.B = out->site[(size_t)i];
        s.Dl = (int)W->dims[(size_t)i];
        s.Dr = (int)W->dims[(size_t)i + 1];
        s.cl = (int)psi->dims[(size_t)i];
        s.cr = (int)psi->dims[(size_t)i + 1];
        s.R = (long long)s.Dl * s.cl;
        s.row_tiles = (int)((s.R + tile_rows - 1) / tile_rows);
        s.beta_tiles = (s.cr + kTB - 1) / kTB;
        s.b_chunks = (s.Dr + nbv - 1) / nbv;
        s.pad = 0;
        s.block_begin = blocks;
        blocks += (long long)s.row_tiles * s.beta_tiles * s.b_chunks;
    }
    QIL_REQUIRE(blocks < (1LL << 31), QIL_EINVAL_ARG, "apply: grid too large (%lld workgroups)", blocks);
    const size_t bytes = tab.size() * sizeof(ApplySite);
    void *pin = nullptr, *dev = nullptr;
    int slot = 0;
    QIL_TRY(qil_ctx_desc_acquire(ctx, bytes, &pin, &dev, &slot));
    memcpy(pin, tab.data(), bytes);
    QIL_HIP(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, qil_stream(ctx)));
    QIL_TRY(qil_ctx_prof_begin(ctx));
    const ApplySite* dtab = static_cast<const ApplySite*>(dev);
    const dim3 grid((unsigned)blocks), block(kRows);
    const bool wc = W->dtype == QIL_C64, ac = psi->dtype == QIL_C64;
#define QIL_APPLY_LAUNCH(TW, TA) \
    hipLaunchKernelGGL((site_apply_grouped<TW, TA, true, 16, true>), grid, block, 0, qil_stream(ctx), dtab, (int)n)
    if (wc && ac) QIL_APPLY_LAUNCH(c64, c64);
    else if (wc) QIL_APPLY_LAUNCH(c64, double);
    else if (ac) QIL_APPLY_LAUNCH(double, c64);
    else QIL_APPLY_LAUNCH(double, double);
#undef QIL_APPLY_LAUNCH
    QIL_HIP(hipGetLastError());
    QIL_TRY(qil_ctx_prof_end(ctx));
    return qil_ctx_desc_commit(ctx, slot);
}

}  // namespace

extern "C" int qil_apply_into(const qil_mpo* W, const qil_mps* psi, qil_mps* out) {
    QIL_TRY(check_apply_operands(W, psi));
    QIL_REQUIRE(out, QIL_EINVAL_ARG, "qil_apply_into: null out");
    QIL_REQUIRE(out->ctx == W->ctx && out->n() == W->n(), QIL_EINVAL_LENGTH,
                "qil_apply_into: output handle has the wrong number of sites");
    const int odt = (W->dtype == QIL_C64 || psi->dtype == QIL_C64) ? QIL_C64 : QIL_F64;
    QIL_REQUIRE(out->dtype == odt, QIL_EINVAL_ARG, "qil_apply_into: output dtype mismatch");
    for (int64_t i = 0; i <= W->n(); ++i)
        QIL_REQUIRE(out->dims[(size_t)i] == W->dims[(size_t)i] * psi->dims[(size_t)i], QIL_EINVAL_LENGTH,
                    "qil_apply_into: output bond %lld has dimension %lld, expected %lld", (long long)i,
                    (long long)out->dims[(size_t)i], (long long)(W->dims[(size_t)i] * psi->dims[(size_t)i]));
    QIL_TRY(qil_ctx_activate(W->ctx));
    qil_call_scope call_scope(W->ctx);
    out->amplitude = psi->amplitude;  // apply.jl:121, :216
    out->site_ids = psi->site_ids;
    out->paired = psi->paired;
    return launch_apply(W, psi, out);
}

static int apply_new(const qil_mpo* W, const qil_mps* psi, qil_mps** out, bool shared_state);

extern "C" int qil_apply(const qil_mpo* W, const qil_mps* psi, qil_mps** out) { return apply_new(W, psi, out, false); }

// apply with the result (and every launch) in W's context while psi belongs to another context of the same device: the items of
// a batch share the caller's state instead of cloning it into every slot (r04: the 64 slots of a damping sweep made 3 024
// device-to-device copies of psi's sites per sweep)
int qil_apply_shared_state(const qil_mpo* W, const qil_mps* psi, qil_mps** out) { return apply_new(W, psi, out, true); }

static int apply_new(const qil_mpo* W, const qil_mps* psi, qil_mps** out, bool shared_state) {
    QIL_TRY(check_apply_operands(W, psi, shared_state));
    QIL_REQUIRE(out, QIL_EINVAL_ARG, "qil_apply: null out");
    const int64_t n = W->n();
    std::vector<int64_t> bonds((size_t)(n > 1 ? n - 1 : 0));
    for (int64_t i = 0; i + 1 < n; ++i) bonds[(size_t)i] = W->dims[(size_t)i + 1] * psi->dims[(size_t)i + 1];
    const int odt = (W->dtype == QIL_C64 || psi->dtype == QIL_C64) ? QIL_C64 : QIL_F64;
    qil_mps* res = nullptr;
    QIL_TRY(qil_mps_alloc(W->ctx, n, odt, psi->paired, bonds.data(), psi->site_ids.data(), psi->amplitude, &res));
    int s = launch_apply(W, psi, res);
    if (s != QIL_OK) {
        qil_mps_destroy(res);
        return s;
    }
    *out = res;
    return QIL_OK;
}

static int apply_mpo_mpo_impl(const qil_mpo* W1, const qil_mpo* W2, qil_mpo** out, bool shared_second);

extern "C" int qil_apply_mpo_mpo(const qil_mpo* W1, const qil_mpo* W2, qil_mpo** out) {
    return apply_mpo_mpo_impl(W1, W2, out, false);
}

// the product in W1's context while W2 lives in another context of the same device, read-only for the duration of the call
// (the items of a build_zt_mpo batch share ONE paired QFT chain; the batch's `ready` event orders the slots behind its producer)
int qil_apply_mpo_mpo_shared(const qil_mpo* W1, const qil_mpo* W2, qil_mpo** out) { return apply_mpo_mpo_impl(W1, W2, out, true); }

static int apply_mpo_mpo_impl(const qil_mpo* W1, const qil_mpo* W2, qil_mpo** out, bool shared_second) {
    QIL_REQUIRE(W1 && W2 && out, QIL_EINVAL_ARG, "apply: null handle");
    QIL_REQUIRE(shared_second || W1->ctx == W2->ctx, QIL_EINVAL_ARG, "apply: MPOs belong to different contexts");
    QIL_REQUIRE(W1->paired == W2->paired, QIL_EINVAL_ARG, "apply: cannot mix paired and single-site MPOs");
    qil_context* ctx = W1->ctx;
    QIL_TRY(qil_ctx_activate(ctx));
    qil_call_scope call_scope(ctx);
    const int64_t n1 = W1->n(), n2 = W2->n();
    // 1. window (apply.jl:128-139)
    int64_t start1 = -1, start2 = -1;
    for (int64_t i = 0; i < n1 && start1 < 0; ++i)
        for (int64_t j = 0; j < n2; ++j)
            if (W1->site_ids[(size_t)i] == W2->site_ids[(size_t)j]) {
                start1 = i;
                start2 = j;
                break;
            }
    QIL_REQUIRE(start1 >= 0, QIL_EINVAL_SITES, "apply: No matching sites found");
    int64_t match = 0;
    while (start1 + match < n1 && start2 + match < n2 &&
           W1->site_ids[(size_t)(start1 + match)] == W2->site_ids[(size_t)(start2 + match)])
        ++match;
    // 2. base = the longer MPO, W1 if equal (apply.jl:141-147)
    const qil_mpo* base = n1 >= n2 ? W1 : W2;
    const int64_t base_start = n1 >= n2 ? start1 : start2;
    const int64_t nb = base->n();
    // The shorter operand must lie entirely inside the window: otherwise its bond at the window edge would
    // dangle (the reference leaves that index on the edge tensor and its SingleSiteMPO constructor throws,
    // mpo.jl check_singlesitempo) -- and the fused edge bond would not match the copied base neighbour.
    {
        const qil_mpo* other = n1 >= n2 ? W2 : W1;
        const int64_t other_start = n1 >= n2 ? start2 : start1;
        QIL_REQUIRE(other_start == 0 && match == other->n(), QIL_EINVAL_SITES,
                    "apply: MPOs overlap only partially (sites %lld..%lld of the shorter operand's %lld): its "
                    "bond at the window edge would be left dangling",
                    (long long)other_start + 1, (long long)(other_start + match), (long long)other->n());
    }
    std::vector<int64_t> dims(base->dims);
    for (int64_t i = 0; i <= match; ++i) {
        // bond to the left of window site i (i == match: right of the last window site)
        const int64_t d1 = W1->dims[(size_t)(start1 + i)], d2 = W2->dims[(size_t)(start2 + i)];
        dims[(size_t)(base_start + i)] = d1 * d2;
    }
    const int odt = (W1->dtype == QIL_C64 || W2->dtype == QIL_C64) ? QIL_C64 : QIL_F64;
    qil_mpo* res = nullptr;
    QIL_TRY(qil_mpo_alloc(ctx, nb, odt, base->paired, dims.data() + 1, base->site_ids.data(), &res));
    for (int64_t i = 0; i < nb; ++i) {
        const int64_t w = i - base_start;
        if (w >= 0 && w < match) {
            const int64_t i1 = start1 + w, i2 = start2 + w;
            const int D1l = (int)W1->dims[(size_t)i1], D1r = (int)W1->dims[(size_t)i1 + 1];
            const int D2l = (int)W2->dims[(size_t)i2], D2r = (int)W2->dims[(size_t)i2 + 1];
            const long long total = (long long)D1l * D2l * 4 * D1r * D2r;
            const int threads = 256;
            const int blocks = (int)std::min<long long>((total + threads - 1) / threads, 4096);
            const bool c1 = W1->dtype == QIL_C64, c2 = W2->dtype == QIL_C64;
            void* o = res->site[(size_t)i];
            const void *p1 = W1->site[(size_t)i1], *p2 = W2->site[(size_t)i2];
            if (c1 && c2)
                hipLaunchKernelGGL((mpo_compose_site<c64, c64>), dim3(blocks), dim3(threads), 0, qil_stream(ctx),
                                   (const c64*)p1, (const c64*)p2, (c64*)o, D1l, D1r, D2l, D2r);
            else if (c1)
                hipLaunchKernelGGL((mpo_compose_site<c64, double>), dim3(blocks), dim3(threads), 0, qil_stream(ctx),
                                   (const c64*)p1, (const double*)p2, (c64*)o, D1l, D1r, D2l, D2r);
            else if (c2)
                hipLaunchKernelGGL((mpo_compose_site<double, c64>), dim3(blocks), dim3(threads), 0, qil_stream(ctx),
                                   (const double*)p1, (const c64*)p2, (c64*)o, D1l, D1r, D2l, D2r);
            else
                hipLaunchKernelGGL((mpo_compose_site<double, double>), dim3(blocks), dim3(threads), 0,
                                   qil_stream(ctx), (const double*)p1, (const double*)p2, (double*)o, D1l, D1r, D2l,
                                   D2r);
        } else {
            // non-overlapping sites are copied from the base MPO (apply.jl:150-153); promote if needed
            if (base->dtype == odt) {
                QIL_HIP(hipMemcpyAsync(res->site[(size_t)i], base->site[(size_t)i], base->site_bytes(i),
                                       hipMemcpyDeviceToDevice, qil_stream(ctx)));
            } else {
                const long long ne = base->site_elems(i);
                hipLaunchKernelGGL(widen_real_to_complex, dim3((unsigned)std::min<long long>((ne + 255) / 256, 1024)),
                                   dim3(256), 0, qil_stream(ctx), (const double*)base->site[(size_t)i],
                                   (c64*)res->site[(size_t)i], ne);
            }
        }
    }
    QIL_HIP(hipGetLastError());
    *out = res;
    return QIL_OK;
}
