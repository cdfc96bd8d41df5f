// The 64-column diagonal block of the Cholesky + inverse (chol_inv_block16, qil_linalg.hip) on a Hermitian positive definite
// block: checked against R^H R = G, R X = I, then timed as a launch train (the register-tiled kernel it replaced: 35.1 us
// f64 / 46.9 us c64 at 64 columns, r03 measurement, profiles/archive_r01-r04.tar.gz).  Includes the library source so that the product's own kernel runs:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I include -I qilaplace.jl_amd/csrc \
//         tools/micro/chol_block_cost.hip -o tools/micro/chol_block_cost.bin -L qilaplace.jl_amd/lib -lqilhip \
//         -Wl,-rpath,'$ORIGIN/../../qilaplace.jl_amd/lib'
#include "../../qilaplace.jl_amd/csrc/qil_linalg.hip"
#include <complex>
#include <random>

template <class T>
static void run(int nb, const char* name) {
    constexpr int NC = sizeof(T) / 8;
    using H = typename std::conditional<NC == 2, std::complex<double>, double>::type;
    const int m = 4 * nb;
    std::mt19937_64 rng(11);
    std::normal_distribution<double> nd;
    std::vector<H> A((size_t)m * nb), Gh((size_t)nb * nb);
    for (auto& v : A) {
        if constexpr (NC == 2) v = H(nd(rng), nd(rng));
        else v = nd(rng);
    }
    for (int i = 0; i < nb; ++i)
        for (int k = 0; k < nb; ++k) {
            H acc = 0;
            for (int r = 0; r < m; ++r) {
                if constexpr (NC == 2) acc += std::conj(A[r + (size_t)m * i]) * A[r + (size_t)m * k];
                else acc += A[r + (size_t)m * i] * A[r + (size_t)m * k];
            }
            Gh[i + (size_t)nb * k] = acc;
        }
    T *G, *R, *X;
    int* flag;
    (void)hipMalloc(&G, Gh.size() * sizeof(T));
    (void)hipMalloc(&R, Gh.size() * sizeof(T));
    (void)hipMalloc(&X, Gh.size() * sizeof(T));
    (void)hipMalloc(&flag, 256);
    (void)hipMemset(flag, 0, 256);
    (void)hipMemcpy(G, Gh.data(), Gh.size() * sizeof(T), hipMemcpyHostToDevice);
    auto kern = &qil_k1<chol_inv_block16_k<T>, const T*, long long, int, T*, long long, T*, long long, double, int*>;
    const size_t lds = chol16_lds<T>();
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int reps = 200;
    for (int it = 0; it < 2; ++it) {
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < reps; ++r)
            hipLaunchKernelGGL(kern, dim3(1), dim3(256), lds, 0, (const T*)G, (long long)nb, nb, R, (long long)nb, X, (long long)nb, 1e-11, flag);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    int hflag = 0;
    (void)hipMemcpy(&hflag, flag, 4, hipMemcpyDeviceToHost);
    std::vector<H> Rh(Gh.size()), Xh(Gh.size());
    (void)hipMemcpy(Rh.data(), R, Gh.size() * sizeof(T), hipMemcpyDeviceToHost);
    (void)hipMemcpy(Xh.data(), X, Gh.size() * sizeof(T), hipMemcpyDeviceToHost);
    double e1m = 0, e2m = 0, gmax = 0;
    for (int i = 0; i < nb; ++i)
        for (int k = 0; k < nb; ++k) {
            H a = 0, b = 0;
            for (int t = 0; t < nb; ++t) {
                H rti = Rh[t + (size_t)nb * i];
                if constexpr (NC == 2) rti = std::conj(rti);
                a += rti * Rh[t + (size_t)nb * k];
                b += Rh[i + (size_t)nb * t] * Xh[t + (size_t)nb * k];
            }
            e1m = std::max(e1m, std::abs(a - Gh[i + (size_t)nb * k]));
            e2m = std::max(e2m, std::abs(b - H(i == k ? 1.0 : 0.0)));
            gmax = std::max(gmax, std::abs(Gh[i + (size_t)nb * k]));
        }
    printf("%s nb=%d: %.2f us per launch; |R^H R - G| / |G| %.1e, |R X - I| %.1e, flag %d\n", name, nb, 1e3 * ms / reps, e1m / gmax, e2m, hflag);
}

int main() {
    for (int nb : {64, 48, 40, 17, 16, 5}) run<double>(nb, "f64");
    for (int nb : {64, 33}) run<c64>(nb, "c64");
    return 0;
}
