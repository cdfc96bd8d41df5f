// The 64-column diagonal block of the Cholesky + inverse (chol_inv_block16 against the register-tiled chol_inv_block,
// qil_linalg.hip): both kernels on the same Hermitian positive definite block, checked against each other and against
// R^H R = G, R X = I, then timed as a launch train.  Includes the library source so that the product's own kernels run:
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
    T *G, *R[2], *X[2];
    int* flag;
    (void)hipMalloc(&G, Gh.size() * sizeof(T));
    for (int v = 0; v < 2; ++v) {
        (void)hipMalloc(&R[v], Gh.size() * sizeof(T));
        (void)hipMalloc(&X[v], Gh.size() * sizeof(T));
        (void)hipMemset(R[v], 0, Gh.size() * sizeof(T));
        (void)hipMemset(X[v], 0, Gh.size() * sizeof(T));
    }
    (void)hipMalloc(&flag, 256);
    (void)hipMemset(flag, 0, 256);
    (void)hipMemcpy(G, Gh.data(), Gh.size() * sizeof(T), hipMemcpyHostToDevice);
    auto k_old = &qil_k1<chol_inv_block_k<T, 2, 32>, const T*, long long, int, T*, long long, T*, long long, double, int*>;
    auto k_new = &qil_k1<chol_inv_block16_k<T>, const T*, long long, int, T*, long long, T*, long long, double, int*>;
    const size_t lds = chol16_lds<T>();
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_new), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    double us[2];
    for (int v = 0; v < 2; ++v) {
        const int reps = 200;
        for (int it = 0; it < 2; ++it) {
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0, 0);
            for (int r = 0; r < reps; ++r) {
                if (v == 0) hipLaunchKernelGGL(k_old, dim3(1), dim3(1024), 0, 0, (const T*)G, (long long)nb, nb, R[0], (long long)nb, X[0], (long long)nb, 1e-11, flag);
                else hipLaunchKernelGGL(k_new, dim3(1), dim3(256), lds, 0, (const T*)G, (long long)nb, nb, R[1], (long long)nb, X[1], (long long)nb, 1e-11, flag);
            }
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
        }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        us[v] = 1e3 * ms / reps;
    }
    int hflag = 0;
    (void)hipMemcpy(&hflag, flag, 4, hipMemcpyDeviceToHost);
    std::vector<H> Rh[2], Xh[2];
    double err_rr[2], err_rx[2];
    for (int v = 0; v < 2; ++v) {
        Rh[v].resize(Gh.size());
        Xh[v].resize(Gh.size());
        (void)hipMemcpy(Rh[v].data(), R[v], Gh.size() * sizeof(T), hipMemcpyDeviceToHost);
        (void)hipMemcpy(Xh[v].data(), X[v], Gh.size() * sizeof(T), hipMemcpyDeviceToHost);
        double e1m = 0, e2m = 0, gmax = 0;
        for (int i = 0; i < nb; ++i)
            for (int k = 0; k < nb; ++k) {
                H a = 0, b = 0;
                for (int t = 0; t < nb; ++t) {
                    H rti = Rh[v][t + (size_t)nb * i];
                    if constexpr (NC == 2) rti = std::conj(rti);
                    a += rti * Rh[v][t + (size_t)nb * k];
                    b += Rh[v][i + (size_t)nb * t] * Xh[v][t + (size_t)nb * k];
                }
                e1m = std::max(e1m, std::abs(a - Gh[i + (size_t)nb * k]));
                e2m = std::max(e2m, std::abs(b - H(i == k ? 1.0 : 0.0)));
                gmax = std::max(gmax, std::abs(Gh[i + (size_t)nb * k]));
            }
        err_rr[v] = e1m / gmax;
        err_rx[v] = e2m;
    }
    double dr = 0, dx = 0;
    for (size_t t = 0; t < Gh.size(); ++t) {
        dr = std::max(dr, std::abs(Rh[0][t] - Rh[1][t]));
        dx = std::max(dx, std::abs(Xh[0][t] - Xh[1][t]));
    }
    printf("%s nb=%d: register-tiled %.2f us, 16-column sub-blocks %.2f us per launch; |R^H R - G| / |G| %.1e / %.1e, |R X - I| %.1e / %.1e, "
           "old vs new: |dR| %.1e |dX| %.1e, flag %d\n",
           name, nb, us[0], us[1], err_rr[0], err_rr[1], err_rx[0], err_rx[1], dr, dx, hflag);
}

int main() {
    for (int nb : {64, 48, 40, 17, 16, 5}) run<double>(nb, "f64");
    for (int nb : {64, 33}) run<c64>(nb, "c64");
    return 0;
}
