// Where the Householder panel kernel of the blocked QR spends its time (s_memtime stamps, 100 MHz): staging, the column
// loop (one barrier per column), R out, explicit Q.  Includes the library source so the product's own kernel is timed:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I include -I qilaplace.jl_amd/csrc \
//         tools/micro/hh_panel_cost.hip -o tools/micro/hh_panel_cost.bin
#include "../../qilaplace.jl_amd/csrc/qil_linalg.hip"
#include <random>

template <class T, int KM>
static void run(int m, int b, const char* name) {
    auto kern = &qil_k1<hh_panel_k<T, KM, true>, T*, long long, int, int, T*, long long, const double*, long long*>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    const size_t lds = (size_t)((m + 1) | 1) * b * sizeof(T) + 2048;
    long long* prof;
    (void)hipMalloc(&prof, 64);
    std::vector<double> h((size_t)m * b * (sizeof(T) / 8));
    std::mt19937_64 rng(7);
    std::normal_distribution<double> nd;
    for (auto& v : h) v = nd(rng);
    T *P0, *P, *R;
    (void)hipMalloc(&P0, h.size() * 8);
    (void)hipMalloc(&P, h.size() * 8);
    (void)hipMalloc(&R, (size_t)b * b * sizeof(T));
    (void)hipMemcpy(P0, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipMemset(prof, 0, 64);
    const int reps = 20;
    float tot = 0;
    for (int r = 0; r < reps + 2; ++r) {
        (void)hipMemcpy(P, P0, h.size() * 8, hipMemcpyDeviceToDevice);
        if (r == 2) (void)hipMemset(prof, 0, 64);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(1), dim3(64 * hh_panel_waves<T, KM>()), lds, 0, P, (long long)m, m, b, R, (long long)b,
                           (const double*)nullptr, prof);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2) tot += ms;
    }
    long long hp[5];
    (void)hipMemcpy(hp, prof, 40, hipMemcpyDeviceToHost);
    const double nl = (double)hp[4];
    printf("%s %d x %d (KM %d, %d waves): %.1f us per launch; in-kernel (us at 100 MHz) stage %.1f, column loop %.1f (%.2f per column), "
           "R out %.1f, form Q %.1f\n", name, m, b, KM, hh_panel_waves<T, KM>(), 1e3 * tot / reps, hp[0] / nl / 100.0, hp[1] / nl / 100.0,
           hp[1] / nl / 100.0 / b, hp[2] / nl / 100.0, hp[3] / nl / 100.0);
    (void)hipFree(P0); (void)hipFree(P); (void)hipFree(R); (void)hipFree(prof);
}

int main() {
    run<double, 4>(256, 32, "f64");
    run<double, 8>(512, 32, "f64");
    run<double, 2>(128, 32, "f64");
    run<c64, 4>(256, 32, "c64");
    run<c64, 8>(512, 32, "c64");
    run<double, 13>(800, 32, "f64");
    return 0;
}
