// Where a block round of the mid-size Jacobi SVD spends its time: fixed cost (launch + staging of the two column blocks)
// versus cost per inner round, from event-timed launch trains with 0 extra / 7 extra inner rounds.  Includes the library
// source so that the very kernel the product runs is timed:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I include -I qilaplace.jl_amd/csrc \
//         tools/micro/jacobi_round_cost.hip -o tools/micro/jacobi_round_cost.bin
#include "../../qilaplace.jl_amd/csrc/qil_linalg.hip"
#include <random>

template <class T, int BB, int KM, int G>
static void run(int k, const char* name) {
    const int nblk = ((k + BB - 1) / BB + 1) / 2 * 2;
    const size_t lds = block_round_nov_lds<T, BB, KM, G>();
    auto kern_ap = &qil_k1<jacobi_block_round_nov_k<T, BB, KM, G, true, true>, T*, long long, int, int, int, int, double, int*, const double*, long long*>;
    auto kern_x = &qil_k1<jacobi_block_round_nov_k<T, BB, KM, G, false, true>, T*, long long, int, int, int, int, double, int*, const double*, long long*>;
    long long* prof;
    (void)hipMalloc(&prof, 64);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern_ap), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern_x), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    std::vector<double> h((size_t)k * k * (sizeof(T) / 8));
    std::mt19937_64 rng(7);
    std::normal_distribution<double> nd;
    T* X;
    int* flag;
    (void)hipMalloc(&X, h.size() * 8);
    (void)hipMalloc(&flag, 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {   // 0: cross rounds (BB inner), 1: all-pairs rounds (2 BB - 1 inner), 2: cross, tol = 1 (no rotation)
        for (auto& v : h) v = nd(rng);
        (void)hipMemcpy(X, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        const int sweeps = 3;
        (void)hipMemset(prof, 0, 64);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        int launches = 0;
        for (int s = 0; s < sweeps; ++s)
            for (int r = 0; r < nblk - 1; ++r, ++launches)
                hipLaunchKernelGGL(mode == 1 ? kern_ap : kern_x, dim3(nblk / 2), dim3(BB * G), lds, 0, X, (long long)k, k, k, nblk, r,
                                   mode == 2 ? 1.0 : 1e-15, flag, (const double*)nullptr, prof);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        long long hp[5];
        (void)hipMemcpy(hp, prof, 40, hipMemcpyDeviceToHost);
        const double nl = (double)hp[4];
        printf("%s k=%d BB=%d G=%d mode=%d: %.2f us per launch (%d WGs, %d inner rounds); in-kernel cycles stage-in %.0f, inner %.0f "
               "(%.0f per round), stage-out %.0f; in-kernel %.2f us at %.0f MHz\n",
               name, k, BB, G, mode, 1e3 * ms / launches, nblk / 2, mode == 1 ? 2 * BB - 1 : BB, hp[0] / nl, hp[1] / nl,
               hp[1] / nl / (mode == 1 ? 2 * BB - 1 : BB), hp[2] / nl, hp[3] / nl / 100.0,
               (hp[0] + hp[1] + hp[2]) / (double)hp[3] * 100.0);
    }
    (void)hipFree(X);
    (void)hipFree(flag);
}

int main() {
    run<double, 8, 4, 64>(256, "f64");
    run<double, 8, 3, 64>(133, "f64");
    run<double, 16, 8, 32>(256, "f64");
    run<double, 8, 8, 32>(256, "f64");
    run<double, 4, 4, 64>(256, "f64");
    run<double, 8, 8, 64>(512, "f64");
    run<qil_dev::c64, 8, 4, 64>(256, "c64");
    run<qil_dev::c64, 16, 8, 32>(256, "c64");
    run<qil_dev::c64, 8, 8, 32>(256, "c64");
    return 0;
}
