// Where a Gram-matrix block round (gram_block_round, qil_linalg.hip) spends its time: in-kernel shader-clock stamps per phase
// (staging, Gram on the matrix cores, flags, two-sided rotation rounds, update on the matrix cores + store), and the
// event-timed launch train.  Includes the library source so that the product's own kernel is timed:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I include -I qilaplace.jl_amd/csrc \
//         tools/micro/gram_round_cost.hip -o tools/micro/gram_round_cost.bin
#include "../../qilaplace.jl_amd/csrc/qil_linalg.hip"
#include <random>

template <class T, int BB>
static void run(int k, const char* name) {
    const int nblk = ((k + BB - 1) / BB + 1) / 2 * 2;
    const size_t lds = gram_round_lds<T, BB>(k);
    auto kern_ap = &qil_k1<gram_block_round_k<T, BB, true>, gram_round_args<T>>;
    auto kern_x = &qil_k1<gram_block_round_k<T, BB, false>, gram_round_args<T>>;
    long long* prof;
    (void)hipMalloc(&prof, 64);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern_ap), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern_x), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
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
    for (int mode = 0; mode < 2; ++mode) {   // 0: cross rounds (BB inner), 1: all-pairs rounds (2 BB - 1 inner)
        for (auto& v : h) v = nd(rng);
        (void)hipMemcpy(X, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        const int sweeps = 3;
        (void)hipMemset(prof, 0, 64);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        int launches = 0;
        for (int s = 0; s < sweeps; ++s)
            for (int r = 0; r < nblk - 1; ++r, ++launches) {
                gram_round_args<T> a{X, (long long)k, k, k, nblk, r, 1e-15, flag, nullptr, nullptr, prof};
                hipLaunchKernelGGL(mode == 1 ? kern_ap : kern_x, dim3(nblk / 2), dim3(512), lds, 0, a);
            }
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        long long hp[6];
        (void)hipMemcpy(hp, prof, 48, hipMemcpyDeviceToHost);
        const double nl = (double)hp[5];
        const int nin = mode == 1 ? 2 * BB - 1 : BB;
        printf("%s k=%d BB=%d mode=%d: %.2f us per launch (%d WGs, lds %zu); cycles: stage %.0f, gram %.0f, flags %.0f, rotations %.0f "
               "(%.0f per inner round x %d), update+store %.0f\n",
               name, k, BB, mode, 1e3 * ms / launches, nblk / 2, lds, hp[0] / nl, hp[1] / nl, hp[2] / nl, hp[3] / nl, hp[3] / nl / nin,
               nin, hp[4] / nl);
    }
    (void)hipFree(X);
    (void)hipFree(flag);
}

int main() {
    run<double, 16>(256, "f64");
    run<double, 8>(256, "f64");
    run<double, 16>(128, "f64");
    run<double, 16>(512, "f64");
    run<qil_dev::c64, 8>(256, "c64");
    run<qil_dev::c64, 16>(128, "c64");
    return 0;
}
