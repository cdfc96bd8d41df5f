// Raw issue rate of v_mfma_f64_16x16x4_f64 on gfx950: NACC independent accumulators per wave,
// W waves per SIMD, every CU busy.  Prints TFLOP/s.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_peak mfma_f64_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu) {
    int cus = 256;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    cus = p.multiProcessorCount;
    double* d;
    hipMalloc(&d, sizeof(double) * 256 * cus * blocks_per_cu);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, d, 100, 1.0, 0.5);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, d, iters, 1.0, 0.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flop = 2048.0 * NACC * iters * 4.0 * cus * blocks_per_cu;   // 4 waves per block
    double cyc_per_mfma = (ms * 1e-3 * 2.4e9) / ((double)NACC * iters * blocks_per_cu);
    printf("NACC=%d waves/SIMD=%d: %.2f ms  %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", NACC,
           blocks_per_cu, ms, flop / ms / 1e9, cyc_per_mfma);
    hipFree(d);
}

int main() {
    run<1>(1);
    run<4>(1);
    run<8>(1);
    run<4>(2);
    run<8>(2);
    run<4>(4);
    return 0;
}
