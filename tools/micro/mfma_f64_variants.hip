// Why does a pure v_mfma_f64_16x16x4 stream reach only ~0.6 of the 64-cycle issue rate?  Variants:
//   agpr  : accumulators in AGPRs (what hipcc picks for the builtin)            -- baseline
//   vgpr  : accumulators forced into VGPRs (inline asm, "+v")
//   4x4   : v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4x4x4, 512 flops)
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_variants.bin mfma_f64_variants.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_agpr(double* out, int iters, double a0, double b0) {
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
__global__ __launch_bounds__(256) void k_vgpr(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_4x4(double* out, int iters, double a0, double b0) {
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0;
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
void run(const char* name, K kern, int nacc, double flop_per_mfma, int blocks_per_cu) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    double* d;
    hipMalloc(&d, sizeof(double) * 256 * cus * blocks_per_cu);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(cus * blocks_per_cu), dim3(256), 0, 0, d, 100, 1.0, 0.5);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(cus * blocks_per_cu), dim3(256), 0, 0, d, iters, 1.0, 0.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = flop_per_mfma * nacc * iters * 4.0 * cus * blocks_per_cu;
    printf("{\"variant\": \"%s\", \"nacc\": %d, \"waves_per_simd\": %d, \"ms\": %.2f, \"tflops\": %.1f, \"clock_mhz\": %d}\n", name, nacc,
           blocks_per_cu, ms, flop / ms / 1e9, p.clockRate / 1000);
    hipFree(d);
}

int main() {
    run("agpr", k_agpr<4>, 4, 2048.0, 2);
    run("agpr", k_agpr<8>, 8, 2048.0, 2);
    run("vgpr", k_vgpr<4>, 4, 2048.0, 1);
    run("vgpr", k_vgpr<4>, 4, 2048.0, 2);
    run("vgpr", k_vgpr<8>, 8, 2048.0, 2);
    run("vgpr", k_vgpr<4>, 4, 2048.0, 4);
    run("4x4x4", k_4x4<8>, 8, 512.0, 2);
    run("4x4x4", k_4x4<16>, 16, 512.0, 2);
    run("4x4x4", k_4x4<8>, 8, 512.0, 4);
    return 0;
}
