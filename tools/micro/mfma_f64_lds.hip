// Does feeding v_mfma_f64_16x16x4 (VGPR accumulators) from LDS-loaded operands cost issue rate?
// Per iteration: NR ds_read_b64 (operands), wait, 8 MFMAs on a 4x2 tile grid -- the inner step of the GEMM.
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o mfma_f64_lds.bin mfma_f64_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, int ld) {
    __shared__ double lds[16 * 200];
    for (int i = threadIdx.x; i < 16 * 200; i += 256) lds[i] = 1e-3 * i;
    __syncthreads();
    const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    d4 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = d4{0, 0, 0, 0};
    double are[4] = {1, 2, 3, 4}, bre[2] = {5, 6};
    for (int it = 0; it < iters; ++it) {
        const int kk = (it & 3) * 4;
        if (MODE >= 1) {
#pragma unroll
            for (int t = 0; t < 4; ++t) are[t] = lds[(kk + l4) * ld + 16 * t + l15];
#pragma unroll
            for (int t = 0; t < 2; ++t) bre[t] = lds[(kk + l4) * ld + 64 + 16 * t + l15];
        }
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
                acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(bre[tj], are[ti], acc[ti][tj], 0, 0, 0);
        if (MODE == 2 && (it & 3) == 3) __syncthreads();
    }
    double s = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
void run(const char* name, K kern, int blocks_per_cu, int ld) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    double* d;
    hipMalloc(&d, sizeof(double) * 256 * cus * blocks_per_cu);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(cus * blocks_per_cu), dim3(256), 0, 0, d, 100, ld);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(cus * blocks_per_cu), dim3(256), 0, 0, d, iters, ld);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2048.0 * 8 * iters * 4.0 * cus * blocks_per_cu;
    printf("{\"variant\": \"%s\", \"waves_per_simd\": %d, \"lds_ld\": %d, \"ms\": %.2f, \"tflops\": %.1f}\n", name, blocks_per_cu, ld, ms,
           flop / ms / 1e9);
    hipFree(d);
}

int main() {
    run("registers only", k<0>, 1, 130);
    run("registers only", k<0>, 2, 130);
    run("operands from LDS", k<1>, 1, 130);
    run("operands from LDS", k<1>, 2, 130);
    run("operands from LDS", k<1>, 2, 136);
    run("operands from LDS + barrier every 4 steps", k<2>, 1, 130);
    run("operands from LDS + barrier every 4 steps", k<2>, 2, 130);
    return 0;
}
