// Host round trip of the pattern every data-dependent decision of the truncation chains uses: a short kernel, a small
// device-to-host copy, hipStreamSynchronize, next launch.  Default device scheduling against hipDeviceScheduleSpin.
// Build: hipcc --offload-arch=gfx950 -O3 -o sync_latency.bin sync_latency.hip ; run: ./sync_latency.bin [spin]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>

__global__ void tiny(int* p, int v) { if (threadIdx.x == 0) p[0] = v; }
__global__ void spin_us(long long cycles) {
    const long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) {}
}

int main(int argc, char** argv) {
    const bool spin = argc > 1 && !strcmp(argv[1], "spin");
    if (spin) (void)hipSetDeviceFlags(hipDeviceScheduleSpin);
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    int* d;
    int* h;
    (void)hipMalloc(&d, 256);
    (void)hipHostMalloc(&h, 256);
    for (int body_us : {0, 20, 250}) {
        const int n = 2000;
        for (int warm = 0; warm < 2; ++warm) {
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; ++i) {
                if (body_us) hipLaunchKernelGGL(spin_us, dim3(1), dim3(64), 0, st, (long long)body_us * 100);
                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, d, i);
                (void)hipMemcpyAsync(h, d, 8, hipMemcpyDeviceToHost, st);
                (void)hipStreamSynchronize(st);
            }
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
            if (warm) printf("%s: kernel body %3d us + flag kernel + 8-byte D2H + sync: %.1f us per iteration (overhead %.1f us)\n",
                             spin ? "spin" : "default", body_us, us, us - body_us);
        }
    }
    return 0;
}
