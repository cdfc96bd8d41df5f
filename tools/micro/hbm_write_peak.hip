// Store-only HBM bandwidth ceiling on gfx950: what a kernel that does NOTHING but write 16 B per lane reaches,
// for plain and non-temporal stores, grid-stride vs one-shot grids, and hipMemsetAsync for reference.
// Build: hipcc --offload-arch=gfx950 -O3 -o hbm_write_peak.bin hbm_write_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));

template <bool NT>
__global__ __launch_bounds__(256) void fill_stride(d2* __restrict__ p, long long n, double v) {
    const d2 val{v, v + 1.0};
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(val, p + i);
        else p[i] = val;
    }
}
// each workgroup owns one contiguous span (like one (row-panel, column) tile of an apply output)
template <bool NT>
__global__ __launch_bounds__(256) void fill_span(d2* __restrict__ p, long long span, double v) {
    const d2 val{v, v + 1.0};
    d2* q = p + blockIdx.x * span;
    for (long long i = threadIdx.x; i < span; i += 256) {
        if (NT) __builtin_nontemporal_store(val, q + i);
        else q[i] = val;
    }
}

int main() {
    const long long bytes = 16LL << 30;
    const long long n = bytes / 16;
    d2* d;
    if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch) {
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("{\"kernel\": \"%s\", \"GB\": %.1f, \"ms\": %.3f, \"TBps\": %.3f}\n", name, bytes / 1e9, ms / 5,
               bytes / 1e12 / (ms / 5 * 1e-3));
    };
    timeit("hipMemsetAsync", [&] { hipMemsetAsync(d, 0, bytes, 0); });
    for (int g : {2048, 8192, 32768, 131072}) {
        char nm[64];
        snprintf(nm, 64, "stride_plain_g%d", g);
        timeit(nm, [&] { hipLaunchKernelGGL(fill_stride<false>, dim3(g), dim3(256), 0, 0, d, n, 1.0); });
        snprintf(nm, 64, "stride_nt_g%d", g);
        timeit(nm, [&] { hipLaunchKernelGGL(fill_stride<true>, dim3(g), dim3(256), 0, 0, d, n, 1.0); });
    }
    for (long long span_kb : {64, 256, 1024}) {
        const long long span = span_kb * 1024 / 16;
        const unsigned g = (unsigned)(n / span);
        char nm[64];
        snprintf(nm, 64, "span%lldKB_plain", span_kb);
        timeit(nm, [&] { hipLaunchKernelGGL(fill_span<false>, dim3(g), dim3(256), 0, 0, d, span, 1.0); });
        snprintf(nm, 64, "span%lldKB_nt", span_kb);
        timeit(nm, [&] { hipLaunchKernelGGL(fill_span<true>, dim3(g), dim3(256), 0, 0, d, span, 1.0); });
    }
    hipFree(d);
    return 0;
}
