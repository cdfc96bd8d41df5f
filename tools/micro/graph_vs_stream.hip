// Launch-bound inner loop: K dependent short kernels per "sweep".  Stream launches vs one captured hipGraph
// replayed per sweep (capture + instantiate timed separately).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ __launch_bounds__(256) void step(double* p, int n, int round) {
    // ~ what jacobi_round does per pair: two passes over two columns + a reduction
    __shared__ double red[4];
    double* a = p + (size_t)blockIdx.x * 2 * n;
    double s = 0;
    for (int i = threadIdx.x; i < n; i += 256) s += a[i] * a[i + n];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const double t = (red[0] + red[1] + red[2] + red[3]) * 1e-30 + round * 1e-300;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double x = a[i], y = a[i + n];
        a[i] = x - t * y;
        a[i + n] = y + t * x;
    }
}
int main() {
    const int pairs = 135, n = 540, K = 269, sweeps = 8;
    double* d;
    hipMalloc(&d, sizeof(double) * pairs * 2 * n);
    hipMemset(d, 0, sizeof(double) * pairs * 2 * n);
    hipStream_t st;
    hipStreamCreate(&st);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = now();
        for (int s = 0; s < sweeps; ++s)
            for (int r = 0; r < K; ++r) hipLaunchKernelGGL(step, dim3(pairs), dim3(256), 0, st, d, n, r);
        hipStreamSynchronize(st);
        auto t1 = now();
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int r = 0; r < K; ++r) hipLaunchKernelGGL(step, dim3(pairs), dim3(256), 0, st, d, n, r);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        auto t2 = now();
        for (int s = 0; s < sweeps; ++s) hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        auto t3 = now();
        printf("{\"launches\": %d, \"stream_ms\": %.3f, \"us_per_launch_stream\": %.2f, \"graph_capture_instantiate_ms\": %.3f, "
               "\"graph_replay_ms\": %.3f, \"us_per_node_graph\": %.2f}\n",
               K * sweeps, ms(t0, t1), 1e3 * ms(t0, t1) / (K * sweeps), ms(t1, t2), ms(t2, t3), 1e3 * ms(t2, t3) / (K * sweeps));
        hipGraphExecDestroy(ge);
        hipGraphDestroy(g);
    }
    return 0;
}
