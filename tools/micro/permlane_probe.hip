// What v_permlane32_swap / v_permlane16_swap / row_newbcast do to a wave's lanes on gfx950 (lane ids in, lane ids out).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
    const unsigned l = threadIdx.x;
    const v2u a = __builtin_amdgcn_permlane32_swap(l, l + 100, false, false);
    const v2u b = __builtin_amdgcn_permlane16_swap(l, l + 100, false, false);
    const unsigned d = __builtin_amdgcn_update_dpp(0u, l, 0x150 + 5, 0xf, 0xf, false);
    out[l] = a[0];
    out[64 + l] = a[1];
    out[128 + l] = b[0];
    out[192 + l] = b[1];
    out[256 + l] = d;
}
int main() {
    unsigned* d;
    (void)hipMalloc(&d, 320 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[320];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[5] = {"permlane32_swap(l, l+100)[0]", "permlane32_swap(l, l+100)[1]", "permlane16_swap(l, l+100)[0]", "permlane16_swap(l, l+100)[1]", "dpp row_newbcast:5 (l)"};
    for (int r = 0; r < 5; ++r) {
        printf("%s:\n", names[r]);
        for (int l = 0; l < 64; ++l) printf("%4u%s", h[64 * r + l], (l & 15) == 15 ? "\n" : "");
    }
    return 0;
}
