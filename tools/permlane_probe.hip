// v_permlane32_swap semantics on gfx950: r = __builtin_amdgcn_permlane32_swap(a, b, false, false) with a = lane, b = 1000 + lane.
//   hipcc --offload-arch=gfx950 -O2 tools/permlane_probe.hip -o /tmp/plp && /tmp/plp
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned a = threadIdx.x, b = 1000 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x * 2] = r[0];
    out[threadIdx.x * 2 + 1] = r[1];
}
int main() {
    unsigned* d; unsigned h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 8) { for (int j = l; j < l + 8; ++j) printf("lane %2d: (%4u,%4u)  ", j, h[2 * j], h[2 * j + 1]); printf("\n"); }
    return 0;
}
