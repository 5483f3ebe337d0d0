// What does a launch cost before it computes anything?  256 workgroups x 512 threads (one per CU) with 0 / 144 KiB of dynamic LDS,
// an empty body, a body that stores `mb` megabytes in 16-byte pieces, and the same with nontemporal stores.
//   hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

extern __shared__ char smem[];

template <int MODE>
__global__ __launch_bounds__(512) void k(float4* out, int per_thread, int touch_lds) {
    if (touch_lds) smem[threadIdx.x] = 1;
    if (MODE == 0) return;
    const size_t base = (size_t)blockIdx.x * blockDim.x * per_thread + threadIdx.x;
    float4 v = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
    for (int i = 0; i < per_thread; ++i) {
        if (MODE == 1) out[base + (size_t)i * blockDim.x] = v;
        else { typedef float f4 __attribute__((ext_vector_type(4))); f4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<f4*>(out + base + (size_t)i * blockDim.x)); }
    }
}

template <int MODE>
static void run(const char* name, int grid, int block, size_t lds, float4* out, int per_thread) {
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(block), lds, 0, out, per_thread, lds ? 1 : 0);
    hipEventRecord(e0, 0);
    const int n = 200;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(block), lds, 0, out, per_thread, lds ? 1 : 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s grid %4d block %3d lds %6zu  stores %6.1f MB : %6.2f us per launch\n", name, grid, block, lds,
           (double)grid * block * per_thread * 16 / 1e6, ms * 1e3 / n);
}

int main() {
    float4* out;
    hipMalloc(&out, (size_t)256 << 20);
    run<0>("empty", 256, 512, 0, out, 0);
    run<0>("empty", 256, 512, 144 * 1024, out, 0);
    run<0>("empty", 256, 256, 64 * 1024, out, 0);
    run<0>("empty", 512, 256, 64 * 1024, out, 0);
    run<0>("empty", 2048, 256, 0, out, 0);
    for (int pt : {1, 2, 4, 8, 16, 32}) {   // 256*512*16 B = 2 MB per unit
        run<1>("store", 256, 512, 144 * 1024, out, pt);
        run<2>("store nontemporal", 256, 512, 144 * 1024, out, pt);
    }
    run<1>("store (no lds)", 256, 512, 0, out, 16);
    run<1>("store 2048 wg", 2048, 256, 0, out, 4);
    return 0;
}
