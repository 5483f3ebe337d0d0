// Cost of one s_barrier per 16 MFMAs in a 16-wave (1024-thread) workgroup that owns the whole CU, vs 4-wave groups.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_barrier.hip -o /tmp/mfma_barrier && /tmp/mfma_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int THREADS, int PER_BARRIER, bool BAR>
__global__ __launch_bounds__(THREADS) void k(float* out, int iters) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += THREADS) lds[i] = (float)(i & 7);
    __syncthreads();
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* p = lds + (threadIdx.x & 63);
    float a[2][4], b[2][4];
    for (int u = 0; u < 4; ++u) { a[0][u] = p[u * 64]; b[0][u] = p[u * 64 + 32]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < PER_BARRIER / 4; ++g) {
            const int h = g & 1;
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[h ^ 1][u] = p[((it + g) & 7) * 256 + u * 64]; b[h ^ 1][u] = p[((it + g) & 7) * 256 + u * 64 + 40]; }
            if (BAR && g == (PER_BARRIER / 8)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[h][u], b[h][u], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 123.456f) out[0] = s;
}
template <int THREADS, int PER_BARRIER, bool BAR>
void run(int blocks_per_cu, int iters) {
    float* d; (void)hipMalloc(&d, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * blocks_per_cu;
    k<THREADS, PER_BARRIER, BAR><<<blocks, THREADS>>>(d, iters); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<THREADS, PER_BARRIER, BAR><<<blocks, THREADS>>>(d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double fl = 2.0 * 32 * 32 * 2 * (double)PER_BARRIER * iters * (THREADS / 64) * blocks;
    printf("threads=%4d blocks/CU=%d mfma/barrier=%2d barrier=%d : %.3f ms %.1f TFLOP/s\n", THREADS, blocks_per_cu, PER_BARRIER, (int)BAR, ms, fl / ms / 1e9);
}
int main() {
    run<1024, 16, false>(1, 1000); run<1024, 16, true>(1, 1000); run<1024, 32, true>(1, 500); run<1024, 8, true>(1, 2000);
    run<512, 16, false>(2, 1000);  run<512, 16, true>(2, 1000);
    run<256, 16, false>(4, 1000);  run<256, 16, true>(4, 1000);
    run<256, 16, true>(2, 1000);   run<512, 16, true>(1, 1000);
    return 0;
}
