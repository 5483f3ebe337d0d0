// How fast does the fp32 MFMA pipe run when every MFMA's operands come from a fresh ds_read (as in a GEMM loop)?
// hipcc --offload-arch=gfx950 -O3 tools/mfma_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// NACC independent accumulators per wave, GROUP MFMAs issued per LDS-read batch (operands read one batch ahead)
template <int NACC, int GROUP, bool PIN>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)(i & 7);
    __syncthreads();
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float* p = lds + (threadIdx.x & 63);
    float a[2][GROUP], b[2][GROUP];
    for (int u = 0; u < GROUP; ++u) { a[0][u] = p[u * 64]; b[0][u] = p[u * 64 + 32]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int u = 0; u < GROUP; ++u) { a[h ^ 1][u] = p[(it & 7) * 256 + u * 64 + h * 8]; b[h ^ 1][u] = p[(it & 7) * 256 + u * 64 + 40 + h * 8]; }
            if (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < GROUP; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[h][u], b[h][u], acc[u % NACC], 0, 0, 0);
            if (PIN) __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
}
template <int NACC, int GROUP, bool PIN>
void run(int blocks_per_cu, int iters) {
    float* d; (void)hipMalloc(&d, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * blocks_per_cu;
    k<NACC, GROUP, PIN><<<blocks, 256>>>(d, iters); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<NACC, GROUP, PIN><<<blocks, 256>>>(d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double fl = 2.0 * 32 * 32 * 2 * 2.0 * GROUP * iters * 4.0 * blocks;
    printf("nacc=%d group=%d pin=%d blocks/CU=%d : %.3f ms %.1f TFLOP/s\n", NACC, GROUP, (int)PIN, blocks_per_cu, ms, fl / ms / 1e9);
}
int main() {
    for (int bpc = 1; bpc <= 4; ++bpc) {
        run<1, 4, true>(bpc, 4000 / bpc);
        run<1, 1, true>(bpc, 16000 / bpc);
        run<2, 4, true>(bpc, 4000 / bpc);
        run<4, 4, true>(bpc, 4000 / bpc);
        run<1, 4, false>(bpc, 4000 / bpc);
    }
    return 0;
}
