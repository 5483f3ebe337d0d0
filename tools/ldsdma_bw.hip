// L2/MALL -> LDS bandwidth of the LDS-DMA stream (global_load_lds_dwordx4) as a GEMM operand ring uses it: what does the
// piece SHAPE (bytes that are contiguous in global memory per 1-KiB piece), the ring depth, the number of waves per CU and the
// source footprint do to the sustained bytes per clock per CU?  (The bf16 GEMM of gemm_x3.hip is bound by this stream.)
//   hipcc --offload-arch=gfx950 -O3 tools/ldsdma_bw.hip -o /tmp/ldsdma_bw && /tmp/ldsdma_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void dma_piece(const void* src, unsigned dst_bytes) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(dst_bytes)
        : "memory");
}

// One workgroup streams `ktiles` tiles of TILE_KB KiB through an NS-stage LDS ring.  A tile is TILE_KB pieces of 1 KiB; a piece
// covers 1024 / ROWB "rows" of ROWB contiguous bytes, consecutive rows `row_stride` bytes apart (ROWB = 1024: fully linear).
// Per tile: every wave issues its share of the pieces of tile t + NS - 1, waits (counted vmcnt) for tile t, one barrier.
template <int WAVES, int NS, int TILE_KB, int ROWB, bool BARRIER, int NREAD = 0, int NMFMA = 0>
__global__ __launch_bounds__(WAVES * 64) void k(const char* __restrict__ src, size_t slab_bytes, int slabs, size_t row_stride,
                                                size_t tile_stride, int ktiles, float* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int PPW = TILE_KB / WAVES;          // pieces per wave per tile
    static_assert(TILE_KB % WAVES == 0, "pieces must divide over the waves");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (size_t)(blockIdx.x % slabs) * slab_bytes;
    constexpr int LPR = ROWB / 16;                // lanes per row
    // piece p of a tile: rows [p * 1024/ROWB, ...), lane -> (row, 16-byte column)
    size_t lane_off[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int p = wave * PPW + i;
        const int row = p * (1024 / ROWB) + lane / LPR;
        lane_off[i] = (size_t)row * row_stride + (size_t)(lane % LPR) * 16;
    }
    auto issue = [&](int t) {
        const int st = t % NS;
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            dma_piece(base + (size_t)t * tile_stride + lane_off[i], (unsigned)(st * TILE_KB * 1024 + (wave * PPW + i) * 1024));
    };
    for (int t = 0; t < NS - 1 && t < ktiles; ++t) issue(t);
    float acc = 0.f;
    f32x16 macc;
    for (int r = 0; r < 16; ++r) macc[r] = 0.f;
    for (int t = 0; t < ktiles; ++t) {
        if (t + NS - 1 < ktiles) issue(t + NS - 1);
        // tile t complete when at most (NS-1)*PPW younger pieces of this wave are outstanding
        if (t + NS - 1 < ktiles) {
            if (PPW * (NS - 1) == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if (PPW * (NS - 1) == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (PPW * (NS - 1) == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (PPW * (NS - 1) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (PPW * (NS - 1) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (PPW * (NS - 1) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (PPW * (NS - 1) == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else if (PPW * (NS - 1) == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (PPW * (NS - 1) == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (PPW * (NS - 1) == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (BARRIER) __builtin_amdgcn_s_barrier();
        // touch the stage (one ds_read per lane) so the ring is really consumed
        acc += *reinterpret_cast<const float*>(lds + (t % NS) * TILE_KB * 1024 + threadIdx.x * 4 % (TILE_KB * 1024));
        // optional consumer work of a GEMM wave: NREAD ds_read_b128 fragment reads (conflict-free: lane-linear) + NMFMA bf16 MFMAs
        if (NREAD > 0) {
            bf16x8 fr[NREAD > 0 ? NREAD : 1];
#pragma unroll
            for (int i = 0; i < NREAD; ++i)
                fr[i] = *reinterpret_cast<const bf16x8*>(lds + (t % NS) * TILE_KB * 1024 + ((wave * NREAD + i) * 1024 + lane * 16) % (TILE_KB * 1024));
#pragma unroll
            for (int i = 0; i < NMFMA; ++i) macc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % NREAD], fr[(i + 1) % NREAD], macc, 0, 0, 0);
            if (NMFMA == 0)
#pragma unroll
                for (int i = 0; i < NREAD; ++i) acc += (float)fr[i][0];
        }
        if (BARRIER) __builtin_amdgcn_s_barrier();
    }
    for (int r = 0; r < 16; ++r) acc += macc[r];
    if (acc == 123.456f) out[0] = acc;
}

static double clk_ghz = 2.4;

template <int WAVES, int NS, int TILE_KB, int ROWB, bool BARRIER, int NREAD = 0, int NMFMA = 0>
void run(const char* name, const char* src, size_t src_bytes, int wg_per_cu, int slabs, size_t row_stride, int ktiles, float* out) {
    // a slab = the region one workgroup streams: ktiles tiles; tile t starts at t * tile_stride
    // ROWB < 1024: operand [rows][K] with row_stride bytes per row, a tile = (TILE_KB*1024/ROWB) rows x ROWB bytes, next tile = +ROWB bytes
    // ROWB = 1024: linear, next tile = + TILE_KB KiB
    const size_t rows = (size_t)TILE_KB * 1024 / ROWB;
    size_t tile_stride, slab_bytes;
    if (ROWB == 1024) { tile_stride = (size_t)TILE_KB * 1024; row_stride = 1024; slab_bytes = tile_stride * ktiles; }
    else { tile_stride = ROWB; slab_bytes = rows * row_stride; }
    if ((size_t)slabs * slab_bytes > src_bytes) { printf("%-44s skipped (needs %zu MB)\n", name, (size_t)slabs * slab_bytes >> 20); return; }
    const int blocks = 256 * wg_per_cu;
    const size_t lds = (size_t)NS * TILE_KB * 1024;
    auto kern = k<WAVES, NS, TILE_KB, ROWB, BARRIER, NREAD, NMFMA>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) kern<<<blocks, WAVES * 64, lds>>>(src, slab_bytes, slabs, row_stride, tile_stride, ktiles, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) kern<<<blocks, WAVES * 64, lds>>>(src, slab_bytes, slabs, row_stride, tile_stride, ktiles, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    hipError_t e = hipGetLastError();
    const double bytes = (double)blocks * ktiles * TILE_KB * 1024;
    printf("%-44s wg/cu %d slabs %4d foot %6.1f MB : %7.1f us  %6.2f TB/s  %5.1f B/clk/CU  (%.2f us per tile)%s\n", name, wg_per_cu, slabs,
           slabs * slab_bytes / 1e6, ms * 1e3, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / (clk_ghz * 1e9), ms * 1e3 / ktiles,
           e == hipSuccess ? "" : hipGetErrorString(e));
}

int main() {
    const size_t src_bytes = 1ull << 30;
    char* src; (void)hipMalloc(&src, src_bytes); (void)hipMemset(src, 1, src_bytes);
    float* out; (void)hipMalloc(&out, 4);
    const size_t RS = 5056;     // row stride of a [rows][2528] bf16 operand
    // --- 1. piece shape, 8 waves, 16 KiB tiles (128x128x32 bf16), 4 stages, every workgroup its own slab (MALL/HBM) vs shared (L2)
    printf("== piece shape (8 waves, 16 KiB tile, 4 stages, 1 WG/CU)\n");
    for (int slabs : {256, 32, 8}) {
        run<8, 4, 16, 64, true>("rows of 64 B (RC, BK=32)", src, src_bytes, 1, slabs, RS, 79, out);
        run<8, 4, 16, 128, true>("rows of 128 B (RC, BK=64)", src, src_bytes, 1, slabs, RS, 39, out);
        run<8, 4, 16, 256, true>("rows of 256 B", src, src_bytes, 1, slabs, RS, 19, out);
        run<8, 4, 16, 1024, true>("linear 1 KiB pieces", src, src_bytes, 1, slabs, RS, 79, out);
    }
    printf("== ring depth (linear, 8 waves, 16 KiB tile)\n");
    run<8, 2, 16, 1024, true>("2 stages", src, src_bytes, 1, 256, RS, 79, out);
    run<8, 3, 16, 1024, true>("3 stages", src, src_bytes, 1, 256, RS, 79, out);
    run<8, 4, 16, 1024, true>("4 stages", src, src_bytes, 1, 256, RS, 79, out);
    run<8, 7, 16, 1024, true>("7 stages", src, src_bytes, 1, 256, RS, 79, out);
    run<8, 4, 32, 1024, true>("4 stages x 32 KiB", src, src_bytes, 1, 256, RS, 40, out);
    printf("== no barrier (linear, 8 waves, 16 KiB tile, 4 stages)\n");
    run<8, 4, 16, 1024, false>("no barrier", src, src_bytes, 1, 256, RS, 79, out);
    run<8, 4, 16, 64, false>("no barrier, rows of 64 B", src, src_bytes, 1, 256, RS, 79, out);
    printf("== waves per CU (linear, 4 stages)\n");
    run<4, 4, 16, 1024, true>("4 waves, 16 KiB", src, src_bytes, 1, 256, RS, 79, out);
    run<16, 4, 16, 1024, true>("16 waves, 16 KiB", src, src_bytes, 1, 256, RS, 79, out);
    run<8, 4, 16, 1024, true>("8 waves x 2 WG/CU, 16 KiB", src, src_bytes, 2, 512, RS, 79, out);
    run<4, 4, 8, 1024, true>("4 waves x 4 WG/CU, 8 KiB", src, src_bytes, 4, 1024, RS, 79, out);
    run<8, 4, 16, 64, true>("8 waves x 2 WG/CU, rows of 64 B", src, src_bytes, 2, 512, RS, 79, out);
    printf("== L2-resident source (8 slabs shared by all workgroups)\n");
    run<8, 4, 16, 1024, true>("linear, shared", src, src_bytes, 1, 8, RS, 79, out);
    run<8, 4, 16, 1024, true>("linear, shared, 2 WG/CU", src, src_bytes, 2, 8, RS, 79, out);
    run<16, 4, 32, 1024, true>("16 waves, 32 KiB, shared", src, src_bytes, 1, 8, RS, 40, out);
    printf("== with the consumer work of a GEMM wave (linear, 8 waves, 16 KiB, 4 stages, 41 MB)\n");
    run<8, 4, 16, 1024, true, 0, 0>("DMA only", src, src_bytes, 1, 32, RS, 79, out);
    run<8, 4, 16, 1024, true, 6, 0>("+ 6 ds_read_b128 per wave", src, src_bytes, 1, 32, RS, 79, out);
    run<8, 4, 16, 1024, true, 6, 4>("+ 6 ds_read_b128 + 4 MFMA", src, src_bytes, 1, 32, RS, 79, out);
    run<8, 4, 16, 1024, true, 12, 0>("+ 12 ds_read_b128", src, src_bytes, 1, 32, RS, 79, out);
    run<8, 4, 16, 1024, true, 6, 4>("+ 6 reads + 4 MFMA, 2 WG/CU", src, src_bytes, 2, 32, RS, 79, out);
    run<8, 4, 16, 64, true, 6, 4>("rows of 64 B + 6 reads + 4 MFMA", src, src_bytes, 1, 32, RS, 79, out);
    run<8, 4, 16, 64, true, 6, 4>("rows of 64 B + 6 reads + 4 MFMA, 2 WG/CU", src, src_bytes, 2, 32, RS, 79, out);
    run<4, 4, 16, 1024, true, 8, 8>("4 waves: 8 reads + 8 MFMA", src, src_bytes, 1, 32, RS, 79, out);
    run<8, 4, 16, 1024, false, 6, 4>("no barrier + 6 reads + 4 MFMA", src, src_bytes, 1, 32, RS, 79, out);
    return 0;
}
