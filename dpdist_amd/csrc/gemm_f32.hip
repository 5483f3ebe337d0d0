// fp32 MFMA GEMM for the decoder's dense layers (gfx950 / CDNA4).
//
//   C[M,N] = epi( op(A) * op(B) ),  exact fp32: v_mfma_f32_32x32x2_f32 is bitwise a k-ordered fmaf chain,
//   64 cycles per instruction per SIMD = 64 FLOP/clk/SIMD = 157.3 TFLOP/s chip peak.
//
// Structure (written for 64-wide wavefronts, not a warp tiling):
//   * 256 threads = 4 waves in a 2x2 grid; each wave owns a WM x WN sub-tile built from 32x32 MFMA tiles,
//     so one A register + one B register per lane feed a whole 32x32x2 product;
//   * LDS tiles are K-major ([BK][BM+pad]); an MFMA operand read is 32 consecutive floats per half-wave
//     (conflict free); operands whose global layout is K-contiguous are transposed on the LDS write with an
//     odd leading dimension (conflict-free ds_write_b32), MN-contiguous operands are written as ds_write_b128;
//   * register-prefetch double buffering: the global loads of K-tile t+1 are issued before the MFMAs of tile
//     t and written to the other LDS buffer afterwards -> one barrier per K-tile;
//   * 1-D grid with a bijective XCD remap so each XCD (private 4 MiB L2) works on a contiguous band of tiles;
//   * optional split-K into fp32 slabs + a fused reduce/epilogue kernel (deterministic, no atomics).
//
// Roofline: MFMA-bound.  Algorithmic flops = 2*M*N*K; bytes/flop of a 128x128 tile = 1/32 -> 8 B/clk/CU from L2.
#include "common.h"

namespace dpd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_RELU = 2, EPI_GATE = 3 };

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const float* gate;
    int M, N, K;
    int lda, ldb, ldc;
    int epi;
    int split_k;      // >= 1
    int k_chunk;      // K range per split (multiple of BK)
    long slab_stride; // floats between split-K slabs (0 when split_k == 1)
};

// Loads a [BMN x BK] operand tile into registers and later stores it K-major into LDS.
//   KCONTIG = true : element (mn,k) at p[mn*ld + k]   (transposed on the LDS write)
//   KCONTIG = false: element (mn,k) at p[k*ld + mn]   (direct ds_write_b128)
template <int BMN, int BK, bool KCONTIG>
struct TileStage {
    static constexpr int NV = (BMN * BK / 4) / 256;
    static constexpr int LD = KCONTIG ? BMN + 1 : BMN + 4;
    float4 r[NV];

    __device__ __forceinline__ void load(const float* __restrict__ p, int ld, int mn0, int k0, int MN, int Kend,
                                         int tid) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = tid + v * 256;
            int gmn, gk;
            if (KCONTIG) {
                gmn = mn0 + idx / (BK / 4);
                gk = k0 + (idx % (BK / 4)) * 4;
            } else {
                gk = k0 + idx / (BMN / 4);
                gmn = mn0 + (idx % (BMN / 4)) * 4;
            }
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gmn < MN && gk < Kend) {
                const float* src = KCONTIG ? p + (size_t)gmn * ld + gk : p + (size_t)gk * ld + gmn;
                x = *reinterpret_cast<const float4*>(src);
            }
            r[v] = x;
        }
    }

    __device__ __forceinline__ void store(float* __restrict__ s, int tid) const {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = tid + v * 256;
            if (KCONTIG) {
                const int row = idx / (BK / 4), kq = (idx % (BK / 4)) * 4;
                s[(kq + 0) * LD + row] = r[v].x;
                s[(kq + 1) * LD + row] = r[v].y;
                s[(kq + 2) * LD + row] = r[v].z;
                s[(kq + 3) * LD + row] = r[v].w;
            } else {
                const int krow = idx / (BMN / 4), q4 = (idx % (BMN / 4)) * 4;
                *reinterpret_cast<float4*>(&s[krow * LD + q4]) = r[v];
            }
        }
    }
};

// ABL: ablation switches for tools/gemm_bench.py (results are WRONG when != 0): bit0 = no global loads after the
// first K-tile, bit1 = no LDS refill + no barrier in the loop.
template <int BM, int BN, int BK, bool AK, bool BKC, int ABL = 0>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    constexpr int WM = BM / 2, WN = BN / 2;   // 2x2 wave grid
    constexpr int TM = WM / 32, TN = WN / 32; // 32x32 MFMA tiles per wave
    using StA = TileStage<BM, BK, AK>;
    using StB = TileStage<BN, BK, BKC>;
    constexpr int LDA = StA::LD, LDB = StB::LD;
    constexpr int A_BUF = BK * LDA, B_BUF = BK * LDB;
    static_assert((2 * A_BUF) % 4 == 0, "B tile must stay 16-byte aligned");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * A_BUF;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;

    const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
    const int per_z = tilesM * tilesN;
    const int sid = xcd_remap(blockIdx.x, per_z * g.split_k);
    const int z = sid / per_z, t = sid % per_z;
    const int m0 = (t / tilesN) * BM, n0 = (t % tilesN) * BN;
    const int kbeg = z * g.k_chunk;
    const int kend = min(g.K, kbeg + g.k_chunk);
    const int nt = (kend - kbeg + BK - 1) / BK;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    StA sa;
    StB sb;
    sa.load(g.A, g.lda, m0, kbeg, g.M, kend, tid);
    sb.load(g.B, g.ldb, n0, kbeg, g.N, kend, tid);
    sa.store(As, tid);
    sb.store(Bs, tid);
    __syncthreads();

    for (int it = 0; it < nt; ++it) {
        const int cur = it & 1;
        if (it + 1 < nt && !(ABL & 1)) {  // prefetch the next K-tile into registers while this one is multiplied
            sa.load(g.A, g.lda, m0, kbeg + (it + 1) * BK, g.M, kend, tid);
            sb.load(g.B, g.ldb, n0, kbeg + (it + 1) * BK, g.N, kend, tid);
        }
        const float* as = As + cur * A_BUF + half * LDA + wm0 + l31;
        const float* bs = Bs + cur * B_BUF + half * LDB + wn0 + l31;
        // MFMA operand fragments are double-buffered in registers: the ds_reads of k-step kk+2 are issued before
        // the MFMAs of step kk, so the 64-cycle MFMAs never wait on LDS latency (one wave per SIMD has no other
        // wave to hide it).  The LDS write of the prefetched K-tile is issued half way through the step so that it
        // overlaps the remaining MFMAs instead of trailing them.
        float a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = as[i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = bs[j * 32];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int c = (kk >> 1) & 1, n = c ^ 1;
            if (kk + 2 < BK) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[n][i] = as[(kk + 2) * LDA + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[n][j] = bs[(kk + 2) * LDB + j * 32];
            }
            if (kk == BK / 2 && it + 1 < nt && !(ABL & 2)) {
                sa.store(As + (cur ^ 1) * A_BUF, tid);
                sb.store(Bs + (cur ^ 1) * B_BUF, tid);
            }
            // pin the order [ds_reads of step kk+2 | MFMAs of step kk]: left alone, the machine scheduler sinks the
            // reads back to just before their use and every step stalls on LDS latency
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(ABL & 2)) __syncthreads();
    }

    // epilogue: acc[i][j][r] is C[row = (r&3) + 8*(r>>2) + 4*half][col = l31] of the 32x32 tile
    float* Cz = g.C + (size_t)z * g.slab_stride;
    const int epi = g.epi;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn0 + j * 32 + l31;
        if (col >= g.N) continue;
        const float bv = (epi == EPI_BIAS || epi == EPI_BIAS_RELU) ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row >= g.M) continue;
                float v = acc[i][j][r] + bv;
                if (epi == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
                if (epi == EPI_GATE) v = (g.gate[(size_t)row * g.ldc + col] > 0.f) ? v : 0.f;
                Cz[(size_t)row * g.ldc + col] = v;
            }
        }
    }
}

// out[m,n] = epi( sum_z slab[z][m,n] ); N % 4 == 0, slabs are dense [M,N].
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int split_k, long slab_stride,
                                                           int M, int N, float* __restrict__ C, int ldc,
                                                           const float* __restrict__ bias, const float* __restrict__ gate,
                                                           int epi) {
    const long total4 = (long)M * N / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long e = i * 4;
        const int row = (int)(e / N), col = (int)(e % N);
        float4 s = *reinterpret_cast<const float4*>(slabs + e);
        for (int z = 1; z < split_k; ++z) {
            const float4 x = *reinterpret_cast<const float4*>(slabs + (size_t)z * slab_stride + e);
            s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
        }
        if (epi == EPI_BIAS || epi == EPI_BIAS_RELU) {
            const float4 b = *reinterpret_cast<const float4*>(bias + col);
            s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
        }
        if (epi == EPI_BIAS_RELU) {
            s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f);
        }
        if (epi == EPI_GATE) {
            const float4 gt = *reinterpret_cast<const float4*>(gate + (size_t)row * ldc + col);
            s.x = gt.x > 0.f ? s.x : 0.f; s.y = gt.y > 0.f ? s.y : 0.f;
            s.z = gt.z > 0.f ? s.z : 0.f; s.w = gt.w > 0.f ? s.w : 0.f;
        }
        *reinterpret_cast<float4*>(C + (size_t)row * ldc + col) = s;
    }
}

// Opt-in in-stream profiler (bench.py's roofline leg): a hipEvent pair around every GEMM kernel launch, on the
// stream the kernel is launched on.  Off by default; costs two event records per launch when on.
struct GemmProf {
    bool on = false;
    int n = 0;
    static constexpr int kMax = 8192;
    hipEvent_t ev[2 * kMax];
    double flops[kMax];
    bool have_events = false;
};
static GemmProf g_prof;

template <int BM, int BN, int BK, bool AK, bool BKC, int ABL = 0>
static int launch_cfg(const GemmArgs& g, hipStream_t s) {
    using StA = TileStage<BM, BK, AK>;
    using StB = TileStage<BN, BK, BKC>;
    constexpr size_t lds = (size_t)(2 * BK * StA::LD + 2 * BK * StB::LD) * sizeof(float);
    auto kern = gemm_f32_kernel<BM, BN, BK, AK, BKC, ABL>;
    if (lds > 64 * 1024) {
        static bool done = false;  // benign race: idempotent attribute
        if (!done) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            done = true;
        }
    }
    const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
    const int nblk = tilesM * tilesN * g.split_k;
    const bool prof = g_prof.on && g_prof.n < GemmProf::kMax;
    if (prof) (void)hipEventRecord(g_prof.ev[2 * g_prof.n], s);
    DPD_LAUNCH(kern, dim3(nblk), dim3(256), lds, s, g);
    if (prof) {
        (void)hipEventRecord(g_prof.ev[2 * g_prof.n + 1], s);
        g_prof.flops[g_prof.n] = 2.0 * g.M * g.N * g.K;
        ++g_prof.n;
    }
    return (int)hipGetLastError();
}

template <bool AK, bool BKC>
static int launch_tile(int tile, const GemmArgs& g, hipStream_t s) {
    switch (tile) {
        case 1: return launch_cfg<128, 128, 32, AK, BKC>(g, s);
        case 2: return launch_cfg<128, 64, 32, AK, BKC>(g, s);
        case 3: return launch_cfg<64, 64, 32, AK, BKC>(g, s);
        case 11: return launch_cfg<128, 128, 32, AK, BKC, 1>(g, s);   // ablations (wrong results, timing only)
        case 21: return launch_cfg<128, 128, 32, AK, BKC, 2>(g, s);
        case 31: return launch_cfg<128, 128, 32, AK, BKC, 3>(g, s);
        case 13: return launch_cfg<64, 64, 32, AK, BKC, 1>(g, s);
        case 23: return launch_cfg<64, 64, 32, AK, BKC, 2>(g, s);
        case 33: return launch_cfg<64, 64, 32, AK, BKC, 3>(g, s);
        default: return DPD_E_UNSUPPORTED;
    }
}

// efficiency of covering an M x N output with BMxBN tiles on 256 CUs (whole "rounds" of blocks)
static double tile_eff(int M, int N, int BM, int BN, int split) {
    const long nblk = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN) * split;
    const long rounds = (nblk + 255) / 256;
    return (double)M * N * split / ((double)rounds * 256.0 * BM * BN);
}

int gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
             int ldc, const float* bias, const float* gate, int epilogue, int split_k, int tile, void* ws,
             size_t ws_bytes, hipStream_t s) {
    if (!A || !B || !C) return DPD_E_NULL;
    if (M <= 0 || N <= 0 || K <= 0 || split_k < 1) return DPD_E_DIM;
    if ((K & 3) || (N & 3) || (lda & 3) || (ldb & 3) || (ldc & 3)) return DPD_E_UNSUPPORTED;
    if (transA && (M & 3)) return DPD_E_UNSUPPORTED;
    if (transA && transB) return DPD_E_UNSUPPORTED;
    if ((epilogue == EPI_BIAS || epilogue == EPI_BIAS_RELU) && !bias) return DPD_E_NULL;
    if (epilogue == EPI_GATE && !gate) return DPD_E_NULL;
    if (epilogue < 0 || epilogue > 3) return DPD_E_UNSUPPORTED;

    if (tile == 0) {
        // Measured on MI355X (tools/gemm_bench.py, profiles/): the 64x64 tile (4 resident blocks = 16 waves per CU)
        // beats 128x64 and 128x128 at every decoder shape (112 vs 100 vs 90 TFLOP/s on layer 1) because the other
        // blocks' MFMAs cover each block's barrier / global-load latency; MFMA is so slow in fp32 that the extra L2
        // traffic of the small tile (16 flop/B) is irrelevant.  Larger tiles only when the grid would not fill.
        (void)tile_eff;
        tile = 3;
    }
    GemmArgs g{};
    g.A = A; g.B = B; g.bias = bias; g.gate = gate;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb;
    g.split_k = split_k;
    if (split_k > 1) {
        const int chunk = (((K + split_k - 1) / split_k) + 31) / 32 * 32;
        if ((size_t)split_k * M * N * sizeof(float) > ws_bytes || !ws) return DPD_E_WORKSPACE;
        g.k_chunk = chunk; g.C = (float*)ws; g.ldc = N; g.slab_stride = (long)M * N; g.epi = EPI_NONE;
    } else {
        g.k_chunk = (K + 31) / 32 * 32; g.C = C; g.ldc = ldc; g.slab_stride = 0; g.epi = epilogue;
    }
    int rc;
    if (!transA && !transB) rc = launch_tile<true, false>(tile, g, s);       // NN: A[M,K], B[K,N]
    else if (!transA && transB) rc = launch_tile<true, true>(tile, g, s);    // NT: A[M,K], B[N,K]
    else rc = launch_tile<false, false>(tile, g, s);                          // TN: A[K,M], B[K,N]
    if (rc) return rc;
    if (split_k > 1) {
        const long total4 = (long)M * N / 4;
        const int blocks = (int)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048);
        DPD_LAUNCH(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)ws, split_k, (long)M * N, M,
                           N, C, ldc, bias, gate, epilogue);
        return (int)hipGetLastError();
    }
    return 0;
}

}  // namespace dpd

// ---- profiler C ABI -------------------------------------------------------------------------------------
extern "C" int dpd_prof_enable(int on) {
    using dpd::g_prof;
    if (on && !g_prof.have_events) {
        for (int i = 0; i < 2 * dpd::GemmProf::kMax; ++i) DPD_HIP(hipEventCreate(&g_prof.ev[i]));
        g_prof.have_events = true;
    }
    g_prof.on = on != 0;
    if (on) g_prof.n = 0;
    return 0;
}

// Synchronises with the recorded events; returns the number of GEMM launches seen since dpd_prof_enable(1) and
// fills total milliseconds / total (padded-shape) flops 2*M*N*K of those launches.
extern "C" int dpd_prof_collect(double* total_ms, double* total_flops) {
    using dpd::g_prof;
    double ms = 0.0, fl = 0.0;
    for (int i = 0; i < g_prof.n; ++i) {
        if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) return -1;
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return -1;
        ms += t;
        fl += g_prof.flops[i];
    }
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    return g_prof.n;
}

extern "C" int dpd_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B,
                            int ldb, float* Cout, int ldc, const float* bias, const float* gate, int epilogue,
                            int split_k, int tile, void* ws, size_t ws_bytes, void* stream) {
    return dpd::gemm_f32(transA, transB, M, N, K, A, lda, B, ldb, Cout, ldc, bias, gate, epilogue, split_k, tile, ws,
                         ws_bytes, (hipStream_t)stream);
}
