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
#include <mutex>

#include "gemm_shared.h"
#include "gemm_rs.h"

namespace dpd {

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

template <int BM, int BN, int BK, bool AK, bool BKC>
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
        if (it + 1 < nt) {  // prefetch the next K-tile into registers while this one is multiplied
            const int kn = kbeg + (it + 1) * BK;
            sa.load(g.A, g.lda, m0, kn, g.M, kend, tid);
            sb.load(g.B, g.ldb, n0, kn, g.N, kend, tid);
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
            if (kk == BK / 2 && it + 1 < nt) {
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
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) store_tile(g, acc[i][j], z, m0 + wm0 + i * 32, n0 + wn0 + j * 32 + l31, half);
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

int splitk_reduce(const float* slabs, int split_k, long slab_stride, int M, int N, float* C, int ldc, const float* bias,
                  const float* gate, int epi, hipStream_t s) {
    const long total4 = (long)M * N / 4;
    const int blocks = (int)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048);
    DPD_LAUNCH(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, slabs, split_k, slab_stride, M, N, C, ldc, bias, gate, epi);
    return (int)hipGetLastError();
}

// =========================================================================================================
// LDS-DMA kernel: (32*WR) x (32*WC) x 32 tiles, one 32x32 MFMA tile per wave, 4-stage LDS ring fed by
// global_load_lds_dwordx4 (no VGPR staging, no ds_write).
//
//   * every wave issues PPW = (BM+BN)/8/(WR*WC) 1-KiB DMA pieces per K-tile, THREE K-tiles ahead of the MFMAs;
//     a K-tile has two full tile-times (~3 us) to land before it is needed;
//   * ONE raw s_barrier per K-tile, placed in the MIDDLE of the tile (between k-blocks 1 and 2): it certifies that
//     K-tile t+1 has landed for every wave and that everybody is past K-tile t-1 (whose stage is then refilled).
//     The fragments of k-block 2 are read before the barrier and the first fragments of K-tile t+1 are read during
//     k-block 3, so the MFMA stream never waits at a tile boundary even when all waves of a CU run in lock step;
//   * the DMA writes LDS lane-linearly, so layouts are chosen on the SOURCE address:
//       K-contiguous operand  -> image [BMN rows][8 x 16-B slots], slot XOR-swizzled with (row>>1)&7; one
//                                ds_read_b128 per lane then carries the operand of FOUR k-steps: the k order inside
//                                an 8-deep block is permuted (half-wave h takes k = 8kb+4h+s at step s), which is
//                                legal because A and B use the same permutation of the contraction index;
//       MN-contiguous operand -> image [32 k][BMN] dense, ds_read_b32 per k-step (conflict free);
//   * out-of-range rows/columns are CLAMPED on the source address (garbage rows/cols are never stored), so there is
//     no predication in the loop; K must be a multiple of 32 (the decoder pads 2503 -> 2528 for this);
//   * WR x WC = 4x4 (1024 threads, 128x128, 32 flop per L2 byte, one block per CU with 4 waves per SIMD),
//     4x2 (128x64) and 2x2 (64x64) cover the smaller grids.
// =========================================================================================================
// MFMA operand fragments of one 8-deep k-block kb for the 32 rows/cols starting at mn0w (wave offset in the tile)
template <bool KCONTIG, int BMN>
__device__ __forceinline__ void load_frag(const float* img, int mn0w, int l31, int half, int kb, float (&f)[4]) {
    if (KCONTIG) {
        const int row = mn0w + l31;
        const int slot = (2 * kb + half) ^ ((row >> 1) & 7);
        const float4 v = *reinterpret_cast<const float4*>(img + row * 32 + slot * 4);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) f[s] = img[(8 * kb + 4 * half + s) * BMN + mn0w + l31];
    }
}

template <int WR, int WC, int NS, bool AK, bool BKC, int TM = 1, int TN = 1>
__global__ __launch_bounds__(64 * WR * WC) void gemm_dma_kernel(GemmArgs g) {
    constexpr bool PRIO = WR * WC >= 16;
    constexpr int BM = 32 * WR * TM, BN = 32 * WC * TN, BK = 32, NW = WR * WC;   // wave tile (32*TM) x (32*TN)
    static_assert(NS >= 3 && NS <= 5, "ring depth");
    constexpr int A_IMG = BM * BK, B_IMG = BN * BK, STAGE = A_IMG + B_IMG;   // floats
    constexpr int PA = BM / 8, PB = BN / 8;                                  // 1-KiB pieces per K-tile
    constexpr int PPW = (PA + PB) / NW;                                      // pieces per wave per K-tile
    static_assert((PA + PB) % NW == 0, "piece split");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave / WC) * 32 * TM, wn0 = (wave % WC) * 32 * TN;

    const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
    const int per_z = tilesM * tilesN;
    const int ngrp = g.A2 ? 2 : 1;
    const int sid = xcd_remap(blockIdx.x, per_z * g.split_k * ngrp);
    const int grp = sid / (per_z * g.split_k);            // grouped launch: second problem of identical shape
    const int sid1 = sid % (per_z * g.split_k);
    const int z = sid1 / per_z, t = sid1 % per_z;
    const int m0 = (t / tilesN) * BM, n0 = (t % tilesN) * BN;
    const int kbeg = z * g.k_chunk;
    const int kend = min(g.K, kbeg + g.k_chunk);
    const int nt = max(0, (kend - kbeg) / BK);
    const float* gA = grp ? g.A2 : g.A;
    const float* gB = grp ? g.B2 : g.B;

    // this wave's DMA pieces: piece p = wave + j*NW; p < PA -> A piece p, else B piece p-PA
    const float* src[PPW];
    size_t step[PPW];
    unsigned dst[PPW];
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int p = wave + j * NW;
        const bool isA = p < PA;
        const int c = isA ? p : p - PA;
        const bool kc = isA ? AK : BKC;
        const float* base = isA ? gA : gB;
        const int ld = isA ? g.lda : g.ldb;
        const int mn0 = isA ? m0 : n0;
        const int MN = isA ? g.M : g.N;
        const int BMN = isA ? BM : BN;
        dst[j] = lds_base + (unsigned)((isA ? 0 : A_IMG) + c * 256) * 4u;
        if (kc) {
            const int row = c * 8 + (lane >> 3), slot = lane & 7;
            src[j] = base + (size_t)min(mn0 + row, MN - 1) * ld + kbeg + 4 * (slot ^ ((row >> 1) & 7));
            step[j] = 32;
        } else {
            const int lin = c * 256 + lane * 4;
            src[j] = base + (size_t)(kbeg + lin / BMN) * ld + min(mn0 + lin % BMN, MN - 4);
            step[j] = (size_t)32 * ld;
        }
    }
    auto issue = [&](int stage) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            dma_piece(src[j], dst[j] + (unsigned)(stage * STAGE) * 4u);
            src[j] += step[j];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // prologue: K-tiles 0 .. NS-2 in flight; wait for tile 0
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < nt) issue(p);
    {
        const int later = min(NS - 2, nt - 1);        // tiles after tile 0 that are in flight
        if (later >= 3) wait_vmcnt<3 * PPW>();
        else if (later == 2) wait_vmcnt<2 * PPW>();
        else if (later == 1) wait_vmcnt<PPW>();
        else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();

    // 4-deep fragment ring: operands are requested TWO k-blocks ahead of their MFMAs
    float fa[4][TM][4], fb[4][TN][4];
    auto frags = [&](const float* stA, int kb, int buf) {
#pragma unroll
        for (int i = 0; i < TM; ++i) load_frag<AK, BM>(stA, wm0 + 32 * i, l31, half, kb, fa[buf][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) load_frag<BKC, BN>(stA + A_IMG, wn0 + 32 * j, l31, half, kb, fb[buf][j]);
    };
    frags(smem, 0, 0);
    frags(smem, 1, 1);

    for (int it = 0; it < nt; ++it) {
        const float* sA = smem + (it % NS) * STAGE;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const int c = kb, n = (kb + 2) & 3;
            // k-blocks 0,1 request k-blocks 2,3 of this tile; k-blocks 2,3 request 0,1 of the next tile (after this tile's
            // barrier, which certifies that tile)
            if (kb < 2) frags(sA, kb + 2, n);
            if (kb == 2) {
                // mid-tile sync: my pieces of K-tile it+1 have landed once only tiles it+2 .. it+NS-2 may be outstanding
                const int later = min(NS - 3, nt - 2 - it);
                if (later >= 2) wait_vmcnt<2 * PPW>();
                else if (later == 1) wait_vmcnt<PPW>();
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                // refill of the stage freed by that barrier with K-tile it+NS-1 (issuing it later / staggered over the
                // following k-blocks was measured 7 % slower: the data then has less time to land)
                if (it + NS - 1 < nt) issue((it + NS - 1) % NS);
            }
            if (kb >= 2 && it + 1 < nt) frags(smem + ((it + 1) % NS) * STAGE, kb - 2, n);
            // MFMA issue is arbitrated by priority, then age.  With equal priorities the oldest wave of a SIMD runs its
            // whole barrier interval first and the youngest runs last and ALONE, with nobody to cover its LDS/barrier
            // stalls.  Priority falls as a wave advances through the interval (k-blocks 2,3,0,1 -> 3,2,1,0), so laggards
            // overtake leaders and the four waves of a SIMD reach the barrier together (+6 % on the 16-wave kernels,
            // neutral to slightly negative on the 4-wave ones, where other workgroups already fill the gaps).
            if (PRIO) {
                if (kb == 2) __builtin_amdgcn_s_setprio(3);
                else if (kb == 3) __builtin_amdgcn_s_setprio(2);
                else if (kb == 0) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i][s2], fb[c][j][s2], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    GemmArgs gs = g;
    if (grp) gs.C = g.C2;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) store_tile(gs, acc[i][j], z, m0 + wm0 + 32 * i, n0 + wn0 + 32 * j + l31, half);
}

template <int WR, int WC, int NS, bool AK, bool BKC, int TM = 1, int TN = 1>
static int launch_dma(const GemmArgs& g, hipStream_t s) {
    constexpr int BM = 32 * WR * TM, BN = 32 * WC * TN;
    constexpr size_t lds = NS * (size_t)(BM + BN) * 32 * sizeof(float);
    auto kern = gemm_dma_kernel<WR, WC, NS, AK, BKC, TM, TN>;
    static LdsOptIn lds_opt;   // one per template instantiation
    if (int rc = ensure_dyn_lds(lds_opt, (const void*)kern, lds)) return rc;
    const int nblk = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN) * g.split_k * (g.A2 ? 2 : 1);
    DPD_LAUNCH(kern, dim3(nblk), dim3(64 * WR * WC), lds, s, g);
    return (int)hipGetLastError();
}

// Opt-in in-stream profiler (bench.py's roofline leg): a hipEvent pair around every GEMM kernel launch, on the
// stream the kernel is launched on.  Off by default; costs two event records per launch when on.
struct GemmProf {
    bool on = false;
    int mode = 0;      // 1: GEMM launches only (dpd_prof_collect); 2: also the bandwidth-bound stages (dpd_prof_collect_stage)
    int n = 0;
    static constexpr int kMax = 8192;
    hipEvent_t ev[2 * kMax];
    double flops[kMax];   // flops (tag 0) or algorithmic HBM bytes (stage tags)
    int tag[kMax];
    signed char form[kMax];   // GEMM launches: 0 = NN / NT, 1 = TN (dpd_prof_collect_form)
    int created = 0;   // events [0, created) exist
};
static GemmProf g_prof;
static std::mutex g_prof_mu;   // the profiler is process-wide (one GEMM stream at a time is the supported use); the lock keeps it memory-safe

// Events are created on demand and destroyed by dpd_prof_enable(0): thousands of live timing events slow every
// later kernel launch of the process down (host side; measured as sporadic 3x slower steps after a profiled pass).
bool prof_begin(hipStream_t s) {
    if (!g_prof.on) return false;                       // the common case takes no lock
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!(g_prof.on && g_prof.n < GemmProf::kMax)) return false;
    for (int i = 2 * g_prof.n; i < 2 * g_prof.n + 2; ++i)
        if (i >= g_prof.created) {
            if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) return false;
            g_prof.created = i + 1;
        }
    (void)hipEventRecord(g_prof.ev[2 * g_prof.n], s);
    return true;
}
void prof_end(bool on, hipStream_t s, double flops, int form) {
    if (!on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof.on) return;
    (void)hipEventRecord(g_prof.ev[2 * g_prof.n + 1], s);
    g_prof.flops[g_prof.n] = flops;
    g_prof.tag[g_prof.n] = DPD_STAGE_GEMM;
    g_prof.form[g_prof.n] = (signed char)form;
    ++g_prof.n;
}
// stages never nest inside each other or inside a GEMM bracket (one open pair at a time: the pair of slot n)
bool prof_begin_stage(hipStream_t s) {
    if (!g_prof.on || g_prof.mode < 2) return false;
    return prof_begin(s);
}
void prof_end_stage(bool on, hipStream_t s, int tag, double bytes) {
    if (!on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof.on) return;
    (void)hipEventRecord(g_prof.ev[2 * g_prof.n + 1], s);
    g_prof.flops[g_prof.n] = bytes;
    g_prof.tag[g_prof.n] = tag;
    ++g_prof.n;
}

template <int BM, int BN, int BK, bool AK, bool BKC>
static int launch_cfg(const GemmArgs& g, hipStream_t s) {
    using StA = TileStage<BM, BK, AK>;
    using StB = TileStage<BN, BK, BKC>;
    constexpr size_t lds = (size_t)(2 * BK * StA::LD + 2 * BK * StB::LD) * sizeof(float);
    auto kern = gemm_f32_kernel<BM, BN, BK, AK, BKC>;
    static LdsOptIn lds_opt;   // one per template instantiation
    if (int rc = ensure_dyn_lds(lds_opt, (const void*)kern, lds)) return rc;
    const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
    const int nblk = tilesM * tilesN * g.split_k;
    DPD_LAUNCH(kern, dim3(nblk), dim3(256), lds, s, g);
    return (int)hipGetLastError();
}

template <bool AK, bool BKC>
static int launch_tile(int tile, const GemmArgs& g, hipStream_t s) {
    if (tile >= 30 && tile <= 33) return launch_rs_tile<AK, BKC>(tile, g, s);   // register-streamed kernels (gemm_rs.h)
    switch (tile) {
        // LDS-DMA ring kernels (round 1): tile 8 serves the NT forms g W^T when the caller passes no transposed weight copies, tile 9 is the
        // 128x128 reference of tools/x3_bench.py.  (The other ring / register-staged configurations were A/B references of rounds 1-3; removed
        // in round 6 -- profiles/r03_gemm_bench.txt has their numbers.)
        case 8: return launch_dma<2, 2, 3, AK, BKC>(g, s);   // LDS-DMA ring,  64x64,  256 thr, 48 KiB  (3 blocks/CU)
        case 9: return launch_dma<4, 4, 3, AK, BKC>(g, s);   // LDS-DMA ring, 128x128, 1024 thr, 96 KiB (1 block/CU)
        case 3: return launch_cfg<64, 64, 32, AK, BKC>(g, s);   // register-staged 64x64: any K % 4 == 0 (the fallback for ragged K)
        default: return DPD_E_UNSUPPORTED;
    }
}

int gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
             int ldc, const float* bias, const float* gate, int epilogue, int split_k, int tile, void* ws,
             size_t ws_bytes, hipStream_t s, float* colsum, const float* A2, const float* B2, float* C2, const ColsumTwoStep* cs2) {
    if (!A || !B || !C) return DPD_E_NULL;
    // split_k == 0: "tail split" (gemm_rs.h): whole-K tiles, only the partial last round of tiles is cut along K (needs ws for
    // (pieces - 1) slabs of M*N floats; pieces <= 4).  Silently a plain launch when it does not apply.
    bool tail_auto = false;
    if (split_k == 0) {
        split_k = 1;
        tail_auto = tile >= 30 && tile <= 33 && ws && ws_bytes >= (size_t)3 * M * N * sizeof(float) && epilogue == EPI_NONE;
    }
    if (M <= 0 || N <= 0 || K <= 0 || split_k < 1) return DPD_E_DIM;
    if ((K & 3) || (N & 3) || (lda & 3) || (ldb & 3) || (ldc & 3)) return DPD_E_UNSUPPORTED;
    if (transA && (M & 3)) return DPD_E_UNSUPPORTED;
    if (transA && transB) return DPD_E_UNSUPPORTED;
    if ((epilogue == EPI_BIAS || epilogue == EPI_BIAS_RELU) && !bias) return DPD_E_NULL;
    if (epilogue == EPI_GATE && !gate) return DPD_E_NULL;
    if (epilogue < 0 || epilogue > 3) return DPD_E_UNSUPPORTED;

    if (tile == 0) tile = 3;     // the register-staged 64x64 kernel: takes every shape
    const bool whole_tiles = (tile == 8 || tile == 9) || (tile >= 30 && tile <= 33);   // kernels that need whole 32-deep K-tiles
    if (whole_tiles && (K % 32 != 0 || M < 4 || N < 4 || (split_k > 1 && ((K + split_k - 1) / split_k + 31) / 32 * 32 * (split_k - 1) >= K)))
        tile = 3;   // these kernels need whole K-tiles (and a non-empty last split): fall back to the register-staged kernel
    GemmArgs g{};
    g.A = A; g.B = B; g.bias = bias; g.gate = gate;
    g.colsum = (split_k > 1) ? nullptr : colsum;
    g.A2 = A2; g.B2 = B2; g.C2 = C2;
    if (tail_auto && !colsum && !A2) { g.tail_split = -1; g.tail_slab = (float*)ws; }
    if (cs2) {   // deterministic bias gradients in two steps (register-streamed kernels only; rows of a partial block = 32)
        if (!(tile >= 30 && tile <= 33) || (cs2->part_out && (split_k > 1 || colsum))) return DPD_E_UNSUPPORTED;
        g.colsum_part = cs2->part_out;
        g.colsum_part_in = cs2->part_in; g.colsum_part_in2 = cs2->part_in2;
        g.colsum_b = cs2->out; g.colsum_b2 = cs2->out2; g.colsum_nparts = cs2->nparts;
    }
    if (A2 && (!B2 || !C2 || split_k > 1 || epilogue != EPI_NONE || colsum)) return DPD_E_UNSUPPORTED;
    if (A2 && !whole_tiles) return DPD_E_UNSUPPORTED;   // grouped launches exist for the DMA / register-streamed kernels only
    if (colsum && split_k > 1) return DPD_E_UNSUPPORTED;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb;
    g.split_k = split_k;
    if (split_k > 1) {
        const int chunk = (((K + split_k - 1) / split_k) + 31) / 32 * 32;
        if ((size_t)split_k * M * N * sizeof(float) > ws_bytes || !ws) return DPD_E_WORKSPACE;
        g.k_chunk = chunk; g.C = (float*)ws; g.ldc = N; g.slab_stride = (long)M * N; g.epi = EPI_NONE;
    } else {
        g.k_chunk = (K + 31) / 32 * 32; g.C = C; g.ldc = ldc; g.slab_stride = 0; g.epi = epilogue;
    }
    // profiler bracket: the GEMM kernel AND, with split-K, its reduce kernel (both belong to this GEMM)
    struct ProfScope {
        bool on; hipStream_t s; double fl; int form;
        ~ProfScope() { prof_end(on, s, fl, form); }
    } prof_scope{prof_begin(s), s, 2.0 * M * N * K, transA ? 1 : 0};
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
    std::lock_guard<std::mutex> lk(dpd::g_prof_mu);
    g_prof.on = on != 0;
    g_prof.mode = on;
    if (on) {
        g_prof.n = 0;
    } else {   // release the events (dpd_prof_collect must have been called before)
        for (int i = 0; i < g_prof.created; ++i) (void)hipEventDestroy(g_prof.ev[i]);
        g_prof.created = 0;
        g_prof.n = 0;
    }
    return 0;
}

// Synchronises with the recorded events; returns the number of GEMM launches seen since dpd_prof_enable(1) and
// fills total milliseconds / total (padded-shape) flops 2*M*N*K of those launches.
static int prof_collect_gemm(int form, double* total_ms, double* total_flops) {
    using dpd::g_prof;
    std::lock_guard<std::mutex> lk(dpd::g_prof_mu);
    double ms = 0.0, fl = 0.0;
    int n = 0;
    for (int i = 0; i < g_prof.n; ++i) {
        if (g_prof.tag[i] != dpd::DPD_STAGE_GEMM || (form >= 0 && g_prof.form[i] != form)) continue;
        if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) return -1;
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return -1;
        ms += t;
        fl += g_prof.flops[i];
        ++n;
    }
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    return n;
}
extern "C" int dpd_prof_collect(double* total_ms, double* total_flops) { return prof_collect_gemm(-1, total_ms, total_flops); }
// the same for the launches of ONE product form: 0 = NN / NT (forward layers and data gradients: one kernel family), 1 = TN (weight gradients)
extern "C" int dpd_prof_collect_form(int form, double* total_ms, double* total_flops) {
    if (form < 0 || form > 1) return DPD_E_DIM;
    return prof_collect_gemm(form, total_ms, total_flops);
}

// The bandwidth-bound stages recorded since dpd_prof_enable(2): launches of stage `tag` (1 encoder, 2 window gather, 3 fused output
// layer, 4 optimizer, 5 small-gradient reduction, 6 weight copies), their summed duration [ms] and ALGORITHMIC HBM bytes.
extern "C" int dpd_prof_collect_stage(int tag, double* total_ms, double* total_bytes) {
    using dpd::g_prof;
    if (tag <= 0 || tag >= dpd::DPD_STAGE_COUNT) return DPD_E_DIM;
    std::lock_guard<std::mutex> lk(dpd::g_prof_mu);
    double ms = 0.0, by = 0.0;
    int n = 0;
    for (int i = 0; i < g_prof.n; ++i) {
        if (g_prof.tag[i] != tag) continue;
        if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) return -1;
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return -1;
        ms += t;
        by += g_prof.flops[i];
        ++n;
    }
    if (total_ms) *total_ms = ms;
    if (total_bytes) *total_bytes = by;
    return n;
}

extern "C" int dpd_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B,
                            int ldb, float* Cout, int ldc, const float* bias, const float* gate, int epilogue,
                            int split_k, int tile, void* ws, size_t ws_bytes, void* stream) {
    return dpd::gemm_f32(transA, transB, M, N, K, A, lda, B, ldb, Cout, ldc, bias, gate, epilogue, split_k, tile, ws,
                         ws_bytes, (hipStream_t)stream, nullptr, nullptr, nullptr, nullptr, nullptr);
}
