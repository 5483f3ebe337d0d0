// Register-streamed fp32 MFMA GEMM (gfx950 / CDNA4): no LDS, no barriers.
//
// Why: v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate (64 cycles per instruction per SIMD), so one wave needs only
// ONE VGPR per operand per 64 cycles -- 16x less operand bandwidth per flop than a bf16 GEMM.  At that rate the L1/L2 path
// can feed the MFMA operands straight into registers; what limited the LDS-ring kernel of gemm_f32.hip (73 % of the MFMA peak)
// was not data but synchronisation: a workgroup barrier per K-tile plus the LDS-DMA refill cost it 22 % (ablations in
// DESIGN.md section 3.1).  Here every wave is independent:
//   * a wave owns a (32*TM) x (32*TN) output tile (TM*TN accumulators of 16 VGPRs) and streams its own operands with
//     buffer loads, one K-tile (32 contraction steps) ahead of the MFMAs (double-buffered in VGPRs, counted vmcnt by hipcc);
//   * lane (l31, half) of a 32x32x2 MFMA supplies A[l31][k] / B[k][l31] for k = the lane's half of the k pair.  Inside a
//     32-deep K-tile the contraction index is assigned as  k = 32 t + 16 half + 4 b + s  (block b = 0..3, step s = 0..3):
//       K-contiguous operand  -> the lane reads 64 contiguous bytes of its row as four dwordx4 (b = 0..3): the four loads
//                                of a half-wave hit the same 32 x 128-B lines back to back (L1 hits after the first);
//       MN-contiguous operand -> one dword per (b, s): 32 lanes read 128 contiguous bytes of row k (perfectly coalesced);
//     A and B use the same assignment, which is all the contraction needs;
//   * addresses: one buffer descriptor per operand, a loop-invariant per-lane voffset, the K progress in the SCALAR offset
//     -> no vector address arithmetic in the loop; out-of-range rows/columns are clamped (never stored);
//   * the workgroup (WR x WC waves, one per SIMD for 2x2) only exists to place waves that share operand rows/columns on
//     one CU (L1 reuse) and to make the block -> tile map XCD-aware; waves never wait for each other, so ragged tiles
//     simply retire early.
// (A fused-window-gather form of this kernel -- X never materialised, the A operand gathered from the Fisher vectors per lane -- was
// built in round 2, was bitwise, and measured slower (layer 1: 180 us against 147 + 13.5 us: the per-lane address arithmetic sits in the
// in-order issue stream of an MFMA-bound wave; profiles/r02_variants.txt); removed in round 6.)
// Results are bitwise identical to the LDS kernels (a k-ordered fmaf chain per output element; the permutation inside a
// K-tile changes the ORDER of the chain, so "identical" holds between rs configurations, not against the ring kernels).
#pragma once
#include "gemm_shared.h"

namespace dpd {

typedef unsigned rs_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rs_rsrc(const void* p, size_t bytes) {
    // raw (untyped, stride 0) buffer; word3 = 0x00020000 is the gfx9/CDNA data format for dword MUBUF accesses
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0xffffffffull ? 0xffffffffull : bytes), 0x00020000);
}

// One operand tile (32 rows/cols of the wave tile) of one K-tile: v[b][s] = operand value for block b, step s
struct RsFrag {
    float v[4][4];
};

// the loads of one operand tile that belong to step (b, s) of a K-tile (issued one pipeline depth ahead of their use)
template <bool KC>
__device__ __forceinline__ void rs_load_step(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned k0, unsigned ld, int b, int s, RsFrag& f) {
    if (KC) {   // voff = (row*ld + 16*half)*4 ; soffset = k0*4 ; block b -> +16 B ; one dwordx4 carries the four steps
        if (s == 0) {
            const rs_u4 x = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16u * b, k0 * 4u, 0);
            f.v[b][0] = __uint_as_float(x.x); f.v[b][1] = __uint_as_float(x.y);
            f.v[b][2] = __uint_as_float(x.z); f.v[b][3] = __uint_as_float(x.w);
        }
    } else {    // voff = (16*half*ld + col)*4 ; soffset = (k0 + 4b + s)*ld*4
        f.v[b][s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, (k0 + 4u * b + s) * ld * 4u, 0));
    }
}

// D = pipeline depth in K-tiles (operand register sets): the loads of K-tile t+D-1 are issued, step by step, between the
// MFMAs of K-tile t, so every operand has D-1 whole tile times ((TM*TN*16) MFMAs x 64 cycles each) to arrive.
template <bool AK, bool BKC, int TM, int TN, int WR, int WC, int D>
__global__ __launch_bounds__(64 * WR * WC) void gemm_rs_kernel(GemmArgs g) {
    constexpr int BM = 32 * TM * WR, BN = 32 * TN * WC;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;

    const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
    const int per_z = tilesM * tilesN;
    const int ngrp = g.A2 ? 2 : 1;
    int grp = 0, z, t;
    unsigned kbeg, kend;
    if (g.tail_split > 1) {     // whole-K tiles first, then the K pieces of the last round's tiles
        const int sid = xcd_remap(blockIdx.x, g.tail_first + (per_z - g.tail_first) * g.tail_split);
        if (sid < g.tail_first) { t = sid; z = 0; kbeg = 0; kend = g.K; }
        else {
            const int u = sid - g.tail_first;
            t = g.tail_first + u / g.tail_split; z = u % g.tail_split;
            kbeg = z * g.tail_chunk; kend = min(g.K, (int)kbeg + g.tail_chunk);
        }
    } else {
        const int sid = xcd_remap(blockIdx.x, per_z * g.split_k * ngrp);
        grp = sid / (per_z * g.split_k);
        const int sid1 = sid % (per_z * g.split_k);
        z = sid1 / per_z; t = sid1 % per_z;
        kbeg = z * g.k_chunk; kend = min(g.K, (int)kbeg + g.k_chunk);
    }
    const int m0 = (t / tilesN) * BM + (wave / WC) * 32 * TM, n0 = (t % tilesN) * BN + (wave % WC) * 32 * TN;
    if (m0 >= g.M || n0 >= g.N) return;   // ragged edge: this wave has no output (no barriers anywhere, so it may leave)
    const int nt = (int)(kend - kbeg) / 32;
    const float* gA = grp ? g.A2 : g.A;
    const float* gB = grp ? g.B2 : g.B;
    const unsigned lda = g.lda, ldb = g.ldb;

    const __amdgpu_buffer_rsrc_t ra = rs_rsrc(gA, AK ? ((size_t)(g.M - 1) * lda + g.K) * 4 : ((size_t)(g.K - 1) * lda + g.M) * 4);
    const __amdgpu_buffer_rsrc_t rb = rs_rsrc(gB, BKC ? ((size_t)(g.N - 1) * ldb + g.K) * 4 : ((size_t)(g.K - 1) * ldb + g.N) * 4);
    unsigned va[TM], vb[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const unsigned r = min(m0 + 32 * i + l31, g.M - 1);
        va[i] = AK ? (r * lda + 16u * half) * 4u : (16u * half * lda + r) * 4u;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const unsigned c = min(n0 + 32 * j + l31, g.N - 1);
        vb[j] = BKC ? (c * ldb + 16u * half) * 4u : (16u * half * ldb + c) * 4u;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- epilogue operands, requested before the main loop (they depend on nothing computed here): the bias value of this
    // lane's column and, for the ReLU-gate epilogue of the backward data GEMMs, the 16 gate values per accumulator tile.  Held
    // in registers when the wave tile is small enough (<= 2 accumulator tiles); otherwise fetched at the end like before.
    constexpr bool PRE = TM * TN <= 2;
    const int epi = g.epi;
    float ebias[TN], egate[PRE ? TM : 1][PRE ? TN : 1][16];
#pragma unroll
    for (int j = 0; j < TN; ++j)
        ebias[j] = (epi == EPI_BIAS || epi == EPI_BIAS_RELU) ? g.bias[min(n0 + 32 * j + l31, g.N - 1)] : 0.f;
    if (PRE && epi == EPI_GATE && g.split_k == 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half, g.M - 1);
                    egate[i][j][r] = g.gate[(size_t)row * g.ldc + min(n0 + 32 * j + l31, g.N - 1)];
                }
    }
    RsFrag fa[D][TM], fb[D][TN];
    // all loads of K-tile kt (clamped to the last tile: a redundant reload instead of a branch) into register set d
    auto load_step = [&](int kt, int d, int b, int s) {
        const int ktc = min(kt, nt - 1);
        const unsigned k0 = kbeg + 32u * ktc;
#pragma unroll
        for (int i = 0; i < TM; ++i) rs_load_step<AK>(ra, va[i], k0, lda, b, s, fa[d][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) rs_load_step<BKC>(rb, vb[j], k0, ldb, b, s, fb[d][j]);
    };
    // One K-tile: 16 steps of TM*TN MFMAs on register set `cur`; the loads of K-tile `kt_next` go into set `nxt` in the
    // same (b, s) order in which they will be consumed.  sched_barrier pins [loads of the step | MFMAs of the step]: left
    // alone, the machine scheduler sinks the loads to the end of the tile and halves the prefetch distance.
    auto tile = [&](int cur, int nxt, int kt_next, bool do_load) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (do_load) load_step(kt_next, nxt, b, s);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].v[b][s], fb[cur][j].v[b][s], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
    };

    // prologue: K-tiles 0 .. D-2 in flight
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int s = 0; s < 4; ++s) load_step(d, d, b, s);
    // The loop body is branch-free on purpose: with a conditional load hipcc's waitcnt pass must assume the not-taken path
    // and waits for the loads it has just issued; a straight-line body gets the exact counted vmcnt.
    int kt = 0;
    for (; kt + D <= nt; kt += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) tile(j, (j + D - 1) % D, kt + j + D - 1, true);
    }
    // remainder (< D tiles): their operands are already in flight in sets 0 .. rem-1
#pragma unroll
    for (int j = 0; j < D - 1; ++j)
        if (kt + j < nt) tile(j, 0, 0, false);

    GemmArgs gs = g;
    if (grp) gs.C = g.C2;
    if (g.tail_split > 1 && z > 0) {      // K piece of a tail tile: its own slab, dense [M, N]
        gs.C = g.tail_slab + (size_t)(z - 1) * g.M * g.N;
        gs.ldc = g.N;
    }
    const int zs = (g.tail_split > 1) ? 0 : z;     // slab index for put_tile (split-K slabs only)
    // Bias gradient, second step (see GemmArgs::colsum_part): the first row block's waves add the 32-row partial column sums an
    // EARLIER launch stored, in block order -> deterministic, no atomics, no extra launch.  (Adding them up inside the K loop of
    // this kernel instead was measured 25 % slower on the dW GEMMs: VALU work between the dependent MFMAs of a single accumulator.)
    float* csb_out = grp ? g.colsum_b2 : g.colsum_b;
    const float* csb_in = grp ? g.colsum_part_in2 : g.colsum_part_in;
    if (csb_out && csb_in && m0 == 0 && z == 0 && half == 0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + 32 * j + l31;
            if (col < g.N) {
                float t = 0.f;
                for (int pb0 = 0; pb0 < g.colsum_nparts; pb0 += 16) {     // 16 loads in flight, added in block order
                    float v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = (pb0 + u < g.colsum_nparts) ? csb_in[(size_t)(pb0 + u) * g.N + col] : 0.f;
#pragma unroll
                    for (int u = 0; u < 16; ++u) t += v[u];
                }
                csb_out[col] = t;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (PRE && g.split_k == 1 && epi != EPI_NONE) {     // epilogue on the preloaded operands (same arithmetic as tile_values)
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float x = acc[i][j][r] + ebias[j];
                    if (epi == EPI_BIAS_RELU) x = fmaxf(x, 0.f);
                    if (epi == EPI_GATE) x = (egate[i][j][r] > 0.f) ? x : 0.f;
                    v[r] = x;
                }
                put_tile(gs, v, zs, m0 + 32 * i, n0 + 32 * j + l31, half);
            } else {
                store_tile(gs, acc[i][j], zs, m0 + 32 * i, n0 + 32 * j + l31, half);
            }
        }
}

// C[r][:] += sum_z slab_z[r][:] for the rows of the tail tiles (fixed order; float4, N % 4 == 0)
__global__ __launch_bounds__(256) void tail_reduce_kernel(float* __restrict__ C, int ldc, const float* __restrict__ slab, int nslab,
                                                           long slab_stride, int row0, int M, int N) {
    const long total4 = (long)(M - row0) * N / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long e = i * 4;
        const int row = row0 + (int)(e / N), col = (int)(e % N);
        float4 a = *reinterpret_cast<const float4*>(C + (size_t)row * ldc + col);
        for (int z = 0; z < nslab; ++z) {
            const float4 x = *reinterpret_cast<const float4*>(slab + (size_t)z * slab_stride + (size_t)row * N + col);
            a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
        }
        *reinterpret_cast<float4*>(C + (size_t)row * ldc + col) = a;
    }
}

template <bool AK, bool BKC, int TM, int TN, int WR, int WC, int D>
static int launch_rs(const GemmArgs& g_in, hipStream_t s) {
    constexpr int BM = 32 * TM * WR, BN = 32 * TN * WC;
    GemmArgs g = g_in;
    const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN, T = tilesM * tilesN;
    int nblk = T * g.split_k * (g.A2 ? 2 : 1);
    int row0 = 0;
    if (g.tail_split == -1) {
        // Tail split requested (tail_slab given): T workgroup tiles on 256 CUs run in ceil(T / 256) rounds; when the last round
        // is less than 3/4 full its tiles are cut into K pieces so that it takes 1/pieces of a round (dW1: 640 tiles = 2.5 rounds
        // -> 512 whole tiles + 128 tiles x 2 halves: 3 rounds become 2.5).  Whole tile ROWS only, >= 8 K-tiles per piece.
        g.tail_split = 0;
        const int r = T % 256, nfull = T - r;
        if (T > 256 && r > 0 && r < 192 && g.split_k == 1 && !g.A2 && g.epi == EPI_NONE && !g.colsum && !g.colsum_part && g.tail_slab) {
            int pieces = (256 + r / 2) / r;
            pieces = pieces > 4 ? 4 : pieces;
            const int first = (nfull / tilesN) * tilesN;
            const int chunk = ((g.K / 32 + pieces - 1) / pieces) * 32;
            if (pieces >= 2 && first > 0 && chunk >= 256 && chunk * (pieces - 1) < g.K) {
                g.tail_first = first; g.tail_split = pieces; g.tail_chunk = chunk;
                nblk = first + (T - first) * pieces;
                row0 = (first / tilesN) * BM;
            }
        }
    }
    DPD_LAUNCH((gemm_rs_kernel<AK, BKC, TM, TN, WR, WC, D>), dim3(nblk), dim3(64 * WR * WC), 0, s, g);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return (int)e;
    if (g.tail_split > 1) {
        const long total4 = (long)(g.M - row0) * g.N / 4;
        const int blocks = (int)((total4 + 255) / 256 < 1024 ? (total4 + 255) / 256 : 1024);
        DPD_LAUNCH(tail_reduce_kernel, dim3(blocks), dim3(256), 0, s, g.C, g.ldc, (const float*)g.tail_slab, g.tail_split - 1,
                   (long)g.M * g.N, row0, g.M, g.N);
        return (int)hipGetLastError();
    }
    return 0;
}

// tile codes 30..33: register-streamed kernels (wave tile, waves per workgroup, pipeline depth)
template <bool AK, bool BKC>
static int launch_rs_tile(int tile, const GemmArgs& g, hipStream_t s) {
    switch (tile) {
        case 30: return launch_rs<AK, BKC, 2, 2, 2, 2, 2>(g, s);   // 128x128 workgroup, 4 waves of 64x64, 2 operand sets
        case 31: return launch_rs<AK, BKC, 2, 1, 2, 2, 2>(g, s);   // 128x64,  4 waves of 64x32
        case 32: return launch_rs<AK, BKC, 1, 2, 2, 2, 2>(g, s);   //  64x128, 4 waves of 32x64
        case 33: return launch_rs<AK, BKC, 1, 1, 2, 2, 2>(g, s);   //  64x64,  4 waves of 32x32
        default: return DPD_E_UNSUPPORTED;
    }
}

}  // namespace dpd
