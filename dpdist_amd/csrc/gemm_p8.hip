// Phase-staggered plane GEMM (gemm_p8_kernel) and the chained persistent launch built on its tile routine (gemm_chain_kernel).
// Split from gemm_x3.hip in round 5 (its own translation unit: the two halves compile in parallel).
#include "gemm_x3.h"

namespace dpd {

// ---------------------------------------------------------------------------------------------------------
// gemm_p8_kernel: one bf16 plane, BK = 64, phase-staggered schedule (the "8-phase" structure of the CDNA4 guide, section 5,
// re-derived for this library's chunked plane layouts and 32x32x16 MFMAs).
//
// The lock-step ring kernel above is additive: every wave issues its LDS-DMA pieces, then its fragment reads, then its MFMAs,
// and all eight waves do each of these at the same time (MFMA 25 us + DMA issue 16 us + fragment reads 7 us on
// the layer-1 shape).  Here the workgroup is two GROUPS of NW/2 waves (waves w and w + NW/2 share a SIMD) that run the same
// program ONE BARRIER APART, so that on every SIMD one wave is inside its MFMA cluster (at raised priority) while the other
// issues ds_reads and DMA pieces:
//
//     K-tile t = phases 2t (k16 steps 0,1) and 2t+1 (steps 2,3); stage = t % 3 (three whole K-tiles of LDS).
//     phase p of a wave:   LOAD(p): fragment reads of phase p; DMA pieces: p = 2t   -> second half of this wave's pieces of K-tile t+1
//                                                                          p = 2t+1 -> first half of K-tile t+2;
//                                   odd p: s_waitcnt vmcnt(first half of t+2 stays in flight)  => my pieces of K-tile t+1 landed
//                          s_barrier (B1)    MFMA(p): TM*TN*2 MFMAs, s_setprio 1    s_barrier (B2)
//     group 1 executes one extra barrier before phase 0 and group 0 one after the last phase: between two consecutive
//     workgroup barriers one group is in LOAD, the other in MFMA.
//
// Hazards (global barrier index: group 0 passes 2p / 2p+1 around MFMA(p), group 1 passes 2p+1 / 2p+2):
//   RAW  K-tile t+1 is read from LOAD(2t+2) on.  Every wave waits for its own pieces of t+1 before ITS B1(2t+1) (index 4t+2 for
//        group 0, 4t+3 for group 1); group 0's LOAD(2t+2) starts after index 4t+3, group 1's after 4t+4: behind both.
//   WAR  K-tile t+2 goes to stage (t+2)%3 = (t-1)%3, last read in LOAD(2t-1), whose reads have returned before that wave's
//        MFMA(2t-1) ends (the MFMAs consume them), i.e. before index 4t-1 (group 0) / 4t (group 1).  The first pieces of t+2 are
//        issued in LOAD(2t+1): after index 4t+1 (group 0) / 4t+2 (group 1): behind both.
// K % 64 == 32 (the decoder's 2528): the lanes whose chunk lies beyond K in the last K-tile fetch a zero chunk instead.
// ---------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) const unsigned g_zero_chunk[4] = {0u, 0u, 0u, 0u};
#ifdef DPD_ABLATIONS
// ABL & 32: wave 0 of every workgroup leaves s_memtime stamps at the kernel's milestones (tools/p8_stamps.py)
__device__ unsigned long long g_p8_stamps[1024 * 8];
#define P8_STAMP(i) do { if ((ABL & 32) && tid == 0) g_p8_stamps[(blockIdx.x & 1023) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define P8_STAMP(i) do { } while (0)
#endif

// ABL (timing-only, -DDPD_ABLATIONS): 1 = no LDS-DMA in the loop, 2 = no barriers, 4 = no fragment reads, 8 = no stagger, 16 = no setprio
// NP planes (1: BK = 64, two k16 steps per phase; 3: BK = 32, one k16 step = six MFMA terms per phase): a K-tile is 48 KiB of LDS for
// a 256x128 (NP = 1) or 128x128 (NP = 3) tile either way.
// One output tile of the phase-staggered schedule: prologue, K loop, epilogue.  Shared by gemm_p8_kernel (one tile per workgroup) and
// gemm_chain_kernel (persistent workgroups, several dependent GEMMs per launch).
//   SC1:   plane outputs by write-through stores (x3_epilogue).
//   SPLIT: the prologue issues the B pieces (weights: they never depend on another workgroup) of K-tiles 0 and 1 FIRST, then calls dep()
//          -- the chained kernel's wait for the producers of this tile's A rows -- and only then issues the A pieces.
template <int NP, bool AK, bool BKC, int WR, int WC, int TM, int TN, bool LATE_WAIT, int ABL, bool SC1, bool SPLIT, typename Dep>
__device__ __forceinline__ void p8_tile(const X3Args& g, char* smem_x3, int grp, const uint16_t* gA, const uint16_t* gB, int m0, int n0, int tid,
                                        Dep dep) {
    constexpr int BK = NP == 1 ? 64 : 32, NS = 3, CPR = BK / 8, KS = BK / 32;   // KS = k16 steps per phase (half a K-tile)
    constexpr int BM = 32 * WR * TM, BN = 32 * WC * TN, NW = WR * WC;
    constexpr int A_IMG = BM * CPR, B_IMG = BN * CPR, PL = A_IMG + B_IMG, STAGE = NP * PL;   // chunks of 16 B
    constexpr int PA = A_IMG / 64, PB = B_IMG / 64;                            // 1-KiB pieces per plane image
    constexpr int PPW = NP * (PA + PB) / NW, HP = PPW / 2;                     // pieces per wave per K-tile / per phase
    static_assert((NP * (PA + PB)) % NW == 0 && PPW % 2 == 0, "piece split");
    static_assert(NW % 2 == 0, "two wave groups");
    static_assert(AK || BM % 64 == 0, "R8 images need 64-row pieces");
    static_assert(BKC || BN % 64 == 0, "R8 images need 64-row pieces");
    static_assert(NS * STAGE * 16 <= 160 * 1024, "LDS");
    constexpr int JA = PA / NW;                                                // SPLIT: pieces [0, JA) of every wave are A pieces, the rest B
    static_assert(!SPLIT || (NP == 1 && PA % NW == 0 && PB % NW == 0), "split prologue: whole A / B piece ranges per wave");

    P8_STAMP(0);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wgrp = __builtin_amdgcn_readfirstlane(wave / (NW / 2));          // 0: waves 0..NW/2-1, 1: the rest
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave / WC) * 32 * TM, wn0 = (wave % WC) * 32 * TN;

    // (every field of `g` the address set-up needs, read ONCE and unconditionally: in the chained kernel `g` is a run-time choice among
    //  kernel-argument blocks, and a conditional use such as isA ? g.lda : g.ldb turns into a dependent scalar load per DMA piece)
    const int M = g.e.M, N = g.e.N;
    const int K = g.e.K;
    const int g_lda = g.lda, g_ldb = g.ldb;
    const long g_a_plane = NP > 1 ? g.a_plane : 0, g_b_plane = NP > 1 ? g.b_plane : 0;
    const int nt = (K + BK - 1) / BK;
    const int tail_groups = (K % BK) / 8;      // != 0: the last K-tile has this many valid k-groups (K % 8 == 0)
    const bool ktail = tail_groups != 0;

    const uint16_t* src[PPW];
    long step[PPW];
    unsigned dst[PPW];
    unsigned tail_ok = 0;                      // bit j: this lane's chunk of piece j is inside K in the tail K-tile
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem_x3;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int p = wave + j * NW;
        const int plane = p / (PA + PB), w = p % (PA + PB);
        const bool isA = w < PA;
        const int c = isA ? w : w - PA;
        const bool kc = isA ? AK : BKC;
        const uint16_t* base = isA ? gA + plane * g_a_plane : gB + plane * g_b_plane;
        const int ld = isA ? g_lda : g_ldb;
        const int o0 = isA ? m0 : n0;
        const int O = isA ? M : N;
        const int BO = isA ? BM : BN;
        dst[j] = lds_base + (unsigned)(plane * PL + (isA ? 0 : A_IMG) + c * 64) * 16u;
        int kg;
        if (kc) {
            const int row = c * (64 / CPR) + lane / CPR, slot = lane % CPR;
            kg = slot ^ ((row / (16 / CPR)) & (CPR - 1));
            src[j] = base + (size_t)min(o0 + row, O - 1) * ld + 8 * kg;
            step[j] = BK;
        } else {
            const int lin = c * 64 + lane;
            kg = lin / BO;
            const int o = lin % BO;
            src[j] = base + ((size_t)kg * ld + min(o0 + o, O - 1)) * 8;
            step[j] = (long)CPR * ld * 8;
        }
        tail_ok |= (kg < tail_groups ? 1u : 0u) << j;
    }
    // pieces [j0, j0 + cnt) of K-tile `tile` into stage `stage`; every piece is issued exactly once per K-tile, in K-tile order
    auto issue = [&](int tile, int stage, auto j0c, auto cntc) {
        constexpr int j0 = decltype(j0c)::value, cnt = decltype(cntc)::value;
        const bool tail = ktail && tile == nt - 1;
#pragma unroll
        for (int j = j0; j < j0 + cnt; ++j) {
            const void* sp = src[j];
            if (tail && !((tail_ok >> j) & 1u)) sp = g_zero_chunk;
            dma_piece(sp, dst[j] + (unsigned)(stage * STAGE) * 16u);
            src[j] += step[j];
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using IH = std::integral_constant<int, HP>;
    using IP = std::integral_constant<int, PPW>;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // prologue: K-tiles 0 and 1 whole; K-tile 0 landed and visible before anybody's LOAD(0)
    P8_STAMP(1);
    if (SPLIT) {
        using IA = std::integral_constant<int, JA>;
        using IB = std::integral_constant<int, PPW - JA>;
        issue(0, 0, IA{}, IB{});
        if (nt > 1) issue(1, 1, IA{}, IB{});
        dep();
        issue(0, 0, I0{}, IA{});
        if (nt > 1) {
            issue(1, 1, I0{}, IA{});
            wait_vm<JA>();          // in flight at most: my A pieces of K-tile 1 (everything of K-tile 0, and the B pieces of K-tile 1, were issued before them)
        } else {
            wait_vm<0>();
        }
    } else {
        issue(0, 0, I0{}, IP{});
        if (nt > 1) {
            issue(1, 1, I0{}, IP{});
            wait_vm<PPW>();
        } else {
            wait_vm<0>();
        }
    }
    __builtin_amdgcn_s_barrier();
    P8_STAMP(2);
    if (wgrp == 1 && !(ABL & 8)) __builtin_amdgcn_s_barrier();      // the stagger: group 1 runs one barrier behind group 0

    bf16x8 fa[KS][NP][TM], fb[KS][NP][TN];
    auto phase = [&](int t, auto stc, auto hc) {
        constexpr int st = decltype(stc)::value, h = decltype(hc)::value;
        const char* sbase = smem_x3 + (size_t)st * STAGE * 16;
        // ---- LOAD(p) ----
#pragma unroll
        for (int s2 = 0; s2 < ((ABL & 4) ? (t == 0 && h == 0 ? KS : 0) : KS); ++s2) {
            const int kg = 2 * (KS * h + s2) + half;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[s2][p][i] = *reinterpret_cast<const bf16x8*>(sbase + (p * PL + chunk_of<AK, BM, CPR>(wm0 + 32 * i + l31, kg)) * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[s2][p][j] =
                        *reinterpret_cast<const bf16x8*>(sbase + (p * PL + A_IMG + chunk_of<BKC, BN, CPR>(wn0 + 32 * j + l31, kg)) * 16);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ABL & 1) {
        } else if (h == 0) {
            if (t >= 1 && t + 1 < nt) issue(t + 1, (st + 1) % NS, IH{}, IH{});
        } else {
            if (t + 2 < nt) issue(t + 2, (st + 2) % NS, I0{}, IH{});
            if (!LATE_WAIT || wgrp == 1) {
                if (t + 1 < nt) {
                    if (t + 2 < nt) wait_vm<HP>();
                    else wait_vm<0>();
                }
            }
        }
        if (!(ABL & 2)) __builtin_amdgcn_s_barrier();                  // B1
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMA(p) ----
        if (!(ABL & 16)) __builtin_amdgcn_s_setprio(1);
        if (NP == 3) {      // lo*hi + hi*lo + mid*mid + mid*hi + hi*mid + hi*hi, small terms first (same order as gemm_x3_kernel)
            constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2)
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s2][ta[q] < NP ? ta[q] : 0][i], fb[s2][tb[q] < NP ? tb[q] : 0][j],
                                                                                acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s2][0][i], fb[s2][0][j], acc[i][j], 0, 0, 0);
        }
        if (!(ABL & 16)) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (LATE_WAIT && h == 1 && wgrp == 0 && !(ABL & 1)) {        // group 0's B2 is the barrier group 1 waits before: one MFMA cluster more to land
            if (t + 1 < nt) {
                if (t + 2 < nt) wait_vm<HP>();
                else wait_vm<0>();
            }
        }
        if (!(ABL & 2)) __builtin_amdgcn_s_barrier();                  // B2
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    int t = 0;
    for (; t + 3 <= nt; t += 3) {
        phase(t, C0{}, C0{});
        phase(t, C0{}, C1{});
        phase(t + 1, C1{}, C0{});
        phase(t + 1, C1{}, C1{});
        phase(t + 2, C2{}, C0{});
        phase(t + 2, C2{}, C1{});
    }
    if (t < nt) {
        phase(t, C0{}, C0{});
        phase(t, C0{}, C1{});
        if (t + 1 < nt) {
            phase(t + 1, C1{}, C0{});
            phase(t + 1, C1{}, C1{});
        }
    }
    if (wgrp == 0 && !(ABL & 8)) __builtin_amdgcn_s_barrier();      // group 0 catches up: every wave has passed the same number of barriers
    P8_STAMP(3);
    x3_epilogue<BM, BN, NW, TM, TN, NP, SC1>(g, acc, smem_x3, grp, 0, m0, n0, wm0, wn0, tid, l31, half);
    P8_STAMP(4);
}

struct NoDep {
    __device__ __forceinline__ void operator()() const {}
};

template <int NP, bool AK, bool BKC, int WR, int WC, int TM, int TN, bool LATE_WAIT, int ABL = 0>
__global__ __launch_bounds__(64 * WR * WC) void gemm_p8_kernel(X3Args g) {
    constexpr int BM = 32 * WR * TM, BN = 32 * WC * TN;
    extern __shared__ __attribute__((aligned(16))) char smem_x3[];
    const int tilesM = (g.e.M + BM - 1) / BM, tilesN = (g.e.N + BN - 1) / BN;
    const int per_z = tilesM * tilesN;
    const int sid0 = xcd_remap(blockIdx.x, per_z * (g.A2 ? 2 : 1));
    const int grp = sid0 / per_z;
    const int t0 = sid0 % per_z;
    p8_tile<NP, AK, BKC, WR, WC, TM, TN, LATE_WAIT, ABL, false, false>(g, smem_x3, grp, grp ? g.A2 : g.A, grp ? g.B2 : g.B, (t0 / tilesN) * BM,
                                                                      (t0 % tilesN) * BN, threadIdx.x, NoDep{});
}

// ---------------------------------------------------------------------------------------------------------
// gemm_chain_kernel (round 5): up to three DEPENDENT one-plane GEMMs with the same output shape [M, N] in ONE persistent launch --
// the decoder's forward layers 1 -> 2 -> 3 (NN: rows x weights, bias + ReLU, plane outputs) and its data-gradient chain g3 -> g2 -> g1
// (NT, ReLU gate, plane outputs).  Launched apart, each of these GEMMs pays 10-13 us of fill / plane epilogue / drain around a K loop of
// 8-36 us (DESIGN.md 3.4 b, 3.5 d), and every launch ends with 256 CUs draining and the next starts with 256 CUs filling.  Here a tile of
// stage s + 1 and row band r starts as soon as the tilesN tiles of band r of stage s have been published -- no grid-wide boundary.
//
// Work distribution = TICKETS, which is what makes the launch deadlock-free under ANY residency or placement (HIP promises neither
// dispatch order nor co-residency: MI355X_MICROARCH.md "Workgroup dispatch"): there is one queue per XCD, queue x lists the tiles of the row
// bands x, x + 8, ... stage by stage (all of stage 0, then all of stage 1, ...), band by band, and a workgroup takes its next tile with one
// atomic increment of its queue's counter.  A tile depends only on tiles with LOWER tickets of the SAME queue, and a ticket is only ever
// held by a running workgroup, so by induction every wait ends.  A workgroup that finds its home queue (its own XCD: the consumer then
// finds the producers' rows in its L2's neighbourhood -- speed only) exhausted goes on to the other queues, so every queue drains wherever
// the workgroups landed.
//
// Hand-off (cdna_hip_programming.md Guideline 16, recipe R1): plane outputs by 16-byte write-through (sc1) stores, every storing wave drains
// them (s_waitcnt vmcnt(0)), barrier, ONE lane adds 1 to the band's arrival word; the consumer's lane 0 polls that ONE word relaxed (with
// s_sleep), then ONE agent-scope acquire (drops this CU's stale L1 lines), barrier, then plain LDS-DMA loads.  The weights' pieces of the
// first two K-tiles are issued BEFORE the poll (p8_tile<SPLIT>): they depend on nobody.
//
// State: `sync` = kChainWords 32-bit words that must be ZERO at launch and are left zero: the last workgroup to leave (an exit counter,
// behind a drained vmcnt) clears every word it or anybody else touched.  The error word is sticky: a poll that gives up (spin_limit) sets
// bit 0 and the tile goes ahead (the result is then wrong, the launch still ends) -- dpd_planes_sync_status() reads it.
// Results are bitwise those of the same GEMMs launched apart: same tile routine, same K order, same epilogue.
// ---------------------------------------------------------------------------------------------------------
constexpr int kChainQueues = 8, kChainMaxBands = 320, kChainArr = 256, kChainTickStride = 32;
// ticket counter of queue x at [32 x] (a 128-byte line each: 32 workgroups hammer one), exit counter [8], error [9], arrivals [256 + s * 320 + band]
constexpr int kChainWords = kChainArr + 2 * kChainMaxBands;
static_assert(kChainWords * 4 <= DPD_SYNC_BYTES, "dpd_planes.sync");

struct ChainArgs {
    X3Args st[3];
    unsigned* sync;
    unsigned long long* stamps;      // optional [gridDim][4 tiles][8] s_memtime stamps of lane 0 (tools/chain_stamps.py), NULL = off
    int nst, tilesM, tilesN;
    unsigned spin_limit;
};

__device__ __forceinline__ unsigned chain_add(unsigned* p) { return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one poll of an arrival word on the VECTOR path, past this CU's L1 (sc1): a uniform address would otherwise be polled through the scalar cache
__device__ __forceinline__ unsigned chain_poll(const unsigned* p) {
    unsigned r;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p) : "memory");
    return r;
}

template <bool BKC, int WR, int WC, int TM, int TN, bool LATE_WAIT>
__global__ __launch_bounds__(64 * WR * WC) void gemm_chain_kernel(ChainArgs c) {
    constexpr int BM = 32 * WR * TM, BN = 32 * WC * TN;
    constexpr unsigned RING = 3u * (BM + BN) * 64 * 2;
    extern __shared__ __attribute__((aligned(16))) char smem_x3[];
    // behind the ring: [0] the next ticket, [1 .. 8] a snapshot of the eight queue counters (which queue to go on with when this one is exhausted)
    volatile unsigned* box = reinterpret_cast<volatile unsigned*>(smem_x3 + RING);
    const int tid = threadIdx.x;
    unsigned* tick = c.sync;
    unsigned* arr = c.sync + kChainArr;
    const int nst = c.nst, tilesM = c.tilesM, tilesN = c.tilesN;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    int q = (int)(xcc & (kChainQueues - 1)), done_tiles = 0;
    auto stamp = [&](int i) {
        if (c.stamps && tid == 0 && done_tiles < 4 && blockIdx.x < 256) c.stamps[((size_t)blockIdx.x * 4 + done_tiles) * 8 + i] = __builtin_amdgcn_s_memrealtime();
    };
    auto queue_tiles = [&](int x) { return ((tilesM - x + kChainQueues - 1) / kChainQueues) * tilesN * nst; };
    auto take = [&]() {
        if (tid == 0) box[0] = chain_add(tick + q * kChainTickStride);
    };
    take();
    __syncthreads();
    unsigned ticket = __builtin_amdgcn_readfirstlane(box[0]);
    for (;;) {
        while (ticket >= (unsigned)queue_tiles(q)) {       // queue exhausted: lanes 0..7 read the eight counters (ONE round trip); a queue that is not, or leave
            __syncthreads();
            if (tid < kChainQueues) box[1 + tid] = __hip_atomic_load(tick + tid * kChainTickStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            int nq = -1;
            for (int i = 1; i < kChainQueues; ++i) {
                const int x = (q + i) & (kChainQueues - 1);
                if (nq < 0 && box[1 + x] < (unsigned)queue_tiles(x)) nq = x;
            }
            if (nq < 0) goto leave;
            q = nq;
            __syncthreads();
            take();
            __syncthreads();
            ticket = __builtin_amdgcn_readfirstlane(box[0]);
        }
        const int per = queue_tiles(q) / nst;
        const int stg = (int)ticket / per, rem = (int)ticket % per;
        const int band = q + kChainQueues * (rem / tilesN), col = rem % tilesN;
        stamp(0);
        const X3Args& g = c.st[stg];
        auto dep = [&]() {
            if (stg > 0) {
                if (tid == 0) {
                    unsigned* w = arr + (stg - 1) * kChainMaxBands + band;
                    unsigned spins = 0;
                    while (chain_poll(w) < (unsigned)tilesN) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > c.spin_limit) {
                            __hip_atomic_fetch_or(c.sync + 9, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
            }
            stamp(1);
        };
        p8_tile<1, true, BKC, WR, WC, TM, TN, LATE_WAIT, 0, true, true>(g, smem_x3, 0, g.A, g.B, band * BM, col * BN, tid, dep);
        stamp(2);
        // the next ticket: requested behind this wave's stores, so that its round trip hides under their drain (hipcc waits for a returning
        // atomic at the end of the lane-0 branch: anywhere earlier that wait would stall wave 0, and with it the workgroup, ~1 us per tile)
        take();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // EVERY storing wave: its write-through stores are acknowledged
        __syncthreads();
        if (tid == 0 && stg + 1 < nst) chain_add(arr + stg * kChainMaxBands + band);
        stamp(3);
        ++done_tiles;
        ticket = __builtin_amdgcn_readfirstlane(box[0]);
    }
leave:
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // all of this workgroup's accesses to the words have been performed
        if (chain_add(c.sync + 8) == gridDim.x - 1) {                              // the last one out leaves the words as it found them: zero
            for (int i = 0; i < kChainQueues; ++i) __hip_atomic_store(tick + i * kChainTickStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int st = 0; st + 1 < nst; ++st)
                for (int b = 0; b < tilesM; ++b) __hip_atomic_store(arr + st * kChainMaxBands + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(c.sync + 8, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <bool BKC, int WR, int WC, int TM, int TN, bool LATE_WAIT>
static int launch_chain(const ChainArgs& c, hipStream_t s) {
    constexpr int BM = 32 * WR * TM, BN = 32 * WC * TN;
    constexpr size_t lds = (size_t)3 * (BM + BN) * 64 * 2 + 48;
    static_assert(lds <= 160 * 1024, "LDS");
    auto kern = gemm_chain_kernel<BKC, WR, WC, TM, TN, LATE_WAIT>;
    static LdsOptIn lds_opt;
    if (int rc = ensure_dyn_lds(lds_opt, (const void*)kern, lds)) return rc;
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    const int tiles = c.tilesM * c.tilesN;
    DPD_LAUNCH(kern, dim3(tiles < cus[dev] ? tiles : cus[dev]), dim3(64 * WR * WC), lds, s, c);
    return (int)hipGetLastError();
}

// tile: 21 = 256x128 (8 waves of 64x64), 23 = 128x128 (8 waves of 32x64).  Returns DPD_E_UNSUPPORTED for anything the chained form does not
// take (the caller then launches the GEMMs apart): ragged tiles, fp32 outputs of an intermediate stage, more than one plane.
int gemm_chain(int nst, const ChainStage* st, int M, int N, int tile, unsigned* sync, unsigned long long* stamps, hipStream_t s) {
    if (nst < 2 || nst > 3 || !st || !sync) return DPD_E_UNSUPPORTED;
    const int bm = tile == 21 ? 256 : (tile == 23 ? 128 : 0), bn = 128;
    if (!bm || M <= 0 || N <= 0 || (M % bm) || (N % bn)) return DPD_E_UNSUPPORTED;
    if (M / bm > kChainMaxBands) return DPD_E_UNSUPPORTED;
    ChainArgs c{};
    double flops = 0.0;
    for (int i = 0; i < nst; ++i) {
        const ChainStage& t = st[i];
        const bool last = i + 1 == nst;
        if (!t.A || !t.B || t.K <= 0 || (t.K % 32) || (t.lda & 7) || (t.ldb & 7) || t.b_fmt != st[0].b_fmt) return DPD_E_UNSUPPORTED;
        if (!last && (t.C || !t.out.rc || st[i + 1].A != t.out.rc || st[i + 1].lda != t.out.ld_rc || st[i + 1].K != N)) return DPD_E_UNSUPPORTED;
        if (!t.C && !t.out.rc && !t.out.r8) return DPD_E_NULL;
        if ((t.out.rc || t.out.r8) && (t.out.np != 1 || (t.out.ld_rc & 7) || (t.out.r8 && (t.out.r8_rows % bm)))) return DPD_E_UNSUPPORTED;
        if (t.C && ((t.ldc & 3) || (N & 3))) return DPD_E_UNSUPPORTED;
        if ((t.epilogue == EPI_BIAS || t.epilogue == EPI_BIAS_RELU) && !t.bias) return DPD_E_NULL;
        if (t.epilogue == EPI_GATE && !t.gate16) return DPD_E_NULL;
        if (t.epilogue < 0 || t.epilogue > 3) return DPD_E_UNSUPPORTED;
        X3Args& g = c.st[i];
        g.e.C = t.C; g.e.bias = t.bias; g.e.gate16 = t.gate16; g.e.gate16_r8 = t.gate16_r8; g.e.colsum = t.colsum;
        g.e.M = M; g.e.N = N; g.e.K = t.K; g.e.ldc = t.C ? t.ldc : N; g.e.epi = t.epilogue;
        g.e.split_k = 1; g.e.k_chunk = t.K; g.e.slab_stride = 0;
        g.A = t.A; g.B = t.B; g.a_plane = (long)M * t.K; g.b_plane = (long)t.K * N; g.lda = t.lda; g.ldb = t.ldb;
        g.out_rc = t.out.rc; g.out_r8 = t.out.r8; g.rc_plane = t.out.rc_plane; g.r8_plane = t.out.r8_plane;
        g.ld_rc = t.out.ld_rc; g.r8_rows = t.out.r8_rows; g.np_out = 1;
        flops += 2.0 * M * N * t.K;
    }
    c.sync = sync; c.stamps = stamps; c.nst = nst; c.tilesM = M / bm; c.tilesN = N / bn;
    static const unsigned spin = [] { const char* e = getenv("DPD_CHAIN_SPIN_LIMIT"); return e ? (unsigned)strtoul(e, nullptr, 10) : 4000000u; }();
    c.spin_limit = spin;
    struct ProfScope {
        bool on; hipStream_t s; double fl;
        ~ProfScope() { prof_end(on, s, fl); }
    } prof_scope{prof_begin(s), s, flops};
    const bool nt = st[0].b_fmt == 0;      // B as RC planes (k contiguous): the data-gradient chain
    if (tile == 21) return nt ? launch_chain<true, 4, 2, 2, 2, true>(c, s) : launch_chain<false, 4, 2, 2, 2, true>(c, s);
    return nt ? launch_chain<true, 4, 2, 1, 2, false>(c, s) : launch_chain<false, 4, 2, 1, 2, false>(c, s);
}

template <int NP, bool AK, bool BKC, int WR, int WC, int TM, int TN, bool LATE_WAIT, int ABL = 0>
static int launch_p8(const X3Args& g, hipStream_t s) {
    constexpr int BM = 32 * WR * TM, BN = 32 * WC * TN, BK = NP == 1 ? 64 : 32;
    constexpr size_t ring = (size_t)3 * NP * (BM + BN) * BK * 2, stage = (size_t)BM * (BN + 4) * 4;
    constexpr size_t lds = ring > stage ? ring : stage;
    static_assert(lds <= 160 * 1024, "LDS");
    auto kern = gemm_p8_kernel<NP, AK, BKC, WR, WC, TM, TN, LATE_WAIT, ABL>;
    static LdsOptIn lds_opt;
    if (int rc = ensure_dyn_lds(lds_opt, (const void*)kern, lds)) return rc;
    const int nblk = ((g.e.M + BM - 1) / BM) * ((g.e.N + BN - 1) / BN) * (g.A2 ? 2 : 1);
    DPD_LAUNCH(kern, dim3(nblk), dim3(64 * WR * WC), lds, s, g);
    return (int)hipGetLastError();
}


template <bool AK, bool BKC>
static int launch_p8_tile(int np, int tile, const X3Args& g, hipStream_t s) {
    switch (tile) {
        case 20: if (np == 1) return launch_p8<1, AK, BKC, 4, 2, 2, 2, false>(g, s); return DPD_E_UNSUPPORTED;   // 256x128, 8 waves of 64x64
        case 21: if (np == 1) return launch_p8<1, AK, BKC, 4, 2, 2, 2, true>(g, s); return DPD_E_UNSUPPORTED;    // ... group 0 waits after its MFMAs
        case 22: if (np == 1) return launch_p8<1, AK, BKC, 2, 4, 2, 2, false>(g, s); return DPD_E_UNSUPPORTED;   // 128x256, 8 waves of 64x64
        case 23: if (np == 1) return launch_p8<1, AK, BKC, 4, 2, 1, 2, false>(g, s); return DPD_E_UNSUPPORTED;   // 128x128, 8 waves of 32x64
        case 24: if (np == 3) return launch_p8<3, AK, BKC, 4, 2, 1, 2, true>(g, s); return DPD_E_UNSUPPORTED;    // 128x128, 8 waves of 32x64, 3 planes
        case 25: if (np == 3) return launch_p8<3, AK, BKC, 2, 4, 2, 1, true>(g, s); return DPD_E_UNSUPPORTED;    // 128x128, 8 waves of 64x32, 3 planes
        case 26: if (np == 3) return launch_p8<3, AK, BKC, 4, 2, 1, 2, false>(g, s); return DPD_E_UNSUPPORTED;   // 24 with both groups waiting before B1
#ifdef DPD_ABLATIONS
#define DPD_P8_ABL(code) case 200 + code: if (np == 1) return launch_p8<1, AK, BKC, 4, 2, 2, 2, true, code>(g, s); return DPD_E_UNSUPPORTED;
        DPD_P8_ABL(32) DPD_P8_ABL(1) DPD_P8_ABL(2) DPD_P8_ABL(3) DPD_P8_ABL(4) DPD_P8_ABL(5) DPD_P8_ABL(7) DPD_P8_ABL(8) DPD_P8_ABL(16) DPD_P8_ABL(24)
#undef DPD_P8_ABL
#endif
        default: return DPD_E_UNSUPPORTED;
    }
}

int launch_p8_code(int np, bool ak, bool bkc, int tile, const X3Args& g, hipStream_t s) {
    if (ak && !bkc) return launch_p8_tile<true, false>(np, tile, g, s);      // NN
    if (ak && bkc) return launch_p8_tile<true, true>(np, tile, g, s);        // NT
    if (!ak && !bkc) return launch_p8_tile<false, false>(np, tile, g, s);    // TN
    return DPD_E_UNSUPPORTED;
}

}  // namespace dpd

#ifdef DPD_ABLATIONS
extern "C" int dpd_debug_p8_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(dpd::g_p8_stamps), sizeof(unsigned long long) * 1024 * 8, 0, hipMemcpyDeviceToHost);
}
#endif
