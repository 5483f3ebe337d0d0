// Phase-staggered plane GEMM (gemm_p8_kernel).
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

// (Round 5 built a persistent CHAINED launch on p8_tile -- layers 1 -> 2 -> 3 / g3 -> g2 -> g1 as one launch with per-XCD ticket queues and
// per-band arrival words; SC1 / SPLIT / Dep are its hooks.  Bitwise the separate launches, 11 us SLOWER per chain at B = 64
// (profiles/r05_chain_bench.txt: a hand-off costs what a kernel boundary costs, and a tile takes as long inside the persistent launch as
// apart); removed in round 6, DESIGN.md section 3.6 keeps the finding.)

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
        case 21: if (np == 1) return launch_p8<1, AK, BKC, 4, 2, 2, 2, true>(g, s); return DPD_E_UNSUPPORTED;    // 256x128, 8 waves of 64x64 (group 0 waits after its MFMAs)
        case 23: if (np == 1) return launch_p8<1, AK, BKC, 4, 2, 1, 2, false>(g, s); return DPD_E_UNSUPPORTED;   // 128x128, 8 waves of 32x64
        case 24: if (np == 3) return launch_p8<3, AK, BKC, 4, 2, 1, 2, true>(g, s); return DPD_E_UNSUPPORTED;    // 128x128, 8 waves of 32x64, 3 planes
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
