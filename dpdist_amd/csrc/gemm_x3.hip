// fp32-accurate GEMM on the bf16 matrix cores (gfx950 / CDNA4): split-bf16 ("bf16x3") MFMA.
//
// gfx950 has no reduced-precision fast path for fp32 inputs (no xf32) and its fp32 MFMA runs at 1/16 of the bf16
// rate (157 vs 2500 TFLOP/s).  An fp32 value is therefore carried as THREE bf16 planes
//      a = hi + mid + lo,   hi = bf16(a), mid = bf16(a - hi), lo = bf16(a - hi - mid)     (|a - hi - mid - lo| <= 2^-27 |a|)
// and a product a*b is evaluated with the six bf16 MFMAs whose weight is >= 2^-16:
//      lo*hi + hi*lo + mid*mid + mid*hi + hi*mid + hi*hi          (dropped: mid*lo, lo*mid, lo*lo <= 3 * 2^-25 |a b|)
// Each bf16 product is exact in fp32 and the matrix core accumulates in fp32, so the result has fp32-class error
// (measured against fp64 in tests/test_gpu_parity.py next to the exact-fp32 MFMA kernel of gemm_f32.hip) at 6/16 of
// the fp32 instruction time: 419 TFLOP/s of fp32-equivalent peak.  NP = 1 runs the same kernel on a single bf16 plane
// (the plain bf16 training path, BASELINE config 3).
//
// Operand planes live in HBM in one of two chunked layouts (chunk = 8 bf16 = 16 B = one lane's MFMA operand):
//      RC  (k = column index):  plane[r][c]            -> chunk (o = r, kg = c/8) at (r*ld + 8*kg)
//      R8  (k = row index):     plane[r/8][c][r%8]     -> chunk (o = c, kg = r/8) at (kg*ld + c)*8
// so every fragment is ONE ds_read_b128 whichever way the GEMM contracts, and every LDS image is lane-linear for the
// LDS-DMA (global_load_lds_dwordx4): RC images are [rows][BK/8 slots] with the slot XOR-swizzled on the source address,
// R8 images are [BK/8][rows] dense.  Producers (split_planes_kernel here, later the fused epilogues) write the planes.
//
// Kernel structure = gemm_f32.hip's LDS-DMA ring: NS stages, pieces issued NS-1 K-tiles ahead, one raw s_barrier per
// K-tile, counted vmcnt, fragments requested one k16-step ahead of their MFMAs.
#include "gemm_x3.h"

namespace dpd {

// In-launch split-K reduction (cdna_hip_programming.md, "In-launch split-K reduction"; the dW GEMMs of the bf16 step: K = query rows
// is long, M x N gives 64-160 tiles of 128x128 for 256 CUs).  One agent-scope release per slice, one agent-scope acquire per tile:
//   every slice: raw accumulators -> its slab (plain 16-byte stores, lane-linear) -> every wave s_waitcnt vmcnt(0) -> barrier ->
//                thread 0: release fence (agent) + the restated vmcnt(0) wait -> relaxed arrival at the tile's counter word;
//   the slice that draws the last ticket: thread 0 acquire fence (agent) -> barrier -> all waves add the slabs of slice 0, 1, ...
//                in THAT order (fp32 addition is not associative: a fixed order makes the result independent of who arrives last,
//                bitwise reproducible) -> the ordinary epilogue.
// Correct for any placement of a tile's slices over XCDs; the block-id map only makes the common placement fast (slices of a tile are
// neighbours in the logical id, i.e. on one XCD: the reducer then finds the slabs in its own L2).
// The arrival word carries the launch's generation: a word left by another generation (uninitialised workspace, an aborted launch)
// counts as zero arrivals, and the last arriver leaves {generation, 0} behind so that a replay of the same captured launch starts clean.
// Returns true in the reducing slice (acc = the sum).  `flag` = one dword of the idle LDS ring (no second __shared__ object).
template <int NW, int TM, int TN>
__device__ __forceinline__ bool inlaunch_reduce(const X3Args& g, f32x16 (&acc)[TM][TN], char* smem, int tile_id, int z, int tid) {
    const int split = g.e.split_k;
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int ITEM4 = NW * TM * TN * 4 * 64;                      // float4s per slice slab
    float4* slab = reinterpret_cast<float4*>(g.red_slab) + (size_t)tile_id * split * ITEM4;
    float4* mine = slab + (size_t)z * ITEM4 + (size_t)wave * (TM * TN * 4 * 64) + lane;
    const bool sc1 = g.red_sc1 != 0;
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slab, 0, (int)((size_t)split * ITEM4 * 16), 0x00020000);
    const unsigned voff_w = (unsigned)(wave * (TM * TN * 4 * 64) + lane) * 16u;
    if (sc1) {      // write-through: the slab is visible to every XCD once the stores have been acknowledged (no buffer_wbl2)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u4v v = {__float_as_uint(acc[i][j][4 * q]), __float_as_uint(acc[i][j][4 * q + 1]), __float_as_uint(acc[i][j][4 * q + 2]),
                                   __float_as_uint(acc[i][j][4 * q + 3])};
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff_w + (unsigned)(((i * TN + j) * 4 + q) * 64) * 16u,
                                                           (unsigned)z * (unsigned)(ITEM4 * 16), 16);
                }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    mine[((i * TN + j) * 4 + q) * 64] = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(smem);
    if (tid == 0) {
        if (!sc1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        unsigned long long* c = g.red_cnt + tile_id;
        const unsigned long long gen = (unsigned long long)g.red_gen << 32;
        unsigned long long old = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned n;
        do {
            n = ((old >> 32) == g.red_gen ? (unsigned)old : 0u) + 1u;
        } while (!__hip_atomic_compare_exchange_strong(c, &old, gen | n, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const bool last = n == (unsigned)split;
        if (last) {
            __hip_atomic_store(c, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!sc1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        *flag = last ? 1u : 0u;
    }
    __syncthreads();
    if (!*flag) return false;
    const float4* src = slab + (size_t)wave * (TM * TN * 4 * 64) + lane;
    for (int zz = 0; zz < split; ++zz) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v;
                    if (sc1) {
                        const u4v u = __builtin_amdgcn_raw_buffer_load_b128(rs, voff_w + (unsigned)(((i * TN + j) * 4 + q) * 64) * 16u,
                                                                            (unsigned)zz * (unsigned)(ITEM4 * 16), 16);
                        v = make_float4(__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3]));
                    } else {
                        v = src[(size_t)zz * ITEM4 + ((i * TN + j) * 4 + q) * 64];
                    }
                    if (zz == 0) {
                        acc[i][j][4 * q] = v.x; acc[i][j][4 * q + 1] = v.y; acc[i][j][4 * q + 2] = v.z; acc[i][j][4 * q + 3] = v.w;
                    } else {
                        acc[i][j][4 * q] += v.x; acc[i][j][4 * q + 1] += v.y; acc[i][j][4 * q + 2] += v.z; acc[i][j][4 * q + 3] += v.w;
                    }
                }
    }
    return true;
}

// ABL (timing-only ablations, instantiated only with -DDPD_ABLATIONS; results are wrong by construction):
//   1 = no LDS-DMA refill in the K loop, 2 = no barrier, 4 = no fragment reads in the loop, 8 = one MFMA per step only.
// TR: operands that are not K-contiguous come as RCT images of their RC planes (above) instead of R8 planes
template <int NP, bool AK, bool BKC, int WR, int WC, int TM, int TN, int NS, int BK, int ABL = 0, bool TR = false>
__global__ __launch_bounds__(64 * WR * WC) void gemm_x3_kernel(X3Args g) {
    constexpr int BM = 32 * WR * TM, BN = 32 * WC * TN, NW = WR * WC;
    constexpr int CPR = BK / 8, KB = BK / 16;               // chunks per row, k16 steps per K-tile
    constexpr int A_IMG = BM * CPR, B_IMG = BN * CPR;       // chunks per plane image
    constexpr int PL = A_IMG + B_IMG, STAGE = NP * PL;      // chunks
    constexpr int PA = A_IMG / 64, PB = B_IMG / 64;         // 1-KiB pieces per plane
    constexpr int PPW = NP * (PA + PB) / NW;                // pieces per wave per K-tile
    static_assert((NP * (PA + PB)) % NW == 0, "piece split");
    static_assert(AK || BM % 64 == 0, "R8 images need 64-row pieces");
    static_assert(BKC || BN % 64 == 0, "R8 images need 64-row pieces");
    constexpr int NT = NP == 3 ? 6 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem_x3[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave / WC) * 32 * TM, wn0 = (wave % WC) * 32 * TN;

    const int N = g.e.N;
    const int tilesN = (N + BN - 1) / BN;
    // up to three grouped problems (rows M0, M1, M2): block id -> (problem, tile, K slice)
    const int Mp1 = g.A2 ? (g.M2 ? g.M2 : g.e.M) : 0, Mp2 = g.A3 ? g.M3 : 0;
    const int nt0 = ((g.e.M + BM - 1) / BM) * tilesN, nt1 = ((Mp1 + BM - 1) / BM) * tilesN, nt2 = ((Mp2 + BM - 1) / BM) * tilesN;
    const int SK = g.e.split_k;
    const int sid0 = xcd_remap(blockIdx.x, (nt0 + nt1 + nt2) * SK);
    const int grp = sid0 >= (nt0 + nt1) * SK ? 2 : (sid0 >= nt0 * SK ? 1 : 0);
    const int per_z = grp == 0 ? nt0 : (grp == 1 ? nt1 : nt2);
    const int tile_base = grp == 0 ? 0 : (grp == 1 ? nt0 : nt0 + nt1);
    const int sid = sid0 - tile_base * SK;
    const int M = grp == 0 ? g.e.M : (grp == 1 ? Mp1 : Mp2);
    // slab split-K: slice-major (a slab is written tile after tile); in-launch reduction: tile-major, so that the slices of a tile are
    // neighbours in the logical id and land on one XCD (a speed choice only: inlaunch_reduce is placement-independent)
    const int z = g.red_cnt ? sid % SK : sid / per_z, t = g.red_cnt ? sid / SK : sid % per_z;
    const uint16_t* gA = grp == 0 ? g.A : (grp == 1 ? g.A2 : g.A3);
    const uint16_t* gB = grp == 0 ? g.B : (grp == 1 ? g.B2 : g.B3);
    const int lda_g = grp == 0 ? g.lda : (grp == 1 ? (g.lda2 ? g.lda2 : g.lda) : g.lda3);
    const long a_plane_g = grp == 0 ? g.a_plane : (grp == 1 ? (g.a_plane2 ? g.a_plane2 : g.a_plane) : g.a_plane3);
    const int m0 = (t / tilesN) * BM, n0 = (t % tilesN) * BN;
    const int kbeg = z * g.e.k_chunk;
    const int kend = min(g.e.K, kbeg + g.e.k_chunk);
    const int nt = (kend - kbeg) / BK;

    // this wave's DMA pieces: piece p = wave + j*NW -> (plane, operand, 1-KiB piece c of that plane image)
    const uint16_t* src[PPW];
    long step[PPW];
    unsigned dst[PPW];
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem_x3;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int p = wave + j * NW;
        const int plane = p / (PA + PB), w = p % (PA + PB);
        const bool isA = w < PA;
        const int c = isA ? w : w - PA;
        const bool kc = isA ? AK : BKC;
        const uint16_t* base = isA ? gA + plane * a_plane_g : gB + plane * g.b_plane;
        const int ld = isA ? lda_g : g.ldb;
        const int o0 = isA ? m0 : n0;
        const int O = isA ? M : N;
        const int BO = isA ? BM : BN;
        dst[j] = lds_base + (unsigned)(plane * PL + (isA ? 0 : A_IMG) + c * 64) * 16u;
        if (kc) {
            const int row = c * (64 / CPR) + lane / CPR, slot = lane % CPR;
            const int kg = slot ^ ((row / (16 / CPR)) & (CPR - 1));
            src[j] = base + (size_t)min(o0 + row, O - 1) * ld + kbeg + 8 * kg;
            step[j] = BK;
        } else if (TR) {      // RCT image: whole row segments of the RC plane, chunk slot swizzled by the row
            const int CH = BO / 8;
            const int row = c * (64 / CH) + lane / CH, slot = lane % CH;
            const int chunk = slot ^ ((row & 3) << 1);
            src[j] = base + (size_t)(kbeg + row) * ld + min(o0 + 8 * chunk, O - 8);
            step[j] = (long)BK * ld;
        } else {
            const int lin = c * 64 + lane;
            const int kg = lin / BO, o = lin % BO;
            src[j] = base + ((size_t)(kbeg / 8 + kg) * ld + min(o0 + o, O - 1)) * 8;
            step[j] = (long)CPR * ld * 8;
        }
    }
    auto issue = [&](int stage) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            dma_piece(src[j], dst[j] + (unsigned)(stage * STAGE) * 16u);
            src[j] += step[j];
        }
    };

    static_assert((NS - 2) * PPW <= 63, "vmcnt range");
    auto wait_later = [&](int later) {   // leave `later` whole K-tiles of this wave's pieces in flight
        if (later >= 6) wait_vm<(NS >= 8 ? 6 : 0) * PPW>();
        else if (later == 5) wait_vm<(NS >= 7 ? 5 : 0) * PPW>();
        else if (later == 4) wait_vm<(NS >= 6 ? 4 : 0) * PPW>();
        else if (later == 3) wait_vm<(NS >= 5 ? 3 : 0) * PPW>();
        else if (later == 2) wait_vm<(NS >= 4 ? 2 : 0) * PPW>();
        else if (later == 1) wait_vm<PPW>();
        else wait_vm<0>();
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
    wait_later(min(NS - 2, nt - 1));
    __builtin_amdgcn_s_barrier();

    bf16x8 fa[2][NP][TM], fb[2][NP][TN];
    auto frags = [&](int stage, int kb, int buf) {
        const char* st = smem_x3 + (size_t)stage * STAGE * 16;
        const int kg = 2 * kb + half;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[buf][p][i] = (TR && !AK) ? rct_frag<BM>(st + (size_t)p * PL * 16, wm0 + 32 * i, kb, lane)
                                            : *reinterpret_cast<const bf16x8*>(st + (p * PL + chunk_of<AK, BM, CPR>(wm0 + 32 * i + l31, kg)) * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[buf][p][j] = (TR && !BKC) ? rct_frag<BN>(st + (size_t)(p * PL + A_IMG) * 16, wn0 + 32 * j, kb, lane)
                                             : *reinterpret_cast<const bf16x8*>(st + (p * PL + A_IMG + chunk_of<BKC, BN, CPR>(wn0 + 32 * j + l31, kg)) * 16);
        }
    };
    frags(0, 0, 0);

    // one k16 step; CUR (compile time) = fragment buffer holding this step's operands
    auto do_step = [&](int it, int kb, auto curc) {
        constexpr int cur = decltype(curc)::value;
        if (kb == KB - 1) {
            // my pieces of K-tile it+1 have landed once only tiles it+2 .. it+NS-2 may be outstanding
            if (!(ABL & 1)) wait_later(min(NS - 3, nt - 2 - it));
            if (!(ABL & 2)) __builtin_amdgcn_s_barrier();
            // everybody is past K-tile it-1: refill its stage with K-tile it+NS-1
            if (!(ABL & 1) && it + NS - 1 < nt) issue((it + NS - 1) % NS);
            // unconditional (the last iteration reads a stale stage and never uses it): behind a branch hipcc falls back to
            // s_waitcnt lgkmcnt(0) before the MFMAs below, i.e. they would wait for the reads that were only just issued
            if (!(ABL & 4)) frags((it + 1) % NS, 0, cur ^ 1);
        } else {
            if (!(ABL & 4)) frags(it % NS, kb + 1, cur ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (NP == 3) {
            constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int q = 0; q < NT; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] =
                            __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][ta[q]][i], fb[cur][tb[q]][j], acc[i][j], 0, 0, 0);
        } else if (ABL & 8) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][0][0], fb[cur][0][0], acc[0][0], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(ABL & 4) ? 0 : cur][0][i], fb[(ABL & 4) ? 0 : cur][0][j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    if (KB == 4) {          // BK = 64: four k16 steps per K-tile, one barrier per 4 * TM * TN MFMAs
        for (int it = 0; it < nt; ++it) {
            do_step(it, 0, C0{});
            do_step(it, 1, C1{});
            do_step(it, 2, C0{});
            do_step(it, 3, C1{});
        }
    } else if (KB == 2) {
        for (int it = 0; it < nt; ++it) {
            do_step(it, 0, C0{});
            do_step(it, 1, C1{});
        }
    } else {
        for (int it = 0; it < nt; it += 2) {
            do_step(it, 0, C0{});
            if (it + 1 < nt) do_step(it + 1, 0, C1{});
        }
    }
    if (g.red_cnt) {
        if (!inlaunch_reduce<NW, TM, TN>(g, acc, smem_x3, tile_base + t, z, tid)) return;
        x3_epilogue<BM, BN, NW, TM, TN, NP>(g, acc, smem_x3, grp, 0, m0, n0, wm0, wn0, tid, l31, half);
        return;
    }
    x3_epilogue<BM, BN, NW, TM, TN, NP>(g, acc, smem_x3, grp, z, m0, n0, wm0, wn0, tid, l31, half);
}

template <int NP, bool AK, bool BKC, int WR, int WC, int TM, int TN, int NS, int BK, int ABL = 0, bool TR = false>
static int launch_x3(const X3Args& g, hipStream_t s) {
    constexpr int BM = 32 * WR * TM, BN = 32 * WC * TN;
    constexpr size_t ring = (size_t)NS * NP * (BM + BN) * BK * 2, stage = (size_t)BM * (BN + 4) * 4;
    constexpr size_t lds = ring > stage ? ring : stage;   // the plane epilogue stages the fp32 tile in the ring's LDS
    static_assert(lds <= 160 * 1024, "LDS");
    auto kern = gemm_x3_kernel<NP, AK, BKC, WR, WC, TM, TN, NS, BK, ABL, TR>;
    static LdsOptIn lds_opt;   // one per template instantiation
    if (int rc = ensure_dyn_lds(lds_opt, (const void*)kern, lds)) return rc;
    const int tn = (g.e.N + BN - 1) / BN;
    const int nblk = (((g.e.M + BM - 1) / BM) + (g.A2 ? ((g.M2 ? g.M2 : g.e.M) + BM - 1) / BM : 0) + (g.A3 ? (g.M3 + BM - 1) / BM : 0)) * tn * g.e.split_k;
    DPD_LAUNCH(kern, dim3(nblk), dim3(64 * WR * WC), lds, s, g);
    return (int)hipGetLastError();
}

// tile codes: 1 = 128x128 (4 waves of 64x64), 2 = 128x128 (8 waves of 64x32), 3 = 64x128, 4 = 128x64, 5 = 64x64, 13 = 192x128 (BK 64, one plane);
// 21 / 23 / 24 = the phase-staggered kernels of gemm_p8.hip
// TN products with both operands read from their RC planes (RCT images + LDS transpose reads)
template <int NP>
static int launch_x3_tile_tr(int tile, const X3Args& g, hipStream_t s) {
    switch (tile) {
        case 1: return launch_x3<NP, false, false, 2, 2, 2, 2, NP == 3 ? 3 : 4, 32, 0, true>(g, s);
        case 2: return launch_x3<NP, false, false, 2, 4, 2, 1, NP == 3 ? 3 : 4, 32, 0, true>(g, s);
        case 3: return launch_x3<NP, false, false, 2, 2, 1, 2, 4, 32, 0, true>(g, s);
        case 5: return launch_x3<NP, false, false, 2, 2, 1, 1, 4, 32, 0, true>(g, s);
        default: return DPD_E_UNSUPPORTED;
    }
}

template <int NP, bool AK, bool BKC>
static int launch_x3_tile(int tile, const X3Args& g, hipStream_t s) {
    switch (tile) {
        case 1: return launch_x3<NP, AK, BKC, 2, 2, 2, 2, NP == 3 ? 3 : 4, 32>(g, s);
        case 2: return launch_x3<NP, AK, BKC, 2, 4, 2, 1, NP == 3 ? 3 : 4, 32>(g, s);
        case 3: return launch_x3<NP, AK, BKC, 2, 2, 1, 2, 4, 32>(g, s);
        case 4: return launch_x3<NP, AK, BKC, 2, 2, 2, 1, 4, 32>(g, s);
        case 5: return launch_x3<NP, AK, BKC, 2, 2, 1, 1, 4, 32>(g, s);
        // 192x128 at BK = 64 (round 4, one plane only: a 3-plane stage would not fit the LDS): the tile with the smallest BM + BN (= LDS fill bytes
        // per flop) that covers dW1 + dW2 + dW3 in at most one workgroup per CU (112 + 48 + 48 = 208 tiles): 8 waves of 96x32, 3 stages x 40 KiB.
        // (BK = 16 rings, the other BK = 64 shapes and the four-wave 96x64 / 64x96 forms were A/B references of rounds 2-4; removed in round 6 --
        // profiles/r03_x3_bench*.txt, r04_bf16_trio_sweep.txt hold their numbers.)
        case 13: if (NP == 1) return launch_x3<1, AK, BKC, 2, 4, 3, 1, 3, 64>(g, s); return DPD_E_UNSUPPORTED;
        // phase-staggered kernels (gemm_p8.hip; K % 32 == 0, no split-K): one plane at BK = 64 (20..23), three planes at BK = 32 (24..26)
        case 21: case 23: case 24:
            return g.e.split_k == 1 ? launch_p8_code(NP, AK, BKC, tile, g, s) : DPD_E_UNSUPPORTED;
#ifdef DPD_ABLATIONS
        case 232: case 201: case 202: case 203: case 204: case 205: case 207: case 208: case 216: case 224:
            return launch_p8_code(NP, AK, BKC, tile, g, s);
#define DPD_X3_ABL(code) case 100 + code: if (NP == 1) return launch_x3<1, AK, BKC, 2, 4, 2, 1, 4, 32, code>(g, s); return DPD_E_UNSUPPORTED;
        DPD_X3_ABL(1) DPD_X3_ABL(2) DPD_X3_ABL(3) DPD_X3_ABL(4) DPD_X3_ABL(5) DPD_X3_ABL(7) DPD_X3_ABL(8) DPD_X3_ABL(9) DPD_X3_ABL(12) DPD_X3_ABL(13) DPD_X3_ABL(15)
#undef DPD_X3_ABL
#endif
        default: return DPD_E_UNSUPPORTED;
    }
}

// C[M,N] (fp32) = epi( op(A) op(B) ) from bf16 planes.  a_fmt/b_fmt: 0 = RC (k contiguous), 1 = R8 (k = row index),
// 2 = RC plane of an operand whose k is its ROW index (both operands: A stored [K][M], B stored [K][N]; lda / ldb = row strides).
int gemm_x3(int np, int a_fmt, int b_fmt, int M, int N, int K, const uint16_t* A, int lda, long a_plane, const uint16_t* B,
            int ldb, long b_plane, float* C, int ldc, const float* bias, const float* gate, int epilogue, int tile,
            hipStream_t s, float* colsum, const X3Out* out, const X3Extra* ex, int split_k, void* ws,
            size_t ws_bytes, const uint16_t* gate16, int gate16_r8, void* red_cnt, int red_cnt_words) {
    const uint16_t* A2 = ex ? ex->A2 : nullptr;
    const uint16_t* B2 = ex ? ex->B2 : nullptr;
    float* C2 = ex ? ex->C2 : nullptr;
    const int M2 = (ex && ex->M2) ? ex->M2 : M;
    const bool three = ex && ex->A3;
    const bool uneven = A2 && (M2 != M || three);      // problems of different rows / a third problem: ring kernels, TN only
    if (A2 && (!B2 || !C2 || out || colsum || epilogue != EPI_NONE)) return DPD_E_UNSUPPORTED;
    if (three && (!A2 || !ex->B3 || !ex->C3 || ex->M3 <= 0)) return DPD_E_UNSUPPORTED;
    if (uneven && (a_fmt != 1 || b_fmt != 1 || tile < 1 || (tile > 5 && tile != 13) || (M2 & 7) || (three && (ex->M3 & 7))))
        return DPD_E_UNSUPPORTED;
    const int nprob = A2 ? (three ? 3 : 2) : 1;
    // split-K (deterministic slabs in `ws` + the reduce kernel of gemm_f32.hip): plain products only (the dW shapes: K = query rows
    // is long, M x N gives too few 128x128 tiles for 256 CUs)
    // red_cnt != NULL: the slices are reduced INSIDE the launch (inlaunch_reduce above; ring kernels, tiles 1-5); `ws` then holds the
    // raw accumulator slabs of every slice (tile-padded) and red_cnt [>= tiles] arrival words of 8 bytes
    int chunk = K;
    if (split_k > 1) {
        chunk = (((K + split_k - 1) / split_k) + 63) / 64 * 64;
        if (out || colsum || epilogue != EPI_NONE || !C || chunk * (split_k - 1) >= K) return DPD_E_UNSUPPORTED;
        if (!red_cnt && (uneven || !ws || (size_t)split_k * M * N * sizeof(float) * nprob > ws_bytes)) return DPD_E_WORKSPACE;
    } else {
        split_k = 1;
    }
    const bool planes_out = out && (out->rc || out->r8);
    if (!A || !B || (!C && !planes_out)) return DPD_E_NULL;
    if (planes_out && (out->np != np || (M & 7) || (N & 7) || (out->r8_rows & 7) || (out->rc && (out->ld_rc & 7))))
        return DPD_E_UNSUPPORTED;
    if (M <= 0 || N <= 0 || K <= 0) return DPD_E_DIM;
    if (np != 1 && np != 3) return DPD_E_UNSUPPORTED;
    if ((K % 32) || (N & 3) || (ldc & 3) || (lda & 7) || (ldb & 7)) return DPD_E_UNSUPPORTED;
    if (tile == 13 && (K % 64)) return DPD_E_UNSUPPORTED;   // BK = 64 kernels take whole 64-deep K-tiles
    if ((epilogue == EPI_BIAS || epilogue == EPI_BIAS_RELU) && !bias) return DPD_E_NULL;
    if (epilogue == EPI_GATE && !gate && !gate16) return DPD_E_NULL;
    if (epilogue < 0 || epilogue > 3) return DPD_E_UNSUPPORTED;
    if (a_fmt && b_fmt == 0) return DPD_E_UNSUPPORTED;   // (R8, RC) never occurs in the decoder
    if ((a_fmt == 2) != (b_fmt == 2)) return DPD_E_UNSUPPORTED;   // the transpose-read form exists for TN with both operands as RC planes
    if (a_fmt == 2 && ((M & 7) || (N & 7))) return DPD_E_UNSUPPORTED;
    X3Args g{};
    g.e.C = C; g.e.bias = bias; g.e.gate = gate; g.e.gate16 = gate ? nullptr : gate16; g.e.gate16_r8 = gate16_r8 && !(M & 7); g.e.colsum = colsum;
    g.e.M = M; g.e.N = N; g.e.K = K; g.e.ldc = ldc; g.e.epi = epilogue;
    g.e.split_k = 1; g.e.k_chunk = K; g.e.slab_stride = 0;
    g.A = A; g.B = B; g.a_plane = a_plane; g.b_plane = b_plane; g.lda = lda; g.ldb = ldb;
    g.A2 = A2; g.B2 = B2; g.C2 = C2;
    if (uneven) {      // TN on R8 planes: the row count of a problem is its A's k-group row length and (with K) its plane stride
        g.M2 = M2; g.lda2 = M2; g.a_plane2 = (long)K * M2;
        if (three) { g.A3 = ex->A3; g.B3 = ex->B3; g.C3 = ex->C3; g.M3 = ex->M3; g.lda3 = ex->M3; g.a_plane3 = (long)K * ex->M3; }
    }
    if (tile == 0) tile = 1;
    if (split_k > 1 && red_cnt) {
        int bm, bn;
        switch (tile) {
            case 1: case 2: bm = 128; bn = 128; break;
            case 3: bm = 64; bn = 128; break;
            case 4: bm = 128; bn = 64; break;
            case 5: bm = 64; bn = 64; break;
            default: return DPD_E_UNSUPPORTED;
        }
        const size_t tn = (size_t)((N + bn - 1) / bn);
        const size_t tiles = ((size_t)((M + bm - 1) / bm) + (A2 ? (M2 + bm - 1) / bm : 0) + (three ? (ex->M3 + bm - 1) / bm : 0)) * tn;
        if (!ws || tiles * split_k * bm * bn * sizeof(float) > ws_bytes) return DPD_E_WORKSPACE;
        if (tiles > (size_t)red_cnt_words) return DPD_E_WORKSPACE;
        static std::atomic<unsigned> generation{0x5eed0000u};
        g.e.split_k = split_k; g.e.k_chunk = chunk;          // C / C2 / ldc stay the real output: the reducing slice stores it
        g.red_slab = (float*)ws; g.red_cnt = (unsigned long long*)red_cnt; g.red_gen = ++generation;
        static const int sc1_mode = [] { const char* e = getenv("DPD_RED_SC1"); return e ? atoi(e) : 1; }();
        g.red_sc1 = sc1_mode;
    } else if (split_k > 1) {
        g.e.split_k = split_k; g.e.k_chunk = chunk; g.e.slab_stride = (long)M * N; g.e.ldc = N;
        g.e.C = (float*)ws; g.C2 = (float*)ws + (size_t)split_k * M * N;
    }
    if (planes_out) {
        g.out_rc = out->rc; g.out_r8 = out->r8; g.rc_plane = out->rc_plane; g.r8_plane = out->r8_plane;
        g.ld_rc = out->ld_rc; g.r8_rows = out->r8_rows; g.np_out = out->np;
    }
    struct ProfScope {
        bool on; hipStream_t s; double fl; int form;
        ~ProfScope() { prof_end(on, s, fl, form); }
    } prof_scope{prof_begin(s), s, 2.0 * N * K * ((double)M + (A2 ? M2 : 0) + (three ? ex->M3 : 0)), a_fmt ? 1 : 0};
    int rc;
    if (a_fmt == 2) {
        rc = np == 3 ? launch_x3_tile_tr<3>(tile, g, s) : launch_x3_tile_tr<1>(tile, g, s);
    } else if (np == 3) {
        if (!a_fmt && b_fmt) rc = launch_x3_tile<3, true, false>(tile, g, s);       // NN
        else if (!a_fmt && !b_fmt) rc = launch_x3_tile<3, true, true>(tile, g, s);  // NT
        else rc = launch_x3_tile<3, false, false>(tile, g, s);                      // TN
    } else {
        if (!a_fmt && b_fmt) rc = launch_x3_tile<1, true, false>(tile, g, s);
        else if (!a_fmt && !b_fmt) rc = launch_x3_tile<1, true, true>(tile, g, s);
        else rc = launch_x3_tile<1, false, false>(tile, g, s);
    }
    if (rc || split_k == 1 || red_cnt) return rc;
    if ((rc = splitk_reduce(g.e.C, split_k, (long)M * N, M, N, C, ldc, nullptr, nullptr, EPI_NONE, s))) return rc;
    if (A2) rc = splitk_reduce(g.C2, split_k, (long)M * N, M, N, C2, ldc, nullptr, nullptr, EPI_NONE, s);
    return rc;
}

// ---------------------------------------------------------------------------------------------------------
// fp32 [R, C] (row stride ld) -> np bf16 planes in RC and/or R8 layout.  R % 8 == 0, C % 8 == 0.
// Block = 256 threads over an 8-row x 256-column strip; HBM-bound (4 B read, 2*np B written per layout).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_planes_kernel(SplitJobs jobs) {
    int ji = 0;
    for (int t = 1; t < jobs.n; ++t)
        if ((int)blockIdx.x >= jobs.j[t].blk0) ji = t;
    const SplitJob& jb = jobs.j[ji];
    const int C = jb.C, ld = jb.ld, np = jb.np;
    const float* __restrict__ src = jb.src;
    const int blk = blockIdx.x - jb.blk0;
    const int strips = (C + 255) / 256;
    const int rg = blk / strips, c0 = (blk % strips) * 256;
    const int tid = threadIdx.x;
    if (jb.r8) {   // thread = one column: 8 rows -> one 16-B chunk per plane
        const int c = c0 + tid;
        if (c < C) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(8 * rg + j) * ld + c];
            uint4 w[3];
            split_chunk(v, w);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (q < np) *reinterpret_cast<uint4*>(jb.r8 + q * jb.r8_plane + ((size_t)rg * C + c) * 8) = w[q];
        }
    }
    if (jb.rc) {   // thread = 8 consecutive columns of one row
        const int r = 8 * rg + tid / 32, c = c0 + (tid % 32) * 8;
        if (c < C) {
            const float4 x0 = *reinterpret_cast<const float4*>(src + (size_t)r * ld + c);
            const float4 x1 = *reinterpret_cast<const float4*>(src + (size_t)r * ld + c + 4);
            const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            uint4 w[3];
            split_chunk(v, w);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (q < np) *reinterpret_cast<uint4*>(jb.rc + q * jb.rc_plane + (size_t)r * jb.ld_rc + c) = w[q];
        }
    }
}

static bool split_job_ok(const SplitJob& j) {
    return j.src && (j.rc || j.r8) && j.R > 0 && j.C > 0 && !(j.R & 7) && !(j.C & 7) && !(j.ld & 3) && (j.np == 1 || j.np == 3);
}

// up to 8 independent split jobs in ONE launch (the weight planes of a step)
int split_planes_multi(SplitJobs jobs, hipStream_t s) {
    if (jobs.n < 1 || jobs.n > 8) return DPD_E_DIM;
    int total = 0;
    for (int t = 0; t < jobs.n; ++t) {
        if (!split_job_ok(jobs.j[t])) return jobs.j[t].src ? DPD_E_UNSUPPORTED : DPD_E_NULL;
        jobs.j[t].blk0 = total;
        total += (jobs.j[t].R / 8) * ((jobs.j[t].C + 255) / 256);
    }
    DPD_LAUNCH(split_planes_kernel, dim3(total), dim3(256), 0, s, jobs);
    return (int)hipGetLastError();
}

int split_planes(const float* src, int R, int C, int ld, int np, uint16_t* rc, int ld_rc, long rc_plane, uint16_t* r8,
                 long r8_plane, hipStream_t s) {
    SplitJobs jobs{};
    jobs.n = 1;
    jobs.j[0] = SplitJob{src, rc, r8, rc_plane, r8_plane, R, C, ld, ld_rc, np, 0};
    return split_planes_multi(jobs, s);
}

}  // namespace dpd


// ---- C ABI (building blocks; the decoder entry points use them when dtype != 0) -----------------------------
extern "C" int dpd_split_planes(const float* src, int R, int C, int ld, int np, void* rc, int ld_rc, long rc_plane, void* r8,
                                long r8_plane, void* stream) {
    return dpd::split_planes(src, R, C, ld, np, (uint16_t*)rc, ld_rc, rc_plane, (uint16_t*)r8, r8_plane, (hipStream_t)stream);
}

extern "C" int dpd_gemm_planes(int np, int a_fmt, int b_fmt, int M, int N, int K, const void* A, int lda, long a_plane,
                               const void* B, int ldb, long b_plane, float* C, int ldc, const float* bias, const float* gate,
                               int epilogue, int tile, void* out_rc, void* out_r8, int r8_rows, void* stream) {
    dpd::X3Out o;
    o.rc = (uint16_t*)out_rc; o.r8 = (uint16_t*)out_r8; o.np = np; o.ld_rc = N; o.r8_rows = r8_rows;
    o.rc_plane = (long)M * N; o.r8_plane = (long)r8_rows * N;
    return dpd::gemm_x3(np, a_fmt, b_fmt, M, N, K, (const uint16_t*)A, lda, a_plane, (const uint16_t*)B, ldb, b_plane, C, ldc,
                        bias, gate, epilogue, tile, (hipStream_t)stream, nullptr, (out_rc || out_r8) ? &o : nullptr, nullptr, 1, nullptr, 0, nullptr, 0, nullptr, 0);
}
