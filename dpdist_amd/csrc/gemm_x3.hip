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
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "gemm_shared.h"

namespace dpd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct X3Args {
    GemmArgs e;           // epilogue view: C, bias, gate, colsum, M, N, K, ldc, epi, split (A/B/lda/ldb unused)
    const uint16_t* A;    // plane 0 of A
    const uint16_t* B;
    long a_plane, b_plane;   // elements between planes
    int lda, ldb;            // RC: row stride (elements); R8: entries per k-group row
    // optional plane outputs of C (np_out planes each), written by the LDS-staged epilogue:
    uint16_t* out_rc;        // RC planes [np_out][M][ld_rc]   (C is the k-contiguous operand of the next GEMM)
    uint16_t* out_r8;        // R8 planes [np_out][r8_rows/8][N][8], rows < r8_rows only (C as a k = row operand)
    long rc_plane, r8_plane;
    int ld_rc, r8_rows, np_out;
    // optional second problem of identical shape and layout (grouped launch): blocks [per_z, 2*per_z)
    const uint16_t* A2;
    const uint16_t* B2;
    float* C2;
    // round 4 (ring kernel only, plain products): the grouped problems may differ in M (rows of A^T / C), and there may be three of
    // them -- the three weight gradients of the bf16 step in ONE launch (dW1 2528 x 1024, dW2 and dW3 1024 x 1024, K = query rows):
    // each alone leaves 96-192 of the 256 CUs idle for the 27 us its K loop takes.  0 = same as problem 0.
    int M2, lda2;
    long a_plane2;
    const uint16_t* A3;
    const uint16_t* B3;
    float* C3;
    int M3, lda3;
    long a_plane3;
    // in-launch split-K (red_cnt != NULL; e.split_k slices of e.k_chunk per output tile): every slice parks its raw accumulators in
    // red_slab, the slice that arrives LAST at the tile's counter adds all slices in slice order and runs the normal epilogue on C
    float* red_slab;               // [tiles (x2 grouped)][split][BM*BN] floats, accumulator-register order (lane-linear 16-byte pieces)
    unsigned long long* red_cnt;   // [tiles (x2 grouped)] arrival words {generation : 32, arrivals : 32}
    unsigned red_gen;              // generation of this launch: a word of another generation (workspace garbage, an aborted launch) counts as 0
    int red_sc1;                   // != 0: slabs published by write-through (sc1) stores and read by sc1 loads, no release / acquire fence
};

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// chunk index inside an operand-plane image
template <bool KC, int BO, int CPR>
__device__ __forceinline__ int chunk_of(int o, int kg) {
    return KC ? o * CPR + (kg ^ ((o / (16 / CPR)) & (CPR - 1))) : kg * BO + o;
}

// ---- "RCT" operand images (round 3): an operand whose contraction index is its ROW index, read straight from its RC plane.
// The R8 planes exist only so that such an operand's MFMA fragment (8 consecutive k for one row/column) is one ds_read_b128; on
// gfx950 the LDS transpose read does the same from a row-major image: ds_read_b64_tr_b16 gives every lane of a 16-lane group four
// consecutive ROWS of its own column (measured semantics, tools/tr_read_probe.hip: output lane i, element j = element i % 4 of the
// 8-byte piece addressed by lane i/4 + 4j of the group).  Image [BK rows (k)][BO columns] bf16, lane-linear for the LDS-DMA (a 1-KiB
// piece = 64 / (BO/8) whole row segments); the 16-byte chunk c of row r sits in slot c ^ 2 (r & 3), chosen on the DMA's per-lane
// SOURCE address, so that the four rows a 16-lane group reads fall on different banks.  With it the activations, the pre-activation
// gradients and the gathered rows would need no R8 plane at all (63 MB less to write per bf16 step at B = 64).  MEASURED SLOWER and therefore
// opt-in only (a_fmt = b_fmt = 2 of dpd_gemm_planes, tested like every other form): dW1 at B = 64 40.2 -> 47.1 us, one dW2 32.5 -> 40.8 us,
// three planes 89.8 -> 96.4 / 54.1 -> 66.9 us (tools/tr_probe2.py) -- two LDS reads per fragment instead of one and 256-byte row
// segments instead of fully linear 1-KiB DMA pieces cost as much as the R8 planes do.
typedef short v4i16 __attribute__((ext_vector_type(4)));

template <int BO>
__device__ __forceinline__ int rct_chunk(int row, int c) {
    return row * (BO / 8) + (c ^ ((row & 3) << 1));
}

// fragment of the 32 columns starting at `o32` (multiple of 32) for the k16 step kb of a K-tile: lane (l31, half) <- rows 16 kb + 8 half + 0..7
template <int BO>
__device__ __forceinline__ bf16x8 rct_frag(const char* img, int o32, int kb, int lane) {
    const int s16 = lane & 15, grp = lane >> 4;
    const int col = o32 + 16 * (grp & 1) + 4 * (s16 & 3);                 // first column of the 8-byte piece this lane SUPPLIES
    const int row = 16 * kb + 8 * (grp >> 1) + (s16 >> 2);                // its row (first read); +4 for the second read
    typedef __attribute__((address_space(3))) v4i16* lp;
    const v4i16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)((unsigned)(uintptr_t)(lds_ptr_t)img + rct_chunk<BO>(row, col >> 3) * 16 + (col & 7) * 2));
    const v4i16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)((unsigned)(uintptr_t)(lds_ptr_t)img + rct_chunk<BO>(row + 4, col >> 3) * 16 + (col & 7) * 2));
    typedef short v8i16 __attribute__((ext_vector_type(8)));
    const v8i16 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// Epilogue shared by the plane GEMM kernels: fp32 store with bias / ReLU / gate / column sums, and -- when plane outputs are
// requested -- the finished tile staged through the (idle) LDS ring so that the next GEMMs find their operands as bf16 planes.
// NP = planes of the kernel = planes of its plane outputs (gemm_x3() checks it): compile time, so that the one-plane type converts
// each value once instead of running the three-plane split and dropping two thirds of it (the split was most of this epilogue's
// time: 11.4k cycles to write a 64 KB RC plane of a 256x128 tile against 9.1k for the 128 KB fp32 tile, tools/p8_stamps.py)
template <int BM, int BN, int NW, int TM, int TN, int NP>
__device__ __forceinline__ void x3_epilogue(const X3Args& g, f32x16 (&acc)[TM][TN], char* smem_x3, int grp, int z, int m0, int n0,
                                            int wm0, int wn0, int tid, int l31, int half) {
    const int M = g.e.M, N = g.e.N;
    if (!g.out_rc && !g.out_r8) {
        GemmArgs ge = g.e;
        if (grp == 1) { ge.C = g.C2; if (g.M2) ge.M = g.M2; }
        if (grp == 2) { ge.C = g.C3; ge.M = g.M3; }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) store_tile(ge, acc[i][j], z, m0 + wm0 + 32 * i, n0 + wn0 + 32 * j + l31, half);
        return;
    }
    // Plane outputs straight from the accumulator registers (round 3; the round-2 form staged the tile through the idle LDS ring as fp32
    // and re-read it in both chunk orientations: two barriers, 64 ds_write_b32 and 24 LDS reads per lane).  A lane of a 32x32 accumulator
    // tile holds ONE column and the rows (r & 3) + 8 (r >> 2) + 4 half, so after packing, dword pair g of a lane = rows 8g + 4 half + 0..3:
    //   R8 chunks (8 consecutive rows of one column): rows 8g .. 8g+3 sit in this lane's half, 8g+4 .. 8g+7 in lane + 32: one
    //   v_permlane32_swap per packed dword hands the lower half the chunks of the even row groups and the upper half those of the odd
    //   ones (semantics probed in tools/permlane_probe.hip) -> one 16-byte store per pair of row groups and plane, lanes contiguous.
    //   RC chunks (8 consecutive columns of one row) need a 16-bit transpose.  Interior tiles: every wave parks its packed dword pairs
    //   in a private 2.25-KiB strip of the idle LDS ring (4 ds_write_b64) and takes them back through the gfx950 transpose read
    //   (ds_read_b64_tr_b16: output lane i, element j of a 16-lane group = element i % 4 of the piece addressed by lane i/4 + 4j,
    //   tools/tr_read_probe.hip): lane s of a group addresses column 8 (s & 3) + (s >> 2) of one four-row group, so that output lane i
    //   receives columns 8 (i >> 2) + 0..3 of row i & 3, a second read (+4 columns) completes the 16-byte chunk -> 4 LDS writes, 4 LDS
    //   reads and 2 stores per tile and plane, no cross-lane VALU work (the all-VALU form -- in-quad DPP transpose, then lanes 4 apart
    //   trading row groups -- is ~100 VALU instructions per tile and plane; it stays below for the tiles that cross the matrix edge).
    //   Strip layout: piece (column c, four-row group rg) at ((36 rg + c) * 8 bytes: writes are lane-linear, the 16 pieces of a
    //   transpose read hit 16 different bank pairs and the neighbouring group (rg + 1, +288 bytes) the other 16.
    // Bit-identical to converting the stored fp32 tile (tests: test_gemm_planes_fused_outputs).
    constexpr int STRIP = 36 * 8 * 8;                      // bytes per wave and plane
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool interior = m0 + wm0 + 32 * TM <= M && n0 + wn0 + 32 * TN <= N && (!g.out_r8 || m0 + wm0 + 32 * TM <= g.r8_rows) &&
                          !(g.ld_rc & 7) && !(N & 7);
    if (g.out_rc) __builtin_amdgcn_s_barrier();            // every wave is done reading the ring (all DMA pieces were waited for in the K loop)
    if (interior) {
        const int lane = tid & 63, s16 = lane & 15, G = lane >> 4;
        const unsigned strip = (unsigned)(uintptr_t)(lds_ptr_t)smem_x3 + wave_id * (NP * STRIP);
        const unsigned wr_addr = strip + (half * 36 + l31) * 8;                                   // + 576 g (+ STRIP q)
        const unsigned rd_addr = strip + (G * 36 + 8 * (s16 & 3) + (s16 >> 2)) * 8;               // + 32 (second read) + 1152 p (+ STRIP q)
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(3))) u32x2* lds_u2;
        typedef __attribute__((address_space(3))) v4i16* lds_v4;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float v[16];
                const int grow0 = m0 + wm0 + 32 * i, gcol0 = n0 + wn0 + 32 * j;
                tile_values(g.e, acc[i][j], grow0, gcol0 + l31, half, v);
                put_tile(g.e, v, z, grow0, gcol0 + l31, half);
                unsigned pk[NP][8];                              // pk[q][2g + h] = rows 8g + 4 half + 2h, + 2h + 1 of plane q
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    unsigned lo[3], hi[3];
                    if (NP == 1) {
                        lo[0] = bf16_bits(v[r]);
                        hi[0] = bf16_bits(v[r + 1]);
                    } else {
                        split3(v[r], lo);
                        split3(v[r + 1], hi);
                    }
#pragma unroll
                    for (int q = 0; q < NP; ++q) pk[q][r >> 1] = lo[q] | (hi[q] << 16);
                }
                if (g.out_rc) {
#pragma unroll
                    for (int q = 0; q < NP; ++q)
#pragma unroll
                        for (int gi = 0; gi < 4; ++gi)
                            *(lds_u2)(uintptr_t)(wr_addr + q * STRIP + 576 * gi) = u32x2{pk[q][2 * gi], pk[q][2 * gi + 1]};
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int q = 0; q < NP; ++q)
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            const v4i16 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(uintptr_t)(rd_addr + q * STRIP + 1152 * p));
                            const v4i16 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(uintptr_t)(rd_addr + q * STRIP + 1152 * p + 32));
                            const uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
                            const int row = grow0 + 16 * p + 4 * G + (s16 & 3);
                            *reinterpret_cast<uint4*>(g.out_rc + q * g.rc_plane + (size_t)row * g.ld_rc + gcol0 + 8 * (s16 >> 2)) =
                                make_uint4(ua.x, ua.y, ub.x, ub.y);
                        }
                    asm volatile("" ::: "memory");
                }
                if (g.out_r8) {
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp)
#pragma unroll
                        for (int q = 0; q < NP; ++q) {
                            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[q][4 * gp], pk[q][4 * gp + 2], false, false);
                            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[q][4 * gp + 1], pk[q][4 * gp + 3], false, false);
                            const int rowg = grow0 + 8 * (2 * gp + half);
                            *reinterpret_cast<uint4*>(g.out_r8 + q * g.r8_plane + ((size_t)(rowg >> 3) * N + gcol0 + l31) * 8) =
                                make_uint4(s0[0], s1[0], s0[1], s1[1]);
                        }
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float v[16];
            const int grow0 = m0 + wm0 + 32 * i, gcol0 = n0 + wn0 + 32 * j;
            tile_values(g.e, acc[i][j], grow0, gcol0 + l31, half, v);
            put_tile(g.e, v, z, grow0, gcol0 + l31, half);
            unsigned pv[16][NP];                              // bf16 planes of the 16 values
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (NP == 1) {
                    pv[r][0] = bf16_bits(v[r]);
                } else {
                    unsigned p3[3];
                    split3(v[r], p3);
#pragma unroll
                    for (int q = 0; q < NP; ++q) pv[r][q] = p3[q];
                }
            }
            if (g.out_rc) {
                // after the in-quad transpose lane 4q+j holds columns 4q..4q+3 of row 8 gi + 4 half + j (8 bytes per plane); lanes 4 apart
                // (q even / odd) then trade row groups pairwise, so that every lane owns ONE 16-byte chunk (8 columns) per pair of row groups
                const int jq = l31 & 3, qodd = (l31 >> 2) & 1;
                const int col8 = gcol0 + ((l31 >> 3) << 3);
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    uint2 w[4];
#pragma unroll
                    for (int gi = 0; gi < 4; ++gi) {
                        float a[4] = {__uint_as_float(pv[4 * gi][q]), __uint_as_float(pv[4 * gi + 1][q]), __uint_as_float(pv[4 * gi + 2][q]),
                                      __uint_as_float(pv[4 * gi + 3][q])};     // (16-bit payloads moved as 32-bit lanes)
                        quad_transpose4(a, l31);
                        w[gi] = make_uint2(__float_as_uint(a[0]) | (__float_as_uint(a[1]) << 16), __float_as_uint(a[2]) | (__float_as_uint(a[3]) << 16));
                    }
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        // even-q lanes keep row group 2 gp and receive its upper four columns from lane + 4; odd-q lanes keep 2 gp + 1
                        // and receive its lower four columns from lane - 4
                        const uint2 give = qodd ? w[2 * gp] : w[2 * gp + 1], keep = qodd ? w[2 * gp + 1] : w[2 * gp];
                        const unsigned ux = (unsigned)__builtin_amdgcn_mov_dpp((int)give.x, 0x104, 0xf, 0xf, true);   // row_shl:4: from lane + 4
                        const unsigned uy = (unsigned)__builtin_amdgcn_mov_dpp((int)give.y, 0x104, 0xf, 0xf, true);
                        const unsigned dx = (unsigned)__builtin_amdgcn_mov_dpp((int)give.x, 0x114, 0xf, 0xf, true);   // row_shr:4: from lane - 4
                        const unsigned dy = (unsigned)__builtin_amdgcn_mov_dpp((int)give.y, 0x114, 0xf, 0xf, true);
                        const uint4 chunk = qodd ? make_uint4(dx, dy, keep.x, keep.y) : make_uint4(keep.x, keep.y, ux, uy);
                        const int row = grow0 + 8 * (2 * gp + qodd) + 4 * half + jq;
                        if (row < M && col8 < N) *reinterpret_cast<uint4*>(g.out_rc + q * g.rc_plane + (size_t)row * g.ld_rc + col8) = chunk;
                    }
                }
            }
            if (g.out_r8) {
                const int col = gcol0 + l31;
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    const int rowg = grow0 + 8 * (2 * gp + half);            // first row of the row group this lane ends up holding
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        const unsigned X0 = pv[8 * gp][q] | (pv[8 * gp + 1][q] << 16), X1 = pv[8 * gp + 2][q] | (pv[8 * gp + 3][q] << 16);
                        const unsigned Y0 = pv[8 * gp + 4][q] | (pv[8 * gp + 5][q] << 16), Y1 = pv[8 * gp + 6][q] | (pv[8 * gp + 7][q] << 16);
                        const auto s0 = __builtin_amdgcn_permlane32_swap(X0, Y0, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(X1, Y1, false, false);
                        if (rowg < g.r8_rows && col < N)
                            *reinterpret_cast<uint4*>(g.out_r8 + q * g.r8_plane + ((size_t)(rowg >> 3) * N + col) * 8) =
                                make_uint4(s0[0], s1[0], s0[1], s1[1]);
                    }
                }
            }
        }
}

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
template <int NP, bool AK, bool BKC, int WR, int WC, int TM, int TN, bool LATE_WAIT, int ABL = 0>
__global__ __launch_bounds__(64 * WR * WC) void gemm_p8_kernel(X3Args g) {
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
    extern __shared__ __attribute__((aligned(16))) char smem_x3[];

    const int tid = threadIdx.x;
    P8_STAMP(0);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wgrp = __builtin_amdgcn_readfirstlane(wave / (NW / 2));          // 0: waves 0..NW/2-1, 1: the rest
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave / WC) * 32 * TM, wn0 = (wave % WC) * 32 * TN;

    const int M = g.e.M, N = g.e.N;
    const int tilesM = (M + BM - 1) / BM, tilesN = (N + BN - 1) / BN;
    const int per_z = tilesM * tilesN;
    const int sid0 = xcd_remap(blockIdx.x, per_z * (g.A2 ? 2 : 1));
    const int grp = sid0 / per_z;
    const int t0 = sid0 % per_z;
    const uint16_t* gA = grp ? g.A2 : g.A;
    const uint16_t* gB = grp ? g.B2 : g.B;
    const int m0 = (t0 / tilesN) * BM, n0 = (t0 % tilesN) * BN;
    const int K = g.e.K;
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
        const uint16_t* base = isA ? gA + plane * g.a_plane : gB + plane * g.b_plane;
        const int ld = isA ? g.lda : g.ldb;
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
    issue(0, 0, I0{}, IP{});
    if (nt > 1) {
        issue(1, 1, I0{}, IP{});
        wait_vm<PPW>();
    } else {
        wait_vm<0>();
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
    x3_epilogue<BM, BN, NW, TM, TN, NP>(g, acc, smem_x3, grp, 0, m0, n0, wm0, wn0, tid, l31, half);
    P8_STAMP(4);
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

// tile codes: 1 = 128x128 (4 waves of 64x64), 2 = 128x128 (8 waves of 64x32), 3 = 64x128, 4 = 128x64, 5 = 64x64
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
        case 6: return launch_x3<NP, AK, BKC, 2, 2, 2, 2, 6, 16>(g, s);   // BK = 16: finer, deeper ring
        case 7: return launch_x3<NP, AK, BKC, 2, 4, 2, 1, 6, 16>(g, s);
        // BK = 64 (one plane only: a 3-plane stage would not fit the LDS): 4x fewer barriers per MFMA
        case 8: if (NP == 1) return launch_x3<1, AK, BKC, 4, 2, 2, 2, 3, 64>(g, s); return DPD_E_UNSUPPORTED;   // 256x128, 8 waves of 64x64, 144 KiB
        case 9: if (NP == 1) return launch_x3<1, AK, BKC, 2, 4, 2, 1, 4, 64>(g, s); return DPD_E_UNSUPPORTED;   // 128x128, 8 waves of 64x32, 128 KiB
        case 10: if (NP == 1) return launch_x3<1, AK, BKC, 2, 2, 2, 2, 4, 64>(g, s); return DPD_E_UNSUPPORTED;  // 128x128, 4 waves of 64x64
        case 11: if (NP == 1) return launch_x3<1, AK, BKC, 2, 4, 2, 2, 3, 64>(g, s); return DPD_E_UNSUPPORTED;  // 128x256, 8 waves of 64x64
        case 12: if (NP == 1) return launch_x3<1, AK, BKC, 2, 2, 1, 2, 4, 64>(g, s); return DPD_E_UNSUPPORTED;  // 64x128, 4 waves of 32x64
        // 192x128 (round 4): the tile with the smallest BM + BN (= LDS fill bytes per flop) that covers dW1 + dW2 + dW3 in at most one
        // workgroup per CU (112 + 48 + 48 = 208 tiles): 8 waves of 96x32, BK = 64, 3 stages x 40 KiB
        case 13: if (NP == 1) return launch_x3<1, AK, BKC, 2, 4, 3, 1, 3, 64>(g, s); return DPD_E_UNSUPPORTED;
        case 14: if (NP == 1) return launch_x3<1, AK, BKC, 4, 2, 1, 3, 3, 64>(g, s); return DPD_E_UNSUPPORTED;  // 128x192, 8 waves of 32x96
        // the same tiles on FOUR waves of 96x64 / 64x96 (one per SIMD): 5 fragment reads per 6 MFMAs instead of 4 per 3 -- 37 % less LDS read traffic.
        // Measured SLOWER in the grouped weight-gradient launch (0.2877 vs 0.2792 ms per bf16 step at B = 64): the second wave per SIMD is worth more
        // than the fragment traffic; kept selectable (dpd_set_gemm_plan(33, 15 | 16, 1)) and tested
        case 15: if (NP == 1) return launch_x3<1, AK, BKC, 2, 2, 3, 2, 3, 64>(g, s); return DPD_E_UNSUPPORTED;  // 192x128, 4 waves of 96x64
        case 16: if (NP == 1) return launch_x3<1, AK, BKC, 2, 2, 2, 3, 3, 64>(g, s); return DPD_E_UNSUPPORTED;  // 128x192, 4 waves of 64x96
        // phase-staggered kernels (gemm_p8_kernel; K % 32 == 0, no split-K): one plane at BK = 64, three planes at BK = 32
        case 20: if (NP == 1 && g.e.split_k == 1) return launch_p8<1, AK, BKC, 4, 2, 2, 2, false>(g, s); return DPD_E_UNSUPPORTED;   // 256x128, 8 waves of 64x64
        case 21: if (NP == 1 && g.e.split_k == 1) return launch_p8<1, AK, BKC, 4, 2, 2, 2, true>(g, s); return DPD_E_UNSUPPORTED;    // ... group 0 waits after its MFMAs
        case 22: if (NP == 1 && g.e.split_k == 1) return launch_p8<1, AK, BKC, 2, 4, 2, 2, false>(g, s); return DPD_E_UNSUPPORTED;   // 128x256, 8 waves of 64x64
        case 23: if (NP == 1 && g.e.split_k == 1) return launch_p8<1, AK, BKC, 4, 2, 1, 2, false>(g, s); return DPD_E_UNSUPPORTED;   // 128x128, 8 waves of 32x64
        case 24: if (NP == 3 && g.e.split_k == 1) return launch_p8<3, AK, BKC, 4, 2, 1, 2, true>(g, s); return DPD_E_UNSUPPORTED;    // 128x128, 8 waves of 32x64, 3 planes
        case 25: if (NP == 3 && g.e.split_k == 1) return launch_p8<3, AK, BKC, 2, 4, 2, 1, true>(g, s); return DPD_E_UNSUPPORTED;    // 128x128, 8 waves of 64x32, 3 planes
        case 26: if (NP == 3 && g.e.split_k == 1) return launch_p8<3, AK, BKC, 4, 2, 1, 2, false>(g, s); return DPD_E_UNSUPPORTED;   // 24 with both groups waiting before B1
#ifdef DPD_ABLATIONS
#define DPD_P8_ABL(code) case 200 + code: if (NP == 1) return launch_p8<1, AK, BKC, 4, 2, 2, 2, true, code>(g, s); return DPD_E_UNSUPPORTED;
        DPD_P8_ABL(32) DPD_P8_ABL(1) DPD_P8_ABL(2) DPD_P8_ABL(3) DPD_P8_ABL(4) DPD_P8_ABL(5) DPD_P8_ABL(7) DPD_P8_ABL(8) DPD_P8_ABL(16) DPD_P8_ABL(24)
#undef DPD_P8_ABL
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
    if (uneven && (a_fmt != 1 || b_fmt != 1 || tile < 1 || (tile > 5 && (tile < 13 || tile > 16)) || (M2 & 7) || (three && (ex->M3 & 7))))
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
    if (tile >= 8 && tile <= 16 && (K % 64)) return DPD_E_UNSUPPORTED;   // BK = 64 kernels take whole 64-deep K-tiles
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
        bool on; hipStream_t s; double fl;
        ~ProfScope() { prof_end(on, s, fl); }
    } prof_scope{prof_begin(s), s, 2.0 * N * K * ((double)M + (A2 ? M2 : 0) + (three ? ex->M3 : 0))};
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

#ifdef DPD_ABLATIONS
extern "C" int dpd_debug_p8_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(dpd::g_p8_stamps), sizeof(unsigned long long) * 1024 * 8, 0, hipMemcpyDeviceToHost);
}
#endif

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
