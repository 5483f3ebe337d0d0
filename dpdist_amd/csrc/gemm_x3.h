// Pieces shared by the plane GEMM kernels: the LDS-DMA ring kernel (gemm_x3.hip) and the phase-staggered / chained kernels (gemm_p8.hip).
#pragma once
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
// SC1 (persistent chained launches, gemm_chain_kernel below): the plane outputs leave as write-through (sc1) buffer stores, so that a workgroup
// on ANY XCD finds them after the producer's drained flag (cdna_hip_programming.md Guideline 16, recipe R1) -- interior tiles only (the
// launcher admits only shapes whose tiles are all interior)
template <int BM, int BN, int NW, int TM, int TN, int NP, bool SC1 = false>
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
    // the R8 planes cover the rows < r8_rows only (forward: the half of the rows that carries gradient): a wave tile wholly beyond them simply
    // has no R8 output (round 5; before, such tiles fell to the all-VALU edge path below -- half of the forward's tiles; same bits)
    const bool r8_here = g.out_r8 && m0 + wm0 < g.r8_rows;
    const bool interior = m0 + wm0 + 32 * TM <= M && n0 + wn0 + 32 * TN <= N && (!r8_here || m0 + wm0 + 32 * TM <= g.r8_rows) &&
                          !(g.ld_rc & 7) && !(N & 7);
    if (g.out_rc) __builtin_amdgcn_s_barrier();            // every wave is done reading the ring (all DMA pieces were waited for in the K loop)
    if (interior) {
        __amdgpu_buffer_rsrc_t rs_rc, rs_r8;
        if (SC1) {
            rs_rc = __builtin_amdgcn_make_buffer_rsrc((void*)g.out_rc, 0, g.out_rc ? (int)((size_t)NP * g.rc_plane * 2) : 0, 0x00020000);
            rs_r8 = __builtin_amdgcn_make_buffer_rsrc((void*)g.out_r8, 0, g.out_r8 ? (int)((size_t)NP * g.r8_plane * 2) : 0, 0x00020000);
        }
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
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
                            if (SC1)
                                __builtin_amdgcn_raw_buffer_store_b128(u4v{ua.x, ua.y, ub.x, ub.y}, rs_rc,
                                                                       (unsigned)((q * g.rc_plane + (size_t)row * g.ld_rc + gcol0 + 8 * (s16 >> 2)) * 2), 0, 16);
                            else
                                *reinterpret_cast<uint4*>(g.out_rc + q * g.rc_plane + (size_t)row * g.ld_rc + gcol0 + 8 * (s16 >> 2)) =
                                    make_uint4(ua.x, ua.y, ub.x, ub.y);
                        }
                    asm volatile("" ::: "memory");
                }
                if (r8_here) {
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp)
#pragma unroll
                        for (int q = 0; q < NP; ++q) {
                            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[q][4 * gp], pk[q][4 * gp + 2], false, false);
                            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[q][4 * gp + 1], pk[q][4 * gp + 3], false, false);
                            const int rowg = grow0 + 8 * (2 * gp + half);
                            if (SC1)
                                __builtin_amdgcn_raw_buffer_store_b128(u4v{s0[0], s1[0], s0[1], s1[1]}, rs_r8,
                                                                       (unsigned)((q * g.r8_plane + ((size_t)(rowg >> 3) * N + gcol0 + l31) * 8) * 2), 0, 16);
                            else
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

// phase-staggered kernels (gemm_p8.hip): tile codes 20..26 of gemm_x3 (and the 200 + code ablations); DPD_E_UNSUPPORTED for anything else
int launch_p8_code(int np, bool ak, bool bkc, int tile, const X3Args& g, hipStream_t s);

}  // namespace dpd
