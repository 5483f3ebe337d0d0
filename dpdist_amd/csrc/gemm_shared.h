// Pieces shared by the fp32 MFMA GEMM (gemm_f32.hip) and the split-bf16 MFMA GEMM (gemm_x3.hip).
#pragma once
#include "common.h"

namespace dpd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_RELU = 2, EPI_GATE = 3 };

// TF-form Adam on one element (tf.train.AdamOptimizer, epsilon-hat form): ONE definition for the optimizer kernels (loss_adam.hip)
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float lr_t, float b1, float b2, float eps) {
    m = b1 * m + (1.0f - b1) * g;
    v = b2 * v + (1.0f - b2) * g * g;
    p = p - lr_t * m / (sqrtf(v) + eps);
}

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const float* gate;
    const uint16_t* gate16;   // alternative to `gate` (plane GEMMs): the gating activation as a bf16 RC plane [M][ldc]; > 0 <=> non-zero, sign clear
    int gate16_r8;            // ... or (non-zero) as its R8 plane [M/8][N][8]: a lane's four rows 8g + 4 half + 0..3 are ONE 8-byte load
    float* colsum;    // optional [N]: += column sums of the stored values (atomics; caller zeroes it)
    // deterministic bias gradients without atomics or an extra launch: a GEMM whose rows are data rows stores, per 32-row block,
    // the column sums of what it writes (colsum_part [ceil(M/32)][N]); a LATER GEMM's first-row-block waves add those partials
    // in a fixed order into colsum_b [N] (reading colsum_part_in, nparts blocks; *_2: the grouped second problem)
    float* colsum_part;
    const float* colsum_part_in;
    const float* colsum_part_in2;
    float* colsum_b;
    float* colsum_b2;
    int colsum_nparts;
    const float* A2;  // optional second problem of identical shape (grouped launch): blocks [per_z, 2*per_z)
    const float* B2;
    float* C2;
    int M, N, K;
    int lda, ldb, ldc;
    int epi;
    int split_k;      // >= 1
    // "tail split" (register-streamed kernels, split_k == 1, no epilogue): output tiles [0, tail_first) run their whole K range; the
    // tiles of the partial last round, [tail_first, tiles), are cut into tail_split K pieces of tail_chunk so that every CU ends
    // at the same time; piece 0 stores to C, piece z > 0 to tail_slab + (z-1)*M*N, a small kernel adds the slabs (fixed order)
    int tail_first, tail_split, tail_chunk;
    float* tail_slab;
    int k_chunk;      // K range per split (multiple of BK)
    long slab_stride; // floats between split-K slabs (0 when split_k == 1)
};

// host-side bundle of the two-step bias-gradient pointers (see GemmArgs::colsum_part)
struct ColsumTwoStep {
    float* part_out = nullptr;         // step 1: [ceil(M/32)][N] partial column sums of the values this GEMM stores
    const float* part_in = nullptr;    // step 2: partials of an earlier GEMM, `nparts` blocks of N floats ...
    const float* part_in2 = nullptr;
    float* out = nullptr;              // ... summed in block order into out [N] by this GEMM's first-row-block waves
    float* out2 = nullptr;
    int nparts = 0;
};

// Epilogue of one 32x32 MFMA tile: acc[r] is C[row0 + (r&3) + 8*(r>>2) + 4*half][col].  The gate / bias values are
// fetched up front with clamped (always valid) addresses so that the 16 loads are in flight together instead of
// one dependent L2 round trip per row.  tile_values() applies bias / ReLU / gate, put_tile() stores fp32 + column sums.
__device__ __forceinline__ void tile_values(const GemmArgs& g, const f32x16& acc, int row0, int col_in, int half, float (&v)[16]) {
    const int col = col_in < g.N ? col_in : g.N - 1;      // clamp instead of returning: all 64 lanes stay converged
    const int epi = g.epi;
    const float bv = (epi == EPI_BIAS || epi == EPI_BIAS_RELU) ? g.bias[col] : 0.f;
    float gv[16];
    if (epi == EPI_GATE) {
        if (g.gate16 && g.gate16_r8) {   // 4 loads of 8 bytes per tile instead of 16 of 2 (the load ISSUE is what costs: ~16 cycles each)
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                const int rg = min((row0 >> 3) + gi, (g.M >> 3) - 1);
                const uint2 b = *reinterpret_cast<const uint2*>(g.gate16 + ((size_t)rg * g.N + col) * 8 + 4 * half);
                const unsigned e[4] = {b.x & 0xffffu, b.x >> 16, b.y & 0xffffu, b.y >> 16};
#pragma unroll
                for (int k = 0; k < 4; ++k) gv[4 * gi + k] = ((e[k] & 0x7fffu) && !(e[k] >> 15)) ? 1.f : 0.f;
            }
        } else if (g.gate16) {      // bf16 rounding keeps the sign and never turns a positive fp32 into zero (same exponent range)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = min(row0 + (r & 3) + 8 * (r >> 2) + 4 * half, g.M - 1);
                const unsigned b = g.gate16[(size_t)row * g.ldc + col];
                gv[r] = ((b & 0x7fffu) && !(b >> 15)) ? 1.f : 0.f;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = min(row0 + (r & 3) + 8 * (r >> 2) + 4 * half, g.M - 1);
                gv[r] = g.gate[(size_t)row * g.ldc + col];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float x = acc[r] + bv;
        if (epi == EPI_BIAS_RELU) x = fmaxf(x, 0.f);
        if (epi == EPI_GATE) x = (gv[r] > 0.f) ? x : 0.f;
        v[r] = x;
    }
}

// 4x4 transpose inside every quad of lanes (two DPP butterfly stages): in: a[i] on lane j = E[i][j]; out: a[c] on lane j = E[j][c].
__device__ __forceinline__ void quad_transpose4(float (&a)[4], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    // stage 1: register bit 0 <-> lane bit 0 (partner = lane ^ 1, quad_perm [1,0,3,2])
    {
        const float x0 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a[0]), 0xB1, 0xf, 0xf, true));
        const float x1 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a[1]), 0xB1, 0xf, 0xf, true));
        const float x2 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a[2]), 0xB1, 0xf, 0xf, true));
        const float x3 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a[3]), 0xB1, 0xf, 0xf, true));
        const float n0 = b0 ? x1 : a[0], n1 = b0 ? a[1] : x0, n2 = b0 ? x3 : a[2], n3 = b0 ? a[3] : x2;
        a[0] = n0; a[1] = n1; a[2] = n2; a[3] = n3;
    }
    // stage 2: register bit 1 <-> lane bit 1 (partner = lane ^ 2, quad_perm [2,3,0,1])
    {
        const float x0 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a[0]), 0x4E, 0xf, 0xf, true));
        const float x1 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a[1]), 0x4E, 0xf, 0xf, true));
        const float x2 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a[2]), 0x4E, 0xf, 0xf, true));
        const float x3 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a[3]), 0x4E, 0xf, 0xf, true));
        const float n0 = b1 ? x2 : a[0], n1 = b1 ? x3 : a[1], n2 = b1 ? a[2] : x0, n3 = b1 ? a[3] : x1;
        a[0] = n0; a[1] = n1; a[2] = n2; a[3] = n3;
    }
}

// Store of one 32x32 MFMA tile.  A row-per-instruction dword store (32 lanes x 4 B of one row, 16 instructions per tile) is
// store-ISSUE bound on gfx950: ~25 cycles per wave-instruction per CU whatever its width, i.e. ~10 B/clk/CU -- measured with
// s_memtime stamps (tools/p8_stamps.py) at 12.6k cycles for a 256x128 fp32 tile, 5.5 us of a 12.8 us launch.  So the tile is
// transposed inside each quad of lanes (lane 4q+j then holds the four consecutive columns 4q..4q+3 of row 8g + 4 half + j) and
// leaves as four 16-byte stores per lane: the same 128-byte row segments, a quarter of the instructions.  Needs a 16-byte
// aligned C, ldc % 4 == 0, N % 4 == 0 and a tile that starts on a multiple of four columns; anything else keeps the dword form.
__device__ __forceinline__ void put_tile(const GemmArgs& g, float (&v)[16], int z, int row0, int col_in, int half) {
    const bool col_ok = col_in < g.N;
    const int col = col_ok ? col_in : g.N - 1;
    float* Cz = g.C ? g.C + (size_t)z * g.slab_stride : nullptr;
    float cs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < g.M && col_ok) cs += v[r];
    }
    if (Cz) {
        const int l31 = threadIdx.x & 31;
        const int col0 = __builtin_amdgcn_readfirstlane(col_in - l31);   // first column of the tile (wave-uniform)
        const bool wide = !((g.ldc | g.N | col0) & 3) && !((uintptr_t)Cz & 15);
        if (wide) {
            const int cq = col0 + (l31 & ~3), j = l31 & 3;
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                float a[4] = {v[4 * gi], v[4 * gi + 1], v[4 * gi + 2], v[4 * gi + 3]};
                quad_transpose4(a, l31);
                const int row = row0 + 8 * gi + 4 * half + j;
                if (row < g.M && cq < g.N) *reinterpret_cast<float4*>(Cz + (size_t)row * g.ldc + cq) = make_float4(a[0], a[1], a[2], a[3]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < g.M && col_ok) Cz[(size_t)row * g.ldc + col] = v[r];
            }
        }
    }
    if (g.colsum || g.colsum_part) {   // bias gradient fused into the dH GEMM: 32-row partial per wave
        cs += __shfl_xor(cs, 32, 64);
        if (half == 0 && col_ok) {
            if (g.colsum) atomicAdd(g.colsum + col, cs);                            // one atomic per column (order-dependent round-off)
            else g.colsum_part[(size_t)(row0 >> 5) * g.N + col] = cs;              // or a plain store for the deterministic two-step form
        }
    }
}

__device__ __forceinline__ void store_tile(const GemmArgs& g, const f32x16& acc, int z, int row0, int col_in, int half) {
    float v[16];
    tile_values(g, acc, row0, col_in, half, v);
    put_tile(g, v, z, row0, col_in, half);
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// One 1-KiB LDS-DMA piece: LDS[dst + lane*16] <- 16 bytes at this lane's source address.
// Inline asm on purpose: with the builtin, hipcc knows an LDS write is pending and drains vmcnt(0) before the next
// ds_read, which serialises the ring.  Here the pieces are invisible to its waitcnt bookkeeping and are retired by the
// counted s_waitcnt vmcnt(N) in the K-loop (every piece is one VM_CNT event).  M0 is written in the same statement
// that uses it and restored afterwards.
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

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// ---- fp32 -> bf16 planes (hi, mid, lo): shared by gemm_x3.hip and the producers that write planes directly ----
// hipcc lowers the float -> __bf16 cast to the gfx950 hardware conversion (round to nearest even, NaN stays NaN)
__device__ __forceinline__ unsigned bf16_bits(float x) {
    const __bf16 h = (__bf16)x;
    return (unsigned)__builtin_bit_cast(unsigned short, h);
}

__device__ __forceinline__ void split3(float a, unsigned (&p)[3]) {
    p[0] = bf16_bits(a);
    const float hi = __uint_as_float(p[0] << 16);
    const bool fin = (__float_as_uint(a) & 0x7f800000u) != 0x7f800000u;   // inf/NaN live in the hi plane only
    const float r1 = fin ? a - hi : 0.f;
    p[1] = bf16_bits(r1);
    const float r2 = r1 - __uint_as_float(p[1] << 16);
    p[2] = bf16_bits(r2);
}

// 8 fp32 values -> one 16-byte chunk per plane
__device__ __forceinline__ void split_chunk(const float (&v)[8], uint4 (&w)[3]) {
    unsigned u[3][4] = {};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        unsigned p[3];
        split3(v[j], p);
#pragma unroll
        for (int q = 0; q < 3; ++q) u[q][j >> 1] |= p[q] << (16 * (j & 1));
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) w[q] = make_uint4(u[q][0], u[q][1], u[q][2], u[q][3]);
}

// optional bf16-plane outputs of a gemm_x3 result (gemm_x3.hip): RC planes [np][M][ld_rc], R8 planes
// [np][r8_rows/8][N][8] for the rows < r8_rows
// out[m,n] = epi( sum_z slabs[z][m,n] ), fixed order (gemm_f32.hip); slabs are dense [M,N], N % 4 == 0
int splitk_reduce(const float* slabs, int split_k, long slab_stride, int M, int N, float* C, int ldc, const float* bias,
                  const float* gate, int epi, hipStream_t s);

struct X3Out {
    uint16_t* rc = nullptr;
    uint16_t* r8 = nullptr;
    long rc_plane = 0, r8_plane = 0;
    int ld_rc = 0, r8_rows = 0, np = 0;
};

// further problems of a grouped plain plane-GEMM launch (gemm_x3.hip): same N, K, B layout and tile; own operands, output and rows.
// M2 / M3 = 0: the rows of problem 0.  Problems with other rows than problem 0, and a third problem, run on the ring kernels only.
struct X3Extra {
    const uint16_t* A2 = nullptr;
    const uint16_t* B2 = nullptr;
    float* C2 = nullptr;
    int M2 = 0;
    const uint16_t* A3 = nullptr;
    const uint16_t* B3 = nullptr;
    float* C3 = nullptr;
    int M3 = 0;
};

struct SplitJob {
    const float* src;
    uint16_t* rc;
    uint16_t* r8;
    long rc_plane, r8_plane;
    int R, C, ld, ld_rc, np, blk0;   // blk0 = first block of this job in a multi-job launch
};
struct SplitJobs {
    int n;
    SplitJob j[8];
};


// in-stream GEMM profiler (gemm_f32.hip): event pair around one GEMM (kernel + split-K reduce)
bool prof_begin(hipStream_t s);
// form: 0 = the contraction runs along A's rows (NN / NT: the forward and data-gradient products), 1 = TN (weight gradients)
void prof_end(bool on, hipStream_t s, double flops, int form = 0);

}  // namespace dpd
