// Query -> voxel lookup and K^3 local-window gather (forward + backward).
//
// Replaces, without ever materialising the [C, m^3, k^3*20] window tensor (164 MB per cloud set at B=32):
//   local_z_3d                               utils/dpdist_util.py:911-930  (tf.extract_volume_patches, SAME)
//   get_pc_grid_binary_mask_from_centers     utils/dpdist_util.py:459-492  (half-open cell test, argmax)
//   get_emb_and_concat                       utils/dpdist_util.py:434-457  (gather_nd of centre-relative xyz + window)
//
// HBM layout: X [Q, KP] fp32, row r = [ window (k^3 neighbours x 20 channels, channel fastest, neighbour order
// (d_axis0, d_axis1, d_axis2) like extract_volume_patches) | q - centre (3) | zero pad ].  Everything is float4
// aligned (20 channels = 5 float4; KP*4 bytes is a multiple of 16).  HBM-bound: algorithmic bytes per query row
// = KP*4 written (+ 12 read; fv[c] = 40 KB per cloud is L2 resident and shared by the cloud's 64 rows).
#include <cstdlib>

#include "gemm_shared.h"
#include "patch_rows_bwd.h"

namespace dpd {

constexpr int kF = DPD_FV_CHANNELS;

// first cell i with  q > c_i - g  &&  q <= c_i + g  (float32, exactly the reference's two comparisons), else -1
__device__ __forceinline__ int cell_of(const GridAxis& ax, int m, float q) {
    int r = -1;
    for (int i = m - 1; i >= 0; --i) {
        const float lo = ax.c[i] - ax.half, hi = ax.c[i] + ax.half;
        if (q > lo && q <= hi) r = i;
    }
    return r;
}

// L2 scale of a cloud's Fisher vectors from the per-slice sums of squares the encoder left (mfv3d.hip: MfvFuse::ssq): the
// slices are added in order, then x * rsqrt(max(sum x^2, 1e-12)) (tf.nn.l2_normalize, dpdist_util.py:124-126).
__device__ __forceinline__ float fv_scale(const float* __restrict__ ssq, int nsl, int c, int ch) {
    float ss = 0.f;
    for (int sl = 0; sl < nsl; ++sl) ss += ssq[((size_t)c * nsl + sl) * kF + ch];
    return 1.0f / sqrtf(fmaxf(ss, 1e-12f));
}

__global__ __launch_bounds__(256) void patch_rows_fwd_kernel(const float* __restrict__ q, const float* __restrict__ fv,
                                                              float* __restrict__ X, float* __restrict__ mask,
                                                              int32_t* __restrict__ vox, int N, int m, int k, int KP,
                                                              GridAxis ax, int Q, const float* __restrict__ ssq, int nsl) {
    __shared__ __attribute__((aligned(16))) float s_sc[2][kF];
    const int half = threadIdx.x >> 7;
    const int r = blockIdx.x * 2 + half, tid = threadIdx.x & 127;   // two rows per workgroup, 128 threads each
    if (ssq) {      // fv is only power-normalised: this kernel applies the per-channel L2 scale of the row's cloud
        if (tid < kF && r < Q) s_sc[half][tid] = fv_scale(ssq, nsl, r / N, tid);
        __syncthreads();
    }
    if (r >= Q) return;
    const int c = r / N;
    const int G = m * m * m, h = (k - 1) / 2;
    const float qx = q[(size_t)r * 3], qy = q[(size_t)r * 3 + 1], qz = q[(size_t)r * 3 + 2];
    int ix = cell_of(ax, m, qx), iy = cell_of(ax, m, qy), iz = cell_of(ax, m, qz);
    const bool valid = (ix >= 0) && (iy >= 0) && (iz >= 0);
    if (!valid) { ix = 0; iy = 0; iz = 0; }   // argmax of an all-zero row is index 0 (:490); output is masked anyway
    const int E4 = k * k * k * (kF / 4);
    const float* fvc = fv + (size_t)c * G * kF;
    float* xr = X + (size_t)r * KP;
    for (int j = tid; j < E4; j += 128) {
        const int nb = j / 5, part = j % 5;
        const int d0 = nb / (k * k), d1 = (nb / k) % k, d2 = nb % k;
        const int g0 = iy + d0 - h, g1 = ix + d1 - h, g2 = iz + d2 - h;   // grid axes are (y, x, z), slowest first
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)g0 < (unsigned)m && (unsigned)g1 < (unsigned)m && (unsigned)g2 < (unsigned)m)
            v = *reinterpret_cast<const float4*>(fvc + (size_t)((g0 * m + g1) * m + g2) * kF + part * 4);
        if (ssq) {
            const float4 sc = *reinterpret_cast<const float4*>(&s_sc[half][part * 4]);
            v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w;
        }
        *reinterpret_cast<float4*>(xr + j * 4) = v;
    }
    const int E = E4 * 4;
    if (tid == 0) {
        xr[E + 0] = qx - ax.c[ix];   // point_cloud - Centers (:491), centre = (l[ix], l[iy], l[iz])
        xr[E + 1] = qy - ax.c[iy];
        xr[E + 2] = qz - ax.c[iz];
        for (int e = E + 3; e < KP; ++e) xr[e] = 0.f;
        mask[r] = valid ? 1.f : 0.f;
        vox[r] = (iy * m + ix) * m + iz;
    }
}

// Forward with operand-plane outputs for the bf16-matrix-core decoder (gemm_x3.hip): RC planes (k contiguous) for all rows and, for the
// rows that carry gradient (< r8_rows), R8 planes (chunks of 8 rows); fp32 X only if requested.
struct RowInfo {
    int ix, iy, iz, cloud;
    float dx, dy, dz;
};

// The round-1 form (a thread gathers one float4 unit of 8 rows and writes both orientations: B = 64: 31 us for 62 MB, no faster with one
// plane than with three) paid five divisions by the RUN-TIME window side per gathered float4 (no integer divider on gfx950: ~35 VALU
// instructions each) and wrote R8 planes as 16-byte pieces 64 bytes apart; removed in round 6.  Here a workgroup owns the 8 rows of one row group:
//   * the window geometry of a float4 unit (offset from the row's own voxel, the three neighbour displacements) is an LDS table
//     built once per workgroup instead of once per gather;
//   * pass A: wave w owns row w, its lanes walk the row's 8-column groups (no index division): two float4 gathers, convert, one
//     uint4 per plane to the RC plane (lanes contiguous: 1 KiB per wave-instruction) and into an LDS image [plane][8 rows][KP];
//   * pass B (rows that carry gradient): work item = one column: its 8 rows from the LDS image, one uint4 per plane to the R8 plane
//     (lanes contiguous).
// 31.2 -> 24 us (one plane, B = 64), 30.8 -> 24.5 us (three planes, B = 32).  What is left is the L2: the 64 rows of a cloud re-read its
// 40 KB of Fisher vectors as 83 MB of scattered 16-byte gathers per launch.  Measured and dropped: the cloud's Fisher vectors staged
// in LDS with several row groups per workgroup (30.8 us: one 8-wave workgroup per CU cannot hide its own barriers), five work items
// per lane with all gathers in flight before the first store (26.6 us), unit pairs of 8 rows per thread (39 us at 180 VGPRs).
#ifdef DPD_ABLATIONS
__device__ unsigned long long g_pr_stamps[1024 * 8];       // s_memtime milestones of thread 0 of every workgroup (tools/gather_stamps.py)
#define PR_STAMP(i) do { if (threadIdx.x == 0) g_pr_stamps[(blockIdx.x & 1023) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PR_STAMP(i) do { } while (0)
#endif
template <int NP>
__global__ __launch_bounds__(512) void patch_rows_planes3_kernel(const float* __restrict__ q, const float* __restrict__ fv,
                                                                 float* __restrict__ X, float* __restrict__ mask,
                                                                 int32_t* __restrict__ vox, int Q, int N, int m, int k, int KP,
                                                                 GridAxis ax, uint16_t* __restrict__ rc, long rc_plane,
                                                                 uint16_t* __restrict__ r8, long r8_plane, int r8_rows,
                                                                 const float* __restrict__ ssq, int nsl) {
    extern __shared__ __attribute__((aligned(16))) int2 s_tab[];               // [KP/4] per float4 unit: {offset in floats from the row's voxel, d0 | d1<<8 | d2<<16 | kind<<24}
    uint16_t* s_img = reinterpret_cast<uint16_t*>(s_tab + KP / 4);             // [NP][8][KP] (only when R8 planes are written)
    __shared__ RowInfo s_row[8];
    __shared__ __attribute__((aligned(16))) float s_sc[8][kF];
    PR_STAMP(0);
    const int tid = threadIdx.x;
    const int G = m * m * m, h = (k - 1) / 2;
    const int E4 = k * k * k * (kF / 4), U = KP / 4, U2 = KP / 8;            // window units; units / 8-column groups per row
    for (int j = tid; j < U; j += 512) {
        int2 e = make_int2(0, (j == E4 ? 1 : 2) << 24);               // y >> 24: 0 = window unit, 1 = the q - centre unit, 2 = zero padding
        if (j < E4) {
            const int nb = j / 5, part = j % 5;
            const int d0 = nb / (k * k), d1 = (nb / k) % k, d2 = nb % k;       // 0 .. k-1 (displacement + h), grid axes (y, x, z)
            e = make_int2((((d0 - h) * m + (d1 - h)) * m + (d2 - h)) * kF + part * 4, d0 | (d1 << 8) | (d2 << 16));
        }
        s_tab[j] = e;
    }
    {
        const int rg = blockIdx.x;
        if (ssq && tid < 8 * kF) {
            const int rr = tid / kF, ch = tid % kF;
            s_sc[rr][ch] = fv_scale(ssq, nsl, (8 * rg + rr) / N, ch);
        }
        if (tid < 8) {
            const int r = 8 * rg + tid;
            const float qx = q[(size_t)r * 3], qy = q[(size_t)r * 3 + 1], qz = q[(size_t)r * 3 + 2];
            int ix = cell_of(ax, m, qx), iy = cell_of(ax, m, qy), iz = cell_of(ax, m, qz);
            const bool valid = (ix >= 0) && (iy >= 0) && (iz >= 0);
            if (!valid) { ix = 0; iy = 0; iz = 0; }
            s_row[tid] = RowInfo{ix, iy, iz, r / N, qx - ax.c[ix], qy - ax.c[iy], qz - ax.c[iz]};
            mask[r] = valid ? 1.f : 0.f;
            vox[r] = (iy * m + ix) * m + iz;
        }
        __syncthreads();
        PR_STAMP(1);
        const bool want_r8 = r8 && (8 * rg < r8_rows);
        // ---- pass A: wave = row ----
        {
            const int rr = tid >> 6, lane = tid & 63;
            const RowInfo ri = s_row[rr];
            const size_t row = (size_t)(8 * rg + rr);
            const int own = ((ri.iy * m + ri.ix) * m + ri.iz) * kF;                               // this row's own voxel
            const float* fvr = fv + (size_t)ri.cloud * G * kF + own;
            const int by = ri.iy - h, bx = ri.ix - h, bz = ri.iz - h;
            for (int t = lane; t < U2; t += 64) {
                float4 v[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int2 e = s_tab[2 * t + u];
                    const int kind = e.y >> 24;
                    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (kind == 0) {
                        const int g0 = by + (e.y & 0xff), g1 = bx + ((e.y >> 8) & 0xff), g2 = bz + ((e.y >> 16) & 0xff);
                        if ((unsigned)g0 < (unsigned)m && (unsigned)g1 < (unsigned)m && (unsigned)g2 < (unsigned)m) {
                            x = *reinterpret_cast<const float4*>(fvr + e.x);
                            if (ssq) {
                                const float4 sc = *reinterpret_cast<const float4*>(&s_sc[rr][((2 * t + u) % 5) * 4]);
                                x.x *= sc.x; x.y *= sc.y; x.z *= sc.z; x.w *= sc.w;
                            }
                        }
                    } else if (kind == 1) {
                        x = make_float4(ri.dx, ri.dy, ri.dz, 0.f);
                    }
                    v[u] = x;
                }
                if (X) {
                    *reinterpret_cast<float4*>(X + row * KP + 8 * t) = v[0];
                    *reinterpret_cast<float4*>(X + row * KP + 8 * t + 4) = v[1];
                }
                const float e8[8] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w};
                uint4 wv[NP];
                if (NP == 1) {
                    unsigned b[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) b[c] = bf16_bits(e8[c]);
                    wv[0] = make_uint4(b[0] | (b[1] << 16), b[2] | (b[3] << 16), b[4] | (b[5] << 16), b[6] | (b[7] << 16));
                } else {
                    uint4 w3[3];
                    split_chunk(e8, w3);
#pragma unroll
                    for (int p = 0; p < NP; ++p) wv[p] = w3[p];
                }
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    if (rc) *reinterpret_cast<uint4*>(rc + p * rc_plane + row * KP + 8 * t) = wv[p];
                    if (want_r8) *reinterpret_cast<uint4*>(s_img + ((size_t)(p * 8 + rr) * KP + 8 * t)) = wv[p];
                }
            }
        }
        PR_STAMP(2);
        if (!want_r8) return;                                         // uniform per workgroup
        __syncthreads();
        PR_STAMP(3);
        // ---- pass B ----
        for (int c = tid; c < KP; c += 512) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                unsigned b[8];
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) b[rr] = s_img[(size_t)(p * 8 + rr) * KP + c];
                *reinterpret_cast<uint4*>(r8 + p * r8_plane + ((size_t)rg * KP + c) * 8) =
                    make_uint4(b[0] | (b[1] << 16), b[2] | (b[3] << 16), b[4] | (b[5] << 16), b[6] | (b[7] << 16));
            }
        }
        PR_STAMP(4);
    }
}

// Round 3, second form (plane compute types that keep no fp32 X): the 8 rows of a row group are 8 queries of ONE cloud (N % 8 == 0), and
// what they gather is that cloud's 512 x 20 Fisher vector, scaled and rounded to the planes -- the same 10240 values whichever row asks.
// So the workgroup converts the cloud's vector ONCE (40 KB of coalesced float4 loads, x * scale, bf16 / three-plane split: the values
// the row-wise kernel computes per gathered element, so the planes keep their bits) into 20 KB of LDS per plane, and both passes read LDS:
//   pass A (wave = row): two 8-byte LDS reads per 8-column group, one uint4 per plane to the RC plane (1 KiB per wave-instruction);
//   pass B (R8 rows): work item = column, its 8 rows are 8 two-byte LDS reads, lanes contiguous in memory; no LDS image of the rows, no
//   barrier between the passes.
// The stamps of the kernel above (tools/gather_stamps.py, B = 64, one plane: 28.6k cycles per workgroup, 9.1k of them before the first
// gather, 13.1k in pass A at the address unit's rate for ~26 cache lines per gather instruction, 45 KB of LDS = 3 workgroups per CU =
// 1.33 rounds for the 1024 workgroups) are what this form removes: 25 KB of LDS (4 workgroups per CU, one round), 83 MB of scattered
// 16-byte L2 gathers become 42 MB of linear reads, 20.7 M conversions become 10.5 M.
template <int NP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(NP == 1 ? 8 : 4)))      // one plane: <= 64 VGPRs = 4 workgroups per CU = one round at B = 64
void patch_rows_planes_lds_kernel(const float* __restrict__ q, const float* __restrict__ fv,
                                                                    float* __restrict__ mask, int32_t* __restrict__ vox, int Q, int N,
                                                                    int m, int k, int KP, GridAxis ax, uint16_t* __restrict__ rc,
                                                                    long rc_plane, uint16_t* __restrict__ r8, long r8_plane, int r8_rows,
                                                                    const float* __restrict__ ssq, int nsl, unsigned mg_k, unsigned mg_kk) {
    extern __shared__ __attribute__((aligned(16))) int2 s_tab2[];             // [KP/4] unit table (as above), then the planes
    const int U = KP / 4, U2 = KP / 8;
    const int G = m * m * m, h = (k - 1) / 2, GF = G * kF;
    uint16_t* s_fv = reinterpret_cast<uint16_t*>(s_tab2 + U);                // [NP][GF]
    __shared__ RowInfo s_row[8];
    __shared__ int2 s_rb[8];                                                  // per row: {offset of its own voxel in the planes, validity bits of the displacements: axis a, d -> bit 8a + d}
    __shared__ __attribute__((aligned(16))) float s_sc[kF];
    __shared__ __attribute__((aligned(8))) uint16_t s_qc[3][8][4];           // planes of (q - centre, 0) of the 8 rows
    PR_STAMP(0);
    const int tid = threadIdx.x;
    // Workgroups go to the 8 XCDs round robin and every XCD has its own L2: with rg = blockIdx.x the N/8 row groups of a cloud would
    // pull its Fisher vector through 8 different L2s.  Cloud c is therefore handled on XCD c % 8 (all its row groups; the clouds that
    // carry gradient rows -- the first half -- stay spread evenly), whenever the cloud count is a multiple of 8.
    int rg = blockIdx.x;
    {
        const int rgpc = N / 8, clouds = (Q / 8) / rgpc;
        if (!(clouds & 7)) {
            const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
            rg = ((i / rgpc) * 8 + xcd) * rgpc + i % rgpc;
        }
    }
    const int cloud = (8 * rg) / N;
    const float4* fvc = reinterpret_cast<const float4*>(fv + (size_t)cloud * GF);
    const int nv = GF / 4;
    // ---- everything global is requested first; the unit table is built in the shadow of those loads ----
    constexpr int PRE = 5;                                                    // m = 8: 2560 float4 = 5 per thread
    float4 pre[PRE];
#pragma unroll
    for (int i = 0; i < PRE; ++i) {
        const int idx = tid + 512 * i;
        pre[i] = idx < nv ? fvc[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < kF) s_sc[tid] = ssq ? fv_scale(ssq, nsl, cloud, tid) : 1.0f;
    if (tid >= 64 && tid < 72) {
        const int rr = tid - 64, r = 8 * rg + rr;
        const float qx = q[(size_t)r * 3], qy = q[(size_t)r * 3 + 1], qz = q[(size_t)r * 3 + 2];
        int ix = cell_of(ax, m, qx), iy = cell_of(ax, m, qy), iz = cell_of(ax, m, qz);
        const bool valid = (ix >= 0) && (iy >= 0) && (iz >= 0);
        if (!valid) { ix = 0; iy = 0; iz = 0; }
        const float dq[4] = {qx - ax.c[ix], qy - ax.c[iy], qz - ax.c[iz], 0.f};
        s_row[rr] = RowInfo{ix, iy, iz, r / N, dq[0], dq[1], dq[2]};
        unsigned vb = 0;
        for (int d = 0; d < k; ++d) {
            if ((unsigned)(iy - h + d) < (unsigned)m) vb |= 1u << d;
            if ((unsigned)(ix - h + d) < (unsigned)m) vb |= 1u << (8 + d);
            if ((unsigned)(iz - h + d) < (unsigned)m) vb |= 1u << (16 + d);
        }
        s_rb[rr] = make_int2(((iy * m + ix) * m + iz) * kF, (int)vb);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            unsigned p3[3];
            split3(dq[c], p3);
            if (NP == 1) p3[0] = bf16_bits(dq[c]);
#pragma unroll
            for (int p = 0; p < NP; ++p) s_qc[p][rr][c] = (uint16_t)p3[p];
        }
        mask[r] = valid ? 1.f : 0.f;
        vox[r] = (iy * m + ix) * m + iz;
    }
    const int E4 = k * k * k * (kF / 4);
    for (int j = tid; j < U; j += 512) {
        int2 e = make_int2(0, (j == E4 ? 1 : 2) << 24);               // y >> 24: 0 = window unit, 1 = the q - centre unit, 2 = zero padding
        if (j < E4) {
            // no integer divider on gfx950 (~35 VALU instructions per run-time division, and 32 waves per CU build this table at the same
            // time: 11k of this kernel's first 15.6k cycles): j / 5 is a compile-time division, the two by k and k*k are multiplications
            // by host-made reciprocals (2^16 / k rounded up; exact on 0 .. k^3 - 1, checked by the launcher)
            const int nb = j / 5, part = j - 5 * nb;
            const int d0 = (int)(((unsigned)nb * mg_kk) >> 16), r = nb - d0 * k * k;
            const int d1 = (int)(((unsigned)r * mg_k) >> 16), d2 = r - d1 * k;       // 0 .. k-1 (displacement + h), grid axes (y, x, z)
            e = make_int2((((d0 - h) * m + (d1 - h)) * m + (d2 - h)) * kF + part * 4, d0 | (d1 << 8) | (d2 << 16));
        }
        s_tab2[j] = e;
    }
    __syncthreads();                                                          // scales, row info, table
    PR_STAMP(1);
    auto stage = [&](int idx, float4 x) {
        const float4 sc = *reinterpret_cast<const float4*>(&s_sc[(idx % 5) * 4]);
        const float v[4] = {x.x * sc.x, x.y * sc.y, x.z * sc.z, x.w * sc.w};
        unsigned pl[4][3];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (NP == 1) pl[c][0] = bf16_bits(v[c]);
            else split3(v[c], pl[c]);
        }
#pragma unroll
        for (int p = 0; p < NP; ++p)
            *reinterpret_cast<uint2*>(s_fv + (size_t)p * GF + 4 * idx) = make_uint2(pl[0][p] | (pl[1][p] << 16), pl[2][p] | (pl[3][p] << 16));
    };
#pragma unroll
    for (int i = 0; i < PRE; ++i)
        if (tid + 512 * i < nv) stage(tid + 512 * i, pre[i]);
    for (int idx = tid + 512 * PRE; idx < nv; idx += 512) stage(idx, fvc[idx]);      // (m > 8 only)
    __syncthreads();
    PR_STAMP(2);
    const bool want_r8 = r8 && (8 * rg < r8_rows);
    // ---- pass A: wave = row ----
    if (rc) {
        const int rr = tid >> 6, lane = tid & 63;
        const size_t row = (size_t)(8 * rg + rr);
        const int own = s_rb[rr].x;
        const unsigned vbits = (unsigned)s_rb[rr].y;
        for (int t = lane; t < U2; t += 64) {
            uint2 w[2][NP];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int2 e = s_tab2[2 * t + u];
                const int kind = e.y >> 24;
#pragma unroll
                for (int p = 0; p < NP; ++p) w[u][p] = make_uint2(0u, 0u);
                if (kind == 0) {
                    if ((vbits >> (e.y & 0xff)) & (vbits >> (8 + ((e.y >> 8) & 0xff))) & (vbits >> (16 + ((e.y >> 16) & 0xff))) & 1u) {
#pragma unroll
                        for (int p = 0; p < NP; ++p) w[u][p] = *reinterpret_cast<const uint2*>(s_fv + (size_t)p * GF + own + e.x);
                    }
                } else if (kind == 1) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) w[u][p] = *reinterpret_cast<const uint2*>(&s_qc[p][rr][0]);
                }
            }
#pragma unroll
            for (int p = 0; p < NP; ++p)
                *reinterpret_cast<uint4*>(rc + p * rc_plane + row * KP + 8 * t) = make_uint4(w[0][p].x, w[0][p].y, w[1][p].x, w[1][p].y);
        }
    }
    PR_STAMP(3);
    if (!want_r8) return;
    // ---- pass B: work item = column: its 8 rows are 8 two-byte LDS reads, one 16-byte chunk per plane, lanes contiguous in memory (1 KiB
    // per store instruction).  Measured against one work item per float4 unit (8-byte LDS reads, v_perm transposes, but 16-byte pieces
    // 64 bytes apart per store instruction): kernel 20.9 -> 16.5 us at B = 64, one plane ----
    int2 rb[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) rb[rr] = s_rb[rr];
    for (int c = tid; c < KP; c += 512) {
        const int2 e = s_tab2[c >> 2];
        const int kind = e.y >> 24, sub = c & 3;
        const int s0 = e.y & 0xff, s1 = 8 + ((e.y >> 8) & 0xff), s2 = 16 + ((e.y >> 16) & 0xff);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            unsigned b[8];
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                b[rr] = 0u;
                const unsigned vb = (unsigned)rb[rr].y;
                if (kind == 0) { if ((vb >> s0) & (vb >> s1) & (vb >> s2) & 1u) b[rr] = s_fv[(size_t)p * GF + rb[rr].x + e.x + sub]; }
                else if (kind == 1) b[rr] = s_qc[p][rr][sub];
            }
            *reinterpret_cast<uint4*>(r8 + p * r8_plane + ((size_t)rg * KP + c) * 8) =
                make_uint4(b[0] | (b[1] << 16), b[2] | (b[3] << 16), b[4] | (b[5] << 16), b[6] | (b[7] << 16));
        }
    }
    PR_STAMP(4);
}

// The same form for the fp32 rows of the exact compute type (X [Q, KP], no planes): the cloud's scaled Fisher vector is staged once
// per workgroup as fp32 (40 KB), wave = row copies 16-byte pieces out of LDS into 1-KiB contiguous stores.  Replaces five run-time
// integer divisions and one scattered 16-byte L2 gather per float4 of patch_rows_fwd_kernel.  Same products (fv * scale), same bits.
__global__ __launch_bounds__(512) void patch_rows_fwd_lds_kernel(const float* __restrict__ q, const float* __restrict__ fv,
                                                                 float* __restrict__ X, float* __restrict__ mask,
                                                                 int32_t* __restrict__ vox, int Q, int N, int m, int k, int KP,
                                                                 GridAxis ax, const float* __restrict__ ssq, int nsl, unsigned mg_k,
                                                                 unsigned mg_kk) {
    extern __shared__ __attribute__((aligned(16))) int2 s_tab3[];             // [KP/4] unit table, then the scaled vector [G*kF] fp32
    const int U = KP / 4;
    const int G = m * m * m, h = (k - 1) / 2, GF = G * kF;
    float* s_fv = reinterpret_cast<float*>(s_tab3 + U);
    __shared__ int2 s_rb[8];
    __shared__ float4 s_dq[8];
    __shared__ __attribute__((aligned(16))) float s_sc[kF];
    const int tid = threadIdx.x;
    int rg = blockIdx.x;
    {
        const int rgpc = N / 8, clouds = (Q / 8) / rgpc;
        if (!(clouds & 7)) {                      // cloud c on XCD c % 8 (one L2 per cloud), as in the plane form
            const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
            rg = ((i / rgpc) * 8 + xcd) * rgpc + i % rgpc;
        }
    }
    const int cloud = (8 * rg) / N;
    const float4* fvc = reinterpret_cast<const float4*>(fv + (size_t)cloud * GF);
    const int nv = GF / 4;
    constexpr int PRE = 5;
    float4 pre[PRE];
#pragma unroll
    for (int i = 0; i < PRE; ++i) {
        const int idx = tid + 512 * i;
        pre[i] = idx < nv ? fvc[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < kF) s_sc[tid] = ssq ? fv_scale(ssq, nsl, cloud, tid) : 1.0f;
    if (tid >= 64 && tid < 72) {
        const int rr = tid - 64, r = 8 * rg + rr;
        const float qx = q[(size_t)r * 3], qy = q[(size_t)r * 3 + 1], qz = q[(size_t)r * 3 + 2];
        int ix = cell_of(ax, m, qx), iy = cell_of(ax, m, qy), iz = cell_of(ax, m, qz);
        const bool valid = (ix >= 0) && (iy >= 0) && (iz >= 0);
        if (!valid) { ix = 0; iy = 0; iz = 0; }
        s_dq[rr] = make_float4(qx - ax.c[ix], qy - ax.c[iy], qz - ax.c[iz], 0.f);      // point_cloud - Centers (:491)
        unsigned vb = 0;
        for (int d = 0; d < k; ++d) {
            if ((unsigned)(iy - h + d) < (unsigned)m) vb |= 1u << d;
            if ((unsigned)(ix - h + d) < (unsigned)m) vb |= 1u << (8 + d);
            if ((unsigned)(iz - h + d) < (unsigned)m) vb |= 1u << (16 + d);
        }
        s_rb[rr] = make_int2(((iy * m + ix) * m + iz) * kF, (int)vb);
        mask[r] = valid ? 1.f : 0.f;
        vox[r] = (iy * m + ix) * m + iz;
    }
    const int E4 = k * k * k * (kF / 4);
    for (int j = tid; j < U; j += 512) {
        int2 e = make_int2(0, (j == E4 ? 1 : 2) << 24);
        if (j < E4) {
            const int nb = j / 5, part = j - 5 * nb;
            const int d0 = (int)(((unsigned)nb * mg_kk) >> 16), r = nb - d0 * k * k;
            const int d1 = (int)(((unsigned)r * mg_k) >> 16), d2 = r - d1 * k;
            e = make_int2((((d0 - h) * m + (d1 - h)) * m + (d2 - h)) * kF + part * 4, d0 | (d1 << 8) | (d2 << 16));
        }
        s_tab3[j] = e;
    }
    __syncthreads();
    auto stage = [&](int idx, float4 x) {
        if (ssq) {      // (without ssq the vector is already normalised: copied as it is, like the row-wise kernel)
            const float4 sc = *reinterpret_cast<const float4*>(&s_sc[(idx % 5) * 4]);
            x.x *= sc.x; x.y *= sc.y; x.z *= sc.z; x.w *= sc.w;
        }
        *reinterpret_cast<float4*>(s_fv + 4 * idx) = x;
    };
#pragma unroll
    for (int i = 0; i < PRE; ++i)
        if (tid + 512 * i < nv) stage(tid + 512 * i, pre[i]);
    for (int idx = tid + 512 * PRE; idx < nv; idx += 512) stage(idx, fvc[idx]);
    __syncthreads();
    const int rr = tid >> 6, lane = tid & 63;
    const size_t row = (size_t)(8 * rg + rr);
    const int own = s_rb[rr].x;
    const unsigned vbits = (unsigned)s_rb[rr].y;
    float* xr = X + row * KP;
    for (int j = lane; j < U; j += 64) {
        const int2 e = s_tab3[j];
        const int kind = e.y >> 24;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kind == 0) {
            if ((vbits >> (e.y & 0xff)) & (vbits >> (8 + ((e.y >> 8) & 0xff))) & (vbits >> (16 + ((e.y >> 16) & 0xff))) & 1u)
                v = *reinterpret_cast<const float4*>(s_fv + own + e.x);
        } else if (kind == 1) {
            v = s_dq[rr];
        }
        *reinterpret_cast<float4*>(xr + 4 * j) = v;
    }
}

// Backward as a gather (patch_rows_bwd.h): block (c, slice) owns a slice of the voxels of cloud c
__global__ __launch_bounds__(256) void patch_rows_bwd_kernel(const float* __restrict__ dX, const int32_t* __restrict__ vox,
                                                              float* __restrict__ dfv, int N, int m, int k, int KP,
                                                              int slices) {
    extern __shared__ int s_vox[];   // [N] voxel coordinates of the cloud's queries, packed (a0 | a1<<8 | a2<<16)
    // (pinning a cloud's slices to one XCD was measured: no change, 16.8 us)
    patch_rows_bwd_block<256>(dX, vox, dfv, N, m, k, KP, blockIdx.x / slices, blockIdx.x % slices, slices, s_vox);
}

__global__ __launch_bounds__(256) void patch_rows_dq_kernel(const float* __restrict__ dX, float* __restrict__ dq, int Q,
                                                             int KP, int E) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < Q * 3) dq[i] = dX[(size_t)(i / 3) * KP + E + i % 3];
}

// End of the as-loss backward (pcrnet-registration/iterative_PCRNet_ours.py:255-257: gradients to input1 / input2 only) in ONE launch:
//   d loss / d pcA = upstream * (encoder route dpts[0:B] + query route of the BA half, dX[(B+b)*N+n, E:E+3])
//   d loss / d pcB = upstream * (encoder route dpts[B:2B] + query route of the AB half, dX[b*N+n, E:E+3])
// (was: a dq extraction launch plus three elementwise torch launches).  `scale` = the upstream gradient, a device scalar (may be NULL).
__global__ __launch_bounds__(256) void asloss_combine_kernel(const float* __restrict__ dpts, const float* __restrict__ dX,
                                                              const float* __restrict__ scale, int BN, int KP, int E,
                                                              float* __restrict__ gA, float* __restrict__ gB) {
    const int i = blockIdx.x * 256 + threadIdx.x;      // over [2, B*N, 3]
    if (i >= 2 * BN * 3) return;
    const float sc = scale ? *scale : 1.0f;
    const int which = i / (BN * 3), r = (i % (BN * 3)) / 3, c = i % 3;
    const int qrow = which ? r : BN + r;               // pcB is queried by the AB half (rows < BN), pcA by the BA half
    const float v = (dpts[i] + dX[(size_t)qrow * KP + E + c]) * sc;
    (which ? gB : gA)[(size_t)r * 3 + c] = v;
}

// pts = [pcA + noise ; pcB] (encoder input), q = [pcB ; pcA] (query clouds): models/dpdist_and_aue.py:45,56-61,69
__global__ __launch_bounds__(256) void stack_clouds_kernel(const float* __restrict__ pcA, const float* __restrict__ pcB,
                                                            const float* __restrict__ noise, int n, float* __restrict__ pts,
                                                            float* __restrict__ q) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float a = pcA[i], b = pcB[i];
    pts[i] = noise ? a + noise[i] : a;
    pts[n + i] = b;
    q[i] = b;
    q[n + i] = a;
}

}  // namespace dpd

extern "C" int dpd_stack_clouds(const float* pcA, const float* pcB, const float* noise, int B, int N, float* pts, float* q,
                                void* stream) {
    if (!pcA || !pcB || !pts || !q) return DPD_E_NULL;
    if (B <= 0 || N <= 0) return DPD_E_DIM;
    const int n = B * N * 3;
    DPD_LAUNCH(dpd::stack_clouds_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pcA, pcB, noise, n, pts, q);
    DPD_CHECK_LAUNCH();
    return 0;
}

extern "C" int dpd_padded_width(int k) { return (k * k * k * DPD_FV_CHANNELS + 3 + 31) / 32 * 32; }

extern "C" int dpd_patch_rows_fwd(const float* q, const float* fv, int C, int N, int m, int k, int KP, float* X,
                                  float* mask, int32_t* vox, const dpd_planes* pl, void* stream) {
    return dpd_patch_rows_fwd_scaled(q, fv, nullptr, C, N, m, k, KP, X, mask, vox, pl, stream);
}

extern "C" int dpd_patch_rows_fwd_scaled(const float* q, const float* fv, const float* ssq, int C, int N, int m, int k, int KP,
                                         float* X, float* mask, int32_t* vox, const dpd_planes* pl, void* stream) {
    using namespace dpd;
    const int Q = C * N;
    const bool planes = pl && (pl->X_rc || pl->X_r8) && !(Q & 7) && !(pl->Qb & 31) && !(KP & 31);
    if (!q || !fv || (!X && !planes) || !mask || !vox) return DPD_E_NULL;
    if (C <= 0 || N <= 0) return DPD_E_DIM;
    if (m < 1 || m > 10 || k < 1 || k > 7 || !(k & 1)) return DPD_E_UNSUPPORTED;
    if (KP < k * k * k * kF + 3 || (KP & 3)) return DPD_E_DIM;
    // algorithmic bytes: the clouds' Fisher vectors and the query points in; the rows out -- fp32 [Q,KP] and / or bf16 planes (RC for
    // all Q rows, R8 for the Qb rows that carry gradient) -- plus mask and voxel id per row
    StageProf prof(stream, DPD_STAGE_GATHER,
                   (double)C * m * m * m * kF * 4.0 + Q * 12.0 + Q * 8.0 + (X ? (double)Q * KP * 4.0 : 0.0) +
                       (planes ? pl->np * 2.0 * KP * ((pl->X_rc ? (double)Q : 0.0) + (pl->X_r8 ? (double)pl->Qb : 0.0)) : 0.0));
    if (planes) {
        if ((pl->np != 1 && pl->np != 3) || pl->Q != Q || pl->Qb > Q || pl->Qb < 0) return DPD_E_DIM;
        if (!X && !(N & 7) &&
                   (size_t)pl->np * m * m * m * kF * sizeof(uint16_t) + (size_t)(KP / 4) * sizeof(int2) <= 100 * 1024) {
            // no fp32 rows wanted and a row group never straddles two clouds: the cloud's planes are made once per workgroup, in LDS
            const size_t lds = (size_t)pl->np * m * m * m * kF * sizeof(uint16_t) + (size_t)(KP / 4) * sizeof(int2);
            const unsigned mg_k = 65536u / k + 1, mg_kk = 65536u / (k * k) + 1;          // reciprocals for the unit table
            for (int x = 0; x < k * k * k; ++x)
                if ((int)((x * mg_kk) >> 16) != x / (k * k) || (int)(((x % (k * k)) * mg_k) >> 16) != (x % (k * k)) / k) return DPD_E_UNSUPPORTED;
            if (pl->np == 1) {
                static LdsOptIn ll1;
                if (int rc2 = ensure_dyn_lds(ll1, (const void*)patch_rows_planes_lds_kernel<1>, lds)) return rc2;
                DPD_LAUNCH(patch_rows_planes_lds_kernel<1>, dim3(Q / 8), dim3(512), lds, (hipStream_t)stream, q, fv, mask, vox, Q, N, m, k, KP,
                           make_axis(m), (uint16_t*)pl->X_rc, (long)Q * KP, (uint16_t*)pl->X_r8, (long)pl->Qb * KP, pl->Qb, ssq, kMfvSlices, mg_k, mg_kk);
            } else {
                static LdsOptIn ll3;
                if (int rc2 = ensure_dyn_lds(ll3, (const void*)patch_rows_planes_lds_kernel<3>, lds)) return rc2;
                DPD_LAUNCH(patch_rows_planes_lds_kernel<3>, dim3(Q / 8), dim3(512), lds, (hipStream_t)stream, q, fv, mask, vox, Q, N, m, k, KP,
                           make_axis(m), (uint16_t*)pl->X_rc, (long)Q * KP, (uint16_t*)pl->X_r8, (long)pl->Qb * KP, pl->Qb, ssq, kMfvSlices, mg_k, mg_kk);
            }
        } else {
            const size_t lds = (pl->X_r8 ? (size_t)pl->np * 8 * KP * sizeof(uint16_t) : 0) + (size_t)(KP / 4) * sizeof(int2);
            if (lds > 150 * 1024) return DPD_E_UNSUPPORTED;
            if (pl->np == 1) {
                static LdsOptIn lo1;
                if (int rc2 = ensure_dyn_lds(lo1, (const void*)patch_rows_planes3_kernel<1>, lds)) return rc2;
                DPD_LAUNCH(patch_rows_planes3_kernel<1>, dim3(Q / 8), dim3(512), lds, (hipStream_t)stream, q, fv, X, mask, vox, Q, N, m, k,
                           KP, make_axis(m), (uint16_t*)pl->X_rc, (long)Q * KP, (uint16_t*)pl->X_r8, (long)pl->Qb * KP, pl->Qb, ssq, kMfvSlices);
            } else {
                static LdsOptIn lo3;
                if (int rc2 = ensure_dyn_lds(lo3, (const void*)patch_rows_planes3_kernel<3>, lds)) return rc2;
                DPD_LAUNCH(patch_rows_planes3_kernel<3>, dim3(Q / 8), dim3(512), lds, (hipStream_t)stream, q, fv, X, mask, vox, Q, N, m, k,
                           KP, make_axis(m), (uint16_t*)pl->X_rc, (long)Q * KP, (uint16_t*)pl->X_r8, (long)pl->Qb * KP, pl->Qb, ssq, kMfvSlices);
            }
        }
        DPD_CHECK_LAUNCH();
        return 0;
    }
    {
        const size_t lds = (size_t)m * m * m * kF * sizeof(float) + (size_t)(KP / 4) * sizeof(int2);
        const unsigned mg_k = 65536u / k + 1, mg_kk = 65536u / (k * k) + 1;
        bool exact = true;
        for (int x = 0; x < k * k * k; ++x)
            if ((int)((x * mg_kk) >> 16) != x / (k * k) || (int)(((x % (k * k)) * mg_k) >> 16) != (x % (k * k)) / k) exact = false;
        if (exact && !(N & 7) && !(Q & 7) && lds <= 100 * 1024) {
            static LdsOptIn lr;
            if (int rc2 = ensure_dyn_lds(lr, (const void*)patch_rows_fwd_lds_kernel, lds)) return rc2;
            DPD_LAUNCH(patch_rows_fwd_lds_kernel, dim3(Q / 8), dim3(512), lds, (hipStream_t)stream, q, fv, X, mask, vox, Q, N, m, k, KP,
                       make_axis(m), ssq, kMfvSlices, mg_k, mg_kk);
            DPD_CHECK_LAUNCH();
            return 0;
        }
    }
    DPD_LAUNCH(patch_rows_fwd_kernel, dim3((C * N + 1) / 2), dim3(256), 0, (hipStream_t)stream, q, fv, X, mask, vox, N, m, k,
                       KP, make_axis(m), C * N, ssq, kMfvSlices);
    DPD_CHECK_LAUNCH();
    return 0;
}

extern "C" int dpd_asloss_combine(const float* dpts, const float* dX, const float* scale, int B, int N, int k, int KP, float* gA,
                                  float* gB, void* stream) {
    using namespace dpd;
    if (!dpts || !dX || !gA || !gB) return DPD_E_NULL;
    if (B <= 0 || N <= 0) return DPD_E_DIM;
    if (k < 1 || k > 7 || !(k & 1)) return DPD_E_UNSUPPORTED;
    if (KP < k * k * k * kF + 3) return DPD_E_DIM;
    const int n = 2 * B * N * 3;
    DPD_LAUNCH(asloss_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dpts, dX, scale, B * N, KP, k * k * k * kF, gA, gB);
    DPD_CHECK_LAUNCH();
    return 0;
}

#ifdef DPD_ABLATIONS
extern "C" int dpd_debug_pr_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(dpd::g_pr_stamps), sizeof(unsigned long long) * 1024 * 8, 0, hipMemcpyDeviceToHost);
}
#endif
extern "C" int dpd_patch_rows_bwd(const float* dX, const int32_t* vox, int C, int N, int m, int k, int KP, float* dq,
                                  float* dfv, void* stream) {
    using namespace dpd;
    if (!dX || !vox) return DPD_E_NULL;
    if (C <= 0 || N <= 0) return DPD_E_DIM;
    if (m < 1 || m > 10 || k < 1 || k > 7 || !(k & 1) || N > 8192) return DPD_E_UNSUPPORTED;
    if (KP < k * k * k * kF + 3 || (KP & 3)) return DPD_E_DIM;
    if (dfv) {
        const int slices = (m * m * m + 50) / 51;      // <= 51 voxels x 5 channel groups = one item per thread, no ragged second pass
        DPD_LAUNCH(patch_rows_bwd_kernel, dim3(C * slices), dim3(256), (size_t)N * sizeof(int), (hipStream_t)stream,
                           dX, vox, dfv, N, m, k, KP, slices);
        DPD_CHECK_LAUNCH();
    }
    if (dq) {
        const int Q = C * N;
        DPD_LAUNCH(patch_rows_dq_kernel, dim3((Q * 3 + 255) / 256), dim3(256), 0, (hipStream_t)stream, dX, dq, Q,
                           KP, k * k * k * kF);
        DPD_CHECK_LAUNCH();
    }
    return 0;
}
