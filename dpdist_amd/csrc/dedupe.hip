// Row de-duplication for the first decoder layer.
//
// The 2500 window columns of a decoder input row depend only on (cloud, voxel of the query): queries of one cloud that
// fall into the same voxel share them, and surface-shaped clouds put 64 queries into ~35-45 distinct voxels
// (get_emb_and_concat gathers the SAME embedding row for them, utils/dpdist_util.py:434-457; only the 3 centre-relative
// coordinates differ, :455).  So layer 1 and its weight gradient run on the UNIQUE rows:
//
//   forward    T[u]  = Xwin[u] W1win                          (GEMM over U <= Q rows, U known only on the device)
//              h1[q] = relu( T[u(q)] + xyz[q] W1xyz + b1 )    (expand kernel; same fmaf chain as the GEMM would run)
//   backward   G[u]  = sum_{q in u} g1[q]                     (segment sum, fixed order, queries of one cloud)
//              dW1win = Xwin^T G   (contraction over U_ab rows), dW1xyz = xyz^T g1 (3 x H, two small kernels)
//
// Results equal the row-by-row evaluation up to the association of the three xyz terms inside the fp32 sum.
//
// STATUS: correct (tests/test_gpu_parity.py::test_unique_row_layer1_matches_plain_path) but NOT the default: at B = 32
// the GEMMs shrink by 51 us per step (U/Q = 0.63 on the ModelNet-shaped bench data) while the bookkeeping kernels here
// cost 83 us as written, and even free bookkeeping would gain little because a <= 1-round grid on 256 CUs finishes in one
// tile-time however many tiles are missing -- the saving needs a stream-K decomposition of the layer-1 GEMM to
// materialise.  Opt in with DPDistTrainer(dedupe=True) / DPD_DEDUPE=1.
// All counts live in `counts` on the device (no host sync): counts[0] = U, [1] = U_ab (unique rows of the first Qb
// queries = the half that carries gradient; they come first), [2] = U_ab rounded up to 32, [3] = U rounded up to 32.
#include "common.h"

namespace dpd {

constexpr int kF = DPD_FV_CHANNELS;

__device__ __forceinline__ int cell_of2(const GridAxis& ax, int m, float q) {   // same two comparisons as patch_rows.hip
    int r = -1;
    for (int i = m - 1; i >= 0; --i) {
        const float lo = ax.c[i] - ax.half, hi = ax.c[i] + ax.half;
        if (q > lo && q <= hi) r = i;
    }
    return r;
}

// ONE workgroup of 1024 threads: voxel lookup of every query, first-occurrence flags inside each cloud, a block-wide
// exclusive scan -> dense unique-row indices in query order.
__global__ __launch_bounds__(1024) void dedupe_rows_kernel(const float* __restrict__ q, int Q, int N, int m, int Qb, GridAxis ax,
                                                           int32_t* __restrict__ u_of_q, int32_t* __restrict__ rep_q,
                                                           int32_t* __restrict__ counts, float* __restrict__ xyz,
                                                           float* __restrict__ mask, int32_t* __restrict__ vox) {
    extern __shared__ int s_i[];
    int* s_vox = s_i;            // [Q] voxel id (unique inside a cloud together with the cloud index)
    int* s_idx = s_i + Q;        // [Q] first-occurrence flag, then exclusive scan
    int* s_lead = s_i + 2 * Q;   // [Q] earliest query of the same (cloud, voxel)
    __shared__ int s_wsum[16];
    const int tid = threadIdx.x;
    for (int r = tid; r < Q; r += 1024) {
        const float qx = q[(size_t)r * 3], qy = q[(size_t)r * 3 + 1], qz = q[(size_t)r * 3 + 2];
        int ix = cell_of2(ax, m, qx), iy = cell_of2(ax, m, qy), iz = cell_of2(ax, m, qz);
        const bool valid = (ix >= 0) && (iy >= 0) && (iz >= 0);
        if (!valid) { ix = 0; iy = 0; iz = 0; }          // argmax of an all-zero row is index 0 (:490); output is masked
        const int v = (iy * m + ix) * m + iz;
        s_vox[r] = v;
        vox[r] = v;
        mask[r] = valid ? 1.f : 0.f;
        xyz[(size_t)r * 3] = qx - ax.c[ix];                 // point_cloud - Centers (:491)
        xyz[(size_t)r * 3 + 1] = qy - ax.c[iy];
        xyz[(size_t)r * 3 + 2] = qz - ax.c[iz];
    }
    __syncthreads();
    for (int r = tid; r < Q; r += 1024) {
        const int c0 = (r / N) * N, v = s_vox[r];
        int lead = r;
        for (int m2 = c0; m2 < r; ++m2)
            if (s_vox[m2] == v) { lead = m2; break; }
        s_lead[r] = lead;
        s_idx[r] = (lead == r) ? 1 : 0;
    }
    __syncthreads();
    // exclusive scan of the flags: thread t owns the contiguous chunk [t*per, (t+1)*per)
    const int per = (Q + 1023) / 1024;
    const int beg = min(Q, tid * per), end = min(Q, beg + per);
    int local = 0;
    for (int r = beg; r < end; ++r) local += s_idx[r];
    int incl = local;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += s_wsum[w];
    int run = base + incl - local;
    for (int r = beg; r < end; ++r) {
        const int f = s_idx[r];
        s_idx[r] = run;
        run += f;
    }
    __syncthreads();
    int total = 0;
    for (int w = 0; w < 16; ++w) total += s_wsum[w];
    for (int r = tid; r < Q; r += 1024) {
        const int lead = s_lead[r];
        const int u = s_idx[lead];
        u_of_q[r] = u;
        if (lead == r) rep_q[u] = r;
    }
    if (tid == 0) {
        const int uab = (Qb >= Q) ? total : s_idx[Qb];     // unique rows among the first Qb queries (cloud boundary)
        counts[0] = total;
        counts[1] = uab;
        counts[2] = (uab + 31) / 32 * 32;
        counts[3] = (total + 31) / 32 * 32;
    }
}

// Window gather of the unique rows: X_u[u] = [window of (cloud, voxel) of rep_q[u] | 0 0 0 | 0 pad]; rows
// [U, round32(U)) are zero-filled (they pad the contraction of the weight-gradient GEMM).
__global__ __launch_bounds__(128) void patch_rows_unique_kernel(const float* __restrict__ fv, const int32_t* __restrict__ rep_q,
                                                                const int32_t* __restrict__ vox, const int32_t* __restrict__ counts,
                                                                float* __restrict__ X, int N, int m, int k, int KP) {
    const int u = blockIdx.x, tid = threadIdx.x;
    const int U = counts[0];
    float* xr = X + (size_t)u * KP;
    if (u >= U) {
        if (u < counts[3])
            for (int j = tid; j < KP / 4; j += 128) *reinterpret_cast<float4*>(xr + j * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int r = rep_q[u], c = r / N, v = vox[r];
    const int iy = v / (m * m), ix = (v / m) % m, iz = v % m;      // voxel id = (iy*m + ix)*m + iz
    const int G = m * m * m, h = (k - 1) / 2;
    const int E4 = k * k * k * (kF / 4);
    const float* fvc = fv + (size_t)c * G * kF;
    for (int j = tid; j < KP / 4; j += 128) {
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < E4) {
            const int nb = j / 5, part = j % 5;
            const int d0 = nb / (k * k), d1 = (nb / k) % k, d2 = nb % k;
            const int g0 = iy + d0 - h, g1 = ix + d1 - h, g2 = iz + d2 - h;   // grid axes are (y, x, z), slowest first
            if ((unsigned)g0 < (unsigned)m && (unsigned)g1 < (unsigned)m && (unsigned)g2 < (unsigned)m)
                val = *reinterpret_cast<const float4*>(fvc + (size_t)((g0 * m + g1) * m + g2) * kF + part * 4);
        }
        *reinterpret_cast<float4*>(xr + j * 4) = val;
    }
}

// h1[q] = relu( ((T[u(q)] + x W[E]) + y W[E+1]) + z W[E+2] + b1 ), one fmaf per term like the matrix core's chain
__global__ __launch_bounds__(256) void expand_rows_kernel(const float* __restrict__ T, const int32_t* __restrict__ u_of_q,
                                                          const float* __restrict__ xyz, const float* __restrict__ W1p,
                                                          const float* __restrict__ b1, float* __restrict__ h1, int Q, int H, int E) {
    const int per_row = H / 4;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)Q * per_row) return;
    const int r = (int)(i / per_row), j = (int)(i % per_row) * 4;
    const float4 t = *reinterpret_cast<const float4*>(T + (size_t)u_of_q[r] * H + j);
    const float x = xyz[(size_t)r * 3], y = xyz[(size_t)r * 3 + 1], z = xyz[(size_t)r * 3 + 2];
    const float4 w0 = *reinterpret_cast<const float4*>(W1p + (size_t)E * H + j);
    const float4 w1 = *reinterpret_cast<const float4*>(W1p + (size_t)(E + 1) * H + j);
    const float4 w2 = *reinterpret_cast<const float4*>(W1p + (size_t)(E + 2) * H + j);
    const float4 b = *reinterpret_cast<const float4*>(b1 + j);
    float4 o;
    o.x = fmaxf(fmaf(z, w2.x, fmaf(y, w1.x, fmaf(x, w0.x, t.x))) + b.x, 0.f);
    o.y = fmaxf(fmaf(z, w2.y, fmaf(y, w1.y, fmaf(x, w0.y, t.y))) + b.y, 0.f);
    o.z = fmaxf(fmaf(z, w2.z, fmaf(y, w1.z, fmaf(x, w0.z, t.z))) + b.z, 0.f);
    o.w = fmaxf(fmaf(z, w2.w, fmaf(y, w1.w, fmaf(x, w0.w, t.w))) + b.w, 0.f);
    *reinterpret_cast<float4*>(h1 + (size_t)r * H + j) = o;
}

// G[u] = sum of g1 over the queries of unique row u (all in the cloud of rep_q[u], fixed order); rows [U_ab, round32) = 0
__global__ __launch_bounds__(256) void segment_sum_kernel(const float* __restrict__ g1, const int32_t* __restrict__ u_of_q,
                                                          const int32_t* __restrict__ rep_q, const int32_t* __restrict__ counts,
                                                          float* __restrict__ G, int N, int H) {
    const int u = blockIdx.x, Uab = counts[1], Upad = counts[2];
    if (u >= Upad) return;
    for (int j = threadIdx.x * 4; j < H; j += 1024) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (u < Uab) {
            const int r0 = (rep_q[u] / N) * N;
            for (int n = 0; n < N; ++n) {
                if (u_of_q[r0 + n] == u) {
                    const float4 v = *reinterpret_cast<const float4*>(g1 + (size_t)(r0 + n) * H + j);
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
            }
        }
        *reinterpret_cast<float4*>(G + (size_t)u * H + j) = acc;
    }
}

// dW1 rows E..E+2 = xyz^T g1 over the first Qb rows: stage 1 = kXyzChunks row chunks -> partial[chunk][3][H]; stage 2 sums
constexpr int kXyzChunks = 32;
__global__ __launch_bounds__(256) void xyz_grad_stage1(const float* __restrict__ xyz, const float* __restrict__ g1, int Qb, int H,
                                                       float* __restrict__ partial) {
    const int j = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y;
    if (j >= H) return;
    const int per = (Qb + kXyzChunks - 1) / kXyzChunks;
    const int r0 = ch * per, r1 = min(Qb, r0 + per);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int r = r0; r < r1; ++r) {
        const float g = g1[(size_t)r * H + j];
        a0 = fmaf(xyz[(size_t)r * 3], g, a0);
        a1 = fmaf(xyz[(size_t)r * 3 + 1], g, a1);
        a2 = fmaf(xyz[(size_t)r * 3 + 2], g, a2);
    }
    float* p = partial + (size_t)ch * 3 * H;
    p[j] = a0; p[H + j] = a1; p[2 * H + j] = a2;
}
__global__ __launch_bounds__(256) void xyz_grad_stage2(const float* __restrict__ partial, int H, float* __restrict__ dW_rows) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * H) return;
    float s = 0.f;
    for (int ch = 0; ch < kXyzChunks; ++ch) s += partial[(size_t)ch * 3 * H + i];
    dW_rows[i] = s;      // rows E, E+1, E+2 are contiguous in W1p [KP, H]
}

int gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
             int ldc, const float* bias, const float* gate, int epilogue, int split_k, int tile, void* ws,
             size_t ws_bytes, hipStream_t s, float* colsum, const float* A2, const float* B2, float* C2, const int* M_dev,
             const int* K_dev);

}  // namespace dpd

extern "C" int dpd_dedupe_rows(const float* q, int C, int N, int m, int Qb, int32_t* u_of_q, int32_t* rep_q, int32_t* counts,
                               float* xyz, float* mask, int32_t* vox, void* stream) {
    using namespace dpd;
    if (!q || !u_of_q || !rep_q || !counts || !xyz || !mask || !vox) return DPD_E_NULL;
    if (C <= 0 || N <= 0 || Qb < 0 || Qb > C * N || (Qb % N)) return DPD_E_DIM;
    if (m < 1 || m > 10) return DPD_E_UNSUPPORTED;
    const int Q = C * N;
    const size_t lds = (size_t)3 * Q * sizeof(int);
    if (lds > 150 * 1024) return DPD_E_UNSUPPORTED;     // Q <= 12800 queries per call
    if (lds > 64 * 1024) {
        static bool done = false;
        if (!done) {
            DPD_HIP(hipFuncSetAttribute((const void*)dedupe_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            done = true;
        }
    }
    DPD_LAUNCH(dedupe_rows_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, q, Q, N, m, Qb, make_axis(m), u_of_q, rep_q, counts,
               xyz, mask, vox);
    DPD_CHECK_LAUNCH();
    return 0;
}

extern "C" int dpd_patch_rows_fwd_unique(const float* fv, int C, int N, int m, int k, int KP, const int32_t* rep_q,
                                         const int32_t* vox, const int32_t* counts, float* X_u, void* stream) {
    using namespace dpd;
    if (!fv || !rep_q || !vox || !counts || !X_u) return DPD_E_NULL;
    if (C <= 0 || N <= 0) return DPD_E_DIM;
    if (m < 1 || m > 10 || k < 1 || k > 7 || !(k & 1)) return DPD_E_UNSUPPORTED;
    if (KP < k * k * k * kF + 3 || (KP & 3)) return DPD_E_DIM;
    DPD_LAUNCH(patch_rows_unique_kernel, dim3(C * N), dim3(128), 0, (hipStream_t)stream, fv, rep_q, vox, counts, X_u, N, m, k, KP);
    DPD_CHECK_LAUNCH();
    return 0;
}

extern "C" int dpd_layer1_fwd_unique(const float* X_u, const int32_t* counts, const int32_t* u_of_q, const float* xyz, int Q, int KP,
                                     int H, int E, const float* W1p, const float* b1, float* T, float* h1, void* stream) {
    using namespace dpd;
    if (!X_u || !counts || !u_of_q || !xyz || !W1p || !b1 || !T || !h1) return DPD_E_NULL;
    if (Q <= 0 || KP <= 0 || H <= 0 || E + 3 > KP) return DPD_E_DIM;
    if ((KP & 31) || (H & 63) || Q < 4) return DPD_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    // T [U, H] = X_u W1p (the xyz / pad columns of X_u are zero); 64x64 tiles: U is data dependent, small tiles waste least
    // (measured at B = 32: 64x64 0.709 ms/step, 128x128 0.734, 128x64 0.742)
    if (int rc = gemm_f32(0, 0, Q, H, KP, X_u, KP, W1p, H, T, H, nullptr, nullptr, 0, 1, 8, nullptr, 0, s, nullptr,
                          nullptr, nullptr, nullptr, counts, nullptr)) return rc;
    const long n4 = (long)Q * (H / 4);
    DPD_LAUNCH(expand_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (const float*)T, u_of_q, xyz, W1p, b1, h1, Q, H, E);
    DPD_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t dpd_layer1_bwd_unique_workspace_bytes(int Qb, int KP, int H) {
    return ((size_t)(Qb + 32) * H + (size_t)dpd::kXyzChunks * 3 * H + (size_t)2 * KP * H) * sizeof(float);
}

extern "C" int dpd_layer1_bwd_weights_unique(const float* X_u, const float* g1, const int32_t* u_of_q, const int32_t* rep_q,
                                             const float* xyz, const int32_t* counts, int N, int Qb, int KP, int H, int E, float* dW1,
                                             void* ws, size_t ws_bytes, void* stream) {
    using namespace dpd;
    if (!X_u || !g1 || !u_of_q || !rep_q || !xyz || !counts || !dW1 || !ws) return DPD_E_NULL;
    if (N <= 0 || Qb <= 0 || (Qb % N) || KP <= 0 || H <= 0 || E + 3 > KP) return DPD_E_DIM;
    if ((KP & 31) || (H & 63) || (H & 3)) return DPD_E_UNSUPPORTED;
    if (ws_bytes < dpd_layer1_bwd_unique_workspace_bytes(Qb, KP, H)) return DPD_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* G = (float*)ws;                                   // [Qb + 32, H]
    float* part = G + (size_t)(Qb + 32) * H;                 // [chunks][3][H]
    float* slabs = part + (size_t)kXyzChunks * 3 * H;        // split-K slabs of the GEMM
    DPD_LAUNCH(segment_sum_kernel, dim3(Qb + 32), dim3(256), 0, s, g1, u_of_q, rep_q, counts, G, N, H);
    DPD_CHECK_LAUNCH();
    // dW1 [KP, H] = X_u^T [KP, U_ab] G [U_ab, H]: contraction length counts[2] (U_ab rounded up to 32) read on the device
    const int Kmax = (Qb + 31) / 32 * 32;
    if (int rc = gemm_f32(1, 0, KP, H, Kmax, X_u, KP, G, H, dW1, H, nullptr, nullptr, 0, 2, 8, slabs, (size_t)2 * KP * H * sizeof(float), s,
                          nullptr, nullptr, nullptr, nullptr, nullptr, counts + 2)) return rc;
    DPD_LAUNCH(xyz_grad_stage1, dim3((H + 255) / 256, kXyzChunks), dim3(256), 0, s, xyz, g1, Qb, H, part);
    DPD_CHECK_LAUNCH();
    DPD_LAUNCH(xyz_grad_stage2, dim3((3 * H + 255) / 256), dim3(256), 0, s, (const float*)part, H, dW1 + (size_t)E * H);
    DPD_CHECK_LAUNCH();
    return 0;
}
