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
#include "common.h"

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

__global__ __launch_bounds__(128) void patch_rows_fwd_kernel(const float* __restrict__ q, const float* __restrict__ fv,
                                                              float* __restrict__ X, float* __restrict__ mask,
                                                              int32_t* __restrict__ vox, int N, int m, int k, int KP,
                                                              GridAxis ax) {
    const int r = blockIdx.x, tid = threadIdx.x;
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

// Backward as a gather (deterministic, no atomics): block (c, slice) owns a slice of the voxels of cloud c and,
// for every (voxel, float4 channel group), sums the window column of every query of the cloud that covers it.
__global__ __launch_bounds__(256) void patch_rows_bwd_kernel(const float* __restrict__ dX, const int32_t* __restrict__ vox,
                                                              float* __restrict__ dfv, int N, int m, int k, int KP,
                                                              int slices) {
    extern __shared__ int s_vox[];   // [N] voxel coordinates of the cloud's queries, packed (a0 | a1<<8 | a2<<16)
    const int c = blockIdx.x / slices, sl = blockIdx.x % slices, tid = threadIdx.x;
    const int G = m * m * m, h = (k - 1) / 2;
    for (int n = tid; n < N; n += 256) {
        const int v = vox[(size_t)c * N + n];
        s_vox[n] = (v / (m * m)) | (((v / m) % m) << 8) | ((v % m) << 16);
    }
    __syncthreads();
    const int gper = (G + slices - 1) / slices;
    const int gbeg = sl * gper, gend = min(G, gbeg + gper);
    const float* dXc = dX + (size_t)c * N * KP;
    for (int item = tid; item < (gend - gbeg) * 5; item += 256) {
        const int g = gbeg + item / 5, part = item % 5;
        const int g0 = g / (m * m) + h, g1 = (g / m) % m + h, g2 = g % m + h;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int n = 0; n < N; ++n) {
            const int pv = s_vox[n];
            const int d0 = g0 - (pv & 255), d1 = g1 - ((pv >> 8) & 255), d2 = g2 - (pv >> 16);
            if ((unsigned)d0 < (unsigned)k && (unsigned)d1 < (unsigned)k && (unsigned)d2 < (unsigned)k) {
                const float4 x = *reinterpret_cast<const float4*>(dXc + (size_t)n * KP + ((d0 * k + d1) * k + d2) * kF + part * 4);
                acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
            }
        }
        *reinterpret_cast<float4*>(dfv + ((size_t)c * G + g) * kF + part * 4) = acc;
    }
}

__global__ __launch_bounds__(256) void patch_rows_dq_kernel(const float* __restrict__ dX, float* __restrict__ dq, int Q,
                                                             int KP, int E) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < Q * 3) dq[i] = dX[(size_t)(i / 3) * KP + E + i % 3];
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
                                  float* mask, int32_t* vox, void* stream) {
    using namespace dpd;
    if (!q || !fv || !X || !mask || !vox) return DPD_E_NULL;
    if (C <= 0 || N <= 0) return DPD_E_DIM;
    if (m < 1 || m > 10 || k < 1 || k > 7 || !(k & 1)) return DPD_E_UNSUPPORTED;
    if (KP < k * k * k * kF + 3 || (KP & 3)) return DPD_E_DIM;
    DPD_LAUNCH(patch_rows_fwd_kernel, dim3(C * N), dim3(128), 0, (hipStream_t)stream, q, fv, X, mask, vox, N, m, k,
                       KP, make_axis(m));
    DPD_CHECK_LAUNCH();
    return 0;
}

extern "C" int dpd_patch_rows_bwd(const float* dX, const int32_t* vox, int C, int N, int m, int k, int KP, float* dq,
                                  float* dfv, void* stream) {
    using namespace dpd;
    if (!dX || !vox) return DPD_E_NULL;
    if (C <= 0 || N <= 0) return DPD_E_DIM;
    if (m < 1 || m > 10 || k < 1 || k > 7 || !(k & 1) || N > 8192) return DPD_E_UNSUPPORTED;
    if (KP < k * k * k * kF + 3 || (KP & 3)) return DPD_E_DIM;
    if (dfv) {
        const int slices = 8;
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
