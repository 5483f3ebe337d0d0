// Window-gather backward as a device routine (shared by patch_rows.hip: patch_rows_bwd_kernel and mfv3d.hip: asloss_tail_a_kernel).
#pragma once
#include "common.h"

namespace dpd {

// Backward as a gather (deterministic, no atomics): block (c, slice) owns a slice of the voxels of cloud c and,
// for every (voxel, float4 channel group), sums the window column of every query of the cloud that covers it.
// NT = threads of the workgroup; s_vox = N ints of LDS.  Every (voxel, channel group) item is computed by ONE thread from the same loads in
// the same order (n ascending) whatever NT / slices are: the result does not depend on the launch shape.
template <int NT>
__device__ __forceinline__ void patch_rows_bwd_block(const float* __restrict__ dX, const int32_t* __restrict__ vox, float* __restrict__ dfv,
                                                     int N, int m, int k, int KP, int c, int sl, int slices, int* s_vox) {
    constexpr int kF = DPD_FV_CHANNELS;
    const int tid = threadIdx.x;
    const int G = m * m * m, h = (k - 1) / 2;
    for (int n = tid; n < N; n += NT) {
        const int v = vox[(size_t)c * N + n];
        s_vox[n] = (v / (m * m)) | (((v / m) % m) << 8) | ((v % m) << 16);
    }
    __syncthreads();
    const int gper = (G + slices - 1) / slices;
    const int gbeg = sl * gper, gend = min(G, gbeg + gper);
    const float* dXc = dX + (size_t)c * N * KP;
    for (int item = tid; item < (gend - gbeg) * 5; item += NT) {
        const int g = gbeg + item / 5, part = item % 5;
        const int g0 = g / (m * m) + h, g1 = (g / m) % m + h, g2 = g % m + h;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // Two passes per 64 queries: (1) which of them cover this voxel -- LDS reads (the same address in every lane) and compares only,
        // a 64-bit hit mask; (2) the hits, sixteen at a time, all sixteen loads issued before the first add (a load inside an `if` made every
        // hit a serial L2 round trip).  A query covers 5^3 of 8^3 voxels, so a thread loads ~16 window columns instead of probing 64
        // (round 2 issued a load for every query and dropped three quarters of them by a select).  The kernel stays latency bound at the
        // PCRNet batch (~12 us back to back, C = 32 clouds of 64 queries; a streaming plane-owner form with the voxels in LDS was 14.7).
        // Same values added in the same order (n ascending) as before.
        for (int n0 = 0; n0 < N; n0 += 64) {
            unsigned long long hm = 0;
            const int lim = min(64, N - n0);
            for (int j = 0; j < lim; ++j) {
                const int pv = s_vox[n0 + j];
                const int d0 = g0 - (pv & 255), d1 = g1 - ((pv >> 8) & 255), d2 = g2 - (pv >> 16);
                if ((unsigned)d0 < (unsigned)k && (unsigned)d1 < (unsigned)k && (unsigned)d2 < (unsigned)k) hm |= 1ull << j;
            }
            while (__any(hm != 0)) {
                float4 x[16];
                bool hit[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    hit[j] = hm != 0;
                    const int bit = hit[j] ? __ffsll((long long)hm) - 1 : 0;
                    hm = hit[j] ? (hm & (hm - 1)) : 0;
                    const int n = n0 + bit;
                    const int pv = s_vox[n];
                    const int d0 = g0 - (pv & 255), d1 = g1 - ((pv >> 8) & 255), d2 = g2 - (pv >> 16);
                    const size_t off = hit[j] ? (size_t)n * KP + ((d0 * k + d1) * k + d2) * kF + part * 4 : 0;
                    x[j] = *reinterpret_cast<const float4*>(dXc + off);
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    acc.x = hit[j] ? acc.x + x[j].x : acc.x; acc.y = hit[j] ? acc.y + x[j].y : acc.y;
                    acc.z = hit[j] ? acc.z + x[j].z : acc.z; acc.w = hit[j] ? acc.w + x[j].w : acc.w;
                }
            }
        }
        *reinterpret_cast<float4*>(dfv + ((size_t)c * G + g) * kF + part * 4) = acc;
    }
}

}  // namespace dpd
