// L1 / mean-prediction losses (+ their gradients) and the TF-form Adam update.  Small HBM-bound kernels.
//
//   dpd_l1_loss  replaces utils/dpdist_util.py:962-980 (get_loss) and TF's autodiff of it
//   dpd_adam_tf  replaces tf.train.AdamOptimizer (train_multi_gpu_pc_compare_dist.py:216,301): epsilon-hat form
//                algorithmic HBM bytes per parameter: 16 read (p,g,m,v) + 12 written (p,m,v)
#include "common.h"

namespace dpd {

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return s;
}

// single block: BN rows of each direction are at most a few 10k
__global__ __launch_bounds__(256) void l1_loss_kernel(const float* __restrict__ pred, const float* __restrict__ labels,
                                                       int BN, int mode, float gscale, float* __restrict__ loss,
                                                       float* __restrict__ dpred) {
    __shared__ float red[4];
    float sl = 0.f, sab = 0.f, sba = 0.f;
    const float inv = 1.0f / (float)BN;
    for (int r = threadIdx.x; r < BN; r += 256) {
        const float pab = pred[(size_t)r * 3];                  // pred_listAB[:,:,:,0]  (:967)
        const float pba = pred[((size_t)BN + r) * 3];           // pred_listBA[:,:,:,0]
        const float d = pab - labels[r];
        sl += fabsf(d);                                         // :972
        sab += pab;
        sba += pba;
        if (mode == 1) {                                        // d mean|pab - label| / d pab  (tf.abs grad = sign)
            const float sg = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
            dpred[(size_t)r * 3] = sg * inv * gscale;
            dpred[(size_t)r * 3 + 1] = 0.f;
            dpred[(size_t)r * 3 + 2] = 0.f;
        } else if (mode == 2) {                                 // d loss_pred / d pred: 0.5/BN on channel 0 of both halves
            const float gv = 0.5f * inv * gscale;
            dpred[(size_t)r * 3] = gv; dpred[(size_t)r * 3 + 1] = 0.f; dpred[(size_t)r * 3 + 2] = 0.f;
            dpred[((size_t)BN + r) * 3] = gv; dpred[((size_t)BN + r) * 3 + 1] = 0.f; dpred[((size_t)BN + r) * 3 + 2] = 0.f;
        }
    }
    sl = block_sum_256(sl, red);
    sab = block_sum_256(sab, red);
    sba = block_sum_256(sba, red);
    if (threadIdx.x == 0) {
        loss[0] = sl * inv;                                     // loss_samples
        loss[1] = (sab * inv + sba * inv) / 2.0f;               // loss_pred  (:976-977)
    }
}

__global__ __launch_bounds__(256) void adam_tf_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, size_t n, float lr_t,
                                                       float b1, float b2, float eps, float gscale) {
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 P = reinterpret_cast<float4*>(p)[i];
        const float4 Gr = reinterpret_cast<const float4*>(g)[i];
        float4 M = reinterpret_cast<float4*>(m)[i];
        float4 V = reinterpret_cast<float4*>(v)[i];
        float* pp = &P.x; const float* gg = &Gr.x; float* mm = &M.x; float* vv = &V.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gr = gg[j] * gscale;
            mm[j] = b1 * mm[j] + (1.0f - b1) * gr;
            vv[j] = b2 * vv[j] + (1.0f - b2) * gr * gr;
            pp[j] = pp[j] - lr_t * mm[j] / (sqrtf(vv[j]) + eps);
        }
        reinterpret_cast<float4*>(p)[i] = P;
        reinterpret_cast<float4*>(m)[i] = M;
        reinterpret_cast<float4*>(v)[i] = V;
    }
    // tail (n not a multiple of 4)
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gr = g[i] * gscale;
        m[i] = b1 * m[i] + (1.0f - b1) * gr;
        v[i] = b2 * v[i] + (1.0f - b2) * gr * gr;
        p[i] = p[i] - lr_t * m[i] / (sqrtf(v[i]) + eps);
    }
}

}  // namespace dpd

extern "C" const char* dpd_version(void) { return "dpdist_hip 0.1 gfx950"; }

extern "C" int dpd_l1_loss(const float* pred, const float* labels, int BN, int mode, float gscale, float* loss,
                           float* dpred, void* stream) {
    if (!pred || !labels || !loss) return DPD_E_NULL;
    if (BN <= 0 || mode < 0 || mode > 2) return DPD_E_DIM;
    if (mode != 0 && !dpred) return DPD_E_NULL;
    DPD_LAUNCH(dpd::l1_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, labels, BN, mode, gscale, loss,
                       dpred);
    DPD_CHECK_LAUNCH();
    return 0;
}

extern "C" int dpd_adam_tf(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2,
                           float eps, float gscale, void* stream) {
    if (!p || !g || !m || !v) return DPD_E_NULL;
    if (n == 0) return DPD_E_DIM;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return DPD_E_UNSUPPORTED;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    DPD_LAUNCH(dpd::adam_tf_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr_t,
                       b1, b2, eps, gscale);
    DPD_CHECK_LAUNCH();
    return 0;
}
