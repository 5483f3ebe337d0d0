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

// Optimizer schedule on the device, so that a captured (hipGraph) training step needs no per-step host parameters.
// state (8 floats, caller-owned, zero-filled = "before the first step" once state[1] = state[2] = 1):
//   [0] global step (int32 bits)   [1] beta1_power   [2] beta2_power   [3] lr_t of the step being taken   [4] its learning rate
// One thread: lr = max(base * rate^floor(step / decay_step), floor) on the step counter BEFORE the increment
// (train_multi_gpu_pc_compare_dist.py:976-990, exponential_decay(staircase=True) + tf.maximum), the beta powers as the running
// fp32 products TensorFlow keeps in its beta1_power / beta2_power variables, lr_t = lr sqrt(1 - b2^t) / (1 - b1^t).
__global__ void adam_sched_kernel(float* __restrict__ st, float base_lr, int decay_step, float decay_rate, float floor_lr,
                                  float b1, float b2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int step = __float_as_int(st[0]);
    const int n = decay_step > 0 ? step / decay_step : 0;
    float lr = base_lr;
    for (int i = 0; i < n && lr > 0.f; ++i) lr *= decay_rate;
    lr = fmaxf(lr, floor_lr);
    const float b1p = st[1] * b1, b2p = st[2] * b2;
    st[0] = __int_as_float(step + 1);
    st[1] = b1p;
    st[2] = b2p;
    st[3] = lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
    st[4] = lr;
}

__global__ __launch_bounds__(256) void adam_tf_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, size_t n, const float* __restrict__ st, float b1,
                                                           float b2, float eps, float gscale) {
    const float lr_t = st[3];
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
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gr = g[i] * gscale;
        m[i] = b1 * m[i] + (1.0f - b1) * gr;
        v[i] = b2 * v[i] + (1.0f - b2) * gr * gr;
        p[i] = p[i] - lr_t * m[i] / (sqrtf(v[i]) + eps);
    }
}

}  // namespace dpd

extern "C" int dpd_adam_sched(float* state, float base_lr, int decay_step, float decay_rate, float floor_lr, float b1, float b2,
                              void* stream) {
    if (!state) return DPD_E_NULL;
    DPD_LAUNCH(dpd::adam_sched_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, base_lr, decay_step, decay_rate, floor_lr, b1, b2);
    DPD_CHECK_LAUNCH();
    return 0;
}

extern "C" int dpd_adam_tf_dev(float* p, const float* g, float* m, float* v, size_t n, const float* state, float b1, float b2,
                               float eps, float gscale, void* stream) {
    if (!p || !g || !m || !v || !state) return DPD_E_NULL;
    if (n == 0) return DPD_E_DIM;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return DPD_E_UNSUPPORTED;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    DPD_LAUNCH(dpd::adam_tf_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, state, b1, b2, eps,
               gscale);
    DPD_CHECK_LAUNCH();
    return 0;
}

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
