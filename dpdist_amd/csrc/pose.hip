// Pose algebra of the iterative registration (BASELINE config 5) as ONE launch per direction.
//
// Around the DPDist loss the reference's registration step (pcrnet-registration/iterative_PCRNet_ours.py:410-470) runs, eight times per
// step, a chain of ~115 tiny element-wise TensorFlow ops on [B,7] / [B,4,4] tensors:
//   models/ipcr_model.py:285-294   quat_normalize: (t, angle, axis) -> (tanh(t) 0.1, cos(a/2), axis sin(a/2)), |a| <= lim_rot degrees
//   helper.py:309-329              transformation_quat2mat: normalise the quaternion (transforms3d.quat2mat), T <- [R t; 0 1] T, move the cloud
//   helper.py:539-570              transformation_quat_tensor: Besl-McKay quaternion -> R, data R^T + t
//   iterative_PCRNet_ours.py:211-224   the training evaluation: quaternion / (|q| + 1e-7), then transformation_quat_tensor
// With eager PyTorch that is ~115 launches per refinement loop (1000 of the ~1300 launches of a registration step, which is host-bound at
// 9.2 ms for 0.4 ms of DPDist: profiles/r05_registration_engine_ab.txt).  Here: one workgroup per cloud pair does the whole chain.
//   dpd_pose_apply_fwd   raw pose-network output [B,7] + source cloud [B,N,3] (+ T [B,4,4]) -> pose [B,7], moved cloud, T_out
//   dpd_pose_apply_bwd   d moved [B,N,3] -> d raw output [B,7]  (training evaluation only: the refinements carry no gradient, :414-441)
// dpdist_amd/registration.py keeps the same algebra as plain torch functions (pinned to the reference's goldens); tests compare the two.
//
// The forward-only refinements (7 of the 8 pose-network evaluations of a step, :414-441, and all 8 of an evaluation batch) also run the
// pose NETWORK here -- models/ipcr_model.py:198-233 (shared MLP 3-64-64-64-128-1024 + max pool) and :273-284 (fc 2048-1024-512-256-7,
// dropout before the last layer) -- in five launches per loop instead of ~25 (dpd_pose_refine):
//   pose_point_kernel   one workgroup per (cloud, 128-column slice of the last layer): all five layers for the cloud's points in LDS,
//                       max pool in the epilogue; the template's features are computed once per call (the template does not move)
//   pose_fc_kernel      the three wide head layers for <= 16 rows: one wave per 4 output columns streams its weight rows once
//   pose_apply_fwd_kernel   fc4 (256 x 7) as its prologue, then the pose chain above
// fp32 FMA throughout.  The training evaluation keeps torch autograd for the network (its backward is torch's).
#include "common.h"

namespace dpd {

struct Pose7 {
    float t[3];
    float q[4];
};

// models/ipcr_model.py:285-294; lim_rad = pi/180 * lim_rot.  lim_rad == 0: the network's output IS the pose (lim_rot falsy).
__device__ __forceinline__ Pose7 quat_normalize_dev(const float* __restrict__ p, float lim_rad) {
    Pose7 o;
    if (lim_rad == 0.f) {
        o.t[0] = p[0]; o.t[1] = p[1]; o.t[2] = p[2];
        o.q[0] = p[3]; o.q[1] = p[4]; o.q[2] = p[5]; o.q[3] = p[6];
        return o;
    }
    const float ang = tanhf(p[3]) * lim_rad;
    const float r = sqrtf(p[4] * p[4] + p[5] * p[5] + p[6] * p[6]) + 1e-6f;
    const float s = sinf(ang / 2.f);
    o.t[0] = tanhf(p[0]) * 0.1f; o.t[1] = tanhf(p[1]) * 0.1f; o.t[2] = tanhf(p[2]) * 0.1f;
    o.q[0] = cosf(ang / 2.f);
    o.q[1] = p[4] / r * s; o.q[2] = p[5] / r * s; o.q[3] = p[6] / r * s;
    return o;
}

// helper.py:552-554 (no normalisation inside)
__device__ __forceinline__ void quat_to_mat_dev(const float* q, float R[3][3]) {
    const float q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    R[0][0] = q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3; R[0][1] = 2.f * (q1 * q2 - q0 * q3); R[0][2] = 2.f * (q1 * q3 + q0 * q2);
    R[1][0] = 2.f * (q1 * q2 + q0 * q3); R[1][1] = q0 * q0 + q2 * q2 - q1 * q1 - q3 * q3; R[1][2] = 2.f * (q2 * q3 - q0 * q1);
    R[2][0] = 2.f * (q1 * q3 - q0 * q2); R[2][1] = 2.f * (q2 * q3 + q0 * q1); R[2][2] = q0 * q0 + q3 * q3 - q1 * q1 - q2 * q2;
}

// one workgroup (one wave) per cloud pair.  mode 0: refinement loop (helper.transformation_quat2mat: quaternion / max(|q|, 1e-12), the
// moved cloud and T_out use the same normalised pose); mode 1: training evaluation (moved cloud from quaternion / (|q| + 1e-7),
// iterative_PCRNet_ours.py:211-224; T_out -- the step's returned transform -- from the max(|q|, 1e-12) form like every other loop).
// fc4 prologue (h3 != nullptr): pred[b] = W4 [7,K4] h3[b] + b4, K4 % 4 == 0 (models/ipcr_model.py:284); pred_out (optional) receives it.
__global__ __launch_bounds__(64) void pose_apply_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ src,
                                                            const float* __restrict__ T_in, int N, float lim_rad, int mode,
                                                            float* __restrict__ pose, float* __restrict__ moved,
                                                            float* __restrict__ T_out, const float* __restrict__ h3,
                                                            const float* __restrict__ W4, const float* __restrict__ b4, int K4,
                                                            float* __restrict__ pred_out) {
    const int b = blockIdx.x;
    float pr[7];
    if (h3) {
        float acc[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[j] = 0.f;
        for (int k = threadIdx.x * 4; k < K4; k += 256) {
            const float4 x = *reinterpret_cast<const float4*>(h3 + (size_t)b * K4 + k);
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const float4 w = *reinterpret_cast<const float4*>(W4 + (size_t)j * K4 + k);
                acc[j] = fmaf(x.x, w.x, fmaf(x.y, w.y, fmaf(x.z, w.z, fmaf(x.w, w.w, acc[j]))));
            }
        }
#pragma unroll
        for (int j = 0; j < 7; ++j) pr[j] = wave_sum(acc[j]) + b4[j];
        if (pred_out && threadIdx.x < 7) {
            float v = pr[0];
#pragma unroll
            for (int j = 1; j < 7; ++j) v = (int)threadIdx.x == j ? pr[j] : v;
            pred_out[(size_t)b * 7 + threadIdx.x] = v;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 7; ++j) pr[j] = pred[(size_t)b * 7 + j];
    }
    const Pose7 P = quat_normalize_dev(pr, lim_rad);
    const float nrm = sqrtf(P.q[0] * P.q[0] + P.q[1] * P.q[1] + P.q[2] * P.q[2] + P.q[3] * P.q[3]);
    const float dc = fmaxf(nrm, 1e-12f), dt = nrm + 1e-7f;
    float qc[4], qm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { qc[i] = P.q[i] / dc; qm[i] = mode == 1 ? P.q[i] / dt : qc[i]; }
    float R[3][3];
    quat_to_mat_dev(qm, R);
    if (moved) {
        for (int n = threadIdx.x; n < N; n += 64) {
            const float* s = src + ((size_t)b * N + n) * 3;
            const float x = s[0], y = s[1], z = s[2];
            float* o = moved + ((size_t)b * N + n) * 3;
            o[0] = (x * R[0][0] + y * R[0][1] + z * R[0][2]) + P.t[0];
            o[1] = (x * R[1][0] + y * R[1][1] + z * R[1][2]) + P.t[1];
            o[2] = (x * R[2][0] + y * R[2][1] + z * R[2][2]) + P.t[2];
        }
    }
    if (threadIdx.x == 0 && pose) {          // the network's pose as registration.PoseNet returns it (quat_normalize applied, not re-normalised)
        float* o = pose + (size_t)b * 7;
        o[0] = P.t[0]; o[1] = P.t[1]; o[2] = P.t[2]; o[3] = P.q[0]; o[4] = P.q[1]; o[5] = P.q[2]; o[6] = P.q[3];
    }
    if (T_out && threadIdx.x < 16) {         // helper.py:309-329: T <- [R(qc) t; 0 1] @ T
        float Rc[3][3];
        quat_to_mat_dev(qc, Rc);
        const int i = threadIdx.x >> 2, j = threadIdx.x & 3;
        float Tc[4];                              // column j of T_in (nullptr: the identity, the first loop of a refinement)
#pragma unroll
        for (int r = 0; r < 4; ++r) Tc[r] = T_in ? T_in[(size_t)b * 16 + r * 4 + j] : (r == j ? 1.f : 0.f);
        float v;
        if (i < 3) v = ((Rc[i][0] * Tc[0] + Rc[i][1] * Tc[1]) + Rc[i][2] * Tc[2]) + P.t[i] * Tc[3];
        else v = Tc[3];
        T_out[(size_t)b * 16 + threadIdx.x] = v;
    }
}

// d moved [B,N,3] -> d pred [B,7] through moved = src R(u)^T + t, u = q / (|q| + 1e-7), (t, q) = quat_normalize(pred)
__global__ __launch_bounds__(64) void pose_apply_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ src,
                                                            const float* __restrict__ dmoved, int N, float lim_rad,
                                                            float* __restrict__ dpred) {
    const int b = blockIdx.x;
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;
    for (int n = threadIdx.x; n < N; n += 64) {
        const float* s = src + ((size_t)b * N + n) * 3;
        const float* g = dmoved + ((size_t)b * N + n) * 3;
        const float x = s[0], y = s[1], z = s[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float gi = g[i];
            acc[i * 3 + 0] += gi * x; acc[i * 3 + 1] += gi * y; acc[i * 3 + 2] += gi * z;      // dR[i][j] = sum_n dm[n][i] src[n][j]
            acc[9 + i] += gi;                                                                   // dt[i]
        }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = wave_sum(acc[i]);
    if (threadIdx.x != 0) return;
    const float* p = pred + (size_t)b * 7;
    const Pose7 P = quat_normalize_dev(p, lim_rad);
    const float nrm = sqrtf(P.q[0] * P.q[0] + P.q[1] * P.q[1] + P.q[2] * P.q[2] + P.q[3] * P.q[3]);
    const float den = nrm + 1e-7f;
    const float u0 = P.q[0] / den, u1 = P.q[1] / den, u2 = P.q[2] / den, u3 = P.q[3] / den;
    const float (*dR)[3] = reinterpret_cast<const float (*)[3]>(acc);
    // d R(u) / d u, from helper.py:552-554
    float du[4];
    du[0] = 2.f * (u0 * dR[0][0] - u3 * dR[0][1] + u2 * dR[0][2] + u3 * dR[1][0] + u0 * dR[1][1] - u1 * dR[1][2] - u2 * dR[2][0] + u1 * dR[2][1] + u0 * dR[2][2]);
    du[1] = 2.f * (u1 * dR[0][0] + u2 * dR[0][1] + u3 * dR[0][2] + u2 * dR[1][0] - u1 * dR[1][1] - u0 * dR[1][2] + u3 * dR[2][0] + u0 * dR[2][1] - u1 * dR[2][2]);
    du[2] = 2.f * (-u2 * dR[0][0] + u1 * dR[0][1] + u0 * dR[0][2] + u1 * dR[1][0] + u2 * dR[1][1] + u3 * dR[1][2] - u0 * dR[2][0] + u3 * dR[2][1] - u2 * dR[2][2]);
    du[3] = 2.f * (-u3 * dR[0][0] - u0 * dR[0][1] + u1 * dR[0][2] + u0 * dR[1][0] - u3 * dR[1][1] + u2 * dR[1][2] + u1 * dR[2][0] + u2 * dR[2][1] + u3 * dR[2][2]);
    // u = q / (|q| + eps):  dq_k = du_k / den - (du . q) / den^2 * q_k / |q|
    const float dot = du[0] * P.q[0] + du[1] * P.q[1] + du[2] * P.q[2] + du[3] * P.q[3];
    float dq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) dq[k] = du[k] / den - (nrm > 0.f ? dot / (den * den) * (P.q[k] / nrm) : 0.f);
    float* o = dpred + (size_t)b * 7;
    if (lim_rad == 0.f) {
        o[0] = acc[9]; o[1] = acc[10]; o[2] = acc[11]; o[3] = dq[0]; o[4] = dq[1]; o[5] = dq[2]; o[6] = dq[3];
        return;
    }
    // quat_normalize: t = tanh(p0..2) 0.1; a = tanh(p3) lim; ax = p4..6 / (|p4..6| + 1e-6); q = (cos(a/2), ax sin(a/2))
    const float th = tanhf(p[3]);
    const float ang = th * lim_rad;
    const float sn = sinf(ang / 2.f), cs = cosf(ang / 2.f);
    const float r0 = sqrtf(p[4] * p[4] + p[5] * p[5] + p[6] * p[6]);
    const float r = r0 + 1e-6f;
    const float ax[3] = {p[4] / r, p[5] / r, p[6] / r};
    const float da = 0.5f * (cs * (ax[0] * dq[1] + ax[1] * dq[2] + ax[2] * dq[3]) - sn * dq[0]);
    o[3] = da * lim_rad * (1.f - th * th);
    const float dax[3] = {sn * dq[1], sn * dq[2], sn * dq[3]};
    const float dotp = dax[0] * p[4] + dax[1] * p[5] + dax[2] * p[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[4 + k] = dax[k] / r - (r0 > 0.f ? dotp / (r * r) * (p[4 + k] / r0) : 0.f);
        const float tk = tanhf(p[k]);
        o[k] = acc[9 + k] * 0.1f * (1.f - tk * tk);
    }
}

// ---------------------------------------------------------------------------------------------------------------- pose network, forward
constexpr int kPP = 64;                 // points per pass of the shared MLP
constexpr int kSlice = 128;             // columns of the last shared layer per workgroup
// LDS (floats).  Weights arrive by LDS-DMA (global_load_lds_dwordx4: no registers, no wait until they are needed), unpadded, with the
// 16-byte chunks of row r stored at position chunk ^ (r & 15) (the DMA writes 1 KiB contiguously per wave, so the swizzle is applied to
// the SOURCE address; a b128 read of 16 consecutive rows at one k then touches every bank once).  Activations are written by ds_write
// and keep 16-byte-aligned padded rows (KIN + 4 floats).  W5's slice comes in two halves of 64 columns: half 0 has its own 32 KiB,
// half 1 replaces W2 | W3 once layer 3 is done with them and is in flight under layer 4.
constexpr int kS64 = 68, kS128 = 132;
constexpr int kW2 = 0, kW3 = 64 * 64, kW4 = 2 * 64 * 64, kW5a = kW4 + 128 * 64, kW5b = 0;
constexpr int kHA = kW5a + 64 * 128, kHB = kHA + 64 * kS64, kW1 = kHB + 64 * kS128, kBias = kW1 + 192, kPts = kBias + 448, kPointLds = kPts + 192;
static_assert(kPointLds * 4 <= 160 * 1024, "LDS budget");

struct PointNetW {
    const float* W[5];   // [out, in] row-major (torch nn.Linear): 64x3, 64x64, 64x64, 128x64, OUTx128
    const float* b[5];
};

typedef __attribute__((address_space(3))) void* lds_addr_t;

// One 1-KiB LDS-DMA piece: LDS[dst + lane * 16] <- 16 bytes at this lane's source address (as csrc/gemm_shared.h: inline asm, retired
// by an explicit s_waitcnt vmcnt(0) before the barrier in front of the first read).
__device__ __forceinline__ void dma_1k(const void* src, unsigned dst_bytes) {
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
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ROWS x KIN weights (row-major, contiguous in global memory) -> LDS at float offset `off`, swizzled as above; 4 waves share the pieces
template <int KIN, int ROWS>
__device__ __forceinline__ void dma_weights(const float* __restrict__ W, unsigned lds_base, int off, int wave, int lane) {
    constexpr int CPR = KIN / 4;                              // 16-byte chunks per row
    constexpr int PIECES = ROWS * KIN * 4 / 1024;
#pragma unroll
    for (int p = 0; p < PIECES / 4; ++p) {
        const int piece = wave + 4 * p;
        const int g = piece * 64 + lane, r = g / CPR, pos = g % CPR;
        dma_1k(W + (size_t)r * KIN + 4 * (pos ^ (r & 15)), lds_base + (unsigned)(off * 4 + piece * 1024));
    }
}

// One layer on the fp32 matrix cores: C[pt][out] = bias[out] + sum_k h[pt][k] W[out][k] for NPT 32-point tiles (ptile0 ..) x one 32-output
// tile (weight rows wrow0 .. + 31) per wave.  v_mfma_f32_32x32x2_f32: lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31];
// here every lane reads ONE float4 of its h row and one of its W row at columns k0 + 4 (l >> 5) .. + 3 and issues four MFMAs (element s of
// both: the same k on both sides, so the order of k inside the step does not matter) -- 8 k per step, an exact fp32 fmaf chain per output.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int KIN, int NPT>
__device__ __forceinline__ void mfma_layer(const float* __restrict__ hin, const float* __restrict__ Ws, float bj, int ptile0, int wrow0,
                                           f32x16& c0, f32x16& c1) {
    const int l = threadIdx.x & 63, i = l & 31, h = l >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c0[r] = bj; c1[r] = bj; }
    const float* ap = hin + (ptile0 * 32 + i) * (KIN + 4) + 4 * h;
    const float* bp = Ws + (wrow0 + i) * KIN;
    const int sw = (wrow0 + i) & 15;
#pragma unroll 2
    for (int k = 0; k < KIN; k += 8) {
        const float4 b = *reinterpret_cast<const float4*>(bp + 4 * (((k >> 2) + h) ^ sw));
        const float4 a0 = *reinterpret_cast<const float4*>(ap + k);
        float4 a1 = a0;
        if (NPT == 2) a1 = *reinterpret_cast<const float4*>(ap + 32 * (KIN + 4) + k);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, c0, 0, 0, 0);
        if (NPT == 2) c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.x, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, c0, 0, 0, 0);
        if (NPT == 2) c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.y, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, c0, 0, 0, 0);
        if (NPT == 2) c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b.z, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, c0, 0, 0, 0);
        if (NPT == 2) c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b.w, c1, 0, 0, 0);
    }
}

// C/D map of the 32x32 MFMA: register r of lane l is row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31
__device__ __forceinline__ int mfma_row(int r, int l) { return (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); }

template <int KOUT>
__device__ __forceinline__ void store_relu_tile(float* __restrict__ hout, int ptile, int otile, const f32x16& c) {
    const int l = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < 16; ++r) hout[(ptile * 32 + mfma_row(r, l)) * (KOUT + 4) + otile * 32 + (l & 31)] = fmaxf(c[r], 0.f);
}

// models/ipcr_model.py:198-233: cloud c (< nA: ptsA[c], else ptsB[c - nA]) -> f[(row0 + c), slice*128 .. +128) = max over the points of
// relu(W5 relu(W4 relu(W3 relu(W2 relu(W1 p + b1) + b2) + b3) + b4) + b5)
__global__ __launch_bounds__(256) void pose_point_kernel(const float* __restrict__ ptsA, const float* __restrict__ ptsB, int nA, int N,
                                                         PointNetW net, int OUT, int row0, float* __restrict__ f) {
    extern __shared__ float lds[];
    const int c = blockIdx.x, slice = blockIdx.y, t = threadIdx.x, og = t & 15, pg = t >> 4, l = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_addr_t)lds;
    const float* pts = c < nA ? ptsA + (size_t)c * N * 3 : ptsB + (size_t)(c - nA) * N * 3;
    const float* W5s = net.W[4] + (size_t)slice * kSlice * 128;
    float vmax = 0.f;                                          // column wv * 32 + (l & 31) of the slice, over this lane's rows; relu outputs are >= 0
    dma_weights<64, 64>(net.W[1], lds_base, kW2, wv, l);
    dma_weights<64, 64>(net.W[2], lds_base, kW3, wv, l);
    dma_weights<64, 128>(net.W[3], lds_base, kW4, wv, l);
    dma_weights<128, 64>(W5s, lds_base, kW5a, wv, l);
    // the biases and W1 once
    if (t < 192) lds[kW1 + t] = net.W[0][t];
    if (t < 64) { lds[kBias + t] = net.b[0][t]; lds[kBias + 64 + t] = net.b[1][t]; lds[kBias + 128 + t] = net.b[2][t]; }
    if (t < 128) { lds[kBias + 192 + t] = net.b[3][t]; lds[kBias + 320 + t] = net.b[4][slice * kSlice + t]; }
    for (int p0 = 0; p0 < N; p0 += kPP) {
        const int np = min(kPP, N - p0);
        if (p0 > 0) {
            __syncthreads();                                   // the previous pass is done with W5's second half: W2 | W3 come back
            dma_weights<64, 64>(net.W[1], lds_base, kW2, wv, l);
            dma_weights<64, 64>(net.W[2], lds_base, kW3, wv, l);
        }
        if (t < 192) lds[kPts + t] = t < np * 3 ? pts[(size_t)p0 * 3 + t] : 0.f;
        dma_wait();
        __syncthreads();
        // layer 1: 3 -> 64 into hA (K = 3: vector ALU; 4 points x 4 outputs per thread)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x = lds[kPts + (pg * 4 + i) * 3], y = lds[kPts + (pg * 4 + i) * 3 + 1], z = lds[kPts + (pg * 4 + i) * 3 + 2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = og + 16 * j;
                const float v = fmaf(z, lds[kW1 + o * 3 + 2], fmaf(y, lds[kW1 + o * 3 + 1], fmaf(x, lds[kW1 + o * 3], lds[kBias + o])));
                lds[kHA + (pg * 4 + i) * kS64 + o] = fmaxf(v, 0.f);
            }
        }
        __syncthreads();
        f32x16 c0, c1;
        // layer 2: hA -> hB (64 wide): wave = (point tile wv & 1, output tile wv >> 1)
        mfma_layer<64, 1>(lds + kHA, lds + kW2, lds[kBias + 64 + (wv >> 1) * 32 + (l & 31)], wv & 1, (wv >> 1) * 32, c0, c1);
        store_relu_tile<64>(lds + kHB, wv & 1, wv >> 1, c0);
        __syncthreads();
        // layer 3: hB -> hA
        mfma_layer<64, 1>(lds + kHB, lds + kW3, lds[kBias + 128 + (wv >> 1) * 32 + (l & 31)], wv & 1, (wv >> 1) * 32, c0, c1);
        store_relu_tile<64>(lds + kHA, wv & 1, wv >> 1, c0);
        __syncthreads();                                       // W2 | W3 are done with: the second half of W5's slice takes their place,
        dma_weights<128, 64>(W5s + 64 * 128, lds_base, kW5b, wv, l);      // in flight under layer 4
        // layer 4: hA -> hB (128 wide): wave = output tile wv, both point tiles
        mfma_layer<64, 2>(lds + kHA, lds + kW4, lds[kBias + 192 + wv * 32 + (l & 31)], 0, wv * 32, c0, c1);
        store_relu_tile<128>(lds + kHB, 0, wv, c0);
        store_relu_tile<128>(lds + kHB, 1, wv, c1);
        dma_wait();
        __syncthreads();
        // layer 5 (slice): waves 0, 1 on the first half of the slice, 2, 3 on the second; max over this lane's valid points
        mfma_layer<128, 2>(lds + kHB, lds + (wv < 2 ? kW5a : kW5b), lds[kBias + 320 + wv * 32 + (l & 31)], 0, (wv & 1) * 32, c0, c1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (mfma_row(r, l) < np) vmax = fmaxf(vmax, c0[r]);          // = max(relu(.)): vmax starts at 0
            if (32 + mfma_row(r, l) < np) vmax = fmaxf(vmax, c1[r]);
        }
    }
    __syncthreads();
    float* red = lds + kHA;                                    // [2 lane halves][128 columns]
    red[(l >> 5) * kSlice + wv * 32 + (l & 31)] = vmax;
    __syncthreads();
    if (t < kSlice) f[(size_t)(row0 + c) * OUT + slice * kSlice + t] = fmaxf(red[t], red[kSlice + t]);
}

// out[r, j] = act(sum_k in[r, k] W[j, k] + bias[j]) (* mask[r, j]) for <= 16 rows per blockIdx.y; in = [inA (KA columns) | inB (K - KA)]
// (the head's first layer reads cat(source feature, template feature) without materialising it).
// One workgroup of 16 waves per 16 output columns: wave w takes K / 16 of the reduction, NIT = K / 256 steps of 16 k each; per step every
// lane loads ONE float4 of W (row j0 + lane % 16, columns k + 4 (lane / 16) .. + 3) and one of `in` (row lane % 16, same columns) and
// issues four v_mfma_f32_16x16x4_f32 (element s of both float4s: A[i][kk] = W[j0 + i][.], B[kk][n] = in[n][.], same k on both sides).
// All 2 NIT loads of a lane are in flight at once; the 16 partial tiles are added in wave order through LDS (deterministic).
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NIT>
__global__ __launch_bounds__(1024) void pose_fc_kernel(const float* __restrict__ inA, const float* __restrict__ inB, int KA,
                                                       const float* __restrict__ W, const float* __restrict__ bias, int J, int R, int relu,
                                                       const float* __restrict__ mask, float* __restrict__ out) {
    constexpr int K = NIT * 256;
    __shared__ float part[16][256];
    const int j0 = blockIdx.x * 16, r0 = blockIdx.y * 16, wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int n = l & 15, kk = l >> 4;
    const int row = min(r0 + n, R - 1);
    const int kw = wv * (K / 16);                              // this wave's K range: [kw, kw + 16 NIT)
    const float* x = kw < KA ? inA + (size_t)row * KA + kw : inB + (size_t)row * (K - KA) + (kw - KA);
    const float* w = W + (size_t)(j0 + n) * K + kw;
    float4 wr[NIT], xr[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        wr[i] = *reinterpret_cast<const float4*>(w + 16 * i + 4 * kk);
        xr[i] = *reinterpret_cast<const float4*>(x + 16 * i + 4 * kk);
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[i].x, xr[i].x, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[i].y, xr[i].y, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[i].z, xr[i].z, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[i].w, xr[i].w, c, 0, 0, 0);
    }
    // lane l holds C[j = 4 (l / 16) + r][n = l % 16], r = 0..3
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wv][l * 4 + r] = c[r];
    __syncthreads();
    if (threadIdx.x < 256) {
        const int e = threadIdx.x;
        float v = part[0][e];
#pragma unroll
        for (int q = 1; q < 16; ++q) v += part[q][e];
        const int el = e >> 2, er = e & 3;
        const int en = el & 15, ej = j0 + 4 * (el >> 4) + er, orow = r0 + en;
        if (orow < R && ej < J) {
            v += bias[ej];
            if (relu) v = fmaxf(v, 0.f);
            if (mask) v *= mask[(size_t)orow * J + ej];
            out[(size_t)orow * J + ej] = v;
        }
    }
}

}  // namespace dpd

extern "C" int dpd_pose_apply_fwd(const float* pred, const float* src, const float* T_in, int B, int N, float lim_rot_deg, int mode,
                                  float* pose, float* moved, float* T_out, void* stream) {
    if (!pred || (moved && !src)) return DPD_E_NULL;         // T_in == NULL with T_out: composed onto the identity
    if (!pose && !moved && !T_out) return DPD_E_NULL;
    if (T_out && T_out == T_in) return DPD_E_UNSUPPORTED;
    if (B <= 0 || N <= 0 || mode < 0 || mode > 1) return DPD_E_DIM;
    const float lim_rad = (float)(3.14159265358979323846 / 180.0 * (double)lim_rot_deg);
    DPD_LAUNCH(dpd::pose_apply_fwd_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, pred, src, T_in, N, lim_rad, mode, pose, moved,
               T_out, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, (float*)nullptr);
    DPD_CHECK_LAUNCH();
    return 0;
}

extern "C" int dpd_pose_apply_bwd(const float* pred, const float* src, const float* dmoved, int B, int N, float lim_rot_deg, float* dpred,
                                  void* stream) {
    if (!pred || !src || !dmoved || !dpred) return DPD_E_NULL;
    if (B <= 0 || N <= 0) return DPD_E_DIM;
    const float lim_rad = (float)(3.14159265358979323846 / 180.0 * (double)lim_rot_deg);
    DPD_LAUNCH(dpd::pose_apply_bwd_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, pred, src, dmoved, N, lim_rad, dpred);
    DPD_CHECK_LAUNCH();
    return 0;
}

namespace {

struct RefineWs {
    float *f, *h1, *h2, *h3, *cloud[2], *T[2];
    size_t total;
};

RefineWs refine_ws(float* base, int B, int N, int OUT) {
    RefineWs w{};
    size_t off = 0;
    auto take = [&](size_t n) { float* p = base ? base + off : nullptr; off += (n + 63) / 64 * 64; return p; };
    w.f = take((size_t)2 * B * OUT); w.h1 = take((size_t)B * 1024); w.h2 = take((size_t)B * 512); w.h3 = take((size_t)B * 256);
    w.cloud[0] = take((size_t)B * N * 3); w.cloud[1] = take((size_t)B * N * 3); w.T[0] = take((size_t)B * 16); w.T[1] = take((size_t)B * 16);
    w.total = off * sizeof(float);
    return w;
}

dpd::LdsOptIn g_point_lds;

}  // namespace

extern "C" size_t dpd_pose_refine_workspace_bytes(int B, int N, int out_features) {
    if (B <= 0 || N <= 0 || out_features <= 0) return 0;
    return refine_ws(nullptr, B, N, out_features).total;
}

extern "C" int dpd_pose_refine(const dpd_pose_net* net, const float* src, const float* tmpl, int B, int N, int loops, float lim_rot_deg,
                               const float* drop_mask, void* ws, size_t ws_bytes, float* moved, float* T_out, float* pred_out, void* stream) {
    using namespace dpd;
    if (!net || !src || !tmpl || !ws || !moved || !T_out) return DPD_E_NULL;
    for (int i = 0; i < 5; ++i)
        if (!net->Wp[i] || !net->bp[i]) return DPD_E_NULL;
    for (int i = 0; i < 4; ++i)
        if (!net->Wh[i] || !net->bh[i]) return DPD_E_NULL;
    if (B <= 0 || N <= 0 || loops <= 0) return DPD_E_DIM;
    const int OUT = net->out_features;
    if (OUT != 1024) return DPD_E_UNSUPPORTED;             // the reference's width (models/ipcr_model.py:226); the head kernel's K is a template parameter
    if (((uintptr_t)ws & 15) != 0) return DPD_E_UNSUPPORTED;
    const RefineWs w = refine_ws((float*)ws, B, N, OUT);
    if (ws_bytes < w.total) return DPD_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const float lim_rad = (float)(3.14159265358979323846 / 180.0 * (double)lim_rot_deg);
    const size_t lds = (size_t)kPointLds * sizeof(float);
    if (int rc = ensure_dyn_lds(g_point_lds, (const void*)pose_point_kernel, lds)) return rc;
    PointNetW pw{};
    for (int i = 0; i < 5; ++i) { pw.W[i] = net->Wp[i]; pw.b[i] = net->bp[i]; }
    for (int it = 0; it < loops; ++it) {
        const float* cur = it == 0 ? src : w.cloud[(it - 1) & 1];
        const float* Tin = it == 0 ? nullptr : w.T[(it - 1) & 1];
        const bool last = it == loops - 1;
        float* nxt = last ? moved : w.cloud[it & 1];
        float* Tn = last ? T_out : w.T[it & 1];
        // shared MLP + max pool: source features every loop; the template's once (the template never moves)
        if (it == 0) {
            DPD_LAUNCH(pose_point_kernel, dim3((unsigned)(2 * B), (unsigned)(OUT / kSlice)), dim3(256), lds, s, cur, tmpl, B, N, pw, OUT, 0, w.f);
        } else {
            DPD_LAUNCH(pose_point_kernel, dim3((unsigned)B, (unsigned)(OUT / kSlice)), dim3(256), lds, s, cur, (const float*)nullptr, B, N, pw, OUT, 0,
                       w.f);
        }
        DPD_CHECK_LAUNCH();
        const unsigned ry = (unsigned)((B + 15) / 16);
        const float* dm = drop_mask ? drop_mask + (size_t)it * B * 256 : (const float*)nullptr;
        if (OUT == 1024) {
            DPD_LAUNCH(pose_fc_kernel<8>, dim3(1024 / 16, ry), dim3(1024), 0, s, (const float*)w.f, (const float*)(w.f + (size_t)B * OUT), OUT,
                       net->Wh[0], net->bh[0], 1024, B, 1, (const float*)nullptr, w.h1);
        } else {      // other feature widths (multiples of 128): the same kernel, K = 2 OUT in 256-column steps
            return DPD_E_UNSUPPORTED;
        }
        DPD_CHECK_LAUNCH();
        DPD_LAUNCH(pose_fc_kernel<4>, dim3(512 / 16, ry), dim3(1024), 0, s, (const float*)w.h1, (const float*)nullptr, 1024, net->Wh[1], net->bh[1],
                   512, B, 1, (const float*)nullptr, w.h2);
        DPD_CHECK_LAUNCH();
        DPD_LAUNCH(pose_fc_kernel<2>, dim3(256 / 16, ry), dim3(1024), 0, s, (const float*)w.h2, (const float*)nullptr, 512, net->Wh[2], net->bh[2], 256,
                   B, 1, dm, w.h3);
        DPD_CHECK_LAUNCH();
        DPD_LAUNCH(pose_apply_fwd_kernel, dim3((unsigned)B), dim3(64), 0, s, (const float*)nullptr, cur, Tin, N, lim_rad, 0, (float*)nullptr, nxt,
                   Tn, (const float*)w.h3, net->Wh[3], net->bh[3], 256, pred_out ? pred_out + (size_t)it * B * 7 : (float*)nullptr);
        DPD_CHECK_LAUNCH();
    }
    return 0;
}
