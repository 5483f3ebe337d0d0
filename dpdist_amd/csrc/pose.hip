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
__global__ __launch_bounds__(64) void pose_apply_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ src,
                                                            const float* __restrict__ T_in, int N, float lim_rad, int mode,
                                                            float* __restrict__ pose, float* __restrict__ moved,
                                                            float* __restrict__ T_out) {
    const int b = blockIdx.x;
    const Pose7 P = quat_normalize_dev(pred + (size_t)b * 7, lim_rad);
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
        const float* Ti = T_in + (size_t)b * 16;
        float v;
        if (i < 3) v = ((Rc[i][0] * Ti[0 * 4 + j] + Rc[i][1] * Ti[1 * 4 + j]) + Rc[i][2] * Ti[2 * 4 + j]) + P.t[i] * Ti[3 * 4 + j];
        else v = Ti[3 * 4 + j];
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

}  // namespace dpd

extern "C" int dpd_pose_apply_fwd(const float* pred, const float* src, const float* T_in, int B, int N, float lim_rot_deg, int mode,
                                  float* pose, float* moved, float* T_out, void* stream) {
    if (!pred || (moved && !src) || (T_out && !T_in)) return DPD_E_NULL;
    if (!pose && !moved && !T_out) return DPD_E_NULL;
    if (B <= 0 || N <= 0 || mode < 0 || mode > 1) return DPD_E_DIM;
    const float lim_rad = (float)(3.14159265358979323846 / 180.0 * (double)lim_rot_deg);
    DPD_LAUNCH(dpd::pose_apply_fwd_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, pred, src, T_in, N, lim_rad, mode, pose, moved,
               T_out);
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
