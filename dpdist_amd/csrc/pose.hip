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
// dropout before the last layer) -- in four launches per loop (+ two per call) instead of ~25 (dpd_pose_refine):
//   pose_point_kernel   one workgroup per (cloud, 128-column slice of the last layer): all five layers for the cloud's points in LDS,
//                       max pool in the epilogue; the template's features are computed once per call (the template does not move)
//   pose_fc_kernel      the three wide head layers for <= 16 rows on v_mfma_f32_16x16x4_f32: workgroup = 16 outputs, sixteen waves split K;
//                       the template's half of fc1's sum once per call (`rowbias`), every loop contracts the source's half only
//   pose_apply_fwd_kernel   fc4 (256 x 7) as its prologue, then the pose chain above: the LAST loop's pose only -- the pose of every other loop
//                       (fc4, chain, move, T composition) is the prologue of the NEXT loop's pose_point_kernel (PoseMove), not a launch
// fp32 throughout (MFMA fp32 / FMA).  The training evaluation of the network and its backward: further down (round 6).
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
    // what the tail needs from global memory is requested here, before the head's products (N <= 64: one point per lane, else the loop below)
    const int n0 = threadIdx.x;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    if (moved && n0 < N) { const float* s = src + ((size_t)b * N + n0) * 3; sx = s[0]; sy = s[1]; sz = s[2]; }
    float Tc[4] = {0.f, 0.f, 0.f, 0.f};              // column j of T_in (nullptr: the identity, the first loop of a refinement), lanes 0..15
    if (T_out && threadIdx.x < 16) {
        const int j = threadIdx.x & 3;
#pragma unroll
        for (int r = 0; r < 4; ++r) Tc[r] = T_in ? T_in[(size_t)b * 16 + r * 4 + j] : (r == j ? 1.f : 0.f);
    }
    float pr[7];
    if (h3) {
        float bb[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) bb[j] = b4[j];
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
        for (int j = 0; j < 7; ++j) pr[j] = wave_sum(acc[j]) + bb[j];
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
        for (int n = n0; n < N; n += 64) {
            float x = sx, y = sy, z = sz;
            if (n != n0) { const float* s = src + ((size_t)b * N + n) * 3; x = s[0]; y = s[1]; z = s[2]; }
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
        const int i = threadIdx.x >> 2;
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
// TRAIN (the training evaluation, dpd_pose_point_fwd_train): the slice-0 workgroup of a cloud also stores the four hidden activations, and every
// workgroup the TIE MASK of its 128 columns -- bit p of ties[cloud][column] is set iff point p attains the column's maximum and that maximum
// is positive (relu' = 0 at 0): what the gradient of reduce_max needs (tf.reduce_max / torch.amax share it evenly among ties).  N <= 64.
struct PointSave {
    float* h[4];                    // [clouds * N, 64] x 3, [clouds * N, 128]
    unsigned long long* ties;       // [clouds, OUT]
};
// Refinement loops 2..n (dpd_pose_refine): the cloud a workgroup reads is the PREVIOUS loop's source moved by the pose the head made of it --
// fc4 (pred = W4 h3 + b4), quat_normalize, R, the move and the T composition are a prologue of the next loop's shared MLP instead of a launch
// of their own (pose_apply_fwd_kernel: 5-9 us per loop for a microsecond of work).  Every (cloud, slice) workgroup recomputes its pair's pose
// (wave 0: the apply kernel's own instruction sequence, so the same bits) while the weight images are in flight; slice 0 stores the moved
// cloud, T and the raw prediction.  h3 == nullptr: no move (first loop, training evaluation).
struct PoseMove {
    const float* h3;                // [B, K4] activations of the head's last hidden layer (previous loop)
    const float* W4;                // [7, K4]
    const float* b4;                // [7]
    const float* T_in;              // [B, 16] or nullptr (identity)
    float* moved;                   // [B, N, 3]
    float* T_out;                   // [B, 16]
    float* pred_out;                // [B, 7] or nullptr
    float lim_rad;
    int K4;
};
template <bool TRAIN>
__device__ __forceinline__ void save_rows(float* __restrict__ dst, const float* __restrict__ src_lds, int np, int W, int stride, int t) {
    for (int e = t; e < np * (W / 4); e += 256) {
        const int r = e / (W / 4), c4 = e % (W / 4);
        *reinterpret_cast<float4*>(dst + (size_t)r * W + 4 * c4) = *reinterpret_cast<const float4*>(src_lds + r * stride + 4 * c4);
    }
}
template <bool TRAIN>
__global__ __launch_bounds__(256) void pose_point_kernel(const float* __restrict__ ptsA, const float* __restrict__ ptsB, int nA, int N,
                                                         PointNetW net, int OUT, int row0, float* __restrict__ f, PointSave sv, PoseMove mv) {
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
    const bool move = !TRAIN && mv.h3 != nullptr && c < nA;
    float R[3][3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}}, tr[3] = {0.f, 0.f, 0.f};
    if (move) {
        float* prs = lds + kHA;                                // scratch until layer 1 writes hA (after the loop's first barrier)
        if (wv == 0) {                                         // pose_apply_fwd_kernel's fc4, lane for lane
            float bb[7], acc[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) { bb[j] = mv.b4[j]; acc[j] = 0.f; }
            for (int k = l * 4; k < mv.K4; k += 256) {
                const float4 x = *reinterpret_cast<const float4*>(mv.h3 + (size_t)c * mv.K4 + k);
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const float4 w = *reinterpret_cast<const float4*>(mv.W4 + (size_t)j * mv.K4 + k);
                    acc[j] = fmaf(x.x, w.x, fmaf(x.y, w.y, fmaf(x.z, w.z, fmaf(x.w, w.w, acc[j]))));
                }
            }
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const float v = wave_sum(acc[j]) + bb[j];
                if (l == 0) prs[j] = v;
            }
        }
        __syncthreads();
        float pr[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) pr[j] = prs[j];
        const Pose7 P = quat_normalize_dev(pr, mv.lim_rad);
        const float nrm = sqrtf(P.q[0] * P.q[0] + P.q[1] * P.q[1] + P.q[2] * P.q[2] + P.q[3] * P.q[3]);
        const float dc = fmaxf(nrm, 1e-12f);
        float qc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) qc[i] = P.q[i] / dc;
        quat_to_mat_dev(qc, R);
        tr[0] = P.t[0]; tr[1] = P.t[1]; tr[2] = P.t[2];
        if (slice == 0) {
            if (mv.pred_out && t < 7) mv.pred_out[(size_t)c * 7 + t] = pr[t];
            if (t < 16) {                                      // helper.py:309-329: T <- [R(qc) t; 0 1] @ T
                const int i = t >> 2, j = t & 3;
                float Tc[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) Tc[r] = mv.T_in ? mv.T_in[(size_t)c * 16 + r * 4 + j] : (r == j ? 1.f : 0.f);
                float v;
                if (i < 3) v = ((R[i][0] * Tc[0] + R[i][1] * Tc[1]) + R[i][2] * Tc[2]) + P.t[i] * Tc[3];
                else v = Tc[3];
                mv.T_out[(size_t)c * 16 + t] = v;
            }
        }
    }
    for (int p0 = 0; p0 < N; p0 += kPP) {
        const int np = min(kPP, N - p0);
        if (p0 > 0) {
            __syncthreads();                                   // the previous pass is done with W5's second half: W2 | W3 come back
            dma_weights<64, 64>(net.W[1], lds_base, kW2, wv, l);
            dma_weights<64, 64>(net.W[2], lds_base, kW3, wv, l);
        }
        if (!move) {
            if (t < 192) lds[kPts + t] = t < np * 3 ? pts[(size_t)p0 * 3 + t] : 0.f;
        } else if (t < kPP) {                                  // one point per thread: moved like pose_apply_fwd_kernel moves it (mode 0)
            float o0 = 0.f, o1 = 0.f, o2 = 0.f;
            if (t < np) {
                const float* sp = pts + (size_t)(p0 + t) * 3;
                const float x = sp[0], y = sp[1], z = sp[2];
                o0 = (x * R[0][0] + y * R[0][1] + z * R[0][2]) + tr[0];
                o1 = (x * R[1][0] + y * R[1][1] + z * R[1][2]) + tr[1];
                o2 = (x * R[2][0] + y * R[2][1] + z * R[2][2]) + tr[2];
                if (slice == 0) {
                    float* o = mv.moved + ((size_t)c * N + p0 + t) * 3;
                    o[0] = o0; o[1] = o1; o[2] = o2;
                }
            }
            lds[kPts + t * 3] = o0; lds[kPts + t * 3 + 1] = o1; lds[kPts + t * 3 + 2] = o2;
        }
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
        if (TRAIN && slice == 0) save_rows<TRAIN>(sv.h[0] + ((size_t)(row0 + c) * N + p0) * 64, lds + kHA, np, 64, kS64, t);
        f32x16 c0, c1;
        // layer 2: hA -> hB (64 wide): wave = (point tile wv & 1, output tile wv >> 1)
        mfma_layer<64, 1>(lds + kHA, lds + kW2, lds[kBias + 64 + (wv >> 1) * 32 + (l & 31)], wv & 1, (wv >> 1) * 32, c0, c1);
        store_relu_tile<64>(lds + kHB, wv & 1, wv >> 1, c0);
        __syncthreads();
        if (TRAIN && slice == 0) save_rows<TRAIN>(sv.h[1] + ((size_t)(row0 + c) * N + p0) * 64, lds + kHB, np, 64, kS64, t);
        // layer 3: hB -> hA
        mfma_layer<64, 1>(lds + kHB, lds + kW3, lds[kBias + 128 + (wv >> 1) * 32 + (l & 31)], wv & 1, (wv >> 1) * 32, c0, c1);
        store_relu_tile<64>(lds + kHA, wv & 1, wv >> 1, c0);
        __syncthreads();                                       // W2 | W3 are done with: the second half of W5's slice takes their place,
        if (TRAIN && slice == 0) save_rows<TRAIN>(sv.h[2] + ((size_t)(row0 + c) * N + p0) * 64, lds + kHA, np, 64, kS64, t);
        dma_weights<128, 64>(W5s + 64 * 128, lds_base, kW5b, wv, l);      // in flight under layer 4
        // layer 4: hA -> hB (128 wide): wave = output tile wv, both point tiles
        mfma_layer<64, 2>(lds + kHA, lds + kW4, lds[kBias + 192 + wv * 32 + (l & 31)], 0, wv * 32, c0, c1);
        store_relu_tile<128>(lds + kHB, 0, wv, c0);
        store_relu_tile<128>(lds + kHB, 1, wv, c1);
        dma_wait();
        __syncthreads();
        if (TRAIN && slice == 0) save_rows<TRAIN>(sv.h[3] + ((size_t)(row0 + c) * N + p0) * 128, lds + kHB, np, 128, kS128, t);
        // layer 5 (slice): waves 0, 1 on the first half of the slice, 2, 3 on the second; max over this lane's valid points
        mfma_layer<128, 2>(lds + kHB, lds + (wv < 2 ? kW5a : kW5b), lds[kBias + 320 + wv * 32 + (l & 31)], 0, (wv & 1) * 32, c0, c1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (mfma_row(r, l) < np) vmax = fmaxf(vmax, c0[r]);          // = max(relu(.)): vmax starts at 0
            if (32 + mfma_row(r, l) < np) vmax = fmaxf(vmax, c1[r]);
        }
        if (TRAIN) {        // (N <= kPP: one pass, the accumulators of every point are still in registers)
            __syncthreads();
            float* red = lds + kHA;
            red[(l >> 5) * kSlice + wv * 32 + (l & 31)] = vmax;
            __syncthreads();
            const int col = wv * 32 + (l & 31);
            const float m = fmaxf(red[col], red[kSlice + col]);
            unsigned lo = 0, hi = 0;                           // rows 0..31 come from c0, rows 32..63 from c1
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mfma_row(r, l);
                if (row < np && m > 0.f && c0[r] == m) lo |= 1u << row;
                if (32 + row < np && m > 0.f && c1[r] == m) hi |= 1u << row;
            }
            lo |= (unsigned)__shfl_xor((int)lo, 32, 64);       // the two lane halves hold the other rows of the same column
            hi |= (unsigned)__shfl_xor((int)hi, 32, 64);
            if ((l >> 5) == 0) sv.ties[(size_t)(row0 + c) * OUT + slice * kSlice + col] = ((unsigned long long)hi << 32) | lo;
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
                                                       const float* __restrict__ mask, float* __restrict__ out, int ldw,
                                                       const float* __restrict__ rowbias) {
    constexpr int K = NIT * 256;
    __shared__ float part[16][256];
    const int j0 = blockIdx.x * 16, r0 = blockIdx.y * 16, wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int n = l & 15, kk = l >> 4;
    const int row = min(r0 + n, R - 1);
    const int kw = wv * (K / 16);                              // this wave's K range: [kw, kw + 16 NIT)
    const float* x = kw < KA ? inA + (size_t)row * KA + kw : inB + (size_t)row * (K - KA) + (kw - KA);
    const float* w = W + (size_t)(j0 + n) * ldw + kw;       // ldw >= K: a column block of a wider weight matrix
    // the epilogue's element (threads 0..255) and what it needs from global memory, requested with the operands
    const int e = threadIdx.x, el = e >> 2, er = e & 3, en = el & 15, ej = j0 + 4 * (el >> 4) + er, orow = r0 + en;
    const bool live = e < 256 && orow < R && ej < J;
    const float bj = live ? (rowbias ? rowbias[(size_t)orow * J + ej] : bias[ej]) : 0.f;      // rowbias [R][J]: a precomputed part of the sum
    const float mk = (live && mask) ? mask[(size_t)orow * J + ej] : 1.f;
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
    if (live) {
        float v = part[0][e];
#pragma unroll
        for (int q = 1; q < 16; ++q) v += part[q][e];
        v += bj;
        if (relu) v = fmaxf(v, 0.f);
        if (mask) v *= mk;
        out[(size_t)orow * J + ej] = v;
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
    float *f, *h1, *h2, *h3, *cloud[2], *T[2], *tb;
    size_t total;
};

RefineWs refine_ws(float* base, int B, int N, int OUT) {
    RefineWs w{};
    size_t off = 0;
    auto take = [&](size_t n) { float* p = base ? base + off : nullptr; off += (n + 63) / 64 * 64; return p; };
    w.f = take((size_t)2 * B * OUT); w.h1 = take((size_t)B * 1024); w.h2 = take((size_t)B * 512); w.h3 = take((size_t)B * 256);
    w.cloud[0] = take((size_t)B * N * 3); w.cloud[1] = take((size_t)B * N * 3); w.T[0] = take((size_t)B * 16); w.T[1] = take((size_t)B * 16);
    w.tb = take((size_t)B * 1024);
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
    if (int rc = ensure_dyn_lds(g_point_lds, (const void*)pose_point_kernel<false>, lds)) return rc;
    PointNetW pw{};
    for (int i = 0; i < 5; ++i) { pw.W[i] = net->Wp[i]; pw.b[i] = net->bp[i]; }
    // loop it: features of source_it (and of the template, once) -> head up to its last hidden layer h3_it.  The pose of loop it - 1 -- fc4,
    // quat_normalize, the move source_{it-1} -> source_it, T <- M T -- is the PROLOGUE of loop it's shared MLP (PoseMove); the last loop's pose is
    // the one pose_apply_fwd_kernel launch of the call: 4 loops + 1 launches.
    const float* cur = src;                 // source_{it-1} while loop it is being enqueued
    const float* Tcur = nullptr;
    for (int it = 0; it < loops; ++it) {
        if (it == 0) {
            DPD_LAUNCH(pose_point_kernel<false>, dim3((unsigned)(2 * B), (unsigned)(OUT / kSlice)), dim3(256), lds, s, cur, tmpl, B, N, pw, OUT, 0, w.f, PointSave{},
                       PoseMove{});
        } else {
            float* nxt = w.cloud[(it - 1) & 1];
            float* Tn = w.T[(it - 1) & 1];
            const PoseMove mv{w.h3, net->Wh[3], net->bh[3], Tcur, nxt, Tn, pred_out ? pred_out + (size_t)(it - 1) * B * 7 : (float*)nullptr, lim_rad, 256};
            DPD_LAUNCH(pose_point_kernel<false>, dim3((unsigned)B, (unsigned)(OUT / kSlice)), dim3(256), lds, s, cur, (const float*)nullptr, B, N, pw, OUT, 0,
                       w.f, PointSave{}, mv);
            cur = nxt;
            Tcur = Tn;
        }
        DPD_CHECK_LAUNCH();
        const unsigned ry = (unsigned)((B + 15) / 16);
        const float* dm = drop_mask ? drop_mask + (size_t)it * B * 256 : (const float*)nullptr;
        if (it == 0) {
            // the template never moves: its half of fc1's sum (W1[:, OUT:] f_tmpl + b1) once per call, every loop then contracts the source's
            // half only (4 MB of weights per loop instead of 8)
            DPD_LAUNCH(pose_fc_kernel<4>, dim3(1024 / 16, ry), dim3(1024), 0, s, (const float*)(w.f + (size_t)B * OUT), (const float*)nullptr, OUT,
                       net->Wh[0] + OUT, net->bh[0], 1024, B, 0, (const float*)nullptr, w.tb, 2 * OUT, (const float*)nullptr);
            DPD_CHECK_LAUNCH();
        }
        DPD_LAUNCH(pose_fc_kernel<4>, dim3(1024 / 16, ry), dim3(1024), 0, s, (const float*)w.f, (const float*)nullptr, OUT, net->Wh[0], net->bh[0], 1024, B, 1,
                   (const float*)nullptr, w.h1, 2 * OUT, (const float*)w.tb);
        DPD_CHECK_LAUNCH();
        DPD_LAUNCH(pose_fc_kernel<4>, dim3(512 / 16, ry), dim3(1024), 0, s, (const float*)w.h1, (const float*)nullptr, 1024, net->Wh[1], net->bh[1],
                   512, B, 1, (const float*)nullptr, w.h2, 1024, (const float*)nullptr);
        DPD_CHECK_LAUNCH();
        DPD_LAUNCH(pose_fc_kernel<2>, dim3(256 / 16, ry), dim3(1024), 0, s, (const float*)w.h2, (const float*)nullptr, 512, net->Wh[2], net->bh[2], 256,
                   B, 1, dm, w.h3, 512, (const float*)nullptr);
        DPD_CHECK_LAUNCH();
    }
    DPD_LAUNCH(pose_apply_fwd_kernel, dim3((unsigned)B), dim3(64), 0, s, (const float*)nullptr, cur, Tcur, N, lim_rad, 0, (float*)nullptr, moved,
               T_out, (const float*)w.h3, net->Wh[3], net->bh[3], 256, pred_out ? pred_out + (size_t)(loops - 1) * B * 7 : (float*)nullptr);
    DPD_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ pose network, training evaluation (round 6)
// The training evaluation of a registration step (pcrnet-registration/iterative_PCRNet_ours.py:442-470) differentiates the pose network
// w.r.t. its WEIGHTS only (the refined source cloud is a constant of the step).  Shared MLP + max pool (models/ipcr_model.py:198-233) here:
//   dpd_pose_point_fwd_train   the forward kernel above with its hidden activations and the max pool's tie masks stored
//   dpd_pose_point_bwd         d features [clouds, OUT] -> dW1..dW5, db1..db5 (TF / torch autodiff of five 1x1 convolutions, ReLU, reduce_max)
// The max pool makes the last layer's gradient SPARSE: per (cloud, column) only the points that attain a positive maximum carry gradient
// (evenly shared among ties, like tf.reduce_max), i.e. one 128-vector per (cloud, column) instead of a [points, 1024] matrix:
//   pose_bwd_w5_dh4_kernel  ONE launch, two block roles:
//     (w5)  wave = output column c: dW5[c, :] = sum over clouds of coef * h4[tied point, :], db5[c] = sum of the column's gradient
//     (dh4) the gradient of layer 4's output, g4 = S W5 with the selection matrix S built from the tie masks, on the fp32
//                           matrix cores (workgroup = cloud x 32x32 tile, wave = 128 columns): independent of how the maxima spread over points
//   pose_bwd_cloud_kernel   workgroup = cloud: layers 4..1 on v_mfma_f32_32x32x2_f32 out of LDS; per-cloud partial weight gradients
//   pose_bwd_reduce_kernel  sums the partials over the clouds in cloud order
// fp32 throughout (MFMA fp32 = an fmaf chain per element); every sum has a fixed order (bitwise reproducible).
namespace dpd {

constexpr int kPart = 128 * 64 + 128 + 64 * 64 + 64 + 64 * 64 + 64 + 64 * 3 + 64;      // dW4 | db4 | dW3 | db3 | dW2 | db2 | dW1 | db1 per cloud

// (wave = output column c; the blocks after the g4 tiles of pose_bwd_w5_dh4_kernel, eight columns each)
__device__ __forceinline__ void bwd_w5_wave(const float* __restrict__ df, const unsigned long long* __restrict__ ties, const float* __restrict__ h4,
                                            int C, int N, int OUT, float* __restrict__ dW5, float* __restrict__ db5, int c) {
    const int l = threadIdx.x & 63;
    if (c >= OUT) return;
    float a0 = 0.f, a1 = 0.f, bsum = 0.f;
    for (int cl0 = 0; cl0 < C; cl0 += 64) {
        // lane = cloud: the column's tie mask and gradient of 64 clouds in ONE load each, so that the row loads below do not wait on them
        const int mine = cl0 + l;
        const unsigned long long mybits = mine < C ? ties[(size_t)mine * OUT + c] : 0ull;
        const float myg = (mine < C && mybits) ? df[(size_t)mine * OUT + c] : 0.f;
        const float mycoef = mybits ? myg / (float)__popcll(mybits) : 0.f;
        const int myfirst = mybits ? __ffsll((long long)mybits) - 1 : 0;
        const bool multi = mybits & (mybits - 1);
        const int n = min(64, C - cl0);
        for (int k0 = 0; k0 < n; k0 += 8) {          // eight clouds at a time: their (single) tied rows are requested together
            float2 h[8];
            float cf[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = min(k0 + u, n - 1);
                cf[u] = (k0 + u < n) ? __shfl(mycoef, k, 64) : 0.f;
                const int p = __shfl(myfirst, k, 64);
                h[u] = *reinterpret_cast<const float2*>(h4 + ((size_t)(cl0 + k) * N + p) * 128 + 2 * l);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { a0 += cf[u] * h[u].x; a1 += cf[u] * h[u].y; }      // (a cloud without a positive maximum: coef 0)
        }
        // further tied points of a column (duplicated points): rare, one at a time, after the first ones, in cloud / point order
        unsigned long long any = __ballot(multi);
        while (any) {
            const int k = __ffsll((long long)any) - 1;
            any &= any - 1;
            unsigned lo = (unsigned)__shfl((int)(unsigned)(mybits & 0xffffffffull), k, 64), hi = (unsigned)__shfl((int)(unsigned)(mybits >> 32), k, 64);
            unsigned long long bits = ((unsigned long long)hi << 32) | lo;
            const float coef = __shfl(mycoef, k, 64);
            bits &= bits - 1;                        // the first point was taken above
            while (bits) {
                const int p = __ffsll((long long)bits) - 1;
                bits &= bits - 1;
                const float2 hh = *reinterpret_cast<const float2*>(h4 + ((size_t)(cl0 + k) * N + p) * 128 + 2 * l);
                a0 += coef * hh.x;
                a1 += coef * hh.y;
            }
        }
        // db5[c] = sum of the column's gradient over the clouds in which the maximum is positive: lane sums in a fixed tree
        float g = myg;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) g += __shfl_xor(g, o, 64);
        bsum += g;
    }
    *reinterpret_cast<float2*>(dW5 + (size_t)c * 128 + 2 * l) = make_float2(a0, a1);
    if (l == 0) db5[c] = bsum;
}

// g4[cloud, p, :] = [h4 > 0] * sum over the columns whose maximum point p attains of coef * W5[column, :], coef = df / (number of tied points).
// As a product on the fp32 matrix cores: g4 = S W5 with S [64 points][1024 columns] holding coef[col] at the point(s) that attain column col's
// maximum -- one nonzero per column -- so the zeros cost 16.8 MFLOP per cloud and NOTHING depends on how the maxima are spread over the points.
// (The sparse forms walked each point's hit list: a few "critical" points attain hundreds of columns while most attain none, and every batch of
// weight rows was a dependent L2 round trip -- 30 us with a wave per point, 39-47 us with a workgroup per point.)
// Workgroup = (cloud, 32 points x 32 inputs tile), eight waves, wave w = columns 128 w .. + 127: the A operand is built in registers from the tie
// word and the coefficient of its column (LDS broadcasts), the B operand W5[col][n0 + lane % 32] is read straight from global memory (coalesced:
// v_mfma_f32_32x32x2_f32 wants one k per lane half), all 64 of them requested before the first product.  The eight partial tiles are added in
// wave order: columns ascending, deterministic; adding an exact zero changes nothing, so this IS the ascending-column sum of the sparse forms.
// ONE launch for both consumers of d features: blocks [0, 8 clouds) the g4 tiles, the rest dW5 / db5 (eight columns per block) -- two launches of
// 10.6 + 5.6 us that read the same df / ties and write disjoint outputs.
__global__ __launch_bounds__(512) void pose_bwd_w5_dh4_kernel(const float* __restrict__ df, const unsigned long long* __restrict__ ties,
                                                               const float* __restrict__ W5, const float* __restrict__ h4, int C, int N, int OUT,
                                                               float* __restrict__ g4g, float* __restrict__ dW5, float* __restrict__ db5) {
    __shared__ unsigned long long sT[1024];
    __shared__ float sC[1024];
    __shared__ float part[8][1024];
    if ((int)blockIdx.x >= C * 8) {
        bwd_w5_wave(df, ties, h4, C, N, OUT, dW5, db5, ((int)blockIdx.x - C * 8) * 8 + (int)(threadIdx.x >> 6));
        return;
    }
    const int c = blockIdx.x >> 3, tile = blockIdx.x & 7, m0 = (tile >> 2) * 32, n0 = (tile & 3) * 32;
    const int t = threadIdx.x, wv = t >> 6, l = t & 63, i = l & 31, h = l >> 5;
    const int kb = wv * 128;
    // every global load of the kernel is requested here, together
    float w[64];
    const float* wp = W5 + (size_t)(kb + h) * 128 + n0 + i;
#pragma unroll
    for (int j = 0; j < 64; ++j) w[j] = wp[(size_t)j * 256];
    unsigned long long tb[2];
    float dv[2], hv[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int col = q * 512 + t;
        tb[q] = ties[(size_t)c * OUT + col];
        dv[q] = df[(size_t)c * OUT + col];
        const int pr = m0 + (col >> 5);                       // (the epilogue's element: row col / 32 of the tile, input n0 + col % 32)
        hv[q] = pr < N ? h4[((size_t)c * N + pr) * 128 + n0 + (col & 31)] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        sT[q * 512 + t] = tb[q];
        sC[q * 512 + t] = tb[q] ? dv[q] / (float)__popcll(tb[q]) : 0.f;
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int sh = m0 + i;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        const int col = kb + 2 * j + h;
        const float a = ((sT[col] >> sh) & 1ull) ? sC[col] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w[j], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wv][mfma_row(r, l) * 32 + i] = acc[r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int o = q * 512 + t, pr = m0 + (o >> 5);
        float sum = part[0][o];
#pragma unroll
        for (int u = 1; u < 8; ++u) sum += part[u][o];
        if (pr < N) g4g[((size_t)c * N + pr) * 128 + n0 + (o & 31)] = hv[q] > 0.f ? sum : 0.f;
    }
}

// ---- layers 4..1 of one cloud on the fp32 matrix cores.  Every product is C[m][n] = sum_k A[m][k] B[n][k] with BOTH operands k-contiguous in
// LDS (rows padded by 4 floats: conflict-free 16-byte reads), so each tensor is kept in the orientation(s) its products contract over:
//   dW_l [o][i] = sum_p g_l[p][o] h_{l-1}[p][i]      A = g_l^T [o][p], B = h_{l-1}^T [i][p]
//   dh   [p][i] = sum_o g_l[p][o] W_l[o][i]          A = g_l [p][o],   B = W_l^T [i][o];   g_{l-1} = dh * [h_{l-1} > 0]
// 32x32 output tiles of v_mfma_f32_32x32x2_f32 (an exact fp32 fmaf chain per element, k ascending), four waves.  The FMA form of this kernel
// (one output row per thread, operands re-read from LDS for every multiply) was LDS-bandwidth bound at 45 us.
template <int K>
__device__ __forceinline__ f32x16 bwd_mm_tile(const float* __restrict__ A, int sa, const float* __restrict__ Bm, int sb, int m0, int n0) {
    const int l = threadIdx.x & 63, i = l & 31, h = l >> 5;
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    const float* ap = A + (m0 + i) * sa + 4 * h;
    const float* bp = Bm + (n0 + i) * sb + 4 * h;
#pragma unroll 4
    for (int k = 0; k < K; k += 8) {
        const float4 a = *reinterpret_cast<const float4*>(ap + k);
        const float4 b = *reinterpret_cast<const float4*>(bp + k);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, c, 0, 0, 0);
    }
    return c;
}
// (the k order inside a step of 8 is 4 h + e for lane half h: the two halves of the wave supply the two k of each 32x32x2 product)

constexpr int kBT = 256;       // threads of the per-cloud backward workgroup (four waves)
constexpr int kP64 = 68, kP128 = 132;
// LDS (floats): G [64 p][132] | GT [128 o][68] | HT [64 i][68] | WT [64 i][132] | GA [64][68] | GAT [64][68] | points [64][4]
//   layer 4: G = g4, GT = g4^T, HT = h3^T, WT = W4^T -> dW4, g3 (GA, GAT)
//   layer 3: HT = h2^T, WT = W3^T (stride 68)         -> dW3, g2 (G as [64][68], GT as [64][68])
//   layer 2: HT = h1^T, WT = W2^T                     -> dW2, g1 (GA)
constexpr int kMG = 0, kMGT = kMG + 64 * kP128, kMHT = kMGT + 128 * kP64, kMWT = kMHT + 64 * kP64, kMGA = kMWT + 64 * kP128, kMGAT = kMGA + 64 * kP64,
              kMPts = kMGAT + 64 * kP64, kBwdLds = kMPts + 256;
static_assert(kBwdLds * 4 <= 160 * 1024, "LDS budget");

// src [rows][cols] row-major in global memory (rows < nrow valid, else zero) -> dst[c][r] (transposed, stride ds), and optionally dst2[r][c]
__device__ __forceinline__ void lds_load_t(const float* __restrict__ src, int nrow, int rows, int cols, float* __restrict__ dstT, int dsT,
                                           float* __restrict__ dst, int ds, int t) {
    const int c4n = cols / 4;
    for (int e = t; e < rows * c4n; e += kBT) {
        const int r = e / c4n, c4 = (e % c4n) * 4;
        const float4 v = r < nrow ? *reinterpret_cast<const float4*>(src + (size_t)r * cols + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        dstT[(c4 + 0) * dsT + r] = v.x; dstT[(c4 + 1) * dsT + r] = v.y; dstT[(c4 + 2) * dsT + r] = v.z; dstT[(c4 + 3) * dsT + r] = v.w;
        if (dst) *reinterpret_cast<float4*>(dst + r * ds + c4) = v;
    }
}

// the same for a [64][64] matrix in two steps, so that the NEXT layer's operands are requested before this layer's products and written to LDS
// after them (a dependent global round trip per layer otherwise): thread t holds rows r = t / 16 + 16 q (q = 0..3), columns 4 (t % 16) .. + 3
struct Pre64 {
    float4 v[4];
};
__device__ __forceinline__ Pre64 pre64_load(const float* __restrict__ src, int nrow, int t) {
    Pre64 x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (t >> 4) + 16 * q, c4 = (t & 15) * 4;
        x.v[q] = r < nrow ? *reinterpret_cast<const float4*>(src + (size_t)r * 64 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return x;
}
__device__ __forceinline__ void pre64_store_t(const Pre64& x, float* __restrict__ dstT, int t) {      // dstT[c][r], stride 68
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (t >> 4) + 16 * q, c4 = (t & 15) * 4;
        dstT[(c4 + 0) * kP64 + r] = x.v[q].x; dstT[(c4 + 1) * kP64 + r] = x.v[q].y; dstT[(c4 + 2) * kP64 + r] = x.v[q].z; dstT[(c4 + 3) * kP64 + r] = x.v[q].w;
    }
}

// dW tile(s) of this wave -> the cloud's partial record; db[o] = sum_p g[p][o] by the first KO threads (G^T rows are contiguous in p)
template <int KO>
__device__ __forceinline__ void bwd_layer_dw(const float* __restrict__ GT, const float* __restrict__ HT, float* __restrict__ dW, float* __restrict__ db,
                                             int np, int t) {
    const int wv = t >> 6, l = t & 63;
    constexpr int TILES = (KO / 32) * 2;                  // 32x32 tiles of dW [KO][64]
    for (int tile = wv; tile < TILES; tile += 4) {
        const int m0 = (tile >> 1) * 32, n0 = (tile & 1) * 32;
        const f32x16 c = bwd_mm_tile<64>(GT, kP64, HT, kP64, m0, n0);
#pragma unroll
        for (int r = 0; r < 16; ++r) dW[(m0 + mfma_row(r, l)) * 64 + n0 + (l & 31)] = c[r];
    }
    if (t < KO) {
        float s = 0.f;
        for (int p = 0; p < np; ++p) s += GT[t * kP64 + p];
        db[t] = s;
    }
}

// g_prev = (g W) * [h_prev > 0]: tile (wave) of [64 p][64 i]; written as [p][i] (stride so) and transposed [i][p] (stride 68)
template <int KO>
__device__ __forceinline__ void bwd_layer_dx(const float* __restrict__ G, int sg, const float* __restrict__ WT, int sw, const float* __restrict__ HT,
                                             float* __restrict__ out, int so, float* __restrict__ outT, int t) {
    const int wv = t >> 6, l = t & 63;
    const int m0 = (wv >> 1) * 32, n0 = (wv & 1) * 32;
    const f32x16 c = bwd_mm_tile<KO>(G, sg, WT, sw, m0, n0);
    const int col = n0 + (l & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + mfma_row(r, l);
        const float v = HT[col * kP64 + row] > 0.f ? c[r] : 0.f;       // (rows >= np: h = 0 there, so g = 0)
        out[row * so + col] = v;
        if (outT) outT[col * kP64 + row] = v;
    }
}

__global__ __launch_bounds__(kBT) void pose_bwd_cloud_kernel(const float* __restrict__ ptsA, const float* __restrict__ ptsB, int nA, int N,
                                                              PointNetW net, const float* __restrict__ g4g, PointSave sv,
                                                              float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* G = lds + kMG;
    float* GT = lds + kMGT;
    float* HT = lds + kMHT;
    float* WT = lds + kMWT;
    float* GA = lds + kMGA;
    float* GAT = lds + kMGAT;
    float* sp = lds + kMPts;
    const int c = blockIdx.x, t = threadIdx.x;
    const int np = N;
    const float* pts = c < nA ? ptsA + (size_t)c * N * 3 : ptsB + (size_t)(c - nA) * N * 3;
    float* out = part + (size_t)c * kPart;
    // layer 4 (W4 [128 o][64 i]): g4 -> G [p][o], GT [o][p]; h3 -> HT [i][p]; W4 -> WT [i][o]
    lds_load_t(g4g + (size_t)c * N * 128, np, 64, 128, GT, kP64, G, kP128, t);
    lds_load_t(sv.h[2] + (size_t)c * N * 64, np, 64, 64, HT, kP64, nullptr, 0, t);
    lds_load_t(net.W[3], 128, 128, 64, WT, kP128, nullptr, 0, t);
    if (t < 192) sp[(t / 3) * 4 + t % 3] = (t / 3) < np ? pts[t] : 0.f;
    Pre64 ph = pre64_load(sv.h[1] + (size_t)c * N * 64, np, t), pw = pre64_load(net.W[2], 64, t);      // layer 3's operands, in flight
    __syncthreads();
    bwd_layer_dw<128>(GT, HT, out, out + 128 * 64, np, t);
    bwd_layer_dx<128>(G, kP128, WT, kP128, HT, GA, kP64, GAT, t);                      // g3 -> GA [p][o3], GAT [o3][p]
    __syncthreads();
    // layer 3 (W3 [64][64]): h2 -> HT, W3 -> WT (stride 68)
    pre64_store_t(ph, HT, t);
    pre64_store_t(pw, WT, t);
    ph = pre64_load(sv.h[0] + (size_t)c * N * 64, np, t);                                              // layer 2's
    pw = pre64_load(net.W[1], 64, t);
    __syncthreads();
    float* o3 = out + 128 * 64 + 128;
    bwd_layer_dw<64>(GAT, HT, o3, o3 + 64 * 64, np, t);
    bwd_layer_dx<64>(GA, kP64, WT, kP64, HT, G, kP64, GT, t);                          // g2 -> G [p][o2] (stride 68), GT [o2][p]
    __syncthreads();
    // layer 2 (W2 [64][64]): h1 -> HT, W2 -> WT
    pre64_store_t(ph, HT, t);
    pre64_store_t(pw, WT, t);
    __syncthreads();
    float* o2 = o3 + 64 * 64 + 64;
    bwd_layer_dw<64>(GT, HT, o2, o2 + 64 * 64, np, t);
    bwd_layer_dx<64>(G, kP64, WT, kP64, HT, GA, kP64, nullptr, t);                     // g1 -> GA [p][o1]
    __syncthreads();
    // layer 1 (64 x 3): dW1[o][i] = sum_p g1[p][o] pts[p][i], db1
    float* o1 = o2 + 64 * 64 + 64;
    if (t < 64) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, bs = 0.f;
        for (int p = 0; p < np; ++p) {
            const float a = GA[p * kP64 + t];
            bs += a; a0 += a * sp[p * 4]; a1 += a * sp[p * 4 + 1]; a2 += a * sp[p * 4 + 2];
        }
        o1[t * 3] = a0; o1[t * 3 + 1] = a1; o1[t * 3 + 2] = a2;
        o1[192 + t] = bs;
    }
}

__global__ __launch_bounds__(256) void pose_bwd_reduce_kernel(const float* __restrict__ part, int C, float* __restrict__ dW4, float* __restrict__ db4,
                                                               float* __restrict__ dW3, float* __restrict__ db3, float* __restrict__ dW2,
                                                               float* __restrict__ db2, float* __restrict__ dW1, float* __restrict__ db1) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kPart) return;
    float s = 0.f;
    int c = 0;
    for (; c + 8 <= C; c += 8) {                       // eight clouds' records requested together; added in cloud order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(c + u) * kPart + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; c < C; ++c) s += part[(size_t)c * kPart + i];
    int k = i;
    if (k < 128 * 64) { dW4[k] = s; return; }
    k -= 128 * 64;
    if (k < 128) { db4[k] = s; return; }
    k -= 128;
    if (k < 64 * 64) { dW3[k] = s; return; }
    k -= 64 * 64;
    if (k < 64) { db3[k] = s; return; }
    k -= 64;
    if (k < 64 * 64) { dW2[k] = s; return; }
    k -= 64 * 64;
    if (k < 64) { db2[k] = s; return; }
    k -= 64;
    if (k < 192) { dW1[k] = s; return; }
    k -= 192;
    db1[k] = s;
}

}  // namespace dpd

namespace {
dpd::LdsOptIn g_point_train_lds, g_point_bwd_lds;
int check_point_net(const dpd_pose_net* net) {
    if (!net) return DPD_E_NULL;
    for (int i = 0; i < 5; ++i)
        if (!net->Wp[i] || !net->bp[i]) return DPD_E_NULL;
    if (net->out_features != 1024) return DPD_E_UNSUPPORTED;
    return 0;
}
}  // namespace

extern "C" size_t dpd_pose_point_bwd_workspace_bytes(int clouds) {      // per-cloud partial weight gradients + g4 [clouds * 64, 128]
    return clouds > 0 ? (size_t)clouds * (dpd::kPart + 64 * 128) * sizeof(float) : 0;
}

extern "C" int dpd_pose_point_fwd_train(const dpd_pose_net* net, const float* ptsA, const float* ptsB, int nA, int nB, int N, float* f,
                                        float* h1, float* h2, float* h3, float* h4, unsigned long long* ties, void* stream) {
    using namespace dpd;
    if (int rc = check_point_net(net)) return rc;
    if (!ptsA || (nB > 0 && !ptsB) || !f || !h1 || !h2 || !h3 || !h4 || !ties) return DPD_E_NULL;
    if (nA <= 0 || nB < 0 || N <= 0) return DPD_E_DIM;
    if (N > kPP) return DPD_E_UNSUPPORTED;                 // the tie mask is one 64-bit word per (cloud, column)
    if ((((uintptr_t)h1 | (uintptr_t)h2 | (uintptr_t)h3 | (uintptr_t)h4) & 15) != 0) return DPD_E_UNSUPPORTED;
    const int OUT = net->out_features;
    const size_t lds = (size_t)kPointLds * sizeof(float);
    if (int rc = ensure_dyn_lds(g_point_train_lds, (const void*)pose_point_kernel<true>, lds)) return rc;
    PointNetW pw{};
    for (int i = 0; i < 5; ++i) { pw.W[i] = net->Wp[i]; pw.b[i] = net->bp[i]; }
    PointSave sv{{h1, h2, h3, h4}, ties};
    DPD_LAUNCH(pose_point_kernel<true>, dim3((unsigned)(nA + nB), (unsigned)(OUT / kSlice)), dim3(256), lds, (hipStream_t)stream, ptsA, ptsB, nA, N, pw,
               OUT, 0, f, sv, PoseMove{});
    DPD_CHECK_LAUNCH();
    return 0;
}

extern "C" int dpd_pose_point_bwd(const dpd_pose_net* net, const float* ptsA, const float* ptsB, int nA, int nB, int N, const float* df,
                                  const float* h1, const float* h2, const float* h3, const float* h4, const unsigned long long* ties,
                                  float* const* dW, float* const* db, void* ws, size_t ws_bytes, void* stream) {
    using namespace dpd;
    if (int rc = check_point_net(net)) return rc;
    if (!ptsA || (nB > 0 && !ptsB) || !df || !h1 || !h2 || !h3 || !h4 || !ties || !dW || !db || !ws) return DPD_E_NULL;
    for (int i = 0; i < 5; ++i)
        if (!dW[i] || !db[i]) return DPD_E_NULL;
    if (nA <= 0 || nB < 0 || N <= 0) return DPD_E_DIM;
    if (N > kPP) return DPD_E_UNSUPPORTED;
    const int C = nA + nB, OUT = net->out_features;
    if (ws_bytes < dpd_pose_point_bwd_workspace_bytes(C)) return DPD_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    PointNetW pw{};
    for (int i = 0; i < 5; ++i) { pw.W[i] = net->Wp[i]; pw.b[i] = net->bp[i]; }
    PointSave sv{{const_cast<float*>(h1), const_cast<float*>(h2), const_cast<float*>(h3), const_cast<float*>(h4)},
                 const_cast<unsigned long long*>(ties)};
    const size_t lds = (size_t)kBwdLds * sizeof(float);
    if (int rc = ensure_dyn_lds(g_point_bwd_lds, (const void*)pose_bwd_cloud_kernel, lds)) return rc;
    float* g4g = (float*)ws + (size_t)C * kPart;                   // [C * N, 128]
    DPD_LAUNCH(pose_bwd_w5_dh4_kernel, dim3((unsigned)(C * 8 + (OUT + 7) / 8)), dim3(512), 0, s, df, ties, net->Wp[4], h4, C, N, OUT, g4g, dW[4], db[4]);
    DPD_CHECK_LAUNCH();
    DPD_LAUNCH(pose_bwd_cloud_kernel, dim3((unsigned)C), dim3(kBT), lds, s, ptsA, ptsB, nA, N, pw, (const float*)g4g, sv, (float*)ws);
    DPD_CHECK_LAUNCH();
    DPD_LAUNCH(pose_bwd_reduce_kernel, dim3((unsigned)((kPart + 255) / 256)), dim3(256), 0, s, (const float*)ws, C, dW[3], db[3], dW[2], db[2], dW[1], db[1],
               dW[0], db[0]);
    DPD_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ pose network head, training evaluation (round 6)
// models/ipcr_model.py:273-284: fc 2048 -> 1024 -> 512 -> 256 (ReLU each, dropout on the last) -> 7, and TF / torch autodiff of it:
//   forward   pose_fc_kernel (above) x 3 with the hidden activations kept, pose_fc4_fwd_kernel
//   backward  pose_fc4_bwd_kernel: gradient of the 256-wide activation (dropout mask and ReLU gate applied) + dW4 / db4
//             pose_fc_dw3_kernel (dW = g^T x, db of the three wide layers in one launch: 16 terms per entry, bound by the 8 MB fc1 writes), per layer
//             pose_fc_dx_kernel (gx = (g W) * [x > 0] on v_mfma_f32_16x16x4_f32; the weight is streamed once: 16 input columns per workgroup,
//             sixteen waves each contracting a sixteenth of the outputs, partial tiles added in wave order -- 128 / 64 / 32 workgroups)
namespace dpd {

// pred[b][o] = sum_k h3[b][k] W4[o][k] + b4[o], o < 7, K4 = 256: one wave per row
__global__ __launch_bounds__(64) void pose_fc4_fwd_kernel(const float* __restrict__ h3, const float* __restrict__ W4, const float* __restrict__ b4,
                                                           int K4, float* __restrict__ pred) {
    const int b = blockIdx.x, l = threadIdx.x;
    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = l; k < K4; k += 64) {
        const float x = h3[(size_t)b * K4 + k];
#pragma unroll
        for (int o = 0; o < 7; ++o) acc[o] += x * W4[(size_t)o * K4 + k];
    }
#pragma unroll
    for (int o = 0; o < 7; ++o) {
        float v = acc[o];
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
        if (l == 0) pred[(size_t)b * 7 + o] = v + b4[o];
    }
}

// blocks [0, B): g3[b][k] = (sum_o dpred[b][o] W4[o][k]) * mask[b][k] * [h3[b][k] > 0]; block B: dW4[o][k] = sum_b dpred[b][o] h3[b][k], db4
__global__ __launch_bounds__(256) void pose_fc4_bwd_kernel(const float* __restrict__ dpred, const float* __restrict__ W4, const float* __restrict__ h3,
                                                            const float* __restrict__ mask, int B, int K4, float* __restrict__ g3,
                                                            float* __restrict__ dW4, float* __restrict__ db4) {
    const int t = threadIdx.x;
    if ((int)blockIdx.x < B) {
        const int b = blockIdx.x;
        float d[7];
#pragma unroll
        for (int o = 0; o < 7; ++o) d[o] = dpred[(size_t)b * 7 + o];
        for (int k = t; k < K4; k += 256) {
            float v = 0.f;
#pragma unroll
            for (int o = 0; o < 7; ++o) v += d[o] * W4[(size_t)o * K4 + k];
            if (mask) v *= mask[(size_t)b * K4 + k];
            g3[(size_t)b * K4 + k] = h3[(size_t)b * K4 + k] > 0.f ? v : 0.f;
        }
        return;
    }
    // workgroup B + o: row o of dW4 (and db4[o]); the batch's terms are requested eight at a time and added in batch order
    const int o = blockIdx.x - B;
    for (int k = t; k < K4; k += 256) {
        float v = 0.f;
        int b = 0;
        for (; b + 8 <= B; b += 8) {
            float d[8], hh[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { d[u] = dpred[(size_t)(b + u) * 7 + o]; hh[u] = h3[(size_t)(b + u) * K4 + k]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) v += d[u] * hh[u];
        }
        for (; b < B; ++b) v += dpred[(size_t)b * 7 + o] * h3[(size_t)b * K4 + k];
        dW4[(size_t)o * K4 + k] = v;
    }
    if (t == 0) {
        float v = 0.f;
        for (int b = 0; b < B; ++b) v += dpred[(size_t)b * 7 + o];
        db4[o] = v;
    }
}

// dW[j][k] = sum_r g[r][j] x[r][k] (r ascending), db[j] = sum_r g[r][j].  x = [xA (KA columns) | xB (K - KA)] like pose_fc_kernel's input.
// block = 16 output rows x 256 columns: thread = (four rows j, one float4 of columns)
struct FcDwJob {               // dW [J,K] = g^T [J,R] x [R,K] (x = [xA | xB], xA KA columns wide), db [J]; K / 256 x J / 64 blocks
    const float* g;
    const float* xA;
    const float* xB;
    float* dW;
    float* db;
    int KA, J, K;
    int blocks;                 // (K / 256) * (J / 64)
};
__device__ __forceinline__ void fc_dw_block(const float* __restrict__ g, const float* __restrict__ xA, const float* __restrict__ xB, int KA, int J,
                                            int K, int R, float* __restrict__ dW, float* __restrict__ db, int bx, int by) {
    // block = 64 outputs j x 256 inputs k: wave w = outputs 16 w .. + 15 (its g values are wave-uniform: scalar loads), lane = four inputs.
    // The first sixteen rows of x stay in registers for all sixteen outputs (one round of loads; a thread of the 4 j x 4 k form re-read them
    // four times as often: 43 MB of L2 traffic for fc1); further rows (batch > 16) are read again per group of outputs.
    const int t = threadIdx.x, kq = t & 63, jg = __builtin_amdgcn_readfirstlane(t >> 6);
    const int k = bx * 256 + 4 * kq, jw = by * 64 + 16 * jg;
    if (k >= K) return;
    const float* x = k < KA ? xA + k : xB + (k - KA);
    const int ldx = k < KA ? KA : K - KA;
    float4 xr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) xr[r] = r < R ? *reinterpret_cast<const float4*>(x + (size_t)r * ldx) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int jc = 0; jc < 4; ++jc) {
        const int j0 = jw + 4 * jc;
        float4 acc[4];
        float bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float4 gv = r < R ? *reinterpret_cast<const float4*>(g + (size_t)r * J + j0) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float gg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[u].x += gg[u] * xr[r].x; acc[u].y += gg[u] * xr[r].y; acc[u].z += gg[u] * xr[r].z; acc[u].w += gg[u] * xr[r].w;
                bs[u] += gg[u];
            }
        }
        for (int r = 16; r < R; ++r) {
            const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)r * ldx);
            const float4 gv = *reinterpret_cast<const float4*>(g + (size_t)r * J + j0);
            const float gg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[u].x += gg[u] * xv.x; acc[u].y += gg[u] * xv.y; acc[u].z += gg[u] * xv.z; acc[u].w += gg[u] * xv.w;
                bs[u] += gg[u];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<float4*>(dW + (size_t)(j0 + u) * K + k) = acc[u];
        if (bx == 0 && kq == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) db[j0 + u] = bs[u];
        }
    }
}
// the three wide layers' weight gradients in ONE launch (they depend on g3, g2, g1 only: after the second dx launch, beside nothing else --
// three launches of 7-8 us each were mostly launch ramp): block ranges [0, n0) fc3, [n0, n0 + n1) fc2, the rest fc1 (8 + 32 + 128 blocks)
__global__ __launch_bounds__(256) void pose_fc_dw3_kernel(FcDwJob a, FcDwJob b, FcDwJob c, int R) {
    int id = blockIdx.x;
    const FcDwJob* j = &a;
    if (id >= a.blocks) { id -= a.blocks; j = &b; }
    if (j == &b && id >= b.blocks) { id -= b.blocks; j = &c; }
    const int nbx = j->K / 256;
    fc_dw_block(j->g, j->xA, j->xB, j->KA, j->J, j->K, R, j->dW, j->db, id % nbx, id / nbx);
}

// gx[r][k] = (sum_j g[r][j] W[j][k]) * [xprev[r][k] > 0] for <= 16 rows per blockIdx.y and the 64 columns k0 .. k0 + 63 of blockIdx.x.
// Eight waves share the reduction (wave w: j in [w J / 8, (w + 1) J / 8), NIT = J / 32 steps of four j); per step a lane loads ONE float4 of W
// (row j0 + lane / 16, columns k0 + 4 (lane % 16) .. + 3) and one g value (row r0 + lane % 16, column j0 + lane / 16) and issues four
// v_mfma_f32_16x16x4_f32: A_e[i][kk] = W[j0 + kk][k0 + 4 i + e], B[kk][n] = g[r0 + n][j0 + kk] -> C_e[i][n] = partial gx[r0 + n][k0 + 4 i + e].
// A lane then holds the 16 consecutive columns k0 + 16 (lane / 16) .. + 15 of row r0 + lane % 16; the eight partial tiles are added in wave
// order through LDS (deterministic).  xprev == NULL: no gate (the pooled features).  out rows: row r -> out + r * ldo (+ column k); for the
// first layer the columns >= split go to the rows of the second cloud set: out[(R + r) * ldo + k - split].
template <int NIT>
__global__ __launch_bounds__(1024) void pose_fc_dx_kernel(const float* __restrict__ g, const float* __restrict__ W, const float* __restrict__ xprev,
                                                           int K, int R, float* __restrict__ out, int ldo, int split) {
    constexpr int J = NIT * 64;
    __shared__ float part[16][256];
    const int k0 = blockIdx.x * 16, r0 = blockIdx.y * 16, wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int n = l & 15, kk = l >> 4;
    const int row = min(r0 + n, R - 1);
    const int jw = wv * (J / 16);
    // the epilogue's element (threads 0..255): row r0 + (t / 4) % 16, column k0 + 4 (t / 64) + t % 4; its ReLU gate is requested with the operands
    const int et = threadIdx.x, err = r0 + ((et >> 2) & 15), ek = k0 + 4 * (et >> 6) + (et & 3);
    const float gate = (et < 256 && err < R && xprev) ? xprev[(size_t)err * K + ek] : 1.f;
    float wr[NIT], gr[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        wr[i] = W[(size_t)(jw + 4 * i + kk) * K + k0 + n];
        gr[i] = g[(size_t)row * J + jw + 4 * i + kk];
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NIT; ++i) c = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[i], gr[i], c, 0, 0, 0);
    // lane l holds C[i = 4 (l / 16) + r][n = l % 16]: column k0 + i of row n
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wv][l * 4 + r] = c[r];
    __syncthreads();
    if (et < 256) {
        float v = part[0][et];
#pragma unroll
        for (int q = 1; q < 16; ++q) v += part[q][et];
        if (err < R) {
            if (!(gate > 0.f)) v = 0.f;
            if (ek < split) out[(size_t)err * ldo + ek] = v;
            else out[(size_t)(R + err) * ldo + ek - split] = v;
        }
    }
}

}  // namespace dpd

extern "C" int dpd_pose_head_fwd_train(const dpd_pose_net* net, const float* f, int B, const float* drop_mask, float* h1, float* h2, float* h3,
                                       float* pred, void* stream) {
    using namespace dpd;
    if (!net || !f || !h1 || !h2 || !h3 || !pred) return DPD_E_NULL;
    for (int i = 0; i < 4; ++i)
        if (!net->Wh[i] || !net->bh[i]) return DPD_E_NULL;
    if (B <= 0) return DPD_E_DIM;
    if (net->out_features != 1024) return DPD_E_UNSUPPORTED;
    const int OUT = 1024;
    hipStream_t s = (hipStream_t)stream;
    const unsigned ry = (unsigned)((B + 15) / 16);
    DPD_LAUNCH(pose_fc_kernel<8>, dim3(1024 / 16, ry), dim3(1024), 0, s, f, f + (size_t)B * OUT, OUT, net->Wh[0], net->bh[0], 1024, B, 1,
               (const float*)nullptr, h1, 2048, (const float*)nullptr);
    DPD_CHECK_LAUNCH();
    DPD_LAUNCH(pose_fc_kernel<4>, dim3(512 / 16, ry), dim3(1024), 0, s, (const float*)h1, (const float*)nullptr, 1024, net->Wh[1], net->bh[1], 512, B, 1,
               (const float*)nullptr, h2, 1024, (const float*)nullptr);
    DPD_CHECK_LAUNCH();
    DPD_LAUNCH(pose_fc_kernel<2>, dim3(256 / 16, ry), dim3(1024), 0, s, (const float*)h2, (const float*)nullptr, 512, net->Wh[2], net->bh[2], 256, B, 1,
               drop_mask, h3, 512, (const float*)nullptr);
    DPD_CHECK_LAUNCH();
    DPD_LAUNCH(pose_fc4_fwd_kernel, dim3((unsigned)B), dim3(64), 0, s, (const float*)h3, net->Wh[3], net->bh[3], 256, pred);
    DPD_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t dpd_pose_head_bwd_workspace_bytes(int B) { return B > 0 ? (size_t)B * (256 + 512 + 1024) * sizeof(float) : 0; }

extern "C" int dpd_pose_head_bwd(const dpd_pose_net* net, const float* f, int B, const float* drop_mask, const float* h1, const float* h2,
                                 const float* h3, const float* dpred, float* const* dW, float* const* db, float* df, void* ws, size_t ws_bytes,
                                 void* stream) {
    using namespace dpd;
    if (!net || !f || !h1 || !h2 || !h3 || !dpred || !dW || !db || !df || !ws) return DPD_E_NULL;
    for (int i = 0; i < 4; ++i)
        if (!net->Wh[i] || !dW[i] || !db[i]) return DPD_E_NULL;
    if (B <= 0) return DPD_E_DIM;
    if (net->out_features != 1024) return DPD_E_UNSUPPORTED;
    if (ws_bytes < dpd_pose_head_bwd_workspace_bytes(B)) return DPD_E_WORKSPACE;
    const int OUT = 1024;
    hipStream_t s = (hipStream_t)stream;
    float* g3 = (float*)ws;                  // [B, 256]
    float* g2 = g3 + (size_t)B * 256;        // [B, 512]
    float* g1 = g2 + (size_t)B * 512;        // [B, 1024]
    const unsigned ry = (unsigned)((B + 15) / 16);
    DPD_LAUNCH(pose_fc4_bwd_kernel, dim3((unsigned)B + 7), dim3(256), 0, s, dpred, net->Wh[3], h3, drop_mask, B, 256, g3, dW[3], db[3]);
    DPD_CHECK_LAUNCH();
    // fc3: W [256, 512]
    DPD_LAUNCH(pose_fc_dx_kernel<4>, dim3(512 / 16, ry), dim3(1024), 0, s, (const float*)g3, net->Wh[2], h2, 512, B, g2, 512, 512);
    DPD_CHECK_LAUNCH();
    // fc2: W [512, 1024]
    DPD_LAUNCH(pose_fc_dx_kernel<8>, dim3(1024 / 16, ry), dim3(1024), 0, s, (const float*)g2, net->Wh[1], h1, 1024, B, g1, 1024, 1024);
    DPD_CHECK_LAUNCH();
    // fc1: W [1024, 2048], input = [features of the first B clouds | of the second B clouds]; its dX is d features [2B, 1024] (no gate)
    {
        const FcDwJob j3{g3, h2, nullptr, dW[2], db[2], 512, 256, 512, (512 / 256) * (256 / 64)};
        const FcDwJob j2{g2, h1, nullptr, dW[1], db[1], 1024, 512, 1024, (1024 / 256) * (512 / 64)};
        const FcDwJob j1{g1, f, f + (size_t)B * OUT, dW[0], db[0], OUT, 1024, 2048, (2048 / 256) * (1024 / 64)};
        DPD_LAUNCH(pose_fc_dw3_kernel, dim3((unsigned)(j3.blocks + j2.blocks + j1.blocks)), dim3(256), 0, s, j3, j2, j1, B);
        DPD_CHECK_LAUNCH();
    }
    DPD_LAUNCH(pose_fc_dx_kernel<16>, dim3(2048 / 16, ry), dim3(1024), 0, s, (const float*)g1, net->Wh[0], (const float*)nullptr, 2048, B, df, OUT, OUT);
    DPD_CHECK_LAUNCH();
    return 0;
}
