// 3D modified Fisher Vector encoder (forward + backward), one workgroup per cloud.
//
// Replaces utils/dpdist_util.py:22-141 (get_3dmfv_tf with full_fv=True, normalize=True) and its TF autodiff.
//
// Data layout in HBM: pts [C,N,3] fp32 (768 B per 64-point cloud), fv [C,G,20] fp32 (40 KB per cloud for m=8).
// Algorithmic HBM bytes per cloud: N*12 read + G*80 written.  The kernel is latency/VALU bound (N*G pdf
// evaluations ~ 1.3 MFLOP per cloud), not a roofline kernel; it exists to keep the [B,N,G,3] TF tiles
// (5 x 12.6 MB at B=32) out of HBM altogether.
//
// Mapping (wave64), forward (details at mfv3d_fwd_kernel): 4 workgroups of 1024 threads per cloud, each owning a slice of the
//   Gaussians; tables zq[axis][point][cell] = {z, e/S} of the FACTORISED responsibilities in LDS; lane&7 <-> Gaussian,
//   lane>>3 <-> one eighth of the points, 20 running statistics (sum/max/min) in registers, the eight point groups merged with
//   three __shfl_xor; power-1/2 values staged through LDS and written with coalesced float4 stores; a second small kernel
//   applies the per-channel L2 norm over the Gaussian axis.  Backward: sliced over the POINTS (two launches) or one launch.
#include <cstdlib>

#include "common.h"
#include "patch_rows_bwd.h"

namespace dpd {

constexpr int kF = DPD_FV_CHANNELS;  // 20
constexpr int kFP = kF + 1;          // padded LDS row

struct MfvConst {
    GridAxis ax;
    int N, m, G;
    float sigma;
    float lognorm;   // 0.5*D*log(2*pi) + D*log(sigma)
    float w;         // 1/G
    float dpi_den;   // sqrt(w) * N            (:78)
    float mu_scale;  // 1/sqrt(w)              (:98)
    float sig_scale; // 1/sqrt(2w)             (:109)
};

// Index arithmetic without the integer divider gfx950 does not have (~35 VALU instructions per run-time division; the table loops of
// these kernels run three of them per entry with four waves per SIMD: 3.3k of the forward kernel's 21.7k cycles at B = 32): the usual
// sizes (m = 8, N = 64, slices of 16 points) are powers of two, so quotients are shifts; anything else keeps the division.
__device__ __forceinline__ int lg_or_neg(int d) { return (d > 0 && !(d & (d - 1))) ? __ffs(d) - 1 : -1; }
__device__ __forceinline__ int qdiv(int x, int d, int lg) { return lg >= 0 ? x >> lg : x / d; }

__device__ __forceinline__ void gauss_centre(const MfvConst& k, int g, float& cx, float& cy, float& cz) {
    // g = i*m*m + j*m + t  ->  (x,y,z) = (l[j], l[i], l[t])   (np.meshgrid 'xy' indexing, :47-48)
    const int m = k.m;
    const int i = g / (m * m), j = (g / m) % m, t = g % m;
    cx = k.ax.c[j];
    cy = k.ax.c[i];
    cz = k.ax.c[t];
}

__device__ __forceinline__ float pdf(const MfvConst& k, float zx, float zy, float zz) {
    // MultivariateNormalDiag.prob (:69-71): exp(-0.5*sum z^2 - (0.5*D*log 2pi + sum log sigma)), fp32 like TF
    return expf(-0.5f * (zx * zx + zy * zy + zz * zz) - k.lognorm);
}

__device__ __forceinline__ float pnorm(float x) {
    // sign(x) * max(|x|,1e-12)^0.5  (:119-121).  NaN in -> NaN out, as in the oracle (sign(NaN) = NaN there); the
    // only source of NaN is the reference's own 0/0 for a point that underflows every pdf.
    if (x != x) return x;
    const float s = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
    return s * sqrtf(fmaxf(fabsf(x), 1e-12f));
}

// ---------------------------------------------------------------------------------------------------------
// Forward: kSlices workgroups of 1024 threads per cloud (each owns a contiguous slice of the Gaussians), then a
// small normalisation kernel.  64 clouds alone would occupy 64 of the 256 CUs; sliced, the chip is full at B = 32.
//
// The Gaussians have diagonal covariance on a product grid, so the responsibility factorises exactly:
//     Q_ng = w p_ng / sum_g' w p_ng' = (ex[n][j]/Sx[n]) * (ey[n][i]/Sy[n]) * (ez[n][t]/Sz[n]),
//     e_a[n][i] = exp(-z_a^2/2),  S_a[n] = sum_i e_a[n][i]          (constants and the weight w cancel)
// which replaces N*G = 32 768 expf + divisions per cloud (:69-74) by 3*N*m = 1 536 of each.  The result differs from
// the reference's fp32 evaluation by a few ulp of Q; the reference's failure mode is reproduced explicitly:
// if w*p_ng underflows to 0 for EVERY Gaussian of some point (0/0 at :74) the whole descriptor is NaN.
//
//   tables  zq[a][n][i] = { z = (p[n][a]-l_i)/sigma , q = e/S }  (float2, one ds_read_b64 per axis per pair)
//   pass 2  lane&7 <-> Gaussian (8 per wave), lane>>3 <-> one eighth of the points; 20 running statistics in registers;
//           the eight point groups are merged with three __shfl_xor per statistic
//   store   power-1/2 values staged through LDS [slice][21] (conflict free), coalesced float4 stores (NaN if 0/0)
//   norm    mfv3d_norm_kernel: per-channel L2 over the Gaussian axis (:124-126), fixed summation order
// dynamic LDS (floats): zq[3*N*m*2] | S[3*N] | minz2[3*N] | stage[slice*21] | flag
// ---------------------------------------------------------------------------------------------------------
constexpr int kFwdThreads = 1024;
constexpr int kSlices = kMfvSlices;   // common.h (the window gather sums the per-slice norms)

// Optional fusions of the training step's front end (all NULL = plain encoder over pts):
//   pcA/pcB/noise: cloud c < B is pcA[c] + noise[c], cloud c >= B is pcB[c - B] (dpdist_and_aue.py:45,56-61); slice 0 of every
//                  cloud also writes the stacked encoder input pts_out [2B,N,3] and the query clouds q_out = [pcB ; pcA] (:69),
//                  so that dpd_stack_clouds is not launched;
//   ssq:           [C][kSlices][20] receives this slice's per-channel sums of squares (slice_ssq order below); the caller then
//                  skips mfv3d_norm_kernel and the window gather applies the L2 scale instead (patch_rows.hip).
struct MfvFuse {
    const float* pcA; const float* pcB; const float* noise;
    float* pts_out; float* q_out;
    float* ssq;
    int B;
};

// Per-channel sum of squares of one slice of Gaussians, in THE summation order both consumers use (so that the separate norm
// kernel and the fused form give the same bits): 8 parts of ceil(gcount / 8) consecutive Gaussians, summed sequentially,
// then the 8 partials in order.  val(g, ch) reads the slice-local value.  Threads 0..159 take part; result valid for tid < 20.
template <typename F>
__device__ __forceinline__ float slice_ssq(int tid, int gcount, float* s_red /* [8][20] */, F val) {
    const int per = (gcount + 7) / 8;
    if (tid < 8 * kF) {
        const int part = tid / kF, ch = tid % kF;
        float ss = 0.f;
        const int g1 = min(gcount, (part + 1) * per);
        // eight loads in flight, then the SAME additions in the same order (a padded 0 * 0 adds an exact zero): the sequential form was a
        // chain of 16 dependent LDS round trips (1.6k of the forward kernel's 16k cycles, tools/mfv_stamps.py)
        for (int g = part * per; g < g1; g += 8) {
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = (g + u < g1) ? val(g + u, ch) : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) ss += x[u] * x[u];
        }
        s_red[part * kF + ch] = ss;
    }
    __syncthreads();
    float t = 0.f;
    if (tid < kF) {
        float x[8];
#pragma unroll
        for (int part = 0; part < 8; ++part) x[part] = s_red[part * kF + tid];
#pragma unroll
        for (int part = 0; part < 8; ++part) t += x[part];
    }
    return t;
}

#ifdef DPD_ABLATIONS
__device__ unsigned long long g_mfv_stamps[1024 * 8];      // s_memtime milestones of thread 0 of every workgroup (tools/mfv_stamps.py)
#define MFV_STAMP(i) do { if (threadIdx.x == 0) g_mfv_stamps[(blockIdx.x & 1023) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MFV_STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(kFwdThreads) void mfv3d_fwd_kernel(const float* __restrict__ pts, float* __restrict__ fv,
                                                                 MfvConst k, int gslice, MfvFuse fu) {
    MFV_STAMP(0);
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int N = k.N, G = k.G, m = k.m;
    float2* s_zq = reinterpret_cast<float2*>(sm);   // [3][N][m]
    float* s_S = sm + 6 * N * m;                    // [3][N]
    float* s_mz = s_S + 3 * N;                      // [3][N] min_i z^2
    float* s_stage = s_mz + 3 * N;                  // [gslice][21]
    int* s_bad = reinterpret_cast<int*>(s_stage + gslice * kFP);

    const int tid = threadIdx.x, c = blockIdx.x / kSlices, sl = blockIdx.x % kSlices;
    const int g0 = sl * gslice, gcount = max(0, min(G, g0 + gslice) - g0);
    const int lane = tid & 63, wave = tid >> 6;
    const float* p = pts ? pts + (size_t)c * N * 3 : (c < fu.B ? fu.pcA + (size_t)c * N * 3 : fu.pcB + (size_t)(c - fu.B) * N * 3);
    const float* nz = (!pts && fu.noise && c < fu.B) ? fu.noise + (size_t)c * N * 3 : nullptr;
    if (tid == 0) *s_bad = 0;
    const int lg_m = lg_or_neg(m), lg_n = lg_or_neg(N);
    for (int e = tid; e < 3 * N * m; e += kFwdThreads) {
        const int em = qdiv(e, m, lg_m), i = e - em * m, a = qdiv(em, N, lg_n), n = em - a * N;
        const float x = nz ? p[n * 3 + a] + nz[n * 3 + a] : p[n * 3 + a];
        const float z = (x - k.ax.c[i]) / k.sigma;               // (batch_points - batch_mu) / batch_sig  (:87)
        s_zq[e] = make_float2(z, expf(-0.5f * (z * z)));
    }
    if (!pts && sl == 0) {      // the stacked tensors the rest of the step reads
        const int twin = c < fu.B ? c + fu.B : c - fu.B;          // q = [pcB ; pcA]: cloud c's raw points are the twin's queries
        for (int e = tid; e < 3 * N; e += kFwdThreads) {
            const float raw = p[e];
            if (fu.pts_out) fu.pts_out[(size_t)c * N * 3 + e] = nz ? raw + nz[e] : raw;
            if (fu.q_out) fu.q_out[(size_t)twin * N * 3 + e] = raw;
        }
    }
    __syncthreads();
    MFV_STAMP(1);
    for (int e = tid; e < 3 * N; e += kFwdThreads) {              // e = a*N + n
        float S = 0.f, mz = INFINITY;
        for (int i = 0; i < m; ++i) {
            const float2 v = s_zq[e * m + i];
            S += v.y;
            mz = fminf(mz, v.x * v.x);
            if (v.x != v.x) mz = v.x;
        }
        s_S[e] = S;
        s_mz[e] = mz;
    }
    __syncthreads();
    for (int e = tid; e < 3 * N * m; e += kFwdThreads) s_zq[e].y = s_zq[e].y / s_S[qdiv(e, m, lg_m)];
    for (int n = tid; n < N; n += kFwdThreads) {
        // the reference's 0/0: every w*p_ng == 0  <=>  the largest one is (p is monotone in -|z|^2)
        const float pmax = pdf(k, sqrtf(s_mz[n]), sqrtf(s_mz[N + n]), sqrtf(s_mz[2 * N + n]));
        if (!(pmax * k.w > 0.f)) atomicOr(s_bad, 1);             // also catches NaN inputs
    }
    __syncthreads();
    MFV_STAMP(2);
    const float2* zqx = s_zq;
    const float2* zqy = s_zq + N * m;
    const float2* zqz = s_zq + 2 * N * m;

    // ---- per-Gaussian statistics over the points ---------------------------------------------------------------
    const float invN = 1.0f / (float)N;
    const float inv_dpi = 1.0f / k.dpi_den;
    const int npg = (N + 7) / 8;
    const int grp = lane >> 3;
    for (int gbase = wave * 8; gbase < gcount; gbase += 16 * 8) {
        const int gl = gbase + (lane & 7);
        const bool live = gl < gcount;
        const int gg = live ? g0 + gl : 0;
        const int gm = qdiv(gg, m, lg_m), t = gg - gm * m, i = qdiv(gm, m, lg_m), j = gm - i * m;   // centre (x,y,z) = (l[j], l[i], l[t])  (:47-48)
        float pi_s = 0.f, pi_mx = -INFINITY;
        float mu_s[3] = {0.f, 0.f, 0.f}, mu_mx[3] = {-INFINITY, -INFINITY, -INFINITY}, mu_mn[3] = {INFINITY, INFINITY, INFINITY};
        float sg_s[3] = {0.f, 0.f, 0.f}, sg_mx[3] = {-INFINITY, -INFINITY, -INFINITY}, sg_mn[3] = {INFINITY, INFINITY, INFINITY};
        const int n1 = min(N, (grp + 1) * npg);
        for (int n = grp * npg; n < n1; ++n) {
            const float2 vx = zqx[n * m + j], vy = zqy[n * m + i], vz = zqz[n * m + t];
            const float z[3] = {vx.x, vy.x, vz.x};
            const float Q = (vx.y * vy.y) * vz.y;                              // :73-74, factorised
            const float dpi = (Q - k.w) * inv_dpi;                             // :78
            pi_s += dpi;
            pi_mx = fmaxf(pi_mx, dpi);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float a = Q * z[d];                                       // :87
                const float b = Q * (z[d] * z[d] - 1.0f);                       // :100
                mu_s[d] += a; mu_mx[d] = fmaxf(mu_mx[d], a); mu_mn[d] = fminf(mu_mn[d], a);
                sg_s[d] += b; sg_mx[d] = fmaxf(sg_mx[d], b); sg_mn[d] = fminf(sg_mn[d], b);
            }
        }
        MFV_STAMP(5);
        // merge the eight point groups (lanes l, l^8, l^16, l^32 ... hold the same Gaussian).  __shfl_xor is a ds_bpermute (LDS crossbar +
        // an address register) per value and stage: 60 of them made this merge as expensive as the point loop, and the section is VALU /
        // issue bound (four waves per SIMD).  xor 8 is a DPP row rotate, xor 32 a v_permlane32_swap (both halves receive lower + upper),
        // only xor 16 still goes through the crossbar (ds_swizzle, no address).  Same operands, same order: the same bits.
        auto x8 = [](float v) { return dpp_move<0x128>(v); };                                                  // row_ror:8 = lane ^ 8
        auto x16 = [](float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F)); };   // bit mode: xor 16
        auto sum32 = [](float v) { const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                                   return __uint_as_float(r[0]) + __uint_as_float(r[1]); };
        auto max32 = [](float v) { const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                                   return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])); };
        auto min32 = [](float v) { const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                                   return fminf(__uint_as_float(r[0]), __uint_as_float(r[1])); };
        pi_s += x8(pi_s); pi_s += x16(pi_s); pi_s = sum32(pi_s);
        pi_mx = fmaxf(pi_mx, x8(pi_mx)); pi_mx = fmaxf(pi_mx, x16(pi_mx)); pi_mx = max32(pi_mx);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            mu_s[d] += x8(mu_s[d]); mu_s[d] += x16(mu_s[d]); mu_s[d] = sum32(mu_s[d]);
            sg_s[d] += x8(sg_s[d]); sg_s[d] += x16(sg_s[d]); sg_s[d] = sum32(sg_s[d]);
            mu_mx[d] = fmaxf(mu_mx[d], x8(mu_mx[d])); mu_mx[d] = fmaxf(mu_mx[d], x16(mu_mx[d])); mu_mx[d] = max32(mu_mx[d]);
            mu_mn[d] = fminf(mu_mn[d], x8(mu_mn[d])); mu_mn[d] = fminf(mu_mn[d], x16(mu_mn[d])); mu_mn[d] = min32(mu_mn[d]);
            sg_mx[d] = fmaxf(sg_mx[d], x8(sg_mx[d])); sg_mx[d] = fmaxf(sg_mx[d], x16(sg_mx[d])); sg_mx[d] = max32(sg_mx[d]);
            sg_mn[d] = fminf(sg_mn[d], x8(sg_mn[d])); sg_mn[d] = fminf(sg_mn[d], x16(sg_mn[d])); sg_mn[d] = min32(sg_mn[d]);
        }
        MFV_STAMP(6);
        if (live && grp == 0) {
            float v[kF];
            v[0] = pi_s * invN;                                                 // :81 reduce_mean
            v[1] = pi_mx;                                                       // :80
#pragma unroll
            for (int d = 0; d < 3; ++d) {                                       // :89-98, :102-109
                v[2 + d] = (mu_s[d] * invN) * k.mu_scale;
                v[5 + d] = mu_mx[d] * k.mu_scale;
                v[8 + d] = mu_mn[d] * k.mu_scale;
                v[11 + d] = (sg_s[d] * invN) * k.sig_scale;
                v[14 + d] = sg_mx[d] * k.sig_scale;
                v[17 + d] = sg_mn[d] * k.sig_scale;
            }
#pragma unroll
            for (int f = 0; f < kF; ++f) s_stage[gl * kFP + f] = v[f];
        }
    }
    MFV_STAMP(7);
    __syncthreads();
    MFV_STAMP(3);

    // ---- power-1/2 normalisation (:119-121) and coalesced store of the slice, one pass of ALL threads: fv[c][g0 + g][f], 4 consecutive f of
    // one g per thread.  (Inside the merge branch above only 8 lanes of 64 were alive for 20 square roots each, four waves per SIMD
    // queueing for the same VALU: 15k of this kernel's 30k cycles, tools/mfv_stamps.py.)  The normalised values go back to the stage for
    // the slice's sums of squares.
    float* out = fv + ((size_t)c * G + g0) * kF;
    const bool bad = *s_bad != 0;
    const float qnan = __int_as_float(0x7fc00000);
    for (int i4 = tid; i4 < gcount * kF / 4; i4 += kFwdThreads) {
        const int e = i4 * 4, g = e / kF, f = e % kF;
        float* sp = s_stage + g * kFP + f;
        float4 o = make_float4(pnorm(sp[0]), pnorm(sp[1]), pnorm(sp[2]), pnorm(sp[3]));
        sp[0] = o.x; sp[1] = o.y; sp[2] = o.z; sp[3] = o.w;
        if (bad) o = make_float4(qnan, qnan, qnan, qnan);   // 0/0 at :74 poisons every statistic of the cloud
        *reinterpret_cast<float4*>(out + e) = o;
    }
    if (fu.ssq) {
        float* s_red = reinterpret_cast<float*>(s_bad) + 4;  // [8][20] partials
        __syncthreads();
        const float t = slice_ssq(tid, gcount, s_red, [&](int g, int ch) { return s_stage[g * kFP + ch]; });
        if (tid < kF) fu.ssq[((size_t)c * kSlices + sl) * kF + tid] = bad ? qnan : t;
    }
    MFV_STAMP(4);
}


// ---------------------------------------------------------------------------------------------------------
// Round 4: the same forward with less VALU work per (Gaussian, point) pair -- the statistics section of the kernel above is
// VALU-issue bound (tools/mfv_stamps.py, profiles/r04_mfv_stamps.txt: 11k of 20.6k cycles at B = 32, 12-16k at B = 64 where two
// workgroups share a CU's VALUs), and of its ~90 instructions per Gaussian and point group half were the merge of the eight point groups.
//   * lanes: lane & 15 <-> Gaussian (16 per wave), lane >> 4 <-> a QUARTER of the points: two merge stages (xor 16, xor 32) for 16
//     Gaussians instead of three for 8 -- a third of the merge work per Gaussian; 512-thread workgroups (8 waves x 16 Gaussians = the
//     slice of 128), so a CU holds the same number of waves as before;
//   * points in PAIRS on the packed fp32 VALU (v_pk_mul_f32 / v_pk_add_f32: two lanes' worth of products and running sums per
//     instruction): the tables hold {z(n), z(n+1), q(n), q(n+1)} as one float4 per axis entry (one ds_read_b128 per axis and pair);
//     max / min have no packed form and stay scalar.
// Same expressions per element as the kernel above (Q = (qx qy) qz, (Q - w) / den, Q z, Q (z z - 1)); the running SUMS are taken over
// even and odd points separately and over four groups instead of eight, i.e. in another order: the statistics differ from the kernel
// above in the last bits (both are within the oracle bars; DPD_MFV_V1=1 selects the old kernel).  Needs N % 8 == 0.
// dynamic LDS (floats): zq4[3*(N/2)*m*4] | S[3*N] | minz2[3*N] | stage[slice*21] | flag | red[8*20]    (same size as above)
// ---------------------------------------------------------------------------------------------------------
constexpr int kFwd2Threads = 512;
typedef float mfv_f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(kFwd2Threads) void mfv3d_fwd2_kernel(const float* __restrict__ pts, float* __restrict__ fv, MfvConst k, int gslice,
                                                                  MfvFuse fu) {
    MFV_STAMP(0);
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int N = k.N, G = k.G, m = k.m, NP = N >> 1;
    float* s_t = sm;                                // [3][N/2][m][4] = {z(2p), z(2p+1), q(2p), q(2p+1)}
    float* s_S = sm + 6 * N * m;                    // [3][N]
    float* s_mz = s_S + 3 * N;                      // [3][N] min_i z^2
    float* s_stage = s_mz + 3 * N;                  // [gslice][21]
    int* s_bad = reinterpret_cast<int*>(s_stage + gslice * kFP);
    auto slot = [&](int a, int n, int i) { return ((a * NP + (n >> 1)) * m + i) * 4 + (n & 1); };    // z; q at + 2

    const int tid = threadIdx.x, c = blockIdx.x / kSlices, sl = blockIdx.x % kSlices;
    const int g0 = sl * gslice, gcount = max(0, min(G, g0 + gslice) - g0);
    const int lane = tid & 63, wave = tid >> 6;
    const float* p = pts ? pts + (size_t)c * N * 3 : (c < fu.B ? fu.pcA + (size_t)c * N * 3 : fu.pcB + (size_t)(c - fu.B) * N * 3);
    const float* nz = (!pts && fu.noise && c < fu.B) ? fu.noise + (size_t)c * N * 3 : nullptr;
    if (tid == 0) *s_bad = 0;
    const int lg_m = lg_or_neg(m), lg_n = lg_or_neg(N);
    // m == 8 (the reference's grid) and whole passes: the 8 entries (a, n, 0..7) of a point and axis sit in 8 consecutive lanes, so the row sum S,
    // the normalised q = e / S and min_i z^2 come out of three DPP steps in registers -- no second pass over the table, no barrier in between
    const bool dpp8 = m == 8 && (3 * N * m) % kFwd2Threads == 0;
    if (dpp8) {
        // (no unroll pragma: the trip count is a run-time value and the body holds convergent DPP operations, so hipcc cannot peel a
        //  remainder -- a requested `unroll 3` was silently refused with a -Wpass-failed warning.  Three iterations at N = 64, 3.0k of the
        //  workgroup's 20.6k cycles (profiles/r04_mfv_stamps.txt): not where this kernel's time goes)
        for (int e = tid; e < 3 * N * m; e += kFwd2Threads) {
            const int em = e >> 3, i = e & 7, a = qdiv(em, N, lg_n), n = em - a * N;
            const float x = nz ? p[n * 3 + a] + nz[n * 3 + a] : p[n * 3 + a];
            const float z = (x - k.ax.c[i]) / k.sigma;               // (batch_points - batch_mu) / batch_sig  (:87)
            const float ex = expf(-0.5f * (z * z));
            float S = ex, mz = z * z, bad = (z != z) ? 1.f : 0.f;
            S += dpp_move<0xB1>(S); mz = fminf(mz, dpp_move<0xB1>(mz)); bad += dpp_move<0xB1>(bad);        // quad_perm [1,0,3,2]
            S += dpp_move<0x4E>(S); mz = fminf(mz, dpp_move<0x4E>(mz)); bad += dpp_move<0x4E>(bad);        // quad_perm [2,3,0,1]
            S += dpp_move<0x141>(S); mz = fminf(mz, dpp_move<0x141>(mz)); bad += dpp_move<0x141>(bad);     // row_half_mirror: the other quad of the 8
            const int o = slot(a, n, i);
            s_t[o] = z;
            s_t[o + 2] = ex / S;
            if (i == 0) s_mz[em] = bad > 0.f ? __int_as_float(0x7fc00000) : mz;
        }
    } else {
        for (int e = tid; e < 3 * N * m; e += kFwd2Threads) {
            const int em = qdiv(e, m, lg_m), i = e - em * m, a = qdiv(em, N, lg_n), n = em - a * N;
            const float x = nz ? p[n * 3 + a] + nz[n * 3 + a] : p[n * 3 + a];
            const float z = (x - k.ax.c[i]) / k.sigma;               // (batch_points - batch_mu) / batch_sig  (:87)
            const int o = slot(a, n, i);
            s_t[o] = z;
            s_t[o + 2] = expf(-0.5f * (z * z));
        }
    }
    if (!pts && sl == 0) {      // the stacked tensors the rest of the step reads
        const int twin = c < fu.B ? c + fu.B : c - fu.B;
        for (int e = tid; e < 3 * N; e += kFwd2Threads) {
            const float raw = p[e];
            if (fu.pts_out) fu.pts_out[(size_t)c * N * 3 + e] = nz ? raw + nz[e] : raw;
            if (fu.q_out) fu.q_out[(size_t)twin * N * 3 + e] = raw;
        }
    }
    __syncthreads();
    MFV_STAMP(1);
    if (!dpp8) {
        for (int e = tid; e < 3 * N; e += kFwd2Threads) {             // e = a*N + n
            const int a = qdiv(e, N, lg_n), n = e - a * N;
            float S = 0.f, mz = INFINITY;
            for (int i = 0; i < m; ++i) {
                const int o = slot(a, n, i);
                const float zz = s_t[o];
                S += s_t[o + 2];
                mz = fminf(mz, zz * zz);
                if (zz != zz) mz = zz;
            }
            s_S[e] = S;
            s_mz[e] = mz;
        }
        __syncthreads();
        for (int e = tid; e < 3 * N * m; e += kFwd2Threads) {
            const int em = qdiv(e, m, lg_m), i = e - em * m, a = qdiv(em, N, lg_n), n = em - a * N;
            const int o = slot(a, n, i) + 2;
            s_t[o] = s_t[o] / s_S[em];
        }
    }
    for (int n = tid; n < N; n += kFwd2Threads) {
        const float pmax = pdf(k, sqrtf(s_mz[n]), sqrtf(s_mz[N + n]), sqrtf(s_mz[2 * N + n]));
        if (!(pmax * k.w > 0.f)) atomicOr(s_bad, 1);             // the reference's 0/0 (also catches NaN inputs)
    }
    __syncthreads();
    MFV_STAMP(2);
    const float4* tx = reinterpret_cast<const float4*>(s_t);
    const float4* ty = tx + NP * m;
    const float4* tz = tx + 2 * NP * m;

    const float invN = 1.0f / (float)N;
    const float inv_dpi = 1.0f / k.dpi_den;
    const int grp = lane >> 4;                      // quarter of the points
    const int ppg = NP >> 2;                        // point pairs per group (N % 8 == 0)
    const mfv_f2 w2 = {k.w, k.w}, inv2 = {inv_dpi, inv_dpi}, one2 = {1.0f, 1.0f};
    for (int gbase = wave * 16; gbase < gcount; gbase += (kFwd2Threads / 64) * 16) {
        const int gl = gbase + (lane & 15);
        const bool live = gl < gcount;
        const int gg = live ? g0 + gl : 0;
        const int gm = qdiv(gg, m, lg_m), t = gg - gm * m, i = qdiv(gm, m, lg_m), j = gm - i * m;   // centre (x,y,z) = (l[j], l[i], l[t])  (:47-48)
        mfv_f2 pi_s2 = {0.f, 0.f}, mu_s2[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}}, sg_s2[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
        float pi_mx = -INFINITY;
        float mu_mx[3] = {-INFINITY, -INFINITY, -INFINITY}, mu_mn[3] = {INFINITY, INFINITY, INFINITY};
        float sg_mx[3] = {-INFINITY, -INFINITY, -INFINITY}, sg_mn[3] = {INFINITY, INFINITY, INFINITY};
        // four pairs per trip with their twelve table reads requested up front: with two waves per SIMD nothing else hides the LDS latency
        // (3.3k cycles for eight pairs before: tools/mfv_stamps.py); a partial last trip re-reads its last valid pair and skips the update
        const int pp_end = (grp + 1) * ppg;
        for (int pp0 = grp * ppg; pp0 < pp_end; pp0 += 4) {
          float4 lx[4], ly[4], lz[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
              const int pq = min(pp0 + u, pp_end - 1);
              lx[u] = tx[pq * m + j]; ly[u] = ty[pq * m + i]; lz[u] = tz[pq * m + t];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (pp0 + u >= pp_end) break;
            const float4 vx = lx[u], vy = ly[u], vz = lz[u];
            const mfv_f2 z[3] = {{vx.x, vx.y}, {vy.x, vy.y}, {vz.x, vz.y}};
            const mfv_f2 qx = {vx.z, vx.w}, qy = {vy.z, vy.w}, qz = {vz.z, vz.w};
            const mfv_f2 Q = (qx * qy) * qz;                                   // :73-74, factorised
            const mfv_f2 dpi = (Q - w2) * inv2;                                // :78
            pi_s2 += dpi;
            pi_mx = fmaxf(fmaxf(pi_mx, dpi.x), dpi.y);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const mfv_f2 a = Q * z[d];                                      // :87
                const mfv_f2 b = Q * (z[d] * z[d] - one2);                      // :100
                mu_s2[d] += a; sg_s2[d] += b;
                mu_mx[d] = fmaxf(fmaxf(mu_mx[d], a.x), a.y); mu_mn[d] = fminf(fminf(mu_mn[d], a.x), a.y);
                sg_mx[d] = fmaxf(fmaxf(sg_mx[d], b.x), b.y); sg_mn[d] = fminf(fminf(sg_mn[d], b.x), b.y);
            }
          }
        }
        MFV_STAMP(5);
        float pi_s = pi_s2.x + pi_s2.y, mu_s[3], sg_s[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) { mu_s[d] = mu_s2[d].x + mu_s2[d].y; sg_s[d] = sg_s2[d].x + sg_s2[d].y; }
        // merge the four point groups (lanes l, l ^ 16, l ^ 32, l ^ 48 hold the same Gaussian): xor 16 through ds_swizzle, xor 32 as a
        // v_permlane32_swap (both halves receive lower + upper)
        auto x16 = [](float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F)); };
        auto sum32 = [](float v) { const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                                   return __uint_as_float(r[0]) + __uint_as_float(r[1]); };
        auto max32 = [](float v) { const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                                   return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])); };
        auto min32 = [](float v) { const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                                   return fminf(__uint_as_float(r[0]), __uint_as_float(r[1])); };
        pi_s += x16(pi_s); pi_s = sum32(pi_s);
        pi_mx = fmaxf(pi_mx, x16(pi_mx)); pi_mx = max32(pi_mx);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            mu_s[d] += x16(mu_s[d]); mu_s[d] = sum32(mu_s[d]);
            sg_s[d] += x16(sg_s[d]); sg_s[d] = sum32(sg_s[d]);
            mu_mx[d] = fmaxf(mu_mx[d], x16(mu_mx[d])); mu_mx[d] = max32(mu_mx[d]);
            mu_mn[d] = fminf(mu_mn[d], x16(mu_mn[d])); mu_mn[d] = min32(mu_mn[d]);
            sg_mx[d] = fmaxf(sg_mx[d], x16(sg_mx[d])); sg_mx[d] = max32(sg_mx[d]);
            sg_mn[d] = fminf(sg_mn[d], x16(sg_mn[d])); sg_mn[d] = min32(sg_mn[d]);
        }
        MFV_STAMP(6);
        if (live && grp == 0) {
            float v[kF];
            v[0] = pi_s * invN;                                                 // :81 reduce_mean
            v[1] = pi_mx;                                                       // :80
#pragma unroll
            for (int d = 0; d < 3; ++d) {                                       // :89-98, :102-109
                v[2 + d] = (mu_s[d] * invN) * k.mu_scale;
                v[5 + d] = mu_mx[d] * k.mu_scale;
                v[8 + d] = mu_mn[d] * k.mu_scale;
                v[11 + d] = (sg_s[d] * invN) * k.sig_scale;
                v[14 + d] = sg_mx[d] * k.sig_scale;
                v[17 + d] = sg_mn[d] * k.sig_scale;
            }
#pragma unroll
            for (int f = 0; f < kF; ++f) s_stage[gl * kFP + f] = v[f];
        }
    }
    MFV_STAMP(7);
    __syncthreads();
    MFV_STAMP(3);
    // power-1/2 normalisation (:119-121) + coalesced store + the slice's sums of squares: as in mfv3d_fwd_kernel
    float* out = fv + ((size_t)c * G + g0) * kF;
    const bool bad = *s_bad != 0;
    const float qnan = __int_as_float(0x7fc00000);
    for (int i4 = tid; i4 < gcount * kF / 4; i4 += kFwd2Threads) {
        const int e = i4 * 4, g = e / kF, f = e % kF;
        float* sp = s_stage + g * kFP + f;
        float4 o = make_float4(pnorm(sp[0]), pnorm(sp[1]), pnorm(sp[2]), pnorm(sp[3]));
        sp[0] = o.x; sp[1] = o.y; sp[2] = o.z; sp[3] = o.w;
        if (bad) o = make_float4(qnan, qnan, qnan, qnan);
        *reinterpret_cast<float4*>(out + e) = o;
    }
    if (fu.ssq) {
        float* s_red = reinterpret_cast<float*>(s_bad) + 4;  // [8][20] partials
        __syncthreads();
        const float t = slice_ssq(tid, gcount, s_red, [&](int g, int ch) { return s_stage[g * kFP + ch]; });
        if (tid < kF) fu.ssq[((size_t)c * kSlices + sl) * kF + tid] = bad ? qnan : t;
    }
    MFV_STAMP(4);
}

}  // namespace dpd
#ifdef DPD_ABLATIONS
extern "C" int dpd_debug_mfv_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(dpd::g_mfv_stamps), sizeof(unsigned long long) * 1024 * 8, 0, hipMemcpyDeviceToHost);
}
#endif
namespace dpd {
// L2 normalisation over the Gaussian axis, per channel (:124-126), in place.  One 256-thread block per cloud.  The per-channel
// sums are taken slice by slice in slice_ssq order and the slices added in order: the same bits as the fused form (ssq from the
// forward kernel + scale applied by the window gather).
__device__ __forceinline__ float l2_scale(float ss) { return 1.0f / sqrtf(fmaxf(ss, 1e-12f)); }   // x * rsqrt(max(sum x^2, eps))

__global__ __launch_bounds__(1024) void mfv3d_norm_kernel(float* __restrict__ fv, int G, int gslice) {
    // thread (slice, part, ch): the same partial sums, in the same order, as slice_ssq computes inside the forward kernel
    __shared__ float s_red[kSlices][8 * kF];
    __shared__ float s_scale[kF];
    const int tid = threadIdx.x, c = blockIdx.x;
    float* base = fv + (size_t)c * G * kF;
    if (tid < kSlices * 8 * kF) {
        const int sl = tid / (8 * kF), part = (tid / kF) % 8, ch = tid % kF;
        const int g0 = sl * gslice, gcount = max(0, min(G, g0 + gslice) - g0);
        const int per = (gcount + 7) / 8, g1 = min(gcount, (part + 1) * per);
        float ss = 0.f;
        for (int g = part * per; g < g1; ++g) { const float x = base[(size_t)(g0 + g) * kF + ch]; ss += x * x; }
        s_red[sl][part * kF + ch] = ss;
    }
    __syncthreads();
    if (tid < kF) {
        float total = 0.f;
        for (int sl = 0; sl < kSlices; ++sl) {
            float t = 0.f;
            for (int part = 0; part < 8; ++part) t += s_red[sl][part * kF + tid];
            total += t;
        }
        s_scale[tid] = l2_scale(total);
    }
    __syncthreads();
    float4* b4 = reinterpret_cast<float4*>(base);
    for (int i = tid; i < G * 5; i += 1024) {
        const int part = i % 5;
        float4 v = b4[i];
        v.x *= s_scale[part * 4]; v.y *= s_scale[part * 4 + 1]; v.z *= s_scale[part * 4 + 2]; v.w *= s_scale[part * 4 + 3];
        b4[i] = v;
    }
}

static int make_const(int N, int m, float sigma, MfvConst& k) {
    if (m < 1 || m > 10) return DPD_E_UNSUPPORTED;
    if (N < 1 || N > 4096) return DPD_E_UNSUPPORTED;
    if (!(sigma > 0.f)) return DPD_E_DIM;
    k.ax = make_axis(m);
    k.N = N; k.m = m; k.G = m * m * m;
    k.sigma = sigma;
    const double D = 3.0;
    k.lognorm = (float)(0.5 * D * log(2.0 * M_PI) + D * log((double)sigma));
    k.w = 1.0f / (float)k.G;
    k.dpi_den = sqrtf(k.w) * (float)N;
    k.mu_scale = 1.0f / sqrtf(k.w);
    k.sig_scale = 1.0f / sqrtf(2.0f * k.w);
    return 0;
}

static size_t fwd_lds_bytes(int N, int m, int gslice) {
    return (size_t)(6 * N * m + 6 * N + gslice * kFP + 4 + 8 * kF) * sizeof(float);
}

// ------------------------------------------------------------------------------------------------------
// Backward: dfv [C,G,20] -> dpts [C,N,3].  Everything the forward produced is recomputed from the same
// factorised tables instead of being saved.  Chain (TF autodiff of :69-126):
//   fv = s * rsqrt(ss_f),  s = pnorm(v)    -> ds = rs*dfv - s * <s,dfv>_G * rs^3            (tf.nn.l2_normalize)
//   v  = stat(raw) * const                 -> d raw: mean -> 1/N to every point, max/min -> ties share evenly
//   raw statistics depend on Q_ng, z_ng    -> dQ_ng and the direct part of dz_ng
//   Q_ng = wp_ng / sum_g wp_ng             -> through the normalisation: dz_ngd -= z_ngd Q_ng (dQ_ng - T_n),
//                                             T_n = sum_g dQ_ng Q_ng ;   dx_n = sum_g dz_ng / sigma
// Same mapping as the forward (1024 threads; wave <-> 32 Gaussians, lane>>5 <-> half of the points), so the per-
// Gaussian gradient record (20 + 20 floats) lives in REGISTERS; the per-point sums over Gaussians are 32-lane
// xor-shuffle reductions + a fixed-order sum over the 16 waves.  Needs G <= 512 (m <= 8).
// dynamic LDS (floats): zq[6*N*m] | S[3*N] | chred[16*40] | ch[40] | T[N] | part[16*N*3]
// ------------------------------------------------------------------------------------------------------
struct PQ {   // per-(point, Gaussian) quantities from the tables (identical expressions in every pass -> exact tie tests)
    float z[3], a[3], b[3], Q, dpi;
};

__device__ __forceinline__ PQ eval_pq(const float2 vx, const float2 vy, const float2 vz, float w, float inv_dpi) {
    PQ r;
    r.z[0] = vx.x; r.z[1] = vy.x; r.z[2] = vz.x;
    r.Q = (vx.y * vy.y) * vz.y;
    r.dpi = (r.Q - w) * inv_dpi;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        r.a[d] = r.Q * r.z[d];
        r.b[d] = r.Q * (r.z[d] * r.z[d] - 1.0f);
    }
    return r;
}

__global__ __launch_bounds__(kFwdThreads) void mfv3d_bwd_kernel(const float* __restrict__ pts, const float* __restrict__ dfv,
                                                                 float* __restrict__ dpts, MfvConst k) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int N = k.N, G = k.G, m = k.m;
    float2* s_zq = reinterpret_cast<float2*>(sm);   // [3][N][m]
    float* s_S = sm + 6 * N * m;                    // [3][N]
    float* s_chred = s_S + 3 * N;                   // [16][40]
    float* s_ch = s_chred + 16 * 2 * kF;            // [40]: ss_f, dot_f
    float* s_T = s_ch + 2 * kF;                     // [N]
    float* s_part = s_T + N;                        // [16][N][3]

    const int tid = threadIdx.x, c = blockIdx.x;
    const int lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const float* p = pts + (size_t)c * N * 3;
    for (int e = tid; e < 3 * N * m; e += kFwdThreads) {
        const int a = e / (N * m), n = (e / m) % N, i = e % m;
        const float z = (p[n * 3 + a] - k.ax.c[i]) / k.sigma;
        s_zq[e] = make_float2(z, expf(-0.5f * (z * z)));
    }
    __syncthreads();
    for (int e = tid; e < 3 * N; e += kFwdThreads) {
        float S = 0.f;
        for (int i = 0; i < m; ++i) S += s_zq[e * m + i].y;
        s_S[e] = S;
    }
    __syncthreads();
    for (int e = tid; e < 3 * N * m; e += kFwdThreads) s_zq[e].y = s_zq[e].y / s_S[e / m];
    __syncthreads();
    MFV_STAMP(2);
    const float2* zqx = s_zq;
    const float2* zqy = s_zq + N * m;
    const float2* zqz = s_zq + 2 * N * m;

    const int g = wave * 32 + (lane & 31);
    const bool live = g < G;
    const int gg = live ? g : 0;
    const int gi = gg / (m * m), gj = (gg / m) % m, gt = gg % m;
    const float invN = 1.0f / (float)N, inv_dpi = 1.0f / k.dpi_den;
    const int hpts = (N + 1) / 2;
    const int nbeg = half * hpts, nend = min(N, (half + 1) * hpts);

    // ---- P1: statistics (both half-waves end up with the merged values) ----------------------------------
    float raw[kF];
    {
        float pi_s = 0.f, pi_mx = -INFINITY;
        float mu_s[3] = {0.f, 0.f, 0.f}, mu_mx[3] = {-INFINITY, -INFINITY, -INFINITY}, mu_mn[3] = {INFINITY, INFINITY, INFINITY};
        float sg_s[3] = {0.f, 0.f, 0.f}, sg_mx[3] = {-INFINITY, -INFINITY, -INFINITY}, sg_mn[3] = {INFINITY, INFINITY, INFINITY};
        for (int n = nbeg; n < nend; ++n) {
            const PQ q = eval_pq(zqx[n * m + gj], zqy[n * m + gi], zqz[n * m + gt], k.w, inv_dpi);
            pi_s += q.dpi; pi_mx = fmaxf(pi_mx, q.dpi);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                mu_s[d] += q.a[d]; mu_mx[d] = fmaxf(mu_mx[d], q.a[d]); mu_mn[d] = fminf(mu_mn[d], q.a[d]);
                sg_s[d] += q.b[d]; sg_mx[d] = fmaxf(sg_mx[d], q.b[d]); sg_mn[d] = fminf(sg_mn[d], q.b[d]);
            }
        }
        pi_s = sum_x32(pi_s);
        pi_mx = max_x32(pi_mx);
        raw[0] = pi_s * invN; raw[1] = pi_mx;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            mu_s[d] = sum_x32(mu_s[d]);
            sg_s[d] = sum_x32(sg_s[d]);
            raw[2 + d] = mu_s[d] * invN;
            raw[5 + d] = max_x32(mu_mx[d]);
            raw[8 + d] = min_x32(mu_mn[d]);
            raw[11 + d] = sg_s[d] * invN;
            raw[14 + d] = max_x32(sg_mx[d]);
            raw[17 + d] = min_x32(sg_mn[d]);
        }
    }
    // ---- channel sums ss_f = sum_g s^2, dot_f = sum_g s*dfv -------------------------------------------------
    const float* df = dfv + ((size_t)c * G + gg) * kF;
    float dr[kF];   // becomes the gradient w.r.t. the raw statistics
    {
        const bool mine = live && half == 0;
#pragma unroll
        for (int f = 0; f < kF; ++f) {
            const float cst = (f < 2) ? 1.0f : ((f < 11) ? k.mu_scale : k.sig_scale);
            const float sv = pnorm(raw[f] * cst);
            const float dy = mine ? df[f] : 0.f;
            const float a = wave_sum(mine ? sv * sv : 0.f), b = wave_sum(sv * dy);
            if (lane == 0) { s_chred[wave * 2 * kF + f] = a; s_chred[wave * 2 * kF + kF + f] = b; }
        }
    }
    __syncthreads();
    if (tid < 2 * kF) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += s_chred[w * 2 * kF + tid];
        s_ch[tid] = t;
    }
    __syncthreads();
    // ---- P2: dfv -> gradient w.r.t. the raw statistics -------------------------------------------------------
#pragma unroll
    for (int f = 0; f < kF; ++f) {
        const float cst = (f < 2) ? 1.0f : ((f < 11) ? k.mu_scale : k.sig_scale);
        const float v = raw[f] * cst;
        const float sv = pnorm(v);
        const float ss = s_ch[f], dot = s_ch[kF + f];
        const float dy = live ? df[f] : 0.f;
        float ds;
        if (ss >= 1e-12f) {
            const float rs = 1.0f / sqrtf(ss);
            ds = rs * dy - sv * dot * rs * rs * rs;
        } else {
            ds = dy * 1e6f;                       // clamp active: y = s * rsqrt(1e-12)
        }
        // s = sign(v) max(|v|,1e-12)^0.5: gradient reaches v only where |v| >= 1e-12 (tf.maximum), tf.sign has none
        float dv = 0.f;
        if (fabsf(v) >= 1e-12f) dv = ds * 0.5f / sqrtf(fabsf(v));
        float d = dv * cst;
        if (f == 0 || (f >= 2 && f < 5) || (f >= 11 && f < 14)) d *= invN;   // reduce_mean: 1/N to every point
        dr[f] = live ? d : 0.f;
    }
    // ---- P1b: tie counts of the max/min statistics (tf.reduce_max/min split the gradient evenly) --------------
    {
        float cnt[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) cnt[i] = 0.f;
        for (int n = nbeg; n < nend; ++n) {
            const PQ q = eval_pq(zqx[n * m + gj], zqy[n * m + gi], zqz[n * m + gt], k.w, inv_dpi);
            cnt[0] += (q.dpi == raw[1]) ? 1.f : 0.f;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                cnt[1 + d] += (q.a[d] == raw[5 + d]) ? 1.f : 0.f;
                cnt[4 + d] += (q.a[d] == raw[8 + d]) ? 1.f : 0.f;
                cnt[7 + d] += (q.b[d] == raw[14 + d]) ? 1.f : 0.f;
                cnt[10 + d] += (q.b[d] == raw[17 + d]) ? 1.f : 0.f;
            }
        }
        const int mm[13] = {1, 5, 6, 7, 8, 9, 10, 14, 15, 16, 17, 18, 19};
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            const float ctot = sum_x32(cnt[i]);
            dr[mm[i]] = dr[mm[i]] / fmaxf(ctot, 1.f);
        }
    }
    // ---- P3: T_n = sum_g dQ_ng Q_ng ---------------------------------------------------------------------------
    for (int n = nbeg; n < nend; ++n) {
        const PQ q = eval_pq(zqx[n * m + gj], zqy[n * m + gi], zqz[n * m + gt], k.w, inv_dpi);
        float dQ = (dr[0] + ((q.dpi == raw[1]) ? dr[1] : 0.f)) * inv_dpi;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float ga = dr[2 + d] + ((q.a[d] == raw[5 + d]) ? dr[5 + d] : 0.f) + ((q.a[d] == raw[8 + d]) ? dr[8 + d] : 0.f);
            const float gb = dr[11 + d] + ((q.b[d] == raw[14 + d]) ? dr[14 + d] : 0.f) + ((q.b[d] == raw[17 + d]) ? dr[17 + d] : 0.f);
            dQ += ga * q.z[d] + gb * (q.z[d] * q.z[d] - 1.0f);
        }
        const float t = half_sum32_hi(live ? dQ * q.Q : 0.f);
        if ((lane & 31) == 31) s_part[wave * N + n] = t;
    }
    __syncthreads();
    for (int n = tid; n < N; n += kFwdThreads) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += s_part[w * N + n];
        s_T[n] = t;
    }
    __syncthreads();
    // ---- P4: dz_ngd = Q (ga + 2 gb z) - z Q (dQ - T_n), summed over the Gaussians ------------------------------
    for (int n = nbeg; n < nend; ++n) {
        const PQ q = eval_pq(zqx[n * m + gj], zqy[n * m + gi], zqz[n * m + gt], k.w, inv_dpi);
        float dQ = (dr[0] + ((q.dpi == raw[1]) ? dr[1] : 0.f)) * inv_dpi;
        float ga[3], gb[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            ga[d] = dr[2 + d] + ((q.a[d] == raw[5 + d]) ? dr[5 + d] : 0.f) + ((q.a[d] == raw[8 + d]) ? dr[8 + d] : 0.f);
            gb[d] = dr[11 + d] + ((q.b[d] == raw[14 + d]) ? dr[14 + d] : 0.f) + ((q.b[d] == raw[17 + d]) ? dr[17 + d] : 0.f);
            dQ += ga[d] * q.z[d] + gb[d] * (q.z[d] * q.z[d] - 1.0f);
        }
        const float u = dQ - s_T[n];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float dz = live ? q.Q * (ga[d] + 2.0f * gb[d] * q.z[d]) - q.z[d] * q.Q * u : 0.f;
            const float t = half_sum32_hi(dz);
            if ((lane & 31) == 31) s_part[(wave * N + n) * 3 + d] = t;
        }
    }
    __syncthreads();
    float* out = dpts + (size_t)c * N * 3;
    for (int i = tid; i < N * 3; i += kFwdThreads) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += s_part[w * N * 3 + i];
        out[i] = t / k.sigma;   // z = (x - mu)/sigma
    }
}

// ------------------------------------------------------------------------------------------------------
// Backward, sliced over the POINTS: kSlices workgroups per cloud (a single workgroup per cloud leaves 3/4 of the chip idle
// at the as-loss batch sizes: 32 clouds at B = 16).  Everything that is a sum over Gaussians for a fixed point (T_n, dx_n)
// is local to the point's slice; the only cross-slice quantity is the per-Gaussian statistic record (7 sums, 7 maxima,
// 6 minima over the points + how many points attain each extreme), exchanged through `part`:
//   mfv3d_bwd_stats_kernel   slice -> part[c][s][33][G]    (sum / max / min over the slice's points, local tie counts)
//   mfv3d_bwd_combine_kernel one workgroup per cloud: merges the records, channel sums, d raw statistics (in place over slice 0)
//   mfv3d_bwd_apply_kernel   (was: every slice combined the kSlices records itself; ties: counts of the slices whose extreme equals the global
//                            one), then runs P2-P4 of mfv3d_bwd_kernel for its own points and writes their dpts.
// Same per-(point, Gaussian) expressions as the monolithic kernel -> identical tie tests; sums differ by association.
// ------------------------------------------------------------------------------------------------------
constexpr int kRec = 33;   // 20 statistics + 13 tie counts

__device__ __forceinline__ void build_tables(const float* p, int n0, int np_, const MfvConst& k, float2* s_zq, float* s_S, int tid) {
    const int m = k.m;
    const int lg_m = lg_or_neg(m), lg_np = lg_or_neg(np_);
    for (int e = tid; e < 3 * np_ * m; e += kFwdThreads) {
        const int em = qdiv(e, m, lg_m), i = e - em * m, a = qdiv(em, np_, lg_np), ln = em - a * np_;
        const float z = (p[(n0 + ln) * 3 + a] - k.ax.c[i]) / k.sigma;
        s_zq[e] = make_float2(z, expf(-0.5f * (z * z)));
    }
    __syncthreads();
    for (int e = tid; e < 3 * np_; e += kFwdThreads) {
        float S = 0.f;
        for (int i = 0; i < m; ++i) S += s_zq[e * m + i].y;
        s_S[e] = S;
    }
    __syncthreads();
    for (int e = tid; e < 3 * np_ * m; e += kFwdThreads) s_zq[e].y = s_zq[e].y / s_S[qdiv(e, m, lg_m)];
    __syncthreads();
}

__device__ __forceinline__ void mfv3d_bwd_stats_block(const float* __restrict__ pts, float* __restrict__ part, const MfvConst& k, int nslice,
                                                      int c, int sl, float* sm) {
    const int N = k.N, G = k.G, m = k.m;
    const int n0 = min(N, sl * nslice), np_ = min(N, n0 + nslice) - n0;
    float2* s_zq = reinterpret_cast<float2*>(sm);   // [3][np_][m]
    float* s_S = sm + 6 * nslice * m;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    build_tables(pts + (size_t)c * N * 3, n0, np_, k, s_zq, s_S, tid);
    const float2* zqx = s_zq;
    const float2* zqy = s_zq + np_ * m;
    const float2* zqz = s_zq + 2 * np_ * m;
    const int g = wave * 32 + (lane & 31);
    const bool live = g < G;
    const int gg = live ? g : 0;
    const int gi = gg / (m * m), gj = (gg / m) % m, gt = gg % m;
    const float inv_dpi = 1.0f / k.dpi_den;
    const int hpts = (np_ + 1) / 2;
    const int nbeg = half * hpts, nend = min(np_, (half + 1) * hpts);
    float rec[kRec];
    {
        float pi_s = 0.f, pi_mx = -INFINITY;
        float mu_s[3] = {0.f, 0.f, 0.f}, mu_mx[3] = {-INFINITY, -INFINITY, -INFINITY}, mu_mn[3] = {INFINITY, INFINITY, INFINITY};
        float sg_s[3] = {0.f, 0.f, 0.f}, sg_mx[3] = {-INFINITY, -INFINITY, -INFINITY}, sg_mn[3] = {INFINITY, INFINITY, INFINITY};
        for (int n = nbeg; n < nend; ++n) {
            const PQ q = eval_pq(zqx[n * m + gj], zqy[n * m + gi], zqz[n * m + gt], k.w, inv_dpi);
            pi_s += q.dpi; pi_mx = fmaxf(pi_mx, q.dpi);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                mu_s[d] += q.a[d]; mu_mx[d] = fmaxf(mu_mx[d], q.a[d]); mu_mn[d] = fminf(mu_mn[d], q.a[d]);
                sg_s[d] += q.b[d]; sg_mx[d] = fmaxf(sg_mx[d], q.b[d]); sg_mn[d] = fminf(sg_mn[d], q.b[d]);
            }
        }
        rec[0] = sum_x32(pi_s);            // SUMS here (the mean's 1/N is applied after the combine)
        rec[1] = max_x32(pi_mx);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            rec[2 + d] = sum_x32(mu_s[d]);
            rec[5 + d] = max_x32(mu_mx[d]);
            rec[8 + d] = min_x32(mu_mn[d]);
            rec[11 + d] = sum_x32(sg_s[d]);
            rec[14 + d] = max_x32(sg_mx[d]);
            rec[17 + d] = min_x32(sg_mn[d]);
        }
    }
    {
        float cnt[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) cnt[i] = 0.f;
        for (int n = nbeg; n < nend; ++n) {
            const PQ q = eval_pq(zqx[n * m + gj], zqy[n * m + gi], zqz[n * m + gt], k.w, inv_dpi);
            cnt[0] += (q.dpi == rec[1]) ? 1.f : 0.f;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                cnt[1 + d] += (q.a[d] == rec[5 + d]) ? 1.f : 0.f;
                cnt[4 + d] += (q.a[d] == rec[8 + d]) ? 1.f : 0.f;
                cnt[7 + d] += (q.b[d] == rec[14 + d]) ? 1.f : 0.f;
                cnt[10 + d] += (q.b[d] == rec[17 + d]) ? 1.f : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 13; ++i) rec[20 + i] = sum_x32(cnt[i]);
    }
    if (live && half == 0) {
        float* out = part + ((size_t)(c * kSlices + sl) * kRec) * G + g;
#pragma unroll
        for (int i = 0; i < kRec; ++i) out[(size_t)i * G] = rec[i];
    }
}

__global__ __launch_bounds__(kFwdThreads) void mfv3d_bwd_stats_kernel(const float* __restrict__ pts, float* __restrict__ part,
                                                                       MfvConst k, int nslice) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    mfv3d_bwd_stats_block(pts, part, k, nslice, blockIdx.x / kSlices, blockIdx.x % kSlices, sm);
}

// As-loss backward (asloss.hip): the window-gather backward (dX -> dfv; patch_rows_bwd.h) and the statistics pass above read nothing of each
// other, so they share ONE launch -- workgroups [0, ngather) gather, the rest take the statistics of their point slice (the two used to be
// 17 + 12 us of latency-bound launches back to back at the PCRNet batch).  Same routines, same bits.
__global__ __launch_bounds__(kFwdThreads) void asloss_tail_a_kernel(const float* __restrict__ dX, const int32_t* __restrict__ vox,
                                                                     float* __restrict__ dfv, const float* __restrict__ pts,
                                                                     float* __restrict__ part, MfvConst k, int nslice, int kwin, int KP,
                                                                     int gslices, int ngather) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    if ((int)blockIdx.x < ngather) {
        patch_rows_bwd_block<kFwdThreads>(dX, vox, dfv, k.N, k.m, kwin, KP, blockIdx.x / gslices, blockIdx.x % gslices, gslices,
                                          reinterpret_cast<int*>(sm));
    } else {
        const int b = blockIdx.x - ngather;
        mfv3d_bwd_stats_block(pts, part, k, nslice, b / kSlices, b % kSlices, sm);
    }
}

// Per-cloud combine (thread = Gaussian): merges the kSlices statistic records, takes the channel sums over
// ALL Gaussians, and turns dfv into the gradient w.r.t. the 20 raw statistics of every Gaussian.  This is the sqrt / divide heavy
// part of the backward; it used to be repeated by both half-waves of every one of the kSlices apply workgroups (8x).  Output, in
// place over slice 0's record of the cloud (every thread only touches its own column g): rows 0..19 = d raw statistic (tie
// counts already divided in), rows 20..32 = the 13 global extrema the apply kernel needs for its tie tests.
// (round 3) FOUR workgroups per cloud, five of the 20 statistics each: the channel sums are per statistic, so the quarters never talk to
// each other, every thread requests 24-40 record values instead of 132 (the loads of one workgroup per cloud went through ONE CU's
// address unit: ~17k cycles of this kernel's 18.9 us at the PCRNet batch), and 128 workgroups instead of 32 share the square roots.
// Same expressions per statistic, same summation orders: the same bits.
__device__ __forceinline__ constexpr bool mfv_is_sum(int f) { return f == 0 || (f >= 2 && f < 5) || (f >= 11 && f < 14); }
__device__ __forceinline__ constexpr bool mfv_is_max(int f) { return f == 1 || (f >= 5 && f < 8) || (f >= 14 && f < 17); }
// index of statistic f among the 13 extrema {1, 5..10, 14..19} (tie-count / extremum rows 20 + index), -1 for a sum
__device__ __forceinline__ constexpr int mfv_tie_index(int f) { return f == 1 ? 0 : (f >= 5 && f < 11 ? f - 4 : (f >= 14 ? f - 7 : -1)); }

template <int F0>
__device__ __forceinline__ void mfv3d_bwd_combine_quarter(const float* __restrict__ dfv, float* __restrict__ part, const MfvConst& k, int c,
                                                          float* s_chred, float* s_ch) {
    constexpr int NF = 5;
    const int N = k.N, G = k.G;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = tid;
    const bool live = g < G;
    const int gg = live ? g : 0;
    const float invN = 1.0f / (float)N;
    const float* pc = part + (size_t)c * kSlices * kRec * G + gg;
    float r[kSlices][NF], rt[kSlices][NF];
#pragma unroll
    for (int s2 = 0; s2 < kSlices; ++s2)
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            constexpr int dummy = 0; (void)dummy;
            r[s2][j] = pc[((size_t)s2 * kRec + (F0 + j)) * G];
            rt[s2][j] = 0.f;
        }
#pragma unroll
    for (int s2 = 0; s2 < kSlices; ++s2)
#pragma unroll
        for (int j = 0; j < NF; ++j)
            if (mfv_tie_index(F0 + j) >= 0) rt[s2][j] = pc[((size_t)s2 * kRec + 20 + mfv_tie_index(F0 + j)) * G];
    float raw[NF], cnt[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int f = F0 + j;
        float v = r[0][j];
#pragma unroll
        for (int s2 = 1; s2 < kSlices; ++s2) v = mfv_is_sum(f) ? v + r[s2][j] : (mfv_is_max(f) ? fmaxf(v, r[s2][j]) : fminf(v, r[s2][j]));
        raw[j] = mfv_is_sum(f) ? v * invN : v;
        float t = 0.f;      // tie count of an extremum: the slices that attain it contribute theirs
#pragma unroll
        for (int s2 = 0; s2 < kSlices; ++s2) t += (r[s2][j] == raw[j]) ? rt[s2][j] : 0.f;
        cnt[j] = t;
    }
    // ---- channel sums ss_f = sum_g s^2, dot_f = sum_g s*dfv ----
    const float* df = dfv + ((size_t)c * G + gg) * kF;
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int f = F0 + j;
        const float cst = (f < 2) ? 1.0f : ((f < 11) ? k.mu_scale : k.sig_scale);
        const float sv = pnorm(raw[j] * cst);
        const float dy = live ? df[f] : 0.f;
        const float a = wave_sum(live ? sv * sv : 0.f), b = wave_sum(sv * dy);
        if (lane == 0) { s_chred[wave * 2 * NF + j] = a; s_chred[wave * 2 * NF + NF + j] = b; }
    }
    __syncthreads();
    if (tid < 2 * NF) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += s_chred[w * 2 * NF + tid];
        s_ch[tid] = t;
    }
    __syncthreads();
    float* out = part + (size_t)c * kSlices * kRec * G + g;      // slice 0's record of this cloud (rows F0 .. F0+4 and their extremum rows)
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int f = F0 + j;
        const float cst = (f < 2) ? 1.0f : ((f < 11) ? k.mu_scale : k.sig_scale);
        const float v = raw[j] * cst;
        const float sv = pnorm(v);
        const float ss = s_ch[j], dot = s_ch[NF + j];
        const float dy = live ? df[f] : 0.f;
        float ds;
        if (ss >= 1e-12f) {
            const float rs = 1.0f / sqrtf(ss);
            ds = rs * dy - sv * dot * rs * rs * rs;
        } else {
            ds = dy * 1e6f;
        }
        float dv = 0.f;
        if (fabsf(v) >= 1e-12f) dv = ds * 0.5f / sqrtf(fabsf(v));
        float d = dv * cst;
        if (mfv_is_sum(f)) d *= invN;
        d = live ? d : 0.f;
        if (mfv_tie_index(f) >= 0) d = d / fmaxf(cnt[j], 1.f);
        if (live) {
            out[(size_t)f * G] = d;
            if (mfv_tie_index(f) >= 0) out[(size_t)(20 + mfv_tie_index(f)) * G] = raw[j];
        }
    }
}

// NOTE on in-place output: quarter q overwrites rows F0..F0+4 and the extremum rows of ITS statistics in slice 0's record, which no
// other quarter reads (each reads only its own statistic and tie rows), so the four workgroups of a cloud need no ordering.
__global__ __launch_bounds__(512) void mfv3d_bwd_combine_kernel(const float* __restrict__ dfv, float* __restrict__ part, MfvConst k) {
    __shared__ float s_chred[8 * 2 * 5];
    __shared__ float s_ch[2 * 5];
    const int c = blockIdx.x >> 2;
    switch (blockIdx.x & 3) {
        case 0: mfv3d_bwd_combine_quarter<0>(dfv, part, k, c, s_chred, s_ch); break;
        case 1: mfv3d_bwd_combine_quarter<5>(dfv, part, k, c, s_chred, s_ch); break;
        case 2: mfv3d_bwd_combine_quarter<10>(dfv, part, k, c, s_chred, s_ch); break;
        default: mfv3d_bwd_combine_quarter<15>(dfv, part, k, c, s_chred, s_ch); break;
    }
}

// FINAL (as-loss backward): instead of dpts, the evaluation's two input gradients leave this kernel (= dpd_asloss_combine on its output):
//   clouds c < B are pcA, queried by the BA half of the rows; c >= B are pcB, queried by the AB half (models/dpdist_and_aue.py:56-69):
//   g[r, d] = (dpts[c, n, d] + dX[qrow, E + d]) * upstream
struct AslossFinal {
    const float* dX;
    const float* scale;
    float* gA;
    float* gB;
    int B, KP, E;
};
template <bool FINAL>
__global__ __launch_bounds__(kFwdThreads) void mfv3d_bwd_apply_kernel(const float* __restrict__ pts, const float* __restrict__ comb,
                                                                       float* __restrict__ dpts, MfvConst k, int nslice, AslossFinal fin) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int N = k.N, G = k.G, m = k.m;
    const int c = blockIdx.x / kSlices, sl = blockIdx.x % kSlices;
    const int n0 = min(N, sl * nslice), np_ = min(N, n0 + nslice) - n0;
    float2* s_zq = reinterpret_cast<float2*>(sm);   // [3][np_][m]
    float* s_S = sm + 6 * nslice * m;               // [3][nslice]
    float* s_T = s_S + 3 * nslice;                  // [nslice]
    float* s_part = s_T + nslice;                   // [16][nslice][3]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    build_tables(pts + (size_t)c * N * 3, n0, np_, k, s_zq, s_S, tid);
    const float2* zqx = s_zq;
    const float2* zqy = s_zq + np_ * m;
    const float2* zqz = s_zq + 2 * np_ * m;
    const int g = wave * 32 + (lane & 31);
    const bool live = g < G;
    const int gg = live ? g : 0;
    const int gi = gg / (m * m), gj = (gg / m) % m, gt = gg % m;
    const float inv_dpi = 1.0f / k.dpi_den;
    const int hpts = (np_ + 1) / 2;
    const int nbeg = half * hpts, nend = min(np_, (half + 1) * hpts);
    // d raw statistic and the global extrema of this lane's Gaussian, from mfv3d_bwd_combine_kernel
    float dr[kF], raw[kF];
    {
        const float* pc = comb + (size_t)c * kSlices * kRec * G + gg;
        const int mm[13] = {1, 5, 6, 7, 8, 9, 10, 14, 15, 16, 17, 18, 19};
#pragma unroll
        for (int f = 0; f < kF; ++f) { dr[f] = live ? pc[(size_t)f * G] : 0.f; raw[f] = 0.f; }
#pragma unroll
        for (int i = 0; i < 13; ++i) raw[mm[i]] = pc[(size_t)(20 + i) * G];
    }
    // ---- T_n and dz for the slice's own points (as P3 / P4 of mfv3d_bwd_kernel) ----------------------------------
    for (int n = nbeg; n < nend; ++n) {
        const PQ q = eval_pq(zqx[n * m + gj], zqy[n * m + gi], zqz[n * m + gt], k.w, inv_dpi);
        float dQ = (dr[0] + ((q.dpi == raw[1]) ? dr[1] : 0.f)) * inv_dpi;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float ga = dr[2 + d] + ((q.a[d] == raw[5 + d]) ? dr[5 + d] : 0.f) + ((q.a[d] == raw[8 + d]) ? dr[8 + d] : 0.f);
            const float gb = dr[11 + d] + ((q.b[d] == raw[14 + d]) ? dr[14 + d] : 0.f) + ((q.b[d] == raw[17 + d]) ? dr[17 + d] : 0.f);
            dQ += ga * q.z[d] + gb * (q.z[d] * q.z[d] - 1.0f);
        }
        const float t = half_sum32_hi(live ? dQ * q.Q : 0.f);
        if ((lane & 31) == 31) s_part[wave * nslice + n] = t;
    }
    __syncthreads();
    for (int n = tid; n < np_; n += kFwdThreads) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += s_part[w * nslice + n];
        s_T[n] = t;
    }
    __syncthreads();
    for (int n = nbeg; n < nend; ++n) {
        const PQ q = eval_pq(zqx[n * m + gj], zqy[n * m + gi], zqz[n * m + gt], k.w, inv_dpi);
        float dQ = (dr[0] + ((q.dpi == raw[1]) ? dr[1] : 0.f)) * inv_dpi;
        float ga[3], gb[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            ga[d] = dr[2 + d] + ((q.a[d] == raw[5 + d]) ? dr[5 + d] : 0.f) + ((q.a[d] == raw[8 + d]) ? dr[8 + d] : 0.f);
            gb[d] = dr[11 + d] + ((q.b[d] == raw[14 + d]) ? dr[14 + d] : 0.f) + ((q.b[d] == raw[17 + d]) ? dr[17 + d] : 0.f);
            dQ += ga[d] * q.z[d] + gb[d] * (q.z[d] * q.z[d] - 1.0f);
        }
        const float u = dQ - s_T[n];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float dz = live ? q.Q * (ga[d] + 2.0f * gb[d] * q.z[d]) - q.z[d] * q.Q * u : 0.f;
            const float t = half_sum32_hi(dz);
            if ((lane & 31) == 31) s_part[(wave * nslice + n) * 3 + d] = t;
        }
    }
    __syncthreads();
    float* out = FINAL ? nullptr : dpts + ((size_t)c * N + n0) * 3;
    for (int i = tid; i < np_ * 3; i += kFwdThreads) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += s_part[w * nslice * 3 + i];
        const float v = t / k.sigma;
        if (FINAL) {
            const int which = c >= fin.B ? 1 : 0, BN = fin.B * N;
            const int r = (c - which * fin.B) * N + n0 + i / 3, d = i % 3;
            const int qrow = which ? r : BN + r;
            const float sc = fin.scale ? *fin.scale : 1.0f;
            (which ? fin.gB : fin.gA)[(size_t)r * 3 + d] = (v + fin.dX[(size_t)qrow * fin.KP + fin.E + d]) * sc;
        } else {
            out[i] = v;
        }
    }
}

static size_t bwd_sliced_lds_bytes(int nslice, int m) {
    return (size_t)(6 * nslice * m + 3 * nslice + nslice + 16 * nslice * 3 + 4) * sizeof(float);
}
static size_t bwd_lds_bytes(int N, int m) {
    return (size_t)(6 * N * m + 3 * N + 16 * 2 * kF + 2 * kF + N + 16 * N * 3 + 4) * sizeof(float);
}

template <typename K>
static int set_lds(K kern, size_t lds) {
    static LdsOptIn lds_opt;   // one per kernel (template on the kernel's type)
    return ensure_dyn_lds(lds_opt, (const void*)kern, lds);
}

}  // namespace dpd

// round-4 forward kernel (pairs of points on the packed VALU, four point groups): N % 8 == 0; DPD_MFV_V1=1 keeps the round-3 kernel
static bool use_fwd2(int N) {
    static const bool v1 = getenv("DPD_MFV_V1") != nullptr;
    return !v1 && N >= 8 && !(N & 7);
}

extern "C" int dpd_mfv3d_fwd(const float* pts, int C, int N, int m, float sigma, float* fv, void* stream) {
    using namespace dpd;
    if (!pts || !fv) return DPD_E_NULL;
    if (C <= 0) return DPD_E_DIM;
    MfvConst k{};
    if (int rc = make_const(N, m, sigma, k)) return rc;
    const int gslice = (k.G + kSlices - 1) / kSlices;
    const size_t lds = fwd_lds_bytes(N, m, gslice);
    StageProf prof(stream, DPD_STAGE_ENCODER, (double)C * (N * 12.0 + k.G * 80.0));      // points in, [G,20] Fisher vector out
    if (use_fwd2(N)) {
        if (int rc = set_lds(mfv3d_fwd2_kernel, lds)) return rc;
        DPD_LAUNCH(mfv3d_fwd2_kernel, dim3(C * kSlices), dim3(kFwd2Threads), lds, (hipStream_t)stream, pts, fv, k, gslice, MfvFuse{});
    } else {
        if (int rc = set_lds(mfv3d_fwd_kernel, lds)) return rc;
        DPD_LAUNCH(mfv3d_fwd_kernel, dim3(C * kSlices), dim3(kFwdThreads), lds, (hipStream_t)stream, pts, fv, k, gslice, MfvFuse{});
    }
    DPD_CHECK_LAUNCH();
    DPD_LAUNCH(mfv3d_norm_kernel, dim3(C), dim3(1024), 0, (hipStream_t)stream, fv, k.G, gslice);
    DPD_CHECK_LAUNCH();
    return 0;
}

extern "C" int dpd_mfv3d_fwd_stacked(const float* pcA, const float* pcB, const float* noise, int B, int N, int m, float sigma,
                                     float* pts, float* q, float* fv, float* ssq, void* stream) {
    using namespace dpd;
    if (!pcA || !pcB || !fv) return DPD_E_NULL;
    if (B <= 0) return DPD_E_DIM;
    MfvConst k{};
    if (int rc = make_const(N, m, sigma, k)) return rc;
    const int C = 2 * B, gslice = (k.G + kSlices - 1) / kSlices;
    const size_t lds = fwd_lds_bytes(N, m, gslice);
    // points (+ noise) in; stacked pts / q and the [G,20] Fisher vector (+ per-slice sums of squares) out
    StageProf prof(stream, DPD_STAGE_ENCODER, (double)C * (N * 12.0 * (noise ? 1.5 : 1.0) + N * 24.0 + k.G * 80.0 + (ssq ? kSlices * 80.0 : 0.0)));
    if (use_fwd2(N)) {
        if (int rc = set_lds(mfv3d_fwd2_kernel, lds)) return rc;
        DPD_LAUNCH(mfv3d_fwd2_kernel, dim3(C * kSlices), dim3(kFwd2Threads), lds, (hipStream_t)stream, (const float*)nullptr, fv, k, gslice,
                   MfvFuse{pcA, pcB, noise, pts, q, ssq, B});
    } else {
        if (int rc = set_lds(mfv3d_fwd_kernel, lds)) return rc;
        DPD_LAUNCH(mfv3d_fwd_kernel, dim3(C * kSlices), dim3(kFwdThreads), lds, (hipStream_t)stream, (const float*)nullptr, fv, k, gslice,
                   MfvFuse{pcA, pcB, noise, pts, q, ssq, B});
    }
    DPD_CHECK_LAUNCH();
    if (!ssq) {
        DPD_LAUNCH(mfv3d_norm_kernel, dim3(C), dim3(1024), 0, (hipStream_t)stream, fv, k.G, gslice);
        DPD_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" size_t dpd_mfv3d_bwd_workspace_bytes(int C, int m) {
    return (size_t)(C > 0 ? C : 0) * dpd::kSlices * dpd::kRec * (size_t)(m * m * m) * sizeof(float);
}

extern "C" int dpd_mfv3d_bwd(const float* pts, const float* dfv, int C, int N, int m, float sigma, float* dpts, void* ws,
                             size_t ws_bytes, void* stream) {
    using namespace dpd;
    if (!pts || !dfv || !dpts) return DPD_E_NULL;
    if (C <= 0) return DPD_E_DIM;
    MfvConst k{};
    if (int rc = make_const(N, m, sigma, k)) return rc;
    if (k.G > 512) return DPD_E_UNSUPPORTED;   // per-Gaussian gradient record lives in registers: one Gaussian per lane pair
    if (ws && ws_bytes >= dpd_mfv3d_bwd_workspace_bytes(C, m) && N >= 2 * kSlices) {
        // sliced over the points: kSlices workgroups per cloud, per-Gaussian statistic records exchanged through `ws`
        const int nslice = (N + kSlices - 1) / kSlices;
        const size_t l1 = (size_t)(6 * nslice * m + 3 * nslice + 4) * sizeof(float), l2 = bwd_sliced_lds_bytes(nslice, m);
        if (int rc = set_lds(mfv3d_bwd_stats_kernel, l1)) return rc;
        if (int rc = set_lds(mfv3d_bwd_apply_kernel<false>, l2)) return rc;
        DPD_LAUNCH(mfv3d_bwd_stats_kernel, dim3(C * kSlices), dim3(kFwdThreads), l1, (hipStream_t)stream, pts, (float*)ws, k, nslice);
        DPD_CHECK_LAUNCH();
        DPD_LAUNCH(mfv3d_bwd_combine_kernel, dim3(C * 4), dim3(512), 0, (hipStream_t)stream, dfv, (float*)ws, k);
        DPD_CHECK_LAUNCH();
        DPD_LAUNCH(mfv3d_bwd_apply_kernel<false>, dim3(C * kSlices), dim3(kFwdThreads), l2, (hipStream_t)stream, pts, (const float*)ws, dpts, k,
                   nslice, AslossFinal{});
        DPD_CHECK_LAUNCH();
        return 0;
    }
    const size_t lds = bwd_lds_bytes(N, m);
    if (int rc = set_lds(mfv3d_bwd_kernel, lds)) return rc;
    DPD_LAUNCH(mfv3d_bwd_kernel, dim3(C), dim3(kFwdThreads), lds, (hipStream_t)stream, pts, dfv, dpts, k);
    DPD_CHECK_LAUNCH();
    return 0;
}

// The non-GEMM tail of an as-loss backward in THREE launches instead of six (dpd_patch_rows_bwd + dpd_mfv3d_bwd's three + dpd_asloss_combine):
//   [window-gather backward || encoder statistics]  ->  combine  ->  apply + the two input gradients.
// Bitwise the separate calls (same device routines; tests/test_gpu_parity.py).  DPD_E_UNSUPPORTED for shapes the sliced encoder backward
// does not take (the caller then makes the separate calls).
extern "C" int dpd_asloss_tail(const float* dX, const int32_t* vox, const float* pts, const float* upstream, int B, int N, int m, int k, int KP,
                               float sigma, float* dfv, void* mfv_ws, size_t mfv_ws_bytes, float* gA, float* gB, void* stream) {
    using namespace dpd;
    if (!dX || !vox || !pts || !dfv || !mfv_ws || !gA || !gB) return DPD_E_NULL;
    if (B <= 0 || N <= 0) return DPD_E_DIM;
    if (k < 1 || k > 7 || !(k & 1) || N > 8192) return DPD_E_UNSUPPORTED;
    if (KP < k * k * k * kF + 3 || (KP & 3)) return DPD_E_DIM;
    const int C = 2 * B;
    MfvConst kc{};
    if (int rc = make_const(N, m, sigma, kc)) return rc;
    if (kc.G > 512 || N < 2 * kSlices || mfv_ws_bytes < dpd_mfv3d_bwd_workspace_bytes(C, m)) return DPD_E_UNSUPPORTED;
    const int nslice = (N + kSlices - 1) / kSlices;
    const int gslices = (kc.G * 5 + kFwdThreads - 1) / kFwdThreads;      // one (voxel, channel group) item per thread
    const size_t l1 = (size_t)(6 * nslice * m + 3 * nslice + 4) * sizeof(float), lg = (size_t)N * sizeof(int), l2 = bwd_sliced_lds_bytes(nslice, m);
    if (int rc = set_lds(asloss_tail_a_kernel, l1 > lg ? l1 : lg)) return rc;
    static LdsOptIn opt_final;      // (set_lds keys its flag on the kernel's TYPE, which the two instantiations of the apply kernel share)
    if (int rc = ensure_dyn_lds(opt_final, (const void*)mfv3d_bwd_apply_kernel<true>, l2)) return rc;
    hipStream_t s = (hipStream_t)stream;
    DPD_LAUNCH(asloss_tail_a_kernel, dim3(C * gslices + C * kSlices), dim3(kFwdThreads), l1 > lg ? l1 : lg, s, dX, vox, dfv, pts, (float*)mfv_ws, kc,
               nslice, k, KP, gslices, C * gslices);
    DPD_CHECK_LAUNCH();
    DPD_LAUNCH(mfv3d_bwd_combine_kernel, dim3(C * 4), dim3(512), 0, s, (const float*)dfv, (float*)mfv_ws, kc);
    DPD_CHECK_LAUNCH();
    DPD_LAUNCH(mfv3d_bwd_apply_kernel<true>, dim3(C * kSlices), dim3(kFwdThreads), l2, s, pts, (const float*)mfv_ws, (float*)nullptr, kc, nslice,
               AslossFinal{dX, upstream, gA, gB, B, KP, k * k * k * kF});
    DPD_CHECK_LAUNCH();
    return 0;
}
