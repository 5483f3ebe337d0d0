// L1 / mean-prediction losses (+ their gradients) and the TF-form Adam update.  Small HBM-bound kernels.
//
//   dpd_l1_loss  replaces utils/dpdist_util.py:962-980 (get_loss) and TF's autodiff of it
//   dpd_adam_tf  replaces tf.train.AdamOptimizer (train_multi_gpu_pc_compare_dist.py:216,301): epsilon-hat form
//                algorithmic HBM bytes per parameter: 16 read (p,g,m,v) + 12 written (p,m,v)
#include "gemm_shared.h"

namespace dpd {

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

// One-launch optimizer step: Adam + everything that is derived from the result or feeds its last few elements.
//   role 1 (tile blocks): the matrices that have a transposed copy (W2, W3, optionally W1p) are updated in 64 x 64 tiles and
//           the new values leave the block twice: in place and, through an LDS transpose, as rows of the copy -> the separate
//           transpose launch (and its 2 x 8.4 MB round trip) disappears;
//   role 2 (tail blocks): p[tail_off .. tail_off + 4H + 3) = [b3 | W4 | b4] take their gradient straight from the block partials
//           of the output-layer backward (same fixed summation order as small_grads_reduce, which this replaces; the gradient is
//           also stored, and the two loss values are finished here);
//   role 3 (vector blocks): every other element, float4 grid-stride over up to four ranges.
struct AdamFuse {
    float* WT[3];
    uint16_t* rc[3];            // bf16 operand planes of the updated matrix (gemm_x3.hip formats), np planes each, or NULL
    uint16_t* r8[3];
    int np;
    long w_off[3];
    int w_rows[3], w_cols[3];
    int direct[3];              // 1: no transposed copy, rows % 8 == 0, cols % 128 == 0 -> register-only plane path (one wave = 8 rows x 128 columns)
    int tile_end[3];            // cumulative tile-block counts of the three matrices (0-size matrices repeat the previous value)
    const float* partials;      // [nparts][rec] or NULL
    int nparts, rec, H, Qb;
    long tail_off;
    float* loss;
    int ntail;                  // tail blocks (16 outputs each)
    long v_off[5], v_cnt[5];    // vector ranges: the complement of up to four covered intervals (three matrices + the tail)
};


// The gradient and the two moments are touched once per step and by nobody else: their 16-byte accesses carry the non-temporal hint
// (global_load / global_store ... nt), so that 56 MB of them do not push the operand planes and activations of the next forward out of the
// L2 / Infinity Cache.  Measured (round 6, profiles/r06_adam_nt_ab.txt): bf16 B = 64 step 0.2792 -> 0.2722 ms; f32 B = 32 unchanged
// (0.5657 / 0.5658); hinting the parameters as well (the exact type's forward reads them) gave nothing more (0.2736).
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_stream(const float* p) {
    const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4_stream(float* p, float4 x) {
    nt_f4 v;
    v.x = x.x; v.y = x.y; v.z = x.z; v.w = x.w;
    __builtin_nontemporal_store(v, reinterpret_cast<nt_f4*>(p));
}

// NP = number of bf16 planes written next to the update (0: none / transposed fp32 copies only): compile time, so that the one-plane
// type converts once per value instead of running the three-plane split and discarding two thirds of it
template <int NP>
__global__ __launch_bounds__(256) void adam_fused_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, float lr_t, float b1, float b2, float eps,
                                                          float gscale, AdamFuse f) {
    __shared__ float tile[64][65];
    int b = blockIdx.x;
    const int tid = threadIdx.x;
    if (b < f.tile_end[2]) {
        const int w = b < f.tile_end[0] ? 0 : (b < f.tile_end[1] ? 1 : 2);
        const int t = b - (w ? f.tile_end[w - 1] : 0);
        const int rows = f.w_rows[w], cols = f.w_cols[w], tc = cols / 64;
        const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
        const size_t base = (size_t)f.w_off[w];
        if (NP != 0 && f.direct[w]) {
            // Planes without the LDS tile: a wave owns 8 rows x 128 columns; lane (l31, half) updates rows 4 half .. 4 half + 3 of the columns
            // 4 l31 .. 4 l31 + 3 (float4 loads, 512 contiguous bytes per row and half-wave).  RC chunks halves (4 columns of a row) are
            // the lane's own values; an R8 chunk (8 rows of one column) is this lane's four rows plus the four of lane +- 32: one
            // v_permlane32_swap per packed dword gives the lower half the chunks of columns 0, 2 and the upper half those of 1, 3
            // (same exchange as the plane GEMM epilogue, gemm_x3.hip).  33.6 -> 33.0 us: the kernel moves its 150 MB of mixed traffic at ~4.5 TB/s
            // with or without the LDS tile (DPD_ADAM_LDS_TILES=1 keeps the tile form as an A/B reference).
            const int lane = tid & 63, l31 = lane & 31, half = lane >> 5, segs = cols >> 7;
            const int u = t * 4 + (tid >> 6);
            if (u >= (rows >> 3) * segs) return;
            const int rg = u / segs, c = (u % segs) * 128 + 4 * l31, row0 = 8 * rg + 4 * half;
            float4 P[4], G[4], M[4], V[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const size_t o = base + (size_t)(row0 + j) * cols + c;
                P[j] = *reinterpret_cast<float4*>(p + o);
                G[j] = ld4_stream(g + o);
                M[j] = ld4_stream(m + o);
                V[j] = ld4_stream(v + o);
            }
            unsigned pv[4][4][NP ? NP : 1];      // [row j][column e][plane]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const size_t o = base + (size_t)(row0 + j) * cols + c;
                adam_one(P[j].x, G[j].x * gscale, M[j].x, V[j].x, lr_t, b1, b2, eps);
                adam_one(P[j].y, G[j].y * gscale, M[j].y, V[j].y, lr_t, b1, b2, eps);
                adam_one(P[j].z, G[j].z * gscale, M[j].z, V[j].z, lr_t, b1, b2, eps);
                adam_one(P[j].w, G[j].w * gscale, M[j].w, V[j].w, lr_t, b1, b2, eps);
                *reinterpret_cast<float4*>(p + o) = P[j];
                st4_stream(m + o, M[j]);
                st4_stream(v + o, V[j]);
                const float x[4] = {P[j].x, P[j].y, P[j].z, P[j].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (NP == 1) {
                        pv[j][e][0] = bf16_bits(x[e]);
                    } else {
                        unsigned p3[3];
                        split3(x[e], p3);
#pragma unroll
                        for (int q = 0; q < NP; ++q) pv[j][e][q] = p3[q];
                    }
                }
            }
            const long plane = (long)rows * cols;
            if (uint16_t* rc = f.rc[w]) {
#pragma unroll
                for (int q = 0; q < NP; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *reinterpret_cast<uint2*>(rc + q * plane + (size_t)(row0 + j) * cols + c) =
                            make_uint2(pv[j][0][q] | (pv[j][1][q] << 16), pv[j][2][q] | (pv[j][3][q] << 16));
            }
            if (uint16_t* r8 = f.r8[w]) {
#pragma unroll
                for (int q = 0; q < NP; ++q)
#pragma unroll
                    for (int e0 = 0; e0 < 2; ++e0) {
                        // (lower half-wave: column 2 e0, upper: 2 e0 + 1 -> a lane pair writes 32 contiguous bytes per store instruction)
                        const int ca = 2 * e0, cb = 2 * e0 + 1;
                        const unsigned a0 = pv[0][ca][q] | (pv[1][ca][q] << 16), a1 = pv[2][ca][q] | (pv[3][ca][q] << 16);
                        const unsigned b0 = pv[0][cb][q] | (pv[1][cb][q] << 16), b1_ = pv[2][cb][q] | (pv[3][cb][q] << 16);
                        const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1_, false, false);
                        *reinterpret_cast<uint4*>(r8 + q * plane + ((size_t)rg * cols + c + 2 * e0 + half) * 8) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                    }
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid + 256 * k, r = idx >> 4, c4 = (idx & 15) * 4;
            if (r0 + r < rows) {
                const size_t o = base + (size_t)(r0 + r) * cols + c0 + c4;
                float4 P = *reinterpret_cast<float4*>(p + o);
                const float4 G = ld4_stream(g + o);
                float4 M = ld4_stream(m + o);
                float4 V = ld4_stream(v + o);
                adam_one(P.x, G.x * gscale, M.x, V.x, lr_t, b1, b2, eps);
                adam_one(P.y, G.y * gscale, M.y, V.y, lr_t, b1, b2, eps);
                adam_one(P.z, G.z * gscale, M.z, V.z, lr_t, b1, b2, eps);
                adam_one(P.w, G.w * gscale, M.w, V.w, lr_t, b1, b2, eps);
                *reinterpret_cast<float4*>(p + o) = P;
                st4_stream(m + o, M);
                st4_stream(v + o, V);
                tile[r][c4] = P.x; tile[r][c4 + 1] = P.y; tile[r][c4 + 2] = P.z; tile[r][c4 + 3] = P.w;
            }
        }
        __syncthreads();
        if (float* T = f.WT[w]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = tid + 256 * k, c = idx >> 4, r4 = (idx & 15) * 4;
                if (r0 + r4 < rows)      // rows % 4 == 0: a float4 of rows is valid or absent as a whole
                    *reinterpret_cast<float4*>(T + (size_t)(c0 + c) * rows + r0 + r4) =
                        make_float4(tile[r4][c], tile[r4 + 1][c], tile[r4 + 2][c], tile[r4 + 3][c]);
            }
        }
        // bf16 operand planes of the new weights (what dpd_weights_to_planes would write: same split, same layouts)
        const long plane = (long)rows * cols;
        if (uint16_t* rc = f.rc[w]) {      // RC [np][rows][cols]: a chunk = 8 consecutive columns of one row
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int idx = tid + 256 * k, r = idx >> 3, cg = (idx & 7) * 8;
                if (r0 + r < rows) {
                    float x[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = tile[r][cg + j];
                    uint4 wv[3];
                    if (NP == 1) {
                        unsigned b[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) b[j] = bf16_bits(x[j]);
                        wv[0] = make_uint4(b[0] | (b[1] << 16), b[2] | (b[3] << 16), b[4] | (b[5] << 16), b[6] | (b[7] << 16));
                    } else {
                        split_chunk(x, wv);
                    }
#pragma unroll
                    for (int q = 0; q < NP; ++q) *reinterpret_cast<uint4*>(rc + q * plane + (size_t)(r0 + r) * cols + c0 + cg) = wv[q];
                }
            }
        }
        if (uint16_t* r8 = f.r8[w]) {      // R8 [np][rows/8][cols][8]: a chunk = 8 consecutive rows of one column
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int idx = tid + 256 * k, rg = idx >> 6, c = idx & 63;
                if (r0 + 8 * rg < rows) {  // rows % 8 == 0 when planes are requested
                    float x[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = tile[8 * rg + j][c];
                    uint4 wv[3];
                    if (NP == 1) {
                        unsigned b[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) b[j] = bf16_bits(x[j]);
                        wv[0] = make_uint4(b[0] | (b[1] << 16), b[2] | (b[3] << 16), b[4] | (b[5] << 16), b[6] | (b[7] << 16));
                    } else {
                        split_chunk(x, wv);
                    }
#pragma unroll
                    for (int q = 0; q < NP; ++q) *reinterpret_cast<uint4*>(r8 + q * plane + ((size_t)((r0 >> 3) + rg) * cols + c0 + c) * 8) = wv[q];
                }
            }
        }
        return;
    }
    b -= f.tile_end[2];
    if (b < f.ntail) {
        // = small_grads_reduce (decoder.hip) followed by Adam on the 16 elements of this block
        float (*red)[17] = reinterpret_cast<float (*)[17]>(&tile[0][0]);
        float* s_loss = &tile[8][0];
        const int li = tid & 15, sl = tid >> 4, H = f.H;
        const int i = b * 16 + li;
        const int n = (f.rec > 4 * H + 4 && f.loss) ? 4 * H + 7 : 4 * H + 3;
        float s = 0.f;
        if (i < n)
            for (int k = sl; k < f.nparts; k += 16) s += f.partials[(size_t)k * f.rec + i];
        red[sl][li] = s;
        __syncthreads();
        if (sl == 0 && i < n) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[k][li];
            if (i < 4 * H + 3) {
                const size_t o = (size_t)f.tail_off + i;
                g[o] = t;
                float P = p[o], M = m[o], V = v[o];
                adam_one(P, t * gscale, M, V, lr_t, b1, b2, eps);
                p[o] = P; m[o] = M; v[o] = V;
            } else if (i >= 4 * H + 4) {
                s_loss[i - 4 * H - 4] = t;
            }
        }
        if (n > 4 * H + 3 && b == (4 * H + 4) / 16) {
            __syncthreads();
            if (tid == 0) {
                const float inv = 1.0f / (float)f.Qb;
                f.loss[0] = s_loss[0] * inv;
                f.loss[1] = (s_loss[1] * inv + s_loss[2] * inv) / 2.0f;
            }
        }
        return;
    }
    b -= f.ntail;
    const int nvec = gridDim.x - f.tile_end[2] - f.ntail;
    const size_t stride = (size_t)nvec * 256;
#pragma unroll 1
    for (int r = 0; r < 5; ++r) {
        const size_t off = (size_t)f.v_off[r], cnt = (size_t)f.v_cnt[r], n4 = (off & 3) ? 0 : cnt / 4;
        for (size_t i = (size_t)b * 256 + tid; i < n4; i += stride) {
            const size_t o = off + 4 * i;
            float4 P = *reinterpret_cast<float4*>(p + o);
            const float4 G = ld4_stream(g + o);
            float4 M = ld4_stream(m + o);
            float4 V = ld4_stream(v + o);
            adam_one(P.x, G.x * gscale, M.x, V.x, lr_t, b1, b2, eps);
            adam_one(P.y, G.y * gscale, M.y, V.y, lr_t, b1, b2, eps);
            adam_one(P.z, G.z * gscale, M.z, V.z, lr_t, b1, b2, eps);
            adam_one(P.w, G.w * gscale, M.w, V.w, lr_t, b1, b2, eps);
            *reinterpret_cast<float4*>(p + o) = P;
            st4_stream(m + o, M);
            st4_stream(v + o, V);
        }
        for (size_t i = n4 * 4 + (size_t)b * 256 + tid; i < cnt; i += stride) {
            const size_t o = off + i;
            float P = p[o], M = m[o], V = v[o];
            adam_one(P, g[o] * gscale, M, V, lr_t, b1, b2, eps);
            p[o] = P; m[o] = M; v[o] = V;
        }
    }
}

// Adam with lr_t read from DEVICE memory (state[3]), so that a captured (hipGraph) step carries no per-step host parameter: the consumer's
// optimizer (optim.TFAdam.prepare_replay writes state[3] before every replay of the registration step).
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
    dpd::StageProf prof(stream, dpd::DPD_STAGE_OPTIMIZER, 28.0 * (double)n);      // read p, g, m, v; write p, m, v
    DPD_LAUNCH(dpd::adam_tf_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr_t,
                       b1, b2, eps, gscale);
    DPD_CHECK_LAUNCH();
    return 0;
}

extern "C" int dpd_adam_tf_fused(float* p, float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2, float eps,
                                 float gscale, const dpd_adam_fuse* fu, void* stream) {
    using namespace dpd;
    if (!p || !g || !m || !v || !fu) return DPD_E_NULL;
    if (n == 0) return DPD_E_DIM;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return DPD_E_UNSUPPORTED;
    AdamFuse f{};
    // covered intervals, in ascending order of offset (the caller passes the matrices in flat order)
    long lo[4], hi[4];
    int nc = 0, tiles = 0;
    const bool planes = fu->np != 0;
    if (planes && fu->np != 1 && fu->np != 3) return DPD_E_UNSUPPORTED;
    f.np = fu->np;
    for (int w = 0; w < 3; ++w) {
        f.WT[w] = fu->WT[w]; f.w_off[w] = fu->w_off[w]; f.w_rows[w] = fu->w_rows[w]; f.w_cols[w] = fu->w_cols[w];
        f.rc[w] = planes ? (uint16_t*)fu->W_rc[w] : nullptr;
        f.r8[w] = planes ? (uint16_t*)fu->W_r8[w] : nullptr;
        if (fu->WT[w] || f.rc[w] || f.r8[w]) {
            const long cnt = (long)fu->w_rows[w] * fu->w_cols[w];
            if (fu->w_rows[w] <= 0 || fu->w_cols[w] <= 0 || (fu->w_cols[w] & 63) || (fu->w_rows[w] & 3) || (fu->w_off[w] & 3) ||
                fu->w_off[w] < 0 || (size_t)(fu->w_off[w] + cnt) > n || ((uintptr_t)fu->WT[w] & 15))
                return DPD_E_UNSUPPORTED;
            if ((f.rc[w] || f.r8[w]) && ((fu->w_rows[w] & 7) || (((uintptr_t)f.rc[w] | (uintptr_t)f.r8[w]) & 15))) return DPD_E_UNSUPPORTED;
            if (nc && fu->w_off[w] < hi[nc - 1]) return DPD_E_DIM;
            lo[nc] = fu->w_off[w]; hi[nc] = fu->w_off[w] + cnt; ++nc;
            f.direct[w] = !fu->WT[w] && (f.rc[w] || f.r8[w]) && !(fu->w_cols[w] & 127);
            if (f.direct[w]) tiles += ((fu->w_rows[w] / 8) * (fu->w_cols[w] / 128) + 3) / 4;
            else tiles += ((fu->w_rows[w] + 63) / 64) * (fu->w_cols[w] / 64);
        }
        f.tile_end[w] = tiles;
    }
    if (fu->partials) {
        const long tail = 4L * fu->H + 3;
        // the tail ends the buffer, up to 3 alignment-padding elements (zero parameters with zero gradients: Adam leaves them 0)
        if (fu->nparts <= 0 || fu->H <= 0 || fu->rec < 4 * fu->H + 4 || fu->tail_off < 0 || (size_t)(fu->tail_off + tail) > n ||
            n - (size_t)(fu->tail_off + tail) > 3)
            return DPD_E_DIM;
        if (nc && fu->tail_off < hi[nc - 1]) return DPD_E_DIM;
        if (fu->loss && (fu->rec < 4 * fu->H + 8 || fu->Qb <= 0)) return DPD_E_DIM;
        // the tail blocks own 16 parameters each; the three loss sums (elements 4H+4 .. 4H+6 of a record) must land in ONE block
        if (fu->H % 4 || (4 * fu->H + 4) % 16 > 13) return DPD_E_UNSUPPORTED;
        f.partials = fu->partials; f.nparts = fu->nparts; f.rec = fu->rec; f.H = fu->H; f.Qb = fu->Qb;
        f.tail_off = fu->tail_off; f.loss = fu->loss;
        f.ntail = (4 * fu->H + 7 + 15) / 16;
        lo[nc] = fu->tail_off; hi[nc] = (long)n; ++nc;
    }
    // the vector ranges are the complement of the covered intervals in [0, n)
    long at = 0;
    int nv = 0;
    size_t vec_elems = 0;
    for (int i = 0; i <= nc; ++i) {
        const long end = i < nc ? lo[i] : (long)n;
        if (end > at) {
            if (nv >= (int)(sizeof(f.v_off) / sizeof(f.v_off[0]))) return DPD_E_UNSUPPORTED;   // more gaps than the kernel's range table holds
            f.v_off[nv] = at; f.v_cnt[nv] = end - at; vec_elems += (size_t)(end - at); ++nv;
        }
        if (i < nc) at = hi[i];
    }
    size_t vblocks = (vec_elems / 4 + 255) / 256;
    if (vblocks > 2048) vblocks = 2048;
    if (vblocks == 0 && vec_elems) vblocks = 1;
    const unsigned grid = (unsigned)(tiles + f.ntail + vblocks);
    if (grid == 0) return DPD_E_DIM;
    // algorithmic bytes: 28 B per parameter (read p, g, m, v; write p, m, v) + the copies written in the same pass (transposed fp32: 4 B, bf16 operand planes: np x 2 B per layout) + the block partials
    double by = 0.0;
    for (int w = 0; w < 3; ++w) {
        const double cnt = (double)fu->w_rows[w] * fu->w_cols[w];
        by += cnt * ((f.WT[w] ? 4.0 : 0.0) + f.np * 2.0 * ((f.rc[w] ? 1.0 : 0.0) + (f.r8[w] ? 1.0 : 0.0)));
    }
    by += 28.0 * (double)n + (fu->partials ? (double)fu->nparts * fu->rec * 4.0 : 0.0);
    StageProf prof(stream, DPD_STAGE_OPTIMIZER, by);
    if (f.np == 1) DPD_LAUNCH(adam_fused_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, lr_t, b1, b2, eps, gscale, f);
    else if (f.np == 3) DPD_LAUNCH(adam_fused_kernel<3>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, lr_t, b1, b2, eps, gscale, f);
    else DPD_LAUNCH(adam_fused_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, lr_t, b1, b2, eps, gscale, f);
    DPD_CHECK_LAUNCH();
    return 0;
}
