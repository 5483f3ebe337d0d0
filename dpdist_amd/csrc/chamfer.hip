// Chamfer distance between two point sets (the baseline loss of the AUE task).
//
// Replaces pairwise_diff + chmafer_dist (train_multi_gpu_pc_compare_dist.py:891-916):
//     d(x, y)[b, i, j] = |x_bi - y_bj|^2                                      (:896-905, squared, no sqrt)
//     loss = ( mean_bi min_j d(rec, pc) + mean_bj min_i d(pc, rec) ) / 2      (:913-915)
// without materialising the [B, N, 3, M] tiles of the reference.  One workgroup per (cloud, direction, 256-point
// chunk): the other cloud is staged in LDS, every thread scans it for its own point.  Tiny and latency bound
// (N = M = 64 ... 2048); algorithmic HBM bytes = 12 (N + M) read + 8 (N + M) written per cloud.
// Ties in the minimum (measure zero for float clouds) go to the lowest index; tf.reduce_min would split the
// gradient evenly among them.
#include "common.h"

namespace dpd {

// mins[b][i] = min_j |x_bi - y_bj|^2, arg[b][i] = that j.  dir 0: x = a (N points), y = b (M); dir 1: x = b, y = a.
__global__ __launch_bounds__(256) void chamfer_min_kernel(const float* __restrict__ a, const float* __restrict__ b, int N, int M,
                                                          float* __restrict__ min_a, int32_t* __restrict__ arg_a,
                                                          float* __restrict__ min_b, int32_t* __restrict__ arg_b, int chunks_a,
                                                          int chunks_b) {
    extern __shared__ float s_y[];   // [ny][3]
    const int per = chunks_a + chunks_b;
    const int c = blockIdx.x / per, r = blockIdx.x % per;
    const bool dir = r >= chunks_a;
    const int chunk = dir ? r - chunks_a : r;
    const int nx = dir ? M : N, ny = dir ? N : M;
    const float* x = (dir ? b : a) + (size_t)c * nx * 3;
    const float* y = (dir ? a : b) + (size_t)c * ny * 3;
    for (int e = threadIdx.x; e < ny * 3; e += 256) s_y[e] = y[e];
    __syncthreads();
    const int i = chunk * 256 + threadIdx.x;
    if (i >= nx) return;
    const float px = x[i * 3], py = x[i * 3 + 1], pz = x[i * 3 + 2];
    float best = INFINITY;
    int bj = 0;
    for (int j = 0; j < ny; ++j) {
        const float dx = px - s_y[j * 3], dy = py - s_y[j * 3 + 1], dz = pz - s_y[j * 3 + 2];
        const float d = (dx * dx + dy * dy) + dz * dz;      // reduce_sum over the 3 coordinates, in order (:905)
        if (d < best) { best = d; bj = j; }
    }
    (dir ? min_b : min_a)[(size_t)c * nx + i] = best;
    (dir ? arg_b : arg_a)[(size_t)c * nx + i] = bj;
}

// loss = (mean(min_a) + mean(min_b)) / 2, fixed summation order (one workgroup)
__global__ __launch_bounds__(256) void chamfer_loss_kernel(const float* __restrict__ min_a, long na, const float* __restrict__ min_b,
                                                           long nb, float* __restrict__ loss) {
    __shared__ float red[2][256];
    float sa = 0.f, sb = 0.f;
    for (long i = threadIdx.x; i < na; i += 256) sa += min_a[i];
    for (long i = threadIdx.x; i < nb; i += 256) sb += min_b[i];
    red[0][threadIdx.x] = sa;
    red[1][threadIdx.x] = sb;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (red[0][0] / (float)na + red[1][0] / (float)nb) / 2.0f;
}

// Gradient w.r.t. x (dir 0: a, dir 1: b) as a gather (deterministic, no atomics):
//   dx_i = g * [ wx * 2 (x_i - y_arg_x[i])  +  wy * sum_{j : arg_y[j] == i} 2 (x_i - y_j) ],  wx = 1/(2 B nx), wy = 1/(2 B ny)
__global__ __launch_bounds__(256) void chamfer_grad_kernel(const float* __restrict__ a, const float* __restrict__ b, int N, int M,
                                                           const int32_t* __restrict__ arg_a, const int32_t* __restrict__ arg_b,
                                                           float gscale, int B, float* __restrict__ da, float* __restrict__ db,
                                                           int chunks_a, int chunks_b) {
    extern __shared__ float s_buf[];   // [ny][3] then [ny] arg (as int)
    const int per = chunks_a + chunks_b;
    const int c = blockIdx.x / per, r = blockIdx.x % per;
    const bool dir = r >= chunks_a;
    const int chunk = dir ? r - chunks_a : r;
    const int nx = dir ? M : N, ny = dir ? N : M;
    float* dx = dir ? db : da;
    if (!dx) return;
    const float* x = (dir ? b : a) + (size_t)c * nx * 3;
    const float* y = (dir ? a : b) + (size_t)c * ny * 3;
    const int32_t* arg_x = (dir ? arg_b : arg_a) + (size_t)c * nx;
    const int32_t* arg_y = (dir ? arg_a : arg_b) + (size_t)c * ny;
    float* s_y = s_buf;
    int* s_arg = reinterpret_cast<int*>(s_buf + ny * 3);
    for (int e = threadIdx.x; e < ny * 3; e += 256) s_y[e] = y[e];
    for (int e = threadIdx.x; e < ny; e += 256) s_arg[e] = arg_y[e];
    __syncthreads();
    const int i = chunk * 256 + threadIdx.x;
    if (i >= nx) return;
    const float wx = gscale / (2.0f * (float)B * (float)nx), wy = gscale / (2.0f * (float)B * (float)ny);
    const float px = x[i * 3], py = x[i * 3 + 1], pz = x[i * 3 + 2];
    const int j0 = arg_x[i];
    float gx = wx * 2.f * (px - s_y[j0 * 3]), gy = wx * 2.f * (py - s_y[j0 * 3 + 1]), gz = wx * 2.f * (pz - s_y[j0 * 3 + 2]);
    for (int j = 0; j < ny; ++j) {
        if (s_arg[j] == i) {
            gx += wy * 2.f * (px - s_y[j * 3]);
            gy += wy * 2.f * (py - s_y[j * 3 + 1]);
            gz += wy * 2.f * (pz - s_y[j * 3 + 2]);
        }
    }
    dx[((size_t)c * nx + i) * 3] = gx;
    dx[((size_t)c * nx + i) * 3 + 1] = gy;
    dx[((size_t)c * nx + i) * 3 + 2] = gz;
}

}  // namespace dpd

extern "C" int dpd_chamfer_fwd(const float* a, const float* b, int B, int N, int M, float* min_a, int32_t* arg_a, float* min_b,
                               int32_t* arg_b, float* loss, void* stream) {
    using namespace dpd;
    if (!a || !b || !min_a || !arg_a || !min_b || !arg_b || !loss) return DPD_E_NULL;
    if (B <= 0 || N <= 0 || M <= 0) return DPD_E_DIM;
    if (N > 4096 || M > 4096) return DPD_E_UNSUPPORTED;   // the other cloud lives in LDS (48 KB)
    const int ca = (N + 255) / 256, cb = (M + 255) / 256;
    const size_t lds = (size_t)(N > M ? N : M) * 3 * sizeof(float);
    DPD_LAUNCH(chamfer_min_kernel, dim3(B * (ca + cb)), dim3(256), lds, (hipStream_t)stream, a, b, N, M, min_a, arg_a, min_b, arg_b,
               ca, cb);
    DPD_CHECK_LAUNCH();
    DPD_LAUNCH(chamfer_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)min_a, (long)B * N,
               (const float*)min_b, (long)B * M, loss);
    DPD_CHECK_LAUNCH();
    return 0;
}

extern "C" int dpd_chamfer_bwd(const float* a, const float* b, int B, int N, int M, const int32_t* arg_a, const int32_t* arg_b,
                               float gscale, float* da, float* db, void* stream) {
    using namespace dpd;
    if (!a || !b || !arg_a || !arg_b || (!da && !db)) return DPD_E_NULL;
    if (B <= 0 || N <= 0 || M <= 0) return DPD_E_DIM;
    if (N > 4096 || M > 4096) return DPD_E_UNSUPPORTED;
    const int ca = (N + 255) / 256, cb = (M + 255) / 256;
    const size_t lds = (size_t)(N > M ? N : M) * 4 * sizeof(float);
    DPD_LAUNCH(chamfer_grad_kernel, dim3(B * (ca + cb)), dim3(256), lds, (hipStream_t)stream, a, b, N, M, arg_a, arg_b, gscale, B, da,
               db, ca, cb);
    DPD_CHECK_LAUNCH();
    return 0;
}
