// Shared helpers for the dpdist_hip kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dpdist_capi.h"

// hipGetLastError() is sticky per host thread: a benign error left behind by another library (e.g. the framework
// probing devices) must not be reported as ours, so the state is cleared right before each launch.
#define DPD_LAUNCH(...)                             \
    do {                                            \
        (void)hipGetLastError();                    \
        hipLaunchKernelGGL(__VA_ARGS__);            \
    } while (0)

#define DPD_CHECK_LAUNCH()                          \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

#define DPD_HIP(call)                               \
    do {                                            \
        hipError_t e__ = (call);                    \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

#include <atomic>

namespace dpd {

constexpr int kMfvSlices = DPD_MFV_SLICES;   // workgroups per cloud in the encoder forward = partial norms per cloud

// Opt a kernel into more than 64 KiB of dynamic LDS, once per (kernel, device): `slot` is a per-call-site static array of
// flags indexed by the CURRENT device, so a process that drives several GPUs configures each of them (the attribute is
// per device), and concurrent first calls are a benign repeat of an idempotent setting.
struct LdsOptIn {
    std::atomic<bool> done[64];
};
inline int ensure_dyn_lds(LdsOptIn& slot, const void* kern, size_t lds) {
    if (lds <= 64 * 1024) return 0;
    if (lds > 160 * 1024) return DPD_E_UNSUPPORTED;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    dev = (dev < 0 || dev >= 64) ? 63 : dev;
    if (slot.done[dev].load(std::memory_order_acquire)) return 0;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    slot.done[dev].store(true, std::memory_order_release);
    return 0;
}

// In-stream stage profiler (gemm_f32.hip; dpd_prof_enable(2)): the bandwidth-bound kernels of the step bracket their launch with an
// event pair and record their ALGORITHMIC HBM bytes (every input read once, every output written once) under a stage tag; bench.py
// turns that into the `roofline_hbm` object of its JSON line.  Off (one branch, no lock) unless the profiler was enabled with mode 2.
enum { DPD_STAGE_GEMM = 0, DPD_STAGE_ENCODER = 1, DPD_STAGE_GATHER = 2, DPD_STAGE_OUT_LAYER = 3, DPD_STAGE_OPTIMIZER = 4,
       DPD_STAGE_SMALL_REDUCE = 5, DPD_STAGE_WEIGHT_COPIES = 6, DPD_STAGE_COUNT = 7 };
bool prof_begin_stage(hipStream_t s);
void prof_end_stage(bool on, hipStream_t s, int tag, double bytes);
struct StageProf {
    bool on;
    hipStream_t s;
    int tag;
    double bytes;
    StageProf(void* stream, int tag_, double bytes_) : on(prof_begin_stage((hipStream_t)stream)), s((hipStream_t)stream), tag(tag_), bytes(bytes_) {}
    ~StageProf() { prof_end_stage(on, s, tag, bytes); }
    StageProf(const StageProf&) = delete;
    StageProf& operator=(const StageProf&) = delete;
};

constexpr int kWave = 64;   // CDNA wavefront
constexpr int kNumXCD = 8;  // MI355X: 8 XCDs, block b runs on XCD b % 8 (speed only, never correctness)

// Axis centres of the m-cell grid on [-1,1], restating numpy's arithmetic in double exactly:
//   np.linspace(-1,1,m,False)+1/m == np.arange(-1,1,2/m)+(2/m)/2 == (-1 + i*(2/m)) + 1/m
// (utils/dpdist_util.py:42 and :987-988), then cast to float32 like tf.constant(x, tf.float32).
struct GridAxis {
    float c[16];
    float half;  // |c[0]-c[1]|/2 in float32 (utils/dpdist_util.py:468)
};

inline GridAxis make_axis(int m) {
    GridAxis a{};
    const double step = 2.0 / (double)m;
    for (int i = 0; i < m && i < 16; ++i) {
        volatile double v = (double)i * step;  // volatile: no FMA contraction on the host
        v = v + (-1.0);
        v = v + 1.0 / (double)m;
        a.c[i] = (float)v;
    }
    a.half = (m > 1) ? fabsf(a.c[0] - a.c[1]) / 2.0f : 1.0f;
    return a;
}

// bijective XCD-aware remap of a 1-D block id: each XCD gets a contiguous chunk of logical ids.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk / kNumXCD, r = nblk % kNumXCD;
    const int xcd = bid % kNumXCD, loc = bid / kNumXCD;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + loc;
}

// Cross-lane sums on the DPP path (VALU, ~10 cycles per step) instead of __shfl_xor (= ds_bpermute_b32 through the LDS crossbar,
// > 100 cycles of latency per step, six dependent steps per sum).  Fixed reduction tree -> deterministic.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_move(float v) {      // lanes without a valid source (or outside ROW_MASK) read 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
// every lane of a 16-lane row gets the sum of its row
__device__ __forceinline__ float row_sum16(float v) {
    v += dpp_move<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);     // row_half_mirror
    v += dpp_move<0x140>(v);     // row_mirror
    return v;
}
// sum over the 64 lanes, returned in every lane
__device__ __forceinline__ float wave_sum(float v) {
    v = row_sum16(v);
    v += dpp_move<0x142, 0xA>(v);    // row_bcast15 into rows 1 and 3: lanes 16-31 = rows 0+1, lanes 48-63 = rows 2+3
    v += dpp_move<0x143, 0xC>(v);    // row_bcast31 into rows 2 and 3: lanes 48-63 = all four rows
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// lane <-> lane ^ 32 combinations through v_permlane32_swap (gfx950; semantics probed in tools/permlane_probe.hip: with both operands
// equal, result 0 holds the lower half-wave's value in every lane and result 1 the upper half-wave's): both halves receive
// lower (op) upper, the same bits as x (op) __shfl_xor(x, 32) for a commutative op -- without the LDS crossbar and its address register
__device__ __forceinline__ float sum_x32(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float max_x32(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float min_x32(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fminf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// sum over a 256-thread block (fixed order), returned in every thread; red = 4 floats of LDS
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return s;
}
// sum over the 32 lanes of each half-wave; valid in lanes 16-31 (lower half) and 48-63 (upper half) -- read it at lane 31 / 63
__device__ __forceinline__ float half_sum32_hi(float v) {
    v = row_sum16(v);
    v += dpp_move<0x142, 0xA>(v);
    return v;
}

}  // namespace dpd
