// Implicit-surface decoder: shared MLP  KP(=2503 padded) -> H -> H -> H -> 3, relu6/3, mask.  Forward and backward.
//
// Replaces utils/dpdist_util.py:513-544 (four tf_util.conv2d "1xW VALID conv" == dense layers,
// utils/tf_util.py:161-228), :691 (relu6/3), :695-698 (split + mask) and TF's autodiff of them.
//
// The three wide layers run on the fp32 MFMA GEMM of gemm_f32.hip (bias+ReLU / ReLU-gate fused in the epilogue);
// the 3-wide output layer and the bias gradients are small HBM-bound kernels in this file.
// Algorithmic work per query row (H = 1024): 4 663 296 MAC forward.  Activations h1,h2,h3 [Q,H] stay in HBM
// (16.8 MB each at Q = 4096, resident in the 256 MiB Infinity Cache) for the backward pass.
#include "gemm_shared.h"

namespace dpd {

int gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
             int ldc, const float* bias, const float* gate, int epilogue, int split_k, int tile, void* ws,
             size_t ws_bytes, hipStream_t s, float* colsum = nullptr, const float* A2 = nullptr, const float* B2 = nullptr,
             float* C2 = nullptr, const ColsumTwoStep* cs2 = nullptr);

int gemm_x3(int np, int a_fmt, int b_fmt, int M, int N, int K, const uint16_t* A, int lda, long a_plane, const uint16_t* B,
            int ldb, long b_plane, float* C, int ldc, const float* bias, const float* gate, int epilogue, int tile,
            hipStream_t s, float* colsum, const X3Out* out = nullptr, const X3Extra* ex = nullptr, int split_k = 1, void* ws = nullptr,
            size_t ws_bytes = 0, const uint16_t* gate16 = nullptr, int gate16_r8 = 0, void* red_cnt = nullptr, int red_cnt_words = 0);
int split_planes(const float* src, int R, int C, int ld, int np, uint16_t* rc, int ld_rc, long rc_plane, uint16_t* r8,
                 long r8_plane, hipStream_t s);
int split_planes_multi(SplitJobs jobs, hipStream_t s);

// process-wide GEMM plan (tile, split_k) per call site; 0 = automatic.  The only global state of the library:
// a tuning knob (dpd_set_gemm_plan), never needed for correctness.
enum { OP_FWD_L1 = 0, OP_FWD_L23 = 1, OP_BWD_DH = 2, OP_BWD_DX = 3, OP_BWD_DW1 = 4, OP_BWD_DW23 = 5, OP_BWD_DH_T = 6, OP_BWD_DX_T = 7,
       OP_COUNT = 8 };   // _T: the same product with a transposed weight copy (NN form)
// Defaults measured on MI355X at B=32 (tools/gemm_bench.py, profiles/): LDS-DMA ring kernels everywhere;
//   fwd L1 (4096x1024x2528)  128x128 16-wave 3-stage ring  ~127 TFLOP/s    fwd L2/3 (K=1024) 128x128 3-stage  ~120
//   bwd dH (2048x1024x1024)   64x64  3-stage               ~ 93..106       bwd dX            64x64 3-stage    ~103
//   bwd dW1 (2528x1024x2048)  64x64  3-stage, split-K 2    ~102            bwd dW2/3         64x64 3-stage    ~ 97
// (run-to-run spread between boxes is ~10 %; the ranking inside one run is stable.  DMA kernels need K % 32 == 0;
//  gemm_f32 falls back to the register-staged 64x64 kernel otherwise.)
static int g_plan_tile[OP_COUNT] = {32, 32, 8, 8, 33, 33, 32, 32};
static int g_plan_split[OP_COUNT] = {1, 1, 1, 1, 1, 1, 1, 1};   // (0 = tail split, gemm_rs.h: measured no gain on dW1, 0.525 vs 0.5215 ms of GEMM per step)
static int g_x3_tile[OP_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};   // one-plane (bf16) tile override per call site, 0 = automatic
// split-K of the plane weight-gradient GEMMs (K = query rows is long, M x N gives 64-160 tiles of 128x128 for 256 CUs):
//   n > 1: n slices per output tile, reduced INSIDE the launch by the last-arriving slice in slice order (gemm_x3.hip: inlaunch_reduce;
//          deterministic); n < -1: the round-2 form, |n| fp32 slabs + a reduce launch (A/B reference); 1: off; 0: automatic
static int g_x3_split[OP_COUNT] = {1, 1, 1, 1, 0, 0, 1, 1};
static int g_x3_pair_tile = 0, g_x3_pair_split = 0;   // plane dW2+dW3 pair: tile (0 = automatic), split-K (as above)
static int g_x3_trio_tile = 0, g_x3_trio_split = 1;   // plane dW1+dW2+dW3 in one launch (dpd_decoder_bwd_weights_trio): tile, split-K
constexpr size_t kRedCntBytes = 8192;                 // arrival words of the in-launch reduction: the last 8 KiB of the base workspace

// Compute type of the three wide layers (the `dtype` argument of the decoder entry points):
//   0  exact fp32 on the fp32 MFMA (gemm_f32.hip)
//   1  fp32-equivalent on the bf16 MFMA: operands split into 3 bf16 planes, 6 MFMA terms (gemm_x3.hip)
//   2  bf16 operands, fp32 accumulation and fp32 outputs (mixed-precision training, BASELINE config 3)
// For 1 and 2 each GEMM first writes the planes of its two operands into `scr` (unfused producers).
struct Scratch {
    char* p;
    size_t bytes;
};
static size_t plane_bytes(int dtype, int Q, int KP, int H) {
    if (dtype == 0) return 0;
    const size_t np = dtype == 1 ? 3 : 1;
    const size_t big = (size_t)(KP > H ? KP : H);
    return np * 2 * ((size_t)Q * big + big * H) + 256;
}

// Automatic split-K of a plane weight-gradient GEMM: enough slices that the launch has about two 128x128 workgroups per CU (their
// 64 KiB LDS rings let two co-reside: one wave of tiles alone cannot keep the LDS-DMA ring of a CU full), each with >= 1024 deep K.
static int x3_auto_split(int np, int tile, int M, int N, int K) {
    if (np != 1 || tile < 1 || tile > 5) return 1;
    const int bm = (tile == 3 || tile == 5) ? 64 : 128, bn = (tile == 4 || tile == 5) ? 64 : 128;
    const long tiles = (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    int split = (int)(480 / (tiles > 0 ? tiles : 1));
    while (split > 1 && K / split < 1024) --split;
    return split < 1 ? 1 : (split > 4 ? 4 : split);
}

// In-launch split-K parks the raw accumulators of every slice in TILE-PADDED slabs (gemm_x3.hip: tiles * split * bm * bn floats): the largest
// split <= `split` whose slabs fit `slab_bytes` for problems of rows[0 .. nprob) x N (with small H the base workspace holds
// fewer slabs than the automatic split asked for, and the call failed with DPD_E_WORKSPACE instead of running unsplit)
static int x3_fit_split(int tile, int split, const int* rows, int nprob, int N, size_t slab_bytes) {
    if (split <= 1 || tile < 1 || tile > 5) return split;
    const int bm = (tile == 3 || tile == 5) ? 64 : 128, bn = (tile == 4 || tile == 5) ? 64 : 128;
    size_t tiles = 0;
    for (int i = 0; i < nprob; ++i) tiles += (size_t)((rows[i] + bm - 1) / bm) * ((N + bn - 1) / bn);
    while (split > 1 && (tiles * split * bm * bn * sizeof(float) > slab_bytes || tiles > kRedCntBytes / 8)) --split;
    return split;
}

// Apl / Bpl: operand planes that already exist (else the fp32 operand is split into `scr`); out: plane outputs.
static int gemm_dt(int dtype, int op, int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B,
                   int ldb, float* C, int ldc, const float* bias, const float* gate, int epilogue, void* ws, size_t ws_bytes,
                   Scratch scr, hipStream_t s, float* colsum = nullptr, const void* Apl = nullptr, const void* Bpl = nullptr,
                   const X3Out* out = nullptr, const ColsumTwoStep* cs2 = nullptr, const void* gate16 = nullptr, int gate16_r8 = 0) {
    const int ra = transA ? K : M, ca = transA ? M : K;   // stored shape of A, B
    const int rb = transB ? N : K, cb = transB ? K : N;
    const bool planes_ok = dtype != 0 && !(K % 32) && !(ra & 7) && !(ca & 7) && !(rb & 7) && !(cb & 7) && !(transA && transB);
    if (!planes_ok) {   // exact fp32 (also for shapes the plane kernels do not take)
        if (Apl || Bpl || out || (gate16 && !gate)) return DPD_E_UNSUPPORTED;   // callers only pass planes for shapes planes_shape_ok() accepts
        const int split = (dtype == 0) ? g_plan_split[op] : 1;
        int tile = g_plan_tile[op];
        // the 128x128 one-workgroup-per-CU kernels need >= ~200 tiles to fill the chip (B = 32 forward); smaller batches
        // (as-loss mode at the PCRNet batch of 16: M = 2048 -> 128 tiles) run the 64x64 kernel, 3 workgroups per CU
        const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128);
        if (tile == 9 && tiles128 < 200) tile = 8;
        // register-streamed NN products: once there are two 128x128 tiles per CU, 64x64 wave tiles (half the operand loads per
        // flop) keep two waves per SIMD as well -- measured at B = 64: family 0.75 -> 0.85 of peak, step 1.206 -> 1.081 ms
        if (tile == 32 && tiles128 >= 512 && !transA && !transB) tile = 30;
        return gemm_f32(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, gate, epilogue, colsum ? 1 : split, tile, ws, ws_bytes,
                        s, colsum, nullptr, nullptr, nullptr, cs2);
    }
    if (cs2) return DPD_E_UNSUPPORTED;
    const int np = dtype == 1 ? 3 : 1;
    const size_t ae = (size_t)M * K, be = (size_t)K * N;
    const size_t need = (size_t)np * 2 * ((Apl ? 0 : ae) + (Bpl ? 0 : be));
    if (need && (!scr.p || scr.bytes < need)) return DPD_E_WORKSPACE;
    const uint16_t* Ap = (const uint16_t*)Apl;
    const uint16_t* Bp = (const uint16_t*)Bpl;
    uint16_t* cursor = (uint16_t*)scr.p;
    if (!Ap) {
        if (!A) return DPD_E_NULL;
        if (int rc = split_planes(A, ra, ca, lda, np, transA ? nullptr : cursor, ca, (long)ae, transA ? cursor : nullptr, (long)ae, s))
            return rc;
        Ap = cursor;
        cursor += (size_t)np * ae;
    }
    if (!Bp) {
        if (!B) return DPD_E_NULL;
        if (int rc = split_planes(B, rb, cb, ldb, np, transB ? cursor : nullptr, cb, (long)be, transB ? nullptr : cursor, (long)be, s))
            return rc;
        Bp = cursor;
    }
    // largest tile that still gives the 256 CUs at least ~200 workgroups
    auto blocks = [&](int bm, int bn) { return (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    // (measured, tools/x3_bench.py: 160 blocks of 128x128 beat 320 of 64x128 on the 2528x1024 dW1; 64x64 only when even
    //  64x128 leaves most CUs idle)
    int tile = blocks(128, 128) >= 150 ? 2 : (blocks(64, 128) >= 100 ? 3 : 5);
    // one plane (bf16) and enough rows for one 256x128 workgroup per CU (B = 64: 8192 x 1024 -> 256 tiles): the phase-staggered
    // BK = 64 kernel (gemm_p8_kernel, tile 21).  Measured on the forward shapes, profiles/r03_x3_bench_p8.txt: layer 1 53.2 -> 48.8 us,
    // layers 2/3 28.1 -> 27.0 us; the backward shapes (M = 4096: 128 such tiles) stay on the 128x128 ring kernel
    if (np == 1 && !transA && blocks(256, 128) >= 224) tile = 21;
    // three planes: the same schedule at BK = 32 on 128x128 tiles (tile 24) wherever the ring kernel ran 128x128 tiles: layer 1 at
    // B = 32 132 -> 118 us, layers 2/3 52.6 -> 50.0 us, dW1 88 -> 79 us (tools/p8_probe.py, NP=3)
    if (np == 3 && tile == 2) tile = 24;
    if (np == 1 && g_x3_tile[op] && (g_x3_tile[op] >= 20 || !(K % 64))) tile = g_x3_tile[op];   // tuning override (dpd_set_gemm_plan, ops 16..24)
    if (np == 3 && g_x3_tile[op] >= 24 && g_x3_tile[op] <= 26) tile = g_x3_tile[op];
    // plane weight gradients (TN, plain product): split-K with the reduction inside the launch when the tiles alone leave CUs idle
    // (`ws` = the slab region of the base workspace, its last kRedCntBytes the arrival words)
    int split = 1;
    void* cnt = nullptr;
    if (transA && !transB && epilogue == 0 && !colsum && !out && C && ws && ws_bytes > kRedCntBytes) {
        split = g_x3_split[op];
        if (split == 0) split = x3_auto_split(np, tile, M, N, K);
        if (split > 1 && tile >= 1 && tile <= 5) split = x3_fit_split(tile, split, &M, 1, N, ws_bytes - kRedCntBytes);
        if (split > 1 && tile >= 1 && tile <= 5) cnt = (char*)ws + ws_bytes - kRedCntBytes;
        else if (split < -1) split = -split;
        else split = 1;
        const int chunk = (((K + split - 1) / split) + 63) / 64 * 64;
        if (split > 1 && chunk * (split - 1) >= K) { split = 1; cnt = nullptr; }
    }
    return gemm_x3(np, transA, !transB, M, N, K, Ap, ca, (long)ae, Bp, cb, (long)be, C, ldc, bias, gate, epilogue, tile, s, colsum,
                   out, nullptr, split, split > 1 ? ws : nullptr, split > 1 ? ws_bytes - kRedCntBytes : 0,
                   (const uint16_t*)gate16, gate16_r8, cnt, (int)(kRedCntBytes / 8));
}

// `pl` is honoured only for these shapes (everything the fused producers and the plane GEMMs assume)
static const dpd_planes* usable_planes(const dpd_planes* pl, int dtype, int Q, int Qb, int KP, int H) {
    if (!pl || dtype == 0) return nullptr;
    if ((Q & 7) || (Qb & 31) || (KP & 31) || (H & 63)) return nullptr;
    return pl;
}
static int check_planes(const dpd_planes* pl, int dtype) {
    if (!pl) return 0;
    if (pl->np != (dtype == 1 ? 3 : 1)) return DPD_E_DIM;
    return 0;
}
static X3Out make_out(const dpd_planes* pl, void* rc, int rc_rows, void* r8, int r8_rows, int cols) {
    X3Out o;
    o.rc = (uint16_t*)rc; o.r8 = (uint16_t*)r8; o.np = pl->np; o.ld_rc = cols; o.r8_rows = r8_rows;
    o.rc_plane = (long)rc_rows * cols; o.r8_plane = (long)r8_rows * cols;
    return o;
}

// ---- output layer: y = h3 W4 + b4 ; pred = clip(y,0,6)/3 * mask.  One wave per row. ----------------------
// the three dot products of one row with W4 [H,3] (one wave; every lane returns the full sums): ONE definition, so that every
// kernel that evaluates the output layer produces the same bits
// (H % 256 == 0, H <= 1024, W4 16-byte aligned: 16-byte loads of the row AND of W4 -- the 12 floats W4[k..k+3][0..2] are three float4 --
// and the loaded row / column 0 of W4 stay in registers for a caller that goes on to the backward; same additions in the same order)
constexpr int kOutMaxI = 4;
struct OutRowRegs {
    float4 h[kOutMaxI];
    float w0[kOutMaxI][4];
};
__device__ __forceinline__ bool out_row_fast(const float* W4, int H) { return (H & 255) == 0 && H <= 256 * kOutMaxI && !((uintptr_t)W4 & 15); }

__device__ __forceinline__ void out_row_dot(const float* __restrict__ h, const float* __restrict__ W4, int H, int lane, float (&a)[3],
                                            OutRowRegs* keep = nullptr) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (out_row_fast(W4, H)) {
#pragma unroll
        for (int i = 0; i < kOutMaxI; ++i) {
            const int k = 4 * lane + 256 * i;
            if (k < H) {
                const float4 x = *reinterpret_cast<const float4*>(h + k);
                const float4* wp = reinterpret_cast<const float4*>(W4 + (size_t)k * 3);
                const float4 wa = wp[0], wb = wp[1], wc = wp[2];
                a0 += x.x * wa.x; a1 += x.x * wa.y; a2 += x.x * wa.z;
                a0 += x.y * wa.w; a1 += x.y * wb.x; a2 += x.y * wb.y;
                a0 += x.z * wb.z; a1 += x.z * wb.w; a2 += x.z * wc.x;
                a0 += x.w * wc.y; a1 += x.w * wc.z; a2 += x.w * wc.w;
                if (keep) { keep->h[i] = x; keep->w0[i][0] = wa.x; keep->w0[i][1] = wa.w; keep->w0[i][2] = wb.z; keep->w0[i][3] = wc.y; }
            }
        }
    } else if ((H & 255) == 0) {          // 16 bytes per lane per load: H / 256 loads in flight instead of H / 64 dependent-address ones
        for (int k = 4 * lane; k < H; k += 256) {
            const float4 x = *reinterpret_cast<const float4*>(h + k);
            const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0 += xs[e] * W4[(k + e) * 3 + 0];
                a1 += xs[e] * W4[(k + e) * 3 + 1];
                a2 += xs[e] * W4[(k + e) * 3 + 2];
            }
        }
    } else {
        for (int k = lane; k < H; k += 64) {
            const float x = h[k];
            a0 += x * W4[k * 3 + 0];
            a1 += x * W4[k * 3 + 1];
            a2 += x * W4[k * 3 + 2];
        }
    }
    a[0] = wave_sum(a0); a[1] = wave_sum(a1); a[2] = wave_sum(a2);
}

__global__ __launch_bounds__(256) void out_fwd_kernel(const float* __restrict__ h3, const float* __restrict__ W4,
                                                       const float* __restrict__ b4, const float* __restrict__ mask,
                                                       float* __restrict__ y, float* __restrict__ pred, int Q, int H) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Q) return;
    float a[3];
    out_row_dot(h3 + (size_t)row * H, W4, H, lane, a);
    if (lane < 3) {
        const float v = (lane == 0 ? a[0] : (lane == 1 ? a[1] : a[2])) + b4[lane];
        y[(size_t)row * 3 + lane] = v;
        pred[(size_t)row * 3 + lane] = fminf(fmaxf(v, 0.f), 6.f) / 3.0f * mask[row];   // relu6(y)/3 (:691) * mask (:697)
    }
}

// DPDist as a frozen loss (pcrnet-registration/iterative_PCRNet_ours.py:229-257): output layer, loss_pred and -- when g3 != NULL -- the
// output-layer backward of d loss_pred / d pred (= gv = 0.5 / BN on channel 0 of every row, models/dpdist_and_aue.py:976-977) in ONE
// launch instead of three (out_fwd + l1_loss + out_bwd: ~5 us each at the PCRNet batch, all launch-latency bound).  One wave per row.
// Same values as the three-kernel chain (loss_pred within an ulp: exact fixed-point sum instead of fp32 partial sums): y / pred through
// out_row_dot, dy = (gv * mask / 3 * [0 < y0 < 6], 0, 0), g3 = dy0 * W4[:,0] * [h3 > 0] (the chain adds two exact zeros to that).
template <int RW>      // rows per wave
__global__ __launch_bounds__(256) void out_asloss_kernel(const float* __restrict__ h3, const float* __restrict__ W4,
                                                          const float* __restrict__ b4, const float* __restrict__ mask,
                                                          float* __restrict__ y, float* __restrict__ pred, float* __restrict__ dy,
                                                          float* __restrict__ g3, int Q, int H, int BN, float gv,
                                                          float* __restrict__ loss, unsigned long long* __restrict__ acc,
                                                          uint16_t* __restrict__ g3_rc, long g3_plane, int g3_np) {
    // g3_rc (round 5, the as-loss engine of the plane compute types): g3 leaves this kernel as the RC operand plane(s) of the first dH GEMM
    // -- the same split as split_planes_kernel, which this saves (one launch and the fp32 round trip of g3; g3 itself may then be NULL)
    __shared__ float s_p[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float p0 = 0.f;
#pragma unroll 1
    for (int rw = 0; rw < RW; ++rw) {
        const int row = (blockIdx.x * 4 + wave) * RW + rw;
        if (row >= Q) break;
        const float* h = h3 + (size_t)row * H;
        float a[3];
        OutRowRegs rr;
        out_row_dot(h, W4, H, lane, a, &rr);
        const float mk = mask[row];
        float yv[3], pv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            yv[c] = a[c] + b4[c];
            pv[c] = fminf(fmaxf(yv[c], 0.f), 6.f) / 3.0f * mk;
        }
        if (lane < 3) {
            y[(size_t)row * 3 + lane] = lane == 0 ? yv[0] : (lane == 1 ? yv[1] : yv[2]);
            pred[(size_t)row * 3 + lane] = lane == 0 ? pv[0] : (lane == 1 ? pv[1] : pv[2]);
        }
        p0 += pv[0];
        if (dy) {
            const float d0 = (yv[0] > 0.f && yv[0] < 6.f) ? gv * mk / 3.0f : 0.f;
            if (lane < 3) dy[(size_t)row * 3 + lane] = lane == 0 ? d0 : 0.f;
            float* g = g3 ? g3 + (size_t)row * H : nullptr;
            if (out_row_fast(W4, H)) {
#pragma unroll
                for (int i = 0; i < kOutMaxI; ++i) {
                    const int k = 4 * lane + 256 * i;
                    if (k < H) {
                        const float4 x = rr.h[i];
                        float4 o;
                        o.x = x.x > 0.f ? d0 * rr.w0[i][0] : 0.f;
                        o.y = x.y > 0.f ? d0 * rr.w0[i][1] : 0.f;
                        o.z = x.z > 0.f ? d0 * rr.w0[i][2] : 0.f;
                        o.w = x.w > 0.f ? d0 * rr.w0[i][3] : 0.f;
                        if (g) *reinterpret_cast<float4*>(g + k) = o;
                        if (g3_rc) {
                            const float ov[4] = {o.x, o.y, o.z, o.w};
                            unsigned pl4[4][3];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if (g3_np == 1) { pl4[e][0] = bf16_bits(ov[e]); pl4[e][1] = 0; pl4[e][2] = 0; }
                                else split3(ov[e], pl4[e]);
                            }
#pragma unroll
                            for (int q = 0; q < 3; ++q)
                                if (q < g3_np)
                                    *reinterpret_cast<uint2*>(g3_rc + q * g3_plane + (size_t)row * H + k) =
                                        make_uint2(pl4[0][q] | (pl4[1][q] << 16), pl4[2][q] | (pl4[3][q] << 16));
                        }
                    }
                }
            } else if (g) {
                for (int k = lane; k < H; k += 64) g[k] = h[k] > 0.f ? d0 * W4[k * 3] : 0.f;
            }
        }
    }
    // loss_pred = sum over ALL rows of pred[:,0] / (2 BN).  The sum is taken in 2^-32 fixed point: integer addition is associative, so
    // the blocks can add their part in any order with ONE relaxed device-scope atomic each and still produce the same bits on every run;
    // the block count rides in the top 16 bits of the same word, so the block that completes it needs no fence and no second look at
    // memory (a __threadfence() per block costs an L2 write-back on this chip: this kernel took 23 us with a ticket + fence, ~6 without)
    if (lane == 0) s_p[wave] = p0;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long part = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) part += (unsigned long long)__double2ll_rn((double)s_p[w] * 4294967296.0);   // pred >= 0
        const unsigned long long add = part + (1ull << 48);
        const unsigned long long old = atomicAdd(acc, add);
        if ((old >> 48) + 1 == (unsigned long long)gridDim.x) {
            const unsigned long long total = (old + add) & ((1ull << 48) - 1);
            loss[0] = (float)((double)total * (1.0 / 4294967296.0) / (2.0 * (double)BN));      // (:976-977)
            atomicExch(acc, 0ull);                                                               // ready for the next launch
        }
    }
}

// dy = dpred * mask * [0 < y < 6] / 3 ;  g3 = (dy W4^T) * [h3 > 0].  One wave per row.
struct ZeroList {   // small accumulators cleared by the first kernel of the backward chain (saves memset launches)
    float* p[5];
    int n[5];
};

__global__ __launch_bounds__(256) void out_bwd_kernel(const float* __restrict__ dpred, const float* __restrict__ mask,
                                                       const float* __restrict__ y, const float* __restrict__ h3,
                                                       const float* __restrict__ W4, float* __restrict__ dy,
                                                       float* __restrict__ g3, int Qb, int H, ZeroList zl) {
    if (blockIdx.x < 5 && zl.p[blockIdx.x]) {
        for (int i = threadIdx.x; i < zl.n[blockIdx.x]; i += 256) zl.p[blockIdx.x][i] = 0.f;
    }
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Qb) return;
    float d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float yv = y[(size_t)row * 3 + c];
        d[c] = (yv > 0.f && yv < 6.f) ? dpred[(size_t)row * 3 + c] * mask[row] / 3.0f : 0.f;   // relu6 gradient is 1 on (0,6)
    }
    if (lane < 3) dy[(size_t)row * 3 + lane] = d[lane];
    const float* h = h3 + (size_t)row * H;
    float* g = g3 + (size_t)row * H;
    for (int k = lane; k < H; k += 64) {
        const float v = d[0] * W4[k * 3] + d[1] * W4[k * 3 + 1] + d[2] * W4[k * 3 + 2];
        g[k] = (h[k] > 0.f) ? v : 0.f;
    }
}

// Fused output-layer backward: ONE pass over h3 produces dy, g3 AND the per-block partial sums of
//   db3 = colsum(g3), dW4 = h3^T dy, db4 = colsum(dy)
// (block = 32 rows, 4 waves x 8 rows; lane owns columns lane+64j).  Partials go to `scratch` [nblk][4H+4] and are
// summed in fixed order by small_grads_reduce -> deterministic, no atomics, no memset.  H <= 1024.
constexpr int kOBRows = 8;
constexpr int kOBMaxJ = 16;

__global__ __launch_bounds__(256) void out_bwd_fused_kernel(const float* __restrict__ dpred, const float* __restrict__ mask,
                                                             const float* __restrict__ y, const float* __restrict__ h3,
                                                             const float* __restrict__ W4, float* __restrict__ dy,
                                                             float* __restrict__ g3, int Qb, int H, ZeroList zl,
                                                             float* __restrict__ scratch) {
    extern __shared__ float s_acc[];   // [4H + 4]
    if (blockIdx.x < 5 && zl.p[blockIdx.x]) {
        for (int i = threadIdx.x; i < zl.n[blockIdx.x]; i += 256) zl.p[blockIdx.x][i] = 0.f;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nj = H / 64;
    float w4[kOBMaxJ][3], s3[kOBMaxJ], a4[kOBMaxJ][3];
#pragma unroll
    for (int j = 0; j < kOBMaxJ; ++j) {
        s3[j] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) { a4[j][c] = 0.f; w4[j][c] = (j < nj) ? W4[(lane + 64 * j) * 3 + c] : 0.f; }
    }
    float dsum[3] = {0.f, 0.f, 0.f};
    for (int rr = 0; rr < kOBRows / 4; ++rr) {
        const int row = blockIdx.x * kOBRows + wave * (kOBRows / 4) + rr;
        if (row >= Qb) break;
        float d[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float yv = y[(size_t)row * 3 + c];
            d[c] = (yv > 0.f && yv < 6.f) ? dpred[(size_t)row * 3 + c] * mask[row] / 3.0f : 0.f;   // relu6 gradient is 1 on (0,6)
            dsum[c] += d[c];
        }
        if (lane < 3) dy[(size_t)row * 3 + lane] = d[lane];
        const float* h = h3 + (size_t)row * H;
        float* g = g3 + (size_t)row * H;
#pragma unroll
        for (int j = 0; j < kOBMaxJ; ++j) {
            if (j < nj) {
                const int k = lane + 64 * j;
                const float hv = h[k];
                const float v = d[0] * w4[j][0] + d[1] * w4[j][1] + d[2] * w4[j][2];
                const float gv = (hv > 0.f) ? v : 0.f;
                g[k] = gv;
                s3[j] += gv;
#pragma unroll
                for (int c = 0; c < 3; ++c) a4[j][c] += hv * d[c];
            }
        }
    }
    // fixed-order reduction over the 4 waves
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int j = 0; j < kOBMaxJ; ++j) {
                if (j < nj) {
                    const int k = lane + 64 * j;
                    if (w == 0) {
                        s_acc[k] = s3[j];
#pragma unroll
                        for (int c = 0; c < 3; ++c) s_acc[H + k * 3 + c] = a4[j][c];
                    } else {
                        s_acc[k] += s3[j];
#pragma unroll
                        for (int c = 0; c < 3; ++c) s_acc[H + k * 3 + c] += a4[j][c];
                    }
                }
            }
            if (lane < 3) {
                const float v = (lane == 0) ? dsum[0] : ((lane == 1) ? dsum[1] : dsum[2]);
                if (w == 0) s_acc[4 * H + lane] = v;
                else s_acc[4 * H + lane] += v;
            }
        }
        __syncthreads();
    }
    float* out = scratch + (size_t)blockIdx.x * (4 * H + 4);
    for (int i = threadIdx.x; i < 4 * H + 3; i += 256) out[i] = s_acc[i];
}

// Same contract as out_bwd_fused_kernel for H % 256 == 0, H <= 1024: a lane owns 4 consecutive columns in each
// 256-column group (float4 loads/stores, both rows of a wave in flight together) and the four waves' partials are
// combined with ONE barrier (each wave has its own LDS slab).  LDS: 4 * (4H + 4) floats.
// L1: optional fused loss (training mode, utils/dpdist_util.py:962-980): with l1.labels the kernel derives
// d loss_samples / d pred_AB = sign(pred_AB[:,0] - labels) * gscale / Qb itself (dpred is not read) and adds the three loss sums
// (sum |pred_AB - labels|, sum pred_AB, sum pred_BA over its rows) to the block record [4H+4 .. 4H+6].
struct L1Fuse {
    const float* pred;     // [2*Qb, 3]
    const float* labels;   // [Qb]
    float gscale;
};
constexpr int kOBRec = 8;   // record = 4H + 8 floats: db3 [H] | dW4 [3H] | db4 [3] | pad | loss sums [3] | pad
// Optional bf16 operand planes of g3 written by the same pass (plane compute types: no fp32 g3, no conversion launch): RC
// [np][Qb][H] straight from the registers, R8 [np][Qb/8][H][8] through an LDS image of the block's 8 rows (behind the four slabs).
struct G3Planes {
    uint16_t* rc;
    uint16_t* r8;
    long plane;     // elements per plane (Qb * H)
    int np;
};
// Optional forward of the output layer inside the same pass (training step: no out_fwd launch, h3 rows of the AB half are read
// once): with y != NULL the kernel computes y = h3 W4 + b4 and pred = relu6(y)/3 * mask for its AB rows AND their BA twins
// (row + Qb), in out_fwd_kernel's summation order (bit-identical), writes both, and uses them instead of the y / l1.pred inputs.
struct OutFwd {
    const float* b4;
    float* y;       // [2*Qb, 3]
    float* pred;    // [2*Qb, 3]
};

#ifdef DPD_ABLATIONS
__device__ unsigned long long g_ob_stamps[1024 * 8];       // s_memtime milestones of thread 0 of every workgroup (tools/ob_stamps.py)
#define OB_STAMP(i) do { if (threadIdx.x == 0) g_ob_stamps[(blockIdx.x & 1023) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define OB_STAMP(i) do { } while (0)
#endif
__global__ __launch_bounds__(256) void out_bwd_fused4_kernel(const float* __restrict__ dpred, const float* __restrict__ mask,
                                                              const float* __restrict__ y, const float* __restrict__ h3,
                                                              const float* __restrict__ W4, float* __restrict__ dy,
                                                              float* __restrict__ g3, int Qb, int H, ZeroList zl,
                                                              float* __restrict__ scratch, L1Fuse l1, OutFwd of, G3Planes gp,
                                                              const uint16_t* __restrict__ h3b) {
    extern __shared__ float s_acc[];   // [4 waves][4H + 8]; reused at the end (gp.r8) as uint16 [np][8][H]
    if (blockIdx.x < 5 && zl.p[blockIdx.x]) {
        for (int i = threadIdx.x; i < zl.n[blockIdx.x]; i += 256) zl.p[blockIdx.x][i] = 0.f;
    }
    OB_STAMP(0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ng = H / 256;            // column groups (<= 4)
    const int P = 4 * H + kOBRec;
    constexpr int RW = kOBRows / 4;    // rows per wave
    float lsum[3] = {0.f, 0.f, 0.f};   // loss sums of this wave's rows (identical in every lane)
    float dsum[3] = {0.f, 0.f, 0.f};
    float d[RW][3];
    float4 hv[RW][4], hb[RW][4];
    const int row0 = blockIdx.x * kOBRows + wave * RW;
    // Every global load of the kernel is requested HERE, before anything waits: W4 above, the rows (and their BA twins) and the per-row
    // scalars below.  Left where they were used, they made three dependent round trips (W4 -> row 0 -> row 1: 5.5k + 3.9k + 5.7k cycles
    // of a 25k-cycle kernel at B = 32, tools/ob_stamps.py); with one workgroup per CU there is nothing else to hide them behind.
    static_assert(RW == 2, "the lane-pair exchanges below assume two rows per wave");
    const bool odd = lane & 1;
    // exchange a 64-bit value with lane ^ 1 (DPP quad_perm [1,0,3,2])
    auto swap_pair = [](uint2 v) {
        return make_uint2((unsigned)__builtin_amdgcn_mov_dpp((int)v.x, 0xB1, 0xf, 0xf, true), (unsigned)__builtin_amdgcn_mov_dpp((int)v.y, 0xB1, 0xf, 0xf, true));
    };
    auto widen = [](uint2 u) {
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    };
    if (h3b) {
        // layer 3's activation as one bf16 plane (DPD_BF16), widened exactly.  A lane's four columns of one row are 8 bytes, and 8-byte
        // accesses run at ~0.6 of the 16-byte rate (round 5: this load phase was 12k of the kernel's 37k cycles at B = 64): the lanes
        // 2p / 2p + 1 therefore load 16 bytes -- the eight columns of BOTH lanes -- of row 0 / row 1 and hand each other the half that
        // belongs to the partner.  Same values in the same registers as two 8-byte loads per lane.
        const int rsel = min(row0 + (odd ? 1 : 0), Qb - 1);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            if (jj < ng) {
                const uint4 va = *reinterpret_cast<const uint4*>(h3b + (size_t)rsel * H + 256 * jj + 8 * (lane >> 1));
                const uint2 ra = swap_pair(odd ? make_uint2(va.x, va.y) : make_uint2(va.z, va.w));
                hv[0][jj] = widen(odd ? ra : make_uint2(va.x, va.y));
                hv[1][jj] = widen(odd ? make_uint2(va.z, va.w) : ra);
                if (of.y) {
                    const uint4 vb = *reinterpret_cast<const uint4*>(h3b + ((size_t)Qb + rsel) * H + 256 * jj + 8 * (lane >> 1));
                    const uint2 rb = swap_pair(odd ? make_uint2(vb.x, vb.y) : make_uint2(vb.z, vb.w));
                    hb[0][jj] = widen(odd ? rb : make_uint2(vb.x, vb.y));
                    hb[1][jj] = widen(odd ? make_uint2(vb.z, vb.w) : rb);
                }
            }
        }
    } else {
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int row = min(row0 + rr, Qb - 1);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (jj < ng) {
                    hv[rr][jj] = *reinterpret_cast<const float4*>(h3 + (size_t)row * H + 256 * jj + 4 * lane);
                    if (of.y) hb[rr][jj] = *reinterpret_cast<const float4*>(h3 + ((size_t)Qb + row) * H + 256 * jj + 4 * lane);
                }
            }
        }
    }
    float w4[4][4][3], s3[4][4], a4[4][4][3];
    const bool w4_vec = !((uintptr_t)W4 & 15);       // the 12 floats W4[k..k+3][0..2] of a lane's four columns are three float4 (k % 4 == 0)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        if (w4_vec && jj < ng) {      // 12 loads per lane instead of 48: the address unit was ~6k of this kernel's 24k cycles at B = 32
            const float4* wp = reinterpret_cast<const float4*>(W4 + (size_t)(256 * jj + 4 * lane) * 3);
            const float4 wa = wp[0], wb = wp[1], wc = wp[2];
            const float w[12] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w, wc.x, wc.y, wc.z, wc.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < 3; ++c) w4[jj][e][c] = w[3 * e + c];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s3[jj][e] = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                a4[jj][e][c] = 0.f;
                if (!(w4_vec && jj < ng)) w4[jj][e][c] = (jj < ng) ? W4[(256 * jj + 4 * lane + e) * 3 + c] : 0.f;
            }
        }
    }
    // the per-row scalars are requested before the row loop: behind the dot products and wave sums they were one more dependent round trip
    float mk_a[RW], mk_b[RW], lab[RW];
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
        const int row = min(row0 + rr, Qb - 1);
        mk_a[rr] = mask[row];
        mk_b[rr] = of.y ? mask[Qb + row] : 0.f;
        lab[rr] = l1.labels ? l1.labels[row] : 0.f;
    }
    OB_STAMP(1);
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
        const int row = min(row0 + rr, Qb - 1);
        const bool live = row0 + rr < Qb;
        float yab[3] = {0.f, 0.f, 0.f}, pab0 = 0.f, pba0 = 0.f;
        if (of.y) {                // forward of the output layer for this AB row and its BA twin (out_fwd_kernel's order)
            float a[3] = {0.f, 0.f, 0.f}, b[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (jj < ng) {
                    const float xa[4] = {hv[rr][jj].x, hv[rr][jj].y, hv[rr][jj].z, hv[rr][jj].w};
                    const float xb[4] = {hb[rr][jj].x, hb[rr][jj].y, hb[rr][jj].z, hb[rr][jj].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int c = 0; c < 3; ++c) { a[c] += xa[e] * w4[jj][e][c]; b[c] += xb[e] * w4[jj][e][c]; }
                }
            }
            if (rr == 0) { asm volatile("" :: "v"(a[0]), "v"(b[2])); OB_STAMP(2); }
            float yba[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { yab[c] = wave_sum(a[c]) + of.b4[c]; yba[c] = wave_sum(b[c]) + of.b4[c]; }
            const float ma = mk_a[rr], mb = mk_b[rr];
            float pa[3], pb[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                pa[c] = fminf(fmaxf(yab[c], 0.f), 6.f) / 3.0f * ma;   // relu6(y)/3 (:691) * mask (:697)
                pb[c] = fminf(fmaxf(yba[c], 0.f), 6.f) / 3.0f * mb;
            }
            pab0 = pa[0]; pba0 = pb[0];
            if (live && lane < 3) {
                const float ya = lane == 0 ? yab[0] : (lane == 1 ? yab[1] : yab[2]), yb = lane == 0 ? yba[0] : (lane == 1 ? yba[1] : yba[2]);
                const float qa = lane == 0 ? pa[0] : (lane == 1 ? pa[1] : pa[2]), qb = lane == 0 ? pb[0] : (lane == 1 ? pb[1] : pb[2]);
                of.y[(size_t)row * 3 + lane] = ya; of.y[((size_t)Qb + row) * 3 + lane] = yb;
                of.pred[(size_t)row * 3 + lane] = qa; of.pred[((size_t)Qb + row) * 3 + lane] = qb;
            }
            if (rr == 0) OB_STAMP(3);
        }
        float dp[3];
        if (l1.labels) {           // d mean|pred_AB[:,0] - labels| / d pred_AB (tf.abs gradient = sign), channels 1, 2 get none
            const float pab = of.y ? pab0 : l1.pred[(size_t)row * 3], pba = of.y ? pba0 : l1.pred[((size_t)Qb + row) * 3];
            const float df = pab - lab[rr];
            dp[0] = ((df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f)) * (1.0f / (float)Qb) * l1.gscale;
            dp[1] = 0.f; dp[2] = 0.f;
            if (live) { lsum[0] += fabsf(df); lsum[1] += pab; lsum[2] += pba; }
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) dp[c] = dpred[(size_t)row * 3 + c];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float yv = of.y ? yab[c] : y[(size_t)row * 3 + c];
            d[rr][c] = (live && yv > 0.f && yv < 6.f) ? dp[c] * mk_a[rr] / 3.0f : 0.f;   // relu6' = 1 on (0,6)
            dsum[c] += d[rr][c];
        }
    }
    OB_STAMP(4);
    uint2 wpk[RW][4][3] = {};
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
        const int row = row0 + rr;
        if (row >= Qb) break;
        if (lane < 3) dy[(size_t)row * 3 + lane] = d[rr][lane];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            if (jj < ng) {
                const float hh[4] = {hv[rr][jj].x, hv[rr][jj].y, hv[rr][jj].z, hv[rr][jj].w};
                float gg[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = d[rr][0] * w4[jj][e][0] + d[rr][1] * w4[jj][e][1] + d[rr][2] * w4[jj][e][2];
                    gg[e] = (hh[e] > 0.f) ? v : 0.f;
                    s3[jj][e] += gg[e];
#pragma unroll
                    for (int c = 0; c < 3; ++c) a4[jj][e][c] += hh[e] * d[rr][c];
                }
                if (g3) *reinterpret_cast<float4*>(g3 + (size_t)row * H + 256 * jj + 4 * lane) = make_float4(gg[0], gg[1], gg[2], gg[3]);
                if (gp.rc || gp.r8) {
                    unsigned pl4[4][3];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (gp.np == 1) { pl4[e][0] = bf16_bits(gg[e]); pl4[e][1] = 0; pl4[e][2] = 0; }     // one plane: one conversion, not the three-plane split
                        else split3(gg[e], pl4[e]);
                    }
                    const int col = 256 * jj + 4 * lane;
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const uint2 w = make_uint2(pl4[0][q] | (pl4[1][q] << 16), pl4[2][q] | (pl4[3][q] << 16));
                        wpk[rr][jj][q] = w;      // kept in registers: the LDS image for the R8 chunks reuses the slabs once they are summed
                    }
                    (void)col;
                }
            }
        }
    }
    if (gp.rc) {
        // RC planes: lane 2p stores the eight columns of lanes 2p / 2p + 1 of row 0, lane 2p + 1 those of row 1 -- one 16-byte store per
        // lane, column group and plane instead of two 8-byte ones (the planes exist only for whole 8-row blocks: both rows are valid)
        uint16_t* rcp = gp.rc + (size_t)(row0 + (odd ? 1 : 0)) * H + 8 * (lane >> 1);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            if (jj < ng) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    if (q < gp.np) {
                        const uint2 got = swap_pair(odd ? wpk[0][jj][q] : wpk[1][jj][q]);
                        const uint2 lo = odd ? got : wpk[0][jj][q], hi = odd ? wpk[1][jj][q] : got;
                        *reinterpret_cast<uint4*>(rcp + q * gp.plane + 256 * jj) = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    }
                }
            }
        }
    }
    OB_STAMP(5);
    float* mine = s_acc + wave * P;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        if (jj < ng) {
            const int k = 256 * jj + 4 * lane;          // four columns: 4 + 12 contiguous floats, 16-byte aligned (H % 256 == 0, P % 4 == 0)
            *reinterpret_cast<float4*>(mine + k) = make_float4(s3[jj][0], s3[jj][1], s3[jj][2], s3[jj][3]);
            float4* m4 = reinterpret_cast<float4*>(mine + H + k * 3);
            m4[0] = make_float4(a4[jj][0][0], a4[jj][0][1], a4[jj][0][2], a4[jj][1][0]);
            m4[1] = make_float4(a4[jj][1][1], a4[jj][1][2], a4[jj][2][0], a4[jj][2][1]);
            m4[2] = make_float4(a4[jj][2][2], a4[jj][3][0], a4[jj][3][1], a4[jj][3][2]);
        }
    }
    if (lane < 3) mine[4 * H + lane] = (lane == 0) ? dsum[0] : ((lane == 1) ? dsum[1] : dsum[2]);
    if (lane < 3) mine[4 * H + 4 + lane] = (lane == 0) ? lsum[0] : ((lane == 1) ? lsum[1] : lsum[2]);
    if (lane == 3) { mine[4 * H + 3] = 0.f; mine[4 * H + 7] = 0.f; }
    __syncthreads();
    OB_STAMP(6);
    float* out = scratch + (size_t)blockIdx.x * P;
    for (int i = threadIdx.x; i < 4 * H + 7; i += 256) out[i] = ((s_acc[i] + s_acc[P + i]) + s_acc[2 * P + i]) + s_acc[3 * P + i];
    if (gp.r8) {      // the block's 8 rows = one row group: column c's chunk = its 8 rows (Qb % 8 == 0 when planes are in use)
        uint16_t* s_g = reinterpret_cast<uint16_t*>(s_acc);      // [np][8][H] <= 48 KiB: inside the four slabs, which are summed by now
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < RW; ++rr)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                if (jj < ng)
                    for (int q = 0; q < gp.np; ++q)
                        *reinterpret_cast<uint2*>(s_g + ((size_t)(q * kOBRows + wave * RW + rr) * H + 256 * jj + 4 * lane)) = wpk[rr][jj][q < 3 ? q : 0];
        __syncthreads();
        for (int c = threadIdx.x; c < H; c += 256)
            for (int q = 0; q < gp.np; ++q) {
                unsigned b[8];
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) b[rr] = s_g[(size_t)(q * kOBRows + rr) * H + c];
                *reinterpret_cast<uint4*>(gp.r8 + q * gp.plane + ((size_t)blockIdx.x * H + c) * 8) =
                    make_uint4(b[0] | (b[1] << 16), b[2] | (b[3] << 16), b[4] | (b[5] << 16), b[6] | (b[7] << 16));
            }
    }
    OB_STAMP(7);
}

// out[i] = sum_b scratch[b][i] in a fixed order: 16 outputs x 16 block-slices per workgroup, LDS tree at the end
// rec = floats per block record (4H + 4, or 4H + 8 when the record carries the three loss sums at [4H+4 .. 4H+6]); loss (may be
// NULL) then receives loss_samples = sum|pred_AB - labels| / Qb and loss_pred = (sum pred_AB / Qb + sum pred_BA / Qb) / 2.
__global__ __launch_bounds__(256) void small_grads_reduce(const float* __restrict__ scratch, int nblk, int H,
                                                           float* __restrict__ db3, float* __restrict__ dW4,
                                                           float* __restrict__ db4, int rec, float* __restrict__ loss, int Qb) {
    __shared__ float red[16][17];
    __shared__ float s_loss[3];
    const int li = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + li;
    const int n = (rec > 4 * H + 4 && loss) ? 4 * H + 7 : 4 * H + 3;
    float s = 0.f;
    if (i < n)
        for (int b = sl; b < nblk; b += 16) s += scratch[(size_t)b * rec + i];
    red[sl][li] = s;
    __syncthreads();
    if (sl == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][li];
        if (i < H) { if (db3) db3[i] = t; }
        else if (i < 4 * H) { if (dW4) dW4[i - H] = t; }
        else if (i < 4 * H + 3) { if (db4) db4[i - 4 * H] = t; }
        else if (i >= 4 * H + 4) s_loss[i - 4 * H - 4] = t;       // the three loss sums land in ONE workgroup (16 outputs each)
    }
    if (n > 4 * H + 3 && blockIdx.x == (4 * H + 4) / 16) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const float inv = 1.0f / (float)Qb;
            loss[0] = s_loss[0] * inv;
            loss[1] = (s_loss[1] * inv + s_loss[2] * inv) / 2.0f;
        }
    }
}

// Column sums in two deterministic stages.  Stage 1: block (colblock, chunk) -> partial[chunk][...].
//   NW = 0: partial[chunk][n]     = sum_r g[r][n]                       (bias gradient)
//   NW = 3: partial[chunk][n*3+c] = sum_r a[r][n] * w[r*3+c]            (dW4 = h3^T dy)
constexpr int kColChunks = 16;

template <int NW>
__global__ __launch_bounds__(256) void colsum_stage1(const float* __restrict__ a, int lda, const float* __restrict__ w,
                                                      int R, int Ncols, float* __restrict__ partial) {
    __shared__ float red[4][64 * (NW ? NW : 1)];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rs = threadIdx.x >> 6, chunk = blockIdx.y;
    const int rper = (R + kColChunks - 1) / kColChunks;
    const int r0 = chunk * rper, r1 = min(R, r0 + rper);
    float acc[NW ? NW : 1];
#pragma unroll
    for (int c = 0; c < (NW ? NW : 1); ++c) acc[c] = 0.f;
    if (col < Ncols) {
        for (int r = r0 + rs; r < r1; r += 4) {
            const float x = a[(size_t)r * lda + col];
            if (NW == 0) acc[0] += x;
            else {
#pragma unroll
                for (int c = 0; c < NW; ++c) acc[c] += x * w[(size_t)r * NW + c];
            }
        }
    }
    constexpr int W = NW ? NW : 1;
#pragma unroll
    for (int c = 0; c < W; ++c) red[rs][(threadIdx.x & 63) * W + c] = acc[c];
    __syncthreads();
    if (rs == 0 && col < Ncols) {
#pragma unroll
        for (int c = 0; c < W; ++c) {
            const int i = (threadIdx.x & 63) * W + c;
            partial[((size_t)chunk * Ncols + col) * W + c] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        }
    }
}

// Single-launch variant: every (column block, row chunk) adds its partial with one atomic per output element
// (output must be zero on entry).  Used for db3 and dW4/db4 inside the backward data chain.
template <int NW>
__global__ __launch_bounds__(256) void colsum_atomic(const float* __restrict__ a, int lda, const float* __restrict__ w,
                                                      int R, int Ncols, float* __restrict__ out, float* __restrict__ wsum) {
    constexpr int W = NW ? NW : 1;
    __shared__ float red[4][64 * W];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rs = threadIdx.x >> 6, chunk = blockIdx.y;
    const int rper = (R + gridDim.y - 1) / gridDim.y;
    const int r0 = chunk * rper, r1 = min(R, r0 + rper);
    float acc[W];
#pragma unroll
    for (int c = 0; c < W; ++c) acc[c] = 0.f;
    if (col < Ncols) {
        for (int r = r0 + rs; r < r1; r += 4) {
            const float x = a[(size_t)r * lda + col];
            if (NW == 0) acc[0] += x;
            else {
#pragma unroll
                for (int c = 0; c < NW; ++c) acc[c] += x * w[(size_t)r * NW + c];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < W; ++c) red[rs][(threadIdx.x & 63) * W + c] = acc[c];
    __syncthreads();
    if (rs == 0 && col < Ncols) {
#pragma unroll
        for (int c = 0; c < W; ++c) {
            const int i = (threadIdx.x & 63) * W + c;
            atomicAdd(out + (size_t)col * W + c, (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]));
        }
    }
    if (NW && wsum && blockIdx.x == 0) {   // column sums of w itself (db4 = colsum(dy)), one block column does it
        float s[W];
#pragma unroll
        for (int c = 0; c < W; ++c) s[c] = 0.f;
        for (int r = r0 + threadIdx.x; r < r1; r += 256)
#pragma unroll
            for (int c = 0; c < W; ++c) s[c] += w[(size_t)r * NW + c];
#pragma unroll
        for (int c = 0; c < W; ++c) {
            s[c] = wave_sum(s[c]);
            if ((threadIdx.x & 63) == 0) atomicAdd(wsum + c, s[c]);
        }
    }
}

__global__ __launch_bounds__(256) void colsum_stage2(const float* __restrict__ partial, int n, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < kColChunks; ++c) s += partial[(size_t)c * n + i];
    out[i] = s;
}

// dy [R,3] column sums (db4): single block
__global__ __launch_bounds__(256) void dy_colsum_kernel(const float* __restrict__ dy, int R, float* __restrict__ db) {
    __shared__ float red[4][3];
    float a[3] = {0.f, 0.f, 0.f};
    for (int r = threadIdx.x; r < R; r += 256) {
        a[0] += dy[(size_t)r * 3]; a[1] += dy[(size_t)r * 3 + 1]; a[2] += dy[(size_t)r * 3 + 2];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        a[c] = wave_sum(a[c]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c] = a[c];
    }
    __syncthreads();
    if (threadIdx.x < 3) db[threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

static size_t colsum_ws_floats(int ncols, int nw) { return (size_t)kColChunks * ncols * (nw ? nw : 1); }

// out[c][r] = in[r][c] for up to three matrices in one launch (64x64 tiles through LDS, both sides coalesced)
struct TransposeJobs {
    const float* in[3];
    float* out[3];
    int R[3], C[3], blk0[4];
    int n;
};
__global__ __launch_bounds__(256) void transpose_kernel(TransposeJobs J) {
    __shared__ float t[64][65];
    int j = 0;
    while (j + 1 < J.n && (int)blockIdx.x >= J.blk0[j + 1]) ++j;
    const int R = J.R[j], C = J.C[j], tc = (C + 63) / 64;
    const int b = blockIdx.x - J.blk0[j];
    const int r0 = (b / tc) * 64, c0 = (b % tc) * 64;
    const float* __restrict__ in = J.in[j];
    float* __restrict__ out = J.out[j];
    const int x4 = (threadIdx.x & 15) * 4, y0 = threadIdx.x >> 4;          // 16 float4 per 64-wide row, 16 rows per pass
    const bool vec = !(C & 3) && !(R & 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = y0 + 16 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + y < R) {
            const float* src = in + (size_t)(r0 + y) * C + c0 + x4;
            if (vec && c0 + x4 + 3 < C) v = *reinterpret_cast<const float4*>(src);
            else {
                if (c0 + x4 < C) v.x = src[0];
                if (c0 + x4 + 1 < C) v.y = src[1];
                if (c0 + x4 + 2 < C) v.z = src[2];
                if (c0 + x4 + 3 < C) v.w = src[3];
            }
        }
        t[y][x4] = v.x; t[y][x4 + 1] = v.y; t[y][x4 + 2] = v.z; t[y][x4 + 3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = y0 + 16 * i;                                           // output row c0 + y, output columns r0 + x4 .. +3
        if (c0 + y < C) {
            const float4 v = make_float4(t[x4][y], t[x4 + 1][y], t[x4 + 2][y], t[x4 + 3][y]);
            float* dst = out + (size_t)(c0 + y) * R + r0 + x4;
            if (vec && r0 + x4 + 3 < R) *reinterpret_cast<float4*>(dst) = v;
            else {
                if (r0 + x4 < R) dst[0] = v.x;
                if (r0 + x4 + 1 < R) dst[1] = v.y;
                if (r0 + x4 + 2 < R) dst[2] = v.z;
                if (r0 + x4 + 3 < R) dst[3] = v.w;
            }
        }
    }
}

}  // namespace dpd

#ifdef DPD_ABLATIONS
extern "C" int dpd_debug_ob_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(dpd::g_ob_stamps), sizeof(unsigned long long) * 1024 * 8, 0, hipMemcpyDeviceToHost);
}
#endif
extern "C" int dpd_set_gemm_plan(int op, int tile, int split_k) {
    // plane GEMMs: ops 16.. = call sites, 32 = the grouped dW2 + dW3 launch; split_k: n > 1 in-launch reduction, n < -1 slabs + reduce
    // launch, 1 off, 0 automatic (decoder.hip: g_x3_split)
    if (op >= 16 && op < 16 + dpd::OP_COUNT && tile >= 0 && tile <= 26 && split_k >= -8 && split_k <= 8) {
        dpd::g_x3_tile[op - 16] = tile;
        if (op - 16 == dpd::OP_BWD_DW1 || op - 16 == dpd::OP_BWD_DW23) dpd::g_x3_split[op - 16] = split_k;
        return 0;
    }
    if (op == 32 && tile >= 0 && tile <= 16 && split_k >= -4 && split_k <= 4) { dpd::g_x3_pair_tile = tile; dpd::g_x3_pair_split = split_k; return 0; }
    if (op == 33 && tile >= 0 && tile <= 16 && split_k >= 1 && split_k <= 4) { dpd::g_x3_trio_tile = tile; dpd::g_x3_trio_split = split_k; return 0; }
    if (op < 0 || op >= dpd::OP_COUNT || tile < 0 || tile > 33 || split_k < 0 || split_k > 8) return DPD_E_DIM;
    dpd::g_plan_tile[op] = tile;
    dpd::g_plan_split[op] = split_k;
    return 0;
}

static size_t base_ws_bytes(int KP, int H) {
    const size_t slabs = (size_t)8 * (size_t)(KP > H ? KP : H) * H * sizeof(float);   // split-K <= 8 of the largest dW
    const size_t cols = dpd::colsum_ws_floats(H, 3) * sizeof(float);
    return (slabs + cols + 255) / 256 * 256 + dpd::kRedCntBytes;
}
// plane scratch = the tail of the workspace
static dpd::Scratch scratch_of(void* ws, size_t ws_bytes, int KP, int H) {
    const size_t base = base_ws_bytes(KP, H);
    if (!ws || ws_bytes <= base) return dpd::Scratch{nullptr, 0};
    return dpd::Scratch{(char*)ws + base, ws_bytes - base};
}

extern "C" size_t dpd_workspace_bytes(int Q, int KP, int H, int dtype) {
    return base_ws_bytes(KP, H) + dpd::plane_bytes(dtype, Q, KP, H);
}

namespace {
struct PlaneSizes {
    size_t X_rc, X_r8, h_rc, h_r8, g, W1, W23;
};
PlaneSizes plane_sizes(int Q, int Qb, int KP, int H, int np) {
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    PlaneSizes z;
    z.X_rc = al((size_t)np * 2 * Q * KP); z.X_r8 = al((size_t)np * 2 * Qb * KP);
    z.h_rc = al((size_t)np * 2 * Q * H);  z.h_r8 = al((size_t)np * 2 * Qb * H);
    z.g = al((size_t)np * 2 * Qb * H);
    z.W1 = al((size_t)np * 2 * KP * H);   z.W23 = al((size_t)np * 2 * H * H);
    return z;
}
}  // namespace

extern "C" size_t dpd_planes_bytes(int Q, int Qb, int KP, int H, int dtype, int with_dx) {
    if (dtype != 1 && dtype != 2) return 0;
    const PlaneSizes z = plane_sizes(Q, Qb, KP, H, dtype == 1 ? 3 : 1);
    return z.X_rc + z.X_r8 + 2 * (z.h_rc + z.h_r8) + (with_dx ? 6 : 5) * z.g + (with_dx ? 2 : 1) * z.W1 + 4 * z.W23 + (dtype == 2 ? z.h_rc : 0);
}

extern "C" int dpd_planes_carve(void* mem, size_t bytes, int Q, int Qb, int KP, int H, int dtype, int with_dx, dpd_planes* out) {
    if (!mem || !out) return DPD_E_NULL;
    if (dtype != 1 && dtype != 2) return DPD_E_UNSUPPORTED;
    if (Q <= 0 || Qb <= 0 || Qb > Q || KP <= 0 || H <= 0) return DPD_E_DIM;
    if (bytes < dpd_planes_bytes(Q, Qb, KP, H, dtype, with_dx)) return DPD_E_WORKSPACE;
    const int np = dtype == 1 ? 3 : 1;
    const PlaneSizes z = plane_sizes(Q, Qb, KP, H, np);
    char* c = (char*)mem;
    auto take = [&](size_t n) { void* r = c; c += n; return r; };
    *out = dpd_planes{};
    out->np = np; out->Q = Q; out->Qb = Qb;
    out->X_rc = take(z.X_rc); out->X_r8 = take(z.X_r8);
    out->h1_rc = take(z.h_rc); out->h1_r8 = take(z.h_r8); out->h2_rc = take(z.h_rc); out->h2_r8 = take(z.h_r8);
    out->g3_rc = take(z.g); out->g3_r8 = take(z.g); out->g2_rc = take(z.g); out->g2_r8 = take(z.g); out->g1_r8 = take(z.g);
    out->g1_rc = with_dx ? take(z.g) : nullptr;
    out->W1_r8 = take(z.W1); out->W2_r8 = take(z.W23); out->W3_r8 = take(z.W23);
    out->W2_rc = take(z.W23); out->W3_rc = take(z.W23);
    out->W1_rc = with_dx ? take(z.W1) : nullptr;
    out->h3_rc = (dtype == 2) ? take(z.h_rc) : nullptr;
    return 0;
}

extern "C" int dpd_weights_to_planes(const dpd_decoder_params* p, int KP, int H, const dpd_planes* pl, void* stream) {
    using namespace dpd;
    if (!p || !pl || !p->W1p || !p->W2 || !p->W3) return DPD_E_NULL;
    if (KP <= 0 || H <= 0 || (KP & 7) || (H & 7) || (pl->np != 1 && pl->np != 3)) return DPD_E_UNSUPPORTED;
    SplitJobs jobs{};
    auto add = [&](const float* src, int R, void* rc, void* r8) {
        if (!rc && !r8) return;
        jobs.j[jobs.n++] = SplitJob{src, (uint16_t*)rc, (uint16_t*)r8, (long)R * H, (long)R * H, R, H, H, H, pl->np, 0};
    };
    add(p->W1p, KP, pl->W1_rc, pl->W1_r8);
    add(p->W2, H, pl->W2_rc, pl->W2_r8);
    add(p->W3, H, pl->W3_rc, pl->W3_r8);
    if (!jobs.n) return 0;
    double by = 0.0;
    for (int t = 0; t < jobs.n; ++t) by += (double)jobs.j[t].R * H * (4.0 + pl->np * 2.0 * ((jobs.j[t].rc ? 1 : 0) + (jobs.j[t].r8 ? 1 : 0)));
    StageProf prof(stream, DPD_STAGE_WEIGHT_COPIES, by);
    return split_planes_multi(jobs, (hipStream_t)stream);
}

extern "C" int dpd_weights_transpose(const dpd_decoder_params* p, int KP, int H, float* W2T, float* W3T, float* W1pT, void* stream) {
    using namespace dpd;
    if (!p || !p->W2 || !p->W3 || !W2T || !W3T) return DPD_E_NULL;
    if (W1pT && !p->W1p) return DPD_E_NULL;
    if (KP <= 0 || H <= 0) return DPD_E_DIM;
    TransposeJobs J{};
    auto add = [&](const float* in, float* out, int R, int C) {
        J.in[J.n] = in; J.out[J.n] = out; J.R[J.n] = R; J.C[J.n] = C;
        J.blk0[J.n + 1] = J.blk0[J.n] + ((R + 63) / 64) * ((C + 63) / 64);
        ++J.n;
    };
    add(p->W2, W2T, H, H);
    add(p->W3, W3T, H, H);
    if (W1pT) add(p->W1p, W1pT, KP, H);
    StageProf prof(stream, DPD_STAGE_WEIGHT_COPIES, 8.0 * ((double)2 * H * H + (W1pT ? (double)KP * H : 0.0)));
    DPD_LAUNCH(transpose_kernel, dim3(J.blk0[J.n]), dim3(256), 0, (hipStream_t)stream, J);
    DPD_CHECK_LAUNCH();
    return 0;
}

extern "C" int dpd_decoder_fwd(const float* X, const float* mask, int Q, int KP, int H, const dpd_decoder_params* p,
                               int dtype, float* h1, float* h2, float* h3, float* y, float* pred, void* ws, size_t ws_bytes,
                               const dpd_planes* pl, void* stream) {
    using namespace dpd;
    pl = dpd::usable_planes(pl, dtype, Q, pl ? pl->Qb : 0, KP, H);
    if (!X && !(pl && pl->X_rc)) return DPD_E_NULL;
    // y = pred = NULL: the output layer is left to dpd_decoder_bwd_data (dpd_small_grads.fwd_y).  h1 / h2 = NULL: plane compute
    // types whose planes keep h1_rc / h2_rc need no fp32 copy (the next layer, the weight gradients and the ReLU gate of the backward
    // all read the planes): 2 x Q x H x 4 bytes less to write per forward
    // h3 = NULL (DPD_BF16, y = pred = NULL): layer 3 leaves its activation as the bf16 plane pl->h3_rc only
    const bool h3_plane = !h3 && pl && pl->np == 1 && pl->h3_rc && !y;
    if (!mask || !p || (!h3 && !h3_plane) || (!y != !pred)) return DPD_E_NULL;
    if ((!h1 && !(pl && pl->h1_rc)) || (!h2 && !(pl && pl->h2_rc))) return DPD_E_NULL;
    if (pl && (pl->Q != Q || pl->Qb > Q)) return DPD_E_DIM;
    if (int rc = dpd::check_planes(pl, dtype)) return rc;
    if (!p->W1p || !p->b1 || !p->W2 || !p->b2 || !p->W3 || !p->b3 || !p->W4 || !p->b4) return DPD_E_NULL;
    if (Q <= 0 || KP <= 0 || H <= 0) return DPD_E_DIM;
    if ((H & 63) || (KP & 3) || dtype < 0 || dtype > 2) return DPD_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const Scratch scr = scratch_of(ws, ws_bytes, KP, H);
    // layer 1..3: h = relu(in W + b)   (tf_util.conv2d: conv2d + bias_add + relu, utils/tf_util.py:213-227)
    if (pl) {   // operands from / results to the persistent planes (no conversion passes)
        X3Out o1 = make_out(pl, pl->h1_rc, Q, pl->h1_r8, pl->Qb, H), o2 = make_out(pl, pl->h2_rc, Q, pl->h2_r8, pl->Qb, H);
        const bool w1 = o1.rc || o1.r8, w2 = o2.rc || o2.r8;
        if (int rc = gemm_dt(dtype, OP_FWD_L1, 0, 0, Q, H, KP, X, KP, p->W1p, H, h1, H, p->b1, nullptr, 2, nullptr, 0, scr, s, nullptr,
                             pl->X_rc, pl->W1_r8, w1 ? &o1 : nullptr)) return rc;
        if (int rc = gemm_dt(dtype, OP_FWD_L23, 0, 0, Q, H, H, h1, H, p->W2, H, h2, H, p->b2, nullptr, 2, nullptr, 0, scr, s, nullptr,
                             pl->h1_rc, pl->W2_r8, w2 ? &o2 : nullptr)) return rc;
        X3Out o3 = make_out(pl, pl->h3_rc, Q, nullptr, 0, H);
        if (int rc = gemm_dt(dtype, OP_FWD_L23, 0, 0, Q, H, H, h2, H, p->W3, H, h3, H, p->b3, nullptr, 2, nullptr, 0, scr, s, nullptr,
                             pl->h2_rc, pl->W3_r8, h3_plane ? &o3 : nullptr)) return rc;
    } else {
        if (dtype != 0 && !scr.p) return DPD_E_WORKSPACE;
        if (int rc = gemm_dt(dtype, OP_FWD_L1, 0, 0, Q, H, KP, X, KP, p->W1p, H, h1, H, p->b1, nullptr, 2, nullptr, 0, scr, s)) return rc;
        if (int rc = gemm_dt(dtype, OP_FWD_L23, 0, 0, Q, H, H, h1, H, p->W2, H, h2, H, p->b2, nullptr, 2, nullptr, 0, scr, s)) return rc;
        if (int rc = gemm_dt(dtype, OP_FWD_L23, 0, 0, Q, H, H, h2, H, p->W3, H, h3, H, p->b3, nullptr, 2, nullptr, 0, scr, s)) return rc;
    }
    if (!y) return 0;
    DPD_LAUNCH(out_fwd_kernel, dim3((Q + 3) / 4), dim3(256), 0, s, h3, p->W4, p->b4, mask, y, pred, Q, H);
    DPD_CHECK_LAUNCH();
    return 0;
}

extern "C" int dpd_decoder_out_asloss(const float* h3, const float* mask, int Q, int H, int BN, const dpd_decoder_params* p, float gscale,
                                      float* y, float* pred, float* loss_pred, float* dy, float* g3, float* scratch, void* stream) {
    return dpd_decoder_out_asloss_planes(h3, mask, Q, H, BN, p, gscale, y, pred, loss_pred, dy, g3, nullptr, scratch, stream);
}

extern "C" int dpd_decoder_out_asloss_planes(const float* h3, const float* mask, int Q, int H, int BN, const dpd_decoder_params* p, float gscale,
                                             float* y, float* pred, float* loss_pred, float* dy, float* g3, const dpd_planes* pl, float* scratch,
                                             void* stream) {
    using namespace dpd;
    if (!h3 || !mask || !p || !p->W4 || !p->b4 || !y || !pred || !loss_pred || !scratch) return DPD_E_NULL;
    // pl with g3_rc (and dy given): g3 also (or, with g3 == NULL, only) as RC operand plane(s); needs the 16-byte fast path of the row kernel
    uint16_t* g3_rc = (pl && dy) ? (uint16_t*)pl->g3_rc : nullptr;
    if (g3_rc && (pl->Qb != Q || (pl->np != 1 && pl->np != 3) || (H & 255) || H > 256 * kOutMaxI || ((uintptr_t)p->W4 & 15) || ((uintptr_t)h3 & 15)))
        return DPD_E_UNSUPPORTED;
    if (dy && !g3 && !g3_rc) return DPD_E_NULL;
    if (!dy && g3) return DPD_E_NULL;
    if (Q <= 0 || H <= 0 || BN <= 0 || Q != 2 * BN) return DPD_E_DIM;
    if ((H & 3) || ((uintptr_t)scratch & 7) || Q > 8 * 65535 || BN >= (1 << 14)) return DPD_E_UNSUPPORTED;   // 16-bit block count; 48-bit sum: Q rows x at most 2.0 x 2^32 must stay below 2^48
    const float gv = 0.5f * (1.0f / (float)BN) * gscale;      // = l1_loss_kernel mode 2
    // two rows per wave: 9.9 us against 11.2 (one) and 11.9 (four) at the PCRNet batch -- the kernel is a chain of round trips (rows and
    // W4 -> wave sums -> stores -> the atomic's return), not a throughput problem
    DPD_LAUNCH(out_asloss_kernel<2>, dim3((Q + 7) / 8), dim3(256), 0, (hipStream_t)stream, h3, p->W4, p->b4, mask, y, pred, dy, g3, Q, H, BN, gv,
               loss_pred, (unsigned long long*)scratch, g3_rc, g3_rc ? (long)Q * H : 0L, g3_rc ? pl->np : 0);
    DPD_CHECK_LAUNCH();
    return 0;
}

extern "C" int dpd_decoder_bwd_data(const float* dpred, const float* mask, const float* y, const float* h1,
                                    const float* h2, const float* h3, int Qb, int KP, int H, const dpd_decoder_params* p,
                                    int dtype, float* dy, float* g3, float* g2, float* g1, float* dX,
                                    const dpd_small_grads* sg, void* ws, size_t ws_bytes, const dpd_planes* pl, int phases,
                                    void* stream) {
    using namespace dpd;
    const bool l1 = sg && sg->l1_labels;        // fused training loss: dpred is derived inside the output-layer kernel
    if (phases <= 0 || phases > 31) return DPD_E_DIM;
    // (phases without 1: the output layer was done before -- an earlier call or dpd_decoder_out_asloss -- and only g3 is read)
    // h3 = NULL: the bf16 plane pl->h3_rc of dpd_decoder_fwd (DPD_BF16; only the fused output-layer kernel reads it: checked below)
    const bool h3_plane = !h3 && pl && pl->np == 1 && pl->h3_rc;
    if (!p || ((phases & 1) && ((!dpred && !l1) || !mask || (!y && !(sg && sg->fwd_y)) || (!h3 && !h3_plane) || !dy))) return DPD_E_NULL;
    if ((!h1 && !(pl && pl->h1_rc)) || (!h2 && !(pl && pl->h2_rc))) return DPD_E_NULL;   // gate from the fp32 activation or its bf16 plane
    // g2 / g1 = NULL: plane compute types whose planes keep g2_rc + g2_r8 (g1_r8, and g1_rc when dX is wanted) need no fp32 copy of
    // the pre-activation gradients either (the next dH GEMM and the weight gradients read the planes; db2 / db1 come out of the
    // epilogue); the block partials then need their own scratch (sg->partials)
    // (phases without 1 -- the as-loss chain behind dpd_decoder_out_asloss -- has no block partials and, without weight gradients to
    //  follow, needs only the RC planes: g2_rc for the next dH GEMM, g1_rc for dX)
    if (!g2 && !(pl && pl->g2_rc && ((pl->g2_r8 && sg && sg->partials) || !(phases & 1)))) return DPD_E_NULL;
    if (!g1 && !(pl && (pl->g1_r8 || pl->g1_rc) && (!dX || pl->g1_rc))) return DPD_E_NULL;
    if (l1 && (!sg->l1_pred || !sg->l1_loss)) return DPD_E_NULL;
    if (!p->W1p || !p->W2 || !p->W3 || !p->W4) return DPD_E_NULL;
    if (Qb <= 0 || KP <= 0 || H <= 0) return DPD_E_DIM;
    if ((H & 63) || (KP & 3) || dtype < 0 || dtype > 2) return DPD_E_UNSUPPORTED;
    if (phases <= 0 || phases > 31) return DPD_E_DIM;
    hipStream_t s = (hipStream_t)stream;
    const Scratch scr = scratch_of(ws, ws_bytes, KP, H);
    pl = usable_planes(pl, dtype, pl ? pl->Q : 0, Qb, KP, H);
    if (pl && pl->Qb != Qb) return DPD_E_DIM;
    if (int rc = check_planes(pl, dtype)) return rc;
    if (dtype != 0 && !pl && !scr.p) return DPD_E_WORKSPACE;
    float* db1 = sg ? sg->db1 : nullptr;
    float* db2 = sg ? sg->db2 : nullptr;
    float* db3 = sg ? sg->db3 : nullptr;
    float* dW4 = sg ? sg->dW4 : nullptr;
    float* db4 = sg ? sg->db4 : nullptr;
    const int nblk = (Qb + kOBRows - 1) / kOBRows;
    const bool fused4 = H % 256 == 0 && H <= 1024;
    const int rec = fused4 ? 4 * H + kOBRec : 4 * H + 4;          // floats per block record
    const bool fused = (db3 || dW4 || db4 || l1) && H <= 64 * kOBMaxJ && (size_t)nblk * rec <= (size_t)Qb * H;
    if (l1 && !(fused && fused4)) return DPD_E_UNSUPPORTED;       // the caller then uses dpd_l1_loss + dpred
    if ((phases & 1) && h3_plane && !(fused && fused4 && pl)) return DPD_E_UNSUPPORTED;
    const L1Fuse lf{l1 ? sg->l1_pred : nullptr, l1 ? sg->l1_labels : nullptr, l1 ? sg->l1_gscale : 1.0f};
    const bool ofwd = sg && (sg->fwd_y || sg->fwd_pred) && (phases & 1);   // output layer's forward inside the same pass (training step)
    if (ofwd && !(l1 && fused4 && sg->fwd_y && sg->fwd_pred && p->b4)) return DPD_E_UNSUPPORTED;
    const OutFwd ofw{ofwd ? p->b4 : nullptr, ofwd ? sg->fwd_y : nullptr, ofwd ? sg->fwd_pred : nullptr};
    float* lossp = l1 ? sg->l1_loss : nullptr;
    const int nred = (4 * H + 7 + 15) / 16;
    // block partials of db3 / dW4 / db4: by default in g2 (free until the first dH GEMM); a caller that defers their reduction
    // (phases 16 / 8: it then runs beside the dH GEMMs on another stream) must provide sg->partials
    float* part = (sg && sg->partials) ? sg->partials : g2;
    if ((phases & 24) && !(sg && sg->partials)) return DPD_E_NULL;
    // g3 = NULL: only when the fused output-layer kernel writes g3 as operand planes itself (plane compute types, below)
    bool g3_planes_done = false;
    // (phases without 1 and no fp32 g3: dpd_decoder_out_asloss_planes has left g3 as the RC plane the first dH GEMM reads)
    if (!g3 && !(pl && pl->g3_rc && ((pl->g3_r8 && fused && fused4 && !(Qb % kOBRows)) || !(phases & 1)))) return DPD_E_NULL;
    if ((phases & 8) && fused) {   // deferred second stage of the small gradients (a side stream / graph branch runs it)
        DPD_LAUNCH(small_grads_reduce, dim3(nred), dim3(256), 0, s, (const float*)part, nblk, H, db3, dW4, db4, rec, lossp, Qb);
        DPD_CHECK_LAUNCH();
    }
    if (!(phases & 1)) {
        // output layer already done by an earlier call
    } else if (fused) {
        // one pass over h3: dy, g3 and block partials of db3 / dW4 / db4 (g2 is free until the first dH GEMM: scratch)
        ZeroList zl{{db1, db2, nullptr, nullptr, nullptr}, {H, H, 0, 0, 0}};
        if (fused4) {
            // plane compute types: g3 leaves this kernel as the operand planes of the first dH GEMM (RC) and of dW3 (R8)
            G3Planes gpl{nullptr, nullptr, (long)Qb * H, pl ? pl->np : 0};
            if (pl && !(Qb % kOBRows)) { gpl.rc = (uint16_t*)pl->g3_rc; gpl.r8 = (uint16_t*)pl->g3_r8; }
            g3_planes_done = gpl.rc || gpl.r8;
            const size_t lds = (size_t)4 * rec * sizeof(float);      // (the R8 image of g3, np * 8 * H * 2 bytes, reuses the slabs)
            static LdsOptIn lds_opt;
            if (int rc = ensure_dyn_lds(lds_opt, (const void*)out_bwd_fused4_kernel, lds)) return rc;
            // algorithmic bytes: layer 3's activation in (the Qb gradient rows, and their Qb twins when the layer's forward runs here
            // too; fp32 or one bf16 plane), g3 out (fp32 or both operand planes), the block partials, mask / labels / y / pred
            const double h3_b = h3 ? 4.0 : 2.0;
            StageProf prof(stream, DPD_STAGE_OUT_LAYER,
                           (double)Qb * H * h3_b * (ofwd ? 2.0 : 1.0) + (g3 ? (double)Qb * H * 4.0 : 0.0) +
                               ((gpl.rc ? 1.0 : 0.0) + (gpl.r8 ? 1.0 : 0.0)) * gpl.np * 2.0 * Qb * H + (double)nblk * rec * 4.0 + Qb * 40.0 + H * 12.0);
            DPD_LAUNCH(out_bwd_fused4_kernel, dim3(nblk), dim3(256), lds, s, dpred, mask, y, h3, p->W4, dy, g3, Qb, H, zl, part, lf, ofw, gpl,
                       h3 ? nullptr : (const uint16_t*)pl->h3_rc);
        } else {
            DPD_LAUNCH(out_bwd_fused_kernel, dim3(nblk), dim3(256), (size_t)(4 * H + 4) * sizeof(float), s, dpred, mask, y, h3,
                       p->W4, dy, g3, Qb, H, zl, part);
        }
        DPD_CHECK_LAUNCH();
        if (!(phases & 16)) {   // 16: the block partials stay in the scratch (= g2, which must not be overwritten before phase 8 ran)
            StageProf prof(stream, DPD_STAGE_SMALL_REDUCE, (double)nblk * rec * 4.0 + (4.0 * H + 3) * 4.0);
            DPD_LAUNCH(small_grads_reduce, dim3(nred), dim3(256), 0, s, (const float*)part, nblk, H, db3, dW4, db4, rec, lossp, Qb);
            DPD_CHECK_LAUNCH();
        }
    } else {
        ZeroList zl{{db1, db2, db3, dW4, db4}, {H, H, H, H * 3, 3}};
        DPD_LAUNCH(out_bwd_kernel, dim3((Qb + 3) / 4), dim3(256), 0, s, dpred, mask, y, h3, p->W4, dy, g3, Qb, H, zl);
        DPD_CHECK_LAUNCH();
        const int chunks = 16;
        if (db3) {   // db3 = colsum(g3)
            DPD_LAUNCH(colsum_atomic<0>, dim3((H + 63) / 64, chunks), dim3(256), 0, s, (const float*)g3, H, (const float*)nullptr, Qb,
                       H, db3, (float*)nullptr);
            DPD_CHECK_LAUNCH();
        }
        if (dW4) {   // dW4 = h3^T dy, db4 = colsum(dy)
            DPD_LAUNCH(colsum_atomic<3>, dim3((H + 63) / 64, chunks), dim3(256), 0, s, h3, H, (const float*)dy, Qb, H, dW4, db4);
            DPD_CHECK_LAUNCH();
        }
    }
    // g2 = (g3 W3^T) * [h2 > 0] ;  g1 = (g2 W2^T) * [h1 > 0];  db2 / db1 = column sums, fused into the epilogue
    if (pl) {
        // g3 from a kernel that wrote no planes (this call's unfused output layer, or -- phases without 1 and an fp32 g3 given --
        // dpd_decoder_out_asloss): one conversion launch for the layouts the planes keep
        if ((((phases & 1) && !g3_planes_done) || (!(phases & 1) && g3 && (phases & 2))) && (pl->g3_rc || pl->g3_r8)) {
            if (!g3) return DPD_E_NULL;
            if (int rc = split_planes(g3, Qb, H, H, pl->np, (uint16_t*)pl->g3_rc, H, (long)Qb * H, (uint16_t*)pl->g3_r8, (long)Qb * H, s))
                return rc;
        }
        X3Out o2 = make_out(pl, pl->g2_rc, Qb, pl->g2_r8, Qb, H), o1 = make_out(pl, dX ? pl->g1_rc : nullptr, Qb, pl->g1_r8, Qb, H);
        const bool w2 = o2.rc || o2.r8, w1 = o1.rc || o1.r8;
        if (phases & 2)
            if (int rc = gemm_dt(dtype, OP_BWD_DH, 0, 1, Qb, H, H, g3, H, p->W3, H, g2, H, nullptr, h2, 3, nullptr, 0, scr, s, db2, pl->g3_rc,
                                 pl->W3_rc, w2 ? &o2 : nullptr, nullptr, h2 ? nullptr : (pl->h2_r8 ? pl->h2_r8 : pl->h2_rc), pl->h2_r8 != nullptr)) return rc;
        if (phases & 4)
            if (int rc = gemm_dt(dtype, OP_BWD_DH, 0, 1, Qb, H, H, g2, H, p->W2, H, g1, H, nullptr, h1, 3, nullptr, 0, scr, s, db1, pl->g2_rc,
                                 pl->W2_rc, w1 ? &o1 : nullptr, nullptr, h1 ? nullptr : (pl->h1_r8 ? pl->h1_r8 : pl->h1_rc), pl->h1_r8 != nullptr)) return rc;
        if (dX && (phases & 4)) {
            if (int rc = gemm_dt(dtype, OP_BWD_DX, 0, 1, Qb, KP, H, g1, H, p->W1p, H, dX, KP, nullptr, nullptr, 0, nullptr, 0, scr, s, nullptr,
                                 pl->g1_rc, pl->W1_rc, nullptr)) return rc;
        }
        return 0;
    }
    // exact-fp32 path with transposed weight copies: the NT products become NN (weights read row-coalesced)
    const bool t3 = dtype == 0 && p->W3T, t2 = dtype == 0 && p->W2T, t1 = dtype == 0 && p->W1pT;
    // deterministic db2 / db1: the dH GEMMs store 32-row partial column sums of g2 / g1 into sg->db_partials ([2][ceil(Qb/32)][H]);
    // dpd_decoder_bwd_weights[_pair] called with the same pointer adds them up.  Needs the register-streamed kernels.
    float* dbp = (sg && sg->db_partials && dtype == 0) ? sg->db_partials : nullptr;
    const int nrb = (Qb + 31) / 32;
    const int op3 = t3 ? OP_BWD_DH_T : OP_BWD_DH, op2 = t2 ? OP_BWD_DH_T : OP_BWD_DH;
    if (dbp && !(g_plan_tile[op3] >= 30 && g_plan_tile[op2] >= 30)) return DPD_E_UNSUPPORTED;
    ColsumTwoStep c2{}, c1{};
    c2.part_out = dbp; c1.part_out = dbp ? dbp + (size_t)nrb * H : nullptr;
    if (phases & 2)
        if (int rc = gemm_dt(dtype, op3, 0, t3 ? 0 : 1, Qb, H, H, g3, H, t3 ? p->W3T : p->W3, H, g2, H, nullptr, h2, 3,
                             nullptr, 0, scr, s, dbp ? nullptr : db2, nullptr, nullptr, nullptr, dbp ? &c2 : nullptr)) return rc;
    if (phases & 4)
        if (int rc = gemm_dt(dtype, op2, 0, t2 ? 0 : 1, Qb, H, H, g2, H, t2 ? p->W2T : p->W2, H, g1, H, nullptr, h1, 3,
                             nullptr, 0, scr, s, dbp ? nullptr : db1, nullptr, nullptr, nullptr, dbp ? &c1 : nullptr)) return rc;
    if (dX && (phases & 4)) {   // as-loss mode: gradient w.r.t. the gathered rows, dX = g1 W1p^T  [Qb,KP]
        if (int rc = gemm_dt(dtype, t1 ? OP_BWD_DX_T : OP_BWD_DX, 0, t1 ? 0 : 1, Qb, KP, H, g1, H, t1 ? p->W1pT : p->W1p, t1 ? KP : H, dX, KP,
                             nullptr, nullptr, 0, nullptr, 0, scr, s)) return rc;
    }
    return 0;
}

extern "C" int dpd_decoder_bwd_weights(int layer, const float* act, int lda, const float* g, int Qb, int Kin, int Nout,
                                       int dtype, float* dW, float* db, void* ws, size_t ws_bytes, const dpd_planes* pl,
                                       const float* db_partials, void* stream) {
    using namespace dpd;
    if (!dW || ((!act || !g) && !(pl && dtype != 0))) return DPD_E_NULL;   // (act / g may be NULL when their R8 planes exist: checked below)
    if (layer == 4 && (!db || !act || !g)) return DPD_E_NULL;
    if (layer < 1 || layer > 4 || Qb <= 0 || Kin <= 0 || Nout <= 0 || lda < Kin) return DPD_E_DIM;
    if (dtype < 0 || dtype > 2) return DPD_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (layer == 4) {
        if (Nout != 3) return DPD_E_DIM;
        if (!ws || ws_bytes < colsum_ws_floats(Kin, 3) * sizeof(float)) return DPD_E_WORKSPACE;
        float* part = (float*)ws;
        DPD_LAUNCH(colsum_stage1<3>, dim3((Kin + 63) / 64, kColChunks), dim3(256), 0, s, act, lda, g, Qb, Kin, part);
        DPD_CHECK_LAUNCH();
        DPD_LAUNCH(colsum_stage2, dim3((Kin * 3 + 255) / 256), dim3(256), 0, s, (const float*)part, Kin * 3, dW);
        DPD_CHECK_LAUNCH();
        DPD_LAUNCH(dy_colsum_kernel, dim3(1), dim3(256), 0, s, g, Qb, db);
        DPD_CHECK_LAUNCH();
        return 0;
    }
    if ((Nout & 3) || (Kin & 3) || (lda & 3)) return DPD_E_UNSUPPORTED;
    const int op = (layer == 1) ? OP_BWD_DW1 : OP_BWD_DW23;
    const int split = dtype == 0 ? g_plan_split[op] : 1;      // 0 = tail split: up to 3 slabs for the last round's K pieces
    size_t slab_bytes = (split > 1) ? (size_t)split * Kin * Nout * sizeof(float) : 0;
    if (split == 0) slab_bytes = (ws && ws_bytes >= (size_t)3 * Kin * Nout * sizeof(float)) ? (size_t)3 * Kin * Nout * sizeof(float) : 0;
    const size_t col_bytes = db ? colsum_ws_floats(Nout, 0) * sizeof(float) : 0;
    if ((slab_bytes + col_bytes) && (!ws || ws_bytes < slab_bytes + col_bytes)) return DPD_E_WORKSPACE;
    const Scratch scr = scratch_of(ws, ws_bytes, Kin > Nout ? Kin : Nout, Nout);
    pl = usable_planes(pl, dtype, pl ? pl->Q : 0, Qb, layer == 1 ? Kin : 32, Nout);
    if (pl && pl->Qb != Qb) return DPD_E_DIM;
    if (int rc = check_planes(pl, dtype)) return rc;
    const void* apl = !pl ? nullptr : (layer == 1 ? pl->X_r8 : (layer == 2 ? pl->h1_r8 : pl->h2_r8));
    const void* gpl = !pl ? nullptr : (layer == 1 ? pl->g1_r8 : (layer == 2 ? pl->g2_r8 : pl->g3_r8));
    if (dtype != 0 && !(apl && gpl) && !scr.p) return DPD_E_WORKSPACE;
    if ((!act && !apl) || (!g && !gpl) || (!g && db)) return DPD_E_NULL;
    // dW [Kin,Nout] = act^T [Kin,Qb] * g [Qb,Nout].  On the register-streamed fp32 kernels the bias gradient db = colsum(g) falls out
    // of the B operand the GEMM streams anyway (deterministic, no atomics, no extra launch).
    // With db_partials (the 32-row partial column sums dpd_decoder_bwd_data stored for this layer's g: layers 1 and 2) the bias
    // gradient is finished by this GEMM's first-row-block waves: deterministic, no atomics, no extra launch.
    const int tile_w = g_plan_tile[op];
    const bool free_db = db && db_partials && (layer == 1 || layer == 2) && dtype == 0 && tile_w >= 30 && tile_w <= 39 && !(Qb % 32);
    if (db && db_partials && !free_db) return DPD_E_UNSUPPORTED;
    ColsumTwoStep cs{};
    cs.part_in = db_partials ? db_partials + (layer == 1 ? (size_t)((Qb + 31) / 32) * Nout : 0) : nullptr;
    cs.out = db; cs.nparts = (Qb + 31) / 32;
    // plane compute types: the slab region + arrival words of the base workspace serve the in-launch split-K of gemm_dt
    const size_t base_b = base_ws_bytes(Kin > Nout ? Kin : Nout, Nout);
    const size_t gemm_ws = dtype == 0 ? slab_bytes : ((ws && ws_bytes >= base_b) ? base_b : 0);
    if (int rc = gemm_dt(dtype, op, 1, 0, Kin, Nout, Qb, act, lda, g, Nout, dW, Nout, nullptr, nullptr, 0, ws, gemm_ws, scr, s, nullptr,
                         apl, gpl, nullptr, free_db ? &cs : nullptr))
        return rc;
    if (!db || free_db) return 0;   // bias gradient not wanted / already produced
    float* part = (float*)((char*)ws + slab_bytes);
    DPD_LAUNCH(colsum_stage1<0>, dim3((Nout + 63) / 64, kColChunks), dim3(256), 0, s, g, Nout, (const float*)nullptr,
                       Qb, Nout, part);
    DPD_CHECK_LAUNCH();
    DPD_LAUNCH(colsum_stage2, dim3((Nout + 255) / 256), dim3(256), 0, s, (const float*)part, Nout, db);
    DPD_CHECK_LAUNCH();
    return 0;
}

// dW of layers 2 and 3 in ONE grouped launch (identical shapes [H,H] = act^T g): 2 x 256 tiles fill the chip twice
// as well as two 256-tile launches and pay one prologue/epilogue instead of two.
extern "C" int dpd_decoder_bwd_weights_pair(const float* actA, const float* gA, float* dWA, const float* actB, const float* gB,
                                            float* dWB, int lda, int Qb, int Kin, int Nout, int dtype, void* ws,
                                            size_t ws_bytes, const dpd_planes* pl, float* dbA, const float* db_partials, void* stream) {
    using namespace dpd;
    if (!dWA || !dWB) return DPD_E_NULL;
    if ((!actA || !actB || !gA || !gB) && !(dtype != 0 && pl && pl->h1_r8 && pl->h2_r8 && pl->g2_r8 && pl->g3_r8)) return DPD_E_NULL;
    if (Qb <= 0 || Kin <= 0 || Nout <= 0 || lda < Kin) return DPD_E_DIM;
    if (dtype < 0 || dtype > 2 || (Nout & 3) || (Kin & 3) || (lda & 3) || (Qb & 31)) return DPD_E_UNSUPPORTED;
    if (dtype != 0 && dbA) return DPD_E_UNSUPPORTED;
    if (dtype != 0) {   // plane path: two launches (the planes of one GEMM at a time live in the scratch)
        const Scratch scr = scratch_of(ws, ws_bytes, Kin > Nout ? Kin : Nout, Nout);
        pl = usable_planes(pl, dtype, pl ? pl->Q : 0, Qb, 32, Nout);
        if (pl && pl->Qb != Qb) return DPD_E_DIM;
        if (int rc = check_planes(pl, dtype)) return rc;
        const bool have = pl && pl->h1_r8 && pl->g2_r8 && pl->h2_r8 && pl->g3_r8;   // pair = (layer 2, layer 3)
        if (!have && !scr.p) return DPD_E_WORKSPACE;
        if (have) {
            // both GEMMs in ONE grouped launch of 64x128 tiles over the whole K (2 x 128 workgroups).  A split-K-2 form on 128x128
            // tiles (2 x 64 x 2 workgroups, two fp32 slabs added in a fixed order) is available through dpd_set_gemm_plan(32,
            // tile, 2) and tested, but measured slower in every compute type (bf16 B=64: 0.3903 vs 0.3861 ms per step; the slab
            // traffic and the reduce launch cost more than the shorter K loop saves).
            // Round 4: split-K with the reduction INSIDE the launch (gemm_x3.hip: inlaunch_reduce; g_x3_pair_split > 1 or automatic):
            // 2 x 64 tiles of 128x128 x 3-4 slices = two workgroups per CU instead of one on half of them.
            const long pe = (long)Kin * Qb, ge = (long)Qb * Nout;
            int split = g_x3_pair_split, ptile = g_x3_pair_tile;
            void* cnt = nullptr;
            const size_t base_b = base_ws_bytes(Kin > Nout ? Kin : Nout, Nout);
            const bool ws_ok = ws && ws_bytes >= base_b;
            if (split == 0) {                                     // automatic
                if (!ptile) ptile = pl->np == 1 ? 2 : 3;
                split = (ws_ok && ptile <= 5) ? x3_auto_split(pl->np, ptile, 2 * Kin, Nout, Qb) : 1;
                if (split == 1 && !g_x3_pair_tile) ptile = 3;
            }
            size_t slab_b = ws_bytes;
            if (split > 1 && ws_ok) {                             // ... with no more slices than the slab region holds (tile-padded)
                const int rows2[2] = {Kin, Kin};
                split = x3_fit_split(ptile ? ptile : 2, split, rows2, 2, Nout, base_b - kRedCntBytes);
                if (split == 1 && !g_x3_pair_tile) ptile = 3;
            }
            if (split > 1) {                                      // in-launch reduction
                const int chunk = (((Qb + split - 1) / split) + 63) / 64 * 64;
                if (!ws_ok || ptile > 5 || chunk * (split - 1) >= Qb) split = 1;
                else { cnt = (char*)ws + base_b - kRedCntBytes; slab_b = base_b - kRedCntBytes; if (!ptile) ptile = 2; }
            } else if (split < -1) {                              // slabs + reduce launch (round 2)
                split = -split;
                if (Qb % (128 * split) || !ws || (size_t)2 * split * Kin * Nout * sizeof(float) > ws_bytes) split = 1;
            } else {
                split = 1;
            }
            if (!ptile) ptile = split > 1 ? 2 : 3;
            if (ptile >= 8 && (pl->np != 1 || Qb % (64 * split))) ptile = split > 1 ? 2 : 3;
            X3Extra ex;
            ex.A2 = (const uint16_t*)pl->h2_r8; ex.B2 = (const uint16_t*)pl->g3_r8; ex.C2 = dWB;
            return gemm_x3(pl->np, 1, 1, Kin, Nout, Qb, (const uint16_t*)pl->h1_r8, Kin, pe, (const uint16_t*)pl->g2_r8, Nout, ge, dWA, Nout,
                           nullptr, nullptr, 0, ptile, (hipStream_t)stream, nullptr, nullptr, &ex, split, ws, slab_b, nullptr, 0, cnt,
                           (int)(kRedCntBytes / 8));
        }
        if (int rc = gemm_dt(dtype, OP_BWD_DW23, 1, 0, Kin, Nout, Qb, actA, lda, gA, Nout, dWA, Nout, nullptr, nullptr, 0, nullptr, 0,
                             scr, (hipStream_t)stream, nullptr, have ? pl->h1_r8 : nullptr, have ? pl->g2_r8 : nullptr, nullptr)) return rc;
        return gemm_dt(dtype, OP_BWD_DW23, 1, 0, Kin, Nout, Qb, actB, lda, gB, Nout, dWB, Nout, nullptr, nullptr, 0, nullptr, 0, scr,
                       (hipStream_t)stream, nullptr, have ? pl->h2_r8 : nullptr, have ? pl->g3_r8 : nullptr, nullptr);
    }
    int tile = g_plan_tile[OP_BWD_DW23];
    if (!((tile == 8 || tile == 9) || (tile >= 30 && tile <= 33))) tile = 8;
    if (dbA && !(db_partials && tile >= 30 && tile <= 33)) return DPD_E_UNSUPPORTED;   // layer 2's bias gradient from the stored partials
    ColsumTwoStep cs{};
    cs.part_in = db_partials; cs.out = dbA; cs.nparts = (Qb + 31) / 32;     // (layer 2 = problem A: the first half of the scratch)
    return gemm_f32(1, 0, Kin, Nout, Qb, actA, lda, gA, Nout, dWA, Nout, nullptr, nullptr, 0, 1, tile, nullptr, 0,
                    (hipStream_t)stream, nullptr, actB, gB, dWB, dbA ? &cs : nullptr);
}

// dW1, dW2 and dW3 of a plane compute type in ONE grouped launch (round 4).  Every weight-gradient GEMM of the step contracts over the
// query rows (K = Qb = 4096 at B = 64): a workgroup's K loop takes ~27 us whatever the grid, and dW1 alone has 160 tiles of 128x128,
// dW2 / dW3 64 each, for 256 CUs -- launched one after the other they leave 96-192 CUs idle for 27 us each time (tools/
// overlap_probe_bf16.py: dW1 37 us, one dW 27-35 us, both on two streams 42 us).  Here the three problems share one grid (288 tiles of
// 128x128): all operands come from the R8 planes (X / g1, h1 / g2, h2 / g3).  DPD_E_UNSUPPORTED when the planes are not all there
// (the caller then uses dpd_decoder_bwd_weights + _pair); bias gradients are not produced here (plane types: dH epilogues).
extern "C" int dpd_decoder_bwd_weights_trio(int Qb, int KP, int H, int dtype, float* dW1, float* dW2, float* dW3, void* ws, size_t ws_bytes,
                                            const dpd_planes* pl, void* stream) {
    using namespace dpd;
    if (!dW1 || !dW2 || !dW3 || !pl) return DPD_E_NULL;
    if (Qb <= 0 || KP <= 0 || H <= 0) return DPD_E_DIM;
    if (dtype != 1 && dtype != 2) return DPD_E_UNSUPPORTED;
    pl = usable_planes(pl, dtype, pl->Q, Qb, KP, H);
    if (!pl) return DPD_E_UNSUPPORTED;
    if (pl->Qb != Qb) return DPD_E_DIM;
    if (int rc = check_planes(pl, dtype)) return rc;
    if (!(pl->X_r8 && pl->g1_r8 && pl->h1_r8 && pl->g2_r8 && pl->h2_r8 && pl->g3_r8) || (KP & 7) || (Qb % 64)) return DPD_E_UNSUPPORTED;
    // Default tile (one plane): 192x128 (gemm_x3.hip tile 13).  A workgroup's K loop is bound by the LDS-DMA fill of its CU
    // (~30 B/clk/CU for linear 1-KiB pieces), i.e. its time is ~ (BM + BN) * K * 2 B / 72 GB/s whatever else runs, and a CU that gets
    // two tiles takes twice as long: the best tile is the one with the smallest BM + BN that still gives every CU at most ONE
    // workgroup -- 128x128 makes 288 tiles (32 CUs get two: 58 us), 192x128 makes 112 + 48 + 48 = 208 (36 us), 256x128 144 (44 us).
    // Measured in the bf16 step at B = 64 (profiles/r04_bf16_trio_sweep.txt): the three launches apart 71 us, grouped on 128x128
    // tiles 67 us, on 192x128 tiles 51 us (step 0.3107 -> 0.2909 ms on that box).
    int tile = g_x3_trio_tile ? g_x3_trio_tile : ((pl->np == 1 && !(Qb % 64)) ? 13 : 2), split = g_x3_trio_split;
    if (tile > 5 && (pl->np != 1 || split > 1 || (Qb % 64))) tile = 2;       // the BK = 64 tiles: one plane, whole K
    void* cnt = nullptr;
    size_t slab_b = 0;
    if (split > 1) {
        const size_t base_b = base_ws_bytes(KP > H ? KP : H, H);
        const int rows3[3] = {KP, H, H};
        if (ws && ws_bytes >= base_b) split = x3_fit_split(tile, split, rows3, 3, H, base_b - kRedCntBytes);
        const int chunk = (((Qb + split - 1) / split) + 63) / 64 * 64;
        if (split <= 1 || !ws || ws_bytes < base_b || chunk * (split - 1) >= Qb) split = 1;
        else { cnt = (char*)ws + base_b - kRedCntBytes; slab_b = base_b - kRedCntBytes; }
    }
    X3Extra ex;
    ex.A2 = (const uint16_t*)pl->h1_r8; ex.B2 = (const uint16_t*)pl->g2_r8; ex.C2 = dW2; ex.M2 = H;
    ex.A3 = (const uint16_t*)pl->h2_r8; ex.B3 = (const uint16_t*)pl->g3_r8; ex.C3 = dW3; ex.M3 = H;
    return gemm_x3(pl->np, 1, 1, KP, H, Qb, (const uint16_t*)pl->X_r8, KP, (long)KP * Qb, (const uint16_t*)pl->g1_r8, H, (long)Qb * H, dW1, H,
                   nullptr, nullptr, 0, tile, (hipStream_t)stream, nullptr, nullptr, &ex, split, split > 1 ? ws : nullptr, slab_b, nullptr, 0,
                   cnt, (int)(kRedCntBytes / 8));
}
