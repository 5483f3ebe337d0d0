// DPDist-as-a-loss engine: the as-loss evaluation (forward; backward to the two clouds) behind one entry point per direction, on buffers
// carved once from one caller-owned allocation (include/dpdist_capi.h: dpd_asloss).  Replaces the reference's spliced graph
//   pcrnet-registration/iterative_PCRNet_ours.py:229-257 (import_meta_graph + input_map, loss = mean of the two output means, gradients
//   w.r.t. the 'Network' scope only, i.e. THROUGH input1) and train_multi_gpu_pc_compare_dist.py:427-463 (the AUE splice).
// It sequences the library's entry points as dpdist_amd/model.py's autograd node does from Python (the backward's non-GEMM tail through
// dpd_asloss_tail: the same device routines in three launches instead of six), so the results are bit for bit those of the one-by-one calls (tests/test_gpu_parity.py); what it removes is ~30 allocations and a dozen
// ctypes calls per evaluation (one forward + backward per registration step at batch 16: the seven refinements before it are pose-network only,
// iterative_PCRNet_ours.py:414-441), and a fixed set of buffers is what lets the whole registration step be captured as a hipGraph.
#include "common.h"

namespace {

struct Layout {
    size_t off[40];
    size_t total;
};

constexpr size_t al(size_t b) { return (b + 255) / 256 * 256; }

struct Sizes {
    int Q, KP, G;
    size_t pts, fv, ssq, row, vox, X, h, q3, dX, scratch, mfv_ws, ws, wT2, wT1, pX, ph, pW1, pW23;
};

Sizes sizes_of(int B, int N, int m, int k, int H, int dtype) {
    Sizes z{};
    z.Q = 2 * B * N; z.KP = dpd_padded_width(k); z.G = m * m * m;
    const size_t f = sizeof(float), Q = (size_t)z.Q, KP = (size_t)z.KP;
    z.pts = al((size_t)2 * B * N * 3 * f); z.fv = al((size_t)2 * B * z.G * DPD_FV_CHANNELS * f);
    z.ssq = al((size_t)2 * B * DPD_MFV_SLICES * DPD_FV_CHANNELS * f);
    z.row = al(Q * f); z.vox = al(Q * sizeof(int32_t)); z.X = al(Q * KP * f); z.h = al(Q * H * f); z.q3 = al(Q * 3 * f); z.dX = al(Q * KP * f);
    z.scratch = 256; z.mfv_ws = al(dpd_mfv3d_bwd_workspace_bytes(2 * B, m)); z.ws = al(dpd_workspace_bytes(z.Q, z.KP, H, dtype));
    z.wT2 = al((size_t)H * H * f); z.wT1 = al((size_t)H * KP * f);
    const size_t np = dtype == DPD_F32_X3 ? 3 : 1;
    z.pX = al(np * 2 * Q * KP); z.ph = al(np * 2 * Q * H); z.pW1 = al(np * 2 * KP * H); z.pW23 = al(np * 2 * (size_t)H * H);
    return z;
}

bool shape_ok(int B, int N, int m, int k, int H, int dtype) {
    return B > 0 && N > 0 && N <= 4096 && (long)B * N < 16384 && m >= 1 && m <= 10 && k >= 1 && k <= 7 && (k & 1) && H > 0 && !(H & 63) &&
           dtype >= 0 && dtype <= 2;
}

// plane compute types only run on planes for these shapes (decoder.hip: usable_planes); other shapes take the exact type's buffers
bool planes_shape(const Sizes& z, int H, int dtype) { return dtype != DPD_F32 && !(z.Q & 31) && !(z.KP & 31) && !(H & 63); }

}  // namespace

extern "C" size_t dpd_asloss_bytes(int B, int N, int m, int k, int H, int dtype) {
    if (!shape_ok(B, N, m, k, H, dtype)) return 0;
    const Sizes z = sizes_of(B, N, m, k, H, dtype);
    size_t n = 2 * z.pts + z.fv + z.ssq + z.row + z.vox + z.h /*h3*/ + 3 * z.q3 + z.h /*g3*/ + z.dX + z.fv /*dfv*/ + z.pts /*dpts*/ + z.scratch +
               z.mfv_ws + z.ws;
    if (planes_shape(z, H, dtype)) n += z.pX + 5 * z.ph + 2 * z.pW1 + 4 * z.pW23;
    else n += z.X + 4 * z.h + 2 * z.wT2 + z.wT1;
    return n;
}

extern "C" int dpd_asloss_carve(void* mem, size_t bytes, int B, int N, int m, int k, int H, int dtype, float sigma, dpd_asloss* out) {
    if (!mem || !out) return DPD_E_NULL;
    if (!shape_ok(B, N, m, k, H, dtype)) return DPD_E_UNSUPPORTED;
    if (((uintptr_t)mem & 255) != 0) return DPD_E_UNSUPPORTED;
    if (bytes < dpd_asloss_bytes(B, N, m, k, H, dtype)) return DPD_E_WORKSPACE;
    const Sizes z = sizes_of(B, N, m, k, H, dtype);
    char* c = (char*)mem;
    auto take = [&](size_t n) { void* r = c; c += n; return r; };
    dpd_asloss e{};
    e.B = B; e.N = N; e.m = m; e.k = k; e.KP = z.KP; e.H = H; e.dtype = dtype; e.sigma = sigma;
    e.pts = (float*)take(z.pts); e.q = (float*)take(z.pts); e.fv = (float*)take(z.fv); e.ssq = (float*)take(z.ssq);
    e.mask = (float*)take(z.row); e.vox = (int32_t*)take(z.vox);
    e.h3 = (float*)take(z.h); e.y = (float*)take(z.q3); e.pred = (float*)take(z.q3); e.dy = (float*)take(z.q3); e.g3 = (float*)take(z.h);
    e.dX = (float*)take(z.dX); e.dfv = (float*)take(z.fv); e.dpts = (float*)take(z.pts);
    e.scratch = (float*)take(z.scratch);
    e.mfv_ws = take(z.mfv_ws); e.mfv_ws_bytes = z.mfv_ws; e.ws = take(z.ws); e.ws_bytes = z.ws;
    if (planes_shape(z, H, dtype)) {
        dpd_planes& pl = e.planes;
        pl.np = dtype == DPD_F32_X3 ? 3 : 1; pl.Q = z.Q; pl.Qb = z.Q;
        pl.X_rc = take(z.pX); pl.h1_rc = take(z.ph); pl.h2_rc = take(z.ph); pl.g3_rc = take(z.ph); pl.g2_rc = take(z.ph); pl.g1_rc = take(z.ph);
        pl.W1_r8 = take(z.pW1); pl.W1_rc = take(z.pW1);
        pl.W2_r8 = take(z.pW23); pl.W3_r8 = take(z.pW23); pl.W2_rc = take(z.pW23); pl.W3_rc = take(z.pW23);
    } else {
        e.X = (float*)take(z.X); e.h1 = (float*)take(z.h); e.h2 = (float*)take(z.h); e.g2 = (float*)take(z.h); e.g1 = (float*)take(z.h);
        e.W2T = (float*)take(z.wT2); e.W3T = (float*)take(z.wT2); e.W1pT = (float*)take(z.wT1);
    }
    *out = e;
    return 0;
}

extern "C" int dpd_asloss_init(const dpd_asloss* e, void* stream) {
    if (!e || !e->scratch) return DPD_E_NULL;
    DPD_HIP(hipMemsetAsync(e->scratch, 0, 256, (hipStream_t)stream));
    return 0;
}

extern "C" int dpd_asloss_set_weights(dpd_asloss* e, const dpd_decoder_params* p, void* stream) {
    if (!e || !p) return DPD_E_NULL;
    if (!p->W1p || !p->b1 || !p->W2 || !p->b2 || !p->W3 || !p->b3 || !p->W4 || !p->b4) return DPD_E_NULL;
    e->params = *p;
    e->params.W2T = e->params.W3T = e->params.W1pT = nullptr;
    if (e->planes.np) return dpd_weights_to_planes(&e->params, e->KP, e->H, &e->planes, stream);
    if (e->dtype != DPD_F32) return 0;        // a plane type on a shape the plane kernels do not take: converts per GEMM into its workspace
    if (int rc = dpd_weights_transpose(&e->params, e->KP, e->H, e->W2T, e->W3T, e->W1pT, stream)) return rc;
    e->params.W2T = e->W2T; e->params.W3T = e->W3T; e->params.W1pT = e->W1pT;
    return 0;
}

extern "C" int dpd_asloss_forward(const dpd_asloss* e, const float* pcA, const float* pcB, int want_grad, float* loss, void* stream) {
    if (!e || !pcA || !pcB || !loss || !e->params.W1p) return DPD_E_NULL;
    const int C = 2 * e->B, Q = C * e->N;
    const dpd_planes* pl = e->planes.np ? &e->planes : nullptr;
    // input stacking + encoder (models/dpdist_and_aue.py:45,56-61,69; utils/dpdist_util.py:22-141), window gather (:911-930, :434-492)
    if (int rc = dpd_mfv3d_fwd_stacked(pcA, pcB, nullptr, e->B, e->N, e->m, e->sigma, e->pts, e->q, e->fv, e->ssq, stream)) return rc;
    if (int rc = dpd_patch_rows_fwd_scaled(e->q, e->fv, e->ssq, C, e->N, e->m, e->k, e->KP, pl ? nullptr : e->X, e->mask, e->vox, pl, stream)) return rc;
    // layers 1-3 (:513-544), then output layer + loss_pred (+ the output-layer backward) in one launch (:691-698, :976-979)
    if (int rc = dpd_decoder_fwd(pl ? nullptr : e->X, e->mask, Q, e->KP, e->H, &e->params, e->dtype, e->h1, e->h2, e->h3, nullptr, nullptr, e->ws,
                                 e->ws_bytes, pl, stream)) return rc;
    // plane compute types (H a multiple of 256): g3 leaves the fused output-layer kernel as the RC plane the first dH GEMM reads -- no fp32
    // g3, no conversion launch (same bits: tests compare the gradients with the entry-by-entry node's)
    const bool g3_plane = pl && want_grad && !(e->H & 255) && e->H <= 1024;
    return dpd_decoder_out_asloss_planes(e->h3, e->mask, Q, e->H, e->B * e->N, &e->params, 1.0f, e->y, e->pred, loss, want_grad ? e->dy : nullptr,
                                         (want_grad && !g3_plane) ? e->g3 : nullptr, g3_plane ? pl : nullptr, e->scratch, stream);
}

extern "C" int dpd_asloss_backward(const dpd_asloss* e, const float* upstream, float* gA, float* gB, void* stream) {
    if (!e || !gA || !gB || !e->params.W1p) return DPD_E_NULL;
    const int C = 2 * e->B, Q = C * e->N;
    const dpd_planes* pl = e->planes.np ? &e->planes : nullptr;
    // TF autodiff of the decoder down to its input rows, of the gather and of the encoder; gradients w.r.t. the two clouds only
    const bool g3_plane = pl && !(e->H & 255) && e->H <= 1024;       // as in dpd_asloss_forward: g3 is already the plane pl->g3_rc
    if (int rc = dpd_decoder_bwd_data(nullptr, nullptr, nullptr, e->h1, e->h2, nullptr, Q, e->KP, e->H, &e->params, e->dtype, nullptr,
                                      g3_plane ? nullptr : e->g3, e->g2,
                                      e->g1, e->dX, nullptr, e->ws, e->ws_bytes, pl, 6, stream)) return rc;
    // the non-GEMM tail: three launches (gather backward || encoder statistics, combine, apply + input gradients); the separate entries for
    // shapes the fused form does not take
    const int rc_tail = dpd_asloss_tail(e->dX, e->vox, e->pts, upstream, e->B, e->N, e->m, e->k, e->KP, e->sigma, e->dfv, e->mfv_ws, e->mfv_ws_bytes,
                                        gA, gB, stream);
    if (rc_tail != DPD_E_UNSUPPORTED) return rc_tail;
    if (int rc = dpd_patch_rows_bwd(e->dX, e->vox, C, e->N, e->m, e->k, e->KP, nullptr, e->dfv, stream)) return rc;
    if (int rc = dpd_mfv3d_bwd(e->pts, e->dfv, C, e->N, e->m, e->sigma, e->dpts, e->mfv_ws, e->mfv_ws_bytes, stream)) return rc;
    return dpd_asloss_combine(e->dpts, e->dX, upstream, e->B, e->N, e->k, e->KP, gA, gB, stream);
}

extern "C" int dpd_asloss_forward_backward(const dpd_asloss* e, const float* pcA, const float* pcB, float* loss, float* gA, float* gB,
                                           void* stream) {
    if (int rc = dpd_asloss_forward(e, pcA, pcB, 1, loss, stream)) return rc;
    return dpd_asloss_backward(e, nullptr, gA, gB, stream);
}
