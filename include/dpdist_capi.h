/*
 * dpdist_capi.h -- C ABI of the MI355X-native DPDist hot path (libdpdist_hip.so).
 *
 * The reference (dahliau/DPDist) has no FFI: its boundary is the Python model-module contract
 *     models/dpdist_and_aue.py:23-86,203-204   placeholder_inputs / get_model / get_loss
 * whose body is ~60 primitive TensorFlow ops in utils/dpdist_util.py.  Each entry point below
 * replaces one group of those ops (file:line cited per function); dpdist_amd/model.py re-assembles
 * them behind the same get_model/get_loss signatures.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous row-major float32 unless stated otherwise;
 *   - the caller owns every buffer: the library never allocates or frees device memory.  Process-wide state is limited to
 *     (a) the GEMM tuning table of dpd_set_gemm_plan (defaults = measured best; not synchronised: set it before use),
 *     (b) the opt-in profiler of dpd_prof_enable (off by default, mutex-protected, owns hipEvents while on) and
 *     (c) a per-(kernel, device) "large LDS opted in" flag set on a kernel's first launch on that device;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and is asynchronous;
 *   - return value: 0 = ok, <0 = argument error (DPD_E_*), >0 = the hipError_t of the failing call;
 *   - thread-safe / re-entrant: safe from several host threads on distinct streams.
 *
 * Shapes: C clouds of N points; grid side m (G = m^3 Gaussians == voxels); window side k;
 *   F = 20 Fisher features per Gaussian; E = k^3*F (2500); rows Q = C*N query points, row r = c*N+n
 *   is query point n of query-cloud c evaluated against the Fisher vector fv[c].
 *   The module contract stacks clouds as  pts = [pcA+noise ; pcB],  q = [pcB ; pcA]  (C = 2B).
 *   Decoder input row layout (internal, 16-byte friendly):  X[r] = [ emb(E) | q-centre (3) | 0-pad ]
 *   with leading dimension KP = dpd_padded_width(k) (2528 for k=5: a whole number of 32-deep K-tiles).  W1 is held in the matching
 *   row order: W1p[0:E] = tf_weights1[3:3+E], W1p[E:E+3] = tf_weights1[0:3], zero pad rows.
 */
#ifndef DPDIST_CAPI_H
#define DPDIST_CAPI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPD_FV_CHANNELS 20
#define DPD_OUT_CHANNELS 3

enum {
    DPD_OK = 0,
    DPD_E_NULL = -1,      /* a required pointer is NULL */
    DPD_E_DIM = -2,       /* non-positive or inconsistent dimension */
    DPD_E_UNSUPPORTED = -3, /* m > 10, k even, k > 7, N > 4096, widths not a multiple of 64 ... */
    DPD_E_WORKSPACE = -4  /* workspace too small (see dpd_workspace_bytes) */
};

/* library version / build target, e.g. "dpdist_hip 0.1 gfx950" */
const char* dpd_version(void);

/* KP: leading dimension of the decoder input rows for window side k (k^3*20+3 rounded up to 32). */
int dpd_padded_width(int k);

/* Input stacking of the module contract (models/dpdist_and_aue.py:45,56-61,69): pcA, pcB, noise [B,N,3]
 * (noise may be NULL) -> pts [2B,N,3] = [pcA+noise ; pcB] (encoder input), q [2B,N,3] = [pcB ; pcA] (queries).  */
int dpd_stack_clouds(const float* pcA, const float* pcB, const float* noise, int B, int N, float* pts, float* q,
                     void* stream);

/* ---------------------------------------------------------------------------------------------
 * 3DmFV encoder.  Replaces utils/dpdist_util.py:22-141 (get_3dmfv_tf, full_fv, normalize=True).
 *   pts [C,N,3] -> fv [C,m^3,20]; Gaussians on the fixed grid of :42-51, uniform weights 1/m^3.
 *   A point further than ~1.7 from every centre underflows every pdf and yields NaN exactly as the
 *   reference does (0/0 at :74).                                                                  */
int dpd_mfv3d_fwd(const float* pts, int C, int N, int m, float sigma, float* fv, void* stream);

/* Backward of the encoder (TF autodiff of :69-126): dfv [C,m^3,20] -> dpts [C,N,3] (overwritten).
 * Max/min gradients are split evenly among ties, like tf.reduce_max/min.                        */
int dpd_mfv3d_bwd(const float* pts, const float* dfv, int C, int N, int m, float sigma, float* dpts, void* ws,
                  size_t ws_bytes, void* stream);
/* ws (optional): dpd_mfv3d_bwd_workspace_bytes(C, m) bytes let the backward run as 4 workgroups per cloud (sliced over
 * the points, two launches) instead of one -- 2-3x faster at the as-loss batch sizes; NULL keeps the one-launch form. */
size_t dpd_mfv3d_bwd_workspace_bytes(int C, int m);

/* ---------------------------------------------------------------------------------------------
 * Query -> voxel lookup + local-window gather.  Replaces local_z_3d (utils/dpdist_util.py:911-930),
 * get_pc_grid_binary_mask_from_centers (:459-492) and get_emb_and_concat (:434-457) WITHOUT
 * materialising the [C,m^3,k^3*20] window tensor.
 *   q [C,N,3], fv [C,m^3,20] -> X [Q,KP] rows, mask [Q] (1/0), vox [Q] (voxel id, 0 if outside). */
typedef struct dpd_planes dpd_planes;   /* bf16 operand planes that persist between entry points; defined with the decoder below */
/* `pl` (may be NULL): also write X as operand planes pl->X_rc (all rows) and pl->X_r8 (rows < pl->Qb), so that
 * the decoder GEMMs of a bf16-matrix-core compute type need no separate conversion pass; X may then be NULL.  */
int dpd_patch_rows_fwd(const float* q, const float* fv, int C, int N, int m, int k, int KP, float* X,
                       float* mask, int32_t* vox, const struct dpd_planes* pl, void* stream);

/* The front end of a training step in TWO launches instead of four (stack, encoder, norm, gather):
 * dpd_mfv3d_fwd_stacked = dpd_stack_clouds + dpd_mfv3d_fwd in one kernel: it reads pcA (+noise) / pcB directly and also
 *   writes pts [2B,N,3] and q [2B,N,3] (either may be NULL).  With ssq != NULL ([2B][DPD_MFV_SLICES][20] floats) fv is
 *   left WITHOUT the final per-channel L2 normalisation (:124-126) and ssq receives the per-slice sums of squares;
 * dpd_patch_rows_fwd_scaled = dpd_patch_rows_fwd that applies that normalisation while it gathers (ssq NULL = plain).
 * Same bits as the four-launch form (both use one summation order for the norms).                                   */
#define DPD_MFV_SLICES 4
int dpd_mfv3d_fwd_stacked(const float* pcA, const float* pcB, const float* noise, int B, int N, int m, float sigma,
                          float* pts, float* q, float* fv, float* ssq, void* stream);
int dpd_patch_rows_fwd_scaled(const float* q, const float* fv, const float* ssq, int C, int N, int m, int k, int KP,
                              float* X, float* mask, int32_t* vox, const struct dpd_planes* pl, void* stream);

/* Backward of the gather: dX [Q,KP] -> dq [C,N,3] (overwritten; = dX[:,E:E+3]) and
 * dfv [C,m^3,20] (overwritten; scatter-add of the window columns).  Either output may be NULL.  */
int dpd_patch_rows_bwd(const float* dX, const int32_t* vox, int C, int N, int m, int k, int KP, float* dq,
                       float* dfv, void* stream);

/* Tail of the as-loss backward (pcrnet-registration/iterative_PCRNet_ours.py:255-257, train_multi_gpu_pc_compare_dist.py:457-463:
 * gradients w.r.t. input1 / input2 only) in one launch: dpts [2B,N,3] from dpd_mfv3d_bwd (encoder route), dX [2BN,KP] from
 * dpd_decoder_bwd_data (its q - centre columns are the query route), `scale` = the upstream gradient as a DEVICE scalar (NULL = 1):
 *   gA [B,N,3] = scale * (dpts[0:B]  + dq[B:2B])      (pcA is the query cloud of the BA half)
 *   gB [B,N,3] = scale * (dpts[B:2B] + dq[0:B])                                                                         */
int dpd_asloss_combine(const float* dpts, const float* dX, const float* scale, int B, int N, int k, int KP, float* gA, float* gB,
                       void* stream);

/* The WHOLE non-GEMM tail of an as-loss backward -- dpd_patch_rows_bwd (dX -> dfv), dpd_mfv3d_bwd (dfv -> dpts) and dpd_asloss_combine -- in
 * THREE launches instead of six: [window-gather backward || the encoder backward's statistics pass] -> combine -> apply + both input
 * gradients.  Bit for bit the separate calls.  pts [2B,N,3] = the encoder input of the forward, vox [2BN] from the window gather, dX
 * [2BN,KP]; scratch: dfv [2B,m^3,20] floats and mfv_ws (dpd_mfv3d_bwd_workspace_bytes(2B, m)).  DPD_E_UNSUPPORTED for shapes the sliced
 * encoder backward does not take (N < 8, m > 8): make the separate calls then.
 * Replaces TF autodiff of local_z_3d / get_3dmfv_tf (utils/dpdist_util.py:22-141,911-930) inside tf.gradients(loss, inputs)
 * (pcrnet-registration/iterative_PCRNet_ours.py:255-257).                                                                        */
int dpd_asloss_tail(const float* dX, const int32_t* vox, const float* pts, const float* upstream, int B, int N, int m, int k, int KP,
                    float sigma, float* dfv, void* mfv_ws, size_t mfv_ws_bytes, float* gA, float* gB, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Implicit decoder (shared MLP).  Replaces tf_util.conv2d x4 (utils/dpdist_util.py:513-544,
 * utils/tf_util.py:161-228), relu6/3 (:691) and the mask multiply (:695-698).
 *   X [Q,KP] -> h1,h2,h3 [Q,H] (post-ReLU, kept for backward), y [Q,3] (pre-activation),
 *   pred [Q,3] = clip(y,0,6)/3 * mask.   W1p [KP,H], W2,W3 [H,H], W4 [H,3], biases [H],[H],[H],[3].
 *   H must be a multiple of 64.  dtype: 0 = fp32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32).     */
typedef struct dpd_decoder_params {
    const float* W1p; const float* b1;
    const float* W2;  const float* b2;
    const float* W3;  const float* b3;
    const float* W4;  const float* b4;
    /* optional (NULL = absent) transposed copies written by dpd_weights_transpose: W2T, W3T [H,H], W1pT [H,KP].  With them
     * the backward data GEMMs g2 = g3 W3^T, g1 = g2 W2^T, dX = g1 W1p^T read their weight operand row-coalesced (the
     * register-streamed fp32 kernel then runs them at the forward's rate); without them the LDS-ring kernel is used.  */
    const float* W2T; const float* W3T; const float* W1pT;
} dpd_decoder_params;

/* Refresh the transposed weight copies (one launch): W2T, W3T [H,H] and, if W1pT != NULL, W1pT [H,KP].  Call after
 * loading weights and after every optimizer step (training), once for frozen weights (as-loss mode).  DPD_F32 only.  */
int dpd_weights_transpose(const dpd_decoder_params* p, int KP, int H, float* W2T, float* W3T, float* W1pT, void* stream);

/* With planes that keep h1_rc / h2_rc (plane compute types), h1 / h2 may be NULL: the fp32 copies are then not written; pass
 * NULL for them to dpd_decoder_bwd_data / dpd_decoder_bwd_weights[_pair] as well (the ReLU gate is read from the bf16 plane).  In the same
 * way g2 / g1 of dpd_decoder_bwd_data may be NULL when the planes keep g2_rc + g2_r8 / g1_r8 (and sg->partials is given).          */
int dpd_decoder_fwd(const float* X, const float* mask, int Q, int KP, int H, const dpd_decoder_params* p,
                    int dtype, float* h1, float* h2, float* h3, float* y, float* pred, void* ws, size_t ws_bytes,
                    const dpd_planes* pl, void* stream);

/* `dtype` of the decoder entry points = compute type of the three wide layers (inputs/outputs are always fp32):
 *   DPD_F32     exact fp32 on the fp32 matrix-core instruction (bitwise an fmaf chain), no workspace needed in
 *               dpd_decoder_fwd / dpd_decoder_bwd_data (ws may be NULL);
 *   DPD_F32_X3  fp32-equivalent on the bf16 matrix cores: every operand is split into three bf16 planes
 *               (hi+mid+lo, exact to 2^-27) and each product is the sum of the six bf16 MFMA terms of weight
 *               >= 2^-16, accumulated in fp32 (error vs fp64 <= the DPD_F32 path's, tests/test_gpu_parity.py);
 *   DPD_BF16    operands rounded to bf16, fp32 accumulation (mixed-precision training, tolerance ~1e-2 relative).
 * DPD_F32_X3 / DPD_BF16 need ws = dpd_workspace_bytes(Q, KP, H, dtype) bytes (operand planes).  Shapes the plane
 * kernels do not take (contraction length not a multiple of 32, dims not multiples of 8) run as DPD_F32.       */
enum dpd_dtype { DPD_F32 = 0, DPD_F32_X3 = 1, DPD_BF16 = 2 };

/* Operand planes that persist between entry points (caller-owned memory; see dpd_planes_bytes / dpd_planes_carve).
 * Layouts as in dpd_split_planes: RC = [np][rows][cols] bf16, R8 = [np][rows/8][cols][8].  A NULL member means
 * "not kept": its producer skips it and its consumers convert from the fp32 tensor into `ws` instead.  A non-NULL
 * member handed to a CONSUMER must hold what its producer wrote for the current tensors:
 *   producer                         members
 *   dpd_patch_rows_fwd               X_rc [Q,KP], X_r8 (rows < Qb)
 *   dpd_decoder_fwd                  h1_rc, h2_rc [Q,H]; h1_r8, h2_r8 (rows < Qb)          consumes X_rc, W*_r8
 *   dpd_decoder_bwd_data             g3_rc, g3_r8, g2_rc, g2_r8, g1_rc, g1_r8 [Qb,H]       consumes W*_rc
 *   dpd_weights_to_planes            W1_r8, W2_r8, W3_r8 (forward), W1_rc, W2_rc, W3_rc (backward dH / dX)
 *   dpd_decoder_bwd_weights[_pair]   -                                                     consumes X_r8/h*_r8, g*_r8
 * Planes are only used when Q % 8 == 0, Qb % 32 == 0, KP % 32 == 0 (otherwise `pl` is ignored).              */
struct dpd_planes {
    int np;       /* 3 for DPD_F32_X3, 1 for DPD_BF16 */
    int Q, Qb;    /* rows of X/h1/h2 planes; rows that carry gradient (R8 planes of X/h*, all g planes) */
    void *X_rc, *X_r8, *h1_rc, *h1_r8, *h2_rc, *h2_r8;
    void *g3_rc, *g3_r8, *g2_rc, *g2_r8, *g1_rc, *g1_r8;
    void *W1_r8, *W2_r8, *W3_r8, *W1_rc, *W2_rc, *W3_rc;
    void* h3_rc;  /* DPD_BF16 only (NULL otherwise): layer 3's activation as ONE bf16 plane [Q,H] instead of fp32 -- dpd_decoder_fwd writes it when
                   * h3 == NULL and y == NULL, the fused output-layer kernel of dpd_decoder_bwd_data (which then runs the output layer's forward
                   * as well: dpd_small_grads.fwd_y) reads it; 2 instead of 4 bytes per element written once and read once per step */
};

/* Bytes for ALL members (with_dx: also g1_rc and W1_rc, needed only when dX is requested), and the carve-up of one
 * caller buffer of that size into the members (host-side pointer arithmetic only).                          */
size_t dpd_planes_bytes(int Q, int Qb, int KP, int H, int dtype, int with_dx);
int dpd_planes_carve(void* mem, size_t bytes, int Q, int Qb, int KP, int H, int dtype, int with_dx, dpd_planes* out);
/* Weight planes from the fp32 variables (one launch): call after loading weights and after every optimizer step. */
int dpd_weights_to_planes(const dpd_decoder_params* p, int KP, int H, const dpd_planes* pl, void* stream);

/* Backward, data chain: dpred [Qb,3] for the FIRST Qb rows (training mode: Qb = Q/2, only the AB half
 * carries gradient, train_multi_gpu_pc_compare_dist.py:274-277; as-loss mode: Qb = Q).
 * Produces the pre-activation gradients g3,g2,g1 [Qb,H] and dy [Qb,3]; if dX != NULL also
 * dX [Qb,KP] = g1 * W1p^T (as-loss mode, TF autodiff through :516).
 * `sg` (may be NULL) lists the small gradients that fall out of this chain for free and are then written here
 * (overwritten): db3/db2/db1 [H] = column sums of g3/g2/g1 (fused into the dH GEMM epilogues, fp32 atomics),
 * dW4 [H,3] = h3^T dy, db4 [3].  Any member may be NULL.
 * `phases` selects the parts of the chain to run, so that a data-parallel caller can interleave the weight-gradient
 * GEMMs (and start their all-reduce) between them: 1 = output layer (dy, g3, db3, dW4, db4; clears db1/db2),
 * 2 = g2 (+db2), 4 = g1 (+db1) and dX; 7 = everything.  The same buffers must be passed to every call.
 * 16 (with 1): leave db3 / dW4 / db4 as per-block partial sums in sg->partials; 8: reduce those partials (a call with
 * phases = 8 alone may run on another stream, beside the dH GEMMs, once the phase-1 call has been ordered before it).  */
typedef struct dpd_small_grads {
    float* db1; float* db2; float* db3; float* dW4; float* db4;
    float* partials;   /* optional scratch, ((Qb + 7) / 8) * (4 * H + 8) floats: required by phases 16 / 8 (see below) */
    /* optional fused training loss (utils/dpdist_util.py:962-980 + its autodiff, the mode-1 call of dpd_l1_loss): with
     * l1_labels [Qb] != NULL the output-layer backward derives d loss_samples / d pred_AB * l1_gscale from l1_pred [2*Qb,3]
     * itself (`dpred` may be NULL) and l1_loss [2] = (loss_samples, loss_pred) comes out of the same reduction as db3/dW4/db4.
     * Needs H % 256 == 0, H <= 1024 (else DPD_E_UNSUPPORTED: use dpd_l1_loss).                                              */
    const float* l1_pred; const float* l1_labels; float* l1_loss; float l1_gscale;
    /* optional scratch, 2 * ((Qb + 31) / 32) * H floats (DPD_F32): instead of atomic column sums into db2 / db1 (order-dependent
     * round-off), the dH GEMMs store 32-row partial sums of g2 / g1 here and dpd_decoder_bwd_weights(layer 2 / 1, db, ...,
     * db_partials = this pointer) or dpd_decoder_bwd_weights_pair(..., dbA, db_partials) finishes them: bitwise reproducible.
     * db1 / db2 above are then ignored.                                                                                       */
    float* db_partials;
    /* optional (both or neither; needs the fused loss above): the FORWARD of the output layer inside the same pass -- y and pred
     * [2*Qb,3] of the AB rows and their BA twins are computed from h3 (bit-identical to dpd_decoder_fwd's), written here and
     * used instead of the `y` / l1_pred inputs, which may then be NULL.  Call dpd_decoder_fwd with y = pred = NULL before.    */
    float* fwd_y; float* fwd_pred;
} dpd_small_grads;

int dpd_decoder_bwd_data(const float* dpred, const float* mask, const float* y, const float* h1, const float* h2,
                         const float* h3, int Qb, int KP, int H, const dpd_decoder_params* p, int dtype,
                         float* dy, float* g3, float* g2, float* g1, float* dX, const dpd_small_grads* sg,
                         void* ws, size_t ws_bytes, const dpd_planes* pl, int phases, void* stream);

/* Backward, weight gradients of ONE layer (1..4) from the buffers above:
 *   layer 1: dW [KP,H] = X^T g1, db = colsum(g1);  2: h1^T g2;  3: h2^T g3;  4: dW [H,3] = h3^T dy.
 * `act` is the layer's input activation (X, h1, h2 or h3) with leading dimension lda, `g` its
 * pre-activation output gradient.  dW/db are overwritten; db may be NULL for layers 1-3 when it was already
 * produced by dpd_decoder_bwd_data.  ws: dpd_workspace_bytes() bytes.                              */
int dpd_decoder_bwd_weights(int layer, const float* act, int lda, const float* g, int Qb, int Kin, int Nout,
                            int dtype, float* dW, float* db, void* ws, size_t ws_bytes, const dpd_planes* pl,
                            const float* db_partials, void* stream);

/* dW of two layers of identical shape (layers 2 and 3: dW = act^T g, [Kin,Nout]) in ONE grouped launch; no bias
 * gradients unless dbA / dbB are given (see below).  Qb must be a multiple of 32.                          */
int dpd_decoder_bwd_weights_pair(const float* actA, const float* gA, float* dWA, const float* actB, const float* gB,
                                 float* dWB, int lda, int Qb, int Kin, int Nout, int dtype, void* ws, size_t ws_bytes,
                                 const dpd_planes* pl, float* dbA, const float* db_partials, void* stream);
/* dbA + db_partials (may be NULL; DPD_F32 only): layer 2's bias gradient colsum(gA) [Nout], finished from the partial sums
 * dpd_decoder_bwd_data stored in dpd_small_grads::db_partials (see there).  db_partials == NULL with db != NULL in
 * dpd_decoder_bwd_weights keeps the older two-kernel column sum.                                                         */

/* Scratch needed by the decoder entry points for the given sizes and compute type (split-K slabs, column-sum
 * partials and, for dtype != DPD_F32, the bf16 operand planes of one GEMM at a time).                   */
size_t dpd_workspace_bytes(int Q, int KP, int H, int dtype);

/* ---------------------------------------------------------------------------------------------
 * Losses.  Replaces utils/dpdist_util.py:962-980.  pred [2*BN,3] (AB rows first), labels [BN].
 *   loss[0] = loss_samples = mean |pred_AB[:,0] - labels|        (:972)
 *   loss[1] = loss_pred    = (mean pred_AB[:,0] + mean pred_BA[:,0]) / 2      (:976-977)
 * mode 0: no gradient.  mode 1 (training): dpred [BN,3] = d loss_samples / d pred_AB * gscale.
 * mode 2 (as-loss): dpred [2*BN,3] = d loss_pred / d pred * gscale.                              */
int dpd_l1_loss(const float* pred, const float* labels, int BN, int mode, float gscale, float* loss,
                float* dpred, void* stream);

/* DPDist as a frozen loss (pcrnet-registration/iterative_PCRNet_ours.py:229-257; AUE splice train_multi_gpu_pc_compare_dist.py:417-431):
 * the output layer of the decoder (utils/dpdist_util.py:691,695-698), loss_pred (:976-977) and -- when dy / g3 are given -- the
 * output-layer backward for d loss_pred / d pred in ONE launch: replaces the tail of dpd_decoder_fwd (call it with y = pred = NULL),
 * dpd_l1_loss(mode 2) and phase 1 of dpd_decoder_bwd_data (call it with phases = 6 and this g3).
 *   h3 [Q,H], mask [Q], Q = 2*BN rows (AB half first), BN <= 16384;  y, pred [Q,3];  loss_pred [1] (the exact sum of pred[:,0] in
 *   2^-32 fixed point / (2 BN): within an ulp of the fp32 means of :976-977, the same bits on every run);
 *   dy [Q,3], g3 [Q,H] (both or neither) = the gradient of gscale * loss_pred;
 *   scratch: 8 bytes, 8-byte aligned, ZERO before the first call (the kernel leaves it zero; one per concurrently running stream). */
int dpd_decoder_out_asloss(const float* h3, const float* mask, int Q, int H, int BN, const dpd_decoder_params* p, float gscale,
                           float* y, float* pred, float* loss_pred, float* dy, float* g3, float* scratch, void* stream);
/* The same with g3 ALSO written as the RC operand plane(s) pl->g3_rc of the first data-gradient GEMM (plane compute types, pl->Qb == Q,
 * H % 256 == 0, H <= 1024; the bits dpd_split_planes would produce from the fp32 g3): g3 may then be NULL, and dpd_decoder_bwd_data
 * (phases = 6, g3 = NULL) goes on from the plane.  pl == NULL: exactly dpd_decoder_out_asloss.                                     */
int dpd_decoder_out_asloss_planes(const float* h3, const float* mask, int Q, int H, int BN, const dpd_decoder_params* p, float gscale,
                                  float* y, float* pred, float* loss_pred, float* dy, float* g3, const dpd_planes* pl, float* scratch,
                                  void* stream);

/* ---------------------------------------------------------------------------------------------
 * DPDist-as-a-loss ENGINE (round 5): the whole as-loss evaluation -- the reference's spliced graph
 *   input1, input2 -> pc_compare/output1, output2 -> (mean(output1[...,0]) + mean(output2[...,0])) / 2
 * (pcrnet-registration/iterative_PCRNet_ours.py:229-257, train_multi_gpu_pc_compare_dist.py:427-453) and its gradient w.r.t. the two
 * clouds -- behind ONE entry point per direction, on buffers sized once for a fixed (B, N).  Same kernels, same order and same bits as
 * calling dpd_mfv3d_fwd_stacked, dpd_patch_rows_fwd_scaled, dpd_decoder_fwd, dpd_decoder_out_asloss | dpd_decoder_bwd_data (phases 6),
 * dpd_patch_rows_bwd, dpd_mfv3d_bwd, dpd_asloss_combine one by one; what it removes is the host: ~30 buffer allocations and a dozen
 * foreign-function calls per evaluation bound the registration loop at its batch of 16 (one forward + backward per training step, one forward per evaluation batch).
 * All members point into ONE caller-owned allocation (dpd_asloss_bytes / dpd_asloss_carve: host-side pointer arithmetic only); the
 * DPDist weights are frozen in this mode: dpd_asloss_set_weights derives what the compute type needs from them (transposed fp32
 * copies or bf16 operand planes) ONCE and keeps the caller's parameter pointers -- call it again after the weights change.
 * B*N < 16384, N <= 4096, H % 64 == 0; one engine per concurrently running stream.                                               */
typedef struct dpd_asloss {
    int B, N, m, k, KP, H, dtype;      /* dtype: enum dpd_dtype */
    float sigma;
    float *pts, *q, *fv, *ssq, *mask;  /* front end: [2B,N,3] x2, [2B,m^3,20], [2B,DPD_MFV_SLICES,20], [Q] */
    int32_t* vox;                      /* [Q] */
    float *X, *h1, *h2;                /* DPD_F32 only (NULL otherwise: the plane types keep them as bf16 planes) [Q,KP], [Q,H] x2 */
    float *h3, *y, *pred, *dy, *g3;    /* [Q,H], [Q,3] x3, [Q,H] */
    float *g2, *g1;                    /* DPD_F32 only */
    float *dX, *dfv, *dpts;            /* [Q,KP], [2B,m^3,20], [2B,N,3] */
    float *W2T, *W3T, *W1pT;           /* DPD_F32 only: transposed weight copies (dpd_asloss_set_weights) */
    float* scratch;                    /* 8 bytes for dpd_decoder_out_asloss: zeroed by dpd_asloss_init */
    void* mfv_ws; size_t mfv_ws_bytes; /* dpd_mfv3d_bwd_workspace_bytes(2B, m) */
    void* ws; size_t ws_bytes;         /* dpd_workspace_bytes(Q, KP, H, dtype) */
    dpd_planes planes;                 /* plane compute types: X, h1, h2, g3, g2, g1 as RC planes, the weights' R8 + RC planes */
    dpd_decoder_params params;         /* filled by dpd_asloss_set_weights */
} dpd_asloss;
size_t dpd_asloss_bytes(int B, int N, int m, int k, int H, int dtype);
int dpd_asloss_carve(void* mem, size_t bytes, int B, int N, int m, int k, int H, int dtype, float sigma, dpd_asloss* out);
/* zero what must be zero before the first evaluation (the loss accumulator): once, on `stream` */
int dpd_asloss_init(const dpd_asloss* e, void* stream);
/* `e` is updated (its params member); p's W*T members are ignored (the engine owns its own transposed copies) */
int dpd_asloss_set_weights(dpd_asloss* e, const dpd_decoder_params* p, void* stream);
/* pcA, pcB [B,N,3] -> loss [1] (loss_pred, utils/dpdist_util.py:976-979).  want_grad != 0 also leaves what dpd_asloss_backward needs
 * (the output-layer backward runs inside the same launch as the output layer).                                                    */
int dpd_asloss_forward(const dpd_asloss* e, const float* pcA, const float* pcB, int want_grad, float* loss, void* stream);
/* gradient of upstream * loss w.r.t. pcA / pcB -> gA, gB [B,N,3]; upstream = DEVICE scalar (NULL = 1).  Needs the state of the last
 * dpd_asloss_forward(want_grad = 1) on this engine; may be called more than once for it.                                           */
int dpd_asloss_backward(const dpd_asloss* e, const float* upstream, float* gA, float* gB, void* stream);
/* both directions with upstream = 1 (callers without an autograd engine in between) */
int dpd_asloss_forward_backward(const dpd_asloss* e, const float* pcA, const float* pcB, float* loss, float* gA, float* gB, void* stream);

/* Host utility: CRC32C (Castagnoli, reflected, init/xorout ~0) of n bytes continuing from `crc` (0 to start); used
 * by the TensorFlow-checkpoint interchange of dpdist_amd/tf_checkpoint.py.  No device work.                 */
uint32_t dpd_crc32c(const void* data, size_t n, uint32_t crc);

/* ---------------------------------------------------------------------------------------------
 * Chamfer distance (baseline loss of the AUE task).  Replaces pairwise_diff + chmafer_dist
 * (train_multi_gpu_pc_compare_dist.py:891-916): a [B,N,3] (pc), b [B,M,3] (rec_pc),
 *   loss[0] = ( mean_bi min_j |b_bi - a_bj|^2 + mean_bj min_i |a_bj - b_bi|^2 ) / 2     (squared distances).
 * min_a/arg_a [B,N], min_b/arg_b [B,M] receive the per-point minima and their indices (kept for the backward).
 * dpd_chamfer_bwd: da, db (either may be NULL) = gscale * d loss / d a, d loss / d b (overwritten).        */
int dpd_chamfer_fwd(const float* a, const float* b, int B, int N, int M, float* min_a, int32_t* arg_a, float* min_b,
                    int32_t* arg_b, float* loss, void* stream);
int dpd_chamfer_bwd(const float* a, const float* b, int B, int N, int M, const int32_t* arg_a, const int32_t* arg_b,
                    float gscale, float* da, float* db, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pose algebra of the iterative registration that consumes DPDist as its loss (row f2; csrc/pose.hip): ONE launch for the chain of
 * ~115 element-wise ops the reference runs per refinement loop.  pred [B,7] = the pose network's raw output (t, angle, axis).
 *   pose   [B,7]   (optional) quat_normalize(pred): (tanh(t) 0.1, cos(a/2), axis sin(a/2)), |a| <= lim_rot_deg
 *                  (pcrnet-registration/models/ipcr_model.py:285-294); lim_rot_deg == 0: pose = pred
 *   moved  [B,N,3] (optional) src R(u)^T + t (helper.py:539-570), u = q / max(|q|, 1e-12) in mode 0 (the forward-only refinements,
 *                  helper.py:309-329) and q / (|q| + 1e-7) in mode 1 (the training evaluation, iterative_PCRNet_ours.py:211-224)
 *   T_out  [B,4,4] (optional) [R(q / max(|q|, 1e-12)) t; 0 1] @ T_in (helper.py:309-329); T_in NULL = the identity; must not alias T_in
 * dpd_pose_apply_bwd: d moved [B,N,3] of a mode-1 forward -> dpred [B,7] (overwritten); src carries no gradient (the refinements
 * are forward-only, iterative_PCRNet_ours.py:414-441).                                                                            */
int dpd_pose_apply_fwd(const float* pred, const float* src, const float* T_in, int B, int N, float lim_rot_deg, int mode, float* pose,
                       float* moved, float* T_out, void* stream);
int dpd_pose_apply_bwd(const float* pred, const float* src, const float* dmoved, int B, int N, float lim_rot_deg, float* dpred,
                       void* stream);

/* The forward-only pose refinements of a registration step (iterative_PCRNet_ours.py:414-441: 7 per training step, 8 per evaluation batch;
 * only `predicted_transformation` is fetched) with the pose NETWORK on the library as well: per loop, shared MLP 3-64-64-64-128-out_features +
 * max pool over the points for source and template (models/ipcr_model.py:198-233; the template's features are computed once: it never
 * moves), fc 2*out_features-1024-512-256, dropout, fc 7 (:273-284), quat_normalize (:285-294), then the source is moved and T composed
 * (helper.py:309-329) -- four launches per loop + two per call (a loop's pose chain and cloud move are the prologue of the next loop's shared MLP; the
 * template's half of the first dense layer's sum is computed once).  Weights are torch.nn.Linear layout: W [out, in] row-major, fp32.
 *   src, tmpl [B,N,3];  drop_mask [loops,B,256] or NULL: multiplied into the 256-wide layer (0 or 1/keep_prob: the caller draws it);
 *   ws: dpd_pose_refine_workspace_bytes(B, N, out_features) bytes, 16-byte aligned;
 *   moved [B,N,3], T_out [B,4,4]: the source and the accumulated transform after `loops` loops (T starts at the identity);
 *   pred_out [loops,B,7] (optional): the network's raw output of every loop.
 * out_features must be 1024, the reference's width (DPD_E_UNSUPPORTED otherwise).                                                        */
typedef struct dpd_pose_net {
    const float* Wp[5]; const float* bp[5];   /* shared MLP: 64x3, 64x64, 64x64, 128x64, out_features x 128 */
    const float* Wh[4]; const float* bh[4];   /* head: 1024 x 2*out_features, 512x1024, 256x512, 7x256 */
    int out_features;
} dpd_pose_net;
size_t dpd_pose_refine_workspace_bytes(int B, int N, int out_features);
int dpd_pose_refine(const dpd_pose_net* net, const float* src, const float* tmpl, int B, int N, int loops, float lim_rot_deg,
                    const float* drop_mask, void* ws, size_t ws_bytes, float* moved, float* T_out, float* pred_out, void* stream);

/* The TRAINING evaluation of the pose network's shared MLP + max pool (models/ipcr_model.py:198-233 inside the step of
 * pcrnet-registration/iterative_PCRNet_ours.py:442-470, which differentiates the network w.r.t. its weights only): forward with what the
 * backward needs, and TF / torch autodiff of the five 1x1 convolutions, their ReLUs and tf.reduce_max.
 *   dpd_pose_point_fwd_train: clouds ptsA [nA,N,3] then ptsB [nB,N,3] (nB may be 0) -> f [nA+nB, out_features]; stored for the backward:
 *     h1, h2, h3 [(nA+nB) N, 64], h4 [(nA+nB) N, 128] (16-byte aligned) and ties [nA+nB, out_features] x 8 bytes: bit p set <=> point p
 *     attains the column's maximum and that maximum is positive (the gradient of reduce_max is shared evenly among ties).  N <= 64.
 *   dpd_pose_point_bwd: df [nA+nB, out_features] -> dW[i], db[i] (i = 0..4, shapes of net->Wp / bp; overwritten).  ws:
 *     dpd_pose_point_bwd_workspace_bytes(nA + nB) bytes.  Deterministic (every sum in a fixed order).  The head's fields of `net` are not read. */
size_t dpd_pose_point_bwd_workspace_bytes(int clouds);
int dpd_pose_point_fwd_train(const dpd_pose_net* net, const float* ptsA, const float* ptsB, int nA, int nB, int N, float* f, float* h1, float* h2,
                             float* h3, float* h4, unsigned long long* ties, void* stream);
int dpd_pose_point_bwd(const dpd_pose_net* net, const float* ptsA, const float* ptsB, int nA, int nB, int N, const float* df, const float* h1,
                       const float* h2, const float* h3, const float* h4, const unsigned long long* ties, float* const* dW, float* const* db,
                       void* ws, size_t ws_bytes, void* stream);

/* The head of the same training evaluation (models/ipcr_model.py:273-284: fc 2 x out_features -> 1024 -> 512 -> 256, ReLU each, dropout on the
 * last, -> 7) and its autodiff.  f [2B, out_features] = the pooled features (rows < B: first cloud set, rows >= B: second; the head reads
 * their concatenation without materialising it); drop_mask [B,256] (0 or 1/keep) or NULL.
 *   dpd_pose_head_fwd_train: -> h1 [B,1024], h2 [B,512], h3 [B,256] (after the mask), pred [B,7] (the network's raw output).
 *   dpd_pose_head_bwd: dpred [B,7] -> dW[i], db[i] (i = 0..3, shapes of net->Wh / bh; overwritten) and df [2B, out_features];
 *     ws: dpd_pose_head_bwd_workspace_bytes(B).  Deterministic.  The shared MLP's fields of `net` are not read.                               */
size_t dpd_pose_head_bwd_workspace_bytes(int B);
int dpd_pose_head_fwd_train(const dpd_pose_net* net, const float* f, int B, const float* drop_mask, float* h1, float* h2, float* h3, float* pred,
                            void* stream);
int dpd_pose_head_bwd(const dpd_pose_net* net, const float* f, int B, const float* drop_mask, const float* h1, const float* h2, const float* h3,
                      const float* dpred, float* const* dW, float* const* db, float* df, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * tf.train.AdamOptimizer step (epsilon-hat form), train_multi_gpu_pc_compare_dist.py:216,301:
 *   g' = g * gscale;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;  p -= lr_t m / (sqrt(v)+eps)
 * with lr_t = lr sqrt(1-b2^t)/(1-b1^t) computed by the caller.  n elements (any n).              */
int dpd_adam_tf(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2,
                float eps, float gscale, void* stream);

/* One-launch optimizer step (same arithmetic, element for element, as dpd_adam_tf):
 *   WT[i] != NULL (or planes, below): the matrix p[w_off[i] ..] of shape [w_rows[i], w_cols[i]] (row-major; cols % 64 == 0, rows % 4 == 0, offsets
 *     ascending) is updated tile-wise and its transposed copy WT[i] [w_cols[i], w_rows[i]] is rewritten in the same pass
 *     (replaces dpd_weights_transpose after the step);
 *   partials != NULL: the LAST 4H+3 elements p[tail_off .. tail_off+4H+3) = [b3 | W4 | b4] (followed by at most 3 zero padding
 *     elements up to n) take their gradient from the block partials
 *     that dpd_decoder_bwd_data(phases | 16) left in dpd_small_grads.partials (nparts = ceil(Qb / 8) records of `rec`
 *     floats); the reduced gradient is stored to g, and when `loss` is given (rec >= 4H+8: fused L1 loss) loss[0..1] are
 *     finished exactly as the deferred reduction would (single-GPU steps only: a data-parallel step needs the reduced
 *     gradient before its all-reduce).                                                                                  */
typedef struct dpd_adam_fuse {
    float* WT[3];
    long w_off[3];
    int w_rows[3], w_cols[3];
    /* bf16-matrix-core compute types: the operand planes of the updated matrices (dpd_planes.W*_rc / W*_r8; either may be NULL),
     * np = 1 or 3 planes each, written in the same pass (replaces dpd_weights_to_planes after the step; rows % 8 == 0);
     * np = 0: none */
    void* W_rc[3];
    void* W_r8[3];
    int np;
    const float* partials;
    int nparts, rec, H, Qb;
    long tail_off;
    float* loss;
} dpd_adam_fuse;
int dpd_adam_tf_fused(float* p, float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2, float eps, float gscale,
                      const dpd_adam_fuse* fuse, void* stream);

/* The same update with lr_t read from DEVICE memory (state[3] of an 8-float caller-owned buffer), so that a captured (hipGraph) step carries
 * no per-step host parameter: the host writes state[3] before every replay (dpdist_amd/optim.py: TFAdam.prepare_replay).  */
int dpd_adam_tf_dev(float* p, const float* g, float* m, float* v, size_t n, const float* state, float b1, float b2, float eps,
                    float gscale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Building block, exported for tests and roofline measurement: C = epi(op(A) op(B)), fp32 MFMA.
 *   transA = 0: A is [M,K] (lda);  1: A is stored [K,M].   transB = 0: B is [K,N];  1: B is [N,K].
 *   epilogue: 0 none, 1 +bias[n], 2 relu(+bias[n]), 3 multiply by (gate[m,n] > 0) (ldg = ldc).
 *   K, N, lda, ldb, ldc multiples of 4; split_k >= 1 (slabs in ws, reduced by a second kernel,
 *   epilogue applied after the reduction); tile: 0 = auto (= 3); 3 = register-staged 64x64 (any K % 4 == 0); LDS-DMA ring
 *   kernels (K % 32 == 0, else 3 is used) 8 = 64x64/3-stage, 9 = 128x128/3-stage; register-streamed kernels (no LDS, no
 *   barriers; K % 32 == 0; csrc/gemm_rs.h): workgroup of 4 waves, wave tile 30 = 64x64, 31 = 64x32, 32 = 32x64, 33 = 32x32. */
int dpd_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                 float* Cout, int ldc, const float* bias, const float* gate, int epilogue, int split_k, int tile,
                 void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * bf16-plane building blocks (gemm_x3.hip).  A plane tensor holds np (1 or 3) bf16 planes of an fp32 matrix in
 * 16-byte chunks of 8 contraction elements:
 *   RC layout (contraction index = column):  plane[r][c],           leading dimension ld_rc (elements)
 *   R8 layout (contraction index = row):     plane[r/8][c][r%8]
 * dpd_split_planes writes either or both (rc / r8 may be NULL) from src [R,C] (row stride ld; R, C multiples of 8).
 * dpd_gemm_planes: C [M,N] fp32 = epi(A B), A/B given as planes; a_fmt/b_fmt: 0 = RC (A stored [M,K], B stored
 * [N,K]), 1 = R8 (A stored [K,M], B stored [K,N]); lda/ldb = row stride for RC, M resp. N for R8; K % 32 == 0;
 * epilogue/bias/gate as in dpd_gemm_f32; tile 0 = default.  out_rc / out_r8 (may be NULL): the result is ALSO
 * written as np planes, RC [np][M][N] and/or R8 [np][r8_rows/8][N][8] (rows < r8_rows), ready to be an operand of
 * the next GEMM; C may then be NULL.                                                                       */
int dpd_split_planes(const float* src, int R, int C, int ld, int np, void* rc, int ld_rc, long rc_plane, void* r8,
                     long r8_plane, void* stream);
int dpd_gemm_planes(int np, int a_fmt, int b_fmt, int M, int N, int K, const void* A, int lda, long a_plane,
                    const void* B, int ldb, long b_plane, float* C, int ldc, const float* bias, const float* gate,
                    int epilogue, int tile, void* out_rc, void* out_r8, int r8_rows, void* stream);

/* Tuning knob (process-wide, never needed for correctness): GEMM tile / split-K per
 * call site.  op: 0 fwd layer 1, 1 fwd layers 2-3, 2 bwd dH, 3 bwd dX, 4 bwd dW1, 5 bwd dW2/3, 6 / 7 bwd dH / dX with a
 * transposed weight copy; 16 + op: one-plane (bf16) tile of gemm_x3.hip for that call site (0 = automatic; 1-5 ring kernels, 13 =
 * 192x128 at BK 64, 21 / 23 / 24 phase-staggered), 32: its grouped dW2/dW3 launch, 33: the grouped dW1/dW2/dW3 launch of
 * dpd_decoder_bwd_weights_trio.  tile as in dpd_gemm_f32 (0 = auto); split_k applies to ops 4, 5 and, for the plane weight gradients
 * (ops 20, 21, 32, 33): n > 1 = n K slices per tile reduced inside the launch (last-arriving slice, slice order: deterministic),
 * n < -1 = |n| fp32 slabs + a reduce launch, 1 = off, 0 = automatic.
 * Defaults are the measured best.                                                                                        */
int dpd_set_gemm_plan(int op, int tile, int split_k);

/* dW1, dW2 and dW3 of a plane compute type (dtype 1 / 2) in ONE grouped launch; all operands are the R8 planes of `pl`
 * (X / g1, h1 / g2, h2 / g3).  Same gradients as dpd_decoder_bwd_weights(1) + dpd_decoder_bwd_weights_pair up to fp32 summation
 * order; no bias gradients (the plane compute types get them from the dH epilogues of dpd_decoder_bwd_data).  DPD_E_UNSUPPORTED
 * when a plane is missing or a shape is not plane-shaped: the caller then makes the separate calls.
 * Replaces TF autodiff of the three tf.nn.conv2d kernels (utils/tf_util.py:213, train_multi_gpu_pc_compare_dist.py:274-277).   */
int dpd_decoder_bwd_weights_trio(int Qb, int KP, int H, int dtype, float* dW1, float* dW2, float* dW3, void* ws, size_t ws_bytes,
                                 const dpd_planes* pl, void* stream);

/* Opt-in profiler for the roofline measurement (bench.py): when enabled, every GEMM kernel launch is bracketed
 * by a hipEvent pair on its own stream.  dpd_prof_collect waits for them and returns the launch count and the
 * summed durations [ms] / flops (2*M*N*K as launched) since dpd_prof_enable(1).                              */
int dpd_prof_enable(int on);
int dpd_prof_collect(double* total_ms, double* total_flops);
/* the same for ONE product form: 0 = NN / NT (forward layers and data gradients -- in DPD_F32 one kernel, the step's dominant one),
 * 1 = TN (weight gradients)                                                                                               */
int dpd_prof_collect_form(int form, double* total_ms, double* total_flops);
/* dpd_prof_enable(2) additionally brackets the bandwidth-bound kernels of the step and records their ALGORITHMIC HBM bytes (every
 * input read once, every output written once) by stage: 1 3DmFV encoder, 2 window gather, 3 fused output layer (+ loss + its backward),
 * 4 optimizer, 5 small-gradient reduction, 6 weight copies (transposes / operand planes).  Returns the launches of that stage since
 * dpd_prof_enable(2) and fills their summed duration [ms] and bytes; dpd_prof_collect keeps counting GEMM launches only.          */
int dpd_prof_collect_stage(int stage, double* total_ms, double* total_bytes);

#ifdef __cplusplus
}
#endif
#endif /* DPDIST_CAPI_H */
