"""Allocation-free DPDist training step on the HIP C ABI (the reference's `sess.run(train_op_s)` hot loop).

One `step(pcA, pcB, labels, noise)` = what `train_multi_gpu_pc_compare_dist.py:786` executes per batch:
    get_model (encoder x2, window gather, decoder both directions)         models/dpdist_and_aue.py:31-86
    get_loss  (loss_samples = mean |pred_AB[...,0] - labels|)              utils/dpdist_util.py:962-980
    compute_gradients(loss_samples, the 8 'pc_compare' variables)         train_multi_gpu...:274-277
    average over towers -> RCCL all-reduce (ddp.py)                        :936-974
    AdamOptimizer.apply_gradients with the staircase learning rate        :216,301,976-990
Only the AB half of the rows carries gradient (loss_samples reads pred_listAB only), so the backward GEMMs run on
B*N rows while the forward runs on 2*B*N.

Input pipeline: `step(..., prefetch=(pcA', pcB', noise'))` runs the NEXT batch's front end (stacking, 3DmFV encoder, window
gather -- none of which depends on the weights) on a side stream while this step's backward and Adam occupy the main
stream; the next `step` on those tensors then starts at the decoder.  Without `prefetch` everything is in stream order.

Backward schedule (chosen for overlap, not for autodiff order): dH chain g3 -> g2 -> g1, then dW1 (largest bucket,
all-reduce launched immediately), then dW2, dW3, dW4 (second bucket), then one fused Adam over the flat buffer.
"""
import math
import os

import torch

from . import lib as L
from .ddp import BucketReducer
from .model import DPDistParams


def learning_rate(step, base=1e-4, decay_step=300 * 512, decay_rate=0.5, floor=1e-7):
    """train_multi_gpu_pc_compare_dist.py:976-990: exponential_decay(base, batch, DECAY_STEP, DECAY_RATE, staircase=True), clipped."""
    return max(base * decay_rate ** math.floor(step / decay_step), floor)


class DPDistTrainer:
    def __init__(self, params: DPDistParams, batch_size, num_point=64, Embedding_Size=512, sigma3dmfv=0.125, base_lr=1e-4,
                 decay_step=300 * 512, decay_rate=0.5, beta1=0.9, beta2=0.999, eps=1e-8, group=None, distributed=None,
                 compute_dtype=None, dedupe=None):
        self.P = params
        self.dt = L.DTYPES[params.compute_dtype if compute_dtype is None else compute_dtype]
        dev = params.flat.device
        self.B, self.N = int(batch_size), int(num_point)
        self.m = int(math.ceil(Embedding_Size ** (1 / 3) - 1e-9))
        self.k, self.sigma = params.k, float(sigma3dmfv)
        self.hp = (base_lr, decay_step, decay_rate, beta1, beta2, eps)
        self.t = 0
        B, N, H, KP = self.B, self.N, params.H, params.KP
        C, Q, BN = 2 * B, 2 * B * N, B * N
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)   # noqa: E731
        self.pts, self.q = f(C, N, 3), f(C, N, 3)
        # Exact-fp32 compute type, opt-in (DPD_FUSED_GATHER=1): the window gather fused into the layer-1 GEMMs (csrc/gemm_rs.h,
        # SURVEY K2) -- X [Q,KP] (41 MB at B = 32, 82 MB at B = 64) is never materialised; bitwise the same results.  Measured
        # SLOWER on MI355X at B = 32 (layer 1: 180 us against 147 + 13.5 us for gather kernel + plain GEMM; dW1: 152 against
        # 91 us: the per-lane address arithmetic sits in the in-order issue stream of an MFMA-bound wave), so the default
        # keeps X.  fv and the q - centre columns share ONE allocation either way (one buffer descriptor).
        G = self.m ** 3
        self.fused = (self.dt == 0 and KP % 32 == 0 and BN % 32 == 0 and not dedupe
                      and os.environ.get("DPD_FUSED_GATHER", "0") == "1" and os.environ.get("DPD_DEDUPE", "0") != "1")
        self._fvx = f(C * G * 20 + Q * 4)
        self.fv = self._fvx[:C * G * 20].view(C, G, 20)
        self.xyz = self._fvx[C * G * 20:].view(Q, 4)
        self.mask = f(Q)
        self.vox = torch.empty(Q, device=dev, dtype=torch.int32)
        if self.fused:
            self.X = None
            self.rowinfo = torch.empty(Q, 2, device=dev, dtype=torch.int32)
            self.ktab = torch.empty(KP // 4, 2, device=dev, dtype=torch.int32)
            L.check(L.load().dpd_gather_table(self.m, params.k, KP, L.ptr(self.ktab), L.cur_stream()), "dpd_gather_table")
            self._gsrc = L.Gather(self.fv.data_ptr(), self.xyz.data_ptr(), self.rowinfo.data_ptr(), self.ktab.data_ptr(), C, G)
        else:
            self.X = f(Q, KP)
        self.h1, self.h2, self.h3 = f(Q, H), f(Q, H), f(Q, H)
        self.y, self.pred = f(Q, 3), f(Q, 3)
        self.dpred, self.dy = f(BN, 3), f(BN, 3)
        self.g1, self.g2, self.g3 = f(BN, H), f(BN, H), f(BN, H)
        self.loss = f(2)
        self.grad = torch.zeros(params.numel, device=dev, dtype=torch.float32)
        self.m_state = torch.zeros_like(self.grad)
        self.v_state = torch.zeros_like(self.grad)
        self.ws = torch.empty((L.load().dpd_workspace_bytes(Q, KP, H, self.dt) + 3) // 4, device=dev, dtype=torch.float32)
        # Layer 1 on the UNIQUE (cloud, voxel) rows (csrc/dedupe.hip): exact-fp32 compute type only.  Experimental and off by
        # default (slower at B = 32 until the layer-1 GEMM gets a stream-K decomposition, see the header of dedupe.hip)
        if dedupe is None:
            dedupe = os.environ.get("DPD_DEDUPE", "0") == "1"
        self.dedupe = bool(dedupe) and self.dt == 0 and KP % 32 == 0 and Q * 3 * 4 <= 150 * 1024 and BN >= 32
        if self.dedupe:
            i32 = lambda n: torch.empty(n, device=dev, dtype=torch.int32)   # noqa: E731
            self.u_of_q, self.rep_q, self.counts = i32(Q), i32(Q), torch.zeros(4, device=dev, dtype=torch.int32)
            self.xyz, self.T = f(Q, 3), f(Q, H)
            self.ws1 = torch.empty(L.load().dpd_layer1_bwd_unique_workspace_bytes(BN, KP, H) // 4 + 1, device=dev,
                                   dtype=torch.float32)
        # bf16-matrix-core compute types: operand planes persist between the kernels (no conversion passes)
        self._planes = None
        if self.dt and Q % 8 == 0 and BN % 32 == 0 and KP % 32 == 0:
            lib = L.load()
            nbytes = lib.dpd_planes_bytes(Q, BN, KP, H, self.dt, 0)
            self._plane_mem = torch.empty(nbytes, device=dev, dtype=torch.uint8)
            self._planes = L.Planes()
            L.check(lib.dpd_planes_carve(L.ptr(self._plane_mem), nbytes, Q, BN, KP, H, self.dt, 0, self._planes), "dpd_planes_carve")
        import torch.distributed as dist
        use_dist = dist.is_initialized() if distributed is None else distributed
        self.reducer = BucketReducer(self.grad, params.bucket_bounds, group,
                                     force=os.environ.get("DPD_FORCE_DIST") == "1") if use_dist else None
        # exact-fp32 compute type: transposed copies of W2 / W3 so that the backward data GEMMs (g W^T) read the weights
        # row-coalesced (register-streamed kernel, csrc/gemm_rs.h); refreshed after every optimizer step
        self.W2T = self.W3T = None
        if self.dt == 0:
            self.W2T, self.W3T = f(H, H), f(H, H)
        self._cparams = L.make_params(*params.views(), self.W2T, self.W3T, None)
        self._gviews = params.views(self.grad)
        gv = self._gviews
        self._csmall = L.make_small_grads(gv[1], gv[3], gv[5], gv[6], gv[7])
        self._after_dw1 = None
        self._side = None          # side stream of the prefetch pipeline (created on first use)
        self._pref_key = None      # identity of the batch whose front end is (being) computed on the side stream
        self._ev_front = self._ev_xfree = self._ev_fwd = None
        self.front_launches = 0    # front ends (stack + encoder + gather) enqueued so far, on either stream
        self.prefetch_hits = 0     # steps that found their front end already computed by the side stream
        self.refresh_weight_planes()

    def refresh_weight_planes(self):
        """Call after changing the weights from outside (load_tf_state_dict): re-derives the bf16 weight planes / the
        transposed fp32 copies."""
        if self._planes is not None:
            L.check(L.load().dpd_weights_to_planes(self._cparams, self.P.KP, self.P.H, self._planes, L.cur_stream()),
                    "dpd_weights_to_planes")
        if self.W2T is not None:
            L.check(L.load().dpd_weights_transpose(self._cparams, self.P.KP, self.P.H, L.ptr(self.W2T), L.ptr(self.W3T), None,
                                                   L.cur_stream()), "dpd_weights_transpose")

    # -- pieces (each enqueues kernels on the current stream; no host sync, no allocation) -----------------
    @staticmethod
    def _key(pcA, pcB, noise):
        return (pcA.data_ptr(), pcB.data_ptr(), None if noise is None else noise.data_ptr(), pcA._version, pcB._version)

    def _front(self, pcA, pcB, noise, gate=None):
        """stacking + encoder + window gather of one batch on the CURRENT stream; `gate`: event to wait for before the
        gather overwrites X / mask / vox (their last reader of the previous step)."""
        self.front_launches += 1
        self._load_batch(pcA, pcB, noise)
        self._encode()
        if gate is not None:
            torch.cuda.current_stream().wait_event(gate)
        self._gather()

    def _load_batch(self, pcA, pcB, noise):
        shp = (self.B, self.N, 3)
        if self.fused:      # stacking + query lookup in one launch (the gather itself happens inside the layer-1 GEMMs)
            L.check(L.load().dpd_front(L.ptr(L.req(pcA, name="pcA", shape=shp)), L.ptr(L.req(pcB, name="pcB", shape=shp)),
                                       None if noise is None else L.ptr(L.req(noise, name="add_noise", shape=shp)), self.B, self.N,
                                       self.m, self.k, L.ptr(self.pts), None, L.ptr(self.mask), L.ptr(self.vox), L.ptr(self.xyz),
                                       L.ptr(self.rowinfo), L.cur_stream()), "dpd_front")
            return
        L.check(L.load().dpd_stack_clouds(L.ptr(L.req(pcA, name="pcA", shape=shp)), L.ptr(L.req(pcB, name="pcB", shape=shp)),
                                          None if noise is None else L.ptr(L.req(noise, name="add_noise", shape=shp)), self.B, self.N,
                                          L.ptr(self.pts), L.ptr(self.q), L.cur_stream()), "dpd_stack_clouds")

    def _encode(self):
        L.check(L.load().dpd_mfv3d_fwd(L.ptr(self.pts), 2 * self.B, self.N, self.m, self.sigma, L.ptr(self.fv), L.cur_stream()),
                "dpd_mfv3d_fwd")

    def _gather(self):
        if self.fused:
            return
        lib, s, P = L.load(), L.cur_stream(), self.P
        C, N = 2 * self.B, self.N
        if self.dedupe:     # voxel lookup + unique-row numbering, then the window gather of the unique rows only
            L.check(lib.dpd_dedupe_rows(L.ptr(self.q), C, N, self.m, self.B * N, L.ptr(self.u_of_q), L.ptr(self.rep_q),
                                        L.ptr(self.counts), L.ptr(self.xyz), L.ptr(self.mask), L.ptr(self.vox), s), "dpd_dedupe_rows")
            L.check(lib.dpd_patch_rows_fwd_unique(L.ptr(self.fv), C, N, self.m, self.k, P.KP, L.ptr(self.rep_q), L.ptr(self.vox),
                                                  L.ptr(self.counts), L.ptr(self.X), s), "dpd_patch_rows_fwd_unique")
            return
        L.check(lib.dpd_patch_rows_fwd(L.ptr(self.q), L.ptr(self.fv), C, N, self.m, self.k, P.KP,
                                       None if self._planes is not None else L.ptr(self.X), L.ptr(self.mask), L.ptr(self.vox),
                                       self._planes, s), "dpd_patch_rows_fwd")

    def forward(self):
        """encoder + gather + decoder of the batch loaded by _load_batch (stream order)."""
        self._encode()
        self._gather()
        self._decode()

    def _decode(self):
        lib, s, P = L.load(), L.cur_stream(), self.P
        Q = 2 * self.B * self.N
        if self.fused:
            L.check(lib.dpd_decoder_fwd_gather(self._gsrc, L.ptr(self.mask), Q, P.KP, P.H, self._cparams, L.ptr(self.h1),
                                               L.ptr(self.h2), L.ptr(self.h3), L.ptr(self.y), L.ptr(self.pred), s),
                    "dpd_decoder_fwd_gather")
            return
        if self.dedupe:     # layer 1 on the unique rows, expanded to h1; the decoder entry point then starts at layer 2
            v = P.views()
            L.check(lib.dpd_layer1_fwd_unique(L.ptr(self.X), L.ptr(self.counts), L.ptr(self.u_of_q), L.ptr(self.xyz), Q, P.KP, P.H,
                                              P.E, L.ptr(v[0]), L.ptr(v[1]), L.ptr(self.T), L.ptr(self.h1), s), "dpd_layer1_fwd_unique")
            L.check(lib.dpd_decoder_fwd(None, L.ptr(self.mask), Q, P.KP, P.H, self._cparams, self.dt, L.ptr(self.h1),
                                        L.ptr(self.h2), L.ptr(self.h3), L.ptr(self.y), L.ptr(self.pred), L.ptr(self.ws),
                                        self.ws.numel() * 4, None, s), "dpd_decoder_fwd")
            return
        L.check(lib.dpd_decoder_fwd(L.ptr(self.X), L.ptr(self.mask), Q, P.KP, P.H, self._cparams, self.dt, L.ptr(self.h1),
                                    L.ptr(self.h2), L.ptr(self.h3), L.ptr(self.y), L.ptr(self.pred), L.ptr(self.ws),
                                    self.ws.numel() * 4, self._planes, s), "dpd_decoder_fwd")

    def backward(self, labels):
        lib, s, P = L.load(), L.cur_stream(), self.P
        BN = self.B * self.N
        L.req(labels, name="labels", numel=BN)
        L.check(lib.dpd_l1_loss(L.ptr(self.pred), L.ptr(labels), BN, 1, 1.0, L.ptr(self.loss), L.ptr(self.dpred), s), "dpd_l1_loss")
        d, wsb = self._gviews, self.ws.numel() * 4

        def data(phases):   # db1..db3, dW4, db4 fall out of the data chain (fused epilogues / one small kernel)
            L.check(lib.dpd_decoder_bwd_data(L.ptr(self.dpred), L.ptr(self.mask), L.ptr(self.y), L.ptr(self.h1), L.ptr(self.h2),
                                             L.ptr(self.h3), BN, P.KP, P.H, self._cparams, self.dt, L.ptr(self.dy), L.ptr(self.g3),
                                             L.ptr(self.g2), L.ptr(self.g1), None, self._csmall, L.ptr(self.ws), wsb, self._planes,
                                             phases, s), "dpd_decoder_bwd_data")

        def dw(layer, act, g, dW):
            if layer == 1 and self.fused:
                L.check(lib.dpd_decoder_bwd_weights_gather(self._gsrc, L.ptr(self.g1), BN, P.KP, P.H, L.ptr(dW), L.ptr(self.ws), wsb, L.cur_stream()),
                        "dpd_decoder_bwd_weights_gather")
                return
            if layer == 1 and self.dedupe:
                L.check(lib.dpd_layer1_bwd_weights_unique(L.ptr(self.X), L.ptr(self.g1), L.ptr(self.u_of_q), L.ptr(self.rep_q),
                                                          L.ptr(self.xyz), L.ptr(self.counts), self.N, BN, P.KP, P.H, P.E, L.ptr(dW),
                                                          L.ptr(self.ws1), self.ws1.numel() * 4, L.cur_stream()),
                        "dpd_layer1_bwd_weights_unique")
                return
            L.check(lib.dpd_decoder_bwd_weights(layer, L.ptr(act), act.stride(0), L.ptr(g), BN, dW.shape[0], dW.shape[1], self.dt,
                                                L.ptr(dW), None, L.ptr(self.ws), wsb, self._planes, L.cur_stream()),
                    "dpd_decoder_bwd_weights(%d)" % layer)

        if self.reducer and os.environ.get("DPD_DP_SCHEDULE", "early") == "early":
            # Data-parallel schedule: every weight gradient is produced as early as its inputs exist, smallest bucket first,
            # so that the all-reduces (serial on the RCCL stream) start ~250 us before the backward ends instead of after dW1:
            #   output layer -> dW3 -> [bucket 2: W3,b3,W4,b4] -> g2 -> dW2 -> [bucket 1: W2,b2] -> g1 -> dW1 -> [bucket 0]
            data(1)
            dw(3, self.h2, self.g3, d[4])
            self.reducer.reduce_async(2)
            data(2)
            dw(2, self.h1, self.g2, d[2])
            self.reducer.reduce_async(1)
            data(4)
            dw(1, self.X, self.g1, d[0])
            if self._after_dw1 is not None:
                self._after_dw1()
            self.reducer.reduce_async(0)
            return
        data(7)
        dw(1, self.X, self.g1, d[0])
        if self._after_dw1 is not None:
            self._after_dw1()             # X / mask are free from here on: the prefetch pipeline hooks in
        if self.reducer:
            self.reducer.reduce_async(0)
        if BN % 32 == 0:      # layers 2 and 3 have identical shapes: one grouped launch
            L.check(lib.dpd_decoder_bwd_weights_pair(L.ptr(self.h1), L.ptr(self.g2), L.ptr(d[2]), L.ptr(self.h2), L.ptr(self.g3),
                                                     L.ptr(d[4]), P.H, BN, P.H, P.H, self.dt, L.ptr(self.ws), wsb, self._planes, L.cur_stream()),
                    "dpd_decoder_bwd_weights_pair")
        else:
            dw(2, self.h1, self.g2, d[2])
            dw(3, self.h2, self.g3, d[4])
        if self.reducer:      # DPD_DP_SCHEDULE=late (A/B reference): plain order, all-reduces start after dW1
            self.reducer.reduce_async(1)
            self.reducer.reduce_async(2)

    def apply_gradients(self):
        base_lr, decay_step, decay_rate, b1, b2, eps = self.hp
        lr = learning_rate(self.t, base_lr, decay_step, decay_rate)     # global_step before the increment (TF semantics)
        self.t += 1
        lr_t = lr * math.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
        gscale = 1.0
        if self.reducer:
            self.reducer.wait()
            gscale = self.reducer.grad_scale
        L.check(L.load().dpd_adam_tf(L.ptr(self.P.flat), L.ptr(self.grad), L.ptr(self.m_state), L.ptr(self.v_state),
                                     self.P.numel, lr_t, b1, b2, eps, gscale, L.cur_stream()), "dpd_adam_tf")
        self.refresh_weight_planes()

    def _take_front(self, pcA, pcB, noise):
        """Make the front-end buffers (pts, q, fv, X, mask, vox) hold this batch on the current stream."""
        main = torch.cuda.current_stream()
        if self._pref_key is not None:
            main.wait_event(self._ev_front)            # whatever the side stream was doing with the buffers is ordered first
            hit = self._pref_key == self._key(pcA, pcB, noise)
            self._pref_key = None
            if hit:
                self.prefetch_hits += 1
                return
        self._front(pcA, pcB, noise)

    @torch.no_grad()
    def step(self, pcA, pcB, labels, noise=None, prefetch=None):
        """One training step.  Returns the device tensor [loss_samples, loss_pred] of THIS rank's shard (no host sync).
        prefetch = (pcA', pcB', noise' or None): the NEXT step's inputs (already complete on the current stream); their
        front end runs on a side stream under this step's backward and optimizer."""
        self._take_front(pcA, pcB, noise)
        self._decode()
        if prefetch is not None:
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.P.flat.device)
                self._ev_front, self._ev_xfree, self._ev_fwd = (torch.cuda.Event() for _ in range(3))
            main = torch.cuda.current_stream()
            self._ev_fwd.record(main)                  # inputs complete + this step's gather/decoder ordered before the side work

            def launch_front():
                self._ev_xfree.record(main)            # recorded right after dW1, the last reader of X / mask
                with torch.cuda.stream(self._side):
                    self._side.wait_event(self._ev_fwd)
                    self._front(*prefetch, gate=self._ev_xfree)
                    self._ev_front.record(self._side)
                self._pref_key = self._key(*prefetch)
            self._after_dw1 = launch_front
        try:
            self.backward(labels.reshape(-1))
        finally:
            self._after_dw1 = None
        self.apply_gradients()
        return self.loss

    @torch.no_grad()
    def evaluate(self, pcA, pcB, labels, noise=None):
        """Forward only (eval_one_epoch_3d, train_multi_gpu...:809-873): returns [loss_samples, loss_pred] and pred_AB[...,0]."""
        self._take_front(pcA, pcB, noise)
        self._decode()
        BN = self.B * self.N
        L.req(labels, name="labels", numel=BN)
        L.check(L.load().dpd_l1_loss(L.ptr(self.pred), L.ptr(labels.reshape(-1)), BN, 0, 1.0, L.ptr(self.loss), None,
                                     L.cur_stream()), "dpd_l1_loss")
        return self.loss, self.pred[:BN, 0].view(self.B, self.N)
