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

import numpy as np
import torch

from . import lib as L
from .ddp import make_reducer
from .model import DPDistParams


def learning_rate(step, base=1e-4, decay_step=300 * 512, decay_rate=0.5, floor=1e-7):
    """train_multi_gpu_pc_compare_dist.py:976-990: exponential_decay(base, batch, DECAY_STEP, DECAY_RATE, staircase=True), clipped."""
    return max(base * decay_rate ** math.floor(step / decay_step), floor)


class DPDistTrainer:
    # stage forms (all on = the measured best; each off = the form it replaced, same results up to the stated tolerance of its test)
    OPTIONS = {
        "fuse_loss": True,     # training loss inside the output-layer backward (off: a separate dpd_l1_loss launch)
        "fuse_out": True,      # ... which also runs the output layer's forward (off: out_fwd launch in the decoder forward)
        "fused_adam": True,    # one optimizer launch incl. the weights' derived copies and the small-gradient reduction (off: three)
        "front2": True,        # front end in two launches (stack + encoder, norm + gather) instead of four
        "det_db": True,        # exact fp32: db1 / db2 as deterministic by-products of the dW GEMMs (off: atomics in the dH epilogues)
        "h3_plane": True,      # bf16: layer 3's activation as ONE bf16 plane (off: fp32 h3)
        "keep_f32_h": False,   # plane types: also write the fp32 h1 / h2 / g1 / g2 (on: for inspection)
        "dw_trio": None,       # plane types: dW1 + dW2 + dW3 as one grouped launch (None: on for one plane, off for three)
        "dp_buckets": 2,       # data-parallel "early" order: 2 = layers 2-4 as one collective, 3 = one per layer
    }

    def __init__(self, params: DPDistParams, batch_size, num_point=64, Embedding_Size=512, sigma3dmfv=0.125, base_lr=1e-4,
                 decay_step=300 * 512, decay_rate=0.5, beta1=0.9, beta2=0.999, eps=1e-8, group=None, distributed=None,
                 compute_dtype=None, adam_on_side=False, options=None):
        """options: overrides of OPTIONS (below) -- the unfused / older forms of single stages, kept because they are the fallbacks for
        shapes the fused forms do not take; tests run them against the defaults.  adam_on_side: see `apply_gradients`."""
        self.P = params
        unknown = set(options or {}) - set(self.OPTIONS)
        if unknown:
            raise ValueError("unknown trainer options: %s" % sorted(unknown))
        opt = dict(self.OPTIONS, **(options or {}))
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
        G = self.m ** 3
        # fv and the q - centre columns share ONE allocation (one buffer descriptor)
        self._fvx = f(C * G * 20 + Q * 4)
        self.fv = self._fvx[:C * G * 20].view(C, G, 20)
        self.xyz = self._fvx[C * G * 20:].view(Q, 4)
        self.mask = f(Q)
        self.vox = torch.empty(Q, device=dev, dtype=torch.int32)
        self.X = f(Q, KP)
        self.h1, self.h2, self.h3 = None, None, f(Q, H)      # h1 / h2: allocated below unless the planes stand in for them
        self.y, self.pred = f(Q, 3), f(Q, 3)
        self.dpred, self.dy = f(BN, 3), f(BN, 3)
        self.g1 = self.g2 = self.g3 = None                   # allocated below unless the planes stand in for them
        self.loss = f(2)
        self.grad = torch.zeros(params.numel, device=dev, dtype=torch.float32)
        self.m_state = torch.zeros_like(self.grad)
        self.v_state = torch.zeros_like(self.grad)
        self.ws = torch.empty((L.load().dpd_workspace_bytes(Q, KP, H, self.dt) + 3) // 4, device=dev, dtype=torch.float32)
        # bf16-matrix-core compute types: operand planes persist between the kernels (no conversion passes)
        self._planes = None
        # (the same predicate as the C side's usable_planes(): with H % 64 != 0 the library ignores the planes and needs the fp32
        # activations, so they must be allocated)
        if self.dt and Q % 8 == 0 and BN % 32 == 0 and KP % 32 == 0 and H % 64 == 0:
            lib = L.load()
            nbytes = lib.dpd_planes_bytes(Q, BN, KP, H, self.dt, 0)
            self._plane_mem = torch.empty(nbytes, device=dev, dtype=torch.uint8)
            self._planes = L.Planes()
            L.check(lib.dpd_planes_carve(L.ptr(self._plane_mem), nbytes, Q, BN, KP, H, self.dt, 0, self._planes), "dpd_planes_carve")
        # Plane compute types: layer 2/3, the weight gradients and the ReLU gate of the backward all read h1 / h2 from their bf16
        # planes, so the fp32 copies are not written at all (2 x 33.5 MB per forward at B = 64); options["keep_f32_h"] keeps them
        if self._planes is None or opt["keep_f32_h"]:
            self.h1, self.h2 = f(Q, H), f(Q, H)
            self.g1, self.g2 = f(BN, H), f(BN, H)
        # g3: the fused output-layer backward writes it as planes when it can (H % 256 == 0, H <= 1024, block partials fit)
        if self.g3 is None and (self.g1 is not None or not (H % 256 == 0 and H <= 1024 and (BN // 8) * (4 * H + 8) <= BN * H)):
            self.g3 = f(BN, H)
        import torch.distributed as dist
        use_dist = dist.is_initialized() if distributed is None else distributed
        self.reducer = make_reducer(self.grad, params.bucket_bounds, group,
                                    force=os.environ.get("DPD_FORCE_DIST") == "1") if use_dist else None
        # exact-fp32 compute type: transposed copies of W2 / W3 so that the backward data GEMMs (g W^T) read the weights
        # row-coalesced (register-streamed kernel, csrc/gemm_rs.h); refreshed after every optimizer step
        self.W2T = self.W3T = None
        if self.dt == 0:
            self.W2T, self.W3T = f(H, H), f(H, H)
        self._cparams = L.make_params(*params.views(), self.W2T, self.W3T, None)
        self._gviews = params.views(self.grad)
        gv = self._gviews
        self._partials = f(((BN + 7) // 8) * (4 * H + 8))      # block partials of db3 / dW4 / db4 (deferred reduction)
        self._csmall = L.make_small_grads(gv[1], gv[3], gv[5], gv[6], gv[7], self._partials)
        # training loss fused into the output-layer backward (no separate loss launch); labels pointer is filled in per step
        self.fuse_loss = H % 256 == 0 and H <= 1024 and bool(opt["fuse_loss"])
        self._db_partials = f(2 * ((BN + 31) // 32) * H)       # 32-row partial column sums of g2 / g1 (deterministic db2 / db1)
        self._last_lr = base_lr
        # one-launch optimizer (dpd_adam_tf_fused): Adam + the transposed copies + (single-GPU steps) the reduction of the
        # output layer's block partials
        self.fused_adam = bool(opt["fused_adam"])
        # training step: the output layer's forward runs inside its fused backward (no out_fwd launch)
        self.fuse_out = self.fuse_loss and bool(opt["fuse_out"])
        self._out_pending = False
        self._h3_in_plane = False      # set by _decode(skip_out=True) of a DPD_BF16 step, read by the backward that must follow it
        # front end in two launches (dpd_mfv3d_fwd_stacked + dpd_patch_rows_fwd_scaled) instead of four
        self.front2 = bool(opt["front2"])
        self._det_db_opt, self._h3_plane_opt, self._dp_buckets = bool(opt["det_db"]), bool(opt["h3_plane"]), int(opt["dp_buckets"])
        self._ssq = f(C * 4 * 20)
        self._fv_scaled = True
        seg = params._segments
        self._afuse = [L.AdamFuse(), L.AdamFuse()]           # [0]: gradients complete; [1]: tail from the block partials
        for i, af in enumerate(self._afuse):
            if self.W2T is not None:
                for j, (n, T) in enumerate((("W2", self.W2T), ("W3", self.W3T))):
                    af.WT[j], af.w_off[j], af.w_rows[j], af.w_cols[j] = T.data_ptr(), seg[n][0], H, H
            elif self._planes is not None and KP % 8 == 0 and H % 64 == 0:      # the weights' bf16 operand planes out of Adam
                pl = self._planes
                af.np = pl.np
                for j, (n, rows) in enumerate((("W1p", KP), ("W2", H), ("W3", H))):
                    af.w_off[j], af.w_rows[j], af.w_cols[j] = seg[n][0], rows, H
                    af.W_rc[j], af.W_r8[j] = getattr(pl, "W%d_rc" % (j + 1)), getattr(pl, "W%d_r8" % (j + 1))
            if i == 1:
                af.partials, af.nparts, af.rec, af.H, af.Qb = self._partials.data_ptr(), (BN + 7) // 8, 4 * H + 8, H, BN
                af.tail_off, af.loss = seg["b3"][0], self.loss.data_ptr()
        # [b3 | W4 | b4] end the flat buffer (up to 3 elements of alignment padding behind them)
        self._tail_ok = 0 <= params.numel - (seg["b3"][0] + 4 * H + 3) <= 3 and H % 256 == 0 and H <= 1024
        # one plane (bf16): dW1 + dW2 + dW3 as one grouped launch; three planes (f32x3): measured SLOWER grouped (0.557 vs 0.523 ms at B = 32:
        # its dW1 alone runs the phase-staggered 128x128 kernel, the grouped launch needs the ring kernel), so opt-in there (options["dw_trio"])
        self._trio = self._planes is not None and bool(self._planes.np == 1 if opt["dw_trio"] is None else opt["dw_trio"])
        # data-parallel steps: Adam on the collectives' stream, joined only where the weights are read next (opt-in: a caller that reads
        # params.flat right after step() must call join_optimizer(); bench.py and dpdist_amd.train switch it on)
        self.adam_on_side = bool(adam_on_side)
        self._ev_opt, self._opt_pending = None, False
        # order of a DATA-PARALLEL backward (see `backward`): "early" = every weight gradient as soon as its inputs exist, buckets
        # all-reduced under the rest of the backward; "grouped" = the single-GPU launch order (ONE grouped dW1 + dW2 + dW3 launch: 20 us
        # less GEMM time at bf16 B = 64) and ONE all-reduce behind it; "late" = plain order, collectives after dW1 (A/B reference).
        # DPD_DP_SCHEDULE pins it; otherwise `select_dp_schedule` MEASURES the candidates at start-up (all ranks together) and the
        # initial value is only what runs until then
        # the DETERMINISTIC default is "early"; measurement is opt-in per caller (bench.py; dpdist_amd.train --dp_schedule auto) or
        # DPD_DP_SCHEDULE=auto
        env_sched = os.environ.get("DPD_DP_SCHEDULE", "early")
        self.dp_schedule = "early" if env_sched == "auto" else env_sched
        self.dp_schedule_info = {"schedule": self.dp_schedule,
                                 "source": "DPD_DP_SCHEDULE" if env_sched not in ("auto",) and "DPD_DP_SCHEDULE" in os.environ else "default"}
        self._wdirty = True        # transposed copies / bf16 planes of the weights need a refresh before their next use
        self._after_dw1 = None
        self._side = None          # side stream of the prefetch pipeline (created on first use)
        self._pref_key = None      # identity of the batch whose front end is (being) computed on the side stream
        self._ev_front = self._ev_xfree = self._ev_fwd = None
        self.front_launches = 0    # front ends (stack + encoder + gather) enqueued so far, on either stream
        self.progress = None       # optional callable, called before every candidate of select_dp_schedule (bench.py: the watchdog's heartbeat)
        self.prefetch_hits = 0     # steps that found their front end already computed by the side stream

    def close(self):
        """Release what outlives Python's garbage collection badly: the RCCL communicators of the direct reducer must be destroyed
        before the process group is (bench.py, tests/rccl_rank.py and train.py call this at tear-down)."""
        self._join_optimizer()
        red, self.reducer = self.reducer, None
        if red is not None and hasattr(red, "close"):
            red.close()

    def refresh_weight_planes(self):
        """Re-derive what is computed FROM the weights (bf16 weight planes / transposed fp32 copies) on the current stream.
        The trainer does this lazily before the first kernel that needs them (`_wdirty`); call it yourself only to force it."""
        self._join_optimizer()
        self._wdirty = False
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
        if self.front2:      # two launches: (stack + encoder), (norm + gather)
            shp = (self.B, self.N, 3)
            L.check(L.load().dpd_mfv3d_fwd_stacked(L.ptr(L.req(pcA, name="pcA", shape=shp)), L.ptr(L.req(pcB, name="pcB", shape=shp)),
                                                   None if noise is None else L.ptr(L.req(noise, name="add_noise", shape=shp)),
                                                   self.B, self.N, self.m, self.sigma, L.ptr(self.pts), L.ptr(self.q), L.ptr(self.fv),
                                                   L.ptr(self._ssq), L.cur_stream()), "dpd_mfv3d_fwd_stacked")
            self._fv_scaled = False
        else:
            self._load_batch(pcA, pcB, noise)
            self._encode()
        if gate is not None:
            gate.wait()
        self._gather()

    def _load_batch(self, pcA, pcB, noise):
        shp = (self.B, self.N, 3)
        L.check(L.load().dpd_stack_clouds(L.ptr(L.req(pcA, name="pcA", shape=shp)), L.ptr(L.req(pcB, name="pcB", shape=shp)),
                                          None if noise is None else L.ptr(L.req(noise, name="add_noise", shape=shp)), self.B, self.N,
                                          L.ptr(self.pts), L.ptr(self.q), L.cur_stream()), "dpd_stack_clouds")

    def _encode(self):
        self._fv_scaled = True
        L.check(L.load().dpd_mfv3d_fwd(L.ptr(self.pts), 2 * self.B, self.N, self.m, self.sigma, L.ptr(self.fv), L.cur_stream()),
                "dpd_mfv3d_fwd")

    def _gather(self):
        lib, s, P = L.load(), L.cur_stream(), self.P
        C, N = 2 * self.B, self.N
        # fv of the two-launch front end still lacks its L2 norm: the gather applies it from the per-slice sums of squares
        L.check(lib.dpd_patch_rows_fwd_scaled(L.ptr(self.q), L.ptr(self.fv), None if self._fv_scaled else L.ptr(self._ssq), C, N,
                                              self.m, self.k, P.KP, None if self._planes is not None else L.ptr(self.X),
                                              L.ptr(self.mask), L.ptr(self.vox), self._planes, s), "dpd_patch_rows_fwd_scaled")

    def forward(self):
        """encoder + gather + decoder of the batch loaded by _load_batch (stream order)."""
        self._encode()
        self._gather()
        self._decode()

    def _decode(self, skip_out=False):
        """skip_out (training step): the output layer's forward is left to the fused output-layer backward, which reads the AB
        half of h3 once for both directions of the layer (dpd_small_grads.fwd_y); `backward` must follow."""
        lib, s, P = L.load(), L.cur_stream(), self.P
        Q = 2 * self.B * self.N
        self._join_optimizer()          # the weights (and what is derived from them) of a side-stream optimizer step
        skip_out = bool(skip_out and self.fuse_out)
        self._out_pending = skip_out
        # DPD_BF16 training step: layer 3's activation leaves its GEMM as ONE bf16 plane and the fused output-layer kernel reads that
        # (17 MB less to write and 33 MB less to read per step at B = 64); options["h3_plane"] = False keeps the fp32 h3
        self._h3_in_plane = bool(skip_out and self.fuse_loss and self._planes is not None and self._planes.np == 1
                                 and self._planes.h3_rc and self._tail_ok and self._h3_plane_opt)
        h3 = None if self._h3_in_plane else self.h3
        if self._wdirty:
            self.refresh_weight_planes()
        L.check(lib.dpd_decoder_fwd(L.ptr(self.X), L.ptr(self.mask), Q, P.KP, P.H, self._cparams, self.dt, L.ptr(self.h1),
                                    L.ptr(self.h2), L.ptr(h3), None if skip_out else L.ptr(self.y),
                                    None if skip_out else L.ptr(self.pred), L.ptr(self.ws), self.ws.numel() * 4, self._planes, s),
                "dpd_decoder_fwd")

    def backward(self, labels, defer_small=False):
        """defer_small: db3 / dW4 / db4 and the loss stay block partials (the fused optimizer launch sums them)."""
        lib, s, P = L.load(), L.cur_stream(), self.P
        BN = self.B * self.N
        L.req(labels, name="labels", numel=BN)
        if self._wdirty:
            self.refresh_weight_planes()
        d, wsb = self._gviews, self.ws.numel() * 4
        gv = self._gviews
        # exact-fp32 compute type: db1 / db2 are by-products of the dW GEMMs (column sums of the operand they stream: deterministic),
        # so the data chain's atomic column sums are switched off; the plane compute types keep the fused-epilogue form
        det_db = self.dt == 0 and BN % 32 == 0 and self._det_db_opt
        sdb1, sdb2 = (None, None) if det_db else (gv[1], gv[3])
        dbp = self._db_partials if det_db else None
        if self.fuse_loss:      # d loss_samples / d pred and the two loss values come out of the output-layer backward
            fwd = (self.y, self.pred) if self._out_pending else (None, None)     # ... and, in a training step, y and pred too
            self._out_pending = False
            small = L.make_small_grads(sdb1, sdb2, gv[5], gv[6], gv[7], self._partials, self.pred, labels, self.loss, 1.0, dbp, *fwd)
        else:
            L.check(lib.dpd_l1_loss(L.ptr(self.pred), L.ptr(labels), BN, 1, 1.0, L.ptr(self.loss), L.ptr(self.dpred), s), "dpd_l1_loss")
            small = L.make_small_grads(sdb1, sdb2, gv[5], gv[6], gv[7], self._partials, db_partials=dbp)

        def data(phases):   # db1..db3, dW4, db4 fall out of the data chain (fused epilogues / one small kernel)
            L.check(lib.dpd_decoder_bwd_data(L.ptr(self.dpred), L.ptr(self.mask), L.ptr(self.y), L.ptr(self.h1), L.ptr(self.h2),
                                             None if self._h3_in_plane else L.ptr(self.h3), BN, P.KP, P.H, self._cparams, self.dt, L.ptr(self.dy), L.ptr(self.g3),
                                             L.ptr(self.g2), L.ptr(self.g1), None, small, L.ptr(self.ws), wsb, self._planes,
                                             phases, L.cur_stream()), "dpd_decoder_bwd_data")   # stream at CALL time (graph branches)

        def dw(layer, act, g, dW):
            db = gv[2 * layer - 1] if (det_db and layer in (1, 2)) else None
            L.check(lib.dpd_decoder_bwd_weights(layer, L.ptr(act), act.stride(0) if act is not None else dW.shape[0], L.ptr(g), BN, dW.shape[0], dW.shape[1], self.dt,
                                                L.ptr(dW), L.ptr(db), L.ptr(self.ws), wsb, self._planes, L.ptr(dbp) if db is not None else None,
                                                L.cur_stream()),
                    "dpd_decoder_bwd_weights(%d)" % layer)

        def dw23():
            if BN % 32 == 0:      # layers 2 and 3 have identical shapes: one grouped launch
                L.check(lib.dpd_decoder_bwd_weights_pair(L.ptr(self.h1), L.ptr(self.g2), L.ptr(d[2]), L.ptr(self.h2), L.ptr(self.g3),
                                                         L.ptr(d[4]), P.H, BN, P.H, P.H, self.dt, L.ptr(self.ws), wsb, self._planes,
                                                         L.ptr(gv[3]) if det_db else None, L.ptr(dbp), L.cur_stream()),
                        "dpd_decoder_bwd_weights_pair")
            else:
                dw(2, self.h1, self.g2, d[2])
                dw(3, self.h2, self.g3, d[4])

        if self.reducer and self._trio and self.dp_schedule == "grouped":
            # DPD_DP_SCHEDULE=grouped (plane compute types, opt-in until an 8-GPU run has compared them): the single-GPU order -- data chain,
            # then dW1 + dW2 + dW3 as ONE grouped launch (20 us less GEMM time at B = 64 than the three early launches) -- and the whole
            # gradient as ONE all-reduce behind it; with the optimizer on the collectives' stream its tail overlaps the next front end
            data(7)
            rc = lib.dpd_decoder_bwd_weights_trio(BN, P.KP, P.H, self.dt, L.ptr(d[0]), L.ptr(d[2]), L.ptr(d[4]), L.ptr(self.ws), wsb, self._planes,
                                                  L.cur_stream())
            if rc == -3:        # DPD_E_UNSUPPORTED (a shape / plane set the grouped launch does not take; every rank sees the same shapes): the
                self._trio = False                                  # separate launches, like the single-GPU path below, from now on
                dw(1, self.X, self.g1, d[0])
                dw23()
            else:
                L.check(rc, "dpd_decoder_bwd_weights_trio")
            if self._after_dw1 is not None:
                self._after_dw1()
            self.reducer.reduce_async(0, upto=len(P.bucket_bounds) - 2)
            return
        if self.reducer and self.dp_schedule in ("early", "grouped"):      # ("grouped" without the grouped launch = "early")
            # Data-parallel schedule: every weight gradient is produced as early as its inputs exist, smallest bucket first,
            # so that the all-reduces (serial on the RCCL stream) start ~250 us before the backward ends instead of after dW1:
            #   output layer -> dW3 -> [bucket 2: W3,b3,W4,b4] -> g2 -> dW2 -> [bucket 1: W2,b2] -> g1 -> dW1 -> [bucket 0]
            # options["dp_buckets"] = 2 (default): layers 2-4 travel as ONE collective after dW2 (8.4 MB, ~140 us of GEMMs still to come)
            # -- every collective costs the compute stream a cross-stream event hop (~20 us on this runtime)
            # and the exposed part is the layer-1 bucket either way; =3: one collective per bucket, the first after dW3.
            three = self._dp_buckets == 3
            data(1)
            if three:
                dw(3, self.h2, self.g3, d[4])
                self.reducer.reduce_async(2)
                data(2)
                dw(2, self.h1, self.g2, d[2])
                self.reducer.reduce_async(1)
            else:
                data(2)
                dw23()                      # dW2 + dW3 as the grouped launch of the single-GPU order
                self.reducer.reduce_async(1, upto=2)
            data(4)
            dw(1, self.X, self.g1, d[0])
            if self._after_dw1 is not None:
                self._after_dw1()
            self.reducer.reduce_async(0)
            return
        data(7 | 16 if defer_small else 7)     # 16: db3 / dW4 / db4 and the loss stay block partials (the optimizer sums them)
        if self._trio and not self.reducer:
            # plane compute types: dW1 + dW2 + dW3 as ONE grouped launch (288 tiles of 128x128 for 256 CUs; apart they leave 96-192
            # CUs idle for the ~27 us a K = 4096 loop takes: DESIGN.md section 3.5)
            rc = lib.dpd_decoder_bwd_weights_trio(BN, P.KP, P.H, self.dt, L.ptr(d[0]), L.ptr(d[2]), L.ptr(d[4]), L.ptr(self.ws), wsb, self._planes,
                                                  L.cur_stream())
            if rc == 0:
                if self._after_dw1 is not None:
                    self._after_dw1()
                return
            if rc != -3:
                L.check(rc, "dpd_decoder_bwd_weights_trio")
            self._trio = False            # DPD_E_UNSUPPORTED: shapes / planes the grouped launch does not take
        dw(1, self.X, self.g1, d[0])
        if self._after_dw1 is not None:
            self._after_dw1()             # X / mask are free from here on: the prefetch pipeline hooks in
        if self.reducer:
            self.reducer.reduce_async(0)
        dw23()
        if self.reducer:      # DPD_DP_SCHEDULE=late (A/B reference): plain order, all-reduces start after dW1
            self.reducer.reduce_async(1)
            self.reducer.reduce_async(2)

    def apply_gradients(self, tail_from_partials=False):
        """tf.train.AdamOptimizer.apply_gradients with the staircase learning rate (train_multi_gpu...:216,301,976-990); lr_t is
        computed on the host (a device-side schedule kernel of one thread costs 4.7 us per step on MI355X: launch latency)."""
        base_lr, decay_step, decay_rate, b1, b2, eps = self.hp
        lr = learning_rate(self.t, base_lr, decay_step, decay_rate)     # global_step before the increment (TF semantics)
        self.t += 1
        self._last_lr = lr
        lr_t = lr * math.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
        gscale = 1.0
        side = None
        if self.reducer:
            self._join_optimizer()
            if self.adam_on_side and self.reducer.active and self.reducer.mode == "allreduce" and self.reducer.backend == "rccl":
                # data-parallel step: the optimizer runs on the stream the collectives ran on, right behind the last one, and the
                # compute stream is NOT joined: the next step's encoder + window gather (which do not read the weights) run meanwhile
                # and the decoder waits for the optimizer's event (_join_optimizer) -- the tail of the last all-reduce, the cross-queue
                # hop and Adam itself hide under ~35 us of front end.  One rank, direct RCCL (profiles/r04_dp_side_ab.txt): f32 0.5947 ->
                # 0.5692 ms, bf16 B=64 0.3395 -> 0.3216; through torch.distributed's streams it LOSES (0.6148 -> 0.6266): direct reducer only
                side = self.reducer.wait_side()
            else:
                self.reducer.wait()
            gscale = self.reducer.grad_scale
            if self.reducer.active and self.reducer.mode == "zero1":
                # sharded optimizer (ZeRO stage 1, ddp.py): this rank holds the summed gradient of its shards only; Adam on those
                # ranges (1/P of the 28 B per parameter), then the updated fp32 parameters are all-gathered and everything derived
                # from the weights (operand planes / transposed copies) is refreshed lazily like after any optimizer step
                lib = L.load()
                pf = self.P.flat.detach()
                for lo, hi in self.reducer.owned_ranges():
                    L.check(lib.dpd_adam_tf(L.ptr(pf[lo:hi]), L.ptr(self.grad[lo:hi]), L.ptr(self.m_state[lo:hi]), L.ptr(self.v_state[lo:hi]),
                                            hi - lo, lr_t, b1, b2, eps, gscale, L.cur_stream()), "dpd_adam_tf(shard)")
                self.reducer.gather_params(pf)
                self._wdirty = True
                self.P.invalidate_derived()
                return
        import contextlib
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            if self.fused_adam and (self.W2T is not None or self._afuse[0].np or tail_from_partials):
                af = self._afuse[1 if tail_from_partials else 0]
                L.check(L.load().dpd_adam_tf_fused(L.ptr(self.P.flat), L.ptr(self.grad), L.ptr(self.m_state), L.ptr(self.v_state),
                                                   self.P.numel, lr_t, b1, b2, eps, gscale, af, L.cur_stream()), "dpd_adam_tf_fused")
                # the transposed copies / operand planes written in the same pass are already those of the new weights
                self._wdirty = self._planes is not None and not self._afuse[0].np
            else:
                L.check(L.load().dpd_adam_tf(L.ptr(self.P.flat), L.ptr(self.grad), L.ptr(self.m_state), L.ptr(self.v_state),
                                             self.P.numel, lr_t, b1, b2, eps, gscale, L.cur_stream()), "dpd_adam_tf")
                self._wdirty = True
            self.P.invalidate_derived()     # DPDistParams.transposed() keys its cache on flat._version, which a raw-pointer update never bumps
            if side is not None:
                if self._ev_opt is None:
                    from .hipevents import LightEvent
                    self._ev_opt = LightEvent()
                self._ev_opt.record(side)
                self._opt_pending = True

    def _join_optimizer(self):
        """Make the current stream wait for an optimizer step that runs on the reducer's side stream (data-parallel steps with
        adam_on_side).  Every method of the trainer that reads the weights or the optimizer state calls this first; code that
        reads `params.flat` directly after `step()` must call `join_optimizer()` itself."""
        if self._opt_pending:
            red = self.reducer
            e0 = red.exposure.begin() if red is not None else None
            self._ev_opt.wait(torch.cuda.current_stream())
            if red is not None:
                red.exposure.end(e0)
            self._opt_pending = False

    join_optimizer = _join_optimizer

    @property
    def lr(self):
        """learning rate of the last step taken."""
        return self._last_lr

    def _take_front(self, pcA, pcB, noise):
        """Make the front-end buffers (pts, q, fv, X, mask, vox) hold this batch on the current stream."""
        main = torch.cuda.current_stream()
        if self._pref_key is not None:
            self._ev_front.wait(main)                  # whatever the side stream was doing with the buffers is ordered first
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
        self._decode(skip_out=True)
        if prefetch is not None:
            if self._side is None:
                from .hipevents import LightEvent
                self._side = torch.cuda.Stream(device=self.P.flat.device)
                # device-local ordering only: events without the system-scope fence (a torch.cuda.Event record in mid-stream costs
                # the compute stream ~6 us, these 0.3 us: tools/event_cost.py)
                self._ev_front, self._ev_xfree, self._ev_fwd = (LightEvent(system_fence=False) for _ in range(3))
            main = torch.cuda.current_stream()
            self._ev_fwd.record(main)                  # inputs complete + this step's gather/decoder ordered before the side work

            def launch_front():
                self._ev_xfree.record(main)            # recorded right after dW1, the last reader of X / mask
                with torch.cuda.stream(self._side):
                    self._ev_fwd.wait(self._side)
                    self._front(*prefetch, gate=self._ev_xfree)
                    self._ev_front.record(self._side)
                self._pref_key = self._key(*prefetch)
            self._after_dw1 = launch_front
        # single-GPU steps: the reduction of the output layer's block partials (and the loss) is deferred into the optimizer launch
        defer = self.fused_adam and self.fuse_loss and self._tail_ok and self.reducer is None
        try:
            self.backward(labels.reshape(-1), defer_small=defer)
        finally:
            self._after_dw1 = None
        self.apply_gradients(tail_from_partials=defer)
        return self.loss

    def dp_schedule_candidates(self, modes=None):
        """Orders of the data-parallel backward, optionally crossed with communication forms ("mode/order"): all of them give the same
        averaged gradient bit for bit except "grouped" (one grouped dW launch: same sums in another fp32 order)."""
        orders = ["early", "grouped", "late"]
        if not modes:
            return orders
        return ["%s/%s" % (m, o) for m in modes for o in orders]

    def _dp_supported(self, name):
        mode, _, order = name.rpartition("/")
        mode = mode or (self.reducer.mode if self.reducer is not None else "allreduce")
        if mode not in ("allreduce", "rs_ag", "zero1") or order not in ("early", "grouped", "late"):
            return False
        if mode == "zero1" and getattr(self.reducer, "wire", "f32") != "f32":
            return False                    # the sharded optimizer keeps fp32 master shards: fp32 wire only
        # (the grouped launch exists for the one-plane type; the sharded optimizer is only exercised with the orders its tests run)
        return order != "grouped" or (self._trio and mode != "zero1")

    def set_dp_mode(self, mode):
        """Switch the communication form of the data-parallel step (allreduce | rs_ag | zero1) by re-creating the reducer.  COLLECTIVE:
        every rank must call it with the same mode (communicators are created, the start-up cross-check runs again)."""
        red = self.reducer
        if red is None or red.mode == mode:
            return
        self._join_optimizer()
        torch.cuda.synchronize()
        group, force = getattr(red, "group", None), os.environ.get("DPD_FORCE_DIST") == "1"
        if red.active and red.mode == "zero1" and red._calls:          # the slots live sharded: make them whole before the partition goes
            red.gather_params(self.m_state)
            red.gather_params(self.v_state)
            torch.cuda.synchronize()
        if hasattr(red, "close"):
            red.close()
        self.reducer = make_reducer(self.grad, self.P.bucket_bounds, group, force=force, mode=mode)

    def select_dp_schedule(self, pcA, pcB, labels, candidates=None, steps=20, warmup=5, spinup=40, modes=None):
        """Decide the data-parallel step's form by measurement: every candidate -- an order of the backward (early / grouped / late),
        crossed with the communication forms in `modes` (allreduce / rs_ag / zero1) when given -- runs `warmup` + `steps` real training
        steps on this batch, collectives, optimizer and all, timed between synchronisations; the times are MAX-reduced over the ranks and
        every rank takes the same winner (ddp.select_schedule; candidates some rank cannot run are dropped collectively beforehand).
        The parameters, the Adam slots and the step counter are restored after every candidate, so the run that follows starts from where
        it was.  COLLECTIVE: every rank must call it, with its own shard.

        Reproducibility: "grouped" sums the weight gradients in another fp32 order than the separate launches, and two runs can measure
        different winners when candidates are close -- a run that must be bitwise reproducible PINS the schedule (DPD_DP_SCHEDULE=<order>,
        DPD_DP_MODE=<mode>; dpdist_amd.train does by default and stores a measured choice next to its checkpoints).  Callers opt in:
        bench.py measures; dpdist_amd.train only with --dp_schedule auto.  Returns the dict for the `dp` record."""
        import time
        from . import ddp
        if self.reducer is None or not getattr(self.reducer, "active", False):
            self.dp_schedule_info = {"schedule": self.dp_schedule, "source": "no collectives: nothing to choose"}
            return self.dp_schedule_info
        if os.environ.get("DPD_DP_SELECT"):          # "steps,warmup,spinup" (tests that time-share one GPU between many ranks shorten it)
            steps, warmup, spinup = (int(x) for x in os.environ["DPD_DP_SELECT"].split(","))
        pinned_order = os.environ.get("DPD_DP_SCHEDULE", "auto") != "auto"
        pinned_mode = "DPD_DP_MODE" in os.environ or not modes
        if pinned_order and pinned_mode:
            self.dp_schedule_info = {"schedule": self.dp_schedule, "mode": self.reducer.mode, "source": "pinned: DPD_DP_SCHEDULE / DPD_DP_MODE"}
            return self.dp_schedule_info
        if candidates is not None:
            cands = list(candidates)
        else:
            orders = [self.dp_schedule] if pinned_order else ["early", "grouped", "late"]
            cands = orders if pinned_mode else ["%s/%s" % (m, o) for m in modes for o in orders]
        self._join_optimizer()
        torch.cuda.synchronize()
        keep = (self.P.flat.detach().clone(), self.m_state.clone(), self.v_state.clone(), self.t)
        mode0 = self.reducer.mode

        def restore():
            self._join_optimizer()
            torch.cuda.synchronize()
            self.P.flat.detach().copy_(keep[0])
            self.m_state.copy_(keep[1])
            self.v_state.copy_(keep[2])
            self.t = keep[3]
            self._wdirty = True
            self.P.invalidate_derived()

        def time_fn(name):
            if self.progress is not None:       # (a launch watchdog's heartbeat: a candidate that runs is progress, whatever the phase limit)
                self.progress()
            mode, _, order = name.rpartition("/")
            if mode:
                self.set_dp_mode(mode)          # collective; every rank walks the same candidate list in the same order
            self.dp_schedule = order
            try:
                for _ in range(warmup):
                    self.step(pcA, pcB, labels)
                self._join_optimizer()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    self.step(pcA, pcB, labels)
                self._join_optimizer()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / steps * 1e3
            finally:
                restore()

        group = getattr(self.reducer, "group", None)
        try:        # the chip needs ~25 ms of sustained work to reach its steady clock: without this the FIRST candidate
            for _ in range(spinup):      # would be timed on the ramp and lose to the ones after it
                self.step(pcA, pcB, labels)
        finally:
            restore()
        choice, table = ddp.select_schedule(cands, time_fn, self.P.flat.device, group, supported=self._dp_supported)
        mode, _, order = choice.rpartition("/")
        self.set_dp_mode(mode or mode0)
        self.dp_schedule = order
        self.dp_schedule_info = {"schedule": order, "mode": self.reducer.mode,
                                 "source": "measured at start-up: %d steps per candidate after a %d-step spin-up, MAX over ranks" % (steps, spinup),
                                 "candidates_ms": table,
                                 "note": "bitwise-equivalent candidates except 'grouped' (another fp32 summation order of dW); pin with DPD_DP_SCHEDULE / DPD_DP_MODE"}
        return self.dp_schedule_info

    # -- optimizer state <-> TF global variables ------------------------------------------------------------------
    def gather_optimizer_state(self):
        """DPD_DP_MODE=zero1: the Adam slots m / v are only current on the rank that owns a shard; all-gather them (COLLECTIVE:
        every rank must call) so that `tf_global_variables` on any rank sees what a replicated optimizer would hold.  No-op for
        the replicated modes."""
        self._join_optimizer()
        red = self.reducer
        if red is not None and red.active and red.mode == "zero1" and red._calls:
            red.gather_params(self.m_state)
            red.gather_params(self.v_state)

    def tf_global_variables(self):
        """Everything the reference's `tf.train.Saver()` stores besides summaries (train_multi_gpu_pc_compare_dist.py:305,
        354-357): the 8 decoder variables, the global step `batch` (float scalar, :205-207), AdamOptimizer's non-slot variables
        `beta1_power` / `beta2_power` (TF keeps beta^(t+1) after t steps) and the slots `<variable>/Adam` (m), `<variable>/Adam_1`
        (v), all in the TF layouts.  Host sync."""
        _, _, _, b1, b2, _ = self.hp
        self._join_optimizer()
        sd = dict(self.P.tf_state_dict())
        for suffix, flat in (("/Adam", self.m_state), ("/Adam_1", self.v_state)):
            for n, a in self.P.tf_state_dict(flat).items():
                sd[n + suffix] = a
        sd["batch"] = np.float32(self.t)
        sd["beta1_power"] = np.float32(b1 ** (self.t + 1))
        sd["beta2_power"] = np.float32(b2 ** (self.t + 1))
        return sd

    @torch.no_grad()
    def load_tf_global_variables(self, sd):
        """Inverse of tf_global_variables; optimizer entries that are absent (a weights-only checkpoint) leave that part of the
        state untouched.  Returns the list of state groups that were restored."""
        _, _, _, b1, b2, _ = self.hp
        self._join_optimizer()
        got = ["weights"]
        self.P.load_tf_state_dict(sd)
        self._wdirty = True
        if all((n + "/Adam") in sd and (n + "/Adam_1") in sd for n in self.P.tf_state_dict()):
            self.P.load_tf_state_dict(sd, flat=self.m_state, suffix="/Adam")
            self.P.load_tf_state_dict(sd, flat=self.v_state, suffix="/Adam_1")
            got.append("adam_slots")
        if "batch" in sd:      # beta1_power / beta2_power are functions of the step count (beta^(t+1)): nothing else to restore
            self.t = int(round(float(np.asarray(sd["batch"]))))
            got.append("schedule")
        return got

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
