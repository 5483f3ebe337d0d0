"""Host-side mirror of the reference's model-module contract, on top of the HIP C ABI.

Reference boundary (relative to /root/reference):
    models/dpdist_and_aue.py:23-28    placeholder_inputs(batch_size, num_point, NUM_DIMS)
    models/dpdist_and_aue.py:31-86    get_model(pcA, pcB, is_training, bn_decay=None, wd=0.0, bn=True, Embedding_Size=512,
                                                pn='pn', sig=True, k=0, overlap=False, localSNmlp=[1024,1024,1024],
                                                full_fv=True, sigma3dmfv=0.125, conv_version=1, add_noise=0)
                                      -> (pred_set{'pred_listAB','pred_listBA'} [B,N,1,3], end_points{}, embedding_set)
    models/dpdist_and_aue.py:203-204  get_loss(pred_set, end_points, labels, loss_type='l1_dist')
                                      -> (pred_listAB[...,0] squeezed [B,N], loss_pred); scalar L1 in collection 'loss_samples'
Same names, argument meaning, defaults and return structure; tensors are torch CUDA float32 instead of TF placeholders.
Variables live in a module-level store keyed like TF's graph ('pc_compare/dpdist_local/mapper_conv{1..4}/{weights,biases}',
dpdist_and_aue.py:36, dpdist_util.py:514-543, tf_util.py:207,217) and are created on first use with TF's Xavier rule.

All arithmetic runs in the HIP kernels of dpdist_amd/csrc via ops.py; autograd is wired by `_DPDistFn`.
"""
import math

import numpy as np
import torch
from torch import nn

from . import asloss, lib as L, ops

TF_NAME = "pc_compare/dpdist_local/mapper_conv%d/%s"
F = 20


def placeholder_inputs(batch_size, num_point, NUM_DIMS=2, device="cuda"):
    """models/dpdist_and_aue.py:23-28, same positional arguments and the same default NUM_DIMS=2 (the trainer always passes
    NUM_DIMS=3, train_multi_gpu_pc_compare_dist.py:192): zero-filled float32 stand-ins for the placeholders input1, input2
    [B,N,NUM_DIMS] and labels12, labels21 [B,N], in that order."""
    z = lambda *s: torch.zeros(*s, device=device, dtype=torch.float32)   # noqa: E731
    return z(batch_size, num_point, NUM_DIMS), z(batch_size, num_point, NUM_DIMS), z(batch_size, num_point), z(batch_size, num_point)


def _align4(n):
    return (n + 3) // 4 * 4


class DPDistParams(nn.Module):
    """The 8 decoder variables in ONE flat fp32 buffer (so gradients, Adam state and all-reduce buckets are flat).

    Internal layout (include/dpdist_capi.h): W1p [KP,H1] with rows [window(E) | xyz(3) | zero pad], i.e.
    W1p[0:E] = tf_w1[3:3+E], W1p[E:E+3] = tf_w1[0:3].  Segment order: [W1p, b1 | W2, b2, W3, b3, W4, b4];
    bucket 0 = layer 1 (10.3 MB), bucket 1 = the rest (8.4 MB).
    """

    def __init__(self, k=5, mlp=(1024, 1024, 1024), device="cuda", init="xavier_tf", compute_dtype="f32"):
        super().__init__()
        # compute type of the three wide layers (include/dpdist_capi.h: enum dpd_dtype): "f32" exact fp32 MFMA,
        # "f32x3" fp32-equivalent split-bf16 MFMA, "bf16" mixed precision (bf16 operands, fp32 accumulate)
        self.compute_dtype = compute_dtype
        self.k = int(k)
        self.mlp = tuple(int(h) for h in mlp)
        if len(set(self.mlp)) != 1 or self.mlp[0] % 64:
            raise NotImplementedError("decoder widths must be equal and a multiple of 64, got %s" % (self.mlp,))
        self.H = self.mlp[0]
        self.E = self.k ** 3 * F
        self.KP = (self.E + 3 + 31) // 32 * 32       # == dpd_padded_width(k): whole 32-deep K-tiles
        H = self.H
        shapes = [("W1p", (self.KP, H)), ("b1", (H,)), ("W2", (H, H)), ("b2", (H,)), ("W3", (H, H)), ("b3", (H,)),
                  ("W4", (H, 3)), ("b4", (3,))]
        self._segments = {}
        off = 0
        for n, shp in shapes:
            cnt = int(np.prod(shp))
            self._segments[n] = (off, cnt, shp)
            off += _align4(cnt)
        self.numel = off
        # gradient buckets in flat order [layer 1 | layer 2 | layers 3-4]; the data-parallel backward of trainer.py
        # produces (and all-reduces) them in the order 2, 1, 0
        self.bucket_bounds = [0, self._segments["W2"][0], self._segments["W3"][0], off]
        self.flat = nn.Parameter(torch.zeros(off, device=device, dtype=torch.float32))
        if init == "xavier_tf":
            self.reset_parameters_tf()

    # -- views ------------------------------------------------------------------------------------
    def view(self, name, flat=None):
        off, cnt, shp = self._segments[name]
        return (self.flat if flat is None else flat).detach()[off:off + cnt].view(*shp)

    def views(self, flat=None):
        return [self.view(n, flat) for n in ("W1p", "b1", "W2", "b2", "W3", "b3", "W4", "b4")]

    def invalidate_derived(self):
        """Drop what is cached FROM the parameter values (transposed fp32 copies, bf16 weight planes of the as-loss node).  Callers that
        update `flat` through raw device pointers (the trainer's Adam kernels never bump `flat._version`) must call this."""
        self._tr_key = None
        self._wplanes = None
        self.__dict__["_derived_gen"] = self.__dict__.get("_derived_gen", 0) + 1    # captured graphs that baked derived copies in re-capture
        for engines in self.__dict__.get("_asloss_engines", {}).values():      # the as-loss engines re-derive their weight copies
            for e in engines:
                e._wkey = None

    def cparams(self, flat=None):
        """The C-ABI pointer block of the 8 variables inside `flat` (include/dpdist_capi.h: dpd_decoder_params), cached per buffer
        address: building the eight views costs the host ~80 us per call, which is what bounds the as-loss node of the plane compute
        types at the registration batch."""
        from . import lib as L
        src = self.flat if flat is None else flat
        key = src.data_ptr()
        c = getattr(self, "_cparams_cache", None)
        if c is None or c[0] != key:
            c = self._cparams_cache = (key, L.make_params(*self.views(src)))
        return c[1]

    def transposed(self, flat=None):
        """(W2T, W3T, W1pT): transposed copies for the fp32 backward data GEMMs (include/dpdist_capi.h:
        dpd_weights_transpose), cached until the parameter buffer changes (frozen weights in as-loss mode: one launch)."""
        from . import lib as L
        src = self.flat if flat is None else flat
        key = (src.data_ptr(), src._version)
        if getattr(self, "_tr_key", None) != key:
            H, KP = self.H, self.KP
            dev = src.device
            t = [torch.empty(H, H, device=dev), torch.empty(H, H, device=dev), torch.empty(H, KP, device=dev)]
            cp = L.make_params(*self.views(src))
            L.check(L.load().dpd_weights_transpose(cp, KP, H, L.ptr(t[0]), L.ptr(t[1]), L.ptr(t[2]), L.cur_stream()),
                    "dpd_weights_transpose")
            self._tr, self._tr_key = t, key
        return self._tr

    # -- TF interchange -----------------------------------------------------------------------------
    def tf_shapes(self):
        H = self.H
        return [[1, self.E + 3, 1, H], [1, 1, H, H], [1, 1, H, H], [1, 1, H, 3]]

    @torch.no_grad()
    def reset_parameters_tf(self, generator=None):
        """tf.contrib.layers.xavier_initializer on the conv kernels (utils/tf_util.py:90-91): uniform +-sqrt(6/(fan_in+fan_out))
        with fan_in = kh*kw*in, fan_out = kh*kw*out; zero biases (:217-218)."""
        sd = {}
        for l, shp in enumerate(self.tf_shapes(), 1):
            recept = shp[0] * shp[1]
            lim = math.sqrt(6.0 / (shp[2] * recept + shp[3] * recept))
            w = (torch.rand(shp, generator=generator) * 2 - 1) * lim
            sd[TF_NAME % (l, "weights")] = w
            sd[TF_NAME % (l, "biases")] = torch.zeros(shp[3])
        self.load_tf_state_dict(sd)

    @torch.no_grad()
    def load_tf_state_dict(self, sd, flat=None, suffix=""):
        """sd: TF variable name -> array in the TF layout ([1,E+3,1,H], [1,1,H,H] x2, [1,1,H,3], biases).
        flat / suffix: fill another flat buffer of the same layout from the variables `<name><suffix>` instead (the Adam slots
        of a TF checkpoint are `<variable>/Adam` and `<variable>/Adam_1`)."""
        dev = self.flat.device
        if flat is not None or suffix:
            sd = {n: sd[n + suffix] for n in (TF_NAME % (l, t) for l in (1, 2, 3, 4) for t in ("weights", "biases"))}
            tgt = self.flat if flat is None else flat
            keep = self.flat.data
            try:                                    # reuse the layout code below on the other buffer
                self.flat.data = tgt.data
                return self.load_tf_state_dict(sd)
            finally:
                self.flat.data = keep
        get = lambda n: torch.as_tensor(np.asarray(sd[n]) if not torch.is_tensor(sd[n]) else sd[n], dtype=torch.float32)  # noqa: E731
        w1 = get(TF_NAME % (1, "weights")).reshape(self.E + 3, self.H)
        W1p = torch.zeros(self.KP, self.H)
        W1p[:self.E] = w1[3:]
        W1p[self.E:self.E + 3] = w1[:3]
        self.flat.zero_()
        self.view("W1p").copy_(W1p.to(dev))
        self.view("W2").copy_(get(TF_NAME % (2, "weights")).reshape(self.H, self.H).to(dev))
        self.view("W3").copy_(get(TF_NAME % (3, "weights")).reshape(self.H, self.H).to(dev))
        self.view("W4").copy_(get(TF_NAME % (4, "weights")).reshape(self.H, 3).to(dev))
        for l in (1, 2, 3, 4):
            self.view("b%d" % l).copy_(get(TF_NAME % (l, "biases")).to(dev))

    @torch.no_grad()
    def tf_state_dict(self, flat=None):
        """Inverse of load_tf_state_dict (also used to express flat gradients in the TF layout)."""
        W1p = self.view("W1p", flat).cpu()
        w1 = torch.cat([W1p[self.E:self.E + 3], W1p[:self.E]], 0)
        sd = {TF_NAME % (1, "weights"): w1.reshape(1, self.E + 3, 1, self.H).numpy()}
        sd[TF_NAME % (2, "weights")] = self.view("W2", flat).cpu().reshape(1, 1, self.H, self.H).numpy()
        sd[TF_NAME % (3, "weights")] = self.view("W3", flat).cpu().reshape(1, 1, self.H, self.H).numpy()
        sd[TF_NAME % (4, "weights")] = self.view("W4", flat).cpu().reshape(1, 1, self.H, 3).numpy()
        for l in (1, 2, 3, 4):
            sd[TF_NAME % (l, "biases")] = self.view("b%d" % l, flat).cpu().numpy()
        return sd


class _DPDistFn(torch.autograd.Function):
    """encoder -> window gather -> decoder, both directions, as one autograd node over the HIP kernels."""

    @staticmethod
    def forward(ctx, pcA, pcB, noise, flat, P, m, k, sigma):
        B, N, _ = pcA.shape
        # pts = [pcA+noise ; pcB] (dpdist_and_aue.py:45,56-61); q = [pcB ; pcA]: the BA half queries the UN-noised pcA (:69)
        pts, q = ops.stack_clouds(pcA, pcB, noise)
        fv = ops.mfv3d_fwd(pts, m, sigma)
        X, mask, vox = ops.patch_rows_fwd(q, fv, m, k, P.KP)
        params = P.views(flat)
        h1, h2, h3, y, pred = ops.decoder_fwd(X, mask, params, P.H, dtype=P.compute_dtype)
        ctx.P, ctx.cfg = P, (B, N, m, k, sigma)
        ctx.has_noise = noise is not None
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(pts, flat, X, mask, vox, h1, h2, h3, y)
        ctx.mark_non_differentiable(mask, vox)
        return pred, fv, mask, vox

    @staticmethod
    def backward(ctx, dpred, dfv_ext, _dm, _dv):
        P = ctx.P
        B, N, m, k, sigma = ctx.cfg
        pts, flat, X, mask, vox, h1, h2, h3, y = ctx.saved_tensors
        need_in = any(ctx.needs_input_grad[:3])
        need_w = ctx.needs_input_grad[3]
        if dpred is None:
            dpred = torch.zeros(2 * B * N, 3, device=pts.device, dtype=torch.float32)
        dpred = dpred.contiguous()
        Q = dpred.shape[0]
        params = P.views(flat)
        dflat, d, small = None, None, None
        if need_w:
            dflat = torch.zeros_like(flat)
            d = P.views(dflat)
            small = (d[1], d[3], d[5], d[6], d[7])     # db1, db2, db3, dW4, db4 come out of the data chain (fused)
        dt = P.compute_dtype
        ws = ops.workspace(Q, P.KP, P.H, flat.device, dt) if (need_w or dt) else None
        wT = P.transposed(flat) if dt in ("f32", 0) else None
        dy, g3, g2, g1, dX = ops.decoder_bwd_data(dpred, mask, y, h1, h2, h3, params, P.KP, need_in, small_grads=small, dtype=dt,
                                                  ws=ws, transposed=wT)
        if need_w:
            ops.decoder_bwd_weights(1, X, g1, Q, d[0], None, ws, dt)
            ops.decoder_bwd_weights(2, h1, g2, Q, d[2], None, ws, dt)
            ops.decoder_bwd_weights(3, h2, g3, Q, d[4], None, ws, dt)
        gA = gB = gN = None
        if need_in:
            dq, dfv = ops.patch_rows_bwd(dX, vox, 2 * B, N, m, k)
            if dfv_ext is not None:
                dfv = dfv + dfv_ext
            dpts = ops.mfv3d_bwd(pts, dfv.contiguous(), m, sigma)
            gA = dpts[:B] + dq[B:]      # encoder route (via pcA+noise) + query route (BA half)
            gB = dpts[B:] + dq[:B]
            gN = dpts[:B] if ctx.has_noise else None
        return gA, gB, gN, dflat, None, None, None, None


class _AsLossFn(torch.autograd.Function):
    """DPDist as a frozen loss in ONE autograd node (pcrnet-registration/iterative_PCRNet_ours.py:229-257, AUE splice
    train_multi_gpu...:417-431):  loss_pred = (mean(output1[...,0]) + mean(output2[...,0])) / 2 of (pcA, pcB), gradients to
    the two clouds only.  Same kernels as _DPDistFn + _L1LossFn, without the slicing / mean / add glue in between."""

    @staticmethod
    def forward(ctx, pcA, pcB, flat, P, m, k, sigma):
        B, N, _ = pcA.shape
        need_grad = pcA.requires_grad or pcB.requires_grad
        ctx.P, ctx.cfg = P, (B, N, m, k, sigma)
        ctx.fused_out = B * N < 16384
        ctx.planes = None
        ctx.engine = None
        # round 5: the whole evaluation behind ONE foreign call per direction on an engine's persistent buffers (dpdist_amd/asloss.py:
        # same kernels, same bits; asloss.ENGINE = False = the entry-by-entry path below)
        eng = asloss.acquire(P, flat, B, N, m, k, sigma, pcA.device) if ctx.fused_out else None
        if eng is not None:
            L.req(pcA, name="pcA", shape=(B, N, 3)), L.req(pcB, name="pcB", shape=(B, N, 3))
            loss = eng.forward(pcA, pcB, need_grad)
            if need_grad:
                ctx.engine, ctx.engine_version = eng, eng.version
                eng.hold(ctx)
            return loss[0]
        if ctx.fused_out and ops.AsLossPlanes.usable(P, 2 * B * N, P.compute_dtype) and asloss.PLANES:
            # plane compute types: the rows, h1, h2 (and g3 / g2 / g1 in the backward) live as bf16 RC planes written by their producers,
            # the frozen weights' planes are cached: no conversion launch per GEMM, no fp32 X / h1 / h2 (round 4; asloss.PLANES = False = before)
            cp = P.cparams(flat)
            pl = ops.AsLossPlanes(P, flat, 2 * B * N, P.compute_dtype, pcA.device)
            pts, mask, vox = ops.front_end_planes(pcA, pcB, m, sigma, k, pl)
            _, _, h3, _, _ = ops.decoder_fwd(None, mask, cp, P.H, dtype=P.compute_dtype, out_layer=False, planes=pl)
            _, _, loss, _, g3 = ops.out_asloss(h3, mask, cp, B * N, want_grad=need_grad)
            if need_grad:
                ctx.planes = pl
                ctx.save_for_backward(pts, flat, vox, g3)
            return loss[0]
        params = P.views(flat)
        pts, X, mask, vox = ops.front_end(pcA, pcB, None, m, sigma, k, P.KP)     # two launches (stack+encoder, norm+gather)
        if ctx.fused_out:
            # output layer, loss_pred AND the output-layer backward of d loss_pred / d pred from one launch (labels only enter
            # loss_samples, which is not used here); the upstream gradient is applied once, at the very end of the backward
            # (everything in between is linear in it)
            h1, h2, h3, _, _ = ops.decoder_fwd(X, mask, params, P.H, dtype=P.compute_dtype, out_layer=False)
            _, _, loss, _, g3 = ops.out_asloss(h3, mask, params, B * N, want_grad=need_grad)
            if need_grad:
                ctx.save_for_backward(pts, flat, vox, h1, h2, g3)
            return loss[0]
        # three-launch form (very large batches: B * N >= 16384): out_fwd, l1_loss(mode 2), out_bwd
        h1, h2, h3, y, pred = ops.decoder_fwd(X, mask, params, P.H, dtype=P.compute_dtype)
        loss, dpred = ops.l1_loss(pred, mask[:B * N], mode=2 if need_grad else 0)
        ctx.save_for_backward(pts, flat, vox, h1, h2, mask, h3, y, dpred if need_grad else pred)
        return loss[1]

    @staticmethod
    def backward(ctx, g):
        P = ctx.P
        B, N, m, k, sigma = ctx.cfg
        Q = 2 * B * N
        dt = P.compute_dtype
        if ctx.engine is not None:
            eng = ctx.engine
            if eng.version != ctx.engine_version:
                raise RuntimeError("DPDist as-loss node: its engine's buffers have been re-used by a later evaluation (a second backward after "
                                   "the first one released them); set dpdist_amd.asloss.ENGINE = False for graphs that are differentiated repeatedly")
            gA, gB = eng.backward(g)
            eng.release()
            return gA, gB, None, None, None, None, None
        if ctx.planes is not None:
            pts, flat, vox, g3 = ctx.saved_tensors
            _, _, _, _, dX = ops.decoder_bwd_data(None, None, None, None, None, None, P.cparams(flat), P.KP, True, dtype=dt, phases=6, g3=g3,
                                                  planes=ctx.planes)
            # (ctx.planes stays: a second backward through this node -- retain_graph -- recomputes g2 / g1 from the saved g3 and the intact h planes)
            _, dfv = ops.patch_rows_bwd(dX, vox, 2 * B, N, m, k, want_dq=False)
            dpts = ops.mfv3d_bwd(pts, dfv, m, sigma)
            gA, gB = ops.asloss_combine(dpts, dX, g, B, N, k)
            return gA, gB, None, None, None, None, None
        if ctx.fused_out:
            pts, flat, vox, h1, h2, g3 = ctx.saved_tensors
        else:
            pts, flat, vox, h1, h2, mask, h3, y, dpred = ctx.saved_tensors
        ws = ops.workspace(Q, P.KP, P.H, flat.device, dt) if dt else None
        wT = P.transposed(flat) if dt in ("f32", 0) else None
        if ctx.fused_out:
            _, _, _, _, dX = ops.decoder_bwd_data(None, None, None, h1, h2, None, P.views(flat), P.KP, True, dtype=dt, ws=ws, transposed=wT,
                                                  phases=6, g3=g3)
        else:
            _, _, _, _, dX = ops.decoder_bwd_data(dpred, mask, y, h1, h2, h3, P.views(flat), P.KP, True, dtype=dt, ws=ws, transposed=wT)
        _, dfv = ops.patch_rows_bwd(dX, vox, 2 * B, N, m, k, want_dq=False)
        dpts = ops.mfv3d_bwd(pts, dfv, m, sigma)
        gA, gB = ops.asloss_combine(dpts, dX, g, B, N, k)           # upstream * (encoder route + query route), one launch
        return gA, gB, None, None, None, None, None


class _L1LossFn(torch.autograd.Function):
    """utils/dpdist_util.py:962-980 on the HIP loss kernel: pred [2BN,3], labels [BN] -> [loss_samples, loss_pred]."""

    @staticmethod
    def forward(ctx, pred, labels):
        loss, _ = ops.l1_loss(pred, labels, mode=0)
        ctx.save_for_backward(pred, labels)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        pred, labels = ctx.saved_tensors
        BN = labels.numel()
        _, d1 = ops.l1_loss(pred, labels, mode=1)     # d loss_samples / d pred_AB
        _, d2 = ops.l1_loss(pred, labels, mode=2)     # d loss_pred / d pred
        g = d2 * dloss[1]
        g[:BN] += d1 * dloss[0]
        return g, None


# ------------------------------------------------------------------------------------------------
# TF-like module-level state: variable store + collections (what the reference's trainer reads back)
# ------------------------------------------------------------------------------------------------
_VARIABLES = {}
_COLLECTIONS = {}


def reset_default_graph():
    _VARIABLES.clear()
    _COLLECTIONS.clear()


def get_variable_store():
    return _VARIABLES


def get_collection(name):
    return list(_COLLECTIONS.get(name, []))


class _LazyEmbeddingSet(dict):
    """embedding_set of the reference ({'embedding_A','embedding_B'} = [B, m^3, k^3*20], 164 MB each at B=32):
    materialised with torch ops only if somebody reads it."""

    def __init__(self, fv, B, m, k):
        super().__init__()
        self._fv, self._B, self._m, self._k = fv, B, m, k

    def _window(self, fv):
        m, k, h = self._m, self._k, (self._k - 1) // 2
        g = torch.nn.functional.pad(fv.reshape(-1, m, m, m, F), (0, 0, h, h, h, h, h, h))
        p = g.unfold(1, k, 1).unfold(2, k, 1).unfold(3, k, 1).permute(0, 1, 2, 3, 5, 6, 7, 4)
        return p.reshape(fv.shape[0], m ** 3, -1)

    def __getitem__(self, key):
        if key == "embedding_A":
            return self._window(self._fv[:self._B])
        if key == "embedding_B":
            return self._window(self._fv[self._B:])
        raise KeyError(key)

    def keys(self):
        return ["embedding_A", "embedding_B"]


def get_model(pcA, pcB, is_training=None, bn_decay=None, wd=0.0, bn=True, Embedding_Size=512, pn="pn", sig=True, k=0,
              overlap=False, localSNmlp=[1024, 1024, 1024], full_fv=True, sigma3dmfv=0.0625 * 2, conv_version=1,
              add_noise=0, params=None):
    """Drop-in for models/dpdist_and_aue.py:get_model (3dmfv encoder, k>0, conv_version=1, no BN -- the configuration
    the reference trains and ships: train_multi_gpu_pc_compare_dist.py:59-67,93,224-229).  Other branches of the
    reference (pointnet encoder, k=0, conv_version 2/3, BN) are outside the hot path and raise NotImplementedError."""
    if pn == "pointnet":
        raise NotImplementedError("pointnet encoder is not on the DPDist hot path")
    if k <= 0 or not (k & 1):
        raise NotImplementedError("only the local-patch decoder (odd k>0) is implemented")
    if conv_version != 1:
        raise NotImplementedError("conv_version %r is not on the hot path" % (conv_version,))
    if bn and int(bn) != 0:
        raise NotImplementedError("BN is off in the reference configuration (--BN 0)")
    if not full_fv:
        raise NotImplementedError("full_fv=False is not implemented")
    if pcA.shape[-1] != 3:
        raise NotImplementedError("NUM_DIMS must be 3")
    m = int(math.ceil(Embedding_Size ** (1 / 3) - 1e-9))       # dpdist_util.py:41
    if m ** 3 != Embedding_Size:
        raise ValueError("Embedding_Size must be a perfect cube")
    if params is None:
        key = "pc_compare"                                        # tf.variable_scope('pc_compare') (:36)
        if key not in _VARIABLES:
            _VARIABLES[key] = DPDistParams(k=k, mlp=tuple(localSNmlp), device=pcA.device)
        params = _VARIABLES[key]
    B, N, _ = pcA.shape
    noise = None
    if torch.is_tensor(add_noise):
        noise = add_noise
    elif add_noise != 0:
        noise = torch.full_like(pcA, float(add_noise))
    pred, fv, mask, vox = _DPDistFn.apply(pcA.contiguous(), pcB.contiguous(), noise, params.flat, params, m, k,
                                          float(sigma3dmfv))
    pred = pred.view(2, B, N, 1, 3)
    pred_set = {"pred_listAB": pred[0], "pred_listBA": pred[1]}   # 'pc_compare/output1', 'pc_compare/output2' (:78-79)
    return pred_set, {}, _LazyEmbeddingSet(fv, B, m, k)


def get_loss(pred_set, end_points, labels, loss_type="l1_dist"):
    """models/dpdist_and_aue.py:203-204 -> utils/dpdist_util.py:962-980."""
    if loss_type != "l1_dist":
        raise NotImplementedError(loss_type)
    ab, ba = pred_set["pred_listAB"], pred_set["pred_listBA"]
    B, N = ab.shape[0], ab.shape[1]
    pred = torch.cat([ab.reshape(B * N, 3), ba.reshape(B * N, 3)], 0)
    loss = _L1LossFn.apply(pred, labels.reshape(-1).contiguous())
    _COLLECTIONS.setdefault("loss_samples", []).append(loss[0])    # tf.add_to_collection('loss_samples', loss) (:974)
    _COLLECTIONS.setdefault("loss_pred", []).append(loss[1])       # (:979)
    loss_samples = ab[:, :, :, 0].squeeze()                         # what the reference actually returns (:967-968,980)
    return loss_samples, loss[1]


class DPDistModel(nn.Module):
    """nn.Module form of the module contract: forward(pcA, pcB, add_noise=None) -> pred_set."""

    def __init__(self, Embedding_Size=512, k=5, localSNmlp=(1024, 1024, 1024), sigma3dmfv=0.125, device="cuda"):
        super().__init__()
        self.params_ = DPDistParams(k=k, mlp=tuple(localSNmlp), device=device)
        self.Embedding_Size, self.k, self.sigma, self.mlp = Embedding_Size, k, sigma3dmfv, list(localSNmlp)

    def forward(self, pcA, pcB, add_noise=None):
        pred_set, _, _ = get_model(pcA, pcB, True, bn=0, Embedding_Size=self.Embedding_Size, pn="3dmfv", k=self.k,
                                   localSNmlp=self.mlp, sigma3dmfv=self.sigma, add_noise=0 if add_noise is None else add_noise,
                                   params=self.params_)
        return pred_set

    def load_tf_state_dict(self, sd):
        self.params_.load_tf_state_dict(sd)

    def tf_state_dict(self):
        return self.params_.tf_state_dict()


class DPDistLoss(nn.Module):
    """DPDist as a frozen loss (pcrnet-registration/iterative_PCRNet_ours.py:229-251):
    loss = (mean(output1[...,0]) + mean(output2[...,0])) / 2 for (source, template); gradients flow to the inputs only."""

    capturable = True       # no host synchronisation, no host-side state per evaluation: the whole evaluation can sit inside a hipGraph

    def __init__(self, model):
        super().__init__()
        self.model = model
        for p in self.model.parameters():
            p.requires_grad_(False)

    def graph_key(self):
        """What a captured graph of this loss has baked in (registration.IterativeRegistration re-captures when it changes): the weight
        buffer, its version, the compute type and the generation of the copies derived from the weights."""
        P = self.model.params_
        return (P.flat.data_ptr(), P.flat._version, P.compute_dtype, P.__dict__.get("_derived_gen", 0))

    def forward(self, source, template):
        mod = self.model
        m = int(math.ceil(mod.Embedding_Size ** (1 / 3) - 1e-9))
        return _AsLossFn.apply(source.contiguous(), template.contiguous(), mod.params_.flat, mod.params_, m, mod.k,
                               float(mod.sigma))
