"""Tensor-level wrappers over the C ABI (one function per entry point of include/dpdist_capi.h).

All tensors are torch CUDA float32 (int32 for `vox`), contiguous; outputs are allocated with torch's
caching allocator and the kernels are enqueued on torch's current HIP stream.
"""
import torch

from . import lib as L

F = 20


def padded_width(k):
    return L.load().dpd_padded_width(int(k))


def mfv3d_fwd(pts, m, sigma):
    """pts [C,N,3] -> fv [C,m^3,20]   (utils/dpdist_util.py:22-141)"""
    L.req(pts, name="pts")
    C, N, _ = pts.shape
    fv = torch.empty(C, m ** 3, F, device=pts.device, dtype=torch.float32)
    L.check(L.load().dpd_mfv3d_fwd(L.ptr(pts), C, N, m, float(sigma), L.ptr(fv), L.cur_stream()), "dpd_mfv3d_fwd")
    return fv


def mfv3d_bwd(pts, dfv, m, sigma, sliced=True):
    """dfv [C,m^3,20] -> dpts [C,N,3]; sliced: 4 workgroups per cloud through a small workspace (default)."""
    L.req(pts, name="pts"), L.req(dfv, name="dfv")
    C, N, _ = pts.shape
    dpts = torch.empty_like(pts)
    lib = L.load()
    ws = torch.empty(lib.dpd_mfv3d_bwd_workspace_bytes(C, m) // 4, device=pts.device, dtype=torch.float32) if sliced else None
    L.check(lib.dpd_mfv3d_bwd(L.ptr(pts), L.ptr(dfv), C, N, m, float(sigma), L.ptr(dpts), L.ptr(ws),
                              ws.numel() * 4 if ws is not None else 0, L.cur_stream()), "dpd_mfv3d_bwd")
    return dpts


def front_end(pcA, pcB, noise, m, sigma, k, KP=None):
    """The front end in two launches (include/dpdist_capi.h: dpd_mfv3d_fwd_stacked + dpd_patch_rows_fwd_scaled):
    -> pts [2B,N,3] (encoder input, kept for the backward), X [2BN,KP], mask, vox.  Same bits as stack_clouds + mfv3d_fwd +
    patch_rows_fwd; the un-normalised Fisher vectors and the per-slice norms stay internal."""
    L.req(pcA, name="pcA"), L.req(pcB, name="pcB")
    if noise is not None:
        L.req(noise, name="add_noise")
    B, N, _ = pcA.shape
    dev, lib, s = pcA.device, L.load(), L.cur_stream()
    KP = KP or padded_width(k)
    f = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)   # noqa: E731
    pts, q, fv, ssq = f(2 * B, N, 3), f(2 * B, N, 3), f(2 * B, m ** 3, F), f(2 * B, 4, F)
    X, mask, vox = f(2 * B * N, KP), f(2 * B * N), torch.empty(2 * B * N, device=dev, dtype=torch.int32)
    L.check(lib.dpd_mfv3d_fwd_stacked(L.ptr(pcA), L.ptr(pcB), L.ptr(noise), B, N, m, float(sigma), L.ptr(pts), L.ptr(q), L.ptr(fv),
                                      L.ptr(ssq), s), "dpd_mfv3d_fwd_stacked")
    L.check(lib.dpd_patch_rows_fwd_scaled(L.ptr(q), L.ptr(fv), L.ptr(ssq), 2 * B, N, m, k, KP, L.ptr(X), L.ptr(mask), L.ptr(vox),
                                          None, s), "dpd_patch_rows_fwd_scaled")
    return pts, X, mask, vox


def patch_rows_fwd(q, fv, m, k, KP=None, out=None):
    """q [C,N,3], fv [C,m^3,20] -> X [C*N,KP], mask [C*N], vox [C*N] int32   (:911-930, :459-492, :434-457)"""
    L.req(q, name="q"), L.req(fv, name="fv")
    C, N, _ = q.shape
    KP = KP or padded_width(k)
    if out is None:
        X = torch.empty(C * N, KP, device=q.device, dtype=torch.float32)
        mask = torch.empty(C * N, device=q.device, dtype=torch.float32)
        vox = torch.empty(C * N, device=q.device, dtype=torch.int32)
    else:
        X, mask, vox = out
    L.check(L.load().dpd_patch_rows_fwd(L.ptr(q), L.ptr(fv), C, N, m, k, KP, L.ptr(X), L.ptr(mask), L.ptr(vox),
                                        None, L.cur_stream()), "dpd_patch_rows_fwd")
    return X, mask, vox


def patch_rows_bwd(dX, vox, C, N, m, k, want_dq=True, want_dfv=True):
    L.req(dX, name="dX"), L.req(vox, torch.int32, "vox")
    KP = dX.shape[1]
    dq = torch.empty(C, N, 3, device=dX.device, dtype=torch.float32) if want_dq else None
    dfv = torch.empty(C, m ** 3, F, device=dX.device, dtype=torch.float32) if want_dfv else None
    L.check(L.load().dpd_patch_rows_bwd(L.ptr(dX), L.ptr(vox), C, N, m, k, KP, L.ptr(dq), L.ptr(dfv), L.cur_stream()),
            "dpd_patch_rows_bwd")
    return dq, dfv


def asloss_combine(dpts, dX, scale, B, N, k):
    """dpts [2B,N,3], dX [2BN,KP], scale = 0-dim device tensor (upstream gradient) -> (d loss / d pcA, d loss / d pcB) [B,N,3]"""
    L.req(dpts, name="dpts"), L.req(dX, name="dX")
    gA = torch.empty(B, N, 3, device=dpts.device, dtype=torch.float32)
    gB = torch.empty_like(gA)
    sc = None
    if scale is not None:
        sc = scale.reshape(1).to(torch.float32).contiguous()
    L.check(L.load().dpd_asloss_combine(L.ptr(dpts), L.ptr(dX), L.ptr(sc), B, N, k, dX.shape[1], L.ptr(gA), L.ptr(gB), L.cur_stream()),
            "dpd_asloss_combine")
    return gA, gB


def _ws_args(ws):
    return (L.ptr(ws), ws.numel() * ws.element_size()) if ws is not None else (None, 0)


def decoder_fwd(X, mask, params, H, bufs=None, dtype=0, ws=None, out_layer=True, planes=None):
    """X [Q,KP] -> (h1,h2,h3 [Q,H], y [Q,3], pred [Q,3])   (:513-544, :691, :695-698); out_layer=False: y = pred = None (left to
    out_asloss / the fused output-layer kernel of the backward).  planes (AsLossPlanes, plane compute types): the rows come from
    planes.X_rc (X may be None), h1 / h2 leave as bf16 planes only (returned as None), h3 stays fp32"""
    L.req(mask, name="mask")
    if planes is not None:
        Q, KP = planes.Q, planes.KP
        dev = mask.device
    else:
        L.req(X, name="X")
        Q, KP = X.shape
        dev = X.device
    if bufs is None:
        h1, h2 = (None, None) if planes is not None else tuple(torch.empty(Q, H, device=dev, dtype=torch.float32) for _ in range(2))
        h3 = torch.empty(Q, H, device=dev, dtype=torch.float32)
        y = torch.empty(Q, 3, device=dev, dtype=torch.float32) if out_layer else None
        pred = torch.empty(Q, 3, device=dev, dtype=torch.float32) if out_layer else None
    else:
        h1, h2, h3, y, pred = bufs
    p = params if isinstance(params, L.DecoderParams) else L.make_params(*params)
    dtype = L.DTYPES[dtype]
    if dtype and ws is None and planes is None:
        ws = workspace(Q, KP, H, dev, dtype)
    L.check(L.load().dpd_decoder_fwd(L.ptr(X), L.ptr(mask), Q, KP, H, p, dtype, L.ptr(h1), L.ptr(h2), L.ptr(h3), L.ptr(y),
                                     L.ptr(pred), *_ws_args(ws), planes.c if planes is not None else None, L.cur_stream()), "dpd_decoder_fwd")
    return h1, h2, h3, y, pred


class AsLossPlanes:
    """bf16 operand planes of ONE as-loss evaluation (plane compute types f32x3 / bf16): the gathered rows, h1, h2 and the
    pre-activation gradients live as RC planes only (there are no weight gradients to feed, so no R8 copies), allocated per call
    (autograd may hold several evaluations); the weights are frozen, so their planes (R8 for the forward, RC for the data
    gradients) are converted once and cached on the parameter object until its buffer changes."""

    def __init__(self, P, flat, Q, dtype, device):
        dt = L.DTYPES[dtype]
        self.np = 3 if dt == 1 else 1
        self.Q, self.KP, self.H = Q, P.KP, P.H
        e = lambda rows, cols: torch.empty(self.np * rows * cols, device=device, dtype=torch.int16)   # noqa: E731
        # one allocation for the six activation / gradient planes (the host, not the GPU, bounds this node at small batches)
        sizes = (("X_rc", Q * P.KP), ("h1_rc", Q * P.H), ("h2_rc", Q * P.H), ("g3_rc", Q * P.H), ("g2_rc", Q * P.H), ("g1_rc", Q * P.H))
        self.arena = torch.empty(self.np * sum(n for _, n in sizes), device=device, dtype=torch.int16)
        base, off, self.act = self.arena.data_ptr(), 0, {}
        for n, cnt in sizes:
            self.act[n] = base + 2 * off
            off += self.np * cnt
        key = (flat.data_ptr(), flat._version, dt)
        cache = getattr(P, "_wplanes", None)
        if cache is None or cache[0] != key:
            w = {"W1_r8": e(P.KP, P.H), "W2_r8": e(P.H, P.H), "W3_r8": e(P.H, P.H), "W1_rc": e(P.KP, P.H), "W2_rc": e(P.H, P.H), "W3_rc": e(P.H, P.H)}
            c = L.Planes()
            c.np, c.Q, c.Qb = self.np, Q, Q
            for n, t in w.items():
                setattr(c, n, t.data_ptr())
            L.check(L.load().dpd_weights_to_planes(L.make_params(*P.views(flat)), P.KP, P.H, c, L.cur_stream()), "dpd_weights_to_planes")
            cache = P._wplanes = (key, w)
        self.w = cache[1]
        self.c = L.Planes()
        self.c.np, self.c.Q, self.c.Qb = self.np, Q, Q
        for n, ptr_ in self.act.items():
            setattr(self.c, n, ptr_)
        for n, t in self.w.items():
            setattr(self.c, n, t.data_ptr())

    @staticmethod
    def usable(P, Q, dtype):
        return L.DTYPES[dtype] != 0 and Q % 32 == 0 and P.KP % 32 == 0 and P.H % 64 == 0


def front_end_planes(pcA, pcB, m, sigma, k, planes):
    """front_end for the plane compute types of the as-loss node: the window gather writes the rows straight into planes.X_rc (no fp32
    X at all: its backward needs only dX, vox).  -> pts, mask, vox"""
    L.req(pcA, name="pcA"), L.req(pcB, name="pcB")
    B, N, _ = pcA.shape
    dev, lib, s = pcA.device, L.load(), L.cur_stream()
    f = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)   # noqa: E731
    pts, q, fv, ssq = f(2 * B, N, 3), f(2 * B, N, 3), f(2 * B, m ** 3, F), f(2 * B, 4, F)
    mask, vox = f(2 * B * N), torch.empty(2 * B * N, device=dev, dtype=torch.int32)
    L.check(lib.dpd_mfv3d_fwd_stacked(L.ptr(pcA), L.ptr(pcB), None, B, N, m, float(sigma), L.ptr(pts), L.ptr(q), L.ptr(fv), L.ptr(ssq), s),
            "dpd_mfv3d_fwd_stacked")
    L.check(lib.dpd_patch_rows_fwd_scaled(L.ptr(q), L.ptr(fv), L.ptr(ssq), 2 * B, N, m, k, planes.KP, None, L.ptr(mask), L.ptr(vox),
                                          planes.c, s), "dpd_patch_rows_fwd_scaled")
    return pts, mask, vox


def stack_clouds(pcA, pcB, noise=None):
    """-> pts [2B,N,3] = [pcA+noise ; pcB], q [2B,N,3] = [pcB ; pcA]   (dpdist_and_aue.py:45,56-61,69)"""
    L.req(pcA, name="pcA"), L.req(pcB, name="pcB")
    if noise is not None:
        L.req(noise, name="add_noise")
    B, N, _ = pcA.shape
    pts = torch.empty(2 * B, N, 3, device=pcA.device, dtype=torch.float32)
    q = torch.empty_like(pts)
    L.check(L.load().dpd_stack_clouds(L.ptr(pcA), L.ptr(pcB), L.ptr(noise), B, N, L.ptr(pts), L.ptr(q), L.cur_stream()),
            "dpd_stack_clouds")
    return pts, q


_OUT_SCRATCH = {}


def out_asloss(h3, mask, params, BN, want_grad=True, gscale=1.0):
    """Output layer + loss_pred (+ its output-layer backward) of the as-loss mode in one launch (include/dpdist_capi.h:
    dpd_decoder_out_asloss).  h3 [2*BN,H] -> y, pred [2*BN,3], loss_pred [1], dy [2*BN,3] | None, g3 [2*BN,H] | None"""
    L.req(h3, name="h3"), L.req(mask, name="mask")
    Q, H = h3.shape
    dev = h3.device
    f = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)   # noqa: E731
    y, pred, loss = f(Q, 3), f(Q, 3), f(1)
    dy, g3 = (f(Q, 3), f(Q, H)) if want_grad else (None, None)
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    scr = _OUT_SCRATCH.get(key)
    if scr is None:          # the 64-bit accumulator (zero once; the kernel leaves it zero); one per stream
        if len(_OUT_SCRATCH) > 16:
            _OUT_SCRATCH.clear()
        scr = _OUT_SCRATCH[key] = torch.zeros(2, device=dev, dtype=torch.float32)
    L.check(L.load().dpd_decoder_out_asloss(L.ptr(h3), L.ptr(mask), Q, H, BN, params if isinstance(params, L.DecoderParams) else L.make_params(*params), float(gscale), L.ptr(y), L.ptr(pred),
                                            L.ptr(loss), L.ptr(dy), L.ptr(g3), L.ptr(scr), L.cur_stream()), "dpd_decoder_out_asloss")
    return y, pred, loss, dy, g3


def decoder_bwd_data(dpred, mask, y, h1, h2, h3, params, KP, want_dX, bufs=None, small_grads=None, dtype=0, ws=None,
                     transposed=None, phases=7, g3=None, planes=None):
    """dpred [Qb,3] (first Qb rows) -> dy [Qb,3], g3,g2,g1 [Qb,H], dX [Qb,KP] or None.
    small_grads = (db1, db2, db3, dW4, db4) tensors (or None each) to be filled by the fused epilogues.
    phases=6 with g3 given: the output layer was done by out_asloss (dpred / y / h3 may be None)."""
    if phases & 1:
        L.req(dpred, name="dpred")
    Qb = dpred.shape[0] if dpred is not None else g3.shape[0]
    H = h1.shape[1] if h1 is not None else planes.H
    dev = h1.device if h1 is not None else g3.device
    if bufs is None and g3 is not None:
        dy = None
        # (planes: g2 / g1 travel as RC planes only)
        g2, g1 = (None, None) if planes is not None else tuple(torch.empty(Qb, H, device=dev, dtype=torch.float32) for _ in range(2))
        dX = torch.empty(Qb, KP, device=dev, dtype=torch.float32) if want_dX else None
    elif bufs is None:
        dy = torch.empty(Qb, 3, device=dev, dtype=torch.float32)
        g3, g2, g1 = (torch.empty(Qb, H, device=dev, dtype=torch.float32) for _ in range(3))
        dX = torch.empty(Qb, KP, device=dev, dtype=torch.float32) if want_dX else None
    else:
        dy, g3, g2, g1, dX = bufs
    p = params if isinstance(params, L.DecoderParams) else L.make_params(*params, *(transposed if transposed is not None else ()))
    sg = L.make_small_grads(*small_grads) if small_grads is not None else None
    dtype = L.DTYPES[dtype]
    if dtype and ws is None and planes is None:
        ws = workspace(Qb, KP, H, dev, dtype)
    L.check(L.load().dpd_decoder_bwd_data(L.ptr(dpred), L.ptr(mask), L.ptr(y), L.ptr(h1), L.ptr(h2), L.ptr(h3), Qb, KP, H,
                                          p, dtype, L.ptr(dy), L.ptr(g3), L.ptr(g2), L.ptr(g1), L.ptr(dX), sg, *_ws_args(ws),
                                          planes.c if planes is not None else None, phases, L.cur_stream()), "dpd_decoder_bwd_data")
    return dy, g3, g2, g1, dX


def workspace(Q, KP, H, device, dtype=0):
    n = L.load().dpd_workspace_bytes(Q, KP, H, L.DTYPES[dtype])
    return torch.empty((n + 3) // 4, device=device, dtype=torch.float32)


def decoder_bwd_weights(layer, act, g, Qb, dW, db, ws, dtype=0):
    """dW/db of one layer from its input activation `act` [>=Qb, Kin] and output gradient `g` [Qb, Nout]."""
    Kin, Nout = dW.shape
    L.check(L.load().dpd_decoder_bwd_weights(layer, L.ptr(act), act.stride(0), L.ptr(g), Qb, Kin, Nout, L.DTYPES[dtype], L.ptr(dW),
                                             L.ptr(db), L.ptr(ws), ws.numel() * 4, None, None, L.cur_stream()),
            "dpd_decoder_bwd_weights(layer=%d)" % layer)


def l1_loss(pred, labels, mode=0, gscale=1.0, dpred=None, loss=None):
    """pred [2*BN,3], labels [BN] -> loss [2] = (loss_samples, loss_pred); dpred per `mode` (:962-980)"""
    L.req(pred, name="pred"), L.req(labels, name="labels")
    BN = labels.numel()
    if loss is None:
        loss = torch.empty(2, device=pred.device, dtype=torch.float32)
    if mode and dpred is None:
        dpred = torch.empty(BN * (1 if mode == 1 else 2), 3, device=pred.device, dtype=torch.float32)
    L.check(L.load().dpd_l1_loss(L.ptr(pred), L.ptr(labels), BN, mode, float(gscale), L.ptr(loss), L.ptr(dpred),
                                 L.cur_stream()), "dpd_l1_loss")
    return loss, dpred


def adam_tf(p, g, m, v, lr_t, b1=0.9, b2=0.999, eps=1e-8, gscale=1.0):
    L.check(L.load().dpd_adam_tf(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), float(lr_t), b1, b2, eps,
                                 float(gscale), L.cur_stream()), "dpd_adam_tf")


def gemm_f32(A, B, transA=False, transB=False, bias=None, gate=None, epilogue=0, split_k=1, tile=0, out=None):
    """C = epi(op(A) op(B)) on the fp32 MFMA kernel (building block; see include/dpdist_capi.h)."""
    L.req(A, name="A"), L.req(B, name="B")
    M = A.shape[1] if transA else A.shape[0]
    K = A.shape[0] if transA else A.shape[1]
    N = B.shape[0] if transB else B.shape[1]
    C = out if out is not None else torch.empty(M, N, device=A.device, dtype=torch.float32)
    ws = torch.empty(split_k * M * N if split_k > 1 else 1, device=A.device, dtype=torch.float32)
    L.check(L.load().dpd_gemm_f32(int(transA), int(transB), M, N, K, L.ptr(A), A.stride(0), L.ptr(B), B.stride(0),
                                  L.ptr(C), C.stride(0), L.ptr(bias), L.ptr(gate), epilogue, split_k, tile, L.ptr(ws),
                                  ws.numel() * 4, L.cur_stream()), "dpd_gemm_f32")
    return C


def set_gemm_plan(op, tile, split_k=1):
    L.check(L.load().dpd_set_gemm_plan(op, tile, split_k), "dpd_set_gemm_plan")
