"""DPDist-as-a-loss on the library's engine (include/dpdist_capi.h: dpd_asloss): ONE foreign call per direction on buffers sized once.

The reference splices the trained DPDist graph into a consumer's graph and differentiates THROUGH it to the consumer's own variables
(pcrnet-registration/iterative_PCRNet_ours.py:229-257: import_meta_graph with input_map, loss = mean of the two output means, gradients of
scope 'Network' only; train_multi_gpu_pc_compare_dist.py:427-463 for the AUE task); `dpdist_amd.model._AsLossFn` is that node for PyTorch
autograd.  Until round 4 it drove the C ABI entry by entry from Python -- about 30 tensor allocations and a dozen ctypes calls per
evaluation, which made the plane compute types HOST-bound at the registration batch (0.37 ms per evaluation over 0.20 ms of kernels,
8 evaluations per training step).  An `Engine` owns one caller-side allocation carved by dpd_asloss_carve and calls
dpd_asloss_forward / dpd_asloss_backward: same kernels, same order, same bits.

Autograd may keep several evaluations alive (two losses before one backward; retain_graph): an engine that holds the state of a node
whose backward has not run is BUSY, `acquire` hands out another one (up to `MAX_ENGINES` per shape and stream, then the caller falls back
to the allocating path), a node that is dropped without a backward frees its engine through a weakref finalizer, and a backward that
finds its engine re-used (a second backward long after the first) raises instead of returning another evaluation's gradient.
"""
import os
import weakref
from ctypes import c_void_p

import torch

from . import lib as L

MAX_ENGINES = 4


AsLoss = L.AsLoss


def _lib():
    return L.load()


class Engine:
    """One dpd_asloss: buffers for a fixed (B, N, m, k, H, compute type) on one device."""

    def __init__(self, P, B, N, m, k, sigma, device):
        lib = _lib()
        dt = L.DTYPES[P.compute_dtype]
        nbytes = lib.dpd_asloss_bytes(B, N, m, k, P.H, dt)
        if nbytes == 0:
            raise ValueError("shape not taken by the as-loss engine")
        self.mem = torch.empty(nbytes + 256, device=device, dtype=torch.uint8)
        base = (self.mem.data_ptr() + 255) // 256 * 256
        self.c = AsLoss()
        L.check(lib.dpd_asloss_carve(c_void_p(base), nbytes, B, N, m, k, P.H, dt, float(sigma), self.c), "dpd_asloss_carve")
        L.check(lib.dpd_asloss_init(self.c, L.cur_stream()), "dpd_asloss_init")
        self.B, self.N, self.device = B, N, device
        self.version = 0            # bumped by every forward that keeps state for a backward
        self.busy = False
        self._wkey = None
        self._fin = None

    def view(self, name, *shape, dtype=torch.float32):
        """a tensor view of one of the engine's buffers (tests / diagnostics: e.g. view('pred', Q, 3))"""
        ptr = getattr(self.c, name)
        n = 1
        for s in shape:
            n *= s
        off = ptr - self.mem.data_ptr()
        return self.mem[off: off + n * 4].view(dtype).view(*shape)

    def set_weights(self, P, flat):
        key = (flat.data_ptr(), flat._version)
        if key != self._wkey:
            p = L.make_params(*P.views(flat))
            L.check(_lib().dpd_asloss_set_weights(self.c, p, L.cur_stream()), "dpd_asloss_set_weights")
            self._wkey = key
            if self.c.dtype != 0:       # the parameter object's record of "weight planes derived for this buffer / version" (model.py)
                P._wplanes = (key + (self.c.dtype, "engine"), None)     # (never equal to an AsLossPlanes key: that path keeps its own planes)

    def forward(self, pcA, pcB, want_grad):
        loss = torch.empty(1, device=self.device, dtype=torch.float32)
        if want_grad:
            self.version += 1
        L.check(_lib().dpd_asloss_forward(self.c, L.ptr(pcA), L.ptr(pcB), 1 if want_grad else 0, L.ptr(loss), L.cur_stream()),
                "dpd_asloss_forward")
        return loss

    def backward(self, upstream):
        gA = torch.empty(self.B, self.N, 3, device=self.device, dtype=torch.float32)
        gB = torch.empty_like(gA)
        sc = None
        if upstream is not None:
            sc = upstream.reshape(1).to(torch.float32).contiguous()
        L.check(_lib().dpd_asloss_backward(self.c, L.ptr(sc), L.ptr(gA), L.ptr(gB), L.cur_stream()), "dpd_asloss_backward")
        return gA, gB

    # ---- ownership by an autograd node
    def hold(self, ctx):
        self.busy = True
        ver = self.version
        self._fin = weakref.finalize(ctx, self._drop, ver)

    def _drop(self, ver):
        if self.version == ver:
            self.busy = False

    def release(self):
        self.busy = False
        if self._fin is not None:
            self._fin.detach()
            self._fin = None


def enabled():
    return os.environ.get("DPD_ASLOSS_ENGINE", "1") == "1"


def acquire(P, flat, B, N, m, k, sigma, device):
    """An idle engine for this shape on the current stream with the weights of `flat` in place, or None (disabled, a shape the engine
    does not take, or MAX_ENGINES evaluations already waiting for their backward): the caller then takes the allocating path."""
    if not enabled() or B * N >= 16384:
        return None
    if L.DTYPES[P.compute_dtype] != 0 and os.environ.get("DPD_ASLOSS_PLANES", "1") != "1":
        return None                 # A/B reference of round 4: the plane types without persistent planes
    pool = P.__dict__.setdefault("_asloss_engines", {})
    key = (B, N, m, k, float(sigma), L.DTYPES[P.compute_dtype], str(device), torch.cuda.current_stream(device).cuda_stream)
    engines = pool.get(key)
    if engines is None:
        if len(pool) > 8:
            pool.clear()
        engines = pool[key] = []
    eng = next((e for e in engines if not e.busy), None)
    if eng is None:
        if len(engines) >= MAX_ENGINES:
            return None
        try:
            eng = Engine(P, B, N, m, k, sigma, device)
        except ValueError:
            return None
        engines.append(eng)
    eng.set_weights(P, flat)
    return eng
