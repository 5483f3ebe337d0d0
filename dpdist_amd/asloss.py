"""DPDist-as-a-loss on the library's engine (include/dpdist_capi.h: dpd_asloss): ONE foreign call per direction on buffers sized once.

The reference splices the trained DPDist graph into a consumer's graph and differentiates THROUGH it to the consumer's own variables
(pcrnet-registration/iterative_PCRNet_ours.py:229-257: import_meta_graph with input_map, loss = mean of the two output means, gradients of
scope 'Network' only; train_multi_gpu_pc_compare_dist.py:427-463 for the AUE task); `dpdist_amd.model._AsLossFn` is that node for PyTorch
autograd.  Until round 4 it drove the C ABI entry by entry from Python -- about 30 tensor allocations and a dozen ctypes calls per
evaluation, which made the plane compute types HOST-bound at the registration batch (0.37 ms per evaluation over 0.20 ms of kernels;
ONE forward + backward per registration training step -- the seven refinements before it fetch the pose only,
pcrnet-registration/iterative_PCRNet_ours.py:414-441 -- and one forward per evaluation batch).  An `Engine` owns one caller-side allocation carved by dpd_asloss_carve and calls
dpd_asloss_forward / dpd_asloss_backward: same kernels, same order, same bits.

Autograd may keep several evaluations alive (two losses before one backward; retain_graph): an engine that holds the state of a node
whose backward has not run is BUSY, `acquire` hands out another one (up to `MAX_ENGINES` per shape and stream, then the caller falls back
to the allocating path), a node that is dropped without a backward frees its engine through a weakref finalizer, and a backward that
finds its engine re-used by ANY later forward (a second backward long after the first, with or without gradients in between) raises
instead of returning another evaluation's gradient.

The per-evaluation outputs (`loss`, `gA`, `gB`: 4 + 2 x 12 B N bytes) are still torch allocations, on purpose: they outlive the call in
the caller's hands (a list of losses, an accumulated gradient), so they cannot be the engine's own re-used buffers.  Everything else --
about 30 intermediates -- lives in the engine.  Idle engines are bounded by `MAX_POOL_BYTES` per parameter set (least recently used
shapes are dropped first); `release_all(P)` drops them all.

hipGraph capture (registration.IterativeRegistration): `private_pool(d)` makes `acquire` draw from a pool the graph owns, keyed without
the stream, so the engine whose buffers the captured launches point into is created (and the frozen weights derived) before the capture
and is never handed to anybody else.
"""
import contextlib
import weakref
from ctypes import c_void_p

import torch

from . import lib as L

MAX_ENGINES = 4
MAX_POOL_BYTES = 4 << 30      # idle + busy engines of one parameter set
# module switches (tests and tools/asloss_bench.py flip them; no environment variables): the engine itself, and -- for the plane compute
# types without it -- the node on persistent planes (round 4) against the round-3 form that converted both operands of every GEMM
ENGINE = True
PLANES = True

_private = None


@contextlib.contextmanager
def private_pool(pool):
    """Inside: `acquire` uses `pool` (a dict the caller keeps alive as long as its captured graph) instead of the parameter set's."""
    global _private
    old, _private = _private, pool
    try:
        yield pool
    finally:
        _private = old


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
        self.nbytes = nbytes
        self.mem = torch.empty(nbytes + 256, device=device, dtype=torch.uint8)
        base = (self.mem.data_ptr() + 255) // 256 * 256
        self.c = AsLoss()
        L.check(lib.dpd_asloss_carve(c_void_p(base), nbytes, B, N, m, k, P.H, dt, float(sigma), self.c), "dpd_asloss_carve")
        L.check(lib.dpd_asloss_init(self.c, L.cur_stream()), "dpd_asloss_init")
        self.B, self.N, self.device = B, N, device
        self.version = 0            # bumped by EVERY forward: any of them overwrites what an earlier node's backward would read
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

    def forward(self, pcA, pcB, want_grad):
        loss = torch.empty(1, device=self.device, dtype=torch.float32)
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
    return ENGINE


def acquire(P, flat, B, N, m, k, sigma, device):
    """An idle engine for this shape on the current stream with the weights of `flat` in place, or None (disabled, a shape the engine
    does not take, or MAX_ENGINES evaluations already waiting for their backward): the caller then takes the allocating path."""
    if not enabled() or B * N >= 16384:
        return None
    if L.DTYPES[P.compute_dtype] != 0 and not PLANES:
        return None                 # A/B reference of round 4: the plane types without persistent planes
    private = _private is not None
    pool = _private if private else P.__dict__.setdefault("_asloss_engines", {})
    key = (B, N, m, k, float(sigma), L.DTYPES[P.compute_dtype], str(device), 0 if private else torch.cuda.current_stream(device).cuda_stream)
    engines = pool.get(key)
    if engines is None:
        engines = pool[key] = []
    elif not private:
        pool[key] = pool.pop(key)           # most recently used shape last (dicts keep insertion order)
    eng = next((e for e in engines if not e.busy), None)
    if eng is None:
        if len(engines) >= MAX_ENGINES:
            return None
        if not private:
            _evict(pool, key, L.load().dpd_asloss_bytes(B, N, m, k, P.H, L.DTYPES[P.compute_dtype]))
        try:
            eng = Engine(P, B, N, m, k, sigma, device)
        except ValueError:
            return None
        engines.append(eng)
    eng.set_weights(P, flat)
    return eng


def _evict(pool, keep_key, incoming):
    """Keep the pool's engines under MAX_POOL_BYTES: drop idle engines of the least recently used shapes first (busy ones hold the
    state of a pending backward and stay)."""
    total = incoming + sum(e.nbytes for engines in pool.values() for e in engines)
    for key in list(pool):
        if total <= MAX_POOL_BYTES:
            break
        if key == keep_key:
            continue
        engines = pool[key]
        for e in [e for e in engines if not e.busy]:
            engines.remove(e)
            total -= e.nbytes
            if total <= MAX_POOL_BYTES:
                break
        if not engines:
            del pool[key]


def pool_bytes(P):
    return sum(e.nbytes for engines in P.__dict__.get("_asloss_engines", {}).values() for e in engines)


def release_all(P):
    """Drop every idle engine of this parameter set (their memory goes back to torch's allocator); busy ones stay until their backward."""
    pool = P.__dict__.get("_asloss_engines", {})
    for key in list(pool):
        pool[key] = [e for e in pool[key] if e.busy]
        if not pool[key]:
            del pool[key]
