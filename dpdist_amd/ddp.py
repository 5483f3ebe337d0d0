"""Data parallelism for the DPDist training step: one process per GPU, RCCL all-reduce over xGMI.

Replaces the reference's in-graph towers (`train_multi_gpu_pc_compare_dist.py:237-302`) and its CPU-side
`average_gradients` (`:936-974`: stack + reduce_mean of 8 variables = 18.67 MB per tower per step over PCIe).

Design for MI355X: weights and Adam state are replicated on every GPU; the flat gradient buffer is cut into three
buckets -- bucket 0 = layer 1 (dW1p+db1, 10.3 MB), bucket 1 = layer 2 (4.2 MB), bucket 2 = layers 3-4 (4.2 MB).  The
data-parallel backward of trainer.py produces every weight gradient as soon as its inputs exist (dW3 right after the
output layer, dW2 after g2, dW1 last) and enqueues each bucket's all-reduce(sum) on a side stream the moment its
producer kernels are enqueued: the collectives are serial on the RCCL stream, so what matters is how early the first one
starts -- here ~250 us of backward GEMMs before the end, and only the tail of the last bucket is exposed.  The 1/world
scale is folded into the Adam kernel (`gscale`).  xGMI is point-to-point (ring all-reduce is per-link bound), so three
messages of 4-10 MB beat the reference's eight small ones.

`shard_range` gives rank r the pairs [r*B/P, (r+1)*B/P) like `tf.slice` at `:241-251`.
"""
import torch
import torch.distributed as dist


def shard_range(global_batch, rank, world):
    if global_batch % world:
        raise ValueError("batch %d not divisible by world size %d" % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per


def _wire_bytes(bounds_calls, world, wire, mode):
    """Bytes ONE GPU sends (= receives) per step for the given (lo, hi) calls: a ring all-reduce moves 2 (P-1)/P of the payload per
    GPU, reduce-scatter and all-gather (P-1)/P each (so rs_ag and zero1 with an fp32 parameter gather move the same bytes as the
    all-reduce; what zero1 saves is the replicated optimizer, and the gather half leaves the critical path of the backward)."""
    el = 2 if wire == "bf16" else 4
    f = (world - 1) / float(world) if world > 1 else 0.0
    return int(sum(2 * f * (hi - lo) * el for lo, hi in bounds_calls))


def zero_partition(lo, hi, world):
    """zero1: [lo, hi) = `world` equal shards of a multiple of 4 elements (float4 alignment of the optimizer kernel survives) + a
    replicated remainder of < 4*world elements that is all-reduced and updated on every rank.  Returns (main, shard)."""
    main = (hi - lo) - (hi - lo) % (4 * world)
    return main, main // world


class _Exposure:
    """In-stream timing of the places where the COMPUTE stream waits for a collective (opt-in: `reducer.measure = True`; a timing
    event record costs the stream a few us, so this runs in a separate pass, never in a timed region)."""

    def __init__(self):
        self.measure = False
        self._pairs = []

    def begin(self):
        if not self.measure:
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return e0

    def end(self, e0):
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self._pairs.append((e0, e1))

    def collect_ms(self):
        """(number of waits, total ms) since the last call; host sync."""
        torch.cuda.synchronize()
        tot = sum(a.elapsed_time(b) for a, b in self._pairs)
        n = len(self._pairs)
        self._pairs = []
        return n, tot


class BucketReducer:
    """All-reduce slices of one flat gradient buffer, asynchronously with respect to the compute stream.

    The collective is issued with async_op=True from the compute stream: ProcessGroupNCCL orders its internal RCCL stream
    after everything already enqueued on the current stream (the producers of the bucket) and `Work.wait()` later makes
    the current stream -- not the host -- wait for it, so one event hop each way is all the synchronisation there is.

    wire = "f32" (default) | "bf16": dtype on the links.  "bf16" halves the bytes per step (18.7 -> 9.3 MB): each bucket is
        rounded into a bf16 staging buffer, reduced, and written back to the fp32 gradient (fp32 master weights and Adam state;
        the cross-rank sum itself is taken in bf16, 2^-8 relative per addend) -- meant for the bf16 compute type, whose step is
        short enough (0.40 ms at 64 pairs) for the fp32 all-reduce to show.
    mode = "allreduce" (default) | "rs_ag" | "zero1":
        rs_ag: reduce-scatter followed by all-gather of the same bucket (what a ring all-reduce does internally, as two collectives:
        the scatter half can start while later buckets are still being produced and the gather half of every bucket is deferred to
        `wait()`, i.e. to just before the optimizer).
        zero1: the optimizer is SHARDED (ZeRO stage 1): every bucket is reduce-scattered; after `wait()` this rank holds the summed
        gradient only on `owned_ranges()` (its shard of every bucket + the replicated remainders), the trainer runs Adam on those
        ranges only (1/P of the optimizer traffic: 150 MB -> 19 MB per step at P = 8 in the bf16 step) and `gather_params(flat)`
        all-gathers the updated fp32 parameters.  Same wire bytes as the all-reduce; bit-identical parameters (elementwise Adam on
        the same sums).  fp32 wire only.
    Expected exposed time on 8 x MI355X (xGMI ring, ~7 x 153 GB/s per GPU, all-reduce moves 2 (P-1)/P of the bytes per link):
    fp32 18.7 MB -> ~33 MB per GPU on the wire ~ 40-60 us, of which only the layer-1 bucket's tail (issued last) cannot hide
    under the remaining backward; bf16 halves it.  None of this could be measured here (one-GPU boxes): see DESIGN.md."""

    backend = "torch"

    def __init__(self, flat_grad, bounds, group=None, force=False, wire=None, mode=None):
        import os
        self.flat = flat_grad
        self.bounds = list(bounds)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # force=True issues the collectives even for a single rank (exercises the RCCL/stream plumbing on one GPU)
        self.active = dist.is_initialized() and (self.world > 1 or force)
        self.wire = wire or os.environ.get("DPD_DP_WIRE", "f32")
        self.mode = mode or os.environ.get("DPD_DP_MODE", "allreduce")
        if self.wire not in ("f32", "bf16") or self.mode not in ("allreduce", "rs_ag", "zero1"):
            raise ValueError("wire must be f32|bf16 and mode allreduce|rs_ag|zero1, got %r / %r" % (self.wire, self.mode))
        if self.mode == "zero1" and self.wire != "f32":
            raise ValueError("zero1 shards fp32 master weights: fp32 wire only")
        self._pending = []
        self._stage = {}            # bucket -> (staging buffer [padded], shard buffer) for the bf16 wire / rs_ag mode
        self._calls = []            # (lo, hi) of this step's reduce_async calls (zero1: what owned_ranges / gather_params cover)
        self._fresh = True
        self.exposure = _Exposure()
        self.crosscheck = None      # filled by make_reducer: {"ok": bool, ...}

    @property
    def measure(self):
        return self.exposure.measure

    @measure.setter
    def measure(self, on):
        self.exposure.measure = bool(on) and self.flat.is_cuda

    @property
    def nranks(self):
        return self.world

    @property
    def wire_bytes_per_step(self):
        """bytes each GPU sends per step for the schedule of the LAST step (2 (P-1)/P of the payload: module docstring)"""
        return _wire_bytes(self._calls, self.world, self.wire, self.mode)

    def _staging(self, bucket, n):
        if bucket not in self._stage:
            dt = torch.bfloat16 if self.wire == "bf16" else torch.float32
            pad = (n + self.world - 1) // self.world * self.world
            full = torch.zeros(pad, device=self.flat.device, dtype=dt)
            shard = torch.empty(pad // self.world, device=self.flat.device, dtype=dt) if self.mode == "rs_ag" else None
            self._stage[bucket] = (full, shard)
        return self._stage[bucket]

    def reduce_async(self, bucket, upto=None):
        """Call right after the kernels producing bucket `bucket` (or the contiguous buckets bucket .. upto) were enqueued on
        the current stream."""
        if not self.active:
            return
        if self._fresh:
            self._calls, self._fresh = [], False
        lo, hi = self.bounds[bucket], self.bounds[(bucket if upto is None else upto) + 1]
        self._calls.append((lo, hi))
        bucket = (bucket, upto)
        g = self.flat[lo:hi]
        if self.mode == "zero1":
            main, shard = zero_partition(lo, hi, self.world)
            if main:
                if bucket not in self._stage:
                    self._stage[bucket] = torch.empty(shard, device=self.flat.device, dtype=torch.float32)
                out = self._stage[bucket]
                h = dist.reduce_scatter_tensor(out, self.flat[lo:lo + main], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self._pending.append((h, ("zero", self.flat[lo + self.rank * shard:lo + (self.rank + 1) * shard], out)))
            if main < hi - lo:
                self._pending.append((dist.all_reduce(self.flat[lo + main:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True), None))
            return
        if self.wire == "f32" and self.mode == "allreduce":
            self._pending.append((dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True), None))
            return
        full, shard = self._staging(bucket, hi - lo)
        full[:hi - lo].copy_(g)                                     # (rounds to bf16 on the bf16 wire)
        if self.mode == "allreduce":
            h = dist.all_reduce(full, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._pending.append((h, (g, full, None)))
        else:
            h = dist.reduce_scatter_tensor(shard, full, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._pending.append((h, (g, full, shard)))

    def wait(self):
        """Make the current stream (or the host, for CPU tensors) wait for every outstanding bucket; finishes the two-step
        forms (all-gather of the reduced shards, copy back into the fp32 gradient)."""
        gathers = []
        e0 = self.exposure.begin() if self._pending else None
        for h, extra in self._pending:
            h.wait()
            if extra is not None and extra[0] == "zero":             # zero1: the reduced shard goes back into its place
                extra[1].copy_(extra[2])
            elif extra is not None and extra[2] is not None:         # rs_ag: second half
                g, full, shard = extra
                gathers.append((dist.all_gather_into_tensor(full, shard, group=self.group, async_op=True), g, full))
            elif extra is not None:
                g, full, _ = extra
                g.copy_(full[:g.numel()])
        for h, g, full in gathers:
            h.wait()
            g.copy_(full[:g.numel()])
        self.exposure.end(e0)
        self._pending = []
        self._fresh = True

    def wait_side(self):
        """Like wait(), but for a SIDE stream: returns a stream on which the reduced gradient is complete, without making the compute
        stream wait (the trainer runs the optimizer there, under the next step's front end: trainer.apply_gradients).  None when
        this reducer has no such stream (CPU tensors, inactive)."""
        if not self.active or not self.flat.is_cuda:
            self.wait()
            return None
        if getattr(self, "_opt_side", None) is None:
            from .hipevents import LightEvent
            self._opt_side, self._ev_side = torch.cuda.Stream(device=self.flat.device), LightEvent()
        self._ev_side.record(torch.cuda.current_stream())
        self._ev_side.wait(self._opt_side)
        with torch.cuda.stream(self._opt_side):
            self.wait()                      # Work.wait() orders the CURRENT stream (= the side stream) behind the collectives
        return self._opt_side

    # -- zero1 -----------------------------------------------------------------------------------------------------------
    def owned_ranges(self):
        """zero1, after wait(): the (lo, hi) element ranges of the flat buffers whose gradient sum this rank holds and whose
        parameters it must update: its shard of every reduce_async call + the replicated remainders."""
        out = []
        for lo, hi in self._calls:
            main, shard = zero_partition(lo, hi, self.world)
            if main:
                out.append((lo + self.rank * shard, lo + (self.rank + 1) * shard))
            if main < hi - lo:
                out.append((lo + main, hi))
        return out

    def gather_params(self, flat):
        """zero1, after the sharded optimizer: all-gather the updated shards of `flat` (same layout as the gradient buffer) on the
        current stream."""
        e0 = self.exposure.begin()
        for lo, hi in self._calls:
            main, shard = zero_partition(lo, hi, self.world)
            if main:
                mine = flat[lo + self.rank * shard:lo + (self.rank + 1) * shard]
                if flat.is_cuda:
                    dist.all_gather_into_tensor(flat[lo:lo + main], mine, group=self.group)     # in place (NCCL/RCCL)
                else:
                    full = torch.empty(main, dtype=flat.dtype)
                    dist.all_gather_into_tensor(full, mine.clone(), group=self.group)
                    flat[lo:lo + main].copy_(full)
        self.exposure.end(e0)

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def close(self):
        pass


# ----------------------------------------------------------------------------------------------------------------------
# RCCL driven directly (ctypes on the librccl torch already loaded): the collectives run on streams WE choose.
# ----------------------------------------------------------------------------------------------------------------------
class _Rccl:
    """Minimal binding: ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy."""
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            import ctypes
            import os
            cand = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so", "librccl.so.1"]
            err = None
            for c in cand:
                try:
                    cls._lib = ctypes.CDLL(c)
                    break
                except OSError as e:
                    err = e
            if cls._lib is None:
                raise RuntimeError("librccl.so not found: %r" % (err,))

            class UniqueId(ctypes.Structure):
                _fields_ = [("internal", ctypes.c_char * 128)]
            cls.UniqueId = UniqueId
            L = cls._lib
            L.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
            L.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
            L.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.c_void_p]
            L.ncclCommDestroy.argtypes = [ctypes.c_void_p]
            L.ncclCommCount.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
            L.ncclReduceScatter.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                            ctypes.c_void_p]
            L.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
            L.ncclGetErrorString.restype = ctypes.c_char_p
            L.ncclGetErrorString.argtypes = [ctypes.c_int]
        return cls._lib

    @classmethod
    def check(cls, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, cls.lib().ncclGetErrorString(rc).decode()))

    @classmethod
    def new_comm(cls, group, device):
        """One communicator over the ranks of `group`; the unique id travels through the existing process group."""
        import ctypes
        L = cls.lib()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        uid = cls.UniqueId()
        rc0 = L.ncclGetUniqueId(ctypes.byref(uid)) if rank == 0 else 0
        # rank 0's status travels with the id: every rank takes part in the broadcast and every rank sees the same failure
        t = torch.frombuffer(bytearray(bytes(uid) + bytes([1 if rc0 == 0 else 0])), dtype=torch.uint8).to(device)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = t.cpu().numpy().tobytes()
        if raw[128] != 1:
            raise RuntimeError("ncclGetUniqueId failed on rank 0")
        ctypes.memmove(ctypes.byref(uid), raw[:128], 128)
        comm = ctypes.c_void_p()
        cls.check(L.ncclCommInitRank(ctypes.byref(comm), world, uid, rank), "ncclCommInitRank")
        return comm

    @classmethod
    def count(cls, comm):
        """number of ranks as the communicator itself reports it (ncclCommCount)"""
        import ctypes
        n = ctypes.c_int(-1)
        cls.check(cls.lib().ncclCommCount(comm, ctypes.byref(n)), "ncclCommCount")
        return n.value


class DirectRcclReducer:
    """BucketReducer's contract (reduce_async / wait / grad_scale / zero1) with RCCL called directly.

    `torch.distributed` puts every collective on ProcessGroupNCCL's own stream and orders it with events recorded in the middle
    of the compute stream -- on this runtime a barrier packet with a system-scope release, ~20 us of idle compute stream each, and
    a cross-queue hop (~14 us) each way for the LAST bucket, which nothing hides (+52..57 us per step before a
    single byte moves).  Here every collective runs on ONE side stream through ONE communicator, behind a
    `hipEventDisableSystemFence` event: recording it costs the compute stream ~0.3 us, the hop is paid by the side stream, and
    `wait()` joins the side stream with one more light event.

    (A second communicator for the last bucket on the compute stream itself -- the round-3 form, -14 us on one rank, two RCCL kernels of
    different communicators in flight on one device -- never ran on more than one GPU and was removed in round 6.)

    wire = "f32" | "bf16" as in BucketReducer (staging copies on the stream of the collective); mode = "allreduce" | "zero1"
    (ncclReduceScatter in place; `gather_params` = ncclAllGather in place on the side stream).  One rank (force) works: RCCL's
    single-rank collectives are copies."""

    backend = "rccl"

    def __init__(self, flat_grad, bounds, group=None, wire=None, mode=None):
        import os
        from .hipevents import LightEvent
        if not (dist.is_initialized() and flat_grad.is_cuda):
            raise RuntimeError("DirectRcclReducer needs an initialised process group and a GPU gradient buffer")
        self.flat, self.bounds, self.group = flat_grad, list(bounds), group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.active = True
        self.wire = wire or os.environ.get("DPD_DP_WIRE", "f32")
        self.mode = mode or os.environ.get("DPD_DP_MODE", "allreduce")
        if self.wire not in ("f32", "bf16"):
            raise ValueError("wire must be f32|bf16")
        if self.mode not in ("allreduce", "zero1"):
            raise ValueError("DirectRcclReducer: mode must be allreduce|zero1 (rs_ag lives in BucketReducer)")
        if self.mode == "zero1" and self.wire != "f32":
            raise ValueError("zero1 shards fp32 master weights: fp32 wire only")
        dev = flat_grad.device
        self._side = torch.cuda.Stream(device=dev)
        self._comm_side = _Rccl.new_comm(group, dev)
        self.nranks = _Rccl.count(self._comm_side)      # as the communicator reports it, not as the environment claims
        if self.nranks != self.world:
            raise RuntimeError("ncclCommCount says %d ranks, the process group %d" % (self.nranks, self.world))
        self._ev_fork = [LightEvent() for _ in range(6)]
        self._ev_join = LightEvent()
        self._nfork = 0
        self._covered = 0           # elements reduced so far in this step
        self._side_used = False
        self._stage = {}
        self._calls, self._fresh = [], True
        self.exposure = _Exposure()
        self.crosscheck = None

    measure = BucketReducer.measure
    wire_bytes_per_step = BucketReducer.wire_bytes_per_step
    owned_ranges = BucketReducer.owned_ranges

    def _allreduce(self, t, comm, stream):
        dt = 9 if t.dtype == torch.bfloat16 else 7          # ncclBfloat16 / ncclFloat
        _Rccl.check(_Rccl.lib().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), dt, 0, comm, stream.cuda_stream), "ncclAllReduce")

    def _staging(self, key, n):
        if key not in self._stage:
            self._stage[key] = torch.zeros(n, device=self.flat.device, dtype=torch.bfloat16)
        return self._stage[key]

    def _fork(self, main):
        """the side stream continues after everything enqueued on `main` so far (light event: ~0.3 us on the compute stream)"""
        ev = self._ev_fork[self._nfork % len(self._ev_fork)]
        self._nfork += 1
        ev.record(main)
        ev.wait(self._side)
        self._side_used = True

    def reduce_async(self, bucket, upto=None):
        if self._fresh:
            self._calls, self._fresh = [], False
        lo, hi = self.bounds[bucket], self.bounds[(bucket if upto is None else upto) + 1]
        self._calls.append((lo, hi))
        g = self.flat[lo:hi]
        self._covered += hi - lo
        main = torch.cuda.current_stream()
        if self.mode == "zero1":
            self._fork(main)
            L = _Rccl.lib()
            main_n, shard = zero_partition(lo, hi, self.world)
            el = self.flat.element_size()
            base = self.flat.data_ptr() + lo * el
            if main_n:      # in place: the receive buffer is this rank's slice of the send buffer
                _Rccl.check(L.ncclReduceScatter(base, base + self.rank * shard * el, shard, 7, 0, self._comm_side, self._side.cuda_stream),
                            "ncclReduceScatter")
            if main_n < hi - lo:
                self._allreduce(self.flat[lo + main_n:hi], self._comm_side, self._side)
            return
        self._fork(main)
        if self.wire == "bf16":
            full = self._staging((lo, hi), hi - lo)
            with torch.cuda.stream(self._side):
                full.copy_(g)
                self._allreduce(full, self._comm_side, self._side)
                g.copy_(full)
        else:
            self._allreduce(g, self._comm_side, self._side)

    def _join(self):
        if self._side_used:
            e0 = self.exposure.begin()
            self._ev_join.record(self._side)
            self._ev_join.wait(torch.cuda.current_stream())
            self.exposure.end(e0)
        self._side_used = False

    def wait(self):
        self._join()
        self._covered = 0
        self._fresh = True

    def wait_side(self):
        """The stream on which the reduced gradient is complete WITHOUT joining it into the compute stream: every collective of the
        step was enqueued on the side stream, so whatever is enqueued there next (the optimizer) follows them in stream order -- no
        event, no hop.  None for the sharded forms: the caller then uses wait()."""
        if self.mode != "allreduce":
            self.wait()
            return None
        self._side_used = False
        self._covered = 0
        self._fresh = True
        return self._side

    def gather_params(self, flat):
        """zero1, after the sharded optimizer (enqueued on the current stream): ncclAllGather of the updated shards, in place, on the
        side stream; the current stream continues after them."""
        L = _Rccl.lib()
        el = flat.element_size()
        self._fork(torch.cuda.current_stream())
        for lo, hi in self._calls:
            main_n, shard = zero_partition(lo, hi, self.world)
            if main_n:
                base = flat.data_ptr() + lo * el
                _Rccl.check(L.ncclAllGather(base + self.rank * shard * el, base, shard, 7, self._comm_side, self._side.cuda_stream),
                            "ncclAllGather")
        self._join()

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def close(self):
        comm, self._comm_side = getattr(self, "_comm_side", None), None
        if comm:
            torch.cuda.synchronize()
            _Rccl.lib().ncclCommDestroy(comm)


def crosscheck(red, group=None):
    """Start-up self-test of a reducer: a known integer-valued pattern (exact in fp32 AND bf16 whatever the summation order: every
    addend in [-amp, amp] with world x amp <= 256) goes through `red` with the trainer's call sequence (layers 2-4, then layer 1) and
    through a plain `torch.distributed.all_reduce`; both results must equal the closed-form sum BIT FOR BIT.  The gradient buffer is
    restored.  Returns a dict for the bench line; every rank must call it (collectives)."""
    flat = red.flat
    if not getattr(red, "active", False):
        return {"ok": True, "skipped": "reducer inactive"}
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    saved = flat.clone()
    n = flat.numel()
    idx = torch.arange(n, device=flat.device, dtype=torch.int64)

    # |addend| <= amp with world * amp <= 256: every partial sum is then an integer of magnitude <= 256, exact in bf16 (8 significant bits)
    # as well as in fp32 (the fixed amplitude of 15 was only exact up to 17 ranks on the bf16 wire)
    amp = max(1, min(15, 256 // max(1, world)))

    def pat(r):
        return ((idx * 2654435761 + r * 40503) % (2 * amp + 1) - amp).to(torch.float32)

    expect = pat(0)
    for r in range(1, world):
        expect += pat(r)
    try:
        flat.copy_(pat(rank))
        nb = len(red.bounds) - 1
        if nb > 1:
            red.reduce_async(1, upto=nb - 1)
        red.reduce_async(0)
        red.wait()
        if red.mode == "zero1":
            own_ok = all(bool(torch.equal(flat[lo:hi], expect[lo:hi])) for lo, hi in red.owned_ranges())
            red.gather_params(flat)
        else:
            own_ok = True
        if flat.is_cuda:
            torch.cuda.synchronize()      # never two communicators' kernels in flight at once (the reducer's and the process group's)
        ref = pat(rank)
        dist.all_reduce(ref, op=dist.ReduceOp.SUM, group=group)
        if flat.is_cuda:
            torch.cuda.synchronize()
        lo, hi = red.bounds[0], red.bounds[-1]
        ok_red = own_ok and bool(torch.equal(flat[lo:hi], expect[lo:hi]))
        ok_ref = bool(torch.equal(ref, expect))
        bad = int((flat[lo:hi] != expect[lo:hi]).sum()) if not ok_red else 0
    finally:
        flat.copy_(saved)
    return {"ok": ok_red and ok_ref, "reducer_bitwise": ok_red, "torch_all_reduce_bitwise": ok_ref, "mismatching_elements": bad,
            "elements": hi - lo, "backend": red.backend, "mode": red.mode, "wire": red.wire}


# ---- what a hardware scaling curve should look like (no multi-GPU box has been available to any round: this is the expectation the first
# real curve is read against, computed from numbers measured on ONE rank)
XGMI_LINK_GBPS_UNIDIR = 76.8      # one xGMI link, one direction (153.6 GB/s per link both ways; 7 links per MI355X, point to point)
XGMI_RING_EFFICIENCY = 0.7        # measured all-reduce bus bandwidth / link peak on MI300X-class 8-GPU nodes for 10-100 MB messages
RCCL_LATENCY_US = (12.0, 3.0)     # launch + protocol latency of one collective: a + b * (N - 1) microseconds


def allreduce_us(nbytes, world, link_gbps=XGMI_LINK_GBPS_UNIDIR, eta=XGMI_RING_EFFICIENCY, latency=RCCL_LATENCY_US):
    """Ring all-reduce of `nbytes` over `world` fully connected GPUs: every GPU sends 2 (N-1)/N of the bytes; RCCL can run up to N-1 rings
    over distinct links (N = 2: ONE link, which is why two GPUs are the worst case per byte), each at the link's one-direction rate."""
    if world <= 1:
        return 0.0
    bus = eta * (world - 1) * link_gbps * 1e9
    return latency[0] + latency[1] * (world - 1) + 2.0 * (world - 1) / world * nbytes / bus * 1e6


def predict_scaling(step_ms_n1, bucket_bytes, windows_us, plumbing_us=0.0, wire="f32", worlds=(2, 4, 8)):
    """Expected data-parallel step for N ranks from one-rank measurements.
      step_ms_n1    the step without any collective (measured)
      bucket_bytes  fp32 bytes of each gradient bucket in the order they are issued (early schedule: [layers 2-4, layer 1])
      windows_us    how much compute is still to run after each bucket is issued, i.e. what its collective can hide under
                    (layers 2-4: the last data-gradient GEMM + dW1; layer 1: the next step's encoder + gather when Adam runs on the
                    collectives' stream) -- from the one-rank kernel timeline
      plumbing_us   cost of the data-parallel plumbing on one rank (measured: forced-distributed step minus plain step)
    Per N: wire bytes per GPU, collective time per bucket, predicted exposed time and weak-scaling efficiency.  A MODEL, with its
    assumptions in the record; it exists so that a measured curve can be compared with something stated beforehand."""
    scale = 0.5 if wire == "bf16" else 1.0
    out = {"assumptions": {"link_GBps_one_direction": XGMI_LINK_GBPS_UNIDIR, "links_used": "N-1 (one ring per direct link)",
                           "ring_efficiency": XGMI_RING_EFFICIENCY, "collective_latency_us": "%g + %g (N-1)" % RCCL_LATENCY_US,
                           "wire": wire, "hide_windows_us": [round(w, 1) for w in windows_us], "plumbing_us_one_rank": round(plumbing_us, 1),
                           "step_ms_one_rank": round(step_ms_n1, 4)},
           "per_world": {}}
    for n in worlds:
        coll = [allreduce_us(b * scale, n) for b in bucket_bytes]
        exposed = sum(max(0.0, c - w) for c, w in zip(coll, windows_us))
        step = step_ms_n1 * 1e3 + plumbing_us + exposed
        out["per_world"][str(n)] = {"bytes": int(sum(bucket_bytes) * scale), "wire_bytes_per_gpu": int(2.0 * (n - 1) / n * sum(bucket_bytes) * scale),
                                    "links": n - 1, "collective_us": [round(c, 1) for c in coll], "predicted_exposed_us": round(exposed, 1),
                                    "predicted_ms_per_step": round(step / 1e3, 4),
                                    "predicted_efficiency": round(step_ms_n1 * 1e3 / step, 4), "predicted_speedup": round(n * step_ms_n1 * 1e3 / step, 2)}
    return out


def select_schedule(candidates, time_fn, device, group=None, supported=None):
    """Pick the data-parallel schedule by MEASUREMENT, on all ranks together (the agree-then-act pattern of make_reducer):
      1. `supported(name) -> bool` (local, NO collectives) says what this rank can run; the flags are MIN-reduced, so a candidate any rank
         cannot run is dropped on EVERY rank before anything is timed;
      2. every rank times every remaining candidate in the same order with `time_fn(name) -> milliseconds per step` (which runs real steps,
         collectives included, so the ranks stay in lock step); the times are MAX-reduced (a step is as slow as its slowest rank) and every
         rank takes the candidate with the smallest maximum -- the same one everywhere, whatever each rank measured locally; ties go to the
         earlier candidate.
    An exception inside time_fn is NOT swallowed: a rank that fails in the middle of a candidate has left its peers inside a
    collective, nothing here can repair that, and the launch watchdog must see -- and report -- the real error.
    Returns (choice, {name: max-over-ranks ms, or None for a candidate dropped in step 1})."""
    cands = list(candidates)
    if not cands:
        raise ValueError("no schedule candidates")
    multi = dist.is_initialized() and dist.get_world_size(group) > 1

    def reduce_(t, op):
        if not multi:
            return t
        if t.is_cuda and dist.get_backend(group) == "gloo":
            tc = t.cpu()
            dist.all_reduce(tc, op=op, group=group)
            return tc
        dist.all_reduce(t, op=op, group=group)
        return t

    ok = [1 if (supported is None or supported(c)) else 0 for c in cands]
    ok = [int(x) for x in reduce_(torch.tensor(ok, device=device, dtype=torch.int32), dist.ReduceOp.MIN).tolist()]
    live = [c for c, o in zip(cands, ok) if o]
    if not live:
        raise RuntimeError("no data-parallel schedule candidate is supported on every rank: %r" % (cands,))
    ms = [float(time_fn(c)) for c in live]
    vals = [float(x) for x in reduce_(torch.tensor(ms, device=device, dtype=torch.float64), dist.ReduceOp.MAX).tolist()]
    best = min(range(len(live)), key=lambda i: (vals[i], i))
    table = {c: None for c in cands}
    table.update({c: round(v, 4) for c, v in zip(live, vals)})
    return live[best], table


def _agree(flag, device, group):
    """MIN over the ranks of a 0/1 flag (through the process group that already works)."""
    t = torch.tensor([1 if flag else 0], device=device, dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item()) == 1


def make_reducer(flat_grad, bounds, group=None, force=False, mode=None):
    """The gradient reducer of a data-parallel trainer: RCCL driven directly on GPU buffers under an NCCL/RCCL process group
    (DPD_DP_BACKEND=torch, DPD_DP_MODE=rs_ag, or a failure to bind librccl keep `BucketReducer`), torch.distributed otherwise
    (gloo / CPU tensors: the tests).

    Every decision is taken by ALL ranks together, and before anything that could leave ranks in different collectives:
      1. bind librccl (local, no communication)            -> agree (MIN)  -> all direct, or all torch.distributed
      2. create the communicator(s)                        -> agree (MIN); a rank that fails inside ncclCommInitRank takes the others
                                                              with it or hangs them: that case belongs to the launch watchdog
      3. `crosscheck` (known pattern, bitwise)             -> agree (MIN)  -> direct reducer, or fall back together
    The torch.distributed reducer runs the same cross-check and raises if IT is wrong (nothing left to fall back to)."""
    import os
    import sys
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    want = os.environ.get("DPD_DP_BACKEND", "rccl")
    mode = mode or os.environ.get("DPD_DP_MODE", "allreduce")      # `mode`: a caller that cannot shard its optimizer pins "allreduce"
    if (dist.is_initialized() and (world > 1 or force) and flat_grad.is_cuda and want == "rccl" and dist.get_backend(group) == "nccl"
            and mode in ("allreduce", "zero1")):
        err = None
        try:
            _Rccl.lib()
        except Exception as e:
            err = e
        red = None
        if _agree(err is None, flat_grad.device, group):
            try:
                red = DirectRcclReducer(flat_grad, bounds, group, mode=mode)
            except Exception as e:      # plumbing only: the torch.distributed path computes the same sums
                err = e
            if _agree(red is not None, flat_grad.device, group):
                try:
                    red.crosscheck = crosscheck(red, group)
                except Exception as e:
                    err, red.crosscheck = e, {"ok": False, "error": repr(e)}
                if not _agree(red.crosscheck["ok"], flat_grad.device, group):
                    err = err or RuntimeError("start-up cross-check failed: %r" % (red.crosscheck,))
                    red.close()
                    red = None
                if red is not None:
                    return red
            elif red is not None:
                red.close()
                red = None
        sys.stderr.write("dpdist_amd.ddp: direct RCCL unavailable on some rank (%r here), using torch.distributed collectives\n" % (err,))
    red = BucketReducer(flat_grad, bounds, group, force=force, mode=mode)
    if red.active:
        red.crosscheck = crosscheck(red, group)
        if not red.crosscheck["ok"]:
            raise RuntimeError("torch.distributed gradient reducer failed its start-up cross-check: %r" % (red.crosscheck,))
    return red
