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


class BucketReducer:
    """All-reduce slices of one flat gradient buffer, asynchronously with respect to the compute stream.

    The collective is issued with async_op=True from the compute stream: ProcessGroupNCCL orders its internal RCCL stream
    after everything already enqueued on the current stream (the producers of the bucket) and `Work.wait()` later makes
    the current stream -- not the host -- wait for it, so one event hop each way is all the synchronisation there is.

    wire = "f32" (default) | "bf16": dtype on the links.  "bf16" halves the bytes per step (18.7 -> 9.3 MB): each bucket is
        rounded into a bf16 staging buffer, reduced, and written back to the fp32 gradient (fp32 master weights and Adam state;
        the cross-rank sum itself is taken in bf16, 2^-8 relative per addend) -- meant for the bf16 compute type, whose step is
        short enough (0.40 ms at 64 pairs) for the fp32 all-reduce to show.
    mode = "allreduce" (default) | "rs_ag": reduce-scatter followed by all-gather of the same bucket (what a ring all-reduce
        does internally, as two collectives: the scatter half can start while later buckets are still being produced and the
        gather half of every bucket is deferred to `wait()`, i.e. to just before the optimizer).
    Expected exposed time on 8 x MI355X (xGMI ring, ~7 x 153 GB/s per GPU, all-reduce moves 2 (P-1)/P of the bytes per link):
    fp32 18.7 MB -> ~33 MB per GPU on the wire ~ 40-60 us, of which only the layer-1 bucket's tail (issued last) cannot hide
    under the remaining backward; bf16 halves it.  None of this could be measured here (one-GPU boxes): see DESIGN.md."""

    def __init__(self, flat_grad, bounds, group=None, force=False, wire=None, mode=None):
        import os
        self.flat = flat_grad
        self.bounds = list(bounds)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # force=True issues the collectives even for a single rank (exercises the RCCL/stream plumbing on one GPU)
        self.active = dist.is_initialized() and (self.world > 1 or force)
        self.wire = wire or os.environ.get("DPD_DP_WIRE", "f32")
        self.mode = mode or os.environ.get("DPD_DP_MODE", "allreduce")
        if self.wire not in ("f32", "bf16") or self.mode not in ("allreduce", "rs_ag"):
            raise ValueError("wire must be f32|bf16 and mode allreduce|rs_ag, got %r / %r" % (self.wire, self.mode))
        self._pending = []
        self._stage = {}            # bucket -> (staging buffer [padded], shard buffer) for the bf16 wire / rs_ag mode

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
        lo, hi = self.bounds[bucket], self.bounds[(bucket if upto is None else upto) + 1]
        bucket = (bucket, upto)
        g = self.flat[lo:hi]
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
        for h, extra in self._pending:
            h.wait()
            if extra is not None and extra[2] is not None:           # rs_ag: second half
                g, full, shard = extra
                gathers.append((dist.all_gather_into_tensor(full, shard, group=self.group, async_op=True), g, full))
            elif extra is not None:
                g, full, _ = extra
                g.copy_(full[:g.numel()])
        for h, g, full in gathers:
            h.wait()
            g.copy_(full[:g.numel()])
        self._pending = []

    @property
    def grad_scale(self):
        return 1.0 / self.world


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


class DirectRcclReducer:
    """BucketReducer's contract (reduce_async / wait / grad_scale) with RCCL called directly.

    `torch.distributed` puts every collective on ProcessGroupNCCL's own stream and orders it with events recorded in the middle
    of the compute stream -- on this runtime a barrier packet with a system-scope release, ~20 us of idle compute stream each, and
    a cross-queue hop (~14 us) each way for the LAST bucket, which nothing hides (+52..57 us per step before a
    single byte moves).  Here:
      * the early buckets (everything but the last call of a step) run on a side stream behind a `hipEventDisableSystemFence` event:
        recording it costs the compute stream ~0.3 us, the hop is paid by the side stream;
      * the LAST bucket's all-reduce is enqueued on the COMPUTE stream itself (its own communicator): no event, no hop -- it starts
        the moment dW1 ends and Adam follows it in stream order;
      * `wait()` joins the side stream with one more light event (its collective finished long before: dW1 ran meanwhile).
    wire = "f32" | "bf16" as in BucketReducer (staging copies on the stream of the collective).  One rank (force) works: RCCL's
    single-rank all-reduce is a copy."""

    def __init__(self, flat_grad, bounds, group=None, wire=None):
        import os
        from .hipevents import LightEvent
        if not (dist.is_initialized() and flat_grad.is_cuda):
            raise RuntimeError("DirectRcclReducer needs an initialised process group and a GPU gradient buffer")
        self.flat, self.bounds, self.group = flat_grad, list(bounds), group
        self.world = dist.get_world_size(group)
        self.active = True
        self.wire = wire or os.environ.get("DPD_DP_WIRE", "f32")
        if self.wire not in ("f32", "bf16"):
            raise ValueError("wire must be f32|bf16")
        self.mode = "allreduce"
        dev = flat_grad.device
        self._side = torch.cuda.Stream(device=dev)
        self._comm_side = _Rccl.new_comm(group, dev)
        self._comm_main = _Rccl.new_comm(group, dev)
        self._ev_fork = [LightEvent() for _ in range(4)]
        self._ev_join = LightEvent()
        self._nfork = 0
        self._covered = 0           # elements reduced so far in this step
        self._side_used = False
        self._stage = {}
        self._copyback = []         # bf16 wire: (stream is_main, g, full) pending conversions back to fp32

    def _allreduce(self, t, comm, stream):
        dt = 9 if t.dtype == torch.bfloat16 else 7          # ncclBfloat16 / ncclFloat
        _Rccl.check(_Rccl.lib().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), dt, 0, comm, stream.cuda_stream), "ncclAllReduce")

    def _staging(self, key, n):
        if key not in self._stage:
            self._stage[key] = torch.zeros(n, device=self.flat.device, dtype=torch.bfloat16)
        return self._stage[key]

    def reduce_async(self, bucket, upto=None):
        lo, hi = self.bounds[bucket], self.bounds[(bucket if upto is None else upto) + 1]
        g = self.flat[lo:hi]
        self._covered += hi - lo
        last = self._covered >= self.bounds[-1] - self.bounds[0]
        main = torch.cuda.current_stream()
        if last:
            if self.wire == "bf16":
                full = self._staging((lo, hi), hi - lo)
                full.copy_(g)
                self._allreduce(full, self._comm_main, main)
                g.copy_(full)
            else:
                self._allreduce(g, self._comm_main, main)
            return
        ev = self._ev_fork[self._nfork % len(self._ev_fork)]
        self._nfork += 1
        ev.record(main)
        ev.wait(self._side)
        self._side_used = True
        if self.wire == "bf16":
            full = self._staging((lo, hi), hi - lo)
            with torch.cuda.stream(self._side):
                full.copy_(g)
                self._allreduce(full, self._comm_side, self._side)
                g.copy_(full)
        else:
            self._allreduce(g, self._comm_side, self._side)

    def wait(self):
        if self._side_used:
            self._ev_join.record(self._side)
            self._ev_join.wait(torch.cuda.current_stream())
        self._side_used = False
        self._covered = 0

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def close(self):
        for c in ("_comm_side", "_comm_main"):
            comm = getattr(self, c, None)
            if comm:
                torch.cuda.synchronize()
                _Rccl.lib().ncclCommDestroy(comm)
                setattr(self, c, None)


def make_reducer(flat_grad, bounds, group=None, force=False):
    """The gradient reducer of a data-parallel trainer: RCCL driven directly on GPU buffers under an NCCL/RCCL process group
    (DPD_DP_BACKEND=torch, a DPD_DP_MODE other than allreduce, or a failure to bind librccl keep `BucketReducer`), torch.distributed
    otherwise (gloo / CPU tensors: the tests)."""
    import os
    import sys
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    want = os.environ.get("DPD_DP_BACKEND", "rccl")
    if (dist.is_initialized() and (world > 1 or force) and flat_grad.is_cuda and want == "rccl" and dist.get_backend(group) == "nccl"
            and os.environ.get("DPD_DP_MODE", "allreduce") == "allreduce"):
        red, err = None, None
        try:
            red = DirectRcclReducer(flat_grad, bounds, group)
        except Exception as e:      # plumbing only: the torch.distributed path computes the same sums
            err = e
        # the choice must be the same on every rank (a rank that fell back alone would wait for collectives the others never issue)
        ok = torch.tensor([1 if red is not None else 0], device=flat_grad.device, dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 1:
            return red
        if red is not None:
            red.close()
        sys.stderr.write("dpdist_amd.ddp: direct RCCL unavailable on some rank (%r here), using torch.distributed collectives\n" % (err,))
    return BucketReducer(flat_grad, bounds, group, force=force)
