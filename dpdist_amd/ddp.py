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
