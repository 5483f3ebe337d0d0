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
    the current stream -- not the host -- wait for it, so one event hop each way is all the synchronisation there is."""

    def __init__(self, flat_grad, bounds, group=None, force=False):
        self.flat = flat_grad
        self.bounds = list(bounds)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # force=True issues the collectives even for a single rank (exercises the RCCL/stream plumbing on one GPU)
        self.active = dist.is_initialized() and (self.world > 1 or force)
        self._pending = []

    def reduce_async(self, bucket):
        """Call right after the kernels producing bucket `bucket` were enqueued on the current stream."""
        if not self.active:
            return
        lo, hi = self.bounds[bucket], self.bounds[bucket + 1]
        self._pending.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        """Make the current stream (or the host, for CPU tensors) wait for every outstanding bucket."""
        for h in self._pending:
            h.wait()
        self._pending = []

    @property
    def grad_scale(self):
        return 1.0 / self.world
