"""Data-parallel path on CPU: world_size 2 over gloo.  The bucketed all-reduce + 1/world scaling of ddp.py must
reproduce the full-batch gradient (mean of equal shard means), the role of average_gradients in the reference
(train_multi_gpu_pc_compare_dist.py:936-974) -- for the fp32 and the bf16 wire, as one all-reduce per bucket and as
reduce-scatter + all-gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dpdist_amd import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat_grad(P, pcA, pcB, lab, mlp):
    """per-shard gradient of loss_samples from the oracle, laid out in the product's flat/bucket layout"""
    from oracle import restate as R
    W = R.as_torch_weights(synth.make_weights("wide", mlp=mlp), torch.float64, requires_grad=True)
    pred, _ = R.get_model(torch.tensor(pcA, dtype=torch.float64), torch.tensor(pcB, dtype=torch.float64), W)
    ls, _ = R.get_loss(pred, torch.tensor(lab, dtype=torch.float64))
    names = sorted(W)
    g = torch.autograd.grad(ls, [W[n] for n in names])
    P.load_tf_state_dict({n: gg.numpy() for n, gg in zip(names, g)})    # reuse the layout code: grads -> flat
    return P.flat.detach().clone()


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from dpdist_amd.ddp import BucketReducer, shard_range
    from dpdist_amd.model import DPDistParams
    mlp = (64, 64, 64)
    GB = 4
    pcA, pcB, lab = synth.s2_modelnet_shaped(GB, 64, 100)
    lo, hi = shard_range(GB, rank, world)
    P = DPDistParams(k=5, mlp=mlp, device="cpu", init=None)
    shard = _flat_grad(P, pcA[lo:hi], pcB[lo:hi], lab[lo:hi], mlp).float()
    full = _flat_grad(P, pcA, pcB, lab, mlp).float() if rank == 0 else None
    res = {}
    for wire, mode in (("f32", "allreduce"), ("f32", "rs_ag"), ("bf16", "allreduce"), ("bf16", "rs_ag")):
        flat = shard.clone()
        red = BucketReducer(flat, P.bucket_bounds, wire=wire, mode=mode)
        assert len(P.bucket_bounds) == 4
        for b in (2, 1, 0):          # layers 3-4, layer 2, layer 1: the order trainer.backward produces them in
            red.reduce_async(b)
        red.wait()
        flat *= red.grad_scale
        if rank == 0:
            res[(wire, mode)] = (float((flat - full).abs().max()), float(full.abs().max()), red.world)
    if rank == 0:
        out.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_equals_full_batch_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for (wire, mode), (err, scale, world) in res.items():
        assert world == 2
        # fp32 wire: summation order only; bf16 wire: every addend and the sum are rounded to 8 bits of mantissa
        tol = 1e-6 * max(1.0, scale) if wire == "f32" else 2.0 ** -7 * scale
        assert err <= tol, (wire, mode, err, scale)


def test_shard_range():
    from dpdist_amd.ddp import shard_range
    assert [shard_range(512, r, 8) for r in (0, 7)] == [(0, 64), (448, 512)]
    with pytest.raises(ValueError):
        shard_range(10, 0, 4)


_RANK_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
t = torch.tensor([float(os.environ["LOCAL_RANK"]) + 1.0])
dist.all_reduce(t)
if dist.get_rank() == 0:
    open(sys.argv[1], "w").write("%d %.1f" % (dist.get_world_size(), t.item()))
dist.destroy_process_group()
if os.environ["RANK"] == sys.argv[2]:
    sys.exit(7)
"""


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` with no launcher: bench.spawn_ranks starts N processes with the torchrun environment
    contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* on 127.0.0.1), waits for them and returns the first failure."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out = tmp_path / "r0.txt"
    assert bench.spawn_ranks(2, [sys.executable, "-c", _RANK_SCRIPT, str(out), "-1"]) == 0
    assert out.read_text() == "2 3.0"
    assert bench.spawn_ranks(2, [sys.executable, "-c", _RANK_SCRIPT, str(out), "1"]) == 7


def test_make_reducer_keeps_torch_distributed_off_the_gpu():
    """ddp.make_reducer: the direct librccl reducer needs an RCCL process group and a GPU buffer; anything else (gloo, CPU tensors,
    no process group) gets the torch.distributed BucketReducer."""
    import torch
    from dpdist_amd.ddp import BucketReducer, make_reducer
    r = make_reducer(torch.zeros(16), [0, 4, 8, 16])
    assert isinstance(r, BucketReducer) and not r.active
