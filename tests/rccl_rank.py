"""One rank of the 2-GPU RCCL parity check (started by tests/test_gpu_parity.py::test_two_ranks_rccl_equals_full_batch via
bench.spawn_ranks).  GPU twin of tests/test_ddp_gloo.py: every rank runs the product's data-parallel backward on its shard of
the global batch (ddp.shard_range, three async buckets, 1/world folded into the scale) and rank 0 compares the averaged
gradient with the gradient of the FULL batch from a non-distributed trainer -- the role of average_gradients
(train_multi_gpu_pc_compare_dist.py:936-974)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main(out_path, dtype, mode="allreduce"):
    os.environ["DPD_DP_MODE"] = mode           # "zero1": sharded optimizer; replicas must still end bit-identical
    rank, world, local = (int(os.environ[k]) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from dpdist_amd import synth
    from dpdist_amd.ddp import shard_range
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    GB = 8 * world
    pcA, pcB, lab = (torch.tensor(x, device=dev) for x in synth.s2_modelnet_shaped(GB, 64, 100))
    lo, hi = shard_range(GB, rank, world)
    W0 = synth.make_weights("wide")
    P = DPDistParams(device=dev, compute_dtype=dtype)
    P.load_tf_state_dict(W0)
    tr = DPDistTrainer(P, hi - lo, base_lr=1e-3)
    assert tr.reducer is not None and tr.reducer.active and tr.reducer.world == world and tr.reducer.nranks == world
    assert tr.reducer.crosscheck["ok"], tr.reducer.crosscheck
    tr._load_batch(pcA[lo:hi].contiguous(), pcB[lo:hi].contiguous(), None)
    tr.forward()
    tr.backward(lab[lo:hi].reshape(-1).contiguous())
    tr.reducer.wait()
    if mode == "zero1":                          # this rank holds the sums of its shards only: gather them for the comparison
        tr.reducer.gather_params(tr.grad)
    g = tr.grad * tr.reducer.grad_scale
    # two full optimizer steps as well: replicas must stay bit-identical (same averaged gradient, same Adam)
    for _ in range(2):
        tr.step(pcA[lo:hi].contiguous(), pcB[lo:hi].contiguous(), lab[lo:hi].contiguous())
    w = P.flat.detach().clone()
    ws = [torch.empty_like(w) for _ in range(world)]
    dist.all_gather(ws, w)
    torch.cuda.synchronize()
    if rank == 0:
        P1 = DPDistParams(device=dev, compute_dtype=dtype)
        P1.load_tf_state_dict(W0)
        t1 = DPDistTrainer(P1, GB, base_lr=1e-3, distributed=False)
        t1._load_batch(pcA, pcB, None)
        t1.forward()
        t1.backward(lab.reshape(-1))
        torch.cuda.synchronize()
        err = float((g - t1.grad).abs().max())
        scale = float(t1.grad.abs().max())
        same = all(bool(torch.equal(ws[0], x)) for x in ws[1:])
        open(out_path, "w").write("%d %.6e %.6e %d" % (world, err, scale, int(same)))
    dist.barrier()
    tr.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "f32", sys.argv[3] if len(sys.argv) > 3 else "allreduce")
