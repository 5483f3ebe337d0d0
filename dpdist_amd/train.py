#!/usr/bin/env python3
"""DPDist trainer loop: the caller of the hot path.

Mirrors `train_multi_gpu_pc_compare_dist.py` for `--train_comp dpdist` (relative to /root/reference):
    flags and defaults                         :41-69,73-136   (same names; `--num_gpus` is replaced by torchrun)
    batch composition from dataset items       :747-766 (train), :831-847 (eval)          -> compose_batch()
    train_one_epoch_3d / eval_one_epoch_3d     :732-807, :809-873
    epoch loop, eval + checkpoint every 10     :347-357
    learning-rate schedule, Adam               :216,976-990                                 -> DPDistTrainer
The ModelNet `*_dist_c_scaled.txt` / `*_neg_{l,u}.txt` files are not in the reference tree, so the dataset is a
synthetic stand-in with the SAME item format as `modelnet_dataset.ModelNetDataset` (`:98-187`): per shape
3*2N points (surface | near-surface | far) and 2*2N ground-truth distances, on analytic surfaces (spheres / boxes
of extent <= 0.8 like `dataset_sample_with_gt.py:82`), with the reference's augmentation (random y-rotation +
shift U(-0.1,0.1), `modelnet_dataset.py:82-95`, `provider.py:32-50,200-211`).

    python -m dpdist_amd.train --max_epoch 20 --batch_size 32            # one GPU
    python -m torch.distributed.run --nproc-per-node 8 -m dpdist_amd.train --batch_size 512   # global batch, DP over RCCL
"""
import argparse
import json
import math
import os
import time

import numpy as np


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--log_dir", default="log/test1_")
    p.add_argument("--num_point", type=int, default=64)
    p.add_argument("--max_epoch", type=int, default=10001)
    p.add_argument("--batch_size", type=int, default=16, help="GLOBAL batch (split evenly over the ranks)")
    p.add_argument("--learning_rate_dpdist", type=float, default=0.0001)
    p.add_argument("--optimizer", default="adam")
    p.add_argument("--decay_step", type=int, default=300 * 512)
    p.add_argument("--decay_rate", type=float, default=0.5)
    p.add_argument("--encoder", default="3dmfv")
    p.add_argument("--embedding_size", type=int, default=8 ** 3)
    p.add_argument("--BN", default="0")
    p.add_argument("--K", default="5")
    p.add_argument("--loss_type", default="l1_dist")
    p.add_argument("--implicit_net_type", default="1")
    p.add_argument("--category", default="chair")
    p.add_argument("--sigma3dmfv", type=float, default=2.0)
    p.add_argument("--add_noise", type=float, default=0.0)
    p.add_argument("--train_shapes", type=int, default=889, help="synthetic stand-in for the 889 train chairs")
    p.add_argument("--test_shapes", type=int, default=100, help="synthetic stand-in for the 100 test chairs")
    p.add_argument("--eval_every", type=int, default=10)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--restore", default="", help="TF V2 checkpoint prefix (model.ckpt) or .npz to start from")
    p.add_argument("--prefetch", type=int, default=0, help="1 = next batch's encoder + gather on a side stream (measured slower on MI355X)")
    p.add_argument("--dp_schedule", default=os.environ.get("DPD_DP_SCHEDULE", "early"), choices=["early", "grouped", "late", "auto"],
                   help="order of the data-parallel backward.  Default 'early': deterministic, a multi-GPU run is bitwise reproducible and "
                        "resumable.  'auto' MEASURES order x communication form on the first batch (all ranks together), stores the choice in "
                        "<log_dir>/dp_schedule.json and re-uses the stored choice of the run it resumes (--restore) instead of measuring again")
    return p


# --------------------------------------------------------------------------------------------------------------
# dataset with the item format of modelnet_dataset.ModelNetDataset
# --------------------------------------------------------------------------------------------------------------
class SyntheticDistanceDataset:
    """Items: point_set [3*npoints,3] = surface | near | far, labels [2*npoints] = near dist | far dist."""

    def __init__(self, n_shapes, npoints, batch_size, split="train", seed=0, shuffle=None):
        from . import synth
        self.npoints, self.batch_size, self.split = npoints, batch_size, split
        self.shuffle = (split == "train") if shuffle is None else shuffle
        rng = np.random.default_rng(seed + (0 if split == "train" else 10 ** 6))
        self.items = []
        for _ in range(n_shapes):
            if rng.random() < 0.5:
                r = rng.uniform(0.3, 0.7)
                samp = lambda n, r=r: synth._sample_sphere(rng, n, r)      # noqa: E731
                dist = lambda p, r=r: synth._dist_sphere(p, r)             # noqa: E731
            else:
                h = rng.uniform(0.2, 0.55, 3)
                samp = lambda n, h=h: synth._sample_box(rng, n, h)         # noqa: E731
                dist = lambda p, h=h: synth._dist_box(p, h)                # noqa: E731
            surf = samp(npoints)
            near = np.zeros((0, 3))
            while len(near) < npoints:
                c = samp(4 * npoints) + rng.standard_normal((4 * npoints, 3)) * 0.04
                d = dist(c)
                near = np.concatenate([near, c[(d > 0.001) & (d < 0.1)]])
            near = near[:npoints]
            far = np.zeros((0, 3))
            while len(far) < npoints:
                c = rng.standard_normal((8 * npoints, 3))
                c = c / np.linalg.norm(c, axis=1, keepdims=True) * rng.random((8 * npoints, 1)) ** (1 / 3) * 0.85
                far = np.concatenate([far, c[dist(c) > 0.1]])
            far = far[:npoints]
            pts = np.concatenate([surf, near, far]).astype(np.float32)
            lab = np.concatenate([dist(near), dist(far)]).astype(np.float32)
            self.items.append((pts, lab))
        self._rng = np.random.default_rng(seed + 17)
        self.reset()

    def num_channel(self):
        return 3

    def reset(self):
        self.idxs = np.arange(len(self.items))
        if self.shuffle:
            self._rng.shuffle(self.idxs)
        self.num_batches = (len(self.items) + self.batch_size - 1) // self.batch_size
        self.batch_idx = 0

    def has_next_batch(self):
        return self.batch_idx < self.num_batches

    def _get_item(self, index):
        pts, lab = self.items[index]
        n = self.npoints
        shuff = self._rng.permutation(n)                           # modelnet_dataset.py:99-110: same permutation for
        pts = pts.reshape(3, n, 3)[:, shuff].reshape(3 * n, 3)      # the three point sets and the two label sets
        lab = lab.reshape(2, n)[:, shuff].reshape(2 * n)
        return pts, lab

    def next_batch(self, augment=False):
        lo = self.batch_idx * self.batch_size
        hi = min((self.batch_idx + 1) * self.batch_size, len(self.items))
        data = np.zeros((hi - lo, 3 * self.npoints, 3), np.float32)
        label = np.zeros((hi - lo, 2 * self.npoints), np.float32)
        for i in range(hi - lo):
            data[i], label[i] = self._get_item(self.idxs[lo + i])
        self.batch_idx += 1
        if augment:                                                 # provider.rotate_point_cloud + shift_point_cloud
            ang = self._rng.uniform(0, 2 * math.pi, hi - lo)
            c, s = np.cos(ang), np.sin(ang)
            R = np.zeros((hi - lo, 3, 3), np.float32)
            R[:, 0, 0], R[:, 0, 2], R[:, 1, 1], R[:, 2, 0], R[:, 2, 2] = c, s, 1, -s, c
            data = np.einsum("bnd,bde->bne", data, R).astype(np.float32)
            data += self._rng.uniform(-0.1, 0.1, (hi - lo, 1, 3)).astype(np.float32)
        return data, label


def compose_batch(batch_data, batch_label, num_point):
    """train_multi_gpu_pc_compare_dist.py:747-766: dataset items -> (pcA, pcB, labels_AB).
    pcA = first N of surface half 1; pcB = [N/2 of surface half 2 | N/4 near | N/4 far[N/4:N/2]];
    labels_AB = [0]*N/2 + near GT + far GT."""
    H = num_point // 2
    q = int(H * 0.5)                                                # split_off_surface = 0.5
    surface, close, far = np.split(batch_data, 3, 1)               # :752
    sA, sB = np.split(surface, 2, 1)                               # :753
    lab_close, lab_far = np.split(batch_label, 2, 1)               # :758
    pcA = sA[:, :num_point]
    labels = np.concatenate([np.zeros((len(pcA), H), np.float32), lab_close[:, :q], lab_far[:, q:H]], 1)   # :759-761
    off = np.concatenate([close[:, :q], far[:, q:H]], 1)           # :764-765
    pcB = np.concatenate([sB[:, :H], off], 1)                      # :766
    return pcA.astype(np.float32), pcB.astype(np.float32), labels.astype(np.float32)


def iter_global_batches(ds, batch_size, num_point, training):
    """Host side of train_one_epoch_3d / eval_one_epoch_3d (:737-766, :820-847): yields (pcA, pcB, labels_AB) of the STATIC
    global batch shape.  Like the reference, the arrays are persistent per-epoch buffers that start at zero; a short last
    batch overwrites rows [0, bsize) and the remaining rows keep the previous batch's content (the yielded arrays are
    the buffers themselves: copy before the next iteration)."""
    cur_A = np.zeros((batch_size, num_point, 3), np.float32)
    cur_B = np.zeros((batch_size, num_point, 3), np.float32)
    cur_lab = np.zeros((batch_size, num_point), np.float32)
    while ds.has_next_batch():
        data, label = ds.next_batch(augment=training)
        bsize = len(data)
        cur_A[:bsize], cur_B[:bsize], cur_lab[:bsize] = compose_batch(data, label, num_point)
        yield cur_A, cur_B, cur_lab


# --------------------------------------------------------------------------------------------------------------
def train(argv=None):
    import torch
    import torch.distributed as dist
    from .ddp import shard_range
    from .model import DPDistParams
    from .tf_checkpoint import read_checkpoint, write_checkpoint
    from .trainer import DPDistTrainer

    F = build_parser().parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    share_gpu = os.environ.get("DPD_TEST_SHARE_GPU") == "1"        # tests only: every rank on GPU 0 over gloo (as in bench.py)
    if share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert F.batch_size % world == 0                               # :122-126
    dev_bs = F.batch_size // world
    N, K = F.num_point, int(F.K)
    sigma = F.sigma3dmfv * 0.0625                                   # :103
    if int(F.BN) or F.encoder != "3dmfv" or int(F.implicit_net_type) != 1 or F.loss_type != "l1_dist" or F.optimizer != "adam":
        raise NotImplementedError("only the reference's shipped configuration is on the hot path")

    os.makedirs(F.log_dir, exist_ok=True)
    log = open(os.path.join(F.log_dir, "log_trainours.txt"), "a") if rank == 0 else None

    def log_string(s):                                              # :930-934
        if rank == 0:
            log.write(s + "\n")
            log.flush()
            print(s, flush=True)

    log_string(str(F))
    train_ds = SyntheticDistanceDataset(F.train_shapes, 2 * N, F.batch_size, "train", F.seed)   # npoints=NUM_POINT*2 (:181)
    test_ds = SyntheticDistanceDataset(F.test_shapes, 2 * N, F.batch_size, "test", F.seed)
    params = DPDistParams(k=K, mlp=(1024, 1024, 1024), device=dev)
    params.reset_parameters_tf(generator=torch.Generator().manual_seed(F.seed))               # replicated variables
    tr = DPDistTrainer(params, dev_bs, num_point=N, Embedding_Size=F.embedding_size, sigma3dmfv=sigma,
                       base_lr=F.learning_rate_dpdist, decay_step=F.decay_step, decay_rate=F.decay_rate,
                       adam_on_side=world > 1)         # optimizer on the collectives' stream (trainer.apply_gradients); the trainer joins it itself
    if F.restore:                                                                            # saver.restore (:443-453)
        # weights AND, when the checkpoint has them (ours do, like the reference's Saver()), global step + Adam state
        got = tr.load_tf_global_variables(dict(np.load(F.restore)) if F.restore.endswith(".npz") else read_checkpoint(F.restore))
        log_string("restored %s from %s (global step %d)" % (", ".join(got), F.restore, tr.t))
    lo, hi = shard_range(F.batch_size, rank, world)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev, non_blocking=True)       # noqa: E731

    def batches(ds, training):
        for pcA, pcB, lab in iter_global_batches(ds, F.batch_size, N, training):
            noise = None
            if F.add_noise > 0.0:                                  # :768-771
                noise = cu((np.random.randn(F.batch_size, N, 3) * F.add_noise).astype(np.float32)[lo:hi])
            yield cu(pcA[lo:hi].copy()), cu(pcB[lo:hi].copy()), cu(lab[lo:hi].copy()), noise

    # data-parallel form of the step: pinned (default "early") unless --dp_schedule auto; a measured choice is stored next to the
    # checkpoints and re-used on resume, so that a resumed run continues in the form it was started in (two measurements can
    # pick different winners when candidates are close, and "grouped" sums dW in another fp32 order)
    sched_done = [world == 1 or F.dp_schedule != "auto"]
    if world > 1 and F.dp_schedule != "auto":
        tr.dp_schedule = F.dp_schedule
        tr.dp_schedule_info = {"schedule": F.dp_schedule, "mode": tr.reducer.mode if tr.reducer is not None else None, "source": "--dp_schedule (pinned)"}
        log_string("data-parallel schedule: %s" % (tr.dp_schedule_info,))
    elif world > 1:
        side = os.path.join(os.path.dirname(os.path.abspath(F.restore)), "dp_schedule.json") if F.restore else ""
        stored = None
        if side and os.path.exists(side):            # every rank reads the same file
            with open(side) as f:
                stored = json.load(f)
        if stored and stored.get("schedule") in ("early", "grouped", "late") and stored.get("mode") in ("allreduce", "rs_ag", "zero1"):
            tr.set_dp_mode(stored["mode"])           # collective
            tr.dp_schedule = stored["schedule"]
            tr.dp_schedule_info = dict(stored, source="re-used from %s (%s)" % (side, stored.get("source")))
            log_string("data-parallel schedule: %s" % (tr.dp_schedule_info,))
            sched_done[0] = True

    def run_epoch(ds, training):
        sums, n = torch.zeros(2, device=dev), 0
        it = batches(ds, training)
        cur = next(it, None)
        if training and not sched_done[0] and cur is not None:
            # data-parallel runs: the order of the backward (early / grouped / late) is measured on the first batch by all ranks together;
            # weights, Adam slots and the global step are restored afterwards (DPDistTrainer.select_dp_schedule)
            os.environ["DPD_DP_SCHEDULE"] = "auto"
            info = tr.select_dp_schedule(cur[0], cur[1], cur[2], modes=None if "DPD_DP_MODE" in os.environ else ("allreduce", "rs_ag", "zero1"))
            log_string("data-parallel schedule: %s" % (info,))
            if rank == 0:
                with open(os.path.join(F.log_dir, "dp_schedule.json"), "w") as f:
                    json.dump(info, f)
            sched_done[0] = True
        while cur is not None:
            nxt = next(it, None)                                   # composed one batch ahead (host work overlaps the GPU step)
            a, b, l, noise = cur
            if training:
                # trainer-side prefetch (encoder + gather of the next batch on a side stream) measured slower than plain
                # stream order on MI355X, see bench.py --prefetch; opt in with --prefetch 1
                pf = (nxt[0], nxt[1], nxt[3]) if (nxt is not None and F.prefetch) else None
                loss = tr.step(a, b, l, noise, prefetch=pf)
            else:
                loss = tr.evaluate(a, b, l, noise)[0]
            sums += loss
            n += 1
            cur = nxt
        ds.reset()
        if world > 1:
            tr.join_optimizer()              # the side stream's collectives + Adam are ordered before the process group's all-reduce
            dist.all_reduce(sums)
            sums /= world
        return (sums / max(n, 1)).tolist()

    t0 = time.time()
    for epoch in range(F.max_epoch):                                # :347
        ls, lp = run_epoch(train_ds, True)
        log_string(" ---- epoch: %03d ----  mean loss: %f  (loss_pred %f)  step %d  %.1fs" % (epoch + 1, ls, lp, tr.t, time.time() - t0))
        if epoch % F.eval_every == 0 or epoch == F.max_epoch - 1:   # :349-357
            es, ep = run_epoch(test_ds, False)
            log_string("eval mean loss: %f" % es)
            tr.gather_optimizer_state()             # zero1: every rank takes part (Adam slots live sharded); no-op otherwise
            if rank == 0:
                sd = tr.tf_global_variables()       # what tf.train.Saver() saves: variables, `batch`, beta powers, Adam slots
                np.savez(os.path.join(F.log_dir, "model.ckpt.npz"), **sd)
                write_checkpoint(os.path.join(F.log_dir, "model.ckpt"), sd)                    # saver.save(...) (:354-357)
                with open(os.path.join(F.log_dir, "metrics.jsonl"), "a") as f:
                    f.write(json.dumps({"epoch": epoch + 1, "step": tr.t, "train_loss_samples": ls, "eval_loss_samples": es}) + "\n")
    tr.close()                                      # RCCL communicators go before the process group does
    if world > 1:
        dist.destroy_process_group()
    return ls


if __name__ == "__main__":
    train()
