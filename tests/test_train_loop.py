"""Trainer loop (row f1): batch composition restates train_multi_gpu_pc_compare_dist.py:747-766; GPU: convergence and
loss-curve tracking against the oracle (BASELINE config 3 bar: within 1 % over 50 steps)."""
import os

import numpy as np
import pytest
import torch

from dpdist_amd import synth
from dpdist_amd.train import SyntheticDistanceDataset, compose_batch, iter_global_batches


def test_compose_batch_follows_the_reference_recipe():
    N = 64
    B = 3
    data = np.arange(B * 3 * 2 * N * 3, dtype=np.float32).reshape(B, 3 * 2 * N, 3)
    label = np.arange(B * 2 * 2 * N, dtype=np.float32).reshape(B, 2 * 2 * N) + 0.5
    pcA, pcB, lab = compose_batch(data, label, N)
    assert pcA.shape == pcB.shape == (B, N, 3) and lab.shape == (B, N)
    surf, close, far = data[:, :128], data[:, 128:256], data[:, 256:]
    assert np.array_equal(pcA, surf[:, :64])                                  # first N of surface half 1
    assert np.array_equal(pcB[:, :32], surf[:, 64:96])                        # N/2 of surface half 2
    assert np.array_equal(pcB[:, 32:48], close[:, :16])                       # N/4 near
    assert np.array_equal(pcB[:, 48:], far[:, 16:32])                         # N/4 far, slice [N/4:N/2]
    assert not lab[:, :32].any()
    assert np.array_equal(lab[:, 32:48], label[:, :16]) and np.array_equal(lab[:, 48:], label[:, 128 + 16:128 + 32])


def test_dataset_item_format_and_interface():
    ds = SyntheticDistanceDataset(10, 128, 4, "train", seed=0)
    assert ds.num_channel() == 3 and ds.num_batches == 3
    seen = 0
    while ds.has_next_batch():
        d, l = ds.next_batch(augment=True)
        assert d.shape[1:] == (384, 3) and l.shape[1:] == (256,)
        assert (l[:, :128] > 0.001).all() and (l[:, :128] < 0.1).all() and (l[:, 128:] > 0.1).all()
        seen += len(d)
    assert seen == 10
    ds.reset()
    assert ds.has_next_batch()


def test_short_last_batch_keeps_the_static_shape():
    """train_multi_gpu_pc_compare_dist.py:737-766: the batch buffers are persistent; a short last batch (here 4 of 16, i.e.
    fewer than half -- the case a wrap-around pad of `data[:pad]` cannot fill) overwrites rows [0, bsize) only."""
    N, Bg = 64, 16
    ds = SyntheticDistanceDataset(36, 2 * N, Bg, "test", seed=3)
    got = [(a.copy(), b.copy(), l.copy()) for a, b, l in iter_global_batches(ds, Bg, N, False)]
    assert len(got) == 3
    for a, b, l in got:
        assert a.shape == b.shape == (Bg, N, 3) and l.shape == (Bg, N)
    ds.reset()
    ds.batch_idx = 2
    d, lab = ds.next_batch()
    assert len(d) == 4
    # rows 0..3 = the 4 remaining shapes (up to the per-item point permutation: compare as sets of label values),
    # rows 4..15 = the previous batch, untouched
    assert np.array_equal(got[2][0][4:], got[1][0][4:]) and np.array_equal(got[2][2][4:], got[1][2][4:])
    assert not np.array_equal(got[2][0][:4], got[1][0][:4])
    # every shard of an 8-rank run has rows (the round-1 code handed 0-row tensors to ranks 4-7 here)
    for r in range(8):
        assert got[2][0][r * 2:(r + 1) * 2].shape == (2, N, 3)


@pytest.mark.gpu
def test_loss_curve_tracks_oracle_50_steps():
    """fp32 HIP trainer vs the torch-CPU oracle + numpy TF-Adam on a fixed batch: relative loss error < 1 % at every step."""
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer, learning_rate
    from oracle import restate as R
    dev = torch.device("cuda:0")
    B = 2
    pcA, pcB, lab = synth.s2_modelnet_shaped(B, 64, 100)
    W0 = synth.make_weights("wide")
    P = DPDistParams(device=dev)
    P.load_tf_state_dict(W0)
    tr = DPDistTrainer(P, B, base_lr=2e-4, distributed=False)
    Wt = {n: torch.tensor(a, dtype=torch.float32, requires_grad=True) for n, a in W0.items()}
    ms = {n: np.zeros_like(a) for n, a in W0.items()}
    vs = {n: np.zeros_like(a) for n, a in W0.items()}
    cu = lambda a: torch.tensor(a, device=dev)   # noqa: E731
    a, b, l = cu(pcA), cu(pcB), cu(lab)
    ta, tb, tl = torch.tensor(pcA), torch.tensor(pcB), torch.tensor(lab)
    names = sorted(Wt)
    first = last = None
    for t in range(1, 51):
        got = tr.step(a, b, l).cpu().numpy()[0]
        pred, _ = R.get_model(ta, tb, Wt)
        ls, _ = R.get_loss(pred, tl)
        ref = ls.item()
        assert abs(got - ref) <= 0.01 * abs(ref), (t, got, ref)
        gs = torch.autograd.grad(ls, [Wt[n] for n in names])
        for n, g in zip(names, gs):
            R.adam_tf_step(Wt[n].detach().numpy(), g.numpy(), ms[n], vs[n], t, learning_rate(t - 1, 2e-4))
        first = ref if first is None else first
        last = ref
    assert last < 0.7 * first        # and it actually trains


@pytest.mark.gpu
def test_train_loop_converges_and_checkpoints(tmp_path):
    from dpdist_amd.train import train
    ls = train(["--log_dir", str(tmp_path), "--max_epoch", "6", "--batch_size", "32", "--train_shapes", "128",
                "--test_shapes", "32", "--eval_every", "5", "--learning_rate_dpdist", "0.0005"])
    log = open(os.path.join(tmp_path, "log_trainours.txt")).read()
    losses = [float(x.split("mean loss:")[1].split()[0]) for x in log.splitlines() if "---- epoch" in x]
    assert len(losses) == 6 and losses[-1] < losses[0] and np.isfinite(ls)
    ck = np.load(os.path.join(tmp_path, "model.ckpt.npz"))
    base = sorted(synth.make_weights("xavier_tf"))
    # what tf.train.Saver() stores: the variables, the global step, the beta powers and the two Adam slots per variable
    assert sorted(ck.files) == sorted(base + [n + s for n in base for s in ("/Adam", "/Adam_1")] + ["batch", "beta1_power", "beta2_power"])
    assert ck["pc_compare/dpdist_local/mapper_conv1/weights"].shape == (1, 2503, 1, 1024)
    assert float(ck["batch"]) == 24.0
    # resume from the TF container: global step and optimizer state come back
    ls2 = train(["--log_dir", str(tmp_path / "resumed"), "--max_epoch", "1", "--batch_size", "32", "--train_shapes", "128",
                 "--test_shapes", "32", "--eval_every", "5", "--learning_rate_dpdist", "0.0005",
                 "--restore", os.path.join(tmp_path, "model.ckpt")])
    log2 = open(os.path.join(tmp_path, "resumed", "log_trainours.txt")).read()
    assert "restored weights, adam_slots, schedule" in log2 and "(global step 24)" in log2 and "step 28" in log2
    assert np.isfinite(ls2) and ls2 < losses[0]
