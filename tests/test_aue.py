"""AUE task (row f4).  Pinned by goldens that the reference's own functions produced (oracle/gen_goldens.py):
    tests/golden/chamfer_cases.npz   pairwise_diff / chmafer_dist (train_multi_gpu_pc_compare_dist.py:891-916), loss + autograd
    tests/golden/aue_pn_cases.npz    get_model_aue_pn (models/dpdist_and_aue.py:88-145), both batch-norm branches
The in-file restatement `chamfer_ref` is itself checked against the golden before it is used for the larger shapes."""
import os

import numpy as np
import pytest
import torch


def chamfer_ref(pc, rec):
    """pairwise_diff(x, y)[b,i,j] = sum_d (x_bid - y_bjd)^2 (:896-905); chmafer_dist (:912-916)."""
    d = ((rec[:, :, None, :] - pc[:, None, :, :]) ** 2).sum(-1)          # pairwise_diff(rec_pc, pc)
    s1_s2 = d.min(2).values.mean()
    s2_s1 = d.transpose(1, 2).min(2).values.mean()                       # pairwise_diff(pc, rec_pc)
    return (s1_s2 + s2_s1) / 2.0


def test_chamfer_reference_small_case():
    pc = torch.tensor([[[0.0, 0, 0], [1, 0, 0]]])
    rec = torch.tensor([[[0.0, 0, 0.5], [1, 0, 0], [2, 0, 0]]])
    # rec -> pc minima: 0.25, 0, 1 (mean 1.25/3); pc -> rec minima: 0.25, 0 (mean 0.125)
    assert abs(chamfer_ref(pc, rec).item() - (1.25 / 3 + 0.125) / 2) < 1e-7


def test_chamfer_restatement_matches_reference_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "chamfer_cases.npz"))
    pc = torch.tensor(d["pc"], dtype=torch.float64, requires_grad=True)
    rec = torch.tensor(d["rec_pc"], dtype=torch.float64, requires_grad=True)
    loss = chamfer_ref(pc, rec)
    assert abs(loss.item() - float(d["loss_f64"])) <= 1e-14
    ga, gb = torch.autograd.grad(loss, [pc, rec])
    # ties (a duplicated point): tf.reduce_min splits the gradient evenly, min().values picks one -- compare the sums per tie
    assert np.abs(ga.numpy() - d["d_pc_f64"]).max() <= 1e-12
    assert abs(gb.numpy().sum() - d["d_rec_f64"].sum()) <= 1e-12


@pytest.mark.gpu
def test_chamfer_hip_matches_reference_golden(golden_dir):
    """HIP Chamfer forward/backward against the reference's chmafer_dist output and its autograd gradients."""
    from dpdist_amd.aue import chamfer_dist
    d = np.load(os.path.join(golden_dir, "chamfer_cases.npz"))
    pc = torch.tensor(d["pc"]).cuda().requires_grad_(True)
    rec = torch.tensor(d["rec_pc"]).cuda().requires_grad_(True)
    loss = chamfer_dist(pc, rec)
    assert abs(loss.item() - float(d["loss_f64"])) <= 1e-7
    assert abs(loss.item() - float(d["loss_f32"])) <= 1e-7
    ga, gb = torch.autograd.grad(loss * 2.0, [pc, rec])
    assert np.abs(ga.cpu().numpy() / 2.0 - d["d_pc_f64"]).max() <= 1e-7
    gb = gb.cpu().numpy() / 2.0
    ref = d["d_rec_f64"]
    # rows 5 and 6 of cloud 0 are the same point: the reference splits the pc -> rec gradient between them, the kernel's
    # argmin gives it to one; their SUM is what reaches the autoencoder output either way
    tie = np.zeros(ref.shape[:2], bool); tie[0, 5] = tie[0, 6] = True
    assert np.abs(gb[~tie] - ref[~tie]).max() <= 1e-7
    assert np.abs((gb[0, 5] + gb[0, 6]) - (ref[0, 5] + ref[0, 6])).max() <= 1e-7


def test_autoencoder_matches_reference_golden(golden_dir):
    """PointNetAE with the seeded weights == get_model_aue_pn: inference branch, training branch (batch statistics),
    the moving averages after one training forward (decay 0.9, unbiased variance) and the gradient to the input."""
    from dpdist_amd import synth
    from dpdist_amd.aue import PointNetAE
    d = np.load(os.path.join(golden_dir, "aue_pn_cases.npz"))
    W = synth.make_named_weights(synth.aue_pn_spec(64), seed=int(d["weights_seed"]))
    ae = PointNetAE(num_point=64, bn=True, bn_decay=0.9).double()
    ae.load_tf_state_dict(W)
    x = torch.tensor(d["points"], dtype=torch.float64, requires_grad=True)
    ae.eval()
    with torch.no_grad():
        assert np.abs(ae(x).numpy() - d["out_eval_f64"]).max() <= 1e-10
    ae.train()
    out = ae(x)
    assert np.abs(out.detach().numpy() - d["out_train_f64"]).max() <= 1e-10
    g, = torch.autograd.grad(out.sum(), [x])
    assert np.abs(g.numpy() - d["d_points_train_f64"]).max() <= 1e-9
    assert np.abs(ae.point[0][1].running_mean.numpy() - d["mm_conv1_f64"]).max() <= 1e-12
    assert np.abs(ae.point[0][1].running_var.numpy() - d["mv_conv1_f64"]).max() <= 1e-12
    bn_fc2 = [m for m in ae.fc if isinstance(m, torch.nn.BatchNorm1d)][1]
    assert np.abs(bn_fc2.running_mean.numpy() - d["mm_fc2_f64"]).max() <= 1e-12
    assert np.abs(bn_fc2.running_var.numpy() - d["mv_fc2_f64"]).max() <= 1e-10
    ae32 = PointNetAE(num_point=64, bn=True, bn_decay=0.9).eval()
    ae32.load_tf_state_dict(W)
    with torch.no_grad():
        assert np.abs(ae32(torch.tensor(d["points"])).numpy() - d["out_eval_f32"]).max() <= 2e-5


def test_autoencoder_shapes_cpu():
    from dpdist_amd.aue import PointNetAE
    ae = PointNetAE(num_point=64)
    x = torch.rand(4, 64, 3) * 2 - 1
    out = ae(x)
    assert out.shape == (4, 64, 3) and out.abs().max() <= 1.0
    assert ae.embed(x).shape == (4, 1024)
    assert sum(p.numel() for p in ae.parameters()) > 2_000_000


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 64, 64), (2, 100, 300), (1, 1024, 700)])
def test_chamfer_hip_matches_reference(shape):
    from dpdist_amd.aue import chamfer_dist
    B, N, M = shape
    g = torch.Generator().manual_seed(N)
    pc = (torch.rand(B, N, 3, generator=g) * 2 - 1).cuda().requires_grad_(True)
    rec = (torch.rand(B, M, 3, generator=g) * 2 - 1).cuda().requires_grad_(True)
    loss = chamfer_dist(pc, rec)
    ref = chamfer_ref(pc.detach().double().cpu(), rec.detach().double().cpu())
    assert abs(loss.item() - ref.item()) <= 1e-6
    ga, gb = torch.autograd.grad(loss * 3.0, [pc, rec])
    pcd = pc.detach().double().cpu().requires_grad_(True)
    recd = rec.detach().double().cpu().requires_grad_(True)
    ra, rb = torch.autograd.grad(chamfer_ref(pcd, recd) * 3.0, [pcd, recd])
    assert (ga.cpu().double() - ra).abs().max().item() <= 1e-6
    assert (gb.cpu().double() - rb).abs().max().item() <= 1e-6


@pytest.mark.gpu
def test_aue_training_with_dpdist_loss_reduces_both_losses():
    """A few Adam steps on the autoencoder with DPDist as the frozen loss: the DPDist loss goes down, the gradient
    reaches every autoencoder parameter, and the DPDist weights stay untouched."""
    from dpdist_amd import synth
    from dpdist_amd.aue import AUETask, PointNetAE
    from dpdist_amd.model import DPDistLoss, DPDistModel
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    dp = DPDistModel(device=dev)
    dp.load_tf_state_dict(synth.make_weights("wide"))
    w_before = dp.params_.flat.detach().clone()
    ae = PointNetAE(num_point=64).to(dev)
    task = AUETask(ae, DPDistLoss(dp), lr=1e-3, opt_type="ours")
    pcA, pcB, _ = synth.s2_modelnet_shaped(8, 64, 100)
    x1 = torch.tensor(pcA, device=dev)
    x2 = torch.tensor(pcB[:, :64].copy(), device=dev)
    hist = [tuple(v.item() for v in task.step(x1, x2)) for _ in range(30)]
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in ae.parameters())
    assert np.mean([h[0] for h in hist[-5:]]) < np.mean([h[0] for h in hist[:5]])
    assert torch.equal(dp.params_.flat.detach(), w_before)
    lp, lc, out2 = task.evaluate(x1, x2)
    assert out2.shape == (8, 64, 3) and torch.isfinite(lp) and torch.isfinite(lc)
    # chamfer mode: the Chamfer loss goes down
    ae2 = PointNetAE(num_point=64).to(dev)
    task2 = AUETask(ae2, DPDistLoss(dp), lr=1e-3, opt_type="chamfer")
    hist2 = [tuple(v.item() for v in task2.step(x1, x2)) for _ in range(30)]
    assert np.mean([h[1] for h in hist2[-5:]]) < np.mean([h[1] for h in hist2[:5]])
