"""The oracle (oracle/restate.py) against the golden vectors produced by running the reference's own
Python under oracle/tfstub (oracle/gen_goldens.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from dpdist_amd import synth
from oracle import restate as R


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("m", [8, 5])
@pytest.mark.parametrize("dt,tag,tol", [(torch.float32, "f32", 2e-6), (torch.float64, "f64", 1e-12)])
def test_fv_matches_reference(golden_dir, m, dt, tag, tol):
    d = _load(golden_dir, "fv_cases.npz")
    fv = R.mfv3d(torch.tensor(d["points"], dtype=dt), m=m, sigma=0.125).numpy()
    ref = d["fv_m%d_%s" % (m, tag)]
    assert fv.shape == ref.shape
    assert np.abs(fv - ref).max() <= tol


@pytest.mark.parametrize("case", ["s1", "boundary"])
@pytest.mark.parametrize("wk", ["xavier_tf", "wide"])
def test_forward_matches_reference(golden_dir, case, wk):
    d = _load(golden_dir, "path_fwd_%s_%s.npz" % (case, wk))
    for dt, tag, tol in ((torch.float32, "f32", 2e-5), (torch.float64, "f64", 1e-10)):
        W = R.as_torch_weights(synth.make_weights(wk), dt)
        pred, aux = R.get_model(torch.tensor(d["pcA"], dtype=dt), torch.tensor(d["pcB"], dtype=dt), W)
        for n in ("pred_listAB", "pred_listBA"):
            ref = d[n + "_" + tag]
            got = pred[n].numpy()
            assert got.shape == ref.shape == (2, 64, 1, 3)
            assert np.abs(got - ref).max() <= tol, (n, tag)
        if tag == "f32" and wk == "wide":
            sel = d["emb_sel"]
            assert np.abs(aux["embA"].numpy()[sel[:, 0], sel[:, 1]] - d["embA_rows"]).max() <= 2e-6
            assert np.abs(aux["embB"].numpy()[sel[:, 0], sel[:, 1]] - d["embB_rows"]).max() <= 2e-6


def test_inputs_regenerate_from_seed(golden_dir):
    d = _load(golden_dir, "path_fwd_s1_wide.npz")
    pcA, pcB = synth.s1_random_patches(32, 64, seed=0)
    assert np.array_equal(pcA[:2], d["pcA"]) and np.array_equal(pcB[:2], d["pcB"])
    d = _load(golden_dir, "path_fwd_boundary_wide.npz")
    pcA, pcB = synth.boundary_cloud(2, 64, seed=7)
    assert np.array_equal(pcA, d["pcA"]) and np.array_equal(pcB, d["pcB"])
    d = _load(golden_dir, "path_bwd_s2_wide.npz")
    pcA, pcB, lab = synth.s2_modelnet_shaped(2, 64, seed=100)
    assert np.array_equal(pcA, d["pcA"]) and np.array_equal(pcB, d["pcB"]) and np.array_equal(lab, d["labels"])


def test_variable_contract(golden_dir):
    """The reference creates exactly these 8 variables / named outputs (checkpoint + graph-name contract)."""
    d = _load(golden_dir, "path_fwd_s1_wide.npz")
    names = list(d["var_names"])
    W = synth.make_weights("wide")
    assert names == sorted(W)
    for n, shp in zip(names, d["var_shapes"]):
        assert list(W[n].shape) == [s for s in shp[:W[n].ndim]]
    assert "pc_compare/output1" in d["named_outputs"] and "pc_compare/output2" in d["named_outputs"]


@pytest.mark.parametrize("m", [8, 5])
def test_small_mlp_and_m5(golden_dir, m):
    d = _load(golden_dir, "path_fwd_mlp64_m%d.npz" % m)
    W = R.as_torch_weights(synth.make_weights("wide", mlp=(64, 64, 64)), torch.float64)
    pred, _ = R.get_model(torch.tensor(d["pcA"], dtype=torch.float64), torch.tensor(d["pcB"], dtype=torch.float64), W, m=m)
    assert np.abs(pred["pred_listAB"].numpy() - d["pred_listAB_f64"]).max() <= 1e-10
    assert np.abs(pred["pred_listBA"].numpy() - d["pred_listBA_f64"]).max() <= 1e-10


def test_losses_and_gradients_match_reference(golden_dir):
    d = _load(golden_dir, "path_bwd_s2_wide.npz")
    for dt, tag, tol in ((torch.float64, "f64", 1e-9), (torch.float32, "f32", 3e-4)):
        W = R.as_torch_weights(synth.make_weights("wide"), dt, requires_grad=True)
        pcA = torch.tensor(d["pcA"], dtype=dt, requires_grad=True)
        pcB = torch.tensor(d["pcB"], dtype=dt, requires_grad=True)
        noise = torch.tensor(d["noise"], dtype=dt, requires_grad=True)
        pred, _ = R.get_model(pcA, pcB, W, add_noise=noise)
        ls, lp = R.get_loss(pred, torch.tensor(d["labels"], dtype=dt))
        assert abs(ls.item() - float(d["loss_samples_" + tag])) <= tol
        assert abs(lp.item() - float(d["loss_pred_" + tag])) <= tol
        gA, gB, gN = torch.autograd.grad(lp, [pcA, pcB, noise], retain_graph=True)
        for g, n in ((gA, "d_pcA"), (gB, "d_pcB"), (gN, "d_noise")):
            ref = d[n + "_" + tag]
            assert np.abs(g.numpy() - ref).max() <= tol * max(1.0, np.abs(ref).max()), n
        names = sorted(W)
        gw = torch.autograd.grad(ls, [W[n] for n in names])
        for n, g in zip(names, gw):
            short = n.split("/")[-2][-1] + ("w" if n.endswith("weights") else "b")
            g = g.numpy()
            g2 = g.reshape(-1, g.shape[-1]) if g.ndim == 4 else g
            nrm = float(d["g%s_norm_%s" % (short, tag)])
            assert abs(np.sqrt((g2.astype(np.float64) ** 2).sum()) - nrm) <= tol * max(1.0, nrm)
            if g.ndim == 4:
                assert np.abs(g2[:16, :16] - d["g%s_corner_%s" % (short, tag)]).max() <= tol * max(1.0, nrm)
                assert np.abs(g2[-16:, -16:] - d["g%s_tail_%s" % (short, tag)]).max() <= tol * max(1.0, nrm)
                assert np.abs(g2.sum(0) - d["g%s_colsum_%s" % (short, tag)]).max() <= tol * max(1.0, nrm) * 30
            else:
                assert np.abs(g2 - d["g%s_%s" % (short, tag)]).max() <= tol * max(1.0, nrm)


def test_gradient_routes(golden_dir):
    """d mean(AB)/d pcA flows only through the encoder (== gradient w.r.t. add_noise);
    d mean(BA)/d add_noise is exactly zero."""
    d = _load(golden_dir, "path_bwd_s2_wide.npz")
    dt = torch.float64
    W = R.as_torch_weights(synth.make_weights("wide"), dt)
    pcA = torch.tensor(d["pcA"], dtype=dt, requires_grad=True)
    pcB = torch.tensor(d["pcB"], dtype=dt)
    noise = torch.zeros_like(pcA, requires_grad=True)
    pred, _ = R.get_model(pcA, pcB, W, add_noise=noise)
    gA, gN = torch.autograd.grad(pred["pred_listAB"][..., 0].mean(), [pcA, noise], retain_graph=True)
    assert torch.equal(gA, gN)
    (gN2,) = torch.autograd.grad(pred["pred_listBA"][..., 0].mean(), [noise], allow_unused=True)
    assert gN2 is None or float(gN2.abs().max()) == 0.0


def test_voxel_per_axis_form_equals_all_centres():
    """The HIP kernel tests each axis against the m cell intervals (lo, hi] with the reference's own
    float32 comparisons instead of all m^3 centres; both forms must agree everywhere, including on cell
    faces and one ulp either side.  (A closed form ceil((q+1)/0.25)-1 is NOT exact in float32: q = 1+ulp
    rounds q+1 to 2.0 and would be accepted, while the reference rejects it.)"""
    rng = np.random.default_rng(0)
    bs = synth.BOUNDARY_SET
    q = np.concatenate([rng.uniform(-1.2, 1.2, (4000, 3)).astype(np.float32),
                        bs[rng.integers(0, 11, (2000, 3))],
                        np.nextafter(bs, np.float32(2))[rng.integers(0, 11, (1000, 3))],
                        np.nextafter(bs, np.float32(-2))[rng.integers(0, 11, (1000, 3))]]).astype(np.float32)
    for m in (8, 5):
        v, mask, _ = R.voxel_lookup(torch.tensor(q)[None], m=m)
        ax = R.grid_axis(m).astype(np.float32)
        g = np.float32(abs(ax[0] - ax[1]) / np.float32(2))
        idx = np.full(q.shape, -1, np.int64)
        for i in range(m - 1, -1, -1):
            hit = (q > ax[i] - g) & (q <= ax[i] + g)
            idx[hit] = i
        valid = (idx >= 0).all(1)
        vv = idx[:, 1] * m * m + idx[:, 0] * m + idx[:, 2]
        assert np.array_equal(valid, mask[0].numpy() > 0)
        assert np.array_equal(vv[valid], v[0].numpy()[valid])


def test_nan_semantics_far_point():
    """a cloud point far from every Gaussian underflows every p -> 0/0 = NaN (reference behaviour)."""
    pts = torch.zeros(1, 64, 3)
    pts[0, 0] = 3.0
    assert torch.isnan(R.mfv3d(pts)).any()
    pts[0, 0] = 1.2
    assert not torch.isnan(R.mfv3d(pts)).any()


def test_restatement_reproduces_the_optimizer_steps_fixture(golden_dir):
    """F4 (tests/golden/step_adam.npz): three steps of the trainer's own optimizer assembly -- two towers, compute_gradients,
    average_gradients, get_learning_rate (staircase, decays after step 2), AdamOptimizer.apply_gradients
    (train_multi_gpu_pc_compare_dist.py:216,241-251,274-277,301,936-990), run by oracle/gen_goldens.py:fx_step -- pin the
    restatement's loss/gradient, adam_tf_step and learning_rate."""
    import math
    d = np.load(os.path.join(golden_dir, "step_adam.npz"))
    mlp = (64, 64, 64)
    W0 = synth.make_weights("wide", mlp=mlp)
    Wt = {n: torch.tensor(a, dtype=torch.float64, requires_grad=True) for n, a in W0.items()}
    ms = {n: np.zeros_like(a, dtype=np.float64) for n, a in W0.items()}
    vs = {n: np.zeros_like(a, dtype=np.float64) for n, a in W0.items()}
    pcA, pcB, lab = (torch.tensor(d[k], dtype=torch.float64) for k in ("pcA", "pcB", "labels"))
    base, dstep, drate = float(d["base_lr"]), int(d["decay_step"]), float(d["decay_rate"])
    for t in range(1, 4):
        lr = R.learning_rate(t - 1, base, dstep, drate)
        assert math.isclose(lr, float(d["lr_f64"][t - 1]), rel_tol=1e-12)
        pred, _ = R.get_model(pcA, pcB, Wt)           # full batch: the mean of two equal tower means
        ls, _ = R.get_loss(pred, lab)
        assert abs(ls.item() - float(d["loss_samples_f64"][t - 1])) <= 1e-10
        names = sorted(Wt)
        gs = torch.autograd.grad(ls, [Wt[n] for n in names])
        for n, g in zip(names, gs):
            R.adam_tf_step(Wt[n].detach().numpy(), g.numpy(), ms[n], vs[n], t, lr)
    for n in Wt:
        short = n.split("/")[-2][-1] + ("w" if n.endswith("weights") else "b")
        assert np.abs(Wt[n].detach().numpy() - d["final_%s_f64" % short]).max() <= 1e-9, n
