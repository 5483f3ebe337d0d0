"""GPU parity tests: the HIP path (through the C ABI) against the oracle and the reference's golden vectors.

Bar (BASELINE.json north_star / SURVEY 8d): fp32, max abs err <= 1e-4 on predicted distances; on the `wide`
weight set additionally max rel err <= 1e-4 on unsaturated outputs.  Integer work (voxel ids, masks) is bit exact.
"""
import os
import time

import numpy as np
import pytest
import torch

from dpdist_amd import synth

pytestmark = pytest.mark.gpu

ABS_TOL = 1e-4
REL_TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    from dpdist_amd import lib
    lib.load()     # raises if the HIP extension is missing -- never fall back
    return torch.device("cuda:0")


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _cu(a, dev):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)


def _model(dev, wk="wide", m=8, mlp=(1024, 1024, 1024)):
    from dpdist_amd.model import DPDistModel
    mod = DPDistModel(Embedding_Size=m ** 3, k=5, localSNmlp=mlp, device=dev)
    mod.load_tf_state_dict(synth.make_weights(wk, mlp=mlp))
    return mod


# ------------------------------------------------------------------------------------------------ encoder
@pytest.mark.parametrize("m", [8, 5])
def test_mfv3d_golden(dev, golden_dir, m):
    from dpdist_amd import ops
    d = _g(golden_dir, "fv_cases.npz")
    fv = ops.mfv3d_fwd(_cu(d["points"], dev), m, 0.125).cpu().numpy()
    assert np.abs(fv - d["fv_m%d_f64" % m]).max() <= 3e-6
    assert np.abs(fv - d["fv_m%d_f32" % m]).max() <= 3e-6


def test_mfv3d_vs_oracle_full_batch(dev):
    from dpdist_amd import ops
    from oracle import restate as R
    pcA, pcB = synth.s1_random_patches(32, 64, 0)
    pts = np.concatenate([pcA, pcB])
    fv = ops.mfv3d_fwd(_cu(pts, dev), 8, 0.125).cpu().numpy()
    ref = R.mfv3d(torch.tensor(pts, dtype=torch.float64)).numpy()
    assert np.abs(fv - ref).max() <= 3e-6
    # L2 normalisation property: every channel has unit norm over the Gaussian axis
    assert np.abs(np.sqrt((fv.astype(np.float64) ** 2).sum(1)) - 1.0).max() <= 1e-5


def test_mfv3d_nan_semantics(dev):
    """Reference behaviour: a point far outside underflows every pdf -> NaN; (1.2,1.2,1.2) stays finite."""
    from dpdist_amd import ops
    pts = torch.zeros(2, 64, 3, device=dev)
    pts[0, 0] = 3.0
    pts[1, 0] = 1.2
    fv = ops.mfv3d_fwd(pts, 8, 0.125)
    assert torch.isnan(fv[0]).any() and not torch.isnan(fv[1]).any()


def test_mfv3d_permutation_invariance(dev):
    from dpdist_amd import ops
    pcA, _ = synth.s1_random_patches(8, 64, 3)
    perm = np.random.default_rng(0).permutation(64)
    a = ops.mfv3d_fwd(_cu(pcA, dev), 8, 0.125)
    b = ops.mfv3d_fwd(_cu(pcA[:, perm], dev), 8, 0.125)
    assert (a - b).abs().max().item() <= 2e-6


def test_mfv3d_backward_vs_oracle(dev):
    """d fv -> d points against the float64 oracle.  The power-1/2 normalisation is ill-conditioned where a statistic
    is ~0 (dv = ds * 0.5/sqrt|v| with |v| down to 1e-12: float32 round-off of v is amplified without bound, in the
    reference too), so the upstream gradient is zeroed on entries with |fv| < 2e-3 for the tight check; the
    unrestricted case is held to the float32 oracle's own distance from float64."""
    from dpdist_amd import ops
    from oracle import restate as R
    rng = np.random.default_rng(4)
    pcA, _ = synth.s1_random_patches(4, 64, 5)
    pcA[3] = pcA[3, :1]                    # all points identical: every max/min is a 64-way tie
    dfv_full = rng.standard_normal((4, 512, 20)).astype(np.float32)
    fv64 = R.mfv3d(torch.tensor(pcA, dtype=torch.float64)).numpy()
    for restricted in (True, False):
        dfv = np.where(np.abs(fv64) > 2e-3, dfv_full, 0).astype(np.float32) if restricted else dfv_full
        grads = {}
        for dt in (torch.float64, torch.float32):
            p = torch.tensor(pcA, dtype=dt, requires_grad=True)
            (R.mfv3d(p) * torch.tensor(dfv, dtype=dt)).sum().backward()
            grads[dt] = p.grad.numpy().astype(np.float64)
        ref = grads[torch.float64]
        got = ops.mfv3d_bwd(_cu(pcA, dev), _cu(dfv, dev), 8, 0.125).cpu().numpy()          # sliced: 4 workgroups per cloud
        mono = ops.mfv3d_bwd(_cu(pcA, dev), _cu(dfv, dev), 8, 0.125, sliced=False).cpu().numpy()
        assert np.isfinite(got).all()
        # the two launch forms differ only by the association of the per-Gaussian sums over the points (amplified without
        # bound in the ill-conditioned unrestricted case, where BOTH are held to the oracle bars below)
        if restricted:
            assert np.abs(got - mono).max() <= 1e-4 * max(1.0, np.abs(mono).max())
        for c in range(4):
            scale = max(1.0, np.abs(ref[c]).max())
            err_m = np.abs(mono[c] - ref[c]).max()
            bar = 1e-4 * scale if restricted else max(10.0 * np.abs(grads[torch.float32][c] - ref[c]).max(), 2e-4 * scale)
            assert err_m <= bar, ("monolithic", c, err_m, bar)
        for c in range(4):
            scale = max(1.0, np.abs(ref[c]).max())
            err = np.abs(got[c] - ref[c]).max()
            if restricted:
                assert err <= 1e-4 * scale, (c, err, scale)
            else:
                err_f32_oracle = np.abs(grads[torch.float32][c] - ref[c]).max()
                assert err <= max(10.0 * err_f32_oracle, 2e-4 * scale), (c, err, err_f32_oracle, scale)


# ------------------------------------------------------------------------------------------------ lookup + gather
@pytest.mark.parametrize("case", ["s1", "boundary"])
@pytest.mark.parametrize("m", [8, 5])
def test_patch_rows_bit_exact(dev, case, m):
    from dpdist_amd import ops
    from oracle import restate as R
    pcA, pcB = synth.s1_random_patches(8, 64, 0) if case == "s1" else synth.boundary_cloud(8, 64, 7)
    fv = torch.tensor(np.random.default_rng(1).standard_normal((8, m ** 3, 20)).astype(np.float32))
    X, mask, vox = ops.patch_rows_fwd(_cu(pcB, dev), fv.to(dev), m, 5)
    v, msk, loc = R.voxel_lookup(torch.tensor(pcB), m)
    emb = R.local_window(fv, m, 5)
    rows = torch.gather(emb, 1, v[..., None].expand(-1, -1, emb.shape[-1])).reshape(8 * 64, -1).numpy()
    X = X.cpu().numpy()
    assert np.array_equal(mask.cpu().numpy(), msk.reshape(-1).numpy())
    assert np.array_equal(vox.cpu().numpy().astype(np.int64), v.reshape(-1).numpy())
    assert np.array_equal(X[:, :2500], rows)                          # pure data movement: bit exact
    assert np.array_equal(X[:, 2500:2503], loc.reshape(-1, 3).numpy())
    assert not X[:, 2503:].any()


def test_patch_rows_backward_is_transpose(dev):
    """<gather(fv), dX> == <fv, scatter(dX)> (adjoint identity) and dq = the xyz columns."""
    from dpdist_amd import ops
    rng = np.random.default_rng(2)
    _, pcB = synth.s1_random_patches(4, 64, 0)
    fv = _cu(rng.standard_normal((4, 512, 20)), dev)
    dX = _cu(rng.standard_normal((256, 2528)), dev)
    X, mask, vox = ops.patch_rows_fwd(_cu(pcB, dev), fv, 8, 5)
    dq, dfv = ops.patch_rows_bwd(dX, vox, 4, 64, 8, 5)
    lhs = (X[:, :2500].double() * dX[:, :2500].double()).sum().item()
    rhs = (fv.double() * dfv.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-6 * max(1.0, abs(lhs))
    assert torch.equal(dq.reshape(-1, 3), dX[:, 2500:2503])


# ------------------------------------------------------------------------------------------------ GEMM building block
@pytest.mark.parametrize("tile", [3, 8, 9, 30, 31, 32, 33])
@pytest.mark.parametrize("mode", ["NN", "NT", "TN"])
def test_gemm_f32(dev, tile, mode):
    from dpdist_amd import lib as L, ops
    rng = np.random.default_rng(10)
    M, N, K = 328, 196, 224            # ragged in M and N against every tile size; asymmetric operands catch transposes
    A = rng.standard_normal((M, K)).astype(np.float32)
    Bm = rng.standard_normal((K, N)).astype(np.float32)
    ref = A.astype(np.float64) @ Bm.astype(np.float64)
    a = _cu(A.T if mode == "TN" else A, dev)
    b = _cu(Bm.T if mode == "NT" else Bm, dev)
    for split in (1, 3):
        c = ops.gemm_f32(a, b, transA=(mode == "TN"), transB=(mode == "NT"), tile=tile, split_k=split).cpu().numpy()
        assert np.abs(c - ref).max() <= 2e-4, (mode, tile, split)
    bias = _cu(rng.standard_normal(N), dev)
    gate = _cu(rng.standard_normal((M, N)), dev)
    c = ops.gemm_f32(a, b, transA=(mode == "TN"), transB=(mode == "NT"), tile=tile, bias=bias, epilogue=2).cpu().numpy()
    assert np.abs(c - np.maximum(ref + bias.cpu().numpy(), 0)).max() <= 2e-4
    c = ops.gemm_f32(a, b, transA=(mode == "TN"), transB=(mode == "NT"), tile=tile, gate=gate, epilogue=3, split_k=2).cpu().numpy()
    assert np.abs(c - ref * (gate.cpu().numpy() > 0)).max() <= 2e-4


def test_gemm_layer1_shape_is_fmaf_exact(dev):
    """fp32 MFMA accumulates like an fmaf chain: compare with float64 at the real layer-1 shape."""
    from dpdist_amd import ops
    rng = np.random.default_rng(11)
    A = rng.standard_normal((512, 2528)).astype(np.float32) * 0.05
    W = rng.standard_normal((2528, 1024)).astype(np.float32) * 0.3
    c = ops.gemm_f32(_cu(A, dev), _cu(W, dev), tile=32).cpu().numpy()
    ref = A.astype(np.float64) @ W.astype(np.float64)
    assert np.abs(c - ref).max() <= 5e-5


# ------------------------------------------------------------------------------------------------ module contract, forward
def _check_pred(got, ref, wide):
    """north_star bar: max abs err <= 1e-4.  On the `wide` weight set (outputs spread over [0,2]) additionally a
    relative criterion so that the absolute bar is not vacuous: err <= 1e-5 + 1e-4*|ref| on unsaturated outputs
    (the 1e-5 floor is float32 round-off of a K=2503 dot product whose terms are O(1))."""
    err = np.abs(got - ref)
    assert err.max() <= ABS_TOL, err.max()
    if wide:
        uns = (ref > 0.0) & (ref < 2.0)
        assert (err[uns] <= 1e-5 + REL_TOL * np.abs(ref[uns])).all(), (err[uns] - REL_TOL * np.abs(ref[uns])).max()
        sat = ~uns
        assert err[sat].max() <= 1e-5 if sat.any() else True


@pytest.mark.parametrize("case", ["s1", "boundary"])
@pytest.mark.parametrize("wk", ["xavier_tf", "wide"])
def test_forward_golden(dev, golden_dir, case, wk):
    d = _g(golden_dir, "path_fwd_%s_%s.npz" % (case, wk))
    mod = _model(dev, wk)
    with torch.no_grad():
        ps = mod(_cu(d["pcA"], dev), _cu(d["pcB"], dev))
    for n in ("pred_listAB", "pred_listBA"):
        got = ps[n].cpu().numpy()
        assert got.shape == (2, 64, 1, 3)
        _check_pred(got, d[n + "_f64"], wk == "wide")
        _check_pred(got, d[n + "_f32"], wk == "wide")


@pytest.mark.parametrize("m", [8, 5])
def test_forward_golden_mlp64(dev, golden_dir, m):
    d = _g(golden_dir, "path_fwd_mlp64_m%d.npz" % m)
    mod = _model(dev, "wide", m=m, mlp=(64, 64, 64))
    with torch.no_grad():
        ps = mod(_cu(d["pcA"], dev), _cu(d["pcB"], dev))
    _check_pred(ps["pred_listAB"].cpu().numpy(), d["pred_listAB_f64"], True)
    _check_pred(ps["pred_listBA"].cpu().numpy(), d["pred_listBA_f64"], True)


@pytest.mark.parametrize("wk", ["xavier_tf", "wide"])
def test_forward_config2_full_batch_vs_oracle(dev, wk):
    """BASELINE config 2: B=32 S1 'random patches', fp32 forward, numerics within 1e-4 of the oracle."""
    from oracle import restate as R
    pcA, pcB = synth.s1_random_patches(32, 64, 0)
    W = synth.make_weights(wk)
    ref, _ = R.get_model(torch.tensor(pcA), torch.tensor(pcB), R.as_torch_weights(W))
    mod = _model(dev, wk)
    with torch.no_grad():
        ps = mod(_cu(pcA, dev), _cu(pcB, dev))
    for n in ("pred_listAB", "pred_listBA"):
        _check_pred(ps[n].cpu().numpy(), ref[n].numpy(), wk == "wide")
    if wk == "wide":   # both relu6 saturations are exercised
        ab = ps["pred_listAB"][..., 0].cpu().numpy()
        assert (ab == 0).mean() > 0.05 and (ab == 2.0).any()


def test_forward_properties_full_size(dev):
    """Size-independent properties at B=64: permuting cloud A's points leaves AB unchanged and permutes BA;
    points outside the cube predict exactly 0."""
    pcA, pcB, _ = synth.s2_modelnet_shaped(64, 64, 100)
    pcB[:, 5] = 1.5
    mod = _model(dev, "wide")
    perm = np.random.default_rng(1).permutation(64)
    with torch.no_grad():
        a = mod(_cu(pcA, dev), _cu(pcB, dev))
        b = mod(_cu(pcA[:, perm], dev), _cu(pcB, dev))
    assert (a["pred_listAB"] - b["pred_listAB"]).abs().max().item() <= 2e-4
    assert (a["pred_listBA"][:, perm] - b["pred_listBA"]).abs().max().item() <= 2e-4
    assert a["pred_listAB"][:, 5].abs().max().item() == 0.0


# ------------------------------------------------------------------------------------------------ losses + gradients
def test_losses_and_input_gradients_golden(dev, golden_dir):
    """As-loss mode (iterative_PCRNet_ours.py:248-257): d loss_pred / d (input1, input2, add_noise)."""
    from dpdist_amd import model as M
    d = _g(golden_dir, "path_bwd_s2_wide.npz")
    mod = _model(dev, "wide")
    pcA = _cu(d["pcA"], dev).requires_grad_(True)
    pcB = _cu(d["pcB"], dev).requires_grad_(True)
    noise = _cu(d["noise"], dev).requires_grad_(True)
    M.reset_default_graph()
    ps = mod(pcA, pcB, add_noise=noise)
    ls_t, lp = M.get_loss(ps, {}, _cu(d["labels"], dev))
    ls = M.get_collection("loss_samples")[0]
    assert abs(ls.item() - float(d["loss_samples_f64"])) <= 2e-5
    assert abs(lp.item() - float(d["loss_pred_f64"])) <= 2e-5
    assert ls_t.shape == (2, 64)
    gA, gB, gN = torch.autograd.grad(lp, [pcA, pcB, noise])
    for g, n in ((gA, "d_pcA"), (gB, "d_pcB"), (gN, "d_noise")):
        ref = d[n + "_f64"]
        # float32 evaluation of the reference itself (golden *_f32) sits this far from float64; allow 4x that
        bar = max(4.0 * np.abs(d[n + "_f32"] - ref).max(), 2e-4 * max(1.0, np.abs(ref).max()))
        assert np.abs(g.cpu().numpy() - ref).max() <= bar, (n, np.abs(g.cpu().numpy() - ref).max(), bar)


def test_weight_gradients_golden(dev, golden_dir):
    """Training mode (train_multi_gpu_pc_compare_dist.py:274-277): d loss_samples / d the 8 variables, TF layout."""
    from dpdist_amd import model as M
    d = _g(golden_dir, "path_bwd_s2_wide.npz")
    mod = _model(dev, "wide")
    M.reset_default_graph()
    ps = mod(_cu(d["pcA"], dev), _cu(d["pcB"], dev), add_noise=_cu(d["noise"], dev))
    M.get_loss(ps, {}, _cu(d["labels"], dev))
    ls = M.get_collection("loss_samples")[0]
    (gflat,) = torch.autograd.grad(ls, [mod.params_.flat])
    gsd = mod.params_.tf_state_dict(gflat)
    for n, g in gsd.items():
        short = n.split("/")[-2][-1] + ("w" if n.endswith("weights") else "b")
        g2 = g.reshape(-1, g.shape[-1]) if g.ndim == 4 else g
        nrm = float(d["g%s_norm_f64" % short])
        tol = 2e-4 * max(1.0, nrm)
        assert abs(np.sqrt((g2.astype(np.float64) ** 2).sum()) - nrm) <= tol, n
        if g.ndim == 4:
            assert np.abs(g2[:16, :16] - d["g%s_corner_f64" % short]).max() <= tol, n
            assert np.abs(g2[-16:, -16:] - d["g%s_tail_f64" % short]).max() <= tol, n
            assert np.abs(g2.sum(0) - d["g%s_colsum_f64" % short]).max() <= tol * 30, n
            assert np.abs(g2.sum(1) - d["g%s_rowsum_f64" % short]).max() <= tol * 30, n
        else:
            assert np.abs(g2 - d["g%s_f64" % short]).max() <= tol, n


def test_trainer_steps_vs_oracle(dev):
    """Three full training steps (fwd, L1 loss, bwd on the AB half, TF-form Adam) against the oracle + numpy Adam."""
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer, learning_rate
    from oracle import restate as R
    B = 4
    pcA, pcB, lab = synth.s2_modelnet_shaped(B, 64, 100)
    W0 = synth.make_weights("wide")
    P = DPDistParams(device=dev)
    P.load_tf_state_dict(W0)
    tr = DPDistTrainer(P, B, base_lr=1e-3, distributed=False)
    Wt = {n: torch.tensor(a, dtype=torch.float64, requires_grad=True) for n, a in W0.items()}
    ms = {n: np.zeros_like(a, dtype=np.float64) for n, a in W0.items()}
    vs = {n: np.zeros_like(a, dtype=np.float64) for n, a in W0.items()}
    for t in range(1, 4):
        loss = tr.step(_cu(pcA, dev), _cu(pcB, dev), _cu(lab, dev)).cpu().numpy()
        pred, _ = R.get_model(torch.tensor(pcA, dtype=torch.float64), torch.tensor(pcB, dtype=torch.float64), Wt)
        ls, lp = R.get_loss(pred, torch.tensor(lab, dtype=torch.float64))
        assert abs(loss[0] - ls.item()) <= 5e-5 and abs(loss[1] - lp.item()) <= 5e-5, (t, loss, ls.item())
        names = sorted(Wt)
        gs = torch.autograd.grad(ls, [Wt[n] for n in names])
        for n, g in zip(names, gs):
            p = Wt[n].detach().numpy()
            R.adam_tf_step(p, g.numpy(), ms[n], vs[n], t, learning_rate(t - 1, 1e-3))
    got = P.tf_state_dict()
    for n in Wt:
        ref = Wt[n].detach().numpy()
        assert np.abs(got[n] - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), n
        assert np.abs(got[n] - W0[n]).max() > 0 or n.endswith("biases")


def test_trainer_steps_vs_reference_optimizer_fixture(dev, golden_dir):
    """F4: the HIP trainer (forward, L1 loss, backward, TF-form Adam kernel, staircase learning rate) against three steps of
    the reference's own optimizer assembly (tests/golden/step_adam.npz; decoder 64-64-64, lr halves after step 2)."""
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    d = _g(golden_dir, "step_adam.npz")
    mlp = (64, 64, 64)
    P = DPDistParams(mlp=mlp, device=dev)
    P.load_tf_state_dict(synth.make_weights("wide", mlp=mlp))
    tr = DPDistTrainer(P, 4, base_lr=float(d["base_lr"]), decay_step=int(d["decay_step"]), decay_rate=float(d["decay_rate"]),
                       distributed=False)
    a, b, l = _cu(d["pcA"], dev), _cu(d["pcB"], dev), _cu(d["labels"], dev)
    for t in range(3):
        loss = tr.step(a, b, l).cpu().numpy()
        assert abs(loss[0] - d["loss_samples_f64"][t]) <= 5e-5, (t, loss[0], d["loss_samples_f64"][t])
    got = P.tf_state_dict()
    for n, w in got.items():
        short = n.split("/")[-2][-1] + ("w" if n.endswith("weights") else "b")
        ref = d["final_%s_f64" % short]
        assert np.abs(w - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), n
        assert np.abs(w - d["final_%s_f32" % short]).max() <= 2e-4 * max(1.0, np.abs(ref).max()), n


def test_placeholder_inputs_contract(dev):
    """models/dpdist_and_aue.py:23-28: (input1, input2, labels12, labels21) with static shapes [B,N,NUM_DIMS] x2, [B,N] x2,
    float32; NUM_DIMS defaults to 2 like the reference (the trainer passes 3, train_multi_gpu_pc_compare_dist.py:192)."""
    from dpdist_amd.model import get_model, placeholder_inputs, reset_default_graph
    pcA, pcB, lab12, lab21 = placeholder_inputs(4, 64, 3, device=dev)
    assert pcA.shape == pcB.shape == (4, 64, 3) and lab12.shape == lab21.shape == (4, 64)
    assert all(t.dtype == torch.float32 and t.is_cuda for t in (pcA, pcB, lab12, lab21))
    assert placeholder_inputs(2, 16, device=dev)[0].shape == (2, 16, 2)          # the reference's default NUM_DIMS
    reset_default_graph()
    pred, end_points, _ = get_model(pcA, pcB, True, bn=0, pn="3dmfv", k=5)      # the placeholders feed get_model as they are
    assert pred["pred_listAB"].shape == (4, 64, 1, 3) and end_points == {}
    with pytest.raises(NotImplementedError):
        get_model(*placeholder_inputs(2, 16, device=dev)[:2], True, bn=0, pn="3dmfv", k=5)   # 2-D inputs: not on the hot path
    reset_default_graph()


def test_embedding_set_matches_reference_rows(dev, golden_dir):
    """The third return of get_model (embedding_A/B = local_z_3d output [B, 512, 2500], utils/dpdist_util.py:911-930),
    materialised lazily by the product, against the rows the reference produced (path_fwd_s1_wide.npz: embA_rows/embB_rows)."""
    d = _g(golden_dir, "path_fwd_s1_wide.npz")
    mod = _model(dev, "wide")
    from dpdist_amd.model import get_model
    with torch.no_grad():
        _, _, emb = get_model(_cu(d["pcA"], dev), _cu(d["pcB"], dev), True, bn=0, pn="3dmfv", k=5, params=mod.params_)
    sel = d["emb_sel"]
    eA, eB = emb["embedding_A"], emb["embedding_B"]
    assert tuple(eA.shape) == (2, 512, 2500) and sorted(emb.keys()) == ["embedding_A", "embedding_B"]
    gotA = eA[sel[:, 0], sel[:, 1]].cpu().numpy()
    gotB = eB[sel[:, 0], sel[:, 1]].cpu().numpy()
    assert np.abs(gotA - d["embA_rows"]).max() <= 3e-6
    assert np.abs(gotB - d["embB_rows"]).max() <= 3e-6


def test_adam_kernel(dev):
    from dpdist_amd import ops
    from oracle import restate as R
    rng = np.random.default_rng(0)
    n = 4099
    p, g = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    P, G = _cu(np.pad(p, (0, 1)), dev)[:n], _cu(np.pad(g, (0, 1)), dev)[:n]
    Mst, V = torch.zeros_like(P), torch.zeros_like(P)
    pr, m, v = p.astype(np.float64), np.zeros(n), np.zeros(n)
    for t in (1, 2, 3):
        import math
        lr_t = 1e-3 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        ops.adam_tf(P, G, Mst, V, lr_t, gscale=0.5)
        R.adam_tf_step(pr, g.astype(np.float64) * 0.5, m, v, t, 1e-3)
    assert np.abs(P.cpu().numpy() - pr).max() <= 1e-6


def test_error_behaviour(dev):
    """Errors are loud: CPU tensors, unsupported sizes and non-contiguous inputs raise."""
    from dpdist_amd import ops
    with pytest.raises(RuntimeError):
        ops.mfv3d_fwd(torch.zeros(1, 64, 3), 8, 0.125)
    with pytest.raises(RuntimeError):
        ops.mfv3d_fwd(torch.zeros(1, 64, 3, device=dev), 11, 0.125)
    with pytest.raises(RuntimeError):
        ops.mfv3d_fwd(torch.zeros(1, 3, 64, device=dev).transpose(1, 2), 8, 0.125)
    with pytest.raises(RuntimeError):
        ops.patch_rows_fwd(torch.zeros(1, 64, 3, device=dev), torch.zeros(1, 512, 20, device=dev), 8, 4)


# ------------------------------------------------------------------------------------------------ shapes / edge cases
def _oracle_forward(pcA, pcB, W, m=8, k=5, noise=None):
    from oracle import restate as R
    n = None if noise is None else torch.tensor(noise)
    ref, _ = R.get_model(torch.tensor(pcA), torch.tensor(pcB), R.as_torch_weights(W), add_noise=n, m=m, k=k)
    return ref


def test_config1_single_pair_forward(dev):
    """BASELINE config 1: one 'chair-like' pair, B=1, forward only."""
    pcA, pcB, _ = synth.s2_modelnet_shaped(1, 64, 100)
    W = synth.make_weights("wide")
    mod = _model(dev, "wide")
    with torch.no_grad():
        ps = mod(_cu(pcA, dev), _cu(pcB, dev))
    ref = _oracle_forward(pcA, pcB, W)
    for n in ("pred_listAB", "pred_listBA"):
        assert ps[n].shape == (1, 64, 1, 3)
        _check_pred(ps[n].cpu().numpy(), ref[n].numpy(), True)


@pytest.mark.parametrize("N", [32, 100, 200])
def test_ragged_point_counts(dev, N):
    """num_point other than 64 (not a multiple of the wavefront): every kernel is shape generic."""
    rng = np.random.default_rng(N)
    pcA = rng.uniform(-0.9, 0.9, (3, N, 3)).astype(np.float32)
    pcB = rng.uniform(-1.1, 1.1, (3, N, 3)).astype(np.float32)          # some queries outside the cube
    noise = (rng.standard_normal((3, N, 3)) * 0.01).astype(np.float32)
    W = synth.make_weights("wide")
    mod = _model(dev, "wide")
    with torch.no_grad():
        ps = mod(_cu(pcA, dev), _cu(pcB, dev), add_noise=_cu(noise, dev))
    ref = _oracle_forward(pcA, pcB, W, noise=noise)
    for n in ("pred_listAB", "pred_listBA"):
        _check_pred(ps[n].cpu().numpy(), ref[n].numpy(), True)


def test_window_k3(dev):
    """K=3 window (27*20+3 = 543 inputs, KP = 544)."""
    from dpdist_amd.model import DPDistModel
    pcA, pcB = synth.s1_random_patches(4, 64, 3)
    W = synth.make_weights("wide", E_plus_D=543, mlp=(128, 128, 128))
    mod = DPDistModel(Embedding_Size=512, k=3, localSNmlp=(128, 128, 128), device=dev)
    mod.load_tf_state_dict(W)
    assert mod.params_.KP == 544
    with torch.no_grad():
        ps = mod(_cu(pcA, dev), _cu(pcB, dev))
    ref = _oracle_forward(pcA, pcB, W, k=3)
    for n in ("pred_listAB", "pred_listBA"):
        _check_pred(ps[n].cpu().numpy(), ref[n].numpy(), True)


def test_large_batch_properties(dev):
    """B=256 (Q = 32768 rows, 4x the bench size): finite, masked rows exactly zero, range [0,2], AB/BA symmetry under
    swapping the clouds."""
    pcA, pcB, _ = synth.s2_modelnet_shaped(256, 64, 7)
    pcB[:, 3] = -1.0                                     # q = -1 sits on the open face: masked out
    mod = _model(dev, "wide")
    with torch.no_grad():
        a = mod(_cu(pcA, dev), _cu(pcB, dev))
        b = mod(_cu(pcB, dev), _cu(pcA, dev))
    for n in ("pred_listAB", "pred_listBA"):
        assert torch.isfinite(a[n]).all() and a[n].min().item() >= 0.0 and a[n].max().item() <= 2.0
    assert a["pred_listAB"][:, 3].abs().max().item() == 0.0
    assert torch.equal(a["pred_listAB"], b["pred_listBA"]) and torch.equal(a["pred_listBA"], b["pred_listAB"])


def test_two_streams_are_independent(dev):
    """The library is stream ordered and re-entrant: two streams with different inputs give the single-stream results."""
    mod = _model(dev, "wide")
    A1, B1 = (_cu(x, dev) for x in synth.s1_random_patches(8, 64, 1))
    A2, B2 = (_cu(x, dev) for x in synth.s1_random_patches(8, 64, 2))
    with torch.no_grad():
        r1, r2 = mod(A1, B1), mod(A2, B2)
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        with torch.cuda.stream(s1):
            p1 = mod(A1, B1)
        with torch.cuda.stream(s2):
            p2 = mod(A2, B2)
        torch.cuda.synchronize()
    assert torch.equal(p1["pred_listAB"], r1["pred_listAB"]) and torch.equal(p2["pred_listBA"], r2["pred_listBA"])


# ------------------------------------------------------------------------------------------------ bf16 matrix-core paths
def _planes(x, np_, want_rc, want_r8):
    from dpdist_amd import lib as L
    R, C = x.shape
    rc = torch.empty(np_, R, C, device=x.device, dtype=torch.int16) if want_rc else None
    r8 = torch.empty(np_, R // 8, C, 8, device=x.device, dtype=torch.int16) if want_r8 else None
    L.check(L.load().dpd_split_planes(L.ptr(x), R, C, x.stride(0), np_, L.ptr(rc), C, R * C, L.ptr(r8), R * C,
                                      L.cur_stream()), "dpd_split_planes")
    return rc, r8


def test_split_planes_reconstruct_exactly(dev):
    """hi + mid + lo reproduces the fp32 value to 2^-27 relative (both layouts agree); NaN/inf stay in the hi plane."""
    x = torch.randn(64, 256, device=dev) * torch.logspace(-20, 20, 256, device=dev)
    x[3, 7] = float("inf")
    x[5, 9] = float("nan")
    rc, r8 = _planes(x, 3, True, True)
    back = (rc.to(torch.int32) << 16).view(torch.float32).double().sum(0)
    fin = torch.isfinite(x)
    assert ((back - x.double()).abs()[fin] <= x.double().abs()[fin] * 2.0 ** -26).all()
    assert torch.isinf(back[3, 7]) and torch.isnan(back[5, 9])
    r8_as_rc = r8.permute(0, 1, 3, 2).reshape(3, 64, 256)     # [p][r/8][c][r%8] -> [p][r][c]
    assert torch.equal(r8_as_rc, rc)


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 13])
@pytest.mark.parametrize("mode", ["NN", "NT", "TN", "TNr"])
@pytest.mark.parametrize("np_", [3, 1])
def test_gemm_planes(dev, np_, mode, tile):
    """Split-bf16 GEMM against fp64: the 3-plane / 6-term form must be at least as accurate as the exact-fp32 MFMA
    GEMM on the same operands (fp32-equivalent); the 1-plane form is a plain bf16 GEMM (2^-8 operand rounding)."""
    from dpdist_amd import lib as L, ops
    if tile == 13 and np_ == 3:
        pytest.skip("the BK = 64 tile exists for one plane only (a 3-plane stage does not fit the LDS)")
    if mode == "TNr" and tile not in (1, 2, 3, 5):
        pytest.skip("the transpose-read TN form (both operands as RC planes) exists for tiles 1, 2, 3, 5")
    M, N, K = 200, 328, 576 if tile == 13 else 544   # ragged M (clamped rows), N % 8 == 0, K a whole number of K-tiles
    g = torch.Generator().manual_seed(tile * 10 + np_)
    A = torch.randn(M, K, generator=g).to(dev)
    B = torch.randn(K, N, generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    ref = torch.relu(A.double() @ B.double() + bias.double())
    lib = L.load()
    C = torch.empty(M, N, device=dev)
    if mode == "NN":
        a, _ = _planes(A, np_, True, False); _, b = _planes(B, np_, False, True)
        args = (np_, 0, 1, M, N, K, L.ptr(a), K, M * K, L.ptr(b), N, K * N)
        c32 = ops.gemm_f32(A, B, bias=bias, epilogue=2, tile=8)
    elif mode == "NT":
        Bt = B.t().contiguous()
        a, _ = _planes(A, np_, True, False); b, _ = _planes(Bt, np_, True, False)
        args = (np_, 0, 0, M, N, K, L.ptr(a), K, M * K, L.ptr(b), K, N * K)
        c32 = ops.gemm_f32(A, Bt, transB=True, bias=bias, epilogue=2, tile=8)
    elif mode == "TN":
        At = A.t().contiguous()
        _, a = _planes(At, np_, False, True); _, b = _planes(B, np_, False, True)
        args = (np_, 1, 1, M, N, K, L.ptr(a), M, K * M, L.ptr(b), N, K * N)
        c32 = ops.gemm_f32(At, B, transA=True, bias=bias, epilogue=2, tile=8)
    else:      # "TNr": the same product with both operands as their RC planes ([K][M], [K][N]) read through LDS transpose reads
        At = A.t().contiguous()
        a, _ = _planes(At, np_, True, False); b, _ = _planes(B, np_, True, False)
        args = (np_, 2, 2, M, N, K, L.ptr(a), M, K * M, L.ptr(b), N, K * N)
        c32 = ops.gemm_f32(At, B, transA=True, bias=bias, epilogue=2, tile=8)
    L.check(lib.dpd_gemm_planes(*args, L.ptr(C), N, L.ptr(bias), None, 2, tile, None, None, 0, L.cur_stream()), "dpd_gemm_planes")
    err = (C.double() - ref).abs().max().item()
    err32 = (c32.double() - ref).abs().max().item()
    if np_ == 3:
        assert err <= max(1.5 * err32, 2e-5), (err, err32)
    else:
        assert err <= 0.02 * K ** 0.5, err          # ~2^-8 per operand, random signs


def _bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float64)


@pytest.mark.parametrize("tile", [21, 23, 24])
@pytest.mark.parametrize("mode", ["NN", "NT", "TN"])
@pytest.mark.parametrize("K", [32, 64, 96, 128, 160, 544, 576, 2528])
def test_gemm_planes_phase_staggered(dev, mode, tile, K):
    """gemm_p8_kernel (two wave groups one barrier apart, three whole K-tiles of LDS, zero chunks beyond a K that ends inside a
    K-tile; tiles 21 / 23: one bf16 plane at BK = 64, tile 24: three planes at BK = 32) against float64: for one plane the EXACT
    product of the bf16-rounded operands (fp32 accumulation is ~1e-6 relative: a chunk read before its LDS-DMA landed, a stale
    stage or a missing zero fill is an O(1) error), for three planes the fp32-equivalence bar of test_gemm_planes.
    K covers 1 .. 5 half/whole K-tiles (prologue / drain corner cases), the unrolled-by-three steady state with and without a
    remainder, and the decoder's 2528; M, N ragged (clamped rows / columns), several output tiles."""
    from dpdist_amd import lib as L, ops
    np_ = 3 if tile >= 24 else 1
    M, N = 600, 328
    g = torch.Generator().manual_seed(tile * 100 + K)
    A = torch.randn(M, K, generator=g).to(dev)
    B = torch.randn(K, N, generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    if np_ == 1:
        ref = torch.relu(_bf16_round(A) @ _bf16_round(B) + bias.double())
    else:
        ref = torch.relu(A.double() @ B.double() + bias.double())
    lib = L.load()
    if mode == "NN":
        a, _ = _planes(A, np_, True, False); _, b = _planes(B, np_, False, True)
        args = (np_, 0, 1, M, N, K, L.ptr(a), K, M * K, L.ptr(b), N, K * N)
    elif mode == "NT":
        a, _ = _planes(A, np_, True, False); b, _ = _planes(B.t().contiguous(), np_, True, False)
        args = (np_, 0, 0, M, N, K, L.ptr(a), K, M * K, L.ptr(b), K, N * K)
    else:
        _, a = _planes(A.t().contiguous(), np_, False, True); _, b = _planes(B, np_, False, True)
        args = (np_, 1, 1, M, N, K, L.ptr(a), M, K * M, L.ptr(b), N, K * N)
    outs = []
    for _ in range(3):                      # same launch three times: a race shows up as run-to-run differences as well
        C = torch.full((M, N), float("nan"), device=dev)
        L.check(lib.dpd_gemm_planes(*args, L.ptr(C), N, L.ptr(bias), None, 2, tile, None, None, 0, L.cur_stream()), "dpd_gemm_planes")
        outs.append(C)
    err = (outs[0].double() - ref).abs().max().item()
    if np_ == 1:
        assert err <= 2e-5 * max(1.0, ref.abs().max().item()) * max(1.0, K / 256), (err, ref.abs().max().item())
    else:
        err32 = (ops.gemm_f32(A, B, bias=bias, epilogue=2, tile=8).double() - ref).abs().max().item()
        assert err <= max(1.5 * err32, 2e-5), (err, err32)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("tile", [21, 24])
def test_gemm_planes_phase_staggered_full_size_is_stable(dev, tile):
    """The layer-1 shape of BASELINE config 3 (8192 x 1024 x 2528; tile 24: the B = 32 shape 4096 x 1024 x 2528 in three planes; one
    workgroup per CU) ten times under a memory-hungry side stream: identical bits every time and the right product."""
    from dpdist_amd import lib as L
    np_ = 3 if tile >= 24 else 1
    M, N, K = (8192 if np_ == 1 else 4096), 1024, 2528
    g = torch.Generator().manual_seed(7)
    A = torch.randn(M, K, generator=g).to(dev)
    B = torch.randn(K, N, generator=g).to(dev)
    ref = (_bf16_round(A[:512]) @ _bf16_round(B)) if np_ == 1 else (A[:512].double() @ B.double())
    a, _ = _planes(A, np_, True, False)
    _, b = _planes(B, np_, False, True)
    lib = L.load()
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, device=dev)
    first = None
    for it in range(10):
        with torch.cuda.stream(side):
            junk.add_(1.0)                  # 512 MB of read-modify-write next to the GEMM
        C = torch.full((M, N), float("nan"), device=dev)
        L.check(lib.dpd_gemm_planes(np_, 0, 1, M, N, K, L.ptr(a), K, M * K, L.ptr(b), N, K * N, L.ptr(C), N, None, None, 0, tile,
                                    None, None, 0, L.cur_stream()), "dpd_gemm_planes")
        if first is None:
            first = C
            assert (C[:512].double() - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()
        else:
            assert torch.equal(C, first), it
    torch.cuda.synchronize()


@pytest.mark.parametrize("case", ["s1", "boundary"])
@pytest.mark.parametrize("wk", ["xavier_tf", "wide"])
def test_forward_golden_f32x3(dev, golden_dir, case, wk):
    """The fp32-equivalent bf16-matrix-core path meets the SAME bar as the exact-fp32 path."""
    d = _g(golden_dir, "path_fwd_%s_%s.npz" % (case, wk))
    mod = _model(dev, wk)
    mod.params_.compute_dtype = "f32x3"
    with torch.no_grad():
        ps = mod(_cu(d["pcA"], dev), _cu(d["pcB"], dev))
    for n in ("pred_listAB", "pred_listBA"):
        _check_pred(ps[n].cpu().numpy(), d[n + "_f64"], wk == "wide")
        _check_pred(ps[n].cpu().numpy(), d[n + "_f32"], wk == "wide")


def test_forward_f32x3_is_as_accurate_as_f32(dev):
    """B=32 (config 2): error of both fp32 paths against the float64 oracle, side by side."""
    from oracle import restate as R
    pcA, pcB = synth.s1_random_patches(32, 64, 0)
    W = synth.make_weights("wide")
    ref, _ = R.get_model(torch.tensor(pcA, dtype=torch.float64), torch.tensor(pcB, dtype=torch.float64),
                         {n: torch.tensor(a, dtype=torch.float64) for n, a in W.items()})
    errs = {}
    for dt in ("f32", "f32x3"):
        mod = _model(dev, "wide")
        mod.params_.compute_dtype = dt
        with torch.no_grad():
            ps = mod(_cu(pcA, dev), _cu(pcB, dev))
        errs[dt] = max(np.abs(ps[n].cpu().numpy() - ref[n].numpy()).max() for n in ("pred_listAB", "pred_listBA"))
    assert errs["f32x3"] <= ABS_TOL and errs["f32x3"] <= 2.0 * errs["f32"] + 2e-6, errs


def test_forward_bf16_tolerance(dev, golden_dir):
    """BASELINE config 3 compute type: bf16 operands, fp32 accumulation.  Stated tolerance: 3e-2 absolute on the
    predicted distances of the `wide` set (outputs in [0,2]; three chained bf16 GEMMs of K=2503/1024/1024)."""
    d = _g(golden_dir, "path_fwd_s1_wide.npz")
    mod = _model(dev, "wide")
    mod.params_.compute_dtype = "bf16"
    with torch.no_grad():
        ps = mod(_cu(d["pcA"], dev), _cu(d["pcB"], dev))
    for n in ("pred_listAB", "pred_listBA"):
        err = np.abs(ps[n].cpu().numpy() - d[n + "_f64"])
        assert err.max() <= 3e-2 and err.mean() <= 5e-3, (err.max(), err.mean())


@pytest.mark.parametrize("dt", ["f32x3", "bf16"])
def test_gradients_golden_plane_paths(dev, golden_dir, dt):
    """Weight and input gradients through the plane GEMMs: f32x3 at the fp32 bars, bf16 at 3e-2 relative (norms)."""
    from dpdist_amd import model as M
    d = _g(golden_dir, "path_bwd_s2_wide.npz")
    mod = _model(dev, "wide")
    mod.params_.compute_dtype = dt
    pcA = _cu(d["pcA"], dev).requires_grad_(True)
    pcB = _cu(d["pcB"], dev).requires_grad_(True)
    noise = _cu(d["noise"], dev).requires_grad_(True)
    M.reset_default_graph()
    ps = mod(pcA, pcB, add_noise=noise)
    _, lp = M.get_loss(ps, {}, _cu(d["labels"], dev))
    ls = M.get_collection("loss_samples")[0]
    rel = 2e-4 if dt == "f32x3" else 3e-2
    assert abs(ls.item() - float(d["loss_samples_f64"])) <= (2e-5 if dt == "f32x3" else 5e-3)
    (gflat,) = torch.autograd.grad(ls, [mod.params_.flat], retain_graph=True)
    gsd = mod.params_.tf_state_dict(gflat)
    for n, g in gsd.items():
        short = n.split("/")[-2][-1] + ("w" if n.endswith("weights") else "b")
        g2 = g.reshape(-1, g.shape[-1]) if g.ndim == 4 else g
        nrm = float(d["g%s_norm_f64" % short])
        assert abs(np.sqrt((g2.astype(np.float64) ** 2).sum()) - nrm) <= rel * max(1.0, nrm), n
        if g.ndim == 4:
            assert np.abs(g2[:16, :16] - d["g%s_corner_f64" % short]).max() <= rel * max(1.0, nrm), n
    gA, gB, gN = torch.autograd.grad(lp, [pcA, pcB, noise])
    for g, n in ((gA, "d_pcA"), (gB, "d_pcB"), (gN, "d_noise")):
        ref = d[n + "_f64"]
        if dt == "f32x3":
            bar = max(4.0 * np.abs(d[n + "_f32"] - ref).max(), 2e-4 * max(1.0, np.abs(ref).max()))
        else:
            bar = 5e-2 * max(1.0, np.abs(ref).max())
        assert np.abs(g.cpu().numpy() - ref).max() <= bar, (n, np.abs(g.cpu().numpy() - ref).max(), bar)


@pytest.mark.parametrize("dt", ["f32x3", "bf16"])
def test_trainer_plane_paths(dev, dt):
    """Training steps on the bf16 matrix cores: f32x3 tracks the fp32 trainer to fp32 round-off; bf16 (mixed precision:
    fp32 master weights, fp32 Adam) tracks it to bf16 round-off and the loss decreases."""
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    B = 8
    pcA, pcB, lab = synth.s2_modelnet_shaped(B, 64, 100)
    W0 = synth.make_weights("wide")
    out = {}
    for name in ("f32", dt):
        P = DPDistParams(device=dev, compute_dtype=name)
        P.load_tf_state_dict(W0)
        tr = DPDistTrainer(P, B, base_lr=1e-3, distributed=False)
        losses = [tr.step(_cu(pcA, dev), _cu(pcB, dev), _cu(lab, dev)).cpu().numpy().copy() for _ in range(4)]
        out[name] = (np.array(losses), P.flat.detach().cpu().numpy().copy())
    l32, w32 = out["f32"]
    lx, wx = out[dt]
    if dt == "f32x3":
        assert np.abs(lx - l32).max() <= 5e-5
        # Adam's first steps move every weight by ~lr whatever the gradient size: tiny gradients may flip sign in round-off
        assert np.abs(wx - w32).max() <= 2.5e-3 * 4 and np.abs(wx - w32).mean() <= 1e-5
    else:
        assert np.abs(lx - l32).max() <= 2e-2
    assert lx[-1, 0] < lx[0, 0]


def test_bf16_training_tracks_fp32_over_50_steps(dev):
    """B=64, bf16 MFMA with fp32 accumulate; the loss curve must track fp32 within 1 %
    over 50 steps.  NOTE: this compares the HIP bf16 path with the HIP fp32 path, i.e. it is a self-comparison -- it is only
    meaningful because the fp32 trainer is pinned to the oracle / the reference's optimizer fixture separately
    (test_trainer_steps_vs_oracle, test_trainer_steps_vs_reference_optimizer_fixture) and because
    test_bf16_step_vs_oracle_b64 below checks the bf16 path against the oracle directly at this batch size."""
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    B = 64
    pcA, pcB, lab = synth.s2_modelnet_shaped(B, 64, 100)
    W0 = synth.make_weights("wide")
    curves = {}
    for name in ("f32", "bf16"):
        P = DPDistParams(device=dev, compute_dtype=name)
        P.load_tf_state_dict(W0)
        tr = DPDistTrainer(P, B, base_lr=1e-4, distributed=False)
        curves[name] = np.array([tr.step(_cu(pcA, dev), _cu(pcB, dev), _cu(lab, dev))[0].item() for _ in range(50)])
    rel = np.abs(curves["bf16"] - curves["f32"]) / curves["f32"]
    assert rel.max() <= 0.01, rel.max()
    assert curves["bf16"][-1] < 0.9 * curves["bf16"][0]


def test_bf16_step_vs_oracle_b64(dev):
    """BASELINE config 3 against the ORACLE (not against our own fp32 path): forward, losses and the first optimizer step of
    the bf16 compute type at B = 64.  Tolerances are bf16's: 2^-8 operand rounding through three 1024-wide layers (outputs in
    [0, 2]: 5e-2 max / 5e-3 mean abs over the 8192 x 3 outputs; mean losses: 1 % relative)."""
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    from oracle import restate as R
    B = 64
    pcA, pcB, lab = synth.s2_modelnet_shaped(B, 64, 100)
    W0 = synth.make_weights("wide")
    P = DPDistParams(device=dev, compute_dtype="bf16")
    P.load_tf_state_dict(W0)
    tr = DPDistTrainer(P, B, base_lr=1e-4, distributed=False)
    loss = tr.step(_cu(pcA, dev), _cu(pcB, dev), _cu(lab, dev)).cpu().numpy()
    pred = tr.pred.cpu().numpy().reshape(2, B, 64, 3)
    torch.set_num_threads(8)
    Wt = R.as_torch_weights(W0, torch.float32, requires_grad=True)
    ref, _ = R.get_model(torch.tensor(pcA), torch.tensor(pcB), Wt)
    ls, lp = R.get_loss(ref, torch.tensor(lab))
    for got, name in ((pred[0], "pred_listAB"), (pred[1], "pred_listBA")):
        err = np.abs(got - ref[name].detach().numpy()[:, :, 0])
        assert err.max() <= 5e-2 and err.mean() <= 5e-3, (name, err.max(), err.mean())   # 8192 outputs in [0, 2]: max 2.5 %, mean 0.25 % of range
    assert abs(loss[0] - ls.item()) <= 0.01 * ls.item() and abs(loss[1] - lp.item()) <= 0.01 * lp.item()
    # the gradient that was just applied: direction and size against the oracle's autodiff (cosine, norm ratio per variable)
    names = sorted(Wt)
    gs = torch.autograd.grad(ls, [Wt[n] for n in names])
    got = P.tf_state_dict(tr.grad)
    for n, g in zip(names, gs):
        a, b = got[n].astype(np.float64).ravel(), g.numpy().astype(np.float64).ravel()
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
        assert cos >= 0.995, (n, cos)
        assert abs(np.linalg.norm(a) / (np.linalg.norm(b) + 1e-30) - 1.0) <= 0.02, n


@pytest.mark.parametrize("tile", [1, 2, 3, 5, 21, 23, 24])
@pytest.mark.parametrize("np_", [3, 1])
def test_gemm_planes_fused_outputs(dev, np_, tile):
    """The LDS-staged epilogue writes the result as operand planes: bit-identical to splitting the fp32 result."""
    from dpdist_amd import lib as L
    if tile >= 20 and (np_ == 3) != (tile >= 24):
        pytest.skip("phase-staggered tiles 21 / 23 take one plane, 24 three")
    M, N, K, R8 = 320, 264, 96, 192
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g).to(dev)
    B = torch.randn(K, N, generator=g).to(dev)
    a, _ = _planes(A, np_, True, False)
    _, b = _planes(B, np_, False, True)
    C = torch.empty(M, N, device=dev)
    rc = torch.zeros(np_, M, N, device=dev, dtype=torch.int16)
    r8 = torch.zeros(np_, R8 // 8, N, 8, device=dev, dtype=torch.int16)
    L.check(L.load().dpd_gemm_planes(np_, 0, 1, M, N, K, L.ptr(a), K, M * K, L.ptr(b), N, K * N, L.ptr(C), N, None, None, 0,
                                     tile, L.ptr(rc), L.ptr(r8), R8, L.cur_stream()), "dpd_gemm_planes")
    want_rc, _ = _planes(C, np_, True, False)
    _, want_r8 = _planes(C[:R8].contiguous(), np_, False, True)
    assert torch.equal(rc, want_rc)
    assert torch.equal(r8, want_r8)
    # planes only (C == NULL)
    rc2 = torch.zeros_like(rc)
    L.check(L.load().dpd_gemm_planes(np_, 0, 1, M, N, K, L.ptr(a), K, M * K, L.ptr(b), N, K * N, None, N, None, None, 0,
                                     tile, L.ptr(rc2), None, 0, L.cur_stream()), "dpd_gemm_planes")
    assert torch.equal(rc2, want_rc)


@pytest.mark.parametrize("np_", [3, 1])
def test_patch_rows_planes_match_split(dev, np_):
    """The window gather writes X directly as operand planes: bit-identical to converting the fp32 rows."""
    from dpdist_amd import lib as L, ops
    B, N, m, k = 4, 64, 8, 5
    pcA, pcB = synth.s1_random_patches(B, N, 3)
    pts, q = ops.stack_clouds(_cu(pcA, dev), _cu(pcB, dev))
    fv = ops.mfv3d_fwd(pts, m, 0.125)
    X, mask, vox = ops.patch_rows_fwd(q, fv, m, k)
    Q, KP, Qb = X.shape[0], X.shape[1], X.shape[0] // 2
    lib = L.load()
    dt = 1 if np_ == 3 else 2
    nbytes = lib.dpd_planes_bytes(Q, Qb, KP, 1024, dt, 0)
    mem = torch.zeros(nbytes, device=dev, dtype=torch.uint8)
    pl = L.Planes()
    L.check(lib.dpd_planes_carve(L.ptr(mem), nbytes, Q, Qb, KP, 1024, dt, 0, pl), "carve")
    X2, mask2, vox2 = torch.zeros_like(X), torch.zeros_like(mask), torch.zeros_like(vox)
    L.check(lib.dpd_patch_rows_fwd(L.ptr(q), L.ptr(fv), 2 * B, N, m, k, KP, L.ptr(X2), L.ptr(mask2), L.ptr(vox2), pl,
                                   L.cur_stream()), "dpd_patch_rows_fwd")
    assert torch.equal(X2, X) and torch.equal(mask2, mask) and torch.equal(vox2, vox)
    want_rc, _ = _planes(X, np_, True, False)
    _, want_r8 = _planes(X[:Qb].contiguous(), np_, False, True)
    off = pl.X_rc - mem.data_ptr()
    got_rc = mem[off:off + np_ * Q * KP * 2].view(torch.int16).view(np_, Q, KP)
    off = pl.X_r8 - mem.data_ptr()
    got_r8 = mem[off:off + np_ * Qb * KP * 2].view(torch.int16).view(np_, Qb // 8, KP, 8)
    assert torch.equal(got_rc, want_rc)
    assert torch.equal(got_r8, want_r8)


def test_prefetch_pipeline_is_bitwise_equivalent(dev):
    """Running the next batch's encoder + gather on the side stream must not change a single bit of the weights
    (except through the fp32 atomics of db1/db2, which are excluded by comparing the GEMM-produced tensors)."""
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    B = 8
    batches = [tuple(_cu(x, dev) for x in synth.s2_modelnet_shaped(B, 64, 100 + i)) for i in range(4)]
    W0 = synth.make_weights("wide")
    outs = []
    for use_prefetch in (False, True):
        P = DPDistParams(device=dev)
        P.load_tf_state_dict(W0)
        tr = DPDistTrainer(P, B, base_lr=1e-3, distributed=False)
        losses = []
        for i, (a, b, l) in enumerate(batches):
            nxt = batches[i + 1][:2] + (None,) if (use_prefetch and i + 1 < len(batches)) else None
            losses.append(tr.step(a, b, l, prefetch=nxt).clone())
            if nxt is not None:     # the side-stream front end of the next batch survives apply_gradients (round-1 bug: it was wiped)
                assert tr._pref_key is not None
        torch.cuda.synchronize()
        # exactly ONE front end per batch either way; with prefetch every step after the first takes the hit path
        assert tr.front_launches == len(batches)
        assert tr.prefetch_hits == (len(batches) - 1 if use_prefetch else 0)
        outs.append((torch.stack(losses), P.view("W1p").detach().clone(), P.view("W4").detach().clone()))
    # first step: identical inputs and weights -> identical loss bits; later steps differ only by atomics-order round-off
    assert torch.equal(outs[0][0][0], outs[1][0][0])
    assert (outs[0][0] - outs[1][0]).abs().max().item() <= 1e-6
    assert (outs[0][1] - outs[1][1]).abs().max().item() <= 2.1e-3      # Adam: a sign flip of a ~0 gradient moves a weight by 2 lr
    assert (outs[0][1] - outs[1][1]).abs().mean().item() <= 1e-6
    # a mismatching batch after a prefetch falls back to recomputing the front end
    P = DPDistParams(device=dev); P.load_tf_state_dict(W0)
    tr = DPDistTrainer(P, B, base_lr=1e-3, distributed=False)
    a, b, l = batches[0]
    tr.step(a, b, l, prefetch=batches[1][:2] + (None,))
    l2 = tr.evaluate(*batches[2])[0].clone()
    P2 = DPDistParams(device=dev); P2.load_tf_state_dict(P.tf_state_dict())
    tr2 = DPDistTrainer(P2, B, distributed=False)
    assert (tr2.evaluate(*batches[2])[0] - l2).abs().max().item() <= 1e-7


@pytest.mark.parametrize("mode", ["allreduce", "zero1"])
@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_two_ranks_rccl_equals_full_batch(dev, dt, mode, tmp_path):
    """Two REAL ranks over RCCL (one process per GPU, started the way `python bench.py --gpus 2` starts them): the averaged
    gradient equals the full-batch gradient, and the replicas stay bit-identical through two optimizer steps.  Skipped on a
    one-GPU box (the driver's GPU box has one)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import sys
    import bench          # tests/conftest.py puts the repo root on sys.path
    out = tmp_path / "r0.txt"
    rc = bench.spawn_ranks(2, [sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_rank.py"), str(out), dt, mode])
    assert rc == 0
    world, err, scale, same = out.read_text().split()
    assert int(world) == 2 and int(same) == 1
    # shard means averaged vs one mean over the full batch: fp32 summation order only (bf16: operand rounding is identical,
    # the split into two M-halves changes only the fp32 accumulation order of dW)
    assert float(err) <= 2e-5 * max(1.0, float(scale)), (err, scale)


@pytest.mark.parametrize("backend", ["rccl", "torch"])
@pytest.mark.parametrize("buckets", ["2", "3"])
@pytest.mark.parametrize("dt", ["f32", "f32x3"])
def test_data_parallel_schedule_matches_plain_backward(dev, dt, buckets, backend, monkeypatch):
    """The interleaved data-parallel backward (phased data chain, dW3 -> dW2 -> dW1; layers 2-4 as one collective after dW2 or
    one collective per bucket; exercised here with a single-rank process group) produces the same gradients as the plain
    schedule.  backend "rccl": ddp.DirectRcclReducer (librccl through ctypes: early buckets on a side stream behind events without
    the system fence, the last bucket on the compute stream itself); "torch": ddp.BucketReducer over torch.distributed."""
    monkeypatch.setenv("DPD_DP_BACKEND", backend)
    import torch.distributed as dist
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    B = 8
    pcA, pcB, lab = (_cu(x, dev) for x in synth.s2_modelnet_shaped(B, 64, 100))
    W0 = synth.make_weights("wide")
    grads = {}
    own_pg = not dist.is_initialized()
    if own_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29631")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        for mode in ("plain", "dp"):
            os.environ["DPD_FORCE_DIST"] = "1" if mode == "dp" else "0"
            P = DPDistParams(device=dev, compute_dtype=dt)
            P.load_tf_state_dict(W0)
            tr = DPDistTrainer(P, B, base_lr=1e-3, distributed=(mode == "dp"), options={"dp_buckets": int(buckets)})
            assert (tr.reducer is not None and tr.reducer.active) == (mode == "dp")
            if mode == "dp":
                assert type(tr.reducer).__name__ == ("DirectRcclReducer" if backend == "rccl" else "BucketReducer")
            tr.step(pcA, pcB, lab)
            tr.step(pcA, pcB, lab)                   # a second step: the per-step state of the reducer resets in wait()
            torch.cuda.synchronize()
            grads[mode] = (tr.grad.clone(), tr.loss.clone())
    finally:
        os.environ.pop("DPD_FORCE_DIST", None)
        if own_pg:
            dist.destroy_process_group()
    g0, g1 = grads["plain"][0], grads["dp"][0]
    assert torch.equal(grads["plain"][1], grads["dp"][1])
    P = DPDistParams(device=dev, init=None)
    for n, (off, cnt, _) in P._segments.items():
        a, b = g0[off:off + cnt], g1[off:off + cnt]
        if n in ("b1", "b2"):      # fp32 atomics in the dH epilogues: order-dependent round-off
            assert (a - b).abs().max().item() <= 1e-6 * max(1.0, a.abs().max().item()), n
        elif dt == "f32":          # dW2/dW3 come from different launches (grouped vs single): same tiles, same k order
            assert torch.equal(a, b), n
        else:                      # plane path: grouped 64x128 tiles vs single 64x64 tiles -> different accumulation order
            assert (a - b).abs().max().item() <= 2e-6 * max(1.0, a.abs().max().item()), n


def _single_rank_group(dev, port):
    import torch.distributed as dist
    own = not dist.is_initialized()
    if own:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(port))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    return own


@pytest.mark.parametrize("backend,mode", [("rccl", "allreduce"), ("rccl", "zero1"), ("torch", "allreduce"), ("torch", "zero1"), ("torch", "rs_ag")])
def test_reducer_startup_crosscheck_and_rank_count(dev, backend, mode, monkeypatch):
    """ddp.make_reducer: every reducer form reduces a known integer pattern (exact in any summation order) with the trainer's call
    sequence and must return the closed-form sum BIT FOR BIT, next to a plain torch.distributed all-reduce of the same pattern;
    the direct reducer reports its rank count from ncclCommCount; the gradient buffer is left untouched.  (Single-rank group on a
    one-GPU box; the same code runs first thing in every N > 1 launch.)"""
    import torch.distributed as dist
    from dpdist_amd import ddp
    monkeypatch.setenv("DPD_DP_BACKEND", backend)
    monkeypatch.setenv("DPD_DP_MODE", mode)
    own = _single_rank_group(dev, 29641)
    try:
        flat = torch.randn(5000, device=dev)
        keep = flat.clone()
        red = ddp.make_reducer(flat, [0, 2000, 3004, 5000], force=True)
        assert red.active and red.backend == (backend if mode != "rs_ag" else "torch") and red.mode == mode
        assert red.crosscheck["ok"] and red.crosscheck["reducer_bitwise"] and red.crosscheck["torch_all_reduce_bitwise"]
        assert red.crosscheck["elements"] == 5000 and red.nranks == 1 and red.wire_bytes_per_step == 0
        assert torch.equal(flat, keep)
        red.measure = True                       # exposed-communication probe: event pairs around the compute stream's waits
        for _ in range(2):
            red.reduce_async(1, upto=2)
            red.reduce_async(0)
            red.wait()
        n, ms = red.exposure.collect_ms()
        assert n >= 2 and 0.0 <= ms < 50.0
        red.close()
    finally:
        if own:
            dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["rccl", "torch"])
@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_zero1_sharded_optimizer_step_is_bitwise_replicated_adam(dev, dt, backend, monkeypatch):
    """DPD_DP_MODE=zero1 through the trainer: reduce-scatter, `dpd_adam_tf` on the owned ranges, all-gather of the fp32 parameters,
    operand planes / transposed copies re-derived lazily -- after three steps parameters and Adam slots equal those of the
    replicated optimizer behind an all-reduce bit for bit (single-rank group here; two ranks: tests/rccl_rank.py)."""
    import torch.distributed as dist
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    monkeypatch.setenv("DPD_DP_BACKEND", backend)
    monkeypatch.setenv("DPD_FORCE_DIST", "1")
    B = 8
    pcA, pcB, lab = (_cu(x, dev) for x in synth.s2_modelnet_shaped(B, 64, 100))
    W0 = synth.make_weights("wide")
    own = _single_rank_group(dev, 29643)
    res = {}
    try:
        for mode in ("allreduce", "zero1"):
            monkeypatch.setenv("DPD_DP_MODE", mode)
            P = DPDistParams(device=dev, compute_dtype=dt)
            P.load_tf_state_dict(W0)
            tr = DPDistTrainer(P, B, base_lr=1e-3, distributed=True)
            assert tr.reducer.mode == mode and tr.reducer.backend == backend
            losses = [tr.step(pcA, pcB, lab).clone() for _ in range(3)]
            tr.gather_optimizer_state()
            torch.cuda.synchronize()
            res[mode] = (P.flat.detach().clone(), tr.m_state.clone(), tr.v_state.clone(), torch.stack(losses))
            tr.close()
    finally:
        if own:
            dist.destroy_process_group()
    if dt == "f32":
        for a, b in zip(res["allreduce"], res["zero1"]):
            assert torch.equal(a, b)
    else:
        # plane compute types: db1 / db2 are fp32 atomics (order-dependent round-off from run to run), so two runs of the SAME mode
        # differ in the last bits too; first step bitwise (same inputs, same weights), afterwards Adam-sized differences only
        la, lz = res["allreduce"][3], res["zero1"][3]
        assert torch.equal(la[0], lz[0]) and (la - lz).abs().max().item() <= 1e-5
        assert (res["allreduce"][0] - res["zero1"][0]).abs().max().item() <= 2.1e-3      # a sign flip of a ~0 gradient moves a weight by 2 lr
        assert (res["allreduce"][0] - res["zero1"][0]).abs().mean().item() <= 1e-6


@pytest.mark.parametrize("backend", ["rccl", "torch"])
@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_optimizer_on_the_collective_stream_is_the_joined_step(dev, dt, backend, monkeypatch):
    """adam_on_side: in a data-parallel step Adam runs on the stream the collectives ran on and the compute stream is joined only
    where the next step first reads the weights (after its encoder + window gather).  Four steps on alternating batches, with a
    memory-hungry kernel train on the compute stream in between, give the parameters, Adam slots and losses of the joined form
    (f32: bit for bit); reading the weights through the trainer (evaluate, tf_global_variables) joins by itself."""
    import torch.distributed as dist
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    monkeypatch.setenv("DPD_DP_BACKEND", backend)
    monkeypatch.setenv("DPD_FORCE_DIST", "1")
    B = 8
    batches = [tuple(_cu(x, dev) for x in synth.s2_modelnet_shaped(B, 64, 100 + i)) for i in range(2)]
    W0 = synth.make_weights("wide")
    own = _single_rank_group(dev, 29645)
    junk = torch.empty(64 << 20, device=dev)
    res = {}
    try:
        for side in ("0", "1"):
            P = DPDistParams(device=dev, compute_dtype=dt)
            P.load_tf_state_dict(W0)
            tr = DPDistTrainer(P, B, base_lr=1e-3, distributed=True, adam_on_side=(side == "1"))
            losses = []
            for i in range(4):
                losses.append(tr.step(*batches[i % 2]).clone())
                junk.add_(1.0)                       # work on the compute stream while the optimizer may still run on the side stream
            assert tr._opt_pending == (side == "1" and backend == "rccl")      # (torch.distributed's streams: measured slower, not used)
            ev = tr.evaluate(*batches[0])[0].clone()  # joins
            assert not tr._opt_pending
            sd = tr.tf_global_variables()
            torch.cuda.synchronize()
            res[side] = (P.flat.detach().clone(), tr.m_state.clone(), tr.v_state.clone(), torch.stack(losses), ev, sd["batch"])
            tr.close()
    finally:
        if own:
            dist.destroy_process_group()
    a, b = res["0"], res["1"]
    assert a[5] == b[5] == 4
    if dt == "f32":
        for x, y in zip(a[:5], b[:5]):
            assert torch.equal(x, y)
    else:       # plane types: db1 / db2 are fp32 atomics (run-to-run round-off)
        assert torch.equal(a[3][0], b[3][0]) and (a[3] - b[3]).abs().max().item() <= 1e-5
        assert (a[0] - b[0]).abs().max().item() <= 2.1e-3 and (a[0] - b[0]).abs().mean().item() <= 1e-6


@pytest.mark.parametrize("backend", ["rccl", "torch"])
def test_grouped_data_parallel_schedule_matches_the_early_one(dev, backend, monkeypatch):
    """DPD_DP_SCHEDULE=grouped (bf16): data chain, ONE grouped dW1 + dW2 + dW3 launch, ONE all-reduce of the whole gradient -- against
    the default early schedule (three weight-gradient launches, two collectives): the same reduced gradients up to fp32 summation
    order / the atomics' round-off of the bias gradients, the same three training steps."""
    import torch.distributed as dist
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    monkeypatch.setenv("DPD_DP_BACKEND", backend)
    monkeypatch.setenv("DPD_FORCE_DIST", "1")
    B = 32
    pcA, pcB, lab = (_cu(x, dev) for x in synth.s2_modelnet_shaped(B, 64, 100))
    own = _single_rank_group(dev, 29647)
    res = {}
    try:
        for sched in ("early", "grouped"):
            monkeypatch.setenv("DPD_DP_SCHEDULE", sched)
            P = DPDistParams(device=dev, compute_dtype="bf16")
            P.load_tf_state_dict(synth.make_weights("wide"))
            tr = DPDistTrainer(P, B, base_lr=1e-3, distributed=True)
            tr._take_front(pcA, pcB, None)
            tr._decode(skip_out=True)
            tr.backward(lab.reshape(-1))
            tr.reducer.wait()
            torch.cuda.synchronize()
            g = tr.grad.clone()
            assert len(tr.reducer._calls) == (2 if sched == "early" else 1)
            losses = torch.stack([tr.step(pcA, pcB, lab).clone() for _ in range(3)])
            tr.join_optimizer()
            torch.cuda.synchronize()
            res[sched] = (g, losses)
            tr.close()
    finally:
        if own:
            dist.destroy_process_group()
    a, b = res["early"][0], res["grouped"][0]
    assert (a - b).abs().max().item() <= 4e-6 * a.abs().max().item()
    assert (res["early"][1] - res["grouped"][1]).abs().max().item() <= 2e-3


def test_bench_watchdog_falls_back_on_the_gpu(dev):
    """bench.py's N > 1 skeleton on one GPU (DPD_FORCE_DIST=1): the worker of attempt 1 stops beating in the timed region (injected);
    the supervisor stops it by PID and reruns with DPD_DP_BACKEND=torch; the JSON line carries dp_backend / fallback / the
    watchdog's record, the start-up cross-check, ncclCommCount's rank count and the exposed-communication probe.  And
    `--gpus 2` on a one-GPU box still ends quickly with code 2."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DPD_FORCE_DIST="1", DPD_WD_INJECT_HANG="timed", DPD_WD_LIMITS="timed=6", MASTER_PORT="29651")
    for k in ("DPD_BENCH_CHILD", "DPD_DP_BACKEND", "DPD_DP_MODE", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-other-dtypes"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["fallback"] is True and rec["dp_backend"] == "torch" and rec["value"] > 0
    assert rec["dp"]["crosscheck"]["ok"] and rec["dp"]["nranks"] == 1 and "exposed_comm_us_per_step" in rec["dp"]
    assert "no heartbeat" in rec["dp"]["watchdog_history"][0]["failure"]
    env.pop("DPD_WD_INJECT_HANG")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)       # healthy run: attempt 1, direct RCCL
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["fallback"] is False and rec["dp_backend"] == "rccl" and rec["dp"]["nranks_source"] == "ncclCommCount"
    if torch.cuda.device_count() < 2:
        t0 = time.time()
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 2 and time.time() - t0 < 120


@pytest.mark.parametrize("wire,mode", [("bf16", "allreduce"), ("f32", "rs_ag"), ("bf16", "rs_ag")])
def test_reducer_variants_on_rccl(dev, wire, mode):
    """BucketReducer's bf16 wire and reduce-scatter + all-gather form through RCCL (single-rank group: the values must come back
    unchanged up to the wire rounding, and the stream ordering with the backward must hold)."""
    import torch.distributed as dist
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    B = 8
    pcA, pcB, lab = (_cu(x, dev) for x in synth.s2_modelnet_shaped(B, 64, 100))
    W0 = synth.make_weights("wide")
    own_pg = not dist.is_initialized()
    if own_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29633")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        grads = {}
        for variant in ("ref", "var"):
            os.environ["DPD_FORCE_DIST"] = "1"
            os.environ["DPD_DP_WIRE"], os.environ["DPD_DP_MODE"] = (wire, mode) if variant == "var" else ("f32", "allreduce")
            P = DPDistParams(device=dev)
            P.load_tf_state_dict(W0)
            tr = DPDistTrainer(P, B, base_lr=1e-3, distributed=True)
            assert tr.reducer.wire == os.environ["DPD_DP_WIRE"] and tr.reducer.mode == os.environ["DPD_DP_MODE"]
            tr._load_batch(pcA, pcB, None)
            tr.forward()
            tr.backward(lab.reshape(-1))
            tr.reducer.wait()
            torch.cuda.synchronize()
            grads[variant] = tr.grad.clone()
    finally:
        for k in ("DPD_FORCE_DIST", "DPD_DP_WIRE", "DPD_DP_MODE"):
            os.environ.pop(k, None)
        if own_pg:
            dist.destroy_process_group()
    a, b = grads["ref"], grads["var"]
    if wire == "f32":
        assert (a - b).abs().max().item() <= 1e-6 * max(1.0, a.abs().max().item())      # db1/db2 atomics order only
    else:
        assert ((a - b).abs() <= 2.0 ** -8 * a.abs() + 1e-12).all()                     # one bf16 rounding per value


def test_fused_as_loss_node_equals_module_contract(dev, golden_dir):
    """DPDistLoss (one fused autograd node) == get_model + get_loss['loss_pred'] of the module contract, value and
    input gradients, and the fused node matches the oracle's float64 autograd (zero add_noise, as the consumers feed it)."""
    from dpdist_amd import model as M
    d = _g(golden_dir, "path_bwd_s2_wide.npz")
    mod = _model(dev, "wide")
    a1 = _cu(d["pcA"], dev).requires_grad_(True)
    b1 = _cu(d["pcB"], dev).requires_grad_(True)
    loss1 = M.DPDistLoss(mod)(a1, b1)
    gA1, gB1 = torch.autograd.grad(loss1 * 2.0, [a1, b1])
    a2 = _cu(d["pcA"], dev).requires_grad_(True)
    b2 = _cu(d["pcB"], dev).requires_grad_(True)
    M.reset_default_graph()
    ps = mod(a2, b2)
    _, lp = M.get_loss(ps, {}, _cu(d["labels"], dev))
    gA2, gB2 = torch.autograd.grad(lp * 2.0, [a2, b2])
    assert abs(loss1.item() - lp.item()) <= 1e-7
    assert (gA1 - gA2).abs().max().item() <= 1e-6 * max(1.0, gA2.abs().max().item())
    assert (gB1 - gB2).abs().max().item() <= 1e-6 * max(1.0, gB2.abs().max().item())
    # ... and the fused node against the ORACLE's float64 autograd on the same inputs (the golden d_pcA / d_pcB of this fixture were
    # generated with a non-zero add_noise, which the as-loss consumers never feed: iterative_PCRNet_ours.py:422-431 feeds zeros)
    from oracle import restate as R

    def oracle(dtype):
        W = R.as_torch_weights(synth.make_weights("wide"), dtype)
        a3 = torch.tensor(d["pcA"], dtype=dtype, requires_grad=True)
        b3 = torch.tensor(d["pcB"], dtype=dtype, requires_grad=True)
        ps3, _ = R.get_model(a3, b3, W)
        _, lp3 = R.get_loss(ps3, torch.tensor(d["labels"], dtype=dtype))
        gs = torch.autograd.grad(lp3 * 2.0, [a3, b3])
        return lp3.item(), [x.double().numpy() for x in gs]

    l64, g64 = oracle(torch.float64)
    _, g32 = oracle(torch.float32)
    assert abs(loss1.item() - l64) <= 2e-5
    for g, ref, r32 in zip((gA1, gB1), g64, g32):
        # same bar as test_losses_and_input_gradients_golden: 4x the distance of the oracle's own float32 run, or 2e-4 of the scale
        bar = max(4.0 * np.abs(r32 - ref).max(), 2e-4 * max(1.0, np.abs(ref).max()))
        assert np.abs(g.cpu().numpy() - ref).max() <= bar, (np.abs(g.cpu().numpy() - ref).max(), bar)


@pytest.mark.parametrize("dt", ["f32x3", "bf16"])
def test_as_loss_node_on_persistent_planes(dev, dt, golden_dir, monkeypatch):
    """The as-loss node of the plane compute types (round 4): rows / h1 / h2 / g3 / g2 / g1 as bf16 RC planes written by their
    producers, the frozen weights' planes cached on the parameter object -- against the round-3 form that converted both operands of
    every GEMM (asloss.PLANES = False): same loss and input gradients (the planes hold the same bits); f32x3 also against the oracle's
    float64 autograd at the bars of the exact type; two evaluations may be alive before either backward runs; a change of the
    weights re-derives the cached weight planes."""
    from dpdist_amd import model as M
    from oracle import restate as R
    d = _g(golden_dir, "path_bwd_s2_wide.npz")
    B = 16
    pcA, pcB, _ = synth.s2_modelnet_shaped(B, 64, 100)
    res = {}
    monkeypatch.setattr("dpdist_amd.asloss.ENGINE", False)          # the plane NODE (model._AsLossFn on ops.AsLossPlanes); the engine has its own tests
    for planes in ("0", "1"):
        monkeypatch.setattr("dpdist_amd.asloss.PLANES", planes == "1")
        mod = _model(dev, "wide")
        mod.params_.compute_dtype = dt
        fn = M.DPDistLoss(mod)
        a1, b1 = _cu(pcA, dev).requires_grad_(True), _cu(pcB, dev).requires_grad_(True)
        a2, b2 = _cu(pcB, dev).requires_grad_(True), _cu(pcA, dev).requires_grad_(True)
        l1 = fn(a1, b1)
        l2 = fn(a2, b2)                                   # a second evaluation before the first backward
        g1 = torch.autograd.grad(l1 * 2.0, [a1, b1])
        g2 = torch.autograd.grad(l2, [a2, b2])
        res[planes] = (l1.detach().clone(), l2.detach().clone(), g1, g2)
        if planes == "1":
            assert getattr(mod.params_, "_wplanes", None) is not None
            key0 = mod.params_._wplanes[0]
            with torch.no_grad():
                mod.params_.flat.mul_(1.0)                # bumps the version: the cached weight planes must be re-derived
            fn(a1, b1)
            assert mod.params_._wplanes[0] != key0
    tol = 1e-6 if dt == "f32x3" else 1e-5
    for x, y in ((res["0"][0], res["1"][0]), (res["0"][1], res["1"][1])):
        assert abs(x.item() - y.item()) <= tol
    for ga, gb in zip(res["0"][2] + res["0"][3], res["1"][2] + res["1"][3]):
        assert (ga - gb).abs().max().item() <= tol * max(1.0, ga.abs().max().item())
    # against the oracle (float64 autograd of the same loss)
    W = R.as_torch_weights(synth.make_weights("wide"), torch.float64)
    a3 = torch.tensor(pcA, dtype=torch.float64, requires_grad=True)
    b3 = torch.tensor(pcB, dtype=torch.float64, requires_grad=True)
    ps3, _ = R.get_model(a3, b3, W)
    _, lp3 = R.get_loss(ps3, torch.ones(B, 64, dtype=torch.float64))
    g64 = torch.autograd.grad(lp3 * 2.0, [a3, b3])
    assert abs(res["1"][0].item() - lp3.item()) <= (2e-5 if dt == "f32x3" else 3e-3)
    for g, ref in zip(res["1"][2], g64):
        scale = max(1.0, ref.abs().max().item())
        err = (g.double().cpu() - ref).abs().max().item()
        assert err <= (2e-3 if dt == "f32x3" else 5e-2) * scale, (err, scale)       # piecewise-smooth loss: gate flips move single entries


@pytest.mark.parametrize("B,N,H", [(16, 64, 1024), (3, 4, 192)])
def test_out_asloss_equals_the_three_kernel_chain(dev, B, N, H):
    """dpd_decoder_out_asloss (output layer + loss_pred + output-layer backward of d loss_pred / d pred, one launch) against
    out_fwd + dpd_l1_loss(mode 2) + phase 1 of dpd_decoder_bwd_data: y, pred, dy, g3 bit for bit (same dot-product code, the chain
    only adds exact zeros), loss_pred up to the summation order; twice in a row (the arrival counter must re-arm itself)."""
    from dpdist_amd import ops
    g = torch.Generator().manual_seed(5)
    Q, KP = 2 * B * N, 64
    h1 = torch.randn(Q, H, generator=g).to(dev)
    h3 = torch.randn(Q, H, generator=g).to(dev)
    mask = (torch.rand(Q, generator=g) > 0.2).float().to(dev)
    W = [torch.randn(*sh, generator=g).to(dev) * 0.05 for sh in ((KP, H), (H,), (H, H), (H,), (H, H), (H,), (H, 3), (3,))]
    W[7] = W[7] + 1.0        # y around 1: both sides of the relu6 gate occur
    hh3, y0, p0 = ops.decoder_fwd(torch.randn(Q, KP, generator=g).to(dev), mask, W, H)[2:]     # the forward's own output layer
    y_ref = h3.double() @ W[6].double() + W[7].double()
    for rep in range(2):
        y, pred, loss, dy, g3 = ops.out_asloss(h3, mask, W, B * N)
        assert (y.double() - y_ref).abs().max().item() <= 1e-4
        l_ref, dpred = ops.l1_loss(pred, mask[:B * N], mode=2)
        assert abs(loss.item() - l_ref[1].item()) <= 2e-7 * max(1.0, abs(l_ref[1].item()))
        dy_r, g3_r, _, _, _ = ops.decoder_bwd_data(dpred, mask, y, h1, h1, h3, W, KP, False)
        assert torch.equal(dy, dy_r)
        assert torch.equal(g3, g3_r)
        assert (pred - torch.clamp(y, 0.0, 6.0) / 3.0 * mask[:, None]).abs().max().item() <= 1e-6
    # and y / pred are the bits of the forward's own output layer on the same rows
    y2, pred2, _, _, _ = ops.out_asloss(hh3, mask, W, B * N, want_grad=False)
    assert torch.equal(y2, y0) and torch.equal(pred2, p0)


def test_bf16_step_with_layer3_activation_as_a_plane(dev):
    """DPD_BF16 training steps keep layer 3's activation as ONE bf16 plane (dpd_planes.h3_rc: written by the layer-3 GEMM, read by the
    fused output-layer kernel) instead of fp32.  It rounds the output layer's input like every other activation of this compute type:
    against the fp32-h3 form (options["h3_plane"] = False) the losses of five steps agree to 2e-3 relative and the weights to the bf16 bar; the
    plane form is the one that runs by default, and evaluation-mode forwards (output layer wanted) still produce the fp32 h3."""
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    B = 32
    batches = [tuple(_cu(x, dev) for x in synth.s2_modelnet_shaped(B, 64, 400 + i)) for i in range(5)]
    W0 = synth.make_weights("wide")
    res = {}
    for v in ("1", "0"):
        P = DPDistParams(device=dev, compute_dtype="bf16")
        P.load_tf_state_dict(W0)
        tr = DPDistTrainer(P, B, 64, options={"h3_plane": v == "1"})
        losses = []
        for b in batches:
            losses.append(tr.step(*b).clone())
            assert tr._h3_in_plane == (v == "1")
        tr._take_front(batches[0][0], batches[0][1], None)
        tr._decode()                                  # evaluation-style forward: fp32 h3, y, pred
        assert not tr._h3_in_plane and torch.isfinite(tr.pred).all()
        torch.cuda.synchronize()
        res[v] = (torch.stack(losses).cpu().numpy(), P.flat.detach().cpu().numpy())
    la, lb = res["1"][0], res["0"][0]
    assert np.abs(la - lb).max() <= 2e-3 * np.abs(lb).max(), (la, lb)
    wa, wb = res["1"][1], res["0"][1]
    assert np.abs(wa - wb).max() <= 2e-3          # five Adam steps of 1e-4: the signs of a few tiny gradients may differ, nothing more


@pytest.mark.parametrize("dt", ["f32x3", "bf16"])
def test_plane_step_without_fp32_copies_is_bitwise_the_step_with_them(dev, dt, monkeypatch):
    """Plane compute types (round 3): fp32 h1 / h2 / g1 / g2 / g3 are not written at all -- layers 2/3 and the weight gradients read
    the bf16 planes, the backward's ReLU gate is taken from the plane (EPI_GATE / gate16), g3 leaves the fused output-layer backward
    as planes.  options["keep_f32_h"] keeps the copies (and the separate g3 conversion launch): same planes, same GEMM results (up to the order of the
    fp32-atomic bias-gradient sums these compute types use either way), three optimizer steps long; also through the data-parallel kernel order (single-rank RCCL group), whose weight-gradient calls differ."""
    import torch.distributed as dist
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    B = 32
    batches = [tuple(_cu(x, dev) for x in synth.s2_modelnet_shaped(B, 64, 100 + i)) for i in range(3)]
    W0 = synth.make_weights("wide")
    own_pg = not dist.is_initialized()
    if own_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29641")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        for dp in (False, True):
            outs = []
            for keep in ("1", "0"):
                monkeypatch.setenv("DPD_FORCE_DIST", "1" if dp else "0")
                P = DPDistParams(device=dev, compute_dtype=dt)
                P.load_tf_state_dict(W0)
                tr = DPDistTrainer(P, B, base_lr=1e-3, distributed=dp, options={"keep_f32_h": keep == "1"})
                assert (tr.h1 is None) == (keep == "0") and (tr.g3 is None) == (keep == "0")
                losses = [tr.step(*b).clone() for b in batches]
                torch.cuda.synchronize()
                outs.append((torch.stack(losses), P.flat.detach().clone(), tr.grad.clone()))
            assert torch.equal(outs[0][0][0], outs[1][0][0]), dp          # first step: identical weights -> identical loss bits
            assert (outs[0][0] - outs[1][0]).abs().max().item() <= 1e-6
            # gradients of the last step: the weight matrices come from the same GEMMs on the same planes; b1 / b2 are fp32-atomic column
            # sums in the plane compute types (order-dependent round-off from run to run, with or without the copies), which reaches the
            # other tensors through Adam only from the second step on
            ga, gb = outs[0][2], outs[1][2]
            for n, (off, cnt, _) in P._segments.items():
                a, b = ga[off:off + cnt], gb[off:off + cnt]
                assert (a - b).abs().max().item() <= 2e-6 * max(1.0, a.abs().max().item()), (dp, n)
            assert (outs[0][1] - outs[1][1]).abs().max().item() <= 2.1e-3      # Adam: a sign flip of a ~0 gradient moves a weight by 2 lr
            assert (outs[0][1] - outs[1][1]).abs().mean().item() <= 1e-6
    finally:
        if own_pg:
            dist.destroy_process_group()


@pytest.mark.parametrize("N", [36, 100])
@pytest.mark.parametrize("dt", ["f32x3", "bf16"])
def test_plane_paths_on_shapes_the_plane_kernels_do_not_take(dev, dt, N):
    """num_point 36 / 100 with B = 3: Q = 216 / 600 rows, contraction lengths that are not multiples of 32 in the backward --
    the decoder entry points run those GEMMs on the exact-fp32 kernel instead (documented fallback), results stay in bar,
    and both trainers (plane path off for these shapes) and the autograd path work."""
    from dpdist_amd import model as M
    from dpdist_amd.trainer import DPDistTrainer
    rng = np.random.default_rng(N)
    pcA = rng.uniform(-0.9, 0.9, (3, N, 3)).astype(np.float32)
    pcB = rng.uniform(-0.9, 0.9, (3, N, 3)).astype(np.float32)
    lab = np.abs(pcB[..., 0]).astype(np.float32)
    W = synth.make_weights("wide")
    ref = _oracle_forward(pcA, pcB, W)
    mod = _model(dev, "wide")
    mod.params_.compute_dtype = dt
    a = _cu(pcA, dev).requires_grad_(True)
    ps = mod(a, _cu(pcB, dev))
    tol = 1e-4 if dt == "f32x3" else 3e-2
    for n in ("pred_listAB", "pred_listBA"):
        assert np.abs(ps[n].detach().cpu().numpy() - ref[n].numpy()).max() <= tol
    loss = M.DPDistLoss(mod)(a, _cu(pcB, dev))
    (ga,) = torch.autograd.grad(loss, [a])
    assert torch.isfinite(ga).all() and ga.abs().max().item() > 0
    curves = {}
    for name in ("f32", dt):                       # the trainer on these shapes tracks the exact-fp32 trainer
        P = M.DPDistParams(device=dev, compute_dtype=name)
        P.load_tf_state_dict(W)
        tr = DPDistTrainer(P, 3, num_point=N, base_lr=1e-4, distributed=False)
        curves[name] = np.array([tr.step(_cu(pcA, dev), _cu(pcB, dev), _cu(lab, dev))[0].item() for _ in range(5)])
    assert np.isfinite(curves[dt]).all()
    assert np.abs(curves[dt] - curves["f32"]).max() <= (1e-4 if dt == "f32x3" else 2e-2)


def test_integration_md_stub_runs_as_written(dev, golden_dir, monkeypatch):
    """The ctypes stub printed in INTEGRATION.md section 2 is executed verbatim and must reproduce the golden forward."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes, torch.*?)```", text, re.S).group(1)
    monkeypatch.chdir(root)                       # the stub opens "dpdist_amd/libdpdist_hip.so" relative to the repo root
    from dpdist_amd import lib as L
    L.load()                                      # torch first, then the library (load order note of section 1)
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    d = _g(golden_dir, "path_fwd_s1_wide.npz")
    mod = _model(dev, "wide")
    v = mod.params_.views()
    params = ns["DecoderParams"](*[t.data_ptr() for t in v])
    pcA, pcB = _cu(d["pcA"], dev), _cu(d["pcB"], dev)
    out = ns["dpdist_forward"](pcA, pcB, torch.zeros_like(pcA), params)
    torch.cuda.synchronize()
    for n in ("pred_listAB", "pred_listBA"):
        _check_pred(out[n].cpu().numpy(), d[n + "_f64"], True)


# ------------------------------------------------------------------------------------------------ fused window gather (K2)
@pytest.mark.gpu
@pytest.mark.parametrize("B", [8, 32, 64])
@pytest.mark.parametrize("dt", ["f32x3", "bf16"])
def test_plane_weight_gradients_in_one_grouped_launch(dev, dt, B, monkeypatch):
    """dpd_decoder_bwd_weights_trio: dW1 (2528 x 1024), dW2 and dW3 (1024 x 1024) of a plane compute type as ONE grouped launch of
    problems with different row counts (plain and with the in-launch split-K) against the separate launches: the same gradients up
    to fp32 summation order, bitwise reproducible, and the same five training steps."""
    from dpdist_amd import ops
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    pcA, pcB, lab = (_cu(x, dev) for x in synth.s2_modelnet_shaped(B, 64, 100))
    res = {}
    try:
        for name, trio, tile, split in (("apart", "0", 0, 1), ("trio", "1", 0, 1), ("trio64", "1", 3, 1), ("trio_split2", "1", 2, 2),
                                        ("trio192", "1", 13, 1)):
            ops.set_gemm_plan(33, tile, split)
            P = DPDistParams(device=dev, compute_dtype=dt)
            P.load_tf_state_dict(synth.make_weights("wide"))
            tr = DPDistTrainer(P, B, base_lr=1e-3, distributed=False, options={"dw_trio": trio == "1"})
            assert tr._trio == (trio == "1")
            grads = []
            for rep in range(2):
                tr._take_front(pcA, pcB, None)
                tr._decode()
                tr.backward(lab.reshape(-1))
                torch.cuda.synchronize()
                grads.append({n: P.view(n, tr.grad).clone() for n in ("W1p", "W2", "W3")})
            assert tr._trio == (trio == "1")          # the grouped launch took these shapes
            for n in grads[0]:
                assert torch.equal(grads[0][n], grads[1][n]), (name, n)
            losses = torch.stack([tr.step(pcA, pcB, lab).clone() for _ in range(5)])
            res[name] = (grads[0], losses)
    finally:
        ops.set_gemm_plan(33, 0, 1)
    for name in ("trio", "trio64", "trio_split2", "trio192"):
        for n in ("W1p", "W2", "W3"):
            a, b = res[name][0][n], res["apart"][0][n]
            assert (a - b).abs().max().item() <= 2e-6 * b.abs().max().item() + 1e-9, (name, n)
        assert (res[name][1] - res["apart"][1]).abs().max().item() <= (1e-5 if dt == "f32x3" else 2e-3), name


@pytest.mark.gpu
@pytest.mark.parametrize("B,N", [(1, 64), (3, 64), (5, 32), (4, 100), (2, 200)])
def test_bf16_step_on_odd_batch_shapes(dev, B, N):
    """The bf16 training step at shapes where the round-4 grouped weight-gradient launch does or does not apply (query rows not a
    multiple of 64 -> the separate launches; not a multiple of 32 -> no planes at all): three steps track the exact-fp32 trainer."""
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    pcA, pcB, lab = (_cu(x, dev) for x in synth.s2_modelnet_shaped(B, N, 100))
    out = {}
    for dt in ("f32", "bf16"):
        P = DPDistParams(device=dev, compute_dtype=dt)
        P.load_tf_state_dict(synth.make_weights("wide"))
        tr = DPDistTrainer(P, B, num_point=N, base_lr=1e-3, distributed=False)
        losses = torch.stack([tr.step(pcA, pcB, lab).clone() for _ in range(3)])
        torch.cuda.synchronize()
        out[dt] = (losses, tr._trio, tr._planes is not None)
    assert torch.isfinite(out["bf16"][0]).all()
    assert (out["bf16"][0] - out["f32"][0]).abs().max().item() <= 3e-2
    BN = B * N
    assert out["bf16"][2] == (BN % 32 == 0 and (2 * BN) % 8 == 0)
    assert out["bf16"][1] == (out["bf16"][2] and BN % 64 == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [32, 64])
@pytest.mark.parametrize("dt", ["f32x3", "bf16"])
def test_plane_weight_gradient_split_k_in_launch(dev, dt, B):
    """Weight gradients of the plane compute types (K = query rows: 2048 / 4096): split-K with the reduction INSIDE the launch
    (gemm_x3.hip: inlaunch_reduce -- raw accumulator slabs, one agent-scope release per slice, the last-arriving slice adds the
    slabs in slice order) for dW1 and the grouped dW2 + dW3 launch, next to the round-2 form (fp32 slabs + a reduce launch) and
    the whole-K launches: same gradients up to fp32 summation order, all within the compute type's bar of the fp32 trainer; the
    in-launch form is BITWISE reproducible from run to run whatever slice arrives last, also when the arrival words start as
    garbage (they carry the launch's generation)."""
    from dpdist_amd import ops
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    pcA, pcB, lab = (_cu(x, dev) for x in synth.s2_modelnet_shaped(B, 64, 100))
    grads = {}
    plans = {"whole": ((0, 1), (3, 1)), "in2": ((2, 2), (2, 2)), "in3": ((2, 3), (2, 3)), "in4": ((3, 2), (3, 4)), "slab2": ((0, 1), (0, -2)),
             "auto": ((0, 0), (0, 0))}
    try:
        for name, ((t1, s1), (t23, s23)) in plans.items():
            ops.set_gemm_plan(16 + 4, t1, s1)          # dW1 of the plane compute types
            ops.set_gemm_plan(32, t23, s23)            # grouped dW2 + dW3
            P = DPDistParams(device=dev, compute_dtype=dt)
            P.load_tf_state_dict(synth.make_weights("wide"))
            tr = DPDistTrainer(P, B, distributed=False)
            runs = []
            for rep in range(3 if name.startswith("in") else 1):
                if rep == 1:
                    tr.ws.view(torch.int32).random_(-2 ** 31, 2 ** 31 - 1)      # arrival words (and slabs) start as garbage
                tr._take_front(pcA, pcB, None)
                tr._decode()
                tr.backward(lab.reshape(-1))
                torch.cuda.synchronize()
                runs.append({n: P.view(n, tr.grad).clone() for n in ("W1p", "W2", "W3")})
            for r in runs[1:]:
                for n in r:
                    assert torch.equal(r[n], runs[0][n]), (name, n)
            grads[name] = runs[0]
    finally:
        ops.set_gemm_plan(16 + 4, 0, 0)
        ops.set_gemm_plan(32, 0, 0)
    P = DPDistParams(device=dev)
    P.load_tf_state_dict(synth.make_weights("wide"))
    tr = DPDistTrainer(P, B, distributed=False)
    tr._take_front(pcA, pcB, None)
    tr._decode()
    tr.backward(lab.reshape(-1))
    for n in ("W1p", "W2", "W3"):
        ref = P.view(n, tr.grad)
        scale = ref.abs().max().item()
        for name in plans:
            a = grads[name][n]
            assert (a - grads["whole"][n]).abs().max().item() <= 4e-6 * scale + 1e-9, (name, n)   # same products, other summation order
            # (W1p: the gradient of the first layer sees every ReLU gate of the chain; a gate that flips on a last-bit difference of its
            # pre-activation moves single entries by ~1e-3 of the largest one in the fp32-equivalent type as well)
            tol = (2e-3 if n == "W1p" else 2e-5) if dt == "f32x3" else 3e-2
            assert (a - ref).abs().max().item() <= tol * scale, (name, n, (a - ref).abs().max().item(), scale)


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,m,k,noise", [(4, 64, 8, 5, True), (3, 50, 8, 5, False), (2, 64, 5, 3, True), (5, 33, 8, 5, True)])
def test_two_launch_front_end_is_bitwise_the_four_launches(dev, B, N, m, k, noise):
    """dpd_mfv3d_fwd_stacked (stacking + encoder, fv left without its L2 norm, per-slice sums of squares out) followed by
    dpd_patch_rows_fwd_scaled (norm applied while gathering) == dpd_stack_clouds, dpd_mfv3d_fwd (+ norm kernel), dpd_patch_rows_fwd."""
    from dpdist_amd import lib as L
    lib = L.load()
    s = L.cur_stream()
    g = torch.Generator().manual_seed(B * 100 + N)
    pcA = (torch.rand(B, N, 3, generator=g) * 2 - 1).to(dev)
    pcB = (torch.rand(B, N, 3, generator=g) * 2 - 1).to(dev)
    nz = (torch.randn(B, N, 3, generator=g) * 0.02).to(dev) if noise else None
    pcA[0, 0] = 5.0                      # a query outside the cube (mask 0)
    C, Q, G, KP = 2 * B, 2 * B * N, m ** 3, lib.dpd_padded_width(k)
    f = lambda *sh: torch.full(sh, float("nan"), device=dev)   # noqa: E731
    # four launches
    pts, q, fv, X, mask = f(C, N, 3), f(C, N, 3), f(C, G, 20), f(Q, KP), f(Q)
    vox = torch.zeros(Q, device=dev, dtype=torch.int32)
    L.check(lib.dpd_stack_clouds(L.ptr(pcA), L.ptr(pcB), L.ptr(nz), B, N, L.ptr(pts), L.ptr(q), s), "stack")
    L.check(lib.dpd_mfv3d_fwd(L.ptr(pts), C, N, m, 0.125, L.ptr(fv), s), "enc")
    L.check(lib.dpd_patch_rows_fwd(L.ptr(q), L.ptr(fv), C, N, m, k, KP, L.ptr(X), L.ptr(mask), L.ptr(vox), None, s), "gather")
    # two launches
    pts2, q2, fv2, X2, mask2, ssq = f(C, N, 3), f(C, N, 3), f(C, G, 20), f(Q, KP), f(Q), f(C, 4, 20)
    vox2 = torch.zeros(Q, device=dev, dtype=torch.int32)
    L.check(lib.dpd_mfv3d_fwd_stacked(L.ptr(pcA), L.ptr(pcB), L.ptr(nz), B, N, m, 0.125, L.ptr(pts2), L.ptr(q2), L.ptr(fv2),
                                      L.ptr(ssq), s), "enc2")
    L.check(lib.dpd_patch_rows_fwd_scaled(L.ptr(q2), L.ptr(fv2), L.ptr(ssq), C, N, m, k, KP, L.ptr(X2), L.ptr(mask2), L.ptr(vox2),
                                          None, s), "gather2")
    # and the stacked encoder WITH its own norm (ssq = NULL) is the plain encoder
    fv3 = f(C, G, 20)
    L.check(lib.dpd_mfv3d_fwd_stacked(L.ptr(pcA), L.ptr(pcB), L.ptr(nz), B, N, m, 0.125, None, None, L.ptr(fv3), None, s), "enc3")
    torch.cuda.synchronize()
    assert torch.equal(pts, pts2) and torch.equal(q, q2)
    assert torch.equal(mask, mask2) and torch.equal(vox, vox2)
    eq = lambda a, b: torch.equal(torch.nan_to_num(a, nan=123.0), torch.nan_to_num(b, nan=123.0))   # noqa: E731
    assert eq(fv, fv3)                   # cloud 0 of the (5.0) case is NaN in both (every pdf underflows)
    assert eq(X, X2)
    scale = 1.0 / torch.sqrt(torch.clamp(ssq.sum(1), min=1e-12))
    assert torch.allclose(torch.nan_to_num(fv2 * scale[:, None, :]), torch.nan_to_num(fv), rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_two_launch_front_end_in_the_trainer(dev, monkeypatch, dt):
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    B = 4
    pcA, pcB, lab = (_cu(x, dev) for x in synth.s2_modelnet_shaped(B, 64, 100))
    outs = []
    for flag in ("1", "0"):
        P = DPDistParams(device=dev, compute_dtype=dt)
        P.load_tf_state_dict(synth.make_weights("wide"))
        tr = DPDistTrainer(P, B, base_lr=1e-3, distributed=False, options={"front2": flag == "1"})
        assert tr.front2 == (flag == "1")
        losses = [tr.step(pcA, pcB, lab).clone() for _ in range(3)]
        ev = tr.evaluate(pcA, pcB, lab)[1].clone()
        outs.append((torch.stack(losses), P.flat.detach().clone(), ev))
    assert torch.equal(outs[0][0][0], outs[1][0][0])            # forward of the first step: same bits in every compute type
    for a, b in zip(*outs):
        if dt == "f32":
            assert torch.equal(a, b)
        else:       # the plane compute types take db1 / db2 with fp32 atomics: later steps agree to summation order only
            assert (a - b).abs().max().item() <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("B", [4, 3])
@pytest.mark.parametrize("mlp", [(1024, 1024, 1024), (256, 256, 256)])
def test_output_layer_forward_inside_its_backward_is_bitwise(dev, monkeypatch, mlp, B):
    """Training step: y / pred of BOTH directions computed inside out_bwd_fused4_kernel (dpd_small_grads.fwd_y, no out_fwd launch)
    against the separate output-layer forward (options["fuse_out"] = False): same y, pred, losses, gradients and weights, bit for bit."""
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    pcA, pcB, lab = (_cu(x, dev) for x in synth.s2_modelnet_shaped(B, 64, 100))
    outs = []
    for flag in ("1", "0"):
        P = DPDistParams(mlp=mlp, device=dev)
        P.load_tf_state_dict(synth.make_weights("wide", mlp=mlp))
        tr = DPDistTrainer(P, B, base_lr=1e-3, distributed=False, options={"fuse_out": flag == "1"})
        assert tr.fuse_out == (flag == "1")
        losses = [tr.step(pcA, pcB, lab).clone() for _ in range(3)]
        torch.cuda.synchronize()
        outs.append((torch.stack(losses), tr.y.clone(), tr.pred.clone(), tr.grad.clone(), P.flat.detach().clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["bf16", "f32x3"])
@pytest.mark.parametrize("mlp", [(1024, 1024, 1024), (64, 64, 64)])
def test_one_launch_optimizer_writes_the_weight_planes(dev, dt, mlp):
    """Plane compute types: dpd_adam_tf_fused leaves the bf16 operand planes of W1/W2/W3 exactly as dpd_weights_to_planes
    would write them from the updated weights (so the separate conversion launch is skipped)."""
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    B = 4
    pcA, pcB, lab = (_cu(x, dev) for x in synth.s2_modelnet_shaped(B, 64, 100))
    P = DPDistParams(mlp=mlp, device=dev, compute_dtype=dt)
    P.load_tf_state_dict(synth.make_weights("wide", mlp=mlp))
    tr = DPDistTrainer(P, B, base_lr=1e-3, distributed=False)
    assert tr._planes is not None and tr._afuse[0].np == (1 if dt == "bf16" else 3)
    for _ in range(3):
        tr.step(pcA, pcB, lab)
    assert not tr._wdirty                                  # nothing left to re-derive after the optimizer
    torch.cuda.synchronize()
    got = tr._plane_mem.clone()
    tr.refresh_weight_planes()                             # the separate conversion launch, from the same weights
    torch.cuda.synchronize()
    assert torch.equal(got, tr._plane_mem)


@pytest.mark.gpu
@pytest.mark.parametrize("mlp", [(1024, 1024, 1024), (64, 64, 64), (256, 256, 256)])
def test_one_launch_optimizer_is_bitwise_the_three_launches(dev, monkeypatch, mlp):
    """dpd_adam_tf_fused (Adam + transposed weight copies + the reduction of the output layer's block partials in ONE launch)
    against options["fused_adam"] = False (dpd_adam_tf, dpd_weights_transpose, small_grads_reduce): same weights, moments, gradients,
    transposed copies and losses, bit for bit, over four steps."""
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    B = 4
    pcA, pcB, lab = (_cu(x, dev) for x in synth.s2_modelnet_shaped(B, 64, 100))
    outs = []
    for fused in ("1", "0"):
        P = DPDistParams(mlp=mlp, device=dev)
        P.load_tf_state_dict(synth.make_weights("wide", mlp=mlp))
        tr = DPDistTrainer(P, B, base_lr=1e-3, distributed=False, options={"fused_adam": fused == "1"})
        assert tr.fused_adam == (fused == "1")
        assert tr._tail_ok == (mlp[0] % 256 == 0)            # the block-partial tail really takes part at H = 1024 / 256
        losses = [tr.step(pcA, pcB, lab).clone() for _ in range(4)]
        if tr._wdirty:
            tr.refresh_weight_planes()
        torch.cuda.synchronize()
        outs.append((torch.stack(losses), P.flat.detach().clone(), tr.m_state.clone(), tr.v_state.clone(), tr.grad.clone(),
                     tr.W2T.clone(), tr.W3T.clone()))
        assert torch.equal(tr.W2T, P.view("W2").t()) and torch.equal(tr.W3T, P.view("W3").t())
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_checkpoint_resume_restores_optimizer_state(dev, tmp_path):
    """What the reference's tf.train.Saver() stores (train_multi_gpu_pc_compare_dist.py:305,354-357) round-trips through the TF
    V2 container: the 8 variables, `batch`, beta1_power / beta2_power and the `<variable>/Adam`, `/Adam_1` slots -- a resumed run
    continues exactly where the uninterrupted one is (global step, staircase learning rate, Adam moments)."""
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.tf_checkpoint import list_variables, read_checkpoint, write_checkpoint
    from dpdist_amd.trainer import DPDistTrainer
    mlp = (64, 64, 64)
    B = 4
    pcA, pcB, lab = (_cu(x, dev) for x in synth.s2_modelnet_shaped(B, 64, 100))

    def fresh():
        P = DPDistParams(mlp=mlp, device=dev)
        P.load_tf_state_dict(synth.make_weights("wide", mlp=mlp))
        return P, DPDistTrainer(P, B, base_lr=1e-3, decay_step=2, decay_rate=0.5, distributed=False)

    P0, t0 = fresh()
    for _ in range(5):
        t0.step(pcA, pcB, lab)
    P1, t1 = fresh()
    for _ in range(3):
        t1.step(pcA, pcB, lab)
    sd = t1.tf_global_variables()
    names = sorted(sd)
    base = sorted(P1.tf_state_dict())
    assert names == sorted(base + [n + s for n in base for s in ("/Adam", "/Adam_1")] + ["batch", "beta1_power", "beta2_power"])
    assert float(sd["batch"]) == 3.0 and abs(float(sd["beta1_power"]) - 0.9 ** 4) < 1e-6 and abs(float(sd["beta2_power"]) - 0.999 ** 4) < 1e-6
    prefix = str(tmp_path / "model.ckpt")
    write_checkpoint(prefix, sd)
    assert sorted(list_variables(prefix)) == names
    P2, t2 = fresh()
    got = t2.load_tf_global_variables(read_checkpoint(prefix))
    assert got == ["weights", "adam_slots", "schedule"] and t2.t == 3
    assert torch.equal(t2.m_state, t1.m_state) and torch.equal(t2.v_state, t1.v_state) and torch.equal(P2.flat, P1.flat)
    for _ in range(2):
        t2.step(pcA, pcB, lab)
    assert abs(t2.lr - t0.lr) == 0.0 and t2.t == t0.t == 5               # 1e-3 * 0.5^floor(4/2)
    assert (P2.flat - P0.flat).abs().max().item() <= 1e-5                 # the 1024-wide layers' db atomics are absent at 64 wide
    # a weights-only checkpoint leaves the optimizer state alone
    P3, t3 = fresh()
    assert t3.load_tf_global_variables(P1.tf_state_dict()) == ["weights"] and t3.t == 0


@pytest.mark.parametrize("shape", [(2528, 1024, 2048, 33), (1184, 512, 1024, 33), (2528, 1024, 4096, 33), (2048, 1024, 1024, 32)])
def test_gemm_tail_split(dev, shape):
    """split_k = 0 ("tail split", csrc/gemm_rs.h): whole-K tiles, only the partial last round of workgroup tiles is cut along K and
    re-added in a fixed order.  Against fp64; bitwise equal to the plain launch wherever a tile was not cut; shapes whose tile
    count fills whole rounds run as a plain launch."""
    from dpdist_amd import ops
    M, N, K, tile = shape
    g = torch.Generator().manual_seed(M + K)
    At = torch.randn(K, M, generator=g).to(dev)          # TN: the weight-gradient form
    Bm = torch.randn(K, N, generator=g).to(dev)
    ref = At.double().t() @ Bm.double()
    plain = ops.gemm_f32(At, Bm, transA=True, tile=tile, split_k=1)
    ws_big = torch.empty(3 * M * N, device=dev)
    C = torch.empty(M, N, device=dev)
    from dpdist_amd import lib as L
    L.check(L.load().dpd_gemm_f32(1, 0, M, N, K, L.ptr(At), M, L.ptr(Bm), N, L.ptr(C), N, None, None, 0, 0, tile, L.ptr(ws_big),
                                  ws_big.numel() * 4, L.cur_stream()), "dpd_gemm_f32")
    scale = float(ref.abs().max())
    assert (C.double() - ref).abs().max().item() <= 2e-6 * scale * (K / 1024) ** 0.5 + 1e-4
    same = (C == plain).all(dim=1)
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    if tiles % 256 == 0 or tiles < 256 or tiles % 256 >= 192:
        assert same.all()                                   # nothing to cut
    else:
        assert same.any() and not same.all()                # the whole-K rows are bitwise the plain result, the tail rows are re-associated
        first_cut = int((~same).nonzero()[0])
        assert same[:first_cut].all() and first_cut % 64 == 0


# ------------------------------------------------------------------------------------------------ the headline workload's backward (round 5)
_BENCH_SHAPE_ORACLE = {}


@pytest.mark.gpu
@pytest.mark.parametrize("B", [32, 64])
@pytest.mark.parametrize("dt", ["f32", "f32x3"])
def test_training_step_at_the_bench_shape_vs_oracle(dev, dt, B):
    """ONE DPDistTrainer.step at the shape bench.py times (B = 32: BASELINE's metric; B = 64: configs 3-4's rows), S2 clouds of bench.py's
    seed, `wide` weights, against the oracle's float64 autograd (train_multi_gpu_pc_compare_dist.py:274-302: loss_samples, the gradients of
    the 8 variables, tf.train.AdamOptimizer): losses, every gradient tensor (norm, 16 x 16 corners, row and column sums, and the WHOLE tensor
    elementwise) and the post-Adam weights.  Here BN = 2048 / 4096 rows carry gradient, i.e. the weight-gradient plan, its tile rounds and
    the deterministic bias sums run as they do in the benchmark (the older step tests stop at B = 4)."""
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer, learning_rate
    from oracle import restate as R
    pcA, pcB, lab = synth.s2_modelnet_shaped(B, 64, 100)
    W0 = synth.make_weights("wide")
    P = DPDistParams(device=dev, compute_dtype=dt)
    P.load_tf_state_dict(W0)
    tr = DPDistTrainer(P, B, base_lr=1e-3, distributed=False)
    loss = tr.step(_cu(pcA, dev), _cu(pcB, dev), _cu(lab, dev)).cpu().numpy()
    grads = P.tf_state_dict(tr.grad)
    after = P.tf_state_dict()
    if B not in _BENCH_SHAPE_ORACLE:      # float64 autograd of the oracle: ~10-40 s of host time, shared by the compute types
        torch.set_num_threads(8)
        Wt = {n: torch.tensor(a, dtype=torch.float64, requires_grad=True) for n, a in W0.items()}
        pred, _ = R.get_model(torch.tensor(pcA, dtype=torch.float64), torch.tensor(pcB, dtype=torch.float64), Wt)
        ls, lp = R.get_loss(pred, torch.tensor(lab, dtype=torch.float64))
        names = sorted(Wt)
        gs = torch.autograd.grad(ls, [Wt[n] for n in names])
        _BENCH_SHAPE_ORACLE[B] = (ls.item(), lp.item(), {n: g.numpy() for n, g in zip(names, gs)})
    ls, lp, gref = _BENCH_SHAPE_ORACLE[B]
    assert abs(loss[0] - ls) <= 2e-5 and abs(loss[1] - lp) <= 2e-5, (loss, ls, lp)
    Wt = {n: torch.tensor(a, dtype=torch.float64) for n, a in W0.items()}
    for n in sorted(gref):
        ref = gref[n]
        got = grads[n].astype(np.float64)
        r2, g2 = (ref.reshape(-1, ref.shape[-1]), got.reshape(-1, got.shape[-1])) if ref.ndim == 4 else (ref, got)
        nrm = float(np.sqrt((r2 ** 2).sum()))
        tol = 2e-4 * max(1.0, nrm)                       # the bar of test_weight_gradients_golden
        assert abs(np.sqrt((g2 ** 2).sum()) - nrm) <= tol, (n, nrm)
        # elementwise: fp32 sums of BN = 2048 / 4096 products per entry against float64.  The loss is only piecewise smooth: a hidden unit
        # that is +1e-8 in float64 and exactly 0 in fp32 flips its ReLU gate and moves ONE row's contribution (~ 0.1 / BN) in the entries
        # it feeds -- seen here in ONE entry of db2 and the same column of dW2 (6e-5 / 3e-4 at B = 32, half that at B = 64, the same in
        # f32 and f32x3) -- so entries beyond the tight bar must sit in at most three columns and stay within a few rows' contributions
        err = np.abs(g2 - r2)
        tight = 1e-5 * max(1.0, np.abs(r2).max()) + 2e-6
        out = err > tight
        if out.any():      # a flipped gate of g[row, col] moves column `col` of that layer's dW (by h[row, :] * g) and entry `col` of its db
            cols = np.unique(np.nonzero(out)[-1])
            assert len(cols) <= 3, (n, len(cols), err.max())
            assert err.max() <= 16.0 / (B * 64), (n, err.max(), np.abs(r2).max())
        if ref.ndim == 4:
            assert np.abs(g2[:16, :16] - r2[:16, :16]).max() <= tol and np.abs(g2[-16:, -16:] - r2[-16:, -16:]).max() <= tol, n
            assert np.abs(g2.sum(0) - r2.sum(0)).max() <= tol * 30 and np.abs(g2.sum(1) - r2.sum(1)).max() <= tol * 30, n
        # tf.train.AdamOptimizer's first step, in float64.  Its update lr * g / (|g| + 3.2e-7) is steep around g = 0 (layer 1's gradients are
        # ~1e-4 here), so the optimizer is pinned on the gradient the GPU produced (itself pinned elementwise above) to fp32 rounding of the
        # parameter; against the oracle's own gradient only the step size is bounded
        p = Wt[n].detach().numpy().copy()
        R.adam_tf_step(p, got, np.zeros_like(p), np.zeros_like(p), 1, learning_rate(0, 1e-3))
        assert np.abs(after[n] - p).max() <= 3e-7 * max(1.0, np.abs(p).max()), (n, np.abs(after[n] - p).max())
        p = Wt[n].detach().numpy().copy()
        R.adam_tf_step(p, ref, np.zeros_like(p), np.zeros_like(p), 1, learning_rate(0, 1e-3))
        assert np.abs(after[n] - p).max() <= 2.1e-3, n          # (at most the two steps apart of a sign flip at g ~ 0)


# ------------------------------------------------------------------------------------------------ as-loss engine (round 5)
@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f32", "f32x3", "bf16"])
def test_as_loss_engine_is_bitwise_the_entry_by_entry_node(dev, dt, monkeypatch):
    """dpd_asloss_forward / dpd_asloss_backward (ONE foreign call per direction on an engine's persistent buffers) against the autograd
    node that drives the C ABI entry by entry (asloss.ENGINE = False): loss and both input gradients BIT FOR BIT at the registration
    batch; forward-only evaluations under no_grad; several evaluations alive before their backwards (each holds its own engine, the
    fifth falls back to the allocating path); a node dropped without a backward frees its engine; a second backward through the same
    node works while its engine has not been re-used and raises once it has."""
    import gc
    from dpdist_amd import asloss, model as M
    B = 16
    pcA, pcB, _ = synth.s2_modelnet_shaped(B, 64, 100)
    res = {}
    for eng in ("0", "1"):
        monkeypatch.setattr("dpdist_amd.asloss.ENGINE", eng == "1")
        mod = _model(dev, "wide")
        mod.params_.compute_dtype = dt
        fn = M.DPDistLoss(mod)
        a1, b1 = _cu(pcA, dev).requires_grad_(True), _cu(pcB, dev).requires_grad_(True)
        a2, b2 = _cu(pcB, dev).requires_grad_(True), _cu(pcA, dev).requires_grad_(True)
        with torch.no_grad():
            l0 = fn(a1, b1).clone()
        l1 = fn(a1, b1)
        l2 = fn(a2, b2)                                   # a second evaluation before the first backward
        g1 = torch.autograd.grad(l1 * 2.0, [a1, b1])
        g2 = torch.autograd.grad(l2, [a2, b2])
        res[eng] = (l0, l1.detach().clone(), l2.detach().clone(), g1, g2)
        if eng == "1":
            pool = mod.params_._asloss_engines
            engines = next(iter(pool.values()))
            assert len(engines) == 2 and not any(e.busy for e in engines)
            # five evaluations alive: four engines, the fifth on the allocating path -- and all five gradients are right
            live = [fn(a1, b1) for _ in range(5)]
            assert len(engines) == asloss.MAX_ENGINES and all(e.busy for e in engines)
            for lv in live:
                g = torch.autograd.grad(lv * 2.0, [a1, b1])
                assert torch.equal(g[0], g1[0]) and torch.equal(g[1], g1[1])
            assert not any(e.busy for e in engines)
            # a node dropped without a backward gives its engine back
            lv = fn(a1, b1)
            assert sum(e.busy for e in engines) == 1
            del lv
            gc.collect()
            assert not any(e.busy for e in engines)
            # retain_graph: a second backward through the same node while its engine still holds that evaluation ...
            lv = fn(a1, b1)
            ga = torch.autograd.grad(lv * 2.0, [a1, b1], retain_graph=True)
            gb = torch.autograd.grad(lv * 2.0, [a1, b1], retain_graph=True)
            assert torch.equal(ga[0], gb[0]) and torch.equal(ga[0], g1[0])
            # ... and a clear error once a later evaluation has taken the engine over
            fn(a2, b2).backward()
            with pytest.raises(RuntimeError, match="re-used"):
                torch.autograd.grad(lv * 2.0, [a1, b1])
            # ... also when that later evaluation kept no state of its own (no_grad): it still overwrote the engine's buffers
            asloss.release_all(mod.params_)
            assert asloss.pool_bytes(mod.params_) == 0
            lv = fn(a1, b1)
            torch.autograd.grad(lv * 2.0, [a1, b1], retain_graph=True)
            with torch.no_grad():
                fn(a2, b2)
            with pytest.raises(RuntimeError, match="re-used"):
                torch.autograd.grad(lv * 2.0, [a1, b1])
            # the pool is bounded in bytes: idle engines of the least recently used shapes go first
            old_cap, asloss.MAX_POOL_BYTES = asloss.MAX_POOL_BYTES, int(2.5 * next(iter(pool.values()))[0].nbytes)
            try:
                for Bx in (15, 14, 13):
                    px, py, _ = synth.s2_modelnet_shaped(Bx, 64, 100)
                    with torch.no_grad():
                        fn(_cu(px, dev), _cu(py, dev))
                assert asloss.pool_bytes(mod.params_) <= asloss.MAX_POOL_BYTES and len(pool) == 2
                assert [k[0] for k in pool] == [14, 13]
            finally:
                asloss.MAX_POOL_BYTES = old_cap
    for x, y in zip(res["0"][:3], res["1"][:3]):
        assert torch.equal(x, y), (x.item(), y.item())
    for ga, gb in zip(res["0"][3] + res["0"][4], res["1"][3] + res["1"][4]):
        assert torch.equal(ga, gb), (ga - gb).abs().max().item()


@pytest.mark.gpu
def test_as_loss_engine_c_entry_forward_backward(dev):
    """dpd_asloss_forward_backward straight through the C ABI (no autograd in between): the same loss and gradients as the autograd
    node, at a batch whose rows are not plane-shaped for the bf16 type (the engine then carves the exact type's buffers)."""
    from dpdist_amd import asloss, lib as L, model as M
    for dt, B, N in (("f32", 16, 64), ("bf16", 3, 36)):
        pcA, pcB, _ = synth.s2_modelnet_shaped(B, N, 7)
        mod = _model(dev, "wide")
        mod.params_.compute_dtype = dt
        a, b = _cu(pcA, dev).requires_grad_(True), _cu(pcB, dev).requires_grad_(True)
        loss = M.DPDistLoss(mod)(a, b)
        gA, gB = torch.autograd.grad(loss, [a, b])
        P = mod.params_
        e = asloss.Engine(P, B, N, 8, 5, mod.sigma if hasattr(mod, "sigma") else 0.125, dev)
        e.set_weights(P, P.flat)
        out = torch.empty(1, device=dev)
        g1, g2 = torch.empty(B, N, 3, device=dev), torch.empty(B, N, 3, device=dev)
        L.check(asloss._lib().dpd_asloss_forward_backward(e.c, L.ptr(a.detach()), L.ptr(b.detach()), L.ptr(out), L.ptr(g1), L.ptr(g2),
                                                          L.cur_stream()), "dpd_asloss_forward_backward")
        assert torch.equal(out[0], loss.detach()) and torch.equal(g1, gA) and torch.equal(g2, gB), dt


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,m,k", [(16, 64, 8, 5), (3, 100, 8, 5), (5, 36, 8, 5), (2, 64, 5, 3), (32, 64, 8, 5)])
def test_as_loss_tail_in_three_launches_is_bitwise_the_six(dev, B, N, m, k):
    """dpd_asloss_tail (round 6: [window-gather backward || encoder statistics] -> combine -> apply + both input gradients) against
    dpd_patch_rows_bwd + dpd_mfv3d_bwd + dpd_asloss_combine on the same dX: the same device routines in another launch shape, so gA / gB
    and the dfv scratch agree bit for bit, with and without an upstream scale; a cloud of fewer than 8 points is refused (the caller falls back)."""
    from dpdist_amd import lib as L, ops
    lib = L.load()
    C, Q, KP = 2 * B, 2 * B * N, ops.padded_width(k)
    pcA, pcB, _ = synth.s2_modelnet_shaped(B, N, 11)
    pts = _cu(np.concatenate([pcA, pcB]), dev)
    q = _cu(np.concatenate([pcB, pcA]), dev)
    q[0, :2] += 5.0                                   # two queries outside the grid
    fv = ops.mfv3d_fwd(pts, m, 0.125)
    _, _, vox = ops.patch_rows_fwd(q, fv, m, k)
    g = torch.Generator().manual_seed(B * 1000 + N)
    dX = (torch.randn(Q, KP, generator=g) * 1e-2).to(dev)
    for scale in (None, torch.tensor([0.37], device=dev)):
        _, dfv = ops.patch_rows_bwd(dX, vox, C, N, m, k, want_dq=False)
        dpts = ops.mfv3d_bwd(pts, dfv, m, 0.125)
        gA, gB = ops.asloss_combine(dpts, dX, scale, B, N, k)
        ws = torch.empty(lib.dpd_mfv3d_bwd_workspace_bytes(C, m) // 4, device=dev)
        dfv2 = torch.full_like(dfv, float("nan"))
        gA2, gB2 = torch.full_like(gA, float("nan")), torch.full_like(gB, float("nan"))
        L.check(lib.dpd_asloss_tail(L.ptr(dX), L.ptr(vox), L.ptr(pts), L.ptr(scale), B, N, m, k, KP, 0.125, L.ptr(dfv2), L.ptr(ws), ws.numel() * 4,
                                    L.ptr(gA2), L.ptr(gB2), L.cur_stream()), "dpd_asloss_tail")
        assert torch.equal(dfv2, dfv) and torch.equal(gA2, gA) and torch.equal(gB2, gB), (B, N, scale)
    small = torch.zeros(2, 4, 3, device=dev)
    assert lib.dpd_asloss_tail(L.ptr(dX), L.ptr(vox), L.ptr(small), None, 1, 4, m, k, KP, 0.125, L.ptr(dfv2), L.ptr(ws), ws.numel() * 4,
                               L.ptr(gA2), L.ptr(gB2), L.cur_stream()) == -3


# ------------------------------------------------------------------------------------------------ data-parallel schedule by measurement (round 5)
@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_data_parallel_schedule_is_selected_by_measurement(dev, dt, monkeypatch):
    """DPDistTrainer.select_dp_schedule on a single-rank RCCL group: every candidate (early / grouped when the grouped weight-gradient
    launch exists / late) is timed over real steps with the reducer in the loop, the winner becomes the trainer's order, the record holds
    every candidate's time -- and the parameters, Adam slots and step counter are BIT FOR BIT what they were, so the training run that
    follows is the run that would have happened without the measurement.  DPD_DP_SCHEDULE still pins the order."""
    import torch.distributed as dist
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    monkeypatch.setenv("DPD_DP_BACKEND", "rccl")
    monkeypatch.setenv("DPD_FORCE_DIST", "1")
    monkeypatch.delenv("DPD_DP_SCHEDULE", raising=False)
    B = 32
    pcA, pcB, lab = (_cu(x, dev) for x in synth.s2_modelnet_shaped(B, 64, 100))
    own = _single_rank_group(dev, 29653)
    try:
        P = DPDistParams(device=dev, compute_dtype=dt)
        P.load_tf_state_dict(synth.make_weights("wide"))
        tr = DPDistTrainer(P, B, base_lr=1e-3, distributed=True)
        for _ in range(2):
            tr.step(pcA, pcB, lab)
        tr.join_optimizer()
        torch.cuda.synchronize()
        before = (P.flat.detach().clone(), tr.m_state.clone(), tr.v_state.clone(), tr.t)
        info = tr.select_dp_schedule(pcA, pcB, lab, steps=4, warmup=1)
        tr.join_optimizer()
        torch.cuda.synchronize()
        cands = ["early", "grouped", "late"] if dt == "bf16" else ["early", "late"]
        table = info["candidates_ms"]
        assert sorted(table) == ["early", "grouped", "late"] and (table["grouped"] is None) == (dt != "bf16")      # dropped collectively, never timed
        assert all(table[c] and table[c] > 0 for c in cands)
        assert info["schedule"] == tr.dp_schedule and info["schedule"] in cands and info["mode"] == tr.reducer.mode == "allreduce"
        assert table[info["schedule"]] <= min(table[c] for c in cands) + 1e-4      # (the table is rounded to 1e-4 ms)
        assert torch.equal(P.flat.detach(), before[0]) and torch.equal(tr.m_state, before[1]) and torch.equal(tr.v_state, before[2])
        assert tr.t == before[3]
        l1 = tr.step(pcA, pcB, lab).clone()
        assert torch.isfinite(l1).all()
        # the wider choice: order x communication form; the reducer is re-created per form (communicators, cross-check),
        # the sharded optimizer's slots are made whole again, and the state is still bit for bit what it was
        tr.join_optimizer()
        torch.cuda.synchronize()
        before = (P.flat.detach().clone(), tr.m_state.clone(), tr.v_state.clone(), tr.t)
        info3 = tr.select_dp_schedule(pcA, pcB, lab, steps=3, warmup=1, spinup=5, modes=("allreduce", "rs_ag", "zero1"))
        tr.join_optimizer()
        torch.cuda.synchronize()
        t3 = info3["candidates_ms"]
        assert sorted(t3) == sorted("%s/%s" % (m, o) for m in ("allreduce", "rs_ag", "zero1") for o in ("early", "grouped", "late"))
        assert t3["zero1/grouped"] is None and (t3["allreduce/grouped"] is None) == (dt != "bf16")
        assert all(t3["%s/%s" % (m, o)] > 0 for m in ("allreduce", "rs_ag", "zero1") for o in ("early", "late"))
        assert info3["mode"] == tr.reducer.mode and info3["schedule"] == tr.dp_schedule and tr.reducer.crosscheck["ok"]
        assert t3["%s/%s" % (info3["mode"], info3["schedule"])] <= min(v for v in t3.values() if v) + 1e-4
        assert torch.equal(P.flat.detach(), before[0]) and torch.equal(tr.m_state, before[1]) and torch.equal(tr.v_state, before[2])
        assert tr.t == before[3]
        assert torch.isfinite(tr.step(pcA, pcB, lab)).all()
        tr.close()
        # pinned by the environment: nothing is measured
        monkeypatch.setenv("DPD_DP_SCHEDULE", "late")
        tr2 = DPDistTrainer(P, B, base_lr=1e-3, distributed=True)
        info2 = tr2.select_dp_schedule(pcA, pcB, lab)
        assert info2["schedule"] == "late" and "candidates_ms" not in info2 and tr2.dp_schedule == "late"
        # ... and with the order pinned the communication form can still be measured (one order x three forms)
        info4 = tr2.select_dp_schedule(pcA, pcB, lab, steps=2, warmup=1, spinup=2, modes=("allreduce", "zero1"))
        assert sorted(info4["candidates_ms"]) == ["allreduce/late", "zero1/late"] and info4["schedule"] == "late"
        tr2.close()
        # DPD_DP_SCHEDULE=auto: the trainer starts on the deterministic default and a caller's select_dp_schedule measures
        monkeypatch.setenv("DPD_DP_SCHEDULE", "auto")
        tr3 = DPDistTrainer(P, B, base_lr=1e-3, distributed=True)
        assert tr3.dp_schedule == "early"
        tr3.close()
    finally:
        if own:
            dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("B", [32, 64])
def test_plane_weight_gradient_pair_on_a_narrow_decoder(dev, B, monkeypatch):
    """with a narrow decoder (H = 64) and >= 2048 gradient rows the automatic in-launch split-K of the grouped dW2 + dW3 launch
    asked for more tile-padded slab space than the base workspace holds and the call failed with DPD_E_WORKSPACE.  The split is now fitted
    to the slab region (down to no split at all): the separate-launch backward (options["dw_trio"] = False: dW1, then the pair) runs and its weight
    gradients agree with the exact-fp32 trainer's to bf16 accuracy."""
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    mlp = (64, 64, 64)
    pcA, pcB, lab = (_cu(x, dev) for x in synth.s2_modelnet_shaped(B, 64, 100))
    grads = {}
    for dt in ("f32", "bf16"):
        P = DPDistParams(mlp=mlp, device=dev, compute_dtype=dt)
        P.load_tf_state_dict(synth.make_weights("wide", mlp=mlp))
        tr = DPDistTrainer(P, B, base_lr=1e-3, distributed=False, options={"dw_trio": False})
        tr._take_front(pcA, pcB, None)
        tr._decode()
        tr.backward(lab.reshape(-1))
        torch.cuda.synchronize()
        grads[dt] = {n: P.view(n, tr.grad).clone() for n in ("W1p", "W2", "W3")}
    for n in ("W1p", "W2", "W3"):
        a, b = grads["bf16"][n].double().flatten(), grads["f32"][n].double().flatten()
        cos = float(a @ b / (a.norm() * b.norm() + 1e-30))
        assert cos >= 0.995 and abs(float(a.norm() / (b.norm() + 1e-30)) - 1.0) <= 0.03, (n, cos)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_ranks_share_the_gpu(dev, world):
    if world == 4 and os.environ.get("DPD_TEST_WORLD4") != "1":
        # On the shared one-GPU test boxes gloo's FOUR-rank path through the host takes 2-20 s per step for the single-collective ("late")
        # forms (candidates_ms of round 6: allreduce/late 2.6 s, rs_ag/late 19.9 s against 12-65 ms for "early"), 5-10 minutes for this case
        # alone -- while 2 ranks run in 5 s and 8 ranks in 20-30 s on the same box.  A property of the test transport, not of the path under
        # test (RCCL on real nodes); the case passes (DPD_TEST_WORLD4=1 runs it), it is just not worth ten minutes of every suite run.
        pytest.skip("world size 4 over gloo on one GPU: minutes per run; set DPD_TEST_WORLD4=1")
    """bench.py's world-size-N control flow for real, on a one-GPU box: DPD_TEST_SHARE_GPU=1 puts all ranks on GPU 0 over gloo (timings
    mean nothing).  Exactly the driver's command (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) for N = 2, 4
    and 8: N supervisors and their workers, the rendezvous, the reducer's start-up cross-check with N ranks (its amplitude adapts to N),
    order x communication form of the backward chosen by ALL ranks together for the headline and for config 4 (the reducer is re-created
    per form: all-reduce, reduce-scatter + all-gather, the sharded optimizer on its real partition), the pinned legs incl. zero1 and the
    bf16 wire, the same-run one-rank leg, the exposed-communication probe, and the final comparison of all N replicas' parameters and
    Adam slots -- which must be bit-identical after all those steps.  The line also carries dp.model (the expectation for real nodes)."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    so.close()
    # (N ranks time-share one GPU and reduce through the host: on a busy box a gloo step of the 18.7 MB gradient was seen to take 2-20 s --
    # one timed step per candidate, no spin-up; bench.py refreshes the watchdog's heartbeat per candidate, so slow is not taken for hung)
    env = dict(os.environ, DPD_TEST_SHARE_GPU="1", DPD_DP_SELECT="1,1,0")
    for k in ("DPD_BENCH_CHILD", "DPD_DP_BACKEND", "DPD_DP_MODE", "DPD_DP_SCHEDULE", "DPD_FORCE_DIST", "RANK", "WORLD_SIZE", "LOCAL_RANK",
              "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines                                    # rank 0 prints ONE line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == world and rec["value"] > 0 and rec["config"]["global_batch"] == 32 * world and "TEST MODE" in rec["data"]
    dp = rec["dp"]
    assert dp["crosscheck"]["ok"] and dp["replicas_bit_identical"] is True and rec["fallback"] is False and dp["nranks"] == world
    forms = {"%s/%s" % (m, o) for m in ("allreduce", "rs_ag", "zero1") for o in ("early", "grouped", "late")}
    sch = dp["schedule"]
    assert set(sch["candidates_ms"]) == forms and sch["schedule"] in ("early", "late") and sch["mode"] == dp["mode"]
    assert all((sch["candidates_ms"][f] is None) == f.endswith("grouped") for f in forms)         # f32: no grouped launch, dropped by all ranks
    assert set(dp["model"]["per_world"]) == {"2", "4", "8"} and 0 < dp["model"]["per_world"]["8"]["predicted_efficiency"] <= 1
    c4 = rec["config4"]
    assert c4["n_gpus"] == world and c4["global_batch"] == 64 * world and c4["dp"]["replicas_bit_identical"] is True
    s4 = c4["dp"]["schedule"]
    assert set(s4["candidates_ms"]) == forms and s4["candidates_ms"]["zero1/grouped"] is None and s4["candidates_ms"]["allreduce/grouped"] > 0
    for leg in ("early_schedule", "grouped_schedule", "zero1", "bf16_wire", "n1_same_run"):
        assert c4[leg]["ms_per_step"] > 0, (leg, c4[leg])
    assert c4["zero1"]["dp"]["mode"] == "zero1" and c4["zero1"]["dp"]["replicas_bit_identical"] is True
    assert c4["bf16_wire"]["dp"]["wire"] == "bf16" and c4["bf16_wire"]["dp"]["crosscheck"]["ok"]
    assert "weak_scaling_efficiency_vs_n1_same_run" in c4 and set(c4["dp"]["model"]["per_world"]) == {"2", "4", "8"}


@pytest.mark.gpu
def test_bench_line_on_one_gpu(dev):
    """`python bench.py` as the driver runs it at N = 1 (short): ONE JSON line with the contract's keys, the roofline object measured by
    hipEvents in this run, and -- SURVEY 8(d), VERDICT r5 #3 -- the forward-only and as-loss legs, config 3 and the data-parallel
    expectation (dp_model) in the same line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DPD_FORCE_DIST", "DPD_TEST_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "3", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["metric"].startswith("query-points/sec") and rec["n_gpus"] == 1 and rec["steps"] == 10 and rec["warmup"] == 3
    assert rec["dtype"] == "f32" and rec["vs_baseline"] is None and rec["scaling"] == "weak" and rec["higher_is_better"] is True
    assert abs(rec["value"] - 2 * 32 * 64 / (rec["ms_per_step"] * 1e-3)) <= 1e-3 * rec["value"]
    roof = rec["roofline"]
    assert roof["bound"] == "mfma" and 0.5 < roof["frac"] <= 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    # roofline.traffic is measured IN this run (two rocprofv3 --pmc child passes): at least the algorithmic bytes, at most ten times them
    assert roof["traffic_note"].startswith("MEASURED in this run"), roof["traffic_note"]
    assert roof["algorithmic_bytes_per_launch"] <= roof["traffic"] <= 10 * roof["algorithmic_bytes_per_launch"], roof
    fo = rec["fwd_only"]
    assert fo["batch"] == 32 and fo["dtype"] == "f32" and 0 < fo["ms_per_eval"] < rec["ms_per_step"] and 0.3 < fo["gemm_frac_of_peak"] <= 1.0
    assert abs(fo["value"] - 2 * 32 * 64 / (fo["ms_per_eval"] * 1e-3)) <= 1e-3 * fo["value"]
    asl = rec["as_loss"]
    for key in ("b16_f32", "b16_bf16", "b32_f32", "b32_bf16"):
        leg = asl[key]
        assert 0 < leg["fwd_only"]["ms_per_eval"] < leg["fwd_bwd"]["ms_per_eval"], key
        assert 0 < leg["fwd_bwd"]["gemm_share_of_eval"] <= 1.0 and leg["fwd_bwd"]["gemm_launches"] == 6, (key, leg)
    assert set(rec["dp_model"]["per_world"]) == {"2", "4", "8"}
    assert rec["config3"]["dtype"] == "bf16" and rec["config3"]["pairs_per_gpu"] == 64 and "dp_model" in rec["config3"]


def _torchrun_shared_gpu(args, timeout=1500, env=None):
    import socket
    import subprocess
    import sys
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    so.close()
    env = dict(os.environ, DPD_TEST_SHARE_GPU="1", **(env or {}))
    for k in ("DPD_BENCH_CHILD", "DPD_DP_BACKEND", "DPD_DP_MODE", "DPD_DP_SCHEDULE", "DPD_FORCE_DIST", "RANK", "WORLD_SIZE", "LOCAL_RANK",
              "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(port)] + args
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.gpu
def test_trainer_loop_two_ranks_share_the_gpu(dev, tmp_path):
    """python -m dpdist_amd.train under torch.distributed.run with two ranks on one GPU (gloo): global batch 16 split 8 + 8, two epochs,
    eval, rank 0's checkpoint.  --dp_schedule auto: order x communication form chosen by both ranks on the first batch and stored in
    dp_schedule.json; the resumed run (--restore) re-uses the stored choice instead of measuring again; without the flag the run is pinned
    to the deterministic default ("early") and nothing is measured."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    log = str(tmp_path / "log")
    base = ["-m", "dpdist_amd.train", "--batch_size", "16", "--train_shapes", "48", "--test_shapes", "16", "--eval_every", "1"]
    r = _torchrun_shared_gpu(base + ["--max_epoch", "2", "--log_dir", log, "--dp_schedule", "auto"], timeout=900, env={"DPD_DP_SELECT": "2,1,2"})
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "data-parallel schedule" in r.stdout and "candidates_ms" in r.stdout
    assert any(f.startswith("model.ckpt") for f in os.listdir(log)), os.listdir(log)
    stored = json.load(open(os.path.join(log, "dp_schedule.json")))
    assert stored["schedule"] in ("early", "late") and stored["mode"] in ("allreduce", "rs_ag", "zero1") and len(stored["candidates_ms"]) == 9
    r2 = _torchrun_shared_gpu(base + ["--max_epoch", "1", "--log_dir", str(tmp_path / "log2"), "--dp_schedule", "auto", "--restore",
                                      os.path.join(log, "model.ckpt.npz")], timeout=900)
    assert r2.returncode == 0, (r2.stdout[-1500:], r2.stderr[-3000:])
    assert "re-used from" in r2.stdout and "candidates_ms" in r2.stdout and "measured at start-up" in r2.stdout      # (the stored record, not a new table)
    assert not os.path.exists(str(tmp_path / "log2" / "dp_schedule.json"))
    r3 = _torchrun_shared_gpu(base + ["--max_epoch", "1", "--log_dir", str(tmp_path / "log3")], timeout=900)
    assert r3.returncode == 0, (r3.stdout[-1500:], r3.stderr[-3000:])
    assert "--dp_schedule (pinned)" in r3.stdout and "candidates_ms" not in r3.stdout


@pytest.mark.gpu
def test_registration_demo_two_ranks_share_the_gpu(dev):
    """BASELINE config 5's data-parallel leg end to end with two ranks on one GPU (gloo): DPDist trained identically on both ranks, frozen,
    the pose network's gradient all-reduced (nothing of DPDist travels), held-out pairs split over the ranks, replicas bit-identical."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = _torchrun_shared_gpu([os.path.join(root, "tools", "registration_demo.py"), "--gpus", "2", "--loss", "ours", "--dp_steps", "60", "--dp_pool", "8",
                              "--reg_steps", "12", "--eval_pairs", "16", "--batch", "4"], timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["workload"]["global_batch"] == 8
    d = rec["pcrnet_ours"]["dp"]
    assert d["nranks"] == 2 and d["replicas_bit_identical"] is True and d["dpdist_collectives"] == 0 and d["crosscheck"]["ok"]
    assert rec["pcrnet_ours"]["pairs"] == 16


@pytest.mark.gpu
@pytest.mark.parametrize("B,N", [(1, 64), (5, 32), (2, 200), (7, 8), (3, 100)])
@pytest.mark.parametrize("dt", ["f32", "bf16", "f32x3"])
def test_as_loss_engine_on_ragged_shapes(dev, dt, B, N, monkeypatch):
    """The engine against the entry-by-entry node on the shapes the reference's consumers may bring (config 1's single pair, ragged point
    counts, row counts that are not plane-shaped -- the engine then carves the exact type's buffers and the GEMMs convert per call):
    loss and both input gradients bit for bit, gradients finite."""
    from dpdist_amd import model as M
    pcA, pcB, _ = synth.s2_modelnet_shaped(B, N, 11)
    res = {}
    for eng in ("0", "1"):
        monkeypatch.setattr("dpdist_amd.asloss.ENGINE", eng == "1")
        mod = _model(dev, "wide")
        mod.params_.compute_dtype = dt
        a, b = _cu(pcA, dev).requires_grad_(True), _cu(pcB, dev).requires_grad_(True)
        loss = M.DPDistLoss(mod)(a, b)
        ga, gb = torch.autograd.grad(loss * 3.0, [a, b])
        res[eng] = (loss.detach().clone(), ga, gb)
        if eng == "1":
            assert getattr(mod.params_, "_asloss_engines", None), "the engine did not take this shape"
    for x, y in zip(res["0"], res["1"]):
        assert torch.equal(x, y), (x - y).abs().max().item()
    assert torch.isfinite(res["1"][1]).all() and torch.isfinite(res["1"][2]).all()
