"""Row f2 (registration harness): pose math against scipy (CPU); GPU: gradients reach the pose network through DPDist."""
import math

import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

from dpdist_amd.registration import (PoseNet, compose, find_errors, quat_normalize, quat_to_mat,
                                     transformation_quat_tensor)


def test_quat_to_mat_matches_scipy():
    rng = np.random.default_rng(0)
    q = rng.standard_normal((16, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    R = quat_to_mat(torch.tensor(q)).numpy()
    ref = Rotation.from_quat(q[:, [1, 2, 3, 0]]).as_matrix()      # scipy is (x,y,z,w); the reference is (q0=w,q1,q2,q3)
    assert np.abs(R - ref).max() < 1e-12


def test_transformation_and_compose_are_consistent():
    rng = np.random.default_rng(1)
    data = torch.tensor(rng.standard_normal((4, 10, 3)))
    T = torch.eye(4, dtype=torch.float64).repeat(4, 1, 1)
    cur = data
    for _ in range(3):
        pose = quat_normalize(torch.tensor(rng.standard_normal((4, 7))))
        cur = transformation_quat_tensor(cur, pose[:, 3:7], pose[:, :3])
        T = compose(T, pose)
    via_T = data @ T[:, :3, :3].transpose(1, 2) + T[:, None, :3, 3]
    assert (cur - via_T).abs().max() < 1e-12


def test_quat_normalize_limits():
    p = quat_normalize(torch.randn(64, 7) * 10, rot_lim=45.0)
    assert (p[:, :3].abs() <= 0.1 + 1e-6).all()
    assert ((p[:, 3:7].norm(dim=-1) - 1).abs() < 1e-5).all()
    ang = 2 * torch.rad2deg(torch.acos(p[:, 3].clamp(-1, 1)))
    assert (ang <= 45.0 + 1e-3).all()


def test_find_errors():
    ang = math.radians(30.0)
    Rg = torch.tensor(Rotation.from_rotvec([0, 0, ang]).as_matrix())[None]
    tg = torch.tensor([[0.05, -0.02, 0.01]], dtype=torch.float64)
    T_perfect = torch.eye(4, dtype=torch.float64)[None].clone()
    T_perfect[:, :3, :3] = Rg.transpose(1, 2)
    T_perfect[:, :3, 3] = -(Rg.transpose(1, 2) @ tg[:, :, None])[:, :, 0]
    te, re = find_errors(T_perfect, Rg, tg)
    assert te.item() < 1e-12 and re.item() < 1e-5
    te, re = find_errors(torch.eye(4, dtype=torch.float64)[None], Rg, tg)
    assert abs(re.item() - 30.0) < 1e-6


@pytest.mark.gpu
def test_dpdist_loss_trains_the_pose_network():
    """Gradients flow source -> transformed source -> (HIP) DPDist backward-to-input -> pose network."""
    from dpdist_amd import synth
    from dpdist_amd.model import DPDistLoss, DPDistModel
    from dpdist_amd.registration import IterativeRegistration
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = DPDistModel(device=dev)
    model.load_tf_state_dict(synth.make_weights("wide"))
    reg = IterativeRegistration(PoseNet().to(dev), DPDistLoss(model), lr=1e-4, max_loops=3)
    pcA, pcB, _ = synth.s2_modelnet_shaped(8, 64, 100)
    src, tmpl = torch.tensor(pcA, device=dev), torch.tensor(pcB, device=dev)
    before = [p.detach().clone() for p in reg.net.parameters()]
    loss, T = reg.train_step(src, tmpl)
    assert torch.isfinite(loss) and T.shape == (8, 4, 4)
    changed = sum(float((p.detach() - b).abs().max()) > 0 for p, b in zip(reg.net.parameters(), before))
    assert changed >= len(before) - 1          # every layer received a gradient
    assert all(not p.requires_grad for p in model.parameters())      # DPDist stays frozen
