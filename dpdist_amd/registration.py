"""Iterative PCRNet registration with DPDist as the (frozen) loss -- SURVEY section 8 row f2, the second consumer of
the hot-path boundary.  Only the LOSS runs on the HIP path; the pose network is the consumer's own small PyTorch model.

Restates (relative to /root/reference/pcrnet-registration):
    models/ipcr_model.py:198-233   pointnet: shared MLP 64-64-64-128-1024 (1xW convs, no BN) + max pool, source and
                                   template stacked along the batch axis
    models/ipcr_model.py:273-294   get_pose: fc 1024-512-256, dropout(keep 0.7), fc 7;  quat_normalize (lim_rot)
    helper.py:539-570              transformation_quat_tensor: Besl-McKay quaternion -> R, data @ R^T + t
    helper.py:309-329              transformation_quat2mat: compose 4x4 transforms, transform the source cloud
    iterative_PCRNet_ours.py:229-257   loss = (mean(output1[...,0]) + mean(output2[...,0]))/2 of the frozen DPDist graph
                                       with input1 = transformed source, input2 = template; gradients to 'Network' only
    iterative_PCRNet_ours.py:410-470   MAX_LOOPS-1 = 7 forward-only pose refinements, then one training step
    results_itrPCRNet_no_stop.py:112-133   find_errors: translation L2 error, rotation angle of R_pred R_gt^-1
"""
import math

import torch
from torch import nn


def quat_to_mat(q):
    """[B,4] (q0=w,q1,q2,q3) -> [B,3,3], helper.py:552-554 (no normalisation inside, like the reference)."""
    q0, q1, q2, q3 = q.unbind(-1)
    R = torch.stack([
        q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3, 2 * (q1 * q2 - q0 * q3), 2 * (q1 * q3 + q0 * q2),
        2 * (q1 * q2 + q0 * q3), q0 * q0 + q2 * q2 - q1 * q1 - q3 * q3, 2 * (q2 * q3 - q0 * q1),
        2 * (q1 * q3 - q0 * q2), 2 * (q2 * q3 + q0 * q1), q0 * q0 + q3 * q3 - q1 * q1 - q2 * q2], -1)
    return R.view(-1, 3, 3)


def transformation_quat_tensor(data, quat, translation):
    """helper.py:539-570: rotate every cloud by its quaternion and add the translation.  data [B,N,3]."""
    return data @ quat_to_mat(quat).transpose(1, 2) + translation[:, None, :]


def quat_normalize(pred, rot_lim=45.0):
    """models/ipcr_model.py:285-294: (t, angle, axis) -> (tanh(t)*0.1, cos(a/2), axis*sin(a/2)), |a| <= rot_lim deg."""
    t, ang, axis = pred[:, :3], pred[:, 3:4], pred[:, 4:7]
    ang = torch.tanh(ang) * (math.pi / 180.0 * rot_lim)
    axis = axis / (axis.norm(dim=-1, keepdim=True) + 1e-6)
    return torch.cat([torch.tanh(t) * 0.1, torch.cos(ang / 2), axis * torch.sin(ang / 2)], -1)


class PoseNet(nn.Module):
    """models/ipcr_model.py:198-233 + :273-284 (1xW convs == per-point linear layers)."""

    def __init__(self, out_features=1024, lim_rot=45.0):
        super().__init__()
        dims = [3, 64, 64, 64, 128, out_features]
        self.point = nn.Sequential(*[m for i in range(5) for m in (nn.Linear(dims[i], dims[i + 1]), nn.ReLU())])
        self.head = nn.Sequential(nn.Linear(2 * out_features, 1024), nn.ReLU(), nn.Linear(1024, 512), nn.ReLU(),
                                  nn.Linear(512, 256), nn.ReLU(), nn.Dropout(p=0.3), nn.Linear(256, 7))
        self.lim_rot = lim_rot

    def forward(self, source, template):
        f = self.point(torch.cat([source, template], 0)).amax(1)            # max pool over the points
        B = source.shape[0]
        pred = self.head(torch.cat([f[:B], f[B:]], 1))
        return quat_normalize(pred, self.lim_rot) if self.lim_rot else pred


def compose(T, pose):
    """helper.py:309-329: T <- [R(q) t; 0 1] @ T.  pose [B,7] = (t, q)."""
    M = torch.zeros_like(T)
    M[:, :3, :3] = quat_to_mat(pose[:, 3:7])
    M[:, :3, 3] = pose[:, :3]
    M[:, 3, 3] = 1
    return M @ T


def find_errors(T_pred, R_gt, t_gt):
    """results_itrPCRNet_no_stop.py:112-133 on matrices: translation L2 error and the rotation angle (deg) of
    R_pred R_gt^-1, for T_pred mapping the source onto the template and (R_gt, t_gt) the pose that created the source."""
    # template = R_gt^-1 (source - t_gt)  ->  ideal prediction R = R_gt^T, t = -R_gt^T t_gt
    R_id = R_gt.transpose(1, 2)
    t_id = -(R_id @ t_gt[:, :, None])[:, :, 0]
    E = T_pred[:, :3, :3] @ R_id.transpose(1, 2)
    cos = ((E.diagonal(dim1=1, dim2=2).sum(-1) - 1) / 2).clamp(-1, 1)
    return (T_pred[:, :3, 3] - t_id).norm(dim=-1), torch.rad2deg(torch.acos(cos))


class IterativeRegistration:
    """One training step = iterative_PCRNet_ours.py:410-470: 7 forward-only refinements (no gradient), then one step
    in which the DPDist loss of (transformed source, template) is back-propagated THROUGH the frozen DPDist path into
    the pose network."""

    def __init__(self, pose_net, dpdist_loss, lr=1e-3, max_loops=8):
        self.net, self.loss_fn, self.max_loops = pose_net, dpdist_loss, max_loops
        self.opt = torch.optim.Adam(pose_net.parameters(), lr=lr)

    def refine(self, source, template, loops):
        T = torch.eye(4, device=source.device).repeat(source.shape[0], 1, 1)
        with torch.no_grad():
            for _ in range(loops):
                pose = self.net(source, template)
                source = transformation_quat_tensor(source, pose[:, 3:7], pose[:, :3])
                T = compose(T, pose)
        return source, T

    def train_step(self, source, template):
        self.net.train()
        src, T = self.refine(source, template, self.max_loops - 1)
        pose = self.net(src, template)
        moved = transformation_quat_tensor(src, pose[:, 3:7], pose[:, :3])
        loss = self.loss_fn(moved, template)                 # (mean(AB[...,0]) + mean(BA[...,0])) / 2, :248-251
        self.opt.zero_grad(set_to_none=True)
        loss.backward()                                      # d loss / d moved comes from the HIP backward-to-input path
        self.opt.step()
        return loss.detach(), compose(T, pose.detach())

    @torch.no_grad()
    def evaluate(self, source, template):
        self.net.eval()
        moved, T = self.refine(source, template, self.max_loops)
        return self.loss_fn(moved, template), T
