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

import numpy as np
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


class _PointFeaturesFn(torch.autograd.Function):
    """Shared MLP + max pool of the pose network (models/ipcr_model.py:198-233) with its weight gradients on the library
    (include/dpdist_capi.h: dpd_pose_point_fwd_train / dpd_pose_point_bwd; csrc/pose.hip): one launch forward, three backward, instead of
    ~16 forward and ~45 backward launches of torch / hipBLASLt kernels at batch 16.  The clouds carry no gradient (the refined source of a
    registration step is a constant of the step, iterative_PCRNet_ours.py:442-470).  Same mathematics as `point(...).amax(1)` -- gradient of
    the max pool shared evenly among ties like tf.reduce_max -- in another fp32 summation order."""

    @staticmethod
    def forward(ctx, clouds, *wb):
        from ctypes import byref
        from . import lib as L
        C, N, _ = clouds.shape
        L.req(clouds, name="clouds", shape=(C, N, 3))
        w = L.PoseNetW()
        for i in range(5):
            w.Wp[i], w.bp[i] = L.req(wb[2 * i], name="weight").data_ptr(), L.req(wb[2 * i + 1], name="bias").data_ptr()
        w.out_features = wb[8].shape[0]
        dev = clouds.device
        f = torch.empty(C, w.out_features, device=dev, dtype=torch.float32)
        h = [torch.empty(C * N, k, device=dev, dtype=torch.float32) for k in (64, 64, 64, 128)]
        ties = torch.empty(C, w.out_features, device=dev, dtype=torch.int64)
        L.check(L.load().dpd_pose_point_fwd_train(byref(w), L.ptr(clouds), None, C, 0, N, L.ptr(f), L.ptr(h[0]), L.ptr(h[1]), L.ptr(h[2]), L.ptr(h[3]),
                                                  ties.data_ptr(), L.cur_stream()), "dpd_pose_point_fwd_train")
        ctx.save_for_backward(clouds, ties, *h, *wb)
        return f

    @staticmethod
    def backward(ctx, df):
        import ctypes
        from ctypes import byref
        from . import lib as L
        clouds, ties, h1, h2, h3, h4 = ctx.saved_tensors[:6]
        wb = ctx.saved_tensors[6:]
        C, N, _ = clouds.shape
        w = L.PoseNetW()
        for i in range(5):
            w.Wp[i], w.bp[i] = wb[2 * i].data_ptr(), wb[2 * i + 1].data_ptr()
        w.out_features = wb[8].shape[0]
        df = df.contiguous()
        grads = [torch.empty_like(t) for t in wb]
        dW = (ctypes.c_void_p * 5)(*[grads[2 * i].data_ptr() for i in range(5)])
        db = (ctypes.c_void_p * 5)(*[grads[2 * i + 1].data_ptr() for i in range(5)])
        lib = L.load()
        nbytes = lib.dpd_pose_point_bwd_workspace_bytes(C)
        ws = torch.empty(nbytes // 4, device=clouds.device, dtype=torch.float32)
        L.check(lib.dpd_pose_point_bwd(byref(w), L.ptr(clouds), None, C, 0, N, L.ptr(df), L.ptr(h1), L.ptr(h2), L.ptr(h3), L.ptr(h4), ties.data_ptr(),
                                       dW, db, L.ptr(ws), nbytes, L.cur_stream()), "dpd_pose_point_bwd")
        return (None,) + tuple(grads)


class _PoseNetRawFn(torch.autograd.Function):
    """The whole pose network of a training evaluation -- shared MLP + max pool (_PointFeaturesFn's kernels) and the head (models/ipcr_model.py:
    273-284) -- as ONE autograd node on the library: five launches forward (dpd_pose_point_fwd_train, dpd_pose_head_fwd_train), eight backward
    (dpd_pose_head_bwd, dpd_pose_point_bwd), gradients w.r.t. the 18 weight tensors only.  `mask` [B,256]: the dropout mask (0 or 1 / keep)
    drawn by the caller, or None (evaluation mode).  `sink`: None, or {id(parameter): tensor of its shape} -- the backward then WRITES the
    gradient of that parameter there (views of the optimizer's flat gradient buffer: no gather copy afterwards) and returns it."""

    @staticmethod
    def forward(ctx, source, template, mask, sink, *wb):
        from ctypes import byref
        from . import lib as L
        B, N, _ = source.shape
        C = 2 * B
        L.req(source, name="source", shape=(B, N, 3)), L.req(template, name="template", shape=(B, N, 3))
        if mask is not None:
            L.req(mask, name="mask", shape=(B, 256))
        w = L.PoseNetW()
        for i in range(5):
            w.Wp[i], w.bp[i] = L.req(wb[2 * i], name="weight").data_ptr(), L.req(wb[2 * i + 1], name="bias").data_ptr()
        for i in range(4):
            w.Wh[i], w.bh[i] = L.req(wb[10 + 2 * i], name="weight").data_ptr(), L.req(wb[11 + 2 * i], name="bias").data_ptr()
        w.out_features = wb[8].shape[0]
        dev, lib, st = source.device, L.load(), L.cur_stream()
        e = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)      # noqa: E731
        f = e(C, w.out_features)
        h = [e(C * N, k) for k in (64, 64, 64, 128)]
        ties = torch.empty(C, w.out_features, device=dev, dtype=torch.int64)
        L.check(lib.dpd_pose_point_fwd_train(byref(w), L.ptr(source), L.ptr(template), B, B, N, L.ptr(f), L.ptr(h[0]), L.ptr(h[1]), L.ptr(h[2]),
                                             L.ptr(h[3]), ties.data_ptr(), st), "dpd_pose_point_fwd_train")
        a1, a2, a3, pred = e(B, 1024), e(B, 512), e(B, 256), e(B, 7)
        L.check(lib.dpd_pose_head_fwd_train(byref(w), L.ptr(f), B, L.ptr(mask), L.ptr(a1), L.ptr(a2), L.ptr(a3), L.ptr(pred), st),
                "dpd_pose_head_fwd_train")
        ctx.has_mask = mask is not None
        ctx.sink = sink
        ctx.save_for_backward(source, template, ties, f, a1, a2, a3, *h, *wb, *([mask] if mask is not None else []))
        return pred

    @staticmethod
    def backward(ctx, dpred):
        import ctypes
        from ctypes import byref
        from . import lib as L
        sv = ctx.saved_tensors
        source, template, ties, f, a1, a2, a3, h1, h2, h3, h4 = sv[:11]
        wb = sv[11:29]
        mask = sv[29] if ctx.has_mask else None
        B, N, _ = source.shape
        C = 2 * B
        w = L.PoseNetW()
        for i in range(5):
            w.Wp[i], w.bp[i] = wb[2 * i].data_ptr(), wb[2 * i + 1].data_ptr()
        for i in range(4):
            w.Wh[i], w.bh[i] = wb[10 + 2 * i].data_ptr(), wb[11 + 2 * i].data_ptr()
        w.out_features = wb[8].shape[0]
        dev, lib, st = source.device, L.load(), L.cur_stream()
        dpred = dpred.contiguous()
        sink = ctx.sink or {}
        grads = [sink.get(id(t)) for t in wb]
        grads = [g if (g is not None and g.shape == t.shape and g.is_contiguous() and g.data_ptr() % 16 == 0) else torch.empty_like(t)
                 for g, t in zip(grads, wb)]
        vp = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])      # noqa: E731
        df = torch.empty_like(f)
        nb = lib.dpd_pose_head_bwd_workspace_bytes(B)
        ws = torch.empty(nb // 4, device=dev, dtype=torch.float32)
        L.check(lib.dpd_pose_head_bwd(byref(w), L.ptr(f), B, L.ptr(mask), L.ptr(a1), L.ptr(a2), L.ptr(a3), L.ptr(dpred), vp(grads[10::2]), vp(grads[11::2]),
                                      L.ptr(df), L.ptr(ws), nb, st), "dpd_pose_head_bwd")
        nb2 = lib.dpd_pose_point_bwd_workspace_bytes(C)
        ws2 = torch.empty(nb2 // 4, device=dev, dtype=torch.float32)
        L.check(lib.dpd_pose_point_bwd(byref(w), L.ptr(source), L.ptr(template), B, B, N, L.ptr(df), L.ptr(h1), L.ptr(h2), L.ptr(h3), L.ptr(h4),
                                       ties.data_ptr(), vp(grads[0:10:2]), vp(grads[1:10:2]), L.ptr(ws2), nb2, st), "dpd_pose_point_bwd")
        return (None, None, None, None) + tuple(grads)


class PoseNet(nn.Module):
    """models/ipcr_model.py:198-233 + :273-284 (1xW convs == per-point linear layers)."""
    next_mask = None         # a dropout mask [B,256] already drawn for the next training evaluation of `raw` (consumed by it)
    grad_sink = None         # set by IterativeRegistration for the span of one training evaluation (see _PoseNetRawFn)
    native_train = True      # the training evaluation's shared MLP + max pool on the library (_PointFeaturesFn) when the shape allows it

    def __init__(self, out_features=1024, lim_rot=45.0, keep_prob=0.7):
        super().__init__()
        dims = [3, 64, 64, 64, 128, out_features]
        self.point = nn.Sequential(*[m for i in range(5) for m in (nn.Linear(dims[i], dims[i + 1]), nn.ReLU())])
        self.head = nn.Sequential(nn.Linear(2 * out_features, 1024), nn.ReLU(), nn.Linear(1024, 512), nn.ReLU(),
                                  nn.Linear(512, 256), nn.ReLU(), nn.Dropout(p=1.0 - keep_prob), nn.Linear(256, 7))
        self.lim_rot = lim_rot
        self.reset_parameters_tf()

    @torch.no_grad()
    def reset_parameters_tf(self):
        """The reference's initialisation (pcrnet-registration/utils/tf_util.py `_variable_with_weight_decay(use_xavier=True)`:
        Xavier-uniform kernels, zero biases) instead of torch's kaiming-uniform(a=sqrt(5)) default with random biases."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)

    @torch.no_grad()
    def load_tf_state_dict(self, sd):
        """TF variables of ipcr_model.pointnet / get_pose: conv{1..5}/{weights [1,kw,cin,cout], biases}, fc{1..4}/{weights
        [in,out], biases} -> the Linear layers (a 1xW VALID conv on [B,N,W,1] / [B,N,1,C] is a per-point linear map)."""
        lin = [m for m in self.point if isinstance(m, nn.Linear)]
        for i, l in enumerate(lin, 1):
            w = torch.as_tensor(np.asarray(sd["conv%d/weights" % i]), dtype=torch.float32)
            l.weight.copy_(w.reshape(-1, w.shape[-1]).t())
            l.bias.copy_(torch.as_tensor(np.asarray(sd["conv%d/biases" % i]), dtype=torch.float32))
        lin = [m for m in self.head if isinstance(m, nn.Linear)]
        for i, l in enumerate(lin, 1):
            l.weight.copy_(torch.as_tensor(np.asarray(sd["fc%d/weights" % i]), dtype=torch.float32).t())
            l.bias.copy_(torch.as_tensor(np.asarray(sd["fc%d/biases" % i]), dtype=torch.float32))

    def _pooled(self, clouds):
        """shared MLP + max pool over the points of every cloud: [C, N, 3] -> [C, out_features]"""
        lin = [m for m in self.point if isinstance(m, nn.Linear)]
        if (self.native_train and torch.is_grad_enabled() and clouds.is_cuda and clouds.dtype == torch.float32 and not clouds.requires_grad
                and clouds.shape[1] <= 64 and [(m.in_features, m.out_features) for m in lin] == [(3, 64), (64, 64), (64, 64), (64, 128), (128, 1024)]
                and any(p.requires_grad for m in lin for p in (m.weight, m.bias))):
            return _PointFeaturesFn.apply(clouds.contiguous(), *[t for m in lin for t in (m.weight, m.bias)])
        return self.point(clouds).amax(1)

    def features(self, source, template):
        """ipcr_model.pointnet (:198-233): (source_global_feature, template_global_feature), each [B, out_features]."""
        f = self._pooled(torch.cat([source, template], 0))
        B = source.shape[0]
        return f[:B], f[B:]

    def _native_raw_ok(self, clouds):
        lin_p = [m for m in self.point if isinstance(m, nn.Linear)]
        lin_h = [m for m in self.head if isinstance(m, nn.Linear)]
        return (self.native_train and torch.is_grad_enabled() and clouds.is_cuda and clouds.dtype == torch.float32 and not clouds.requires_grad
                and clouds.shape[1] <= 64 and [(m.in_features, m.out_features) for m in lin_p] == [(3, 64), (64, 64), (64, 64), (64, 128), (128, 1024)]
                and [(m.in_features, m.out_features) for m in lin_h] == [(2048, 1024), (1024, 512), (512, 256), (256, 7)]
                and any(p.requires_grad for p in self.parameters()))

    def raw(self, source, template):
        """get_pose's fc4 output (:273-284) BEFORE quat_normalize: [B,7] = (t, angle, axis)."""
        if self._native_raw_ok(source) and template.shape == source.shape and template.dtype == source.dtype and not template.requires_grad:
            # the training evaluation on the library (one autograd node): the dropout mask is drawn here (torch's Philox stream, as the
            # refinements' masks are), everything else is csrc/pose.hip
            mask = None
            drop = next((m for m in self.head if isinstance(m, nn.Dropout)), None)
            given, self.next_mask = self.next_mask, None
            if self.training and drop is not None and drop.p > 0:
                keep = 1.0 - drop.p
                if given is not None and given.shape == (source.shape[0], 256) and given.device == source.device:
                    mask = given                                   # drawn with the refinements' masks (IterativeRegistration.refine)
                else:
                    mask = torch.empty(source.shape[0], 256, device=source.device).bernoulli_(keep).div_(keep)
            lin = [m for m in self.point if isinstance(m, nn.Linear)] + [m for m in self.head if isinstance(m, nn.Linear)]
            return _PoseNetRawFn.apply(source.contiguous(), template.contiguous(), mask, getattr(self, "grad_sink", None),
                                       *[t for m in lin for t in (m.weight, m.bias)])
        f = self._pooled(torch.cat([source, template], 0))                  # max pool over the points
        B = source.shape[0]
        return self.head(torch.cat([f[:B], f[B:]], 1))

    def forward(self, source, template):
        pred = self.raw(source, template)
        return quat_normalize(pred, self.lim_rot) if self.lim_rot else pred


def compose(T, pose):
    """helper.py:309-329: T <- [R(q) t; 0 1] @ T.  pose [B,7] = (t, q)."""
    M = torch.zeros_like(T)
    M[:, :3, :3] = quat_to_mat(pose[:, 3:7])
    M[:, :3, 3] = pose[:, :3]
    M[:, 3, 3] = 1
    return M @ T


def euler_to_mat(rx, ry, rz):
    """transforms3d.euler.euler2mat(rz, ry, rx, 'szyx') as the reference calls it (helper.py:301,
    results_itrPCRNet_no_stop.py:121-122): static frame, about z, then y, then x  ->  R = Rx(rx) Ry(ry) Rz(rz)."""
    cx, sx, cy, sy, cz, sz = math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry), math.cos(rz), math.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


def mat_to_euler(R):
    """transforms3d.euler.mat2euler(R, 'szyx') re-ordered like helper.find_final_pose (:331-345): returns (rx, ry, rz)."""
    cy = math.hypot(R[0, 0], R[0, 1])
    if cy > 4 * np.finfo(float).eps:
        return math.atan2(-R[1, 2], R[2, 2]), math.atan2(R[0, 2], cy), math.atan2(-R[0, 1], R[0, 0])
    return math.atan2(R[2, 1], R[1, 1]), math.atan2(R[0, 2], cy), 0.0


def find_errors(gt_pose, final_pose):
    """results_itrPCRNet_no_stop.py:112-133, same signature: poses are 6-vectors (x, y, z, rx, ry, rz), angles in radians.
    Returns (translation L2 error, |angle| in degrees of the axis-angle form of R_pred R_gt^-1)."""
    gt_pose, final_pose = np.asarray(gt_pose, dtype=np.float64), np.asarray(final_pose, dtype=np.float64)
    translation_error = float(np.sqrt(np.sum(np.square(gt_pose[0:3] - final_pose[0:3]))))
    gt_mat = euler_to_mat(gt_pose[3], gt_pose[4], gt_pose[5])
    pt_mat = euler_to_mat(final_pose[3], final_pose[4], final_pose[5])
    error_mat = pt_mat @ np.linalg.inv(gt_mat)
    # transforms3d.axangles.mat2axangle: angle = atan2(sin, cos) with cos = (trace - 1) / 2; its magnitude is acos(cos)
    angle = math.acos(min(1.0, max(-1.0, (np.trace(error_mat) - 1.0) / 2.0)))
    return translation_error, abs(angle * (180 / np.pi))


def transformation_quat2mat(poses, TRANSFORMATIONS, templates_data):
    """helper.py:309-329 (numpy, same argument order): compose the 4x4 transforms with the predicted (t, quaternion) poses
    and move the clouds.  poses [..., B, 7]; TRANSFORMATIONS [B,4,4] and templates_data [B,N,3] are updated in place."""
    poses = np.array(poses)
    poses = poses.reshape(poses.shape[-2], poses.shape[-1])
    for i in range(poses.shape[0]):
        q = poses[i, 3:7] / max(np.linalg.norm(poses[i, 3:7]), np.finfo(float).eps)      # transforms3d.quat2mat normalises
        rot = quat_to_mat(torch.tensor(q[None], dtype=torch.float64))[0].numpy()
        M = np.zeros((4, 4))
        M[3, 3] = 1
        M[0:3, 0:3] = rot
        M[0:3, 3] = poses[i, 0:3]
        TRANSFORMATIONS[i] = M @ TRANSFORMATIONS[i]
        templates_data[i] = (rot @ templates_data[i].T).T + poses[i, 0:3]
    return TRANSFORMATIONS, templates_data


def find_final_pose_inv(TRANSFORMATIONS):
    """helper.py:347-361: the pose of the INVERSE transforms -- what results_itrPCRNet_no_stop.get_error (:465-474) compares
    with the ground-truth pose that created the source from the template."""
    return find_final_pose(np.linalg.inv(np.asarray(TRANSFORMATIONS, dtype=np.float64)))


def find_final_pose(TRANSFORMATIONS):
    """helper.py:331-345: 4x4 transforms -> (x, y, z, rx, ry, rz)."""
    out = np.zeros((TRANSFORMATIONS.shape[0], 6))
    for i in range(TRANSFORMATIONS.shape[0]):
        out[i, 3:6] = mat_to_euler(TRANSFORMATIONS[i, 0:3, 0:3])
        out[i, 0:3] = TRANSFORMATIONS[i, 0:3, 3]
    return out


def pose_errors(T_pred, R_gt, t_gt):
    """find_errors on matrices (batched torch): translation L2 error and the rotation angle (deg) of
    R_pred R_gt^-1, for T_pred mapping the source onto the template and (R_gt, t_gt) the pose that created the source."""
    # template = R_gt^-1 (source - t_gt)  ->  ideal prediction R = R_gt^T, t = -R_gt^T t_gt
    R_id = R_gt.transpose(1, 2)
    t_id = -(R_id @ t_gt[:, :, None])[:, :, 0]
    E = T_pred[:, :3, :3] @ R_id.transpose(1, 2)
    cos = ((E.diagonal(dim1=1, dim2=2).sum(-1) - 1) / 2).clamp(-1, 1)
    return (T_pred[:, :3, 3] - t_id).norm(dim=-1), torch.rad2deg(torch.acos(cos))


def predicted_pose_applied(source, pose):
    """iterative_PCRNet_ours.py:211-224: split the 7-vector, re-normalise the quaternion (norm + 1e-7) and move the source."""
    quat = pose[:, 3:7]
    quat = quat / (quat.square().sum(1, keepdim=True).sqrt() + 1e-7)
    return transformation_quat_tensor(source, quat, pose[:, :3])


def centroid_residual(T_pred, gt_pose, source):
    """Where the registration leaves the SOURCE'S CENTROID, against where the ground-truth pose puts it: |T_pred c - T_ideal c| per
    pair (T_ideal = the inverse of the pose that created the source).  The reference's translation metric compares pose vectors
    (results_itrPCRNet_no_stop.py:112-133); with `centroid_sub=0` the clouds are not centred, so a rotation error of d radians about
    the ORIGIN has to be compensated by a translation of ~d x |centroid| for the clouds to overlap -- the pose-space translation
    error then mostly measures the rotation error times that lever arm (0.035 rad x 0.3 = 0.01 at 2 deg), not how far apart the
    registered clouds are.  This is the number that says the latter.  T_pred [B,4,4]; gt_pose [B,6] (t, rx, ry, rz); source [B,N,3]."""
    gt = np.asarray(gt_pose, dtype=np.float64)
    Tp = np.asarray(T_pred, dtype=np.float64)
    c = np.asarray(source, dtype=np.float64).mean(1)
    out = np.zeros(len(gt))
    for i in range(len(gt)):
        R = euler_to_mat(gt[i, 3], gt[i, 4], gt[i, 5])
        ideal = R.T @ (c[i] - gt[i, :3])                        # source = R x + t  ->  x = R^T (source - t)
        out[i] = np.linalg.norm(Tp[i, :3, :3] @ c[i] + Tp[i, :3, 3] - ideal)
    return out


class _PoseApplyFn(torch.autograd.Function):
    """quat_normalize -> quaternion normalisation -> Besl-McKay R -> moved cloud (-> T composition) as ONE launch of csrc/pose.hip per
    direction instead of ~115 element-wise torch launches (`quat_normalize`, `predicted_pose_applied`, `transformation_quat_tensor`,
    `compose` above are the same algebra in torch and stay the CPU / reference form; tests/test_registration.py compares the two).
    Gradient: d moved -> d pred only (mode 1); the source cloud comes out of the forward-only refinements (:414-441)."""

    @staticmethod
    def forward(ctx, pred, source, T, lim_rot, mode):
        from . import lib as L
        B, N, _ = source.shape
        L.req(pred, name="pred", shape=(B, 7)), L.req(source, name="source", shape=(B, N, 3))
        pose, moved = torch.empty_like(pred), torch.empty_like(source)
        T_out = None
        if T is not None:
            L.req(T, name="T", shape=(B, 4, 4))
            T_out = torch.empty_like(T)
        L.check(L.load().dpd_pose_apply_fwd(L.ptr(pred), L.ptr(source), L.ptr(T), B, N, float(lim_rot or 0.0), int(mode), L.ptr(pose),
                                            L.ptr(moved), L.ptr(T_out), L.cur_stream()), "dpd_pose_apply_fwd")
        ctx.save_for_backward(pred, source)
        ctx.lim_rot, ctx.mode = float(lim_rot or 0.0), int(mode)
        ctx.set_materialize_grads(False)           # (pose / T_out carry no gradient: no zero tensors made for them in the backward)
        ctx.mark_non_differentiable(pose)
        if T_out is None:
            return pose, moved
        ctx.mark_non_differentiable(T_out)
        return pose, moved, T_out

    @staticmethod
    def backward(ctx, *grads):
        from . import lib as L
        if ctx.mode != 1:
            raise RuntimeError("pose_apply: only the training evaluation (mode 1) is differentiable; the refinements are forward-only")
        pred, source = ctx.saved_tensors
        B, N, _ = source.shape
        if grads[1] is None:
            return None, None, None, None, None
        dmoved = grads[1].contiguous()
        dpred = torch.empty_like(pred)
        L.check(L.load().dpd_pose_apply_bwd(L.ptr(pred), L.ptr(source), L.ptr(dmoved), B, N, ctx.lim_rot, L.ptr(dpred), L.cur_stream()),
                "dpd_pose_apply_bwd")
        return dpred, None, None, None, None


def pose_apply(pred, source, T=None, lim_rot=45.0, mode=0):
    """(pose, moved[, T_out]) of the pose network's raw output on the GPU library (include/dpdist_capi.h: dpd_pose_apply_fwd):
    mode 0 = one forward-only refinement (helper.py:309-329), mode 1 = the training evaluation (iterative_PCRNet_ours.py:211-224)."""
    return _PoseApplyFn.apply(pred.contiguous(), source.contiguous(), None if T is None else T.contiguous(), lim_rot, mode)


def native_refine_supported(net):
    """Whether `net` is the architecture csrc/pose.hip's forward implements (models/ipcr_model.py:198-233,273-284 as PoseNet builds it)."""
    if not isinstance(net, PoseNet):
        return False
    lin_p = [m for m in net.point if isinstance(m, nn.Linear)]
    lin_h = [m for m in net.head if isinstance(m, nn.Linear)]
    if len(lin_p) != 5 or len(lin_h) != 4:
        return False
    out = lin_p[4].out_features
    want_p = [(3, 64), (64, 64), (64, 64), (64, 128), (128, out)]
    want_h = [(2 * out, 1024), (1024, 512), (512, 256), (256, 7)]
    return (out == 1024 and [(m.in_features, m.out_features) for m in lin_p] == want_p
            and [(m.in_features, m.out_features) for m in lin_h] == want_h and all(m.bias is not None for m in lin_p + lin_h))


def pose_refine_native(net, source, template, loops, drop_mask=None, want_pred=False):
    """`loops` forward-only refinements (iterative_PCRNet_ours.py:414-441) with the pose network, quat_normalize, the cloud move and the T
    composition all on the library (include/dpdist_capi.h: dpd_pose_refine, four launches per loop + one per call): (moved source, T[, raw outputs
    [loops,B,7]]).  drop_mask [loops,B,256] (0 or 1/keep) or None; the caller draws it (IterativeRegistration.refine does, in train mode)."""
    from ctypes import byref
    from . import lib as L
    B, N, _ = source.shape
    L.req(source, name="source", shape=(B, N, 3)), L.req(template, name="template", shape=(B, N, 3))
    lin_p = [m for m in net.point if isinstance(m, nn.Linear)]
    lin_h = [m for m in net.head if isinstance(m, nn.Linear)]
    w = L.PoseNetW()
    for i, m in enumerate(lin_p):
        w.Wp[i], w.bp[i] = L.req(m.weight, name="weight").data_ptr(), L.req(m.bias, name="bias").data_ptr()
    for i, m in enumerate(lin_h):
        w.Wh[i], w.bh[i] = L.req(m.weight, name="weight").data_ptr(), L.req(m.bias, name="bias").data_ptr()
    w.out_features = lin_p[4].out_features
    lib = L.load()
    nbytes = lib.dpd_pose_refine_workspace_bytes(B, N, w.out_features)
    ws = torch.empty(nbytes // 4, device=source.device, dtype=torch.float32)
    if drop_mask is not None:
        L.req(drop_mask, name="drop_mask", shape=(loops, B, 256))
    moved, T = torch.empty_like(source), torch.empty(B, 4, 4, device=source.device, dtype=torch.float32)
    pred = torch.empty(loops, B, 7, device=source.device, dtype=torch.float32) if want_pred else None
    L.check(lib.dpd_pose_refine(byref(w), L.ptr(source), L.ptr(template), B, N, int(loops), float(net.lim_rot or 0.0), L.ptr(drop_mask),
                                L.ptr(ws), nbytes, L.ptr(moved), L.ptr(T), L.ptr(pred), L.cur_stream()), "dpd_pose_refine")
    return (moved, T, pred) if want_pred else (moved, T)


def flat_gradient_views(params):
    """One flat fp32 buffer with every parameter's `.grad` as a view into it (what `optim.TFAdam` does for its own parameters): the
    data-parallel step all-reduces that ONE buffer instead of a tensor per layer."""
    params = [p for p in params if p.requires_grad]
    n = sum(p.numel() for p in params)
    flat = torch.zeros((n + 3) // 4 * 4, device=params[0].device, dtype=params[0].dtype)
    off = 0
    for p in params:
        p.grad = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
    return flat


class _GraphRec:
    """One captured registration step (or evaluation) for one batch shape: static inputs / outputs, the hipGraph(s) and the as-loss engines
    whose buffers the graph's launches point into (they must live exactly as long as the graph)."""
    __slots__ = ("src", "tmpl", "loss", "T", "g1", "g2", "pool", "loss_key")


class IterativeRegistration:
    """One training step = iterative_PCRNet_ours.py:410-470: 7 forward-only refinements of the pose (`predicted_transformation` only is
    fetched, :414-441: NO DPDist evaluation there), then one step in which the DPDist loss of (transformed source, template) is
    back-propagated THROUGH the frozen DPDist path into the pose network -- one DPDist forward + backward per step; the optimizer is
    `tf.train.AdamOptimizer(learning_rate, name='Adam2')` (:239) = `optim.TFAdam`.
    `loss_fn(moved_source, template) -> scalar` is DPDistLoss (the reference's 'ours') or any other differentiable
    cloud distance (the reference's Chamfer baseline, iterative_PCRNet.py).

    Where the time goes (round 5: 9.2 ms per step at batch 16 for 0.4 ms of DPDist -- ~1300 tiny eager launches, host-bound) and what
    this class does about it on the GPU:
      * `fused_pose` (default on): quat_normalize / normalisation / Besl-McKay R / moved cloud / T composition are ONE
        launch of csrc/pose.hip per loop (dpd_pose_apply_fwd; backward dpd_pose_apply_bwd) instead of ~115 element-wise launches;
      * `native_refine` (default on): the forward-only refinements run the pose NETWORK on the library too (shared MLP +
        max pool in one launch per loop, the template's features once per call, three head launches; fc4 + pose chain + cloud move as
        the prologue of the next loop's shared MLP: dpd_pose_refine, four launches per loop instead of ~25); the training evaluation is
        one autograd node on the library too (PoseNet.native_train);
      * `concat_grads` (default on, TFAdam only): the pose network's 18 gradients are written into the optimizer's flat buffer by ONE
        concatenation instead of 18 in-place accumulations into a zeroed buffer (bit for bit the same update);
      * `graph` (default on; needs optim.TFAdam and a loss with `capturable = True`, e.g. DPDistLoss): after
        `graph_warmup` eager steps for a batch shape the WHOLE step -- 7 refinements, the training forward, DPDist forward + backward on
        a private as-loss engine, the pose network's backward, TF-form Adam with lr_t read from device memory -- is captured once as a
        hipGraph and replayed.  Same kernels in the same order on the same inputs (dropout draws from the same Philox offsets torch's
        eager launches would use): the training is bit for bit the eager one (tests/test_registration.py).

    Data parallel (BASELINE config 5, "8 x MI355X DP"; the reference itself is single-GPU here, iterative_PCRNet_ours.py:196): one
    process per GPU, every rank registers its own pairs; DPDist is FROZEN, so there is no DPDist collective at all --
    the one exchange step is the sum all-reduce of the pose network's flat gradient (0.9 M parameters = 3.7 MB) through the same
    reducer as the DPDist trainer (ddp.make_reducer: RCCL driven directly, start-up cross-check, torch.distributed fallback), then
    every rank applies the same averaged gradient: replicas stay bit-identical.  `distributed=None` follows the process group.  With a
    reducer the step is two graphs (refine + forward + backward | Adam) around the eager collective."""

    def __init__(self, pose_net, dpdist_loss, lr=1e-4, max_loops=8, optimizer=None, distributed=None, group=None, graph=True,
                 fused_pose=True, graph_warmup=2, native_refine=True, concat_grads=True):
        import os
        import torch.distributed as dist
        from .optim import TFAdam
        self.net, self.loss_fn, self.max_loops = pose_net, dpdist_loss, max_loops
        self.opt = optimizer if optimizer is not None else TFAdam(pose_net.parameters(), lr=lr)
        use_dist = (dist.is_available() and dist.is_initialized()) if distributed is None else bool(distributed)
        self.reducer = None
        self._flat_grad = getattr(self.opt, "grad", None) if isinstance(getattr(self.opt, "grad", None), torch.Tensor) else None
        if use_dist:
            from .ddp import make_reducer
            if self._flat_grad is None:
                self._flat_grad = flat_gradient_views(pose_net.parameters())
            self.reducer = make_reducer(self._flat_grad, [0, self._flat_grad.numel()], group,
                                        force=os.environ.get("DPD_FORCE_DIST") == "1", mode="allreduce")
        self.concat_grads = bool(concat_grads)
        dev0 = next(pose_net.parameters()).device
        self._one = torch.ones((), device=dev0) if dev0.type == "cuda" else None      # d loss / d loss of every step
        self.fused_pose = bool(fused_pose)
        self.native_refine = bool(native_refine) and self.fused_pose and native_refine_supported(pose_net)
        self.use_graph = bool(graph) and isinstance(self.opt, TFAdam) and bool(getattr(dpdist_loss, "capturable", False))
        self.graph_warmup = int(graph_warmup)
        self._graphs, self._graph_seen = {}, {}
        self.graph_replays = 0

    def close(self):
        red, self.reducer = self.reducer, None
        if red is not None:
            red.close()
        self._graphs.clear()

    # ---- the step, piece by piece (eager; the graph captures exactly these calls)
    def _fused(self, source):
        return self.fused_pose and source.is_cuda and hasattr(self.net, "raw")

    def refine(self, source, template, loops, train_mask=False):
        """`loops` forward-only refinements.  train_mask (a training step on the library's pose network): the dropout masks of the refinements
        and of the training evaluation that follows are ONE draw of [loops + 1, B, 256]; the last one waits in `net.next_mask` for `raw`."""
        if self.native_refine and source.is_cuda and loops > 0 and source.dtype == torch.float32:
            with torch.no_grad():
                mask = None
                drop = next((m for m in self.net.head if isinstance(m, nn.Dropout)), None)
                if self.net.training and drop is not None and drop.p > 0:      # torch's dropout: keep with probability 1 - p, scale by 1 / (1 - p)
                    keep = 1.0 - drop.p
                    extra = 1 if (train_mask and hasattr(self.net, "next_mask")) else 0
                    mask = torch.empty(loops + extra, source.shape[0], 256, device=source.device).bernoulli_(keep).div_(keep)
                    if extra:
                        self.net.next_mask, mask = mask[loops], mask[:loops]
                return pose_refine_native(self.net, source.contiguous(), template.contiguous(), loops, mask)
        T = torch.eye(4, device=source.device).repeat(source.shape[0], 1, 1)
        fused = self._fused(source)
        with torch.no_grad():
            for _ in range(loops):
                if fused:                # one launch: quat_normalize, normalisation, R, moved cloud, T <- M T (csrc/pose.hip)
                    _, source, T = pose_apply(self.net.raw(source, template), source, T, self.net.lim_rot, 0)
                    continue
                pose = self.net(source, template)
                # helper.transformation_quat2mat (helper.py:309-329) normalises the quaternion (transforms3d.quat2mat)
                pose = torch.cat([pose[:, :3], pose[:, 3:7] / pose[:, 3:7].norm(dim=1, keepdim=True).clamp_min(1e-12)], 1)
                source = transformation_quat_tensor(source, pose[:, 3:7], pose[:, :3])
                T = compose(T, pose)
        return source, T

    def _rebind_flat_gradient(self):
        if self._flat_grad is not None and not hasattr(self.opt, "grad"):
            self._flat_grad.zero_()                          # (a foreign optimizer's zero_grad may have dropped the views: rebind)
            if any(p.grad is None for p in self.net.parameters() if p.requires_grad):
                off = 0
                for p in (q for q in self.net.parameters() if q.requires_grad):
                    p.grad = self._flat_grad[off:off + p.numel()].view_as(p)
                    off += p.numel()

    def _evaluate_grad(self, refined_source, template, T=None):
        """forward + backward of the training evaluation (:468 without the collective and the update): (loss, pose, T' or None);
        this rank's gradients are left in the parameters' `.grad`."""
        Tn = None
        flat, params = getattr(self.opt, "grad", None), getattr(self.opt, "_params", None)
        in_place = (self.concat_grads and isinstance(flat, torch.Tensor) and params is not None
                    and all(p.grad is not None and p.grad.data_ptr() == flat.data_ptr() + 4 * o for p, o in zip(params, self._offsets(params))))
        if in_place and hasattr(self.net, "grad_sink"):
            # for THIS evaluation only (the node keeps the mapping it was built with): its gradients are taken by torch.autograd.grad below,
            # never accumulated into the very views they were written to
            self.net.grad_sink = {id(p): p.grad for p in params}
        try:
            return self._evaluate_grad_body(refined_source, template, T, flat, params, in_place)
        finally:
            if hasattr(self.net, "grad_sink"):
                self.net.grad_sink = None

    def _evaluate_grad_body(self, refined_source, template, T, flat, params, in_place):
        Tn = None
        if self._fused(refined_source):
            pred = self.net.raw(refined_source, template)
            out = pose_apply(pred, refined_source, T, self.net.lim_rot, 1)
            pose, moved = out[0], out[1]
            Tn = out[2] if T is not None else None
        else:
            pose = self.net(refined_source, template)
            moved = predicted_pose_applied(refined_source, pose)
            if T is not None:
                with torch.no_grad():
                    pn = torch.cat([pose[:, :3], pose[:, 3:7] / pose[:, 3:7].norm(dim=1, keepdim=True).clamp_min(1e-12)], 1)
                    Tn = compose(T, pn)
        loss = self.loss_fn(moved, template)                 # (mean(AB[...,0]) + mean(BA[...,0])) / 2, :248-251
        if in_place:
            # TFAdam's flat gradient: the 18 gradients are WRITTEN into it by one concatenation instead of being accumulated into 18 views
            # of a zeroed buffer (19 in-place adds and a fill per step: 70 us of the captured step at batch 16).  0 + g == g: the same update
            # The pose network's node writes them straight into the views (grad_sink): no copy at all; the concatenation is the fallback for
            # whatever did not come back in place (torch pose network, a foreign graph).  The root gradient is a persistent one (no fill).
            one = self._one if (self._one is not None and self._one.device == loss.device and loss.dim() == 0) else None
            grads = torch.autograd.grad(loss, params, grad_outputs=one)
            if not all(g.data_ptr() == p.grad.data_ptr() for g, p in zip(grads, params)):
                n = sum(p.numel() for p in params)
                torch.cat([g.reshape(-1) for g in grads], out=flat[:n])
        else:
            self.opt.zero_grad()
            self._rebind_flat_gradient()
            loss.backward()
        return loss.detach(), pose.detach(), Tn

    @staticmethod
    def _offsets(params):
        off = 0
        for p in params:
            yield off
            off += p.numel()

    def _reduce(self):
        if self.reducer is not None and self.reducer.active:
            # the one collective of this path: mean over the ranks of the pose network's gradient (DPDist is frozen: nothing of it travels)
            self.reducer.reduce_async(0)
            self.reducer.wait()
            self._flat_grad.mul_(self.reducer.grad_scale)

    def loss_and_gradients(self, refined_source, template):
        """The training `sess.run` of :468 without the update: loss, predicted pose; gradients are left in the parameters'
        `.grad` (d loss / d moved source comes from the HIP backward-to-input path when loss_fn is DPDistLoss)."""
        loss, pose, _ = self._evaluate_grad(refined_source, template)
        self._reduce()
        return loss, pose

    def _train_step_eager(self, source, template):
        self.net.train()
        src, T = self.refine(source, template, self.max_loops - 1, train_mask=self._fused(source))
        loss, _, Tn = self._evaluate_grad(src, template, T)
        self._reduce()
        self.opt.step()
        return loss, Tn

    def train_step(self, source, template):
        if self.use_graph and source.is_cuda:
            return self._graph_step("train", source, template)
        return self._train_step_eager(source, template)

    @torch.no_grad()
    def _evaluate_eager(self, source, template):
        self.net.eval()
        moved, T = self.refine(source, template, self.max_loops)
        return self.loss_fn(moved, template), T

    def evaluate(self, source, template):
        if self.use_graph and source.is_cuda:
            return self._graph_step("eval", source, template)
        return self._evaluate_eager(source, template)

    # ---- hipGraph form
    def _loss_key(self):
        k = getattr(self.loss_fn, "graph_key", None)
        return k() if callable(k) else None

    def _graph_step(self, kind, source, template):
        if source.shape != template.shape or source.dtype != torch.float32:
            return self._train_step_eager(source, template) if kind == "train" else self._evaluate_eager(source, template)
        key = (kind, tuple(source.shape), source.device.index)
        rec = self._graphs.get(key)
        if rec is not None and rec.loss_key != self._loss_key():     # the frozen loss changed under the graph (new DPDist weights): recapture
            del self._graphs[key]
            rec = None
        if rec is None:
            seen = self._graph_seen[key] = self._graph_seen.get(key, 0) + 1
            if seen <= self.graph_warmup or len(self._graphs) >= 8:   # real (eager) steps first: library handles, autograd, engine shapes
                return self._train_step_eager(source, template) if kind == "train" else self._evaluate_eager(source, template)
            rec = self._graphs[key] = self._capture(kind, source, template)
        rec.src.copy_(source)
        rec.tmpl.copy_(template)
        if kind == "eval":
            rec.g1.replay()
        elif rec.g2 is None:
            self.opt.prepare_replay()
            rec.g1.replay()
        else:
            rec.g1.replay()
            self._reduce()
            self.opt.prepare_replay()
            rec.g2.replay()
        self.graph_replays += 1
        return rec.loss.clone(), rec.T.clone()

    def _capture(self, kind, source, template):
        from . import asloss
        rec = _GraphRec()
        rec.src, rec.tmpl, rec.pool, rec.g2 = source.clone(), template.clone(), {}, None
        rec.loss_key = self._loss_key()
        self.net.train(kind == "train")
        with asloss.private_pool(rec.pool):
            with torch.no_grad():
                self.loss_fn(rec.src, rec.tmpl)              # creates the graph's own engine and derives the frozen weights OUTSIDE the capture
            torch.cuda.synchronize()
            rec.g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(rec.g1):
                if kind == "eval":
                    with torch.no_grad():
                        moved, rec.T = self.refine(rec.src, rec.tmpl, self.max_loops)
                        rec.loss = self.loss_fn(moved, rec.tmpl)
                else:
                    src, T = self.refine(rec.src, rec.tmpl, self.max_loops - 1, train_mask=self._fused(rec.src))
                    rec.loss, _, rec.T = self._evaluate_grad(src, rec.tmpl, T)
                    if self.reducer is None or not self.reducer.active:
                        self.opt.step()                      # captured: dpd_adam_tf_dev, lr_t from device memory (TFAdam.prepare_replay)
            if kind == "train" and self.reducer is not None and self.reducer.active:
                rec.g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(rec.g2, pool=rec.g1.pool()):
                    self.opt.step()
        return rec
