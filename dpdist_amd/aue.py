"""AUE task -- a PointNet autoencoder trained with DPDist as the (frozen) loss, or with Chamfer as the baseline
(the third consumer of the hot-path boundary).  The losses run on the HIP path; the
autoencoder is the consumer's own small PyTorch model.

Restates (relative to /root/reference):
    models/dpdist_and_aue.py:88-145        get_model_aue_pn: shared MLP 64-64-64-128-1024 (1x3 then 1x1 convs, BN + ReLU),
                                           max pool over the points, FC 1024-1024 (BN + ReLU), FC N*3, tanh
    utils/tf_util.py:558-577               batch_norm_template: tf.contrib.layers.batch_norm (epsilon 1e-3, decay = bn_decay)
    train_multi_gpu_pc_compare_dist.py:891-916   pairwise_diff / chmafer_dist (squared distances, both directions, / 2)
    train_multi_gpu_pc_compare_dist.py:403-470   splice: input1 <- AE output, input2 <- x2, add_noise <- x3;
                                                 loss_p = (mean(output1[...,0]) + mean(output2[...,0])) / 2, loss_c = chamfer(x1, out2);
                                                 gradients to the 'g2' (AE) variables only, Adam
    train_multi_gpu_pc_compare_dist.py:525-566   one step per batch with OPT_TYPE 'ours' (loss_p) or 'chamfer' (loss_c); both logged
"""
import numpy as np
import torch
from torch import nn

from . import lib as L


class _ChamferFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pc, rec):
        L.req(pc, name="pc"), L.req(rec, name="rec_pc")
        B, N, _ = pc.shape
        M = rec.shape[1]
        dev = pc.device
        min_a = torch.empty(B, N, device=dev)
        min_b = torch.empty(B, M, device=dev)
        arg_a = torch.empty(B, N, device=dev, dtype=torch.int32)
        arg_b = torch.empty(B, M, device=dev, dtype=torch.int32)
        loss = torch.empty(1, device=dev)
        L.check(L.load().dpd_chamfer_fwd(L.ptr(pc), L.ptr(rec), B, N, M, L.ptr(min_a), L.ptr(arg_a), L.ptr(min_b), L.ptr(arg_b),
                                         L.ptr(loss), L.cur_stream()), "dpd_chamfer_fwd")
        ctx.save_for_backward(pc, rec, arg_a, arg_b)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        pc, rec, arg_a, arg_b = ctx.saved_tensors
        B, N, _ = pc.shape
        M = rec.shape[1]
        da = torch.empty_like(pc) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(rec) if ctx.needs_input_grad[1] else None
        L.check(L.load().dpd_chamfer_bwd(L.ptr(pc), L.ptr(rec), B, N, M, L.ptr(arg_a), L.ptr(arg_b), 1.0, L.ptr(da), L.ptr(db),
                                         L.cur_stream()), "dpd_chamfer_bwd")
        # the upstream gradient stays on the device (no host sync, any scalar-shaped g)
        return (None if da is None else da.mul_(g)), (None if db is None else db.mul_(g))


def chamfer_dist(pc, rec_pc):
    """chmafer_dist(pc, rec_pc) of train_multi_gpu_pc_compare_dist.py:912-916 on the HIP kernels (squared distances)."""
    return _ChamferFn.apply(pc.contiguous(), rec_pc.contiguous())


class PointNetAE(nn.Module):
    """get_model_aue_pn (models/dpdist_and_aue.py:88-145); 1xW convs on [B,N,3,1] == per-point linear layers."""

    def __init__(self, num_point=64, bn=True, bn_decay=0.9):
        super().__init__()
        mom = 1.0 - bn_decay

        def block(i, o, with_bn):
            layers = [nn.Linear(i, o)]
            if with_bn:
                layers.append(nn.BatchNorm1d(o, eps=1e-3, momentum=mom))
            layers.append(nn.ReLU())
            return layers

        dims = [3, 64, 64, 64, 128, 1024]
        self.point = nn.ModuleList([nn.Sequential(*block(dims[i], dims[i + 1], bn)) for i in range(5)])
        self.fc = nn.Sequential(*block(1024, 1024, bn), *block(1024, 1024, bn), nn.Linear(1024, num_point * 3))
        self.num_point = num_point

    @torch.no_grad()
    def load_tf_state_dict(self, sd):
        """TF variables of get_model_aue_pn (scope 'aue'): <layer>/{weights, biases} and <layer>/bn/{beta, gamma, moving_mean,
        moving_variance} for conv1..5, fc1, fc2; fc3 plain."""
        get = lambda n: torch.as_tensor(np.asarray(sd[n]), dtype=torch.float32)   # noqa: E731

        def put(block, scope, has_bn):
            lin = block[0]
            w = get(scope + "/weights")
            lin.weight.copy_(w.reshape(-1, w.shape[-1]).t())
            lin.bias.copy_(get(scope + "/biases"))
            if has_bn and isinstance(block[1], nn.BatchNorm1d):
                bn = block[1]
                bn.bias.copy_(get(scope + "/bn/beta")); bn.weight.copy_(get(scope + "/bn/gamma"))
                bn.running_mean.copy_(get(scope + "/bn/moving_mean")); bn.running_var.copy_(get(scope + "/bn/moving_variance"))

        for i, blk in enumerate(self.point, 1):
            put(blk, "aue/conv%d" % i, True)
        mods = list(self.fc)
        per = (len(mods) - 1) // 2
        put(mods[:per], "aue/fc1", True)
        put(mods[per:2 * per], "aue/fc2", True)
        put(mods[2 * per:], "aue/fc3", False)

    def embed(self, pc):
        B, N, _ = pc.shape
        x = pc.reshape(B * N, 3)
        for blk in self.point:                       # BatchNorm statistics over (batch, points), like NHWC BN on [B,N,1,C]
            x = blk(x)
        return x.view(B, N, -1).amax(1)              # max_pool2d over the points (:131-132): end_points['embedding']

    def forward(self, pc):
        return torch.tanh(self.fc(self.embed(pc))).view(pc.shape[0], self.num_point, 3)    # :141-144


class AUETask:
    """train_one_epoch_3d_block (train_multi_gpu...:525-566): one Adam step on the autoencoder per batch with
    opt_type 'ours' (DPDist as the frozen loss) or 'chamfer'; both losses are evaluated and returned every step."""

    def __init__(self, autoencoder, dpdist_loss, lr=1e-3, opt_type="ours"):
        if opt_type not in ("ours", "chamfer"):
            raise ValueError("opt_type must be 'ours' or 'chamfer'")
        self.ae, self.loss_p, self.opt_type = autoencoder, dpdist_loss, opt_type
        from .optim import TFAdam            # tf.train.AdamOptimizer (train_multi_gpu_pc_compare_dist.py:216,457-463)
        self.opt = TFAdam(autoencoder.parameters(), lr=lr)

    def step(self, x1, x2):
        """x1: clouds fed to the autoencoder, x2: the second sampling of the same surfaces (input2 of DPDist)."""
        self.ae.train()
        out2 = self.ae(x1)
        loss_p = self.loss_p(out2, x2)               # input1 <- AE output, input2 <- x2, add_noise 0  (:417-424, :553)
        loss_c = chamfer_dist(x1, out2)              # :434
        self.opt.zero_grad()
        (loss_p if self.opt_type == "ours" else loss_c).backward()
        self.opt.step()
        return loss_p.detach(), loss_c.detach()

    @torch.no_grad()
    def evaluate(self, x1, x2):
        self.ae.eval()
        out2 = self.ae(x1)
        return self.loss_p(out2, x2), chamfer_dist(x1, out2), out2
