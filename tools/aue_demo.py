"""AUE task demo (row f4): train the DPDist decoder briefly on synthetic surfaces, freeze it, then train a PointNet
autoencoder with DPDist as the loss ('ours') and with Chamfer, logging both losses like train_one_epoch_3d_block
(train_multi_gpu_pc_compare_dist.py:525-573).      python tools/aue_demo.py [--steps 300]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpdist_amd import synth  # noqa: E402
from dpdist_amd.aue import AUETask, PointNetAE  # noqa: E402
from dpdist_amd.model import DPDistLoss, DPDistModel  # noqa: E402
from dpdist_amd.trainer import DPDistTrainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--pretrain", type=int, default=4000)
ap.add_argument("--batch", type=int, default=16)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
dp = DPDistModel(device=dev)
tr = DPDistTrainer(dp.params_, 32, base_lr=2e-4, distributed=False)
pool = [tuple(torch.tensor(x, device=dev) for x in synth.s2_modelnet_shaped(32, 64, 1000 + i)) for i in range(16)]
for t in range(a.pretrain):                                   # stage 1: train DPDist itself (row f1)
    loss = tr.step(*pool[t % len(pool)])
print("DPDist pre-training: %d steps, loss_samples %.4f" % (a.pretrain, loss[0].item()))
for opt_type in ("ours", "chamfer"):                          # stage 2: the AUE task with the frozen DPDist
    ae = PointNetAE(num_point=64).to(dev)
    task = AUETask(ae, DPDistLoss(dp), lr=1e-3, opt_type=opt_type)
    for t in range(a.steps):
        pcA, pcB, _ = synth.s2_modelnet_shaped(a.batch, 64, 5000 + t % 32)
        x1, x2 = torch.tensor(pcA, device=dev), torch.tensor(pcB[:, :64].copy(), device=dev)
        lp, lc = task.step(x1, x2)
        if t % 50 == 0 or t == a.steps - 1:
            print("opt_type %-7s step %4d   mean loss (DPDist) %.5f   chamf mean loss %.5f" % (opt_type, t, lp.item(), lc.item()))
