#!/usr/bin/env python3
"""Row f2 demo: train DPDist on synthetic shapes, freeze it, train an iterative PCRNet pose network with DPDist as the loss.

    python tools/registration_demo.py [--dp_steps 3000] [--reg_steps 1500] [--batch 16]
"""
import argparse
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dpdist_amd import synth  # noqa: E402
from dpdist_amd.model import DPDistLoss, DPDistModel  # noqa: E402
from dpdist_amd.registration import IterativeRegistration, PoseNet, pose_errors as find_errors, quat_to_mat  # noqa: E402
from dpdist_amd.trainer import DPDistTrainer  # noqa: E402


def make_pairs(B, N, rng, max_deg=45.0, max_t=0.1):
    """template = N surface samples; source = R_gt (other N samples of the same surface) + t_gt."""
    tmpl = np.zeros((B, N, 3), np.float32)
    src = np.zeros((B, N, 3), np.float32)
    Rg = np.zeros((B, 3, 3), np.float32)
    tg = rng.uniform(-max_t, max_t, (B, 3)).astype(np.float32)
    for b in range(B):
        if rng.random() < 0.5:
            h = rng.uniform(0.2, 0.5, 3)
            samp = lambda n: synth._sample_box(rng, n, h)      # noqa: E731
        else:
            r = rng.uniform(0.3, 0.6, 3)                        # ellipsoid (a sphere has no orientation)
            samp = lambda n: synth._sample_sphere(rng, n, 1.0) * r   # noqa: E731
        axis = rng.standard_normal(3)
        axis /= np.linalg.norm(axis)
        ang = math.radians(rng.uniform(-max_deg, max_deg))
        q = torch.tensor([[math.cos(ang / 2), *(axis * math.sin(ang / 2))]], dtype=torch.float32)
        Rg[b] = quat_to_mat(q)[0].numpy()
        tmpl[b] = samp(N)
        src[b] = samp(N) @ Rg[b].T + tg[b]
    return src, tmpl, Rg, tg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dp_steps", type=int, default=3000)
    ap.add_argument("--reg_steps", type=int, default=1500)
    ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    cu = lambda x: torch.tensor(x, device=dev)   # noqa: E731

    # 1. train DPDist (the hot path's own trainer) on synthetic distance data
    model = DPDistModel(device=dev)
    tr = DPDistTrainer(model.params_, 32, base_lr=2e-4, distributed=False)
    pool = [synth.s2_modelnet_shaped(32, 64, 1000 + i) for i in range(64)]
    pool = [tuple(cu(x) for x in p) for p in pool]
    for s in range(a.dp_steps):
        loss = tr.step(*pool[s % len(pool)])
    print("DPDist trained: loss_samples %.4f" % loss[0].item())

    # 2. iterative PCRNet with the frozen DPDist loss
    reg = IterativeRegistration(PoseNet().to(dev), DPDistLoss(model), lr=1e-4)
    ev = make_pairs(64, 64, np.random.default_rng(99))
    es, et, eR, etg = (cu(x) for x in ev)

    def evaluate():
        losses, terr, rerr = [], [], []
        for i in range(0, 64, a.batch):
            l, T = reg.evaluate(es[i:i + a.batch], et[i:i + a.batch])
            te, re = find_errors(T, eR[i:i + a.batch], etg[i:i + a.batch])
            losses.append(l.item()); terr.append(te); rerr.append(re)
        return float(np.mean(losses)), torch.cat(terr).mean().item(), torch.cat(rerr).mean().item()

    ident = find_errors(torch.eye(4, device=dev).repeat(64, 1, 1), eR, etg)
    print("identity pose: trans err %.4f  rot err %.2f deg" % (ident[0].mean().item(), ident[1].mean().item()))
    print("before training:", evaluate())
    for s in range(a.reg_steps):
        src, tmpl, _, _ = make_pairs(a.batch, 64, rng)
        l, _ = reg.train_step(cu(src), cu(tmpl))
        if (s + 1) % 250 == 0:
            print("step %d  train loss %.4f  eval (loss, trans, rot_deg) %s" % (s + 1, l.item(), evaluate()))
    out = evaluate()
    print(json.dumps({"eval_loss": out[0], "trans_err": out[1], "rot_err_deg": out[2], "identity_rot_err_deg": ident[1].mean().item()}))


if __name__ == "__main__":
    main()
