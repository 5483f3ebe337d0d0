#!/usr/bin/env python3
"""Row f2 / BASELINE config 5 demo on one GPU: train DPDist on synthetic chairs, freeze it, train the iterative PCRNet pose
network with DPDist as the loss at the reference's workload (batch 16, 64 points, 8 loops, lim_rot 45, poses U(-45,45)^3 deg
/ U(-0.01,0.01)^3: pcrnet-registration/run_train_and_eval_PCRNet.bash:16-40), and report the reference's metric
(results_itrPCRNet_no_stop.py:112-133,465-474: find_errors(gt_pose, find_final_pose_inv(T))) on held-out pairs.

    python tools/registration_demo.py [--dp_steps 6000] [--reg_steps 6000] [--batch 16] [--loss ours|chamfer|both]

ModelNet40 'chair' is not in the tree: `synth.make_chair` draws box-union chairs (seat, back, legs, optional arm rests).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dpdist_amd import synth  # noqa: E402
from dpdist_amd.aue import chamfer_dist  # noqa: E402
from dpdist_amd.model import DPDistLoss, DPDistModel  # noqa: E402
from dpdist_amd.registration import IterativeRegistration, PoseNet, find_errors, find_final_pose_inv  # noqa: E402
from dpdist_amd.trainer import DPDistTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dp_steps", type=int, default=6000)
    ap.add_argument("--dp_pool", type=int, default=384, help="distinct DPDist training batches (32 chairs each)")
    ap.add_argument("--dp_lr", type=float, default=1e-4)
    ap.add_argument("--tilt", type=float, default=0.0, help="extra x/z tilt (deg) in DPDist's augmentation; the reference has none")
    ap.add_argument("--reg_steps", type=int, default=6000)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--loops", type=int, default=8)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--loss", default="both", choices=["ours", "chamfer", "both"])
    ap.add_argument("--eval_pairs", type=int, default=128)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cu = lambda x: torch.tensor(x, device=dev)   # noqa: E731
    out = {"workload": {"batch": a.batch, "num_point": 64, "loops": a.loops, "lim_rot": 45.0, "poses": "U(-45,45)^3 deg, U(-.01,.01)^3",
                        "shapes": "synthetic box-union chairs"}}

    # 1. DPDist's own trainer (the hot path) on chair distance data: reference recipe = y-rotation + shift augmentation
    t0 = time.time()
    model = DPDistModel(device=dev)
    tr = DPDistTrainer(model.params_, 32, base_lr=a.dp_lr, distributed=False)
    pool = [tuple(cu(x) for x in synth.s2_modelnet_shaped(32, 64, 5000 + i, shapes="chair", tilt_deg=a.tilt)) for i in range(a.dp_pool)]
    held = [tuple(cu(x) for x in synth.s2_modelnet_shaped(32, 64, 9000 + i, shapes="chair", tilt_deg=a.tilt)) for i in range(4)]
    t1 = time.time()
    run = 0.0
    for s in range(a.dp_steps):
        loss = tr.step(*pool[s % len(pool)])
        if (s + 1) % 500 == 0:
            run = loss[0].item()
            print("DPDist step %d  loss_samples %.4f" % (s + 1, run), flush=True)
    torch.cuda.synchronize()
    t2 = time.time()
    with torch.no_grad():
        ev = np.mean([(model(pA, pB)["pred_listAB"][:, :, 0, 0] - lab).abs().mean().item() for pA, pB, lab in held])
    print("DPDist trained: %d steps in %.1f s (+%.1f s of host-side chair generation); held-out mean L1 %.4f "
          "(reference's floor for 64 points: ~0.02, train_multi_gpu_pc_compare_dist.py:51-52)" % (a.dp_steps, t2 - t1, t1 - t0, ev))
    out["dpdist"] = {"steps": a.dp_steps, "train_l1": run, "heldout_l1": float(ev), "tilt_deg": a.tilt}

    # 2. iterative PCRNet, DPDist frozen, as the loss ('ours') and the reference's Chamfer baseline (iterative_PCRNet.py)
    es, et, eg = synth.registration_pairs(a.eval_pairs, 64, seed=99)
    es, et = cu(es), cu(et)
    ident = np.array([find_errors(eg[i], np.zeros(6)) for i in range(len(eg))])
    print("identity pose: trans err %.4f  rot err %.2f deg" % tuple(ident.mean(0)))
    out["identity"] = {"trans_err": float(ident[:, 0].mean()), "rot_err_deg": float(ident[:, 1].mean())}
    dp_loss = DPDistLoss(model)

    def evaluate(reg):
        losses, errs = [], []
        for i in range(0, len(eg), a.batch):
            l, T = reg.evaluate(es[i:i + a.batch], et[i:i + a.batch])
            fp = find_final_pose_inv(T.double().cpu().numpy())
            errs += [find_errors(eg[i + j], fp[j]) for j in range(fp.shape[0])]
            losses.append(l.item())
        errs = np.array(errs)
        return {"eval_loss": float(np.mean(losses)), "trans_err": float(errs[:, 0].mean()), "rot_err_deg": float(errs[:, 1].mean()),
                "rot_err_median_deg": float(np.median(errs[:, 1])), "rot_success_5deg": float((errs[:, 1] < 5).mean())}

    for name in (["ours", "chamfer"] if a.loss == "both" else [a.loss]):
        torch.manual_seed(0)
        rng = np.random.default_rng(0)
        loss_fn = dp_loss if name == "ours" else (lambda moved, tmpl: chamfer_dist(moved, tmpl))
        reg = IterativeRegistration(PoseNet().to(dev), loss_fn, lr=a.lr, max_loops=a.loops)
        print("[%s] before training: %s" % (name, evaluate(reg)), flush=True)
        t0 = time.time()
        gen = 0.0
        for s in range(a.reg_steps):
            tg = time.time()
            src, tmpl, _ = synth.registration_pairs(a.batch, 64, rng=rng)
            gen += time.time() - tg
            l, _ = reg.train_step(cu(src), cu(tmpl))
            if (s + 1) % 500 == 0:
                print("[%s] step %d  train loss %.4f  eval %s" % (name, s + 1, l.item(), evaluate(reg)), flush=True)
        res = evaluate(reg)
        res["train_s"] = time.time() - t0
        res["host_gen_s"] = gen
        res["steps"] = a.reg_steps
        out["pcrnet_" + name] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
