#!/usr/bin/env python3
"""Row f2 / BASELINE config 5 demo on one GPU: train DPDist on synthetic chairs, freeze it, train the iterative PCRNet pose
network with DPDist as the loss at the reference's workload (batch 16, 64 points, 8 loops, lim_rot 45, poses U(-45,45)^3 deg
/ U(-0.01,0.01)^3: pcrnet-registration/run_train_and_eval_PCRNet.bash:16-40), and report the reference's metric
(results_itrPCRNet_no_stop.py:112-133,465-474: find_errors(gt_pose, find_final_pose_inv(T))) on held-out pairs.

    python tools/registration_demo.py [--dp_steps 6000] [--reg_steps 6000] [--batch 16] [--loss ours|chamfer|both] [--gpus N]

--gpus N (BASELINE config 5, "8 x MI355X DP"): one process per GPU, started like `bench.py --gpus N` starts its ranks (every rank under
the launch watchdog of dpdist_amd/launch.py).  DPDist is trained identically on every rank (bitwise reproducible fp32 step: 3 s, no
collective) and frozen; the registration batch is --batch pairs PER GPU (weak scaling), every rank draws its own pairs, the pose
network's flat gradient is all-reduced (registration.IterativeRegistration), the held-out pairs are split over the ranks and the
errors gathered on rank 0.

ModelNet40 'chair' is not in the tree: `synth.make_chair` draws box-union chairs (seat, back, legs, optional arm rests).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


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
    ap.add_argument("--gpus", type=int, default=1)
    a = ap.parse_args()
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    if a.gpus != world:
        if "WORLD_SIZE" in os.environ or a.gpus < 1:
            raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (a.gpus, world))
        import bench                                  # plain `--gpus N`: start the N ranks ourselves (bench.spawn_ranks: by PID, timeout)
        import torch
        if not torch.cuda.is_available() or (torch.cuda.device_count() < a.gpus and os.environ.get("DPD_TEST_SHARE_GPU") != "1"):
            sys.stderr.write("registration_demo --gpus %d: only %d GPU(s) visible\n" % (a.gpus, torch.cuda.device_count() if torch.cuda.is_available() else 0))
            raise SystemExit(2)
        raise SystemExit(bench.spawn_ranks(a.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                           timeout_s=3600.0))
    use_dist = world > 1 or os.environ.get("DPD_FORCE_DIST") == "1"
    from dpdist_amd import launch
    if use_dist:
        os.environ.setdefault("DPD_WD_LIMITS", "train=600,eval=300")      # phases of this script; everything else keeps launch.LIMITS
        launch.maybe_supervise(world)
    hb = launch.Heartbeat(rank)
    hb.beat("start:import")
    import torch
    import torch.distributed as dist
    from dpdist_amd import synth
    from dpdist_amd.aue import chamfer_dist
    from dpdist_amd.model import DPDistLoss, DPDistModel
    from dpdist_amd.registration import IterativeRegistration, PoseNet, centroid_residual, find_errors, find_final_pose_inv
    from dpdist_amd.trainer import DPDistTrainer
    share_gpu = os.environ.get("DPD_TEST_SHARE_GPU") == "1"      # tests only: every rank on GPU 0 over gloo (bench.py has the same switch)
    if share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29521")
        hb.beat("init:process group")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    say = print if rank == 0 else (lambda *x, **k: None)
    cu = lambda x: torch.tensor(x, device=dev)   # noqa: E731
    out = {"n_gpus": world, "workload": {"batch_per_gpu": a.batch, "global_batch": a.batch * world, "num_point": 64, "loops": a.loops, "lim_rot": 45.0, "poses": "U(-45,45)^3 deg, U(-.01,.01)^3",
                        "shapes": "synthetic box-union chairs"}}

    # 1. DPDist's own trainer (the hot path) on chair distance data: reference recipe = y-rotation + shift augmentation
    hb.beat("train:DPDist")
    t0 = time.time()
    # DPDist's TF-Xavier initialisation: the same on every rank AND in every run (round 5: it was unseeded, which made two runs of this demo
    # two different trainings -- the step itself is bitwise reproducible).  The seed matters the way it does in the reference: layer 1's
    # Xavier limit is 0.0015 (utils/tf_util.py:90-91 on a [1,2503,1,1024] kernel), so the initial outputs are ~1e-5 around a zero bias, and
    # an initialisation whose output channel 0 is negative on every input sits on relu6's flat side for good (seeds 0 and 1: the loss stays
    # at the labels' mean 0.08; seeds 2 and 1234 train to 0.013 in 1500 steps)
    torch.manual_seed(1234)
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
            say("DPDist step %d  loss_samples %.4f" % (s + 1, run), flush=True)
            hb.beat("train:DPDist step %d" % (s + 1))
    torch.cuda.synchronize()
    t2 = time.time()
    with torch.no_grad():
        ev = np.mean([(model(pA, pB)["pred_listAB"][:, :, 0, 0] - lab).abs().mean().item() for pA, pB, lab in held])
    say("DPDist trained: %d steps in %.1f s (+%.1f s of host-side chair generation); held-out mean L1 %.4f "
          "(reference's floor for 64 points: ~0.02, train_multi_gpu_pc_compare_dist.py:51-52)" % (a.dp_steps, t2 - t1, t1 - t0, ev))
    out["dpdist"] = {"steps": a.dp_steps, "train_l1": run, "heldout_l1": float(ev), "tilt_deg": a.tilt}

    # 2. iterative PCRNet, DPDist frozen, as the loss ('ours') and the reference's Chamfer baseline (iterative_PCRNet.py)
    es_all, et_all, eg_all = synth.registration_pairs(a.eval_pairs, 64, seed=99)
    per = (a.eval_pairs + world - 1) // world                # held-out pairs are split over the ranks
    sl = slice(rank * per, min(a.eval_pairs, (rank + 1) * per))
    es_np, eg = es_all[sl], eg_all[sl]
    es, et = cu(es_np), cu(et_all[sl])
    ident = np.array([find_errors(eg_all[i], np.zeros(6)) for i in range(len(eg_all))])
    ident_c = centroid_residual(np.tile(np.eye(4), (len(eg_all), 1, 1)), eg_all, es_all)
    say("identity pose: trans err %.4f  rot err %.2f deg  centroid residual %.4f" % (ident[:, 0].mean(), ident[:, 1].mean(), ident_c.mean()))
    out["identity"] = {"trans_err": float(ident[:, 0].mean()), "rot_err_deg": float(ident[:, 1].mean()), "centroid_residual": float(ident_c.mean())}
    dp_loss = DPDistLoss(model)

    def evaluate(reg):
        """the reference's metric on this rank's share of the held-out pairs, gathered on every rank"""
        losses, errs = [], []
        for i in range(0, len(eg), a.batch):
            l, T = reg.evaluate(es[i:i + a.batch], et[i:i + a.batch])
            Tn = T.double().cpu().numpy()
            fp = find_final_pose_inv(Tn)
            cr = centroid_residual(Tn, eg[i:i + a.batch], es_np[i:i + a.batch])
            errs += [find_errors(eg[i + j], fp[j]) + (cr[j],) for j in range(fp.shape[0])]
            losses.append(l.item())
        if use_dist:
            parts = [None] * world
            dist.all_gather_object(parts, (losses, errs))
            losses, errs = sum((p_[0] for p_ in parts), []), sum((p_[1] for p_ in parts), [])
        errs = np.array(errs)
        # median / success rate are the stable figures (the mean is carried by a handful of outliers among the held-out pairs);
        # centroid_residual: where the registered source's centroid ends up -- the pose-space translation error mostly measures
        # rotation error x the centroid's lever arm (registration.centroid_residual)
        return {"eval_loss": float(np.mean(losses)), "pairs": int(len(errs)), "trans_err": float(errs[:, 0].mean()),
                "trans_err_median": float(np.median(errs[:, 0])), "centroid_residual": float(errs[:, 2].mean()),
                "centroid_residual_median": float(np.median(errs[:, 2])), "rot_err_deg": float(errs[:, 1].mean()),
                "rot_err_median_deg": float(np.median(errs[:, 1])), "rot_success_5deg": float((errs[:, 1] < 5).mean())}

    for name in (["ours", "chamfer"] if a.loss == "both" else [a.loss]):
        torch.manual_seed(0)                                  # the same pose network on every rank (replicated variables)
        net = PoseNet().to(dev)
        torch.manual_seed(1000 + rank)                       # ... but its own dropout masks
        rng = np.random.default_rng(1000 * rank)             # ... and its own pairs
        loss_fn = dp_loss if name == "ours" else (lambda moved, tmpl: chamfer_dist(moved, tmpl))
        hb.beat("reducer:pose network")
        reg = IterativeRegistration(net, loss_fn, lr=a.lr, max_loops=a.loops, distributed=use_dist)
        hb.beat("eval:before training")
        say("[%s] before training: %s" % (name, evaluate(reg)), flush=True)
        hb.beat("train:%s" % name)
        t0 = time.time()
        gen = 0.0
        for s in range(a.reg_steps):
            tg = time.time()
            src, tmpl, _ = synth.registration_pairs(a.batch, 64, rng=rng)
            gen += time.time() - tg
            l, _ = reg.train_step(cu(src), cu(tmpl))
            if (s + 1) % 100 == 0:
                hb.beat("train:%s step %d" % (name, s + 1))
            if (s + 1) % 500 == 0:
                hb.beat("eval:step %d" % (s + 1))
                say("[%s] step %d  train loss %.4f  eval %s" % (name, s + 1, l.item(), evaluate(reg)), flush=True)
                hb.beat("train:%s step %d" % (name, s + 1))
        hb.beat("eval:final")
        res = evaluate(reg)
        res["train_s"] = time.time() - t0
        res["host_gen_s"] = gen
        res["steps"] = a.reg_steps
        res["pairs_per_s"] = a.batch * world * a.reg_steps / max(res["train_s"], 1e-9)
        if reg.reducer is not None and reg.reducer.active:
            red = reg.reducer
            res["dp"] = {"backend": red.backend, "fallback": hb.fallback, "nranks": int(red.nranks), "crosscheck": red.crosscheck,
                         "wire_bytes_per_gpu_per_step": red.wire_bytes_per_step, "pose_net_parameters": int(sum(p.numel() for p in net.parameters())),
                         "dpdist_collectives": 0}
            if use_dist:                                      # replicas must have stayed bit-identical
                w = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
                ws = [torch.empty_like(w) for _ in range(world)]
                dist.all_gather(ws, w)
                res["dp"]["replicas_bit_identical"] = bool(all(torch.equal(ws[0], x) for x in ws[1:]))
        reg.close()
        out["pcrnet_" + name] = res
    if rank == 0:
        print(json.dumps(out), flush=True)
    hb.beat("done")
    if use_dist:
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
