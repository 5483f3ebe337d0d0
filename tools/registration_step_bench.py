#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: ms per registration TRAINING step (batch 16, 64 points, 8 loops: 7 pose-only refinements + 1 DPDist
forward/backward + TF-form Adam, pcrnet-registration/iterative_PCRNet_ours.py:410-470) for the three forms of the step:

    eager_torch     torch pose network + torch pose algebra (round 5: ~1300 launches, host-bound)
    eager_fused     torch pose network, one launch per loop for the quaternion chain (csrc/pose.hip: dpd_pose_apply_*)
    eager_native    + the forward-only refinements' pose network on the library (dpd_pose_refine: 4 launches per loop + 1) and its training evaluation
    graph_fused     eager_fused captured once as a hipGraph and replayed
    graph           eager_native captured (the default form; bitwise its eager training: tests/test_registration.py)

Inputs are resident on the GPU before the clock starts (pair generation is the data loader's business, not this path's).  Also prints
what one DPDist forward+backward costs on its own (the as-loss engine, same shape), i.e. the share of the step that IS the hot path.

    python tools/registration_step_bench.py [--batch 16] [--steps 300] [--dtype f32|f32x3|bf16] [--loops 8]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--loops", type=int, default=8)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--forms", default="eager_torch,eager_fused,eager_native,graph_fused,graph")
    ap.add_argument("--train-only", action="store_true", help="profiler runs: exactly 8 + 3 * steps training steps per form, nothing else")
    a = ap.parse_args()
    from dpdist_amd import synth
    from dpdist_amd.model import DPDistLoss, DPDistModel
    from dpdist_amd.registration import IterativeRegistration, PoseNet
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    pool = [tuple(torch.tensor(x, device=dev) for x in synth.registration_pairs(a.batch, 64, rng=rng)[:2]) for _ in range(32)]
    out = {"workload": {"batch": a.batch, "num_point": 64, "loops": a.loops, "dpdist_dtype": a.dtype}}

    def harness(graph, fused, native=True):
        torch.manual_seed(0)
        model = DPDistModel(device=dev)
        model.load_tf_state_dict(synth.make_weights("wide"))
        model.params_.compute_dtype = a.dtype
        net = PoseNet().to(dev)
        torch.manual_seed(1000)
        return model, IterativeRegistration(net, DPDistLoss(model), lr=1e-4, max_loops=a.loops, distributed=False, graph=graph, fused_pose=fused,
                                            native_refine=native)

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    for form in a.forms.split(","):
        graph, fused, native = form.startswith("graph"), form != "eager_torch", form in ("eager_native", "graph")
        model, reg = harness(graph, fused, native)
        for i in range(8):
            reg.train_step(*pool[i % len(pool)])
        ms = min(timed(lambda i: reg.train_step(*pool[i % len(pool)]), a.steps) for _ in range(3))
        if a.train_only:
            print(form, json.dumps({"train_ms_per_step": round(ms, 4), "training_steps_run": 8 + 3 * a.steps}), flush=True)
            reg.close()
            continue
        for i in range(4):
            reg.evaluate(*pool[i % len(pool)])
        ms_eval = min(timed(lambda i: reg.evaluate(*pool[i % len(pool)]), a.steps) for _ in range(2))
        loss, _ = reg.train_step(*pool[0])
        out[form] = {"train_ms_per_step": round(ms, 4), "eval_ms_per_batch": round(ms_eval, 4), "pairs_per_s": round(a.batch / ms * 1e3, 1),
                     "graph_replays": reg.graph_replays, "last_loss": loss.item()}
        if graph:
            # GPU time of one replay alone (events around back-to-back replays of the captured step, no input copies)
            rec = reg._graphs[("train", (a.batch, 64, 3), 0)]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(100):
                reg.opt.prepare_replay()
                rec.g1.replay()
            e1.record()
            torch.cuda.synchronize()
            out[form]["replay_gpu_ms"] = round(e0.elapsed_time(e1) / 100, 4)
        reg.close()
        print(form, json.dumps(out[form]), flush=True)

    if a.train_only:
        return
    # the DPDist share: one as-loss forward + backward at this shape, on its own
    model, _ = harness(False, True)
    fn = DPDistLoss(model)
    s, t = pool[0][0].clone().requires_grad_(True), pool[0][1]

    def one(_):
        s.grad = None
        fn(s, t).backward()
    for i in range(10):
        one(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(200):
        one(i)
    e1.record()
    torch.cuda.synchronize()
    out["dpdist_fwd_bwd_ms"] = round(e0.elapsed_time(e1) / 200, 4)
    with torch.no_grad():
        for i in range(10):
            fn(s, t)
        torch.cuda.synchronize()
        e0.record()
        for i in range(200):
            fn(s, t)
        e1.record()
        torch.cuda.synchronize()
    out["dpdist_fwd_only_ms"] = round(e0.elapsed_time(e1) / 200, 4)
    best = min((out[f]["train_ms_per_step"] for f in a.forms.split(",")), default=None)
    if best:
        out["dpdist_share_of_best_step"] = round(out["dpdist_fwd_bwd_ms"] / best, 3)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
