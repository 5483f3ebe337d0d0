#!/usr/bin/env python3
"""DPDist as a frozen loss (pcrnet-registration/iterative_PCRNet_ours.py:229-257): forward both directions + backward to
the INPUT clouds (decoder dX chain on all 2*B*N rows, window scatter, encoder backward).  GPU only.

    python tools/asloss_bench.py [--batch 16] [--steps 50]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dpdist_amd import synth  # noqa: E402
from dpdist_amd.model import DPDistLoss, DPDistModel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)     # run_train_and_eval_PCRNet.bash:18,72
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=60)     # the chip needs ~25 ms under load to reach its steady clock (tools/ramp_probe.py)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f32x3", "bf16"])
    ap.add_argument("--engine", type=int, default=1, help="0 = the entry-by-entry autograd node (dpdist_amd.asloss.ENGINE = False)")
    ap.add_argument("--plan", default="", help="GEMM plan overrides, e.g. '7:30' = op:tile[:split_k] (include/dpdist_capi.h: dpd_set_gemm_plan)")
    a = ap.parse_args()
    from dpdist_amd import asloss, lib as Lb
    asloss.ENGINE = bool(a.engine)
    for item in filter(None, a.plan.split(",")):
        f = [int(x) for x in item.split(":")]
        Lb.check(Lb.load().dpd_set_gemm_plan(f[0], f[1], f[2] if len(f) > 2 else 1), "dpd_set_gemm_plan")
    dev = torch.device("cuda:0")
    model = DPDistModel(device=dev)
    model.load_tf_state_dict(synth.make_weights("wide"))
    model.params_.compute_dtype = a.dtype
    loss_fn = DPDistLoss(model)
    pcA, pcB, _ = synth.s2_modelnet_shaped(a.batch, 64, 100)
    src = torch.tensor(pcA, device=dev, requires_grad=True)
    tmpl = torch.tensor(pcB, device=dev)

    def step():
        src.grad = None
        loss = loss_fn(src, tmpl)
        loss.backward()
        return loss

    def fwd_only():        # the seven no-gradient refinements of a registration step (iterative_PCRNet_ours.py:414-441)
        with torch.no_grad():
            return loss_fn(src, tmpl)

    def run(fn):
        for _ in range(a.warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            l = fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, l

    engine = bool(a.engine)
    el, l = run(step)
    print(json.dumps({"mode": "as-loss (fwd + bwd to inputs)", "engine": engine, "batch": a.batch, "dtype": a.dtype,
                      "ms_per_step": round(el / a.steps * 1e3, 4), "query_points_per_sec": round(2 * a.batch * 64 * a.steps / el, 1),
                      "loss_pred": round(float(l), 6), "grad_norm": round(float(src.grad.norm()), 6)}))
    el, l = run(fwd_only)
    print(json.dumps({"mode": "as-loss (forward only, no_grad)", "engine": engine, "batch": a.batch, "dtype": a.dtype,
                      "ms_per_step": round(el / a.steps * 1e3, 4), "query_points_per_sec": round(2 * a.batch * 64 * a.steps / el, 1),
                      "loss_pred": round(float(l), 6)}))


if __name__ == "__main__":
    main()
