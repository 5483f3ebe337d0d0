#!/usr/bin/env python3
"""The non-GEMM tail of an as-loss backward, launch by launch (in-stream event pairs, back to back, same buffers):
window-gather backward (dX -> dfv), encoder backward in its forms, the final combine.  GPU only.

    python tools/asloss_tail_bench.py [--batch 16] [--reps 200]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=200)
    a = ap.parse_args()
    from dpdist_amd import lib as L, ops, synth
    dev = torch.device("cuda:0")
    B, N, m, k, sigma = a.batch, 64, 8, 5, 0.125
    C, Q, KP = 2 * B, 2 * B * N, ops.padded_width(5)
    pcA, pcB, _ = synth.s2_modelnet_shaped(B, N, 100)
    pts = torch.tensor(np.concatenate([pcA, pcB]), device=dev)
    q = torch.tensor(np.concatenate([pcB, pcA]), device=dev)
    fv = ops.mfv3d_fwd(pts, m, sigma)
    X, mask, vox = ops.patch_rows_fwd(q, fv, m, k)
    dX = torch.randn(Q, KP, device=dev) * 1e-3
    lib = L.load()
    out = {"batch": B, "clouds": C, "rows": Q}

    def timed(name, fn):
        for _ in range(20):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = round(e0.elapsed_time(e1) / a.reps * 1e3, 2)

    dfv = torch.empty(C, m ** 3, 20, device=dev)
    dpts = torch.empty(C, N, 3, device=dev)
    ws = torch.empty(lib.dpd_mfv3d_bwd_workspace_bytes(C, m) // 4, device=dev)
    gA, gB = torch.empty(B, N, 3, device=dev), torch.empty(B, N, 3, device=dev)
    s = L.cur_stream()
    timed("patch_rows_bwd_us", lambda: L.check(lib.dpd_patch_rows_bwd(L.ptr(dX), L.ptr(vox), C, N, m, k, KP, None, L.ptr(dfv), s), "prb"))
    timed("mfv3d_bwd_sliced_us", lambda: L.check(lib.dpd_mfv3d_bwd(L.ptr(pts), L.ptr(dfv), C, N, m, sigma, L.ptr(dpts), L.ptr(ws), ws.numel() * 4, s), "mb"))
    ref = dpts.clone()
    timed("mfv3d_bwd_one_workgroup_per_cloud_us", lambda: L.check(lib.dpd_mfv3d_bwd(L.ptr(pts), L.ptr(dfv), C, N, m, sigma, L.ptr(dpts), None, 0, s), "mb1"))
    out["max_abs_diff_forms"] = float((dpts - ref).abs().max())
    timed("asloss_combine_us", lambda: L.check(lib.dpd_asloss_combine(L.ptr(dpts), L.ptr(dX), None, B, N, k, KP, L.ptr(gA), L.ptr(gB), s), "ac"))

    def tail():
        L.check(lib.dpd_patch_rows_bwd(L.ptr(dX), L.ptr(vox), C, N, m, k, KP, None, L.ptr(dfv), s), "prb")
        L.check(lib.dpd_mfv3d_bwd(L.ptr(pts), L.ptr(dfv), C, N, m, sigma, L.ptr(dpts), L.ptr(ws), ws.numel() * 4, s), "mb")
        L.check(lib.dpd_asloss_combine(L.ptr(dpts), L.ptr(dX), None, B, N, k, KP, L.ptr(gA), L.ptr(gB), s), "ac")
    timed("tail_as_launched_us", tail)
    if hasattr(lib, "dpd_asloss_tail"):
        gA2, gB2 = torch.empty_like(gA), torch.empty_like(gB)
        tail()
        timed("tail_fused_us", lambda: L.check(lib.dpd_asloss_tail(L.ptr(dX), L.ptr(vox), L.ptr(pts), None, B, N, m, k, KP, sigma, L.ptr(dfv), L.ptr(ws),
                                                               ws.numel() * 4, L.ptr(gA2), L.ptr(gB2), s), "tail"))
        out["fused_max_abs_diff"] = float(max((gA2 - gA).abs().max(), (gB2 - gB).abs().max()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
