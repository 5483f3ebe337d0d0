#!/usr/bin/env python3
"""Micro-benchmark of the fp32 MFMA GEMM at the decoder's shapes (tile / split-K sweep).  GPU only.

    python tools/gemm_bench.py [--batch 32] [--iters 30]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dpdist_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--only", default="", help="substring filter on the shape name")
    ap.add_argument("--tiles", default="1,2,3")
    ap.add_argument("--warm", type=int, default=3)
    ap.add_argument("--splits", default="1,2,4", help="split-K factors tried on the TN (dW) shapes")
    ap.add_argument("--split-all", action="store_true", help="try the split-K factors on every shape, not only TN")
    a = ap.parse_args()
    tiles = [int(t) for t in a.tiles.split(",")]
    dev = torch.device("cuda:0")
    Q, BN, KP, H = 2 * a.batch * 64, a.batch * 64, 2528, 1024
    r = lambda *s: torch.randn(*s, device=dev)   # noqa: E731
    shapes = [
        ("fwd_L1  NN", (r(Q, KP), r(KP, H), False, False)),
        ("fwd_L23 NN", (r(Q, H), r(H, H), False, False)),
        ("bwd_dH  NT", (r(BN, H), r(H, H), False, True)),
        ("bwd_dH  NN", (r(BN, H), r(H, H), False, False)),      # what the step runs: g x W^T as an NN product on the transposed copy
        ("bwd_dX  NT", (r(BN, H), r(KP, H), False, True)),
        ("bwd_dW1 TN", (r(BN, KP), r(BN, H), True, False)),
        ("bwd_dW23 TN", (r(BN, H), r(BN, H), True, False)),
    ]
    for name, (A, B, tA, tB) in shapes:
        if a.only and a.only not in name:
            continue
        M = A.shape[1] if tA else A.shape[0]
        K = A.shape[0] if tA else A.shape[1]
        N = B.shape[0] if tB else B.shape[1]
        flops = 2.0 * M * N * K
        line = "%-12s M=%5d N=%5d K=%5d |" % (name, M, N, K)
        for tile in tiles:
            for split in ((1,) if not (tA or a.split_all) else tuple(int(x) for x in a.splits.split(","))):
                out = torch.empty(M, N, device=dev)
                for _ in range(a.warm):
                    ops.gemm_f32(A, B, transA=tA, transB=tB, tile=tile, split_k=split, out=out)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    ops.gemm_f32(A, B, transA=tA, transB=tB, tile=tile, split_k=split, out=out)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / a.iters
                line += " t%d/s%d %6.1fus %5.1fTF |" % (tile, split, ms * 1e3, flops / ms / 1e9)
        print(line, flush=True)


if __name__ == "__main__":
    main()
