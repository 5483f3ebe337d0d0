"""Race screen for the LDS-DMA GEMM kernels: the rings rely on counted vmcnt waits and one raw barrier per K-tile, and a
misplaced wait would only show up as a rare wrong tile (when a DMA happens to land late).  Every kernel family is run many
times at the decoder's shapes, concurrently with a memory-hungry side stream that perturbs DMA latency, and every result is
compared with a float64 torch.matmul.            python tools/gemm_stress.py [--iters 200]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpdist_amd import lib as L, ops  # noqa: E402


def planes(x, np_, want_rc, want_r8):
    R, C = x.shape
    rc = torch.empty(np_, R, C, device=x.device, dtype=torch.int16) if want_rc else None
    r8 = torch.empty(np_, R // 8, C, 8, device=x.device, dtype=torch.int16) if want_r8 else None
    L.check(L.load().dpd_split_planes(L.ptr(x), R, C, x.stride(0), np_, L.ptr(rc), C, R * C, L.ptr(r8), R * C, L.cur_stream()), "split")
    return rc, r8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = L.load()
    side = torch.cuda.Stream()
    junk = torch.empty(256 << 20, device=dev, dtype=torch.uint8)
    shapes = {"NN": (4096, 1024, 2528), "NT": (2048, 1024, 1024), "TN": (2528, 1024, 2048)}
    bad = 0
    g = torch.Generator().manual_seed(0)
    for mode, (M, N, K) in shapes.items():
        A = torch.randn(M, K, generator=g).to(dev)
        B = torch.randn(K, N, generator=g).to(dev)
        ref = (A.double() @ B.double())
        tol = 5e-6 * K ** 0.5 * 16
        At, Bt = A.t().contiguous(), B.t().contiguous()
        cases = []
        for tile in (8, 9, 32):
            if mode == "NN":
                cases.append(("f32 tile %d" % tile, lambda t=tile: ops.gemm_f32(A, B, tile=t)))
            elif mode == "NT":
                cases.append(("f32 tile %d" % tile, lambda t=tile: ops.gemm_f32(A, Bt, transB=True, tile=t)))
            else:
                cases.append(("f32 tile %d" % tile, lambda t=tile: ops.gemm_f32(At, B, transA=True, tile=t)))
        for np_ in (3, 1):
            if mode == "NN":
                a_, _ = planes(A, np_, True, False); _, b_ = planes(B, np_, False, True)
                args = (np_, 0, 1, M, N, K, L.ptr(a_), K, M * K, L.ptr(b_), N, K * N)
            elif mode == "NT":
                a_, _ = planes(A, np_, True, False); b_, _ = planes(Bt, np_, True, False)
                args = (np_, 0, 0, M, N, K, L.ptr(a_), K, M * K, L.ptr(b_), K, N * K)
            else:
                _, a_ = planes(At, np_, False, True); _, b_ = planes(B, np_, False, True)
                args = (np_, 1, 1, M, N, K, L.ptr(a_), M, K * M, L.ptr(b_), N, K * N)
            # ring kernels 1-7; phase-staggered kernels 20-23 (one plane, BK = 64) / 24-26 (three planes, BK = 32)
            # (13 / 14: the 192x128 / 128x192 BK = 64 ring tiles of round 4: one plane, K % 64 == 0)
            for tile in (1, 2, 3, 5, 7) + ((20, 21, 22, 23) + ((13, 14) if K % 64 == 0 else ()) if np_ == 1 else (24, 25, 26)):
                def run(args=args, tile=tile, keep=(a_, b_)):
                    C = torch.empty(M, N, device=dev)
                    L.check(lib.dpd_gemm_planes(*args, L.ptr(C), N, None, None, 0, tile, None, None, 0, L.cur_stream()), "planes")
                    return C
                cases.append(("planes np=%d tile %d" % (np_, tile), run))
        for name, fn in cases:
            worst, nbad, ndiff, first = 0.0, 0, 0, None
            for it in range(a.iters):
                if it % 3 == 0:
                    with torch.cuda.stream(side):       # perturb the memory system while the GEMM runs
                        junk.add_(1)
                C = fn()
                if first is None:
                    first = C.clone()
                elif not torch.equal(C, first):
                    ndiff += 1
                err = (C.double() - ref).abs().max().item()
                lim = tol if "np=1" not in name else 0.05 * K ** 0.5
                worst = max(worst, err)
                nbad += err > lim
            torch.cuda.synchronize()
            print("%-3s %-22s iters %d  worst |err| %.3e  %s  %s" % (mode, name, a.iters, worst, "OK" if nbad == 0 else "BAD x%d" % nbad,
                                                                     "bitwise stable" if ndiff == 0 else "%d runs differ from the first" % ndiff),
                  flush=True)
            bad += nbad + ndiff
    # round 4: the weight gradients of the plane compute types as they run in the step -- grouped dW1 + dW2 + dW3 launch (192x128 and
    # 128x128 tiles) and the in-launch split-K (arrival counters, last-arriver reduction) -- repeated under the same memory pressure:
    # every run must reproduce the first bit for bit and agree with the separate whole-K launches
    from dpdist_amd import synth
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    B = 64
    pcA, pcB, lab = (torch.tensor(x, device=dev) for x in synth.s2_modelnet_shaped(B, 64, 100))
    base = None
    for name, env, plans in (("apart, whole K", "0", ((20, 0, 1), (32, 3, 1))), ("grouped 192x128", "1", ((33, 13, 1),)), ("grouped 128x128", "1", ((33, 2, 1),)),
                             ("grouped 128x128 split-K 2 in launch", "1", ((33, 2, 2),)), ("apart, dW1 split-K 3 + pair split-K 2 in launch", "0", ((20, 2, 3), (32, 2, 2)))):
        for op, tile, split in plans:
            ops.set_gemm_plan(op, tile, split)
        P = DPDistParams(device=dev, compute_dtype="bf16")
        P.load_tf_state_dict(synth.make_weights("wide"))
        tr = DPDistTrainer(P, B, distributed=False, options={"dw_trio": env == "1"})
        first, ndiff = None, 0
        for it in range(max(20, a.iters // 4)):
            if it % 2 == 0:
                with torch.cuda.stream(side):
                    junk.add_(1)
            tr._take_front(pcA, pcB, None)
            tr._decode()
            tr.backward(lab.reshape(-1))
            g = torch.cat([P.view(n, tr.grad).reshape(-1) for n in ("W1p", "W2", "W3")]).clone()
            if first is None:
                first = g
            elif not torch.equal(g, first):
                ndiff += 1
        torch.cuda.synchronize()
        if base is None:
            base = first
        rel = ((first - base).abs().max() / base.abs().max()).item()
        ok = ndiff == 0 and rel <= 4e-6
        print("bf16 B=64 weight gradients, %-48s %d runs: %s, vs apart max rel diff %.1e  %s" % (name, max(20, a.iters // 4), "bitwise stable" if ndiff == 0 else
              "%d runs differ" % ndiff, rel, "OK" if ok else "BAD"), flush=True)
        bad += 0 if ok else 1
        for op, tile, split in plans:
            ops.set_gemm_plan(op, 0, 1 if op == 33 else 0)
    print("RESULT:", "clean" if bad == 0 else "%d bad results" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
