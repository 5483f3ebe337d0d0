"""(Ablation tile codes need a library built with `python -m dpdist_amd.build --ablations`.)
Correctness (vs fp64) and speed of the split-bf16 MFMA GEMM (gemm_x3.hip) next to the exact-fp32 MFMA GEMM.
    python tools/x3_bench.py [tiles...]"""
import ctypes
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dpdist_amd import lib as L, ops  # noqa: E402

lib = L.load()
P, I, LG = ctypes.c_void_p, ctypes.c_int, ctypes.c_long


dev = "cuda"


def planes(x, np_, want_rc, want_r8):
    R, C = x.shape
    rc = torch.empty(np_, R, C, device=dev, dtype=torch.int16) if want_rc else None
    r8 = torch.empty(np_, R // 8, C, 8, device=dev, dtype=torch.int16) if want_r8 else None
    rcode = lib.dpd_split_planes(L.ptr(x), R, C, x.stride(0), np_, L.ptr(rc), C, R * C, L.ptr(r8), R * C, L.cur_stream())
    assert rcode == 0, rcode
    return rc, r8


def run(mode, M, N, K, np_, tile, iters=20):
    g = torch.Generator(device="cpu").manual_seed(1)
    if mode == "NN":
        A = torch.randn(M, K, generator=g).to(dev); B = torch.randn(K, N, generator=g).to(dev)
        ref = A.double() @ B.double()
        a_rc, _ = planes(A, np_, True, False); _, b_r8 = planes(B, np_, False, True)
        args = (np_, 0, 1, M, N, K, L.ptr(a_rc), K, M * K, L.ptr(b_r8), N, K * N)
        f32 = lambda: ops.gemm_f32(A, B, tile=9 if M * N >= 128 * 128 * 256 else 8)
    elif mode == "NT":
        A = torch.randn(M, K, generator=g).to(dev); B = torch.randn(N, K, generator=g).to(dev)
        ref = A.double() @ B.double().t()
        a_rc, _ = planes(A, np_, True, False); b_rc, _ = planes(B, np_, True, False)
        args = (np_, 0, 0, M, N, K, L.ptr(a_rc), K, M * K, L.ptr(b_rc), K, N * K)
        f32 = lambda: ops.gemm_f32(A, B, transB=True, tile=8)
    elif mode == "TR":      # TN with both operands as RC planes (LDS transpose reads)
        A = torch.randn(K, M, generator=g).to(dev); B = torch.randn(K, N, generator=g).to(dev)
        ref = A.double().t() @ B.double()
        a_rc, _ = planes(A, np_, True, False); b_rc, _ = planes(B, np_, True, False)
        args = (np_, 2, 2, M, N, K, L.ptr(a_rc), M, K * M, L.ptr(b_rc), N, K * N)
        f32 = lambda: ops.gemm_f32(A, B, transA=True, tile=8)
    else:
        A = torch.randn(K, M, generator=g).to(dev); B = torch.randn(K, N, generator=g).to(dev)
        ref = A.double().t() @ B.double()
        _, a_r8 = planes(A, np_, False, True); _, b_r8 = planes(B, np_, False, True)
        args = (np_, 1, 1, M, N, K, L.ptr(a_r8), M, K * M, L.ptr(b_r8), N, K * N)
        f32 = lambda: ops.gemm_f32(A, B, transA=True, tile=8)
    C = torch.empty(M, N, device=dev)
    import os
    outs = os.environ.get("OUTS", "")
    o_rc = torch.empty(np_, M, N, device=dev, dtype=torch.int16) if "rc" in outs else None
    o_r8 = torch.empty(np_, M // 8, N, 8, device=dev, dtype=torch.int16) if "r8" in outs else None
    call = lambda: lib.dpd_gemm_planes(*args, L.ptr(C) if "noc" not in outs else None, N, None, None, 0, tile, L.ptr(o_rc), L.ptr(o_r8),
                                       M if o_r8 is not None else 0, L.cur_stream())
    rc = call()
    if rc != 0:
        print(f"{mode} tile {tile}: rc={rc}"); return
    torch.cuda.synchronize()
    scale = (A.double().abs().mean() * B.double().abs().mean() * K).item()
    err = (C.double() - ref).abs().max().item()
    err32 = (f32().double() - ref).abs().max().item()
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    e0.record()
    for _ in range(iters): f32()
    e1.record(); torch.cuda.synchronize()
    ms32 = e0.elapsed_time(e1) / iters
    fl = 2.0 * M * N * K
    print(f"{mode} {M}x{N}x{K} np={np_} tile={tile}: {ms*1e3:7.1f} us {fl/ms/1e9:7.1f} TF(fp32-equiv)  max|err|={err:.3e} "
          f"(rel {err/scale:.2e}) | f32 MFMA: {ms32*1e3:7.1f} us {fl/ms32/1e9:6.1f} TF err={err32:.3e}", flush=True)


if __name__ == "__main__":
    tiles = [int(t) for t in sys.argv[1:]] or [1, 2, 21, 24]
    import os
    quick = os.environ.get("QUICK")
    if os.environ.get("B64"):
        for np_ in (1, 3):
            for tile in tiles:
                run("NN", 8192, 1024, 2528, np_, tile)
                run("NN", 8192, 1024, 1024, np_, tile)
                run("NT", 4096, 1024, 1024, np_, tile)
                run("TN", 2560, 1024, 4096, np_, tile)
                run("TN", 1024, 1024, 4096, np_, tile)
        sys.exit(0)
    if os.environ.get("MODES"):
        for tile in tiles:
            for mode in ("NN", "NT", "TN"):
                run(mode, 4096, 1024, 2048, 3, tile)
        sys.exit(0)
    for np_ in (3,) if quick else (3, 1):
        for tile in tiles:
            run("NN", 4096, 1024, 2528, np_, tile)
            if quick: continue
            run("NN", 4096, 1024, 1024, np_, tile)
            run("NT", 2048, 1024, 1024, np_, tile)
            run("TN", 2560, 1024, 2048, np_, tile)
            run("TN", 1024, 1024, 2048, np_, tile)
