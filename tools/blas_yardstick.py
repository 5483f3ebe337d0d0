"""Yardstick only (nothing of the product path uses a BLAS library): what torch.matmul's bf16 GEMM (hipBLASLt / rocBLAS kernels tuned by the
vendor) needs for the decoder's GEMM shapes at B = 64, next to this library's plane GEMMs of the same shapes (rocprofv3 averages in
profiles/r05_kernel_stats_bf16_b64.csv).  Plain C = A B in bf16 with fp32 accumulation: no bias / ReLU / gate epilogue, ONE bf16 output (the
plane GEMMs also write the R8 layout of half their rows).   python tools/blas_yardstick.py"""
import torch

dev = torch.device("cuda:0")


def t(fn, n=200):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [("fwd layer 1  8192 x 1024 x 2528 (NN)", 8192, 1024, 2528, False, False),
          ("fwd layer 2/3 8192 x 1024 x 1024 (NN)", 8192, 1024, 1024, False, False),
          ("dH  4096 x 1024 x 1024 (NT)", 4096, 1024, 1024, False, True),
          ("dW1 2528 x 1024 x 4096 (TN)", 2528, 1024, 4096, True, False),
          ("dW2/3 1024 x 1024 x 4096 (TN)", 1024, 1024, 4096, True, False)]
for name, M, N, K, ta, tb in shapes:
    A = torch.randn((K, M) if ta else (M, K), device=dev, dtype=torch.bfloat16)
    B = torch.randn((N, K) if tb else (K, N), device=dev, dtype=torch.bfloat16)
    a = A.t() if ta else A
    b = B.t() if tb else B
    us = t(lambda: torch.matmul(a, b))
    print("%-42s torch.matmul bf16: %6.1f us = %6.0f TFLOP/s" % (name, us, 2.0 * M * N * K / us / 1e6))
