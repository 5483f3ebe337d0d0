"""One plane-GEMM shape for counter passes: python tools/x3_one.py MODE M N K NP TILE [iters]"""
import sys
sys.argv, a = sys.argv[:1], sys.argv[1:]
import x3_bench  # noqa: E402
x3_bench.run(a[0], int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[5]), iters=int(a[6]) if len(a) > 6 else 5)
