import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import x3_bench
for np_ in (1, 3):
    for (M, N, K) in ((2560, 1024, 4096 if np_ == 1 else 2048), (1024, 1024, 4096 if np_ == 1 else 2048)):
        for tile in (2, 3):
            x3_bench.run("TN", M, N, K, np_, tile, iters=30)
            x3_bench.run("TR", M, N, K, np_, tile, iters=30)
