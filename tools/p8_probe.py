"""Fixed cost and ablations of the phase-staggered plane GEMM (needs DPD_ABLATIONS=1 python -m dpdist_amd.build --force for codes 200+).
    python tools/p8_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import x3_bench  # noqa: E402

tiles = [int(t) for t in sys.argv[1:]] or [2, 21, 201, 202, 204, 205, 207, 208, 216]
NP = int(os.environ.get("NP", "1"))
for K in (64, 128, 1024, 2528):
    for tile in tiles:
        x3_bench.run("NN", 8192 if NP == 1 else 4096, 1024, K, NP, tile, iters=30)
if NP == 3:
    for tile in tiles:
        x3_bench.run("NT", 2048, 1024, 1024, 3, tile, iters=30)
        x3_bench.run("TN", 2560, 1024, 2048, 3, tile, iters=30)
        x3_bench.run("TN", 1024, 1024, 2048, 3, tile, iters=30)
