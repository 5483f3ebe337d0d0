"""Fixed cost and ablations of the phase-staggered plane GEMM (needs DPD_ABLATIONS=1 python -m dpdist_amd.build --force for codes 200+).
    python tools/p8_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import x3_bench  # noqa: E402

tiles = [int(t) for t in sys.argv[1:]] or [2, 21, 201, 202, 204, 205, 207, 208, 216]
for K in (64, 128, 1024, 2528):
    for tile in tiles:
        x3_bench.run("NN", 8192, 1024, K, 1, tile, iters=30)
