import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import x3_bench  # noqa: E402
for outs in ("", "rc", "rc r8", "noc rc", "noc rc r8"):
    os.environ["OUTS"] = outs
    print("OUTS=%r" % outs)
    for K in (64, 1024):
        for tile in (2, 21):
            x3_bench.run("NN", 8192, 1024, K, 1, tile, iters=30)
