"""Where does a phase-staggered plane GEMM launch spend its time?  s_memtime stamps of wave 0 of every workgroup (ablation build:
DPD_ABLATIONS=1 python -m dpdist_amd.build --force).   python tools/p8_stamps.py [K ...]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import x3_bench  # noqa: E402
from dpdist_amd import lib as L  # noqa: E402

lib = L.load()
outs = os.environ.get("OUTS", "")
for K in [int(k) for k in sys.argv[1:]] or [64, 1024, 2528]:
    x3_bench.run("NN", 8192, 1024, K, 1, 232, iters=5)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (1024 * 8))()
    f = lib.dpd_debug_p8_stamps
    f.argtypes = [ctypes.c_void_p]
    assert f(buf) == 0
    st = np.array(buf, dtype=np.uint64).reshape(1024, 8)[:256, :5].astype(np.int64)
    t0 = st[:, 0].min()
    rel = st - t0
    names = ["start", "addr setup done", "K-tile 0 landed", "K loop done", "epilogue done"]
    print("K=%d OUTS=%r  (s_memtime ticks; 100 MHz => x10 ns)" % (K, outs))
    for i, n in enumerate(names):
        print("  %-18s median %8.0f   min %8.0f   max %8.0f" % (n, np.median(rel[:, i]), rel[:, i].min(), rel[:, i].max()))
    d = np.diff(st, axis=1)           # per-workgroup phase durations (the s_memtime bases differ between XCDs: only differences within a row mean anything)
    for i, n in enumerate(["addr setup", "first K-tile", "K loop", "epilogue"]):
        print("  d %-16s median %8.0f   min %8.0f   max %8.0f" % (n, np.median(d[:, i]), d[:, i].min(), d[:, i].max()))
