"""Upper bound for a heterogeneous grouped backward launch: the dH GEMM (g W^T as an NN product on the transposed copy) and a dW GEMM
(act^T g) of the f32 step run back to back on one stream vs concurrently on two streams (no cross-stream dependencies).
    python tools/overlap_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpdist_amd import ops  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
Qb, H = 2048, 1024
g3 = torch.randn(Qb, H, device=dev); W3T = torch.randn(H, H, device=dev); h2 = torch.relu(torch.randn(Qb, H, device=dev))
g2 = torch.empty(Qb, H, device=dev); dW3 = torch.empty(H, H, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def dH(tile=32):
    ops.gemm_f32(g3, W3T, gate=h2, epilogue=3, tile=tile, out=g2)


def dW(tile=33):
    ops.gemm_f32(h2, g3, transA=True, tile=tile, out=dW3)


def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def seq():
    dH(); dW()


def par():
    with torch.cuda.stream(s1): dH()
    with torch.cuda.stream(s2): dW()


print("dH alone   %.1f us" % timeit(dH))
print("dW alone   %.1f us" % timeit(dW))
print("sequential %.1f us" % timeit(seq))
print("two streams %.1f us" % timeit(par))
for tH, tW in ((32, 32), (32, 31), (30, 33), (33, 33)):
    def par2():
        with torch.cuda.stream(s1): dH(tH)
        with torch.cuda.stream(s2): dW(tW)
    print("two streams, dH tile %d dW tile %d: %.1f us   (alone: %.1f / %.1f)" % (tH, tW, timeit(par2), timeit(lambda: dH(tH)), timeit(lambda: dW(tW))))
