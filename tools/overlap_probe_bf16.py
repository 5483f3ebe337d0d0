"""Upper bound for a heterogeneous grouped backward launch in the bf16 step at B = 64: the dH GEMM
(g3 W3^T, NT on RC planes, ReLU gate, plane outputs RC + R8) and a dW GEMM (h2^T g3, TN on R8 planes) back to back on one stream vs
concurrently on two streams with no dependency between them.  Also dW1 next to dH.
    python tools/overlap_probe_bf16.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpdist_amd import lib as L  # noqa: E402

lib = L.load()
dev = "cuda"
torch.manual_seed(0)
Qb, H, KP = 4096, 1024, 2528


def planes(x, want_rc, want_r8):
    R, C = x.shape
    rc = torch.empty(1, R, C, device=dev, dtype=torch.int16) if want_rc else None
    r8 = torch.empty(1, R // 8, C, 8, device=dev, dtype=torch.int16) if want_r8 else None
    assert lib.dpd_split_planes(L.ptr(x), R, C, x.stride(0), 1, L.ptr(rc), C, R * C, L.ptr(r8), R * C, L.cur_stream()) == 0
    return rc, r8


g3 = torch.randn(Qb, H, device=dev); W3 = torch.randn(H, H, device=dev); h2 = torch.relu(torch.randn(Qb, H, device=dev))
X = torch.randn(Qb, KP, device=dev)
g3_rc, g3_r8 = planes(g3, True, True)
W3_rc, _ = planes(W3, True, False)
_, h2_r8 = planes(h2, False, True)
_, X_r8 = planes(X, False, True)
g2 = torch.empty(Qb, H, device=dev); dW3 = torch.empty(H, H, device=dev); dW1 = torch.empty(KP, H, device=dev)
o_rc = torch.empty(1, Qb, H, device=dev, dtype=torch.int16); o_r8 = torch.empty(1, Qb // 8, H, 8, device=dev, dtype=torch.int16)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def dH(tile=2):    # NT: M = Qb, N = H, K = H; gate = h2 (fp32), planes out, no fp32 C (as in the training step)
    rc = lib.dpd_gemm_planes(1, 0, 0, Qb, H, H, L.ptr(g3_rc), H, Qb * H, L.ptr(W3_rc), H, H * H, None, H, None, L.ptr(h2), 3, tile,
                             L.ptr(o_rc), L.ptr(o_r8), Qb, L.cur_stream())
    assert rc == 0, rc


def dW(tile=2):    # TN: M = H, N = H, K = Qb
    rc = lib.dpd_gemm_planes(1, 1, 1, H, H, Qb, L.ptr(h2_r8), H, Qb * H, L.ptr(g3_r8), H, Qb * H, L.ptr(dW3), H, None, None, 0, tile,
                             None, None, 0, L.cur_stream())
    assert rc == 0, rc


def dWone(tile=2):  # TN: M = KP, N = H, K = Qb
    rc = lib.dpd_gemm_planes(1, 1, 1, KP, H, Qb, L.ptr(X_r8), KP, Qb * KP, L.ptr(g3_r8), H, Qb * H, L.ptr(dW1), H, None, None, 0, tile,
                             None, None, 0, L.cur_stream())
    assert rc == 0, rc


def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def par(a, b):
    def f():
        with torch.cuda.stream(s1): a()
        with torch.cuda.stream(s2): b()
    return f


print("dH (4096x1024x1024 NT, gate, RC+R8 planes out) alone: %.1f us" % timeit(dH))
for t in (2, 3, 5):
    print("dW (1024x1024x4096 TN) tile %d alone: %.1f us" % (t, timeit(lambda: dW(t))))
print("dW1 (2528x1024x4096 TN) alone: %.1f us" % timeit(dWone))
print("dH ; dW sequential: %.1f us" % timeit(lambda: (dH(), dW())))
for t in (2, 3, 5):
    print("dH || dW tile %d, two streams: %.1f us" % (t, timeit(par(dH, lambda: dW(t)))))
print("dH ; dW1 sequential: %.1f us" % timeit(lambda: (dH(), dWone())))
print("dH || dW1, two streams: %.1f us" % timeit(par(dH, dWone)))
print("dW ; dW1 sequential: %.1f us" % timeit(lambda: (dW(), dWone())))
print("dW || dW1, two streams: %.1f us" % timeit(par(dW, dWone)))
print("dH ; dH ; dW ; dW ; dW1 sequential (the backward): %.1f us" % timeit(lambda: (dH(), dH(), dW(), dW(), dWone())))
