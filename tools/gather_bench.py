"""The window gather alone (fp32 rows / bf16 planes), B pairs: python tools/gather_bench.py [B]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpdist_amd import lib as L, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N, m, k = 64, 8, 5
pcA, pcB, _ = synth.s2_modelnet_shaped(B, N, 100)
pts, q = ops.stack_clouds(torch.tensor(pcA, device=dev), torch.tensor(pcB, device=dev))
fv = ops.mfv3d_fwd(pts, m, 0.125)
lib = L.load()
KP = lib.dpd_padded_width(k)
Q, Qb = 2 * B * N, B * N
X = torch.empty(Q, KP, device=dev); mask = torch.empty(Q, device=dev); vox = torch.empty(Q, device=dev, dtype=torch.int32)


def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


f32 = lambda: lib.dpd_patch_rows_fwd(L.ptr(q), L.ptr(fv), 2 * B, N, m, k, KP, L.ptr(X), L.ptr(mask), L.ptr(vox), None, L.cur_stream())
print("fp32 rows   B=%d: %6.1f us  (%.1f MB written)" % (B, timeit(f32), Q * KP * 4 / 1e6))
for dt, name in ((2, "bf16 planes"), (1, "3 planes")):
    nbytes = lib.dpd_planes_bytes(Q, Qb, KP, 1024, dt, 0)
    mem = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    pl = L.Planes()
    L.check(lib.dpd_planes_carve(L.ptr(mem), nbytes, Q, Qb, KP, 1024, dt, 0, pl), "carve")
    fn = lambda: lib.dpd_patch_rows_fwd(L.ptr(q), L.ptr(fv), 2 * B, N, m, k, KP, None, L.ptr(mask), L.ptr(vox), pl, L.cur_stream())
    npl = 1 if dt == 2 else 3
    print("%-11s B=%d: %6.1f us  (%.1f MB written)" % (name, B, timeit(fn), npl * (Q + Qb) * KP * 2 / 1e6))
    keep = (pl.X_rc, pl.X_r8)
    pl.X_r8 = None
    print("   RC only : %6.1f us  (%.1f MB)" % (timeit(fn), npl * Q * KP * 2 / 1e6))
    pl.X_rc, pl.X_r8 = None, keep[1]
    print("   R8 only : %6.1f us  (%.1f MB)" % (timeit(fn), npl * Qb * KP * 2 / 1e6))
    pl.X_rc, pl.X_r8 = keep
