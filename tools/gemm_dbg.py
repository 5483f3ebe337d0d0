import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpdist_amd import lib as L
lib = L.load()
lib.dpd_gemm_f32_dbg.argtypes = [ctypes.c_int]*3 + [ctypes.c_void_p]*4 + [ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
M, N, K = 4096, 1024, 2528
A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
C = torch.empty(M, N, device=dev)
dbg = torch.zeros(256 * 16 * 4, device=dev)
for _ in range(3):
    L.check(lib.dpd_gemm_f32_dbg(M, N, K, A.data_ptr(), B.data_ptr(), C.data_ptr(), dbg.data_ptr(), 325, L.cur_stream()), "dbg")
torch.cuda.synchronize()
d = dbg.view(256, 16, 4).cpu()
tot, wt, bar, nt = d[..., 0], d[..., 1], d[..., 2], d[..., 3]
print("nt", nt[0, 0].item(), "total cycles/wave mean %.0f  per tile %.0f" % (tot.mean(), tot.mean() / nt[0, 0]))
print("vmcnt wait: mean %.0f (%.1f%%)  max-wave %.0f   per tile %.0f" % (wt.mean(), 100 * wt.mean() / tot.mean(), wt.max(), wt.mean() / nt[0, 0]))
print("barrier   : mean %.0f (%.1f%%)  per tile %.0f" % (bar.mean(), 100 * bar.mean() / tot.mean(), bar.mean() / nt[0, 0]))
print("per-wave vmcnt wait in block 0:", [int(x) for x in wt[0]])
print("per-wave barrier wait in block 0:", [int(x) for x in bar[0]])
