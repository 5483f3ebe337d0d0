"""Milestones of the 3DmFV forward kernel (s_memtime of thread 0 of every workgroup; ablation build).  python tools/mfv_stamps.py [B]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpdist_amd import lib as L, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pcA, pcB, _ = synth.s2_modelnet_shaped(B, 64, 100)
a, b = torch.tensor(pcA, device=dev), torch.tensor(pcB, device=dev)
for _ in range(5):
    pts, X, mask, vox = ops.front_end(a, b, None, 8, 0.125, 5)
torch.cuda.synchronize()
lib = L.load()
buf = (ctypes.c_ulonglong * (1024 * 8))()
f = lib.dpd_debug_mfv_stamps
f.argtypes = [ctypes.c_void_p]
assert f(buf) == 0
n = min(1024, 2 * B * 4)
full = np.array(buf, dtype=np.uint64).reshape(1024, 8)[:n].astype(np.int64)
st = full[:, :5]
d = np.diff(st, axis=1)
names = ["tables (z, exp) + stack", "row sums S + normalise + 0/0 check", "statistics loop + merge + stage", "store + slice norms"]
print("B=%d, %d workgroups; cycles per section (median / max over workgroups), total median %d" % (B, n, np.median(st[:, 4] - st[:, 0])))
for i, nm in enumerate(names):
    print("  %-34s %7.0f / %7.0f" % (nm, np.median(d[:, i]), d[:, i].max()))
if full[:, 5].max() > 0:      # finer stamps inside the statistics section (wave 0's only pass): point loop | merge | power norm + stage
    print("  statistics section: point loop %7.0f, merge of the point groups %7.0f, scale + stage %7.0f, barrier + power norm + barrier %7.0f" % (
        np.median(full[:, 5] - full[:, 2]), np.median(full[:, 6] - full[:, 5]), np.median(full[:, 7] - full[:, 6]), np.median(full[:, 3] - full[:, 7])))
