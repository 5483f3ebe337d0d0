"""Milestones of the window gather into planes (patch_rows_planes3_kernel; s_memtime of thread 0 of every workgroup;
ablation build: DPD_ABLATIONS=1 python -m dpdist_amd.build --force).   python tools/gather_stamps.py [B] [dtype: bf16 | f32x3]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpdist_amd import lib as L, synth  # noqa: E402
from dpdist_amd.model import DPDistParams  # noqa: E402
from dpdist_amd.trainer import DPDistTrainer  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
P = DPDistParams(device=dev, compute_dtype=dt)
P.load_tf_state_dict(synth.make_weights("wide"))
tr = DPDistTrainer(P, B, 64)
batch = [torch.tensor(x, device=dev) for x in synth.s2_modelnet_shaped(B, 64, 100)]
for _ in range(5):
    tr.step(*batch)
torch.cuda.synchronize()
lib = L.load()
buf = (ctypes.c_ulonglong * (1024 * 8))()
f = lib.dpd_debug_pr_stamps
f.argtypes = [ctypes.c_void_p]
assert f(buf) == 0
n = min(1024, 2 * B * 64 // 8)
st = np.array(buf, dtype=np.uint64).reshape(1024, 8)[:n, :5].astype(np.int64)
st = st[st[:, 4] > st[:, 0]]            # workgroups that wrote R8 chunks (the gradient-carrying half)
d = np.diff(st, axis=1)
names = ["loads requested, unit table, row info, barrier", "cloud vector -> planes in LDS, barrier", "pass A: RC planes (wave = row)", "pass B: R8 chunks (work item = column) + store drain"]
print("B=%d %s, %d workgroups with R8; cycles per section (median / max), total median %d" % (B, dt, len(st), np.median(st[:, 4] - st[:, 0])))
for i, nm in enumerate(names):
    print("  %-46s %7.0f / %7.0f" % (nm, np.median(d[:, i]), d[:, i].max()))
