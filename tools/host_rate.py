"""How long does the HOST need to enqueue one training step (Python + ctypes + launches), next to the GPU time per step?
    python tools/host_rate.py [f32|bf16|f32x3] [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpdist_amd import synth  # noqa: E402
from dpdist_amd.model import DPDistParams  # noqa: E402
from dpdist_amd.trainer import DPDistTrainer  # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
P = DPDistParams(device=dev, compute_dtype=dt)
P.reset_parameters_tf(generator=torch.Generator().manual_seed(1))
tr = DPDistTrainer(P, B, distributed=False)
a, b, l = (torch.tensor(x, device=dev) for x in synth.s2_modelnet_shaped(B, 64, 100))
for _ in range(60):
    tr.step(a, b, l)
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    tr.step(a, b, l)
t1 = time.perf_counter()          # everything enqueued (the queue may throttle the host when it is full)
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s B=%d: host enqueue %.3f ms/step, total %.3f ms/step (GPU-bound if enqueue << total)" % (dt, B, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
# host-only cost: the same calls with the GPU kept trivially busy cannot be separated; instead time a step right after a sync
ts = []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(a, b, l)
    ts.append(time.perf_counter() - t0)
print("   host time of one step into an EMPTY queue: median %.3f ms" % (sorted(ts)[len(ts) // 2] * 1e3))
