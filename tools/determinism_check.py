"""Run the fp32 trainer twice from the same weights/batch and report where the parameter buffers first differ."""
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dpdist_amd import synth  # noqa: E402
from dpdist_amd.model import DPDistParams  # noqa: E402
from dpdist_amd.trainer import DPDistTrainer  # noqa: E402

dev = torch.device("cuda:0")
B = 32
pcA, pcB, lab = (torch.tensor(x, device=dev) for x in synth.s2_modelnet_shaped(B, 64, 100))
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
runs = []
for r in range(2):
    g = torch.Generator().manual_seed(1234)
    P = DPDistParams(device=dev, init=None, compute_dtype=dt)
    P.reset_parameters_tf(generator=g)
    tr = DPDistTrainer(P, B, distributed=False)
    snaps = []
    for t in range(6):
        tr.step(pcA, pcB, lab)
        torch.cuda.synchronize()
        snaps.append((P.flat.detach().clone(), tr.grad.clone(), tr.loss.clone()))
    runs.append(snaps)
names = list(P._segments.items())
for t in range(6):
    (w0, g0, l0), (w1, g1, l1) = runs[0][t], runs[1][t]
    msg = []
    for n, (off, cnt, shp) in names:
        dg = (g0[off:off + cnt] - g1[off:off + cnt]).abs().max().item()
        dw = (w0[off:off + cnt] - w1[off:off + cnt]).abs().max().item()
        if dg or dw:
            msg.append("%s dgrad %.2e (|g| %.2e) dW %.2e" % (n, dg, g0[off:off + cnt].abs().max().item(), dw))
    print("step", t + 1, "loss", l0.tolist(), l1.tolist(), "|", "; ".join(msg) if msg else "bitwise identical")
