"""Checksums of one training step's device state (operand planes, gradients, losses, weights after Adam) per compute type: run before and
after a change that must not move a bit.   python tools/step_checksum.py"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dpdist_amd import synth  # noqa: E402
from dpdist_amd.model import DPDistParams  # noqa: E402
from dpdist_amd.trainer import DPDistTrainer  # noqa: E402


def h(t):
    return hashlib.sha1(t.detach().contiguous().view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:16]


dev = torch.device("cuda:0")
for dt, B in (("bf16", 64), ("bf16", 32), ("f32x3", 32), ("f32", 32), ("bf16", 3)):
    pcA, pcB, lab = (torch.tensor(x, device=dev) for x in synth.s2_modelnet_shaped(B, 64, 100))
    P = DPDistParams(device=dev, compute_dtype=dt)
    P.load_tf_state_dict(synth.make_weights("wide"))
    tr = DPDistTrainer(P, B, base_lr=1e-3, distributed=False)
    tr._take_front(pcA, pcB, None)
    tr._decode(skip_out=True)
    tr.backward(lab.reshape(-1))
    torch.cuda.synchronize()
    g = tr.grad.clone()
    # the bias gradients of the plane types are fp32 atomics (order-dependent): hash the matrices only there
    gm = torch.cat([P.view(n, g).reshape(-1) for n in ("W1p", "W2", "W3", "W4")])
    out = {"planes": h(tr._plane_mem[:-4096]) if tr._planes is not None else "-", "dW": h(gm), "loss": h(tr.loss), "pred": h(tr.pred), "y": h(tr.y),
           "dy": h(tr.dy)}
    print(dt, B, " ".join("%s=%s" % kv for kv in out.items()))
