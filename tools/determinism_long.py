"""Run-to-run reproducibility of the training step over MANY steps and several distinct batches (tools/determinism_check.py looks at six steps
of one batch): two trainers from the same weights, the same batch sequence; reports the first step whose loss differs.
    python tools/determinism_long.py [dtype] [steps] [chair]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dpdist_amd import synth  # noqa: E402
from dpdist_amd.model import DPDistParams  # noqa: E402
from dpdist_amd.trainer import DPDistTrainer  # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
shapes = "chair" if len(sys.argv) > 3 else None
dev = torch.device("cuda:0")
B = 32
kw = {"shapes": "chair"} if shapes else {}
pool = [tuple(torch.tensor(x, device=dev) for x in synth.s2_modelnet_shaped(B, 64, 5000 + i, **kw)) for i in range(16)]


def run():
    P = DPDistParams(device=dev, compute_dtype=dt)
    P.reset_parameters_tf(generator=torch.Generator().manual_seed(1234))
    tr = DPDistTrainer(P, B, base_lr=1e-4, distributed=False)
    losses = []
    for s in range(steps):
        losses.append(tr.step(*pool[s % len(pool)]).clone())
    torch.cuda.synchronize()
    return torch.stack(losses).cpu(), P.flat.detach().clone().cpu()


a, wa = run()
b, wb = run()
diff = (a != b).any(dim=1).nonzero().flatten()
print("%s, %d steps over %d batches: %s" % (dt, steps, len(pool), "bitwise identical losses and weights" if len(diff) == 0 and torch.equal(wa, wb) else
      "FIRST DIFFERENCE at step %d: %s vs %s; weights equal: %s" % (int(diff[0]) if len(diff) else -1, a[int(diff[0])].tolist() if len(diff) else None,
                                                                   b[int(diff[0])].tolist() if len(diff) else None, torch.equal(wa, wb))))
