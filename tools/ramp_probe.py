import os, time, torch, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpdist_amd import synth
from dpdist_amd.model import DPDistParams
from dpdist_amd.trainer import DPDistTrainer
dev = torch.device('cuda:0')
B = 32
P = DPDistParams(device=dev); P.reset_parameters_tf(generator=torch.Generator().manual_seed(1))
tr = DPDistTrainer(P, B, distributed=False)
a, b, l = (torch.tensor(x, device=dev) for x in synth.s2_modelnet_shaped(B, 64, 100))
torch.cuda.synchronize()
ts = []
for blk in range(16):
    t0 = time.perf_counter()
    for _ in range(5):
        tr.step(a, b, l)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 5 * 1e3)
print(' '.join('%.3f' % t for t in ts))
time.sleep(2.0)
ts = []
for blk in range(8):
    t0 = time.perf_counter()
    for _ in range(5):
        tr.step(a, b, l)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 5 * 1e3)
print('after 2 s idle:', ' '.join('%.3f' % t for t in ts))
