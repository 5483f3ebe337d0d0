import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpdist_amd import synth
from dpdist_amd.model import DPDistParams
from dpdist_amd.trainer import DPDistTrainer
dev = torch.device('cuda:0')
B = 32
a, b, l = (torch.tensor(x, device=dev) for x in synth.s2_modelnet_shaped(B, 64, 100))

class FakeReducer:
    """same call pattern as BucketReducer, different things done at each reduce_async"""
    def __init__(self, mode):
        self.kind, self.grad_scale = mode, 1.0
        self.active, self.backend, self.mode = True, 'fake', 'allreduce'      # the reducer contract the trainer reads (ddp.py)
        self.side = torch.cuda.Stream()
        from dpdist_amd.hipevents import LightEvent
        light = mode.startswith('light')
        self.kind = mode = mode.replace('light-', '')
        self.evs = [LightEvent() if light else torch.cuda.Event() for _ in range(8)]
        self.light = light
        self.i = 0
    def reduce_async(self, bucket, upto=None):
        if self.kind == 'nothing':
            return
        ev = self.evs[self.i % 8]; self.i += 1
        ev.record()                                   # event on the compute stream
        if self.kind == 'record+sidewait':
            ev.wait(self.side) if self.light else self.side.wait_event(ev)
    def wait(self):
        if self.kind == 'record+sidewait':
            ev = self.evs[self.i % 8]; self.i += 1
            ev.record(self.side)
            ev.wait() if self.light else torch.cuda.current_stream().wait_event(ev)

for mode in ['nothing', 'record', 'record+sidewait', 'light-record', 'light-record+sidewait', 'nothing']:
    P = DPDistParams(device=dev); P.reset_parameters_tf(generator=torch.Generator().manual_seed(1))
    tr = DPDistTrainer(P, B, distributed=False)
    tr.reducer = FakeReducer(mode)
    for _ in range(60): tr.step(a, b, l)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): tr.step(a, b, l)
    torch.cuda.synchronize()
    print(tr.reducer.kind, 'light' if getattr(tr.reducer, 'light', False) else '', '%.4f ms/step' % ((time.perf_counter() - t0) / 200 * 1e3))
