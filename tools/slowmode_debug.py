"""Reproduce/diagnose the sporadic slow third trainer seen in bench.py's other_compute_types pass."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpdist_amd import lib, synth
from dpdist_amd.model import DPDistParams
from dpdist_amd.trainer import DPDistTrainer
L = lib.load()
dev = torch.device("cuda:0")
B, N = 32, 64
pcA, pcB, lab = (torch.tensor(x, device=dev) for x in synth.s2_modelnet_shaped(B, N, 100))

def mk(dt):
    P = DPDistParams(device=dev, compute_dtype=dt); P.reset_parameters_tf(generator=torch.Generator().manual_seed(1234))
    return DPDistTrainer(P, B, distributed=False)

def run(tr, n):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): tr.step(pcA, pcB, lab)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

def segments(tr):
    names = ["front", "decode", "backward", "adam"]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    tot = [0.0] * 4
    for _ in range(20):
        ev[0].record(); tr._front(pcA, pcB, None); ev[1].record(); tr._decode(); ev[2].record()
        tr.backward(lab.reshape(-1)); ev[3].record(); tr.apply_gradients(); ev[4].record()
        torch.cuda.synchronize()
        for i in range(4): tot[i] += ev[i].elapsed_time(ev[i + 1]) / 20
    return dict(zip(names, [round(x, 3) for x in tot]))

tr = mk("f32"); print("f32", run(tr, 10), run(tr, 50))
if "noprof" not in sys.argv:
    L.dpd_prof_enable(1)
    for _ in range(50):
        tr.forward(); tr.backward(lab.reshape(-1))
    torch.cuda.synchronize()
    ms, fl = ctypes.c_double(0), ctypes.c_double(0)
    print("prof launches", L.dpd_prof_collect(ctypes.byref(ms), ctypes.byref(fl))); L.dpd_prof_enable(0)
print("mem MB", torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20)
for dt in ("f32x3", "bf16", "bf16", "f32x3"):
    t2 = mk(dt); a, b = run(t2, 10), run(t2, 50)
    print(dt, round(a, 4), round(b, 4), "mem MB", torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20,
          segments(t2) if b > 1.0 else "")
    if "keep" not in sys.argv: del t2
