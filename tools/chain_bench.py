"""A/B of the chained persistent launches (gemm_chain_kernel) against the separate launches, bf16, and the s_memtime stamps of the
chained launches' workgroups (dpd_set_chain_stamps).   python tools/chain_bench.py [B] [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dpdist_amd import lib as L, ops, synth  # noqa: E402
from dpdist_amd.model import DPDistParams  # noqa: E402
from dpdist_amd.trainer import DPDistTrainer  # noqa: E402


def timed(fn, n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = torch.device("cuda:0")
    pcA, pcB, lab = (torch.tensor(x, device=dev) for x in synth.s2_modelnet_shaped(B, 64, 100))
    P = DPDistParams(device=dev, compute_dtype="bf16")
    P.load_tf_state_dict(synth.make_weights("xavier_tf"))
    tr = DPDistTrainer(P, B, base_lr=1e-4, distributed=False)
    lab = lab.reshape(-1)
    tr._take_front(pcA, pcB, None)

    def fwd():
        tr._decode(skip_out=True)

    def fwd_bwd():
        tr._decode(skip_out=True)
        tr.backward(lab)

    def step():
        tr.step(pcA, pcB, lab)

    for name, f, t21, t23 in (("fwd chain (3 GEMMs)", fwd, 48, None), ("fwd + backward", fwd_bwd, None, None), ("whole step", step, None, None)):
        modes = ((0, 0), (1, 0), (0, 1), (1, 1))
        rows = [[] for _ in modes]
        for rnd in range(5):               # interleaved rounds: the clock ramp of the box hits every mode alike
            for i, mode in enumerate(modes):
                ops.set_gemm_plan(48, mode[0], 0)
                ops.set_gemm_plan(49, mode[1], 0)
                timed(f, 20)
                rows[i].append(timed(f, steps))
        row = [sorted(r)[len(r) // 2] for r in rows]
        print("%-22s B=%d  apart %.1f us | fwd chained %.1f | bwd chained %.1f | both %.1f   (median of 5 interleaved rounds)" % (name, B, *row))
    ops.set_gemm_plan(48, 1, 0)
    ops.set_gemm_plan(49, 1, 0)   # (the stamps below are those of the chained launches; the library's default is 0 = apart)
    # stamps of one chained forward and one chained backward
    st = torch.zeros(256 * 4 * 8, device=dev, dtype=torch.int64)
    for what, f in (("forward", fwd), ("fwd+bwd (last launch = dH chain)", fwd_bwd)):
        L.load().dpd_set_chain_stamps(st.data_ptr())
        st.zero_()
        f()
        torch.cuda.synchronize()
        L.load().dpd_set_chain_stamps(None)
        s = st.view(256, 4, 8).cpu().double()
        t0 = s[:, 0, 0][s[:, 0, 0] > 0].min()
        print("stamps (%s), us after the first workgroup's start (s_memrealtime, 100 MHz); median [min, max] over workgroups" % what)
        for it in range(4):
            if (s[:, it, 0] > 0).sum() == 0:
                continue
            live = s[:, it, 0] > 0
            cols = []
            for i, nm in enumerate(("ticket", "dep done", "tile done", "published")):
                v = (s[live, it, i] - t0) / 100.0
                cols.append("%s %.1f [%.1f, %.1f]" % (nm, v.median().item(), v.min().item(), v.max().item()))
            print("  tile %d (%d wgs): %s" % (it, int(live.sum()), " | ".join(cols)))
    print("sync status", L.load().dpd_planes_sync_status(tr._planes, L.cur_stream()))


if __name__ == "__main__":
    main()
