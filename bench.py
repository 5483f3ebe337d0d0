#!/usr/bin/env python3
"""DPDist hot-path benchmark: query-points/sec of the full training step (fwd + bwd + Adam) on N MI355X GPUs.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: B pairs of 64-point clouds per GPU
(BASELINE.json metric: "query-points/sec (DPDist fwd+bwd), 64-pt patches K=5^3, batch 32"): 3DmFV encode of
2B clouds, 5^3-window gather + shared-MLP decoder for 2*B*64 query points, L1 loss on the AB half, backward to the
8 decoder variables, gradient all-reduce (N>1) and the TF-form Adam update.  Inputs are resident in HBM before the
timed region.  value = 2*B*64 * N * K / max-over-ranks(time).  Weak scaling: per-GPU batch fixed.

`python bench.py --gpus N` with no launcher starts its own N ranks (one process per GPU, RCCL).

Order of one run: auxiliary legs (other compute types, config 3/4; their buffers stay allocated) -> device spin-up (`spinup_ms`
of scratch fp32 GEMMs: an MI355X needs ~25 ms of sustained load to reach its steady clock, tools/ramp_probe.py) -> W untimed
warm-up steps -> K timed steps between barrier + synchronize -> profiler pass (roofline) -> CPU baseline -> ONE JSON line (the
last line of stdout).

Extra objects on the JSON line:
  roofline     -- the fp32 MFMA GEMM kernel family (gemm_rs_kernel<...> + the LDS-ring fallback): algorithmic flops of the GEMM
                  launches of a step / their summed duration, measured with hipEvent pairs recorded in-stream around each
                  launch (library profiler, separate pass of the same K steps); peak = 157.3 TFLOP/s fp32 MFMA.
  roofline_hbm -- the bandwidth-bound kernels (3DmFV encoder, window gather, fused output layer, optimizer, small-gradient reduction,
                  weight copies): algorithmic HBM bytes per launch / in-stream launch duration (dpd_prof_enable(2)) as GB/s and fraction of
                  the 8 TB/s peak, next to the PMC fabric bytes of the committed rocprofv3 summary; also inside config3 for its step.
  dp           -- (N > 1 or DPD_FORCE_DIST=1) the data-parallel plumbing: collective backend, fallback flag and the watchdog's record,
                  ranks as ncclCommCount reports them, start-up cross-check, bytes on the wire per GPU per step, exposed communication.
  cpu_baseline -- the CPU ports (oracle/cpu_ref.c faithful + compact, oracle/restate.py on torch-CPU), each pinned to the
                  reference's goldens, timed on this box's host cores on bounded samples (N = 1 only).
  config3      -- (N = 1) BASELINE config 3: the same step in bf16 at 64 pairs per GPU.
  config4      -- (N > 1) BASELINE config 4: bf16, 64 pairs per GPU (512 global at N = 8), gradients all-reduced over RCCL, next
                  to the same step on rank 0 alone WITHOUT the collectives (`n1_same_run`, the weak-scaling reference).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# torch is imported inside main() / the helpers that need it: the watchdog supervisor of an N > 1 run (dpdist_amd/launch.py) is this
# same file and must start in milliseconds without touching a GPU runtime
torch = dist = None

PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD


def gemm_flops_per_step(B, N, E3, H):
    """Algorithmic flops of the GEMM launches of one training step (true K = 2503, layer 4 excluded:
    it is not a GEMM launch).  fwd: Q rows x (E3*H + 2*H*H); bwd on BN rows: dH chain 2*H*H, dW E3*H + 2*H*H."""
    Q, BN = 2 * B * N, B * N
    fwd = 2.0 * Q * (E3 * H + 2 * H * H)
    bwd = 2.0 * BN * (2 * H * H) + 2.0 * BN * (E3 * H + 2 * H * H)
    return fwd + bwd, 3 + 2 + 3


def gemm_bytes_per_step(B, N, KP, H, dtype="f32"):
    """Algorithmic HBM bytes of the same GEMMs: every operand read once, every result written once, in the compute type's own
    storage -- fp32 (4 B); bf16 planes: 2 B per element per plane (f32x3: three planes); the plane types write an activation /
    pre-activation gradient once per layout its consumers read (RC for the next layer and the dH chain, R8 for the weight gradient)
    and read the ReLU gate as a plane; weight gradients leave as fp32 in every type."""
    Q, BN = 2 * B * N, B * N
    if dtype == "f32":
        fwd = 4.0 * ((Q * KP + KP * H + Q * H) + 2 * (Q * H + H * H + Q * H))
        dh = 4.0 * 2 * (BN * H + H * H + BN * H + BN * H)
        dw = 4.0 * ((BN * KP + BN * H + KP * H) + 2 * (BN * H + BN * H + H * H))
        return fwd + dh + dw
    e = 2.0 * (3 if dtype == "f32x3" else 1)          # bytes per element of an operand (all planes)
    h_out = e * (Q + BN) * H                          # h1 / h2: RC plane for all rows + R8 plane for the gradient rows
    h3_out = (2.0 if dtype == "bf16" else 4.0) * Q * H
    fwd = e * (Q * KP + KP * H) + h_out + e * (Q * H + H * H) + h_out + e * (Q * H + H * H) + h3_out
    dh = 2 * (e * (BN * H + H * H) + e * BN * H + 2 * e * BN * H)        # g in, W in, gate in, g out in both layouts
    dw = e * (BN * KP + BN * H) + 4.0 * KP * H + 2 * (e * (BN * H + BN * H) + 4.0 * H * H)
    return fwd + dh + dw


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_quota():
    """CPUs this container may use: the cgroup v2 quota (cpu.max = "<quota> <period>") when there is one, else the affinity mask.
    The GPU boxes report 256 hardware threads but run the container under a 16-CPU quota: more OpenMP threads than that only thrash."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(round(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def _c_port():
    """oracle/cpu_ref.c (the plain-C restatement, pinned against the reference's goldens in tests/test_oracle_c.py) compiled
    for THIS host (-march=native) into a temp dir; falls back to the portable AVX2 build that travelled with the repo."""
    import subprocess
    import tempfile
    src = os.path.join(ROOT, "oracle", "cpu_ref.c")
    out = os.path.join(tempfile.gettempdir(), "libdpd_cpuref_native_%d.so" % os.getpid())
    try:
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-fopenmp", "-ffp-contract=off", "-shared", "-w", "-o", out, src, "-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lib, how = ctypes.CDLL(out), "gcc -O3 -march=native"
    except Exception:
        lib, how = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libdpd_cpuref.so")), "prebuilt -mavx2 -mfma"
    fp = ctypes.POINTER(ctypes.c_float)
    lib.cpuref_train_step.argtypes = [fp] * 4 + [ctypes.c_int] * 4 + [ctypes.c_float] + [fp] * 8 + [ctypes.c_int] * 3 + [fp] * 11
    return lib, how


def cpu_baseline(B, N, budget_s=26.0):
    """The CPU stand-in for "the reference TF1 CPU path on this box's host cores" (BASELINE.md section 3): fwd + bwd of the same
    S2 workload, bounded samples, query-points/sec.  Three ports, every one pinned to the reference's goldens:
      c_faithful  oracle/cpu_ref.c, the TF graph's dataflow (materialised [C,512,2500] window tensor, all-centres mask + argmax)
      c_compact   oracle/cpu_ref.c without the window tensor
      torch       oracle/restate.py on torch-CPU (oneDNN/MKL GEMMs)
    each at 1 thread and at the best of a few thread counts.  `value` = the best number of all (cores = its thread count)."""
    import numpy as np
    import torch
    from dpdist_amd import synth
    from oracle import restate as R
    ncpu = os.cpu_count() or 1
    quota = min(ncpu, _cpu_quota())
    t_all = time.perf_counter()
    variants = {}
    pcA, pcB, lab = synth.s2_modelnet_shaped(B, N, 100)
    Wnp = synth.make_weights("xavier_tf")

    # ---- the C port
    try:
        lib, how = _c_port()
        omp = ctypes.CDLL("libgomp.so.1")
        fp = ctypes.POINTER(ctypes.c_float)
        ptr = lambda x: x.ctypes.data_as(fp)   # noqa: E731
        nm = "pc_compare/dpdist_local/mapper_conv%d/%s"
        ws = [np.ascontiguousarray(Wnp[nm % (l, t)].reshape(-1, Wnp[nm % (l, t)].shape[-1]) if t == "weights" else Wnp[nm % (l, t)], np.float32)
              for l in (1, 2, 3, 4) for t in ("weights", "biases")]
        loss = np.zeros(2, np.float32)

        def c_step(variant, Bs):
            lib.cpuref_train_step(ptr(pcA[:Bs].copy()), ptr(pcB[:Bs].copy()), None, ptr(lab[:Bs].copy()), Bs, N, 8, 5, 0.125,
                                  *[ptr(w) for w in ws], 1024, variant, 1, ptr(loss), *([None] * 10))

        def timed(fn, max_s, max_n=6):
            fn()                      # page-in
            t0, n = time.perf_counter(), 0
            while True:
                fn(); n += 1
                el = time.perf_counter() - t0
                if el >= max_s or n >= max_n:
                    return n, el

        # compact port: sweep the thread count (the dense layers are a packed, cache-blocked SGEMM: they scale until the memory system
        # does not); then the faithful dataflow at the compact port's best count; one-thread numbers on a 4-pair sample
        best = None
        sweep = {}
        for nt in sorted({min(ncpu, c) for c in (max(2, quota // 2), quota, 2 * quota, 4 * quota)}):
            if nt < 2 or time.perf_counter() - t_all > budget_s * 0.45:
                continue
            omp.omp_set_num_threads(nt)
            n, el = timed(lambda: c_step(0, B), 0.6, 4)
            q = 2 * B * N * n / el
            sweep[nt] = round(q, 1)
            if best is None or q > best[0]:
                best = (q, nt, n, el)
        if best:
            variants["c_compact_best"] = {"value": round(best[0], 1), "threads": best[1], "sample": "%d steps at B=%d in %.1f s" % (best[2], B, best[3]),
                                          "by_threads": sweep}
            omp.omp_set_num_threads(best[1])
            n, el = timed(lambda: c_step(1, B), 0.8, 3)
            variants["c_faithful_best"] = {"value": round(2 * B * N * n / el, 1), "threads": best[1], "sample": "%d steps at B=%d in %.1f s" % (n, B, el)}
        omp.omp_set_num_threads(1)
        Bs = min(B, 4)                                      # one thread: a 4-pair sample
        for vname, vid in (("c_compact", 0), ("c_faithful", 1)):
            n, el = timed(lambda: c_step(vid, Bs), 0.8, 2)
            variants[vname + "_1t"] = {"value": round(2 * Bs * N * n / el, 1), "threads": 1, "sample": "%d steps at B=%d in %.1f s" % (n, Bs, el)}
        variants["c_build"] = how
    except Exception as e:   # the C port must never take the bench down
        variants["c_error"] = repr(e)

    # ---- the torch-CPU oracle
    W = R.as_torch_weights(Wnp, torch.float32, requires_grad=True)
    a, b, l = torch.tensor(pcA), torch.tensor(pcB), torch.tensor(lab)

    def one():
        pred, _ = R.get_model(a, b, W)
        ls, _ = R.get_loss(pred, l)
        torch.autograd.grad(ls, list(W.values()))

    best = None
    for nt in [1] + sorted({min(ncpu, c) for c in (max(2, quota // 2), quota, 2 * quota)}):
        if time.perf_counter() - t_all > budget_s and best is not None:
            break
        torch.set_num_threads(nt)
        one()
        t0, n = time.perf_counter(), 0
        while True:
            one(); n += 1
            el = time.perf_counter() - t0
            if el >= 1.2 or n >= 8:
                break
        q = 2 * B * N * n / el
        if nt == 1:
            variants["torch_1t"] = {"value": round(q, 1), "threads": 1, "sample": "%d steps at B=%d in %.1f s" % (n, B, el)}
        elif best is None or q > best[0]:
            best = (q, nt, n, el)
    if best:
        variants["torch_best"] = {"value": round(best[0], 1), "threads": best[1], "sample": "%d steps at B=%d in %.1f s" % (best[2], B, best[3])}
    top = max((v for v in variants.values() if isinstance(v, dict)), key=lambda v: v["value"])
    which = [k for k, v in variants.items() if v is top][0]
    return {"value": top["value"], "unit": "query-points/sec", "cores": top["threads"], "kind": "port",
            "sample": "fwd+bwd of the same S2 workload, best of the ports below: %s (%s); host: %s, %d hardware threads, cgroup quota %d CPUs; "
                      "total %.0f s of CPU timing" % (which, top["sample"], _cpu_model(), ncpu, quota, time.perf_counter() - t_all),
            "cpu_model": _cpu_model(), "nproc": ncpu, "cpu_quota": quota, "variants": variants}


def pmc_traffic(kernel_substr, suffix=""):
    """Fabric-side bytes per launch of a kernel family from the COMMITTED PMC summary of the same command
    (profiles/rNN_pmc_summary<suffix>.csv: separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, FETCH_SIZE doubled per
    MI355X_MICROARCH.md section HBM).  bench.py cannot collect counters itself; None when no summary is in the tree.
    kernel_substr: one substring or a tuple of alternatives; suffix: "" (f32 B=32), "_bf16_b64", "_bf16", "_f32x3"."""
    import csv
    import glob
    import re
    root = os.path.dirname(os.path.abspath(__file__))
    files = sorted(f for f in glob.glob(os.path.join(root, "profiles", "r*_pmc_summary%s.csv" % suffix))
                   if re.fullmatch(r"r\d+_pmc_summary%s\.csv" % re.escape(suffix), os.path.basename(f)))
    if not files:
        return None, None
    subs = (kernel_substr,) if isinstance(kernel_substr, str) else tuple(kernel_substr)
    tot, n = 0.0, 0
    for r in csv.DictReader(open(files[-1])):
        if any(x in r["kernel"] for x in subs) and r["calls_in_stats"] and r["fabric_read_MB_per_launch(x2 gfx950 correction)"]:
            calls = int(r["calls_in_stats"])
            rd = float(r["fabric_read_MB_per_launch(x2 gfx950 correction)"]) * 1e6
            wr = float(r["WRITE_SIZE_KB_per_launch"] or 0) * 1024.0
            tot += calls * (rd + wr)
            n += calls
    return (tot / n if n else None), os.path.relpath(files[-1], root)


def pmc_traffic_live(kernel_substr, bench_args, timeout_s=75.0):
    """Fabric-side bytes per launch of a kernel family, MEASURED NOW: two child runs of this file under `rocprofv3 --pmc FETCH_SIZE` /
    `--pmc WRITE_SIZE` (separate passes: the two do not fit one; --kernel-trace only, the combination MI355X_MICROARCH.md prescribes), a few
    steps each, no roofline / CPU legs in the children.  2 x FETCH_SIZE (gfx950 tallies 128-byte requests at 64 B) + WRITE_SIZE, launch-
    weighted over the family.  Returns (bytes per launch or None, note).  Never raises; bounded by timeout_s per pass."""
    import csv
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    subs = (kernel_substr,) if isinstance(kernel_substr, str) else tuple(kernel_substr)
    me = os.path.abspath(__file__)
    vals = {}
    tmp = tempfile.mkdtemp(prefix="dpd_pmc_", dir="/tmp")
    try:
        for cname in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, cname)
            cmd = [exe, "--pmc", cname, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable, me,
                   "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-other-dtypes", "--spinup-ms", "0"] + list(bench_args)
            env = dict(os.environ, TMPDIR="/tmp", DPD_WD="0")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DPD_FORCE_DIST"):
                env.pop(k, None)
            try:
                subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=False)
            except subprocess.TimeoutExpired:
                return None, "rocprofv3 --pmc %s did not finish in %.0f s" % (cname, timeout_s)
            hits = []
            for root_, _, files in os.walk(out):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        for r in csv.DictReader(open(os.path.join(root_, f))):
                            if r.get("Counter_Name") == cname and any(x in r.get("Kernel_Name", "") for x in subs):
                                hits.append(float(r["Counter_Value"]))
            if not hits:
                return None, "no %s rows for %s" % (cname, "/".join(subs))
            vals[cname] = sum(hits) / len(hits)          # KiB per launch, launch-weighted (one row per dispatch)
    except Exception as e:      # noqa: BLE001 -- a measurement aid must never take the bench line down
        return None, "live PMC pass failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return 2.0 * vals["FETCH_SIZE"] * 1024.0 + vals["WRITE_SIZE"] * 1024.0, (
        "MEASURED in this run: two child passes of this command under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (--kernel-trace only, 3 steps "
        "each); bytes per launch at the L2's fabric side (2 x FETCH_SIZE: gfx950 correction, + WRITE_SIZE; Infinity-Cache hits included), mean over "
        "the family's dispatches")


PEAK_HBM_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E, 8 TB/s
ACHIEVABLE_HBM_GBPS = 6300.0    # what a pure streaming kernel sustains on this chip (same guide; our Adam kernel reaches it)
# stage tag of the library's in-stream profiler (include/dpdist_capi.h: dpd_prof_collect_stage) -> (name, kernel-name substrings in the
# rocprofv3 summaries, what it replaces in the reference)
HBM_STAGES = {1: ("encoder", ("mfv3d_fwd", "mfv3d_norm_kernel"), "get_3dmfv_tf, utils/dpdist_util.py:22-141"),
              2: ("window_gather", ("patch_rows_",), "local_z_3d + mask + gather, utils/dpdist_util.py:434-492,911-930"),
              3: ("output_layer_fused", ("out_bwd_fused4_kernel",), "layer 4 + relu6/3 + mask + L1 loss + their backward, :540-544,690-698,962-980"),
              4: ("optimizer", ("adam_",), "tf.train.AdamOptimizer.apply_gradients, train_multi_gpu_pc_compare_dist.py:301"),
              5: ("small_grads_reduce", ("small_grads_reduce",), "db3 / dW4 / db4 block partials"),
              6: ("weight_copies", ("transpose_kernel", "split_planes_kernel"), "derived data: transposed fp32 copies / bf16 operand planes")}


def hbm_roofline_pass(L, step_fn, warmup, steps, pmc_suffix=""):
    """`roofline_hbm`: the bandwidth-bound kernels of the step, each against the HBM roofline -- ALGORITHMIC bytes per launch (recorded by
    the launch site: every input read once, every output written once) / average launch duration from hipEvent pairs recorded
    in-stream around the launch (dpd_prof_enable(2); separate pass of the same steps), as GB/s and as a fraction of the 8 TB/s peak
    and of the 6.3 TB/s a streaming kernel sustains; next to it the fabric-side bytes per launch of the same kernel from the
    committed rocprofv3 PMC summary (2 x FETCH_SIZE + WRITE_SIZE): traffic well above the algorithmic bytes = wasted re-reads."""
    import torch
    for it in range(warmup + steps):
        if it == warmup:
            torch.cuda.synchronize()
            L.dpd_prof_enable(2)
        step_fn()
    torch.cuda.synchronize()
    out = {}
    for tag, (name, kerns, what) in HBM_STAGES.items():
        ms_, by_ = ctypes.c_double(0), ctypes.c_double(0)
        n = L.dpd_prof_collect_stage(tag, ctypes.byref(ms_), ctypes.byref(by_))
        if n <= 0 or ms_.value <= 0:
            continue
        gbps = by_.value / (ms_.value * 1e-3) / 1e9
        traffic, src = pmc_traffic(kerns, pmc_suffix)
        out[name] = {"bound": "hbm", "kernel": " / ".join(kerns), "replaces": what, "launches_per_step": round(n / float(steps), 2),
                     "avg_launch_us": round(ms_.value * 1e3 / n, 2), "algorithmic_bytes_per_launch": round(by_.value / n),
                     "achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(gbps / PEAK_HBM_GBPS, 4),
                     "frac_of_achievable_6300": round(gbps / ACHIEVABLE_HBM_GBPS, 4),
                     "traffic": round(traffic) if traffic else None, "traffic_source": src}
        if name == "encoder":
            # 2.8-5.6 MB in 9-13 us: not a bandwidth kernel.  s_memtime stamps (profiles/r04_mfv_stamps.txt, B = 32): 20.6k cycles per
            # workgroup = tables 3.0k + row sums 2.8k + statistics loop / merge / power norm 11.0k (VALU issue: max / min statistics have no
            # packed form, four waves per SIMD share the VALU) + store 3.8k -- three dependent latency chains, no section waits on HBM
            out[name].update({"bound": "latency", "frac": None, "frac_of_achievable_6300": None,
                              "bound_note": "VALU-issue / latency bound: ~53 % of a workgroup's cycles are the statistics loop + merge + power "
                                            "normalisation on the VALU, the rest table set-up and an LDS-staged store (profiles/r04_mfv_stamps.txt); "
                                            "`achieved` GB/s is reported for completeness, the HBM roofline is not its yardstick"})
    L.dpd_prof_enable(0)
    if out:
        tot_us = sum(v["avg_launch_us"] * v["launches_per_step"] for v in out.values())
        out["_sum"] = {"us_per_step": round(tot_us, 1),
                       "note": "in-stream hipEvent pairs around each launch (separate pass; an event pair costs the stream ~2 us between "
                               "launches, not inside them); `traffic` is NOT measured in this run: committed PMC summary named in traffic_source"}
    return out


def profiled_gemm_pass(L, tr, lab, warmup, steps):
    """(launches, summed GEMM ms, flops) of `steps` forward + backward passes with the library's in-stream profiler on (hipEvent pairs
    around every GEMM launch on the stream the kernels run on); best of two passes (a host stall idles the GPU and drops its clock)."""
    import torch
    best = None
    for _pass in range(2):
        for it in range(warmup + steps):
            if it == warmup:
                torch.cuda.synchronize()
                L.dpd_prof_enable(1)
            tr.forward()
            tr.backward(lab.reshape(-1))
        torch.cuda.synchronize()
        ms_, fl_ = ctypes.c_double(0), ctypes.c_double(0)
        n_ = L.dpd_prof_collect(ctypes.byref(ms_), ctypes.byref(fl_))
        L.dpd_prof_enable(0)
        if n_ > 0 and ms_.value > 0 and (best is None or ms_.value < best[1]):
            best = (n_, ms_.value, fl_.value)
    return best


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU ourselves (the reference picks its towers with a
    flag inside one process, train_multi_gpu_pc_compare_dist.py:122-126,237-302; here: one process per GPU over RCCL, the same
    environment contract as `python -m torch.distributed.run --nproc-per-node N`).  Rank 0 prints the JSON line; the exit code
    is the first non-zero rank exit code.  Children are stopped by PID, never by pattern."""
    import torch
    if not torch.cuda.is_available() or (torch.cuda.device_count() < n and os.environ.get("DPD_TEST_SHARE_GPU") != "1"):
        sys.stderr.write("bench.py --gpus %d: only %d GPU(s) visible\n" % (n, torch.cuda.device_count() if torch.cuda.is_available() else 0))
        return 2
    return spawn_ranks(n, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])


def spawn_ranks(n, cmd, timeout_s=None):
    """Run `cmd` n times with RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set (rendezvous on 127.0.0.1, free port).  Every rank of
    bench.py supervises itself (per-phase watchdog + one retry on torch.distributed collectives: dpdist_amd/launch.py); the
    overall limit here (default 1500 s) is the backstop behind that: on expiry the ranks are stopped by PID and
    the code is 124."""
    import socket
    import subprocess
    if timeout_s is None:
        timeout_s = 1500.0
    t_start = time.time()
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    import tempfile
    wd_base = tempfile.mkdtemp(prefix="dpd_wd_")     # heartbeat directory of THIS launch (dpdist_amd/launch.py)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", DPD_WD_DIR_BASE=wd_base)
        env.setdefault("OMP_NUM_THREADS", "8")
        procs.append(subprocess.Popen(list(cmd), env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in procs:        # one rank died: the others would hang in the next collective
                        q.terminate()
            if procs and time.time() - t_start > timeout_s:
                sys.stderr.write("spawn_ranks: %d rank(s) still running after %.0f s, stopping them\n" % (len(procs), timeout_s))
                for q in procs:
                    q.terminate()
                time.sleep(3.0)
                rc = rc or 124
                break
            time.sleep(0.05)
    finally:
        for q in procs:
            q.kill()
        import shutil
        shutil.rmtree(wd_base, ignore_errors=True)
    return rc


DP_MODES = ("allreduce", "rs_ag", "zero1")      # communication forms the start-up measurement chooses among (bit-identical results)
# how much compute is still to run when a gradient bucket's collective is issued in the "early" order, as a fraction of the one-rank step
# (profiles/r05_step_timeline.txt: layers 2-4 are issued before the last data-gradient GEMM + dW1; layer 1 last, with only the next step's
# encoder + gather -- Adam runs on the collectives' stream -- to hide under), and the one-rank cost of the plumbing (profiles/r05_dp_single_rank.txt)
DP_WINDOWS = {"f32": ((35.7 + 91.2) / 561.0, 30.0, 15.0), "f32x3": ((42.0 + 71.6) / 520.0, 33.0, 15.0), "bf16": (45.0 / 278.0, 35.0, 7.0)}


def dp_model(step_ms_n1, dtype, bucket_bounds, wire="f32"):
    """`dp.model`: what the data-parallel step should cost at N = 2 / 4 / 8 given THIS run's one-rank step time (dpdist_amd/ddp.py:
    predict_scaling; assumptions inside the record).  No multi-GPU box has been available: this is the expectation a hardware curve is
    read against."""
    from dpdist_amd import ddp
    frac, late_us, plumb = DP_WINDOWS[dtype]
    bb = list(bucket_bounds)
    early = 4 * (bb[-1] - bb[1])                 # layers 2-4 (issued first), then layer 1
    late = 4 * (bb[1] - bb[0])
    return ddp.predict_scaling(step_ms_n1, [early, late], [frac * step_ms_n1 * 1e3, late_us], plumbing_us=plumb, wire=wire)


def section8d_legs(L, tr, pcA, pcB, lab, dev, steps, warmup):
    """SURVEY 8(d)'s other two figures, in the driver-timed line (N = 1 only; run BEFORE the headline region like config 3):
      fwd_only   eval_one_epoch_3d (train_multi_gpu_pc_compare_dist.py:809-873): encoder + window gather + decoder + loss, both directions,
                 no backward -- the headline's trainer, batch and compute type;
      as_loss    DPDist as a frozen loss (pcrnet-registration/iterative_PCRNet_ours.py:248-257): loss_pred of (source, template) and its
                 gradient w.r.t. both clouds through the as-loss engine, forward + backward and forward only, B = 16 (the registration
                 batch) and 32, f32 and bf16.
    Each leg: ms per evaluation (wall clock over `n` back-to-back evaluations between synchronisations, after a spin-up of its own),
    query-points/s, and from a separate profiled pass (hipEvent pairs around every GEMM launch) the GEMM time per evaluation, its share of
    the evaluation and the GEMM family's fraction of the matrix-core peak of its type."""
    import torch
    from dpdist_amd import synth
    from dpdist_amd.model import DPDistLoss, DPDistModel
    N = 64
    FWD_FLOP = 2.0 * (2503 * 1024 + 2 * 1024 * 1024)            # layers 1-3 per query point (the 1024 x 3 output layer is not a GEMM launch)
    out = {}

    def timed(fn, n, spin_ms=25.0):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < spin_ms:       # this leg's own clock spin-up
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
        best = float("inf")
        for _rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / n * 1e3)
        return best

    def gemm_pass(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        L.dpd_prof_enable(1)
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ms_, fl_ = ctypes.c_double(0), ctypes.c_double(0)
        cnt = L.dpd_prof_collect(ctypes.byref(ms_), ctypes.byref(fl_))
        L.dpd_prof_enable(0)
        return (cnt / float(n), ms_.value / n) if cnt > 0 else (0, 0.0)

    def record(ms, Q, flop_per_q, peak, fn, n):
        launches, gms = gemm_pass(fn, n)
        r = {"ms_per_eval": round(ms, 4), "value": round(Q / (ms * 1e-3), 1), "unit": "query-points/sec"}
        if gms > 0:
            tf = Q * flop_per_q / (gms * 1e-3) / 1e12
            r.update({"gemm_launches": round(launches, 1), "gemm_ms": round(gms, 4), "gemm_share_of_eval": round(gms / ms, 3),
                      "gemm_frac_of_peak": round(tf / peak, 4), "gemm_tflops": round(tf, 1)})
        return r

    def graph_replay_ms(loss_fn, src, tmpl, n):
        from dpdist_amd import asloss
        try:
            pool = {}
            with asloss.private_pool(pool):
                for _ in range(3):
                    torch.autograd.grad(loss_fn(src, tmpl), src)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    torch.autograd.grad(loss_fn(src, tmpl), src)
            for _ in range(10):
                g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(n):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            del g, pool
            return round(ms, 4)
        except Exception as e:      # reported evidence only: never fails the line
            return "capture failed: %s" % (str(e).splitlines()[0][:120] if str(e) else type(e).__name__)

    n = max(steps, 100)         # (sub-millisecond evaluations: 20 of them are 4 ms of wall clock, too few for a steady figure)
    # forward only on the headline trainer
    B = tr.B
    ms = timed(lambda: tr.evaluate(pcA, pcB, lab), n)
    peak = PEAK_FP32_MFMA_TFLOPS if tr.P.compute_dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
    out["fwd_only"] = dict(record(ms, 2 * B * N, FWD_FLOP * (6 if tr.P.compute_dtype == "f32x3" else 1), peak, lambda: tr.evaluate(pcA, pcB, lab), n),
                           what="forward only (eval_one_epoch_3d): encoder + gather + decoder + loss, both directions", batch=B,
                           dtype=tr.P.compute_dtype)
    # as-loss engine
    model = DPDistModel(device=dev)
    model.params_.reset_parameters_tf(generator=torch.Generator().manual_seed(1234))
    loss_fn = DPDistLoss(model)
    asl = {"what": "DPDist as a frozen loss through the as-loss engine (dpd_asloss_forward / _backward): loss_pred and d loss / d both clouds",
           "flop_note": "GEMM flops per query point: forward 9.32 M (layers 1-3), backward-to-input the same again (dX of layers 3-1)"}
    for Bx in (16, 32):
        a_, b_, _ = synth.s2_modelnet_shaped(Bx, N, 100)
        src = torch.tensor(a_, device=dev, requires_grad=True)
        tmpl = torch.tensor(b_, device=dev)
        for dt in ("f32", "bf16"):
            model.params_.compute_dtype = dt
            pk = PEAK_FP32_MFMA_TFLOPS if dt == "f32" else PEAK_BF16_MFMA_TFLOPS

            def fb():
                src.grad = None
                loss_fn(src, tmpl).backward()

            def fo():
                with torch.no_grad():
                    loss_fn(src, tmpl)
            Q = 2 * Bx * N
            asl["b%d_%s" % (Bx, dt)] = {"fwd_bwd": record(timed(fb, n), Q, 2 * FWD_FLOP, pk, fb, n),
                                        "fwd_only": record(timed(fo, n), Q, FWD_FLOP, pk, fo, n)}
            # the same forward + backward captured once as a hipGraph and replayed (how the registration step runs it): GPU time per
            # evaluation by events, independent of the host (the eager figure above pays Python + autograd per evaluation, and a busy host
            # shows in it)
            asl["b%d_%s" % (Bx, dt)]["fwd_bwd"]["graph_replay_ms_per_eval"] = graph_replay_ms(loss_fn, src, tmpl, n)
    out["as_loss"] = asl
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="pairs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f32x3", "bf16"],
                    help="compute type of the three wide decoder layers (include/dpdist_capi.h: enum dpd_dtype)")
    ap.add_argument("--no-other-dtypes", action="store_true",
                    help="skip the extra single-GPU timing of the same step in the other compute types")
    ap.add_argument("--prefetch", action="store_true", help="side-stream input pipeline (DPDistTrainer.step(prefetch=...))")
    ap.add_argument("--plan", default="", help="GEMM plan overrides for tuning, e.g. '0:30,4:33:2' = op:tile[:split_k]")
    ap.add_argument("--spinup-ms", type=float, default=40.0, help="device spin-up before the timed region (0 = off: profiler passes)")
    ap.add_argument("--trace", action="store_true", help="diagnostic: per-10-step times of the timed region (adds syncs)")
    ap.add_argument("--cfg4", action="store_true", help="with DPD_FORCE_DIST=1: also run the config-4 legs on one GPU")
    ap.add_argument("--no-live-pmc", action="store_true", help="roofline.traffic from the committed PMC summary instead of two rocprofv3 child passes "
                                                               "(profiler runs of this file: a profiler cannot nest)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if "WORLD_SIZE" not in os.environ and a.gpus > 1:
            raise SystemExit(self_launch(a.gpus))     # plain `python bench.py --gpus N`: spawn the N ranks ourselves
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (a.gpus, world))
    use_dist = world > 1 or os.environ.get("DPD_FORCE_DIST") == "1"   # the latter: exercise the RCCL path on one GPU
    from dpdist_amd import launch
    if use_dist:
        # N > 1: this process becomes the rank's SUPERVISOR (no torch, no GPU runtime) and re-runs this command line as the worker
        # under a per-phase watchdog; a hang or a dead rank -> the workers are stopped by PID and the run is repeated ONCE with
        # torch.distributed collectives (DPD_DP_BACKEND=torch).  Returns only in the worker.
        launch.maybe_supervise(world)
    hb = launch.Heartbeat(rank)
    hb.beat("start:import")
    global torch, dist
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the DPDist path has no CPU fallback")
    # DPD_TEST_SHARE_GPU=1 (tests only): every rank runs on GPU 0 and the process group is gloo -- the numbers mean nothing (the ranks
    # time-share one device, collectives go through the host), but it is the only way to run this file's world > 1 control flow (two
    # supervisors, the reducer's start-up cross-check, the collective schedule choice, the config-4 legs, the replica comparison) on a
    # one-GPU box: tests/test_gpu_parity.py::test_bench_two_ranks_share_the_gpu
    share_gpu = os.environ.get("DPD_TEST_SHARE_GPU") == "1"
    if share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # data-parallel steps: the optimizer runs on the collectives' stream and is joined where the next step first reads the weights
        # (trainer.apply_gradients; this loop never reads params.flat between steps)
        hb.beat("init:process group")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dist.barrier()

    from dpdist_amd import lib, synth
    from dpdist_amd.model import DPDistParams
    from dpdist_amd.trainer import DPDistTrainer
    L = lib.load()

    for item in filter(None, a.plan.split(",")):
        f = [int(x) for x in item.split(":")]
        lib.check(L.dpd_set_gemm_plan(f[0], f[1], f[2] if len(f) > 2 else 1), "dpd_set_gemm_plan")
    B, N = a.batch, 64
    P = DPDistParams(k=5, mlp=(1024, 1024, 1024), device=dev, compute_dtype=a.dtype)
    g = torch.Generator().manual_seed(1234)            # same random-init weights on every rank (replicated variables)
    P.reset_parameters_tf(generator=g)
    hb.beat("reducer:communicators + start-up cross-check")   # make_reducer: librccl bind, ncclCommInitRank, known-pattern reduce
    tr = DPDistTrainer(P, B, num_point=N, Embedding_Size=512, sigma3dmfv=0.125, base_lr=1e-4, adam_on_side=use_dist)
    hb.beat("crosscheck:passed")
    pcA, pcB, lab = synth.s2_modelnet_shaped(B, N, 100 + rank)
    pcA, pcB, lab = (torch.tensor(x, device=dev) for x in (pcA, pcB, lab))

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    headline_sched = None
    if use_dist:
        dist.barrier()
        hb.beat("aux:schedule of the headline step")
        # collective; parameters and Adam state are restored afterwards.  Order x communication form (all bitwise-equivalent except
        # "grouped"); DPD_DP_SCHEDULE / DPD_DP_MODE pin either axis
        tr.progress = lambda: hb.beat(hb.phase)       # every candidate refreshes the watchdog's heartbeat
        headline_sched = tr.select_dp_schedule(pcA, pcB, lab, modes=DP_MODES)

    keep = []
    try:                  # a descheduled launcher thread idles the GPU within a millisecond (shared host): ask for priority
        os.nice(-10)
    except OSError:
        pass
    import gc
    gc.collect()
    gc.disable()          # like timeit: a generation-2 collection inside a 12-30 ms timed region shows up as a 20 ms host stall

    def dp_report(trn, step_fn):
        """What the data-parallel plumbing of trainer `trn` is and costs: backend, ranks as the communicator reports them, bytes
        each GPU puts on the links per step, the start-up cross-check, and the EXPOSED communication = time the compute stream
        spends waiting for collectives, from in-stream event pairs around the waits in a separate pass of the same steps."""
        red = trn.reducer
        if red is None or not red.active:
            return None
        rep = {"backend": red.backend, "fallback": hb.fallback, "attempt": hb.attempt, "mode": red.mode, "wire": red.wire,
               "optimizer_on_collective_stream": bool(trn.adam_on_side),
               "nranks": int(red.nranks),
               "nranks_source": "ncclCommCount" if red.backend == "rccl" else "torch.distributed.get_world_size",
               "wire_bytes_per_gpu_per_step": red.wire_bytes_per_step, "payload_bytes_per_step": 4 * P.numel,
               "crosscheck": red.crosscheck}
        try:
            red.measure = True
            for _ in range(3):
                step_fn()
            red.exposure.collect_ms()
            for _ in range(a.steps):
                step_fn()
            nwaits, ms = red.exposure.collect_ms()
            red.measure = False
            t = torch.tensor([ms], device=dev, dtype=torch.float64)
            if use_dist:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            rep["exposed_comm_us_per_step"] = round(float(t.item()) * 1e3 / a.steps, 2)
            rep["exposed_waits_per_step"] = nwaits / float(a.steps)
            rep["exposed_note"] = ("max over ranks of the summed in-stream time between event pairs around the compute stream's waits for "
                                   "collectives (incl. a collective enqueued on the compute stream itself), separate pass of %d steps" % a.steps)
        except Exception as e:
            rep["exposed_error"] = repr(e)
        try:      # replicated variables must have stayed bit-identical over the ranks (same averaged gradient, same Adam): two checksums per rank
            trn.join_optimizer()
            w = trn.P.flat.detach()
            chk = torch.stack([w.double().sum(), w.double().abs().sum(), trn.m_state.double().sum(), trn.v_state.double().sum()])
            if red.mode == "zero1":
                chk = chk[:2]         # the Adam slots live sharded
            allc = [torch.empty_like(chk) for _ in range(world)] if use_dist else [chk]
            if use_dist:
                torch.cuda.synchronize()
                dist.all_gather(allc, chk)
            rep["replicas_bit_identical"] = bool(all(torch.equal(allc[0], x) for x in allc[1:])) and bool(torch.isfinite(chk).all())
        except Exception as e:
            rep["replicas_check_error"] = repr(e)
        if os.environ.get("DPD_WD_HISTORY"):
            rep["watchdog_history"] = json.loads(os.environ["DPD_WD_HISTORY"])
        return rep

    def bf16_b64(distributed, label, mode=None, env=None):
        """BASELINE configs 3-4: the same training step in bf16 at 64 pairs per GPU, timed like the headline.  env: environment switches
        that hold for the whole leg (the trainer reads DPD_DP_SCHEDULE at every backward)."""
        saved = {k: os.environ.get(k) for k in (env or {})}
        os.environ.update(env or {})
        try:
            return _bf16_b64(distributed, label, mode)
        finally:
            for k, v in saved.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)

    def _bf16_b64(distributed, label, mode=None):
        hb.beat("aux:" + label[:40])
        B2 = 64
        P2 = DPDistParams(k=5, mlp=(1024, 1024, 1024), device=dev, compute_dtype="bf16")
        P2.reset_parameters_tf(generator=torch.Generator().manual_seed(1234))
        old_mode = os.environ.get("DPD_DP_MODE")
        if mode is not None:
            os.environ["DPD_DP_MODE"] = mode
        try:
            tr2 = DPDistTrainer(P2, B2, num_point=N, Embedding_Size=512, sigma3dmfv=0.125, base_lr=1e-4, distributed=distributed,
                                adam_on_side=bool(distributed))
        finally:
            if mode is not None:
                os.environ.pop("DPD_DP_MODE") if old_mode is None else os.environ.__setitem__("DPD_DP_MODE", old_mode)
        a2, b2, l2 = (torch.tensor(x, device=dev) for x in synth.s2_modelnet_shaped(B2, N, 100 + rank))
        sched = None
        if distributed:      # the order of the data-parallel backward: measured here, decided by all ranks together (unless DPD_DP_SCHEDULE pins it)
            hb.beat("aux:schedule " + label[:30])
            tr2.progress = lambda: hb.beat(hb.phase)
            sched = tr2.select_dp_schedule(a2, b2, l2, modes=None if mode is not None else DP_MODES)
        for _ in range(a.warmup):
            tr2.step(a2, b2, l2)
        # Device spin-up, as for the headline: the chip needs ~25 ms of sustained work of THIS kind before its clock
        # settles, and W + K = 25 steps of 0.28 ms are 7 ms.  The auxiliary leg spins up on its own step (more untimed warm-up steps: 40 ms
        # worth on one rank, a fixed 150 when ranks must stay in step); the W + K steps timed BEFORE it are reported as ms_per_step_cold.
        spin2 = a.spinup_ms
        cold2, spin_steps = None, 0
        if spin2 > 0:
            (sync if distributed else torch.cuda.synchronize)()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                tr2.step(a2, b2, l2)
            (sync if distributed else torch.cuda.synchronize)()
            cold2 = (time.perf_counter() - t1) / a.steps * 1e3
            t1 = time.perf_counter()
            # (TEST MODE: N ranks time-share one GPU and reduce through the host -- a step can take 100 ms on a busy box and no timing means
            # anything: no spin-up; the heartbeat is refreshed as the loop advances, so a slow but advancing leg is not taken for a hang)
            while (spin_steps < (0 if share_gpu else 150)) if distributed else ((time.perf_counter() - t1) * 1e3 < spin2):
                for _ in range(10):
                    tr2.step(a2, b2, l2)
                spin_steps += 10
                if distributed:
                    hb.beat(hb.phase)
                if not distributed:
                    torch.cuda.synchronize()
        e2 = float("inf")
        for _rep in range(2):          # auxiliary line: best of two loops (host stalls on a shared box; the headline is single-shot)
            (sync if distributed else torch.cuda.synchronize)()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                tr2.step(a2, b2, l2)
            (sync if distributed else torch.cuda.synchronize)()
            e = time.perf_counter() - t1
            if distributed:
                tt = torch.tensor([e], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                e = float(tt.item())
            e2 = min(e2, e)
        nr = world if distributed else 1
        out2 = {"what": label, "dtype": "bf16", "pairs_per_gpu": B2, "global_batch": B2 * nr, "n_gpus": nr,
                "ms_per_step": round(e2 / a.steps * 1e3, 4), "value": round(2.0 * B2 * N * nr * a.steps / e2, 1),
                "unit": "query-points/sec", "loss_samples_last": round(float(tr2.loss.cpu()[0]), 6),
                "ms_per_step_cold": round(cold2, 4) if cold2 else None, "spinup_steps": spin_steps}
        if distributed:
            out2["dp"] = dp_report(tr2, lambda: tr2.step(a2, b2, l2))
            out2["dp"]["schedule"] = sched
        else:
            try:
                out2["dp_model"] = dp_model(e2 / a.steps * 1e3, "bf16", P2.bucket_bounds)
                out2["dp_model_bf16_wire"] = dp_model(e2 / a.steps * 1e3, "bf16", P2.bucket_bounds, wire="bf16")["per_world"]
            except Exception as e:
                out2["dp_model"] = {"error": repr(e)}
        if not distributed and not a.no_roofline and rank == 0:
            # the plane-GEMM family of THIS configuration against the dense bf16 matrix-core peak (same method as `roofline` below)
            try:
                tr2._load_batch(a2, b2, None)
                got = profiled_gemm_pass(L, tr2, l2, a.warmup, a.steps)
                if got:
                    n_, ms_, _ = got
                    alg2, _ = gemm_flops_per_step(B2, N, 2503, 1024)
                    ach2 = alg2 * a.steps / (ms_ * 1e-3) / 1e12
                    tr3, tsrc3 = pmc_traffic(("gemm_p8_kernel", "gemm_x3_kernel"), "_bf16_b64")
                    out2["roofline"] = {"bound": "mfma", "kernel": "gemm_p8_kernel / gemm_x3_kernel<1,...> (v_mfma_f32_32x32x16_bf16, one bf16 plane)",
                                        "achieved": round(ach2, 2), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                                        "frac": round(ach2 / PEAK_BF16_MFMA_TFLOPS, 4), "traffic": round(tr3) if tr3 else None,
                                        "traffic_note": ("NOT measured in this run: committed PMC pass %s (2 x FETCH_SIZE + WRITE_SIZE per launch, "
                                                         "launch-weighted over the family)" % tsrc3) if tr3 else None,
                                        "algorithmic_bytes_per_launch": round(gemm_bytes_per_step(B2, N, 2528, 1024, "bf16") / (n_ // a.steps)),
                                        "launches_per_step": n_ // a.steps, "avg_launch_us": round(ms_ * 1e3 / n_, 2),
                                        "gemm_ms_per_step": round(ms_ / a.steps, 4),
                                        "whole_step_frac": round(alg2 / (e2 / a.steps) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)}
            except Exception as e:
                out2["roofline"] = {"error": repr(e)}
            try:
                out2["roofline_hbm"] = hbm_roofline_pass(L, lambda: tr2.step(a2, b2, l2), a.warmup, a.steps, "_bf16_b64")
            except Exception as e:
                out2["roofline_hbm"] = {"error": repr(e)}
        keep.append((tr2, P2))     # freed after the headline: a hipFree of ~1 GB idles the GPU for milliseconds (clock ramp)
        return out2

    cfg34 = None
    if not a.no_other_dtypes and a.dtype == "f32" and B == 32:
        try:
            if world > 1 or (use_dist and a.cfg4):   # (--cfg4: exercise this leg on one GPU)
                # every rank takes part in both legs (the first has collectives)
                # the headline leg runs the schedule that WON the start-up measurement (dp.schedule: candidates_ms, MAX over ranks); the
                # pinned legs below are the record of what each candidate does over the full K steps
                c4 = bf16_b64(True, "BASELINE config 4: data-parallel bf16 step, 64 pairs per GPU, RCCL gradient all-reduce, "
                                    "backward order chosen by measurement at start-up")
                try:
                    c4["early_schedule"] = bf16_b64(True, "config 4 with DPD_DP_SCHEDULE=early (separate dW launches, buckets reduced under the backward)",
                                                    mode="allreduce", env={"DPD_DP_SCHEDULE": "early"})
                except Exception as e:
                    c4["early_schedule"] = {"error": repr(e)}
                try:     # the sharded optimizer (ZeRO-1: reduce-scatter -> Adam on 1/P -> all-gather of the parameters), same step
                    c4["zero1"] = bf16_b64(True, "config 4 with DPD_DP_MODE=zero1 (sharded Adam)", mode="zero1")
                except Exception as e:
                    c4["zero1"] = {"error": repr(e)}
                try:     # bf16 on the links: half the bytes, but the cross-rank sum is taken in bf16 -- it changes numerics, so it is never a
                    # candidate of the measured choice: an explicit pinned leg
                    c4["bf16_wire"] = bf16_b64(True, "config 4 with DPD_DP_WIRE=bf16 (gradients rounded to bf16 on the links; pinned leg, changes numerics)",
                                               mode="allreduce", env={"DPD_DP_WIRE": "bf16"})
                except Exception as e:
                    c4["bf16_wire"] = {"error": repr(e)}
                try:     # the single-GPU launch order with ONE grouped weight-gradient launch and ONE all-reduce behind it (DPD_DP_SCHEDULE=grouped)
                    c4["grouped_schedule"] = bf16_b64(True, "config 4 with DPD_DP_SCHEDULE=grouped (one dW launch, one all-reduce)",
                                                      mode="allreduce", env={"DPD_DP_SCHEDULE": "grouped"})
                except Exception as e:
                    c4["grouped_schedule"] = {"error": repr(e)}
                c4["n1_same_run"] = bf16_b64(False, "the same step on one rank without collectives (all ranks run it concurrently)")
                c4["scaling"] = "weak"
                n1v = c4["n1_same_run"]["value"]
                c4["weak_scaling_efficiency_vs_n1_same_run"] = round(c4["value"] / (c4["n_gpus"] * n1v), 4)
                if c4["n_gpus"] == 1:
                    c4["scaling_note"] = "ONE rank: the efficiency above is the cost of the data-parallel plumbing only; nothing crossed a link"
                for leg in ("zero1", "grouped_schedule", "early_schedule", "bf16_wire"):
                    if "value" in c4[leg]:
                        c4[leg]["weak_scaling_efficiency_vs_n1_same_run"] = round(c4[leg]["value"] / (c4["n_gpus"] * n1v), 4)
                try:     # the expectation for real multi-GPU nodes, from this run's one-rank step
                    c4["dp"]["model"] = dp_model(c4["n1_same_run"]["ms_per_step"], "bf16", P.bucket_bounds)
                    c4["dp"]["model_bf16_wire"] = dp_model(c4["n1_same_run"]["ms_per_step"], "bf16", P.bucket_bounds, wire="bf16")["per_world"]
                except Exception as e:
                    c4["dp"]["model"] = {"error": repr(e)}
                cfg34 = ("config4", c4)
            elif not use_dist:
                cfg34 = ("config3", bf16_b64(False, "BASELINE config 3: bf16 training step, 64 pairs"))
        except Exception as e:   # never take the headline number down
            cfg34 = ("config4" if world > 1 else "config3", {"error": repr(e)})

    legs8d = None
    if rank == 0 and world == 1 and not use_dist and not a.no_other_dtypes and B == 32:
        hb.beat("aux:forward-only and as-loss legs")
        try:
            legs8d = section8d_legs(L, tr, pcA, pcB, lab, dev, a.steps, a.warmup)
        except Exception as e:   # never take the headline number down
            legs8d = {"error": repr(e)}
    hb.beat("aux:other compute types")
    others = None
    if rank == 0 and world == 1 and not use_dist and not a.no_other_dtypes:
        # Same step, same batch, same K steps in the other compute types of the decoder GEMMs (include/dpdist_capi.h:
        # enum dpd_dtype).  Reported next to `value`, never as `value`: f32x3 is fp32-equivalent (tests prove it at
        # least as accurate as the exact-fp32 MFMA path), bf16 is the mixed-precision type of BASELINE configs 3-4.
        others = {}
        for dt in ("f32", "f32x3", "bf16"):
            if dt == a.dtype:
                continue
            try:
                P2 = DPDistParams(k=5, mlp=(1024, 1024, 1024), device=dev, compute_dtype=dt)
                P2.reset_parameters_tf(generator=torch.Generator().manual_seed(1234))
                tr2 = DPDistTrainer(P2, B, num_point=N, Embedding_Size=512, sigma3dmfv=0.125, base_lr=1e-4, distributed=False)
                for _ in range(a.warmup):
                    tr2.step(pcA, pcB, lab)
                e2 = float("inf")
                for _rep in range(2):      # best of two: late in a long process the host occasionally stalls a whole loop
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(a.steps):
                        tr2.step(pcA, pcB, lab)
                    torch.cuda.synchronize()
                    e2 = min(e2, time.perf_counter() - t1)
                others[dt] = {"ms_per_step": round(e2 / a.steps * 1e3, 4), "value": round(2.0 * B * N * a.steps / e2, 1),
                              "loss_samples_last": round(float(tr2.loss.cpu()[0]), 6)}
                del tr2, P2
            except Exception as e:   # never take the headline number down
                others[dt] = {"error": repr(e)}
    # ORDER: the config-3/4 legs and the other compute types run BEFORE the headline region, the profiler pass right after it.  An idle
    # MI355X needs ~40 training steps (25 ms) to reach its steady clock (measured, tools/ramp_probe.py: 0.65 -> 0.58 ms per step
    # over the first 40 steps, again after 2 s of idle); with the driver's `--steps 20 --warmup 5` the headline would otherwise be
    # timed on a chip that is still ramping.  The headline itself is exactly W untimed + K timed steps, as the contract says.
    # --prefetch: every step also runs the NEXT batch's encoder + gather on a side stream (each timed step still executes
    # exactly one front end; the round-1 version of this pipeline silently ran it twice).  Re-measured in round 2:
    # with the LDS-ring GEMMs (144 of 160 KiB of LDS per CU) nothing overlapped (0.672 vs 0.674 ms); with the LDS-free
    # register-streamed GEMMs 0.598-0.605 vs 0.610-0.611 ms (1-2 %).  Off by default: it needs the next batch one step early.
    nxt = (pcA, pcB, None) if a.prefetch else None
    # Device spin-up (NOT training steps, nothing of the model is touched): an MI355X that was idle -- or busy with a different
    # kind of load -- needs ~25 ms of sustained fp32-MFMA work before its power management settles on the steady clock
    # (tools/ramp_probe.py, --trace: 0.645 -> 0.58 ms per step over the first 40 steps, every time).  The driver's
    # `--steps 20 --warmup 5` is a 15 ms measurement; without this it times the ramp, not the step.  Reported as `spinup_ms`.
    spin_ms = a.spinup_ms
    hb.beat("warmup:cold pass + spin-up")
    el_cold = None
    if spin_ms > 0:
        # the number WITHOUT the spin-up, reported next to the headline as ms_per_step_cold: the same W untimed + K timed steps, run
        # first (the chip comes out of the auxiliary legs above or out of idle: this times the clock ramp, see below)
        for _ in range(a.warmup):
            tr.step(pcA, pcB, lab, prefetch=nxt)
        sync()
        t0c = time.perf_counter()
        for i in range(a.steps):
            tr.step(pcA, pcB, lab, prefetch=nxt)
        sync()
        el_cold = time.perf_counter() - t0c
        if use_dist:
            tc = torch.tensor([el_cold], device=dev, dtype=torch.float64)
            dist.all_reduce(tc, op=dist.ReduceOp.MAX)
            el_cold = float(tc.item())
    if spin_ms > 0:
        from dpdist_amd import ops
        sa, sb = torch.randn(4096, 2528, device=dev), torch.randn(2528, 1024, device=dev)
        torch.cuda.synchronize()
        t_spin = time.perf_counter()
        while (time.perf_counter() - t_spin) * 1e3 < spin_ms:
            for _ in range(8):
                ops.gemm_f32(sa, sb, tile=32)
            torch.cuda.synchronize()
    for _ in range(a.warmup):
        tr.step(pcA, pcB, lab, prefetch=nxt)
    hb.beat("timed:%d steps" % a.steps)
    sync()
    trace = a.trace
    marks = []
    t0 = time.perf_counter()
    for i in range(a.steps):
        tr.step(pcA, pcB, lab, prefetch=nxt)
        if trace and i % 10 == 9:
            torch.cuda.synchronize()
            marks.append(time.perf_counter())
    sync()
    el = time.perf_counter() - t0
    if trace and rank == 0:
        print("trace ms/step per 10 steps:", " ".join("%.3f" % ((b - a_) / 10 * 1e3) for a_, b in zip([t0] + marks[:-1], marks)), file=sys.stderr)
    if use_dist:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    loss = tr.loss.cpu().numpy()
    hb.beat("profile:roofline + exposed communication")
    dp = dp_report(tr, lambda: tr.step(pcA, pcB, lab)) if use_dist else None
    if dp is not None:
        dp["schedule"] = headline_sched

    roof = None
    if a.no_roofline:
        gc.enable()
    else:
        # Separate pass of the same K steps (forward + backward, no Adam) with the library's in-stream profiler on.
        # EVERY rank runs it so that the gradient collectives stay matched; only rank 0 records and reports.
        # Two passes, the one with the smaller GEMM time is reported: a host stall (shared box) idles the GPU, its clock drops
        # and the kernels of the next ~25 ms run slower -- seen once as frac 0.765 next to an unaffected headline.
        tr._load_batch(pcA, pcB, None)
        best = None
        best_nn = (0, 0.0)
        for _pass in range(2):
            for it in range(a.warmup + a.steps):   # W unprofiled iterations (first launches load code objects, create events)
                if it == a.warmup:
                    torch.cuda.synchronize()
                    if rank == 0:
                        L.dpd_prof_enable(1)
                tr.forward()
                tr.backward(lab.reshape(-1))
                if tr.reducer:
                    tr.reducer.wait()
            torch.cuda.synchronize()
            if rank == 0:
                ms_, fl_ = ctypes.c_double(0), ctypes.c_double(0)
                n_ = L.dpd_prof_collect(ctypes.byref(ms_), ctypes.byref(fl_))
                ms0, fl0 = ctypes.c_double(0), ctypes.c_double(0)
                n0 = L.dpd_prof_collect_form(0, ctypes.byref(ms0), ctypes.byref(fl0))      # NN / NT launches: forward layers + data gradients
                L.dpd_prof_enable(0)
                if n_ > 0 and ms_.value > 0 and (best is None or ms_.value < best[1].value):
                    best = (n_, ms_, fl_)
                    best_nn = (n0, ms0.value)
        gc.enable()
        if rank == 0:
            n, ms, fl = best if best else (0, ctypes.c_double(0), ctypes.c_double(0))
            alg, per_step = gemm_flops_per_step(B, N, 2503, 1024)
            if n > 0 and ms.value > 0:
                launches = n
                ach = alg * a.steps / (ms.value * 1e-3) / 1e12
                # executed matrix-core flops per algorithmic flop: 1 on the fp32 MFMA, 6 bf16 terms in the split form
                mult, peak, kern = {"f32": (1, PEAK_FP32_MFMA_TFLOPS, "gemm_rs_kernel<...> (fp32 v_mfma_f32_32x32x2, register-streamed operands, no LDS / barriers)"),
                                    "f32x3": (6, PEAK_BF16_MFMA_TFLOPS, "gemm_x3_kernel<3,...> (6 x v_mfma_f32_32x32x16_bf16 per product, LDS-DMA ring)"),
                                    "bf16": (1, PEAK_BF16_MFMA_TFLOPS, "gemm_x3_kernel<1,...> (v_mfma_f32_32x32x16_bf16, LDS-DMA ring)")}[a.dtype]
                sfx = {"f32": "", "f32x3": "_f32x3", "bf16": "_bf16"}[a.dtype] if B == 32 else ("_bf16_b64" if (a.dtype, B) == ("bf16", 64) else "_none")
                fam = ("gemm_rs_kernel",) if a.dtype == "f32" else ("gemm_p8_kernel", "gemm_x3_kernel")
                traffic, tsrc = pmc_traffic(fam, sfx)
                committed = traffic
                live_note = None
                if not a.no_live_pmc and world == 1 and not use_dist:
                    # the headline's own PMC passes, now: the timed region is over, the GPU is ours (the children allocate their own trainer)
                    hb.beat("profile:live PMC passes")
                    torch.cuda.synchronize()
                    t_pmc = time.perf_counter()
                    live, live_note = pmc_traffic_live(fam, ["--dtype", a.dtype, "--batch", str(B)] + (["--plan", a.plan] if a.plan else []))
                    if live:
                        traffic = live
                    live_s = round(time.perf_counter() - t_pmc, 1)
                roof = {"bound": "mfma", "kernel": kern,
                        "achieved": round(ach * mult, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(ach * mult / peak, 4),
                        "traffic": round(traffic) if traffic else None,
                        "traffic_note": (live_note if (live_note and traffic is not committed) else
                                         ("NOT measured in this run%s: read from the COMMITTED PMC pass %s (separate rocprofv3 --pmc FETCH_SIZE / "
                                          "WRITE_SIZE runs of this command); bytes per launch at the L2's fabric side (2 x FETCH_SIZE + "
                                          "WRITE_SIZE, Infinity-Cache hits included), launch-weighted over the family"
                                          % (" (%s)" % live_note if live_note else "", tsrc)) if traffic else
                                         "PMC passes are separate rocprofv3 runs: profiles/"),
                        "algorithmic_bytes_per_launch": round(gemm_bytes_per_step(B, N, 2528, 1024, a.dtype) / (launches / a.steps)),
                        "algorithmic_tflops": round(ach, 2),
                        "launches_per_step": launches // a.steps, "avg_launch_us": round(ms.value * 1e3 / launches, 2),
                        "alg_gflop_per_launch": round(alg / (launches / a.steps) / 1e9, 3),
                        "gemm_ms_per_step": round(ms.value / a.steps, 4)}
                if live_note is not None:
                    roof["traffic_committed_summary"] = round(committed) if committed else None
                    roof["live_pmc_seconds"] = live_s
                # the DOMINANT kernel alone (the family above also holds the weight-gradient kernel): the forward layers and the data
                # gradients are one kernel in f32 (gemm_rs_kernel<true, false, ...>, NN products on transposed weight copies)
                n0, ms0 = best_nn
                if n0 > 0 and ms0 > 0:
                    Qr, BNr = 2 * B * N, B * N
                    alg_nn = 2.0 * Qr * (2503 * 1024 + 2 * 1024 * 1024) + 2.0 * BNr * (2 * 1024 * 1024)
                    ach_nn = alg_nn * a.steps / (ms0 * 1e-3) / 1e12
                    roof["dominant_kernel_frac"] = round(ach_nn * mult / peak, 4)
                    roof["dominant_kernel"] = {"what": "forward layers 1-3 + data gradients g3->g2->g1 (NN / NT products)",
                                               "launches_per_step": n0 // a.steps, "avg_launch_us": round(ms0 * 1e3 / n0, 2),
                                               "achieved": round(ach_nn * mult, 2), "unit": "TFLOP/s"}
    roof_hbm = None
    if not a.no_roofline and rank == 0 and not use_dist:
        try:
            sfx = {"f32": "", "f32x3": "_f32x3", "bf16": "_bf16"}[a.dtype] if B == 32 else ("_bf16_b64" if (a.dtype, B) == ("bf16", 64) else "_none")
            roof_hbm = hbm_roofline_pass(L, lambda: tr.step(pcA, pcB, lab), a.warmup, a.steps, sfx)
        except Exception as e:
            roof_hbm = {"error": repr(e)}
    hb.beat("report:cpu baseline + json")
    if rank == 0:
        qps = 2.0 * B * N * world * a.steps / el
        out = {"metric": "query-points/sec (DPDist fwd+bwd)", "value": round(qps, 1), "unit": "query-points/sec",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(el / a.steps * 1e3, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
               "data": "synthetic" if not share_gpu else "synthetic; TEST MODE DPD_TEST_SHARE_GPU=1: all ranks on ONE GPU over gloo, timings meaningless",
               "config": {"workload": "DPDist training step (3DmFV 8^3 + 5^3-window decoder 2503-1024-1024-1024-3, "
                                      "fwd both directions + bwd AB half + Adam), S2 ModelNet-shaped clouds",
                          "pairs_per_gpu": B, "global_batch": B * world, "num_point": N, "query_points_per_step": 2 * B * N * world,
                          "parallelism": "dp%d" % world, "loss_samples_last": round(float(loss[0]), 6)},
               "roofline": roof, "roofline_hbm": roof_hbm, "spinup_ms": spin_ms,
               "ms_per_step_cold": round(el_cold / a.steps * 1e3, 4) if el_cold else None,
               "cold_note": "ms_per_step_cold = the same W + K steps timed BEFORE the %g ms device spin-up (scratch fp32 GEMMs, nothing of the "
                            "model) that precedes the headline region; the difference is the GPU's clock ramp (DESIGN.md section 5)" % spin_ms}
        if dp is not None:
            out["dp"] = dp
            out["dp_backend"], out["fallback"] = dp["backend"], dp["fallback"]
        try:        # one rank: from this run's step; N ranks: from the one-rank step the plumbing cost is known for (the model's own input)
            step_n1 = el / a.steps * 1e3 - (DP_WINDOWS[a.dtype][2] * 1e-3 if dp is not None else 0.0)
            (out["dp"] if dp is not None else out)["model" if dp is not None else "dp_model"] = dp_model(step_n1, a.dtype, P.bucket_bounds)
        except Exception as e:
            out["dp_model"] = {"error": repr(e)}
        if others is not None:
            out["other_compute_types"] = others
        if legs8d is not None:
            out.update(legs8d) if "error" not in legs8d else out.__setitem__("fwd_only", legs8d)
        if cfg34:
            out[cfg34[0]] = cfg34[1]
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(B, N)
            except Exception as e:   # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"error": repr(e)}
    if rank == 0:
        # RCCL prints its version banner through C stdio (flushed at exit = after anything Python printed): flush it first so that
        # the JSON line is the LAST line of stdout.  The line goes out BEFORE the tear-down: a rank that hangs in ncclCommDestroy
        # must not take the measurement with it (the supervisor treats phase "done" as success).
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    hb.beat("done")
    if use_dist:
        torch.cuda.synchronize()
        for t_ in [tr] + [k[0] for k in keep]:      # communicators of the direct RCCL reducers go before the process group does
            t_.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
