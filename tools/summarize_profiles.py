"""Condense gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU box) into the tracked profiles/<tag>_* files.

    python tools/summarize_profiles.py r01
"""
import csv
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")


def mine(name):
    return name.startswith("dpd::") or name.startswith("void dpd::")


def copy(a, b):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, "%s_%s" % (tag, b)))


# every top-level text / json artefact of the round, as written
if os.path.isdir(src):
    for name in sorted(os.listdir(src)):
        if os.path.isfile(os.path.join(src, name)) and name.endswith((".json", ".txt", ".err")) and not name.endswith(".out"):
            copy(name, name)

# per-kernel stats (our kernels only), one file per compute type
raw = os.path.join(dst, "raw")
os.makedirs(raw, exist_ok=True)
subs = sorted(d for d in (os.listdir(src) if os.path.isdir(src) else []) if d.startswith("stats") and os.path.isdir(os.path.join(src, d)))
for sub in subs:
    out = "kernel_stats%s.csv" % sub[len("stats"):]
    # the raw rocprofv3 stats (every kernel, torch's and RCCL's included) and the first 200 launches of the trace, so that the condensed
    # file below can be re-derived without gpurun_out/
    for kind, keep in (("kernel_stats", None), ("kernel_trace", 201)):
        f0 = os.path.join(src, sub, "%s_%s.csv" % (tag, kind))
        if os.path.exists(f0):
            with open(f0) as fi, open(os.path.join(raw, "%s_%s_%s.csv" % (tag, sub, kind)), "w") as fo:
                for n, line in enumerate(fi):
                    if keep is not None and n >= keep:
                        break
                    fo.write(line)
    f = os.path.join(src, sub, "%s_kernel_stats.csv" % tag)
    if not os.path.exists(f):
        continue
    # (the registration step also runs torch / hipBLASLt kernels -- the pose network's backward: every kernel is listed there)
    rows = [r for r in csv.DictReader(open(f)) if mine(r["Name"]) or "registration" in sub]
    with open(os.path.join(dst, "%s_%s" % (tag, out)), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "percent"])
        for r in rows:
            w.writerow([r["Name"], r["Calls"], "%.1f" % (float(r["TotalDurationNs"]) / 1e3), "%.2f" % (float(r["AverageNs"]) / 1e3),
                        "%.2f" % (float(r["MinNs"]) / 1e3), "%.2f" % (float(r["MaxNs"]) / 1e3), r["Percentage"]])

# PMC summaries: FETCH_SIZE / WRITE_SIZE (KiB per launch) and MFMA-busy fraction per kernel, one file per compute type
def pmc_summary(prefix, stats_sub, out_name):
    acc = defaultdict(lambda: defaultdict(list))
    for cname in ("FETCH_SIZE", "WRITE_SIZE", "MFMA"):
        f = os.path.join(src, prefix + cname, "p_counter_collection.csv")
        if not os.path.exists(f):
            continue
        for r in csv.DictReader(open(f)):
            if mine(r["Kernel_Name"]):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not acc:
        return
    stats = {}
    f = os.path.join(src, stats_sub, "%s_kernel_stats.csv" % tag)
    if os.path.exists(f):
        stats = {r["Name"]: r for r in csv.DictReader(open(f))}
    with open(os.path.join(dst, "%s_%s" % (tag, out_name)), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "calls_in_stats", "avg_us", "FETCH_SIZE_KB_per_launch(raw)", "fabric_read_MB_per_launch(x2 gfx950 correction)",
                    "WRITE_SIZE_KB_per_launch", "MFMA_busy_frac(under PMC collection)"])
        mean = lambda v: sum(v) / len(v) if v else None   # noqa: E731
        for k, c in sorted(acc.items(), key=lambda kv: -float(stats.get(kv[0], {}).get("TotalDurationNs", 0))):
            fe, wr = mean(c.get("FETCH_SIZE", [])), mean(c.get("WRITE_SIZE", []))
            # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
            busy, act = mean(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [])), mean(c.get("GRBM_GUI_ACTIVE", []))
            act = act / 8.0 * 1024.0 if act else act
            st = stats.get(k, {})
            w.writerow([k, st.get("Calls", ""), "%.2f" % (float(st["AverageNs"]) / 1e3) if st else "",
                        "%.0f" % fe if fe is not None else "", "%.1f" % (2 * fe * 1024 / 1e6) if fe is not None else "",
                        "%.0f" % wr if wr is not None else "", "%.3f" % (busy / act) if busy and act else ""])


pmc_summary("pmc_", "stats", "pmc_summary.csv")
pmc_summary("pmc_bf16_b64_", "stats_bf16_b64", "pmc_summary_bf16_b64.csv")

# config 5: where a registration step's GPU time goes (eager form of the default path = what the captured graph replays), per training step
f = os.path.join(src, "stats_registration", "%s_kernel_stats.csv" % tag)
if os.path.exists(f):
    steps = 8 + 3 * 50      # tools/registration_step_bench.py --steps 50 --train-only
    groups = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "dpd::pose_" in n and ("_bwd" in n or "pose_fc_dw" in n or "pose_fc_dx" in n):
            g = "pose network + pose chain, backward of the training evaluation (csrc/pose.hip)"
        elif "dpd::pose_" in n:
            g = "pose network + pose chain, forward: 7 refinements + the training evaluation (csrc/pose.hip)"
        elif "dpd::adam" in n:
            g = "TF-form Adam (dpd_adam_tf)"
        elif "dpd::" in n:
            g = "DPDist forward + backward (as-loss engine)"
        elif n.startswith("Cijk_"):
            g = "torch GEMMs (hipBLASLt)"
        else:
            g = "torch kernels (the dropout mask draw)"
        groups[g][0] += float(r["TotalDurationNs"]) / 1e3
        groups[g][1] += int(r["Calls"])
    tot = sum(v[0] for v in groups.values())
    with open(os.path.join(dst, "%s_registration_breakdown.txt" % tag), "w") as fh:
        fh.write("# kernel time per registration TRAINING step (B=16, 64 points, 8 loops; %d steps under rocprofv3, eager form of the default path)\n" % steps)
        fh.write("# %-75s %10s %10s %8s\n" % ("group", "us/step", "launches", "share"))
        for g, (us, calls) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
            fh.write("%-77s %10.1f %10.1f %7.1f%%\n" % (g, us / steps, calls / steps, 100 * us / tot))
        fh.write("%-77s %10.1f\n" % ("total kernel time per step", tot / steps))
        fh.write("# wall clock per step: see %s_registration_step_bench.txt (eager_native; the captured graph replays these kernels)\n" % tag)
print("wrote", sorted(x for x in os.listdir(dst) if x.startswith(tag)))
