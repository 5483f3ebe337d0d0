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


for a, b in (("bench.json", "bench.json"), ("bench_f32x3.json", "bench_f32x3.json"), ("bench_bf16.json", "bench_bf16.json"),
             ("bench_bf16_b64.json", "bench_bf16_b64.json"), ("bench_f32_b64.json", "bench_f32_b64.json"),
             ("x3_bench.txt", "x3_bench.txt"), ("gemm_bench.txt", "gemm_bench.txt"), ("variants.txt", "variants.txt"),
             ("determinism.txt", "determinism.txt"), ("bench_driver_flags.json", "bench_driver_flags.json"),
             ("asloss_bench.txt", "asloss_bench.txt"), ("ramp_probe.txt", "ramp_probe.txt"), ("host_rate.txt", "host_rate.txt"),
             ("ldsdma_bw.txt", "ldsdma_bw.txt"), ("dp_single_rank.txt", "dp_single_rank.txt"), ("step_timeline.txt", "step_timeline.txt"),
             ("x3_bench_p8.txt", "x3_bench_p8.txt"), ("gather_bench.txt", "gather_bench.txt"), ("event_cost.txt", "event_cost.txt"),
             ("launch_floor.txt", "launch_floor.txt"), ("bf16_trio_sweep.txt", "bf16_trio_sweep.txt"), ("bf16_bwd_sweep.txt", "bf16_bwd_sweep.txt"),
             ("bf16_plan_sweep.txt", "bf16_plan_sweep.txt"), ("trio_other_configs.txt", "trio_other_configs.txt"), ("xcd_band_ab.txt", "xcd_band_ab.txt"),
             ("overlap_probe_bf16.txt", "overlap_probe_bf16.txt"), ("watchdog_fallback.json", "watchdog_fallback.json"),
             ("watchdog_fallback.err", "watchdog_fallback.err"), ("bench_forced_dist_cfg4.json", "bench_forced_dist_cfg4.json"),
             ("registration_demo.txt", "registration_demo.txt"), ("mfv_stamps.txt", "mfv_stamps.txt"), ("chain_bench.txt", "chain_bench.txt"),
             ("asloss_engine_ab.txt", "asloss_engine_ab.txt"), ("registration_engine_ab.txt", "registration_engine_ab.txt")):
    copy(a, b)

# per-kernel stats (our kernels only), one file per compute type
for sub, out in (("stats", "kernel_stats.csv"), ("stats_f32x3", "kernel_stats_f32x3.csv"), ("stats_bf16", "kernel_stats_bf16.csv"),
                 ("stats_bf16_b64", "kernel_stats_bf16_b64.csv")):
    f = os.path.join(src, sub, "%s_kernel_stats.csv" % tag)
    if not os.path.exists(f):
        continue
    rows = [r for r in csv.DictReader(open(f)) if mine(r["Name"])]
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
pmc_summary("pmc_f32x3_", "stats_f32x3", "pmc_summary_f32x3.csv")
pmc_summary("pmc_bf16_", "stats_bf16", "pmc_summary_bf16.csv")
pmc_summary("pmc_bf16_b64_", "stats_bf16_b64", "pmc_summary_bf16_b64.csv")
print("wrote", sorted(x for x in os.listdir(dst) if x.startswith(tag)))
