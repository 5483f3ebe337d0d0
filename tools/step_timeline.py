"""Per-launch timeline of one training step from a rocprofv3 kernel trace: launches in stream order with their average duration
and the average idle gap before each (end of the previous kernel -> start of this one).
    rocprofv3 --kernel-trace --output-format csv -d DIR -o p -- python bench.py ...;  python tools/step_timeline.py DIR/.../p_kernel_trace.csv LAUNCHES_PER_STEP [skip_steps]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("dpd::", "")
    m = re.match(r"([A-Za-z0-9_]+)(<[^(]*>)?", name)
    tmpl = (m.group(2) or "")
    tmpl = re.sub(r"\s+", "", tmpl)
    return (m.group(1) + tmpl)[:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    end_kernel = sys.argv[2] if len(sys.argv) > 2 else "adam"
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    # a step = the launches up to and including the optimizer kernel; keep the steps of the most common shape
    steps, cur = [], []
    for r, nm in zip(rows, names):
        cur.append((r, nm))
        if end_kernel in nm:
            steps.append(cur)
            cur = []
    from collections import Counter
    sig = Counter(tuple(nm for _, nm in st) for st in steps).most_common(1)[0][0]
    good = [i for i, st in enumerate(steps) if tuple(nm for _, nm in st) == sig and i > 0 and tuple(nm for _, nm in steps[i - 1]) == sig]
    good = good[len(good) // 3:]                      # drop the first third (clock ramp)
    n = len(sig)
    print("%d launches in the trace; %d per step; averaging %d steady steps" % (len(rows), n, len(good)))
    tot_d = tot_g = 0.0
    for i in range(n):
        d, g = [], []
        for s in good:
            r = steps[s][i][0]
            prev = steps[s][i - 1][0] if i > 0 else steps[s - 1][-1][0]
            d.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            g.append(int(r["Start_Timestamp"]) - int(prev["End_Timestamp"]))
        ad, ag = sum(d) / len(d) / 1e3, sum(g) / len(g) / 1e3
        tot_d += ad
        tot_g += ag
        print("%2d %-70s %7.2f us  gap %6.2f us" % (i, sig[i], ad, ag))
    print("   kernels %.1f us + gaps %.1f us = %.1f us per step" % (tot_d, tot_g, tot_d + tot_g))


if __name__ == "__main__":
    main()
