#!/bin/bash
# usage: tools/pmc_pass.sh <outdir-tag> "<counters>" -- <bench args...>     (run on the GPU box through gpurun)
# One rocprofv3 --pmc pass (counters in their own run: --kernel-trace only) of a short bench.py; prints per-kernel means.
TAG=$1; CNT=$2; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-other-dtypes "$@" > /dev/null 2>&1
python - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$OUT/p_counter_collection.csv")):
    if "gemm" in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in acc.items():
    print(k, {n: "%.3g" % (sum(v)/len(v)) for n,v in c.items()}, "n=%d" % len(next(iter(c.values()))))
PY
