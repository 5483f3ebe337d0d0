#!/bin/bash
# rocprofv3 kernel stats of the registration step (config 5, B=16, 8 loops), eager form of the default path (the graph replays the same
# kernels).  usage (GPU box): tools/registration_profile.sh <outdir> [dtype]
OUT=${1:-gpurun_out/regprof}; DT=${2:-f32}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT -o reg -- python $R/tools/registration_step_bench.py --forms eager_native --steps 50 --dtype $DT > $R/$OUT/bench.txt 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/$OUT/reg_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel,calls,total_us,avg_us,percent")
for r in rows[:40]:
    print("%s,%s,%.1f,%.2f,%.2f" % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
