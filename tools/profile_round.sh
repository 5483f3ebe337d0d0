#!/bin/bash
# Regenerate the per-round evidence under gpurun_out/<tag>/ (run on the GPU box through gpurun):
#   bench.json                     python bench.py                       (value, roofline, cpu_baseline)
#   stats/*_kernel_stats.csv       rocprofv3 --kernel-trace --stats      (per-kernel average durations)
#   pmc_FETCH_SIZE / pmc_WRITE_SIZE  separate --pmc passes               (HBM/fabric bytes per launch)
# usage: tools/profile_round.sh r01
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_MFMA -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
cat $OUT/bench.json
