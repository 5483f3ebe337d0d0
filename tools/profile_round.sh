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
# compute types on the bf16 matrix cores (same step, --dtype) and the B=64 configuration of BASELINE configs 3-4
for dt in f32x3 bf16; do
  python $R/bench.py --dtype $dt --no-cpu-baseline > $OUT/bench_$dt.json 2>> $OUT/bench.err
done
python $R/bench.py --dtype bf16 --batch 64 --no-cpu-baseline > $OUT/bench_bf16_b64.json 2>> $OUT/bench.err
python $R/bench.py --batch 64 --no-cpu-baseline > $OUT/bench_f32_b64.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_f32x3 -o $TAG -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype f32x3 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bf16 -o $TAG -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype bf16 > /dev/null 2>&1
python $R/tools/x3_bench.py 2 3 5 > $OUT/x3_bench.txt 2>&1
# PMC passes for the plane compute types (same separate-pass rule)
for dt in f32x3 bf16; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${dt}_$c -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-other-dtypes --dtype $dt > /dev/null 2>&1
  done
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_${dt}_MFMA -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-other-dtypes --dtype $dt > /dev/null 2>&1
done
